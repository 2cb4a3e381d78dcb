#!/bin/bash
# Everything profiles/ holds for a round, collected on ONE GPU box — in two calls, each within a gpurun limit:
#   bash tools/collect_round.sh gpurun_out/prof_rNN a     kernel-trace stats, PMC passes, the bench lines
#   bash tools/collect_round.sh gpurun_out/prof_rNN b     other configs, config 5, partition / in-flight times, phases, events
set -e
OUT=$(realpath -m "$1"); PART=${2:-ab}; REPO=$(pwd); mkdir -p "$OUT"
if [[ $PART == *a* ]]; then
  bash tools/profile_round.sh "$OUT" > "$OUT/profile_round.log" 2>&1
  echo "profile_round done"
  python3 tools/trace_union.py "$OUT/stats" > "$OUT/trace_union.txt"
  python3 tools/trace_union.py "$OUT/stats_two_in_flight" >> "$OUT/trace_union.txt"
  cat "$OUT/trace_union.txt"
  cp "$OUT/traffic.json" profiles/traffic_r03.json   # so that the lines below quote the counters of THESE sources
  python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2> /dev/null
  python bench.py > "$OUT/bench_default.json" 2> /dev/null
  python bench.py --batch-inflight 2 --no-cpu-baseline > "$OUT/bench_two_in_flight.json" 2> /dev/null
  python bench.py --dirmask 0 --no-cpu-baseline > "$OUT/bench_dirmask_off.json" 2> /dev/null
  echo "bench lines done"
fi
if [[ $PART == *b* ]]; then
  bash tools/other_configs.sh "$OUT/other_configs_bench.jsonl" 2> /dev/null
  echo "other configs done"
  python tools/config5.py 2>&1 | grep -v amdgpu.ids > "$OUT/config5_one_gpu.txt" || true
  python tools/batch_times.py 2>&1 | grep -v amdgpu.ids > "$OUT/batch_and_partition_times.txt" || true
  python tools/inflight_times.py 2>&1 | grep -v amdgpu.ids > "$OUT/inflight_times.txt" || true
  echo "times done"
  RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 8 2>&1 | grep -v amdgpu.ids > "$OUT/phases.txt"
  RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 1 2>&1 | grep -v amdgpu.ids >> "$OUT/phases.txt"
  RT355_LIB=$REPO/rsraytracingv3_amd/librt355_events.so python tools/phase_profile.py --batch 8 2>&1 | grep "direction-mask scans\|events of that launch" >> "$OUT/phases.txt"
  cat "$OUT/phases.txt"
  bash tools/dirmask_sweep.sh > "$OUT/dirmask_sweep.txt" 2>&1
  python tools/set_scene_time.py 2>&1 | grep -v amdgpu.ids > "$OUT/set_scene_time.txt"
fi
