#!/bin/bash
# Everything profiles/ holds for round 4, collected on ONE GPU box in calls that each fit a gpurun limit:
#   bash tools/collect_round4.sh gpurun_out/prof_r04 a   headline: kernel-trace stats (default = two launches in flight; one at a time;
#                                                       all legs), six PMC passes -> traffic_r04.json, the bench lines
#   bash tools/collect_round4.sh gpurun_out/prof_r04 c   config 3 (1 024 spheres): kernel-trace stats + PMC passes -> traffic_r04_config3.json
#   bash tools/collect_round4.sh gpurun_out/prof_r04 d   the 4K frame, config 5's frame with the 32 MiB and the 512 MiB skybox (fetched against
#                                                       algorithmic bytes), the single-frame kernel: kernel-trace stats + PMC passes
#   bash tools/collect_round4.sh gpurun_out/prof_r04 b   other configs, config 5's accumulation, partition / in-flight / multi / presentation
#                                                       times, phase and event profiles, sweeps
# PMC passes run one launch at a time (--batch-inflight 1): per-dispatch counters of kernels that share the device would count each
# other's cycles.  Never combined with trace domains.
set -e
OUT=$(realpath -m "$1"); PART=${2:-a}; REPO=$(pwd); mkdir -p "$OUT"
PMC_SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES"
          "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")

trace() {  # trace NAME bench-flags...: kernel-trace stats of one bench run -> $OUT/NAME/ + $OUT/NAME.json
  local name=$1; shift
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o bench -- python3 "$REPO/bench.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.log")
  echo "trace $name done"
}
pmc() {  # pmc NAME first-set last-set bench-flags...: PMC passes -> $OUT/NAME_pmcK/
  local name=$1 first=$2 last=$3; shift 3
  for ((i = first; i <= last; i++)); do
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc ${PMC_SETS[$i]} --output-format csv -d "$OUT/${name}_pmc$i" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra-legs \
       --batch-inflight 1 --steps 48 --warmup 16 --min-seconds 0.05 --clock-warmup 0.2 "$@" > "$OUT/${name}_pmc$i.json" 2> "$OUT/${name}_pmc$i.log")
    echo "pmc $name pass $i (${PMC_SETS[$i]}) done"
  done
}
line() { python bench.py "$@" 2> /dev/null; }

if [[ $PART == *a* ]]; then
  trace stats --no-extra-legs
  trace stats_one_at_a_time --no-extra-legs --no-cpu-baseline --batch-inflight 1
  trace stats_all_legs --no-cpu-baseline
  pmc headline 0 5
  python3 tools/pmc_summary.py "$OUT"/headline_pmc[0-9] --json "$OUT/traffic_r04.json" > "$OUT/pmc_summary.txt"
  cat "$OUT/pmc_summary.txt"
  python3 tools/trace_union.py "$OUT/stats" > "$OUT/trace_union.txt"
  python3 tools/trace_union.py "$OUT/stats_one_at_a_time" >> "$OUT/trace_union.txt"
  cat "$OUT/trace_union.txt"
  cp "$OUT/traffic_r04.json" profiles/traffic_r04.json   # so that the lines below quote the counters of THESE sources
  line --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json"
  line > "$OUT/bench_default.json"
  line --batch-inflight 1 --no-cpu-baseline > "$OUT/bench_one_at_a_time.json"
  line --dirmask 0 --no-cpu-baseline > "$OUT/bench_dirmask_off.json"
  echo "bench lines done"
fi
if [[ $PART == *c* ]]; then
  trace config3_stats --no-extra-legs --no-cpu-baseline --spheres 1024
  trace config3_stats_one_at_a_time --no-extra-legs --no-cpu-baseline --spheres 1024 --batch-inflight 1
  pmc config3 0 5 --spheres 1024
  python3 tools/pmc_summary.py "$OUT"/config3_pmc[0-9] --objects 1025 --json "$OUT/traffic_r04_config3.json" > "$OUT/config3_pmc_summary.txt"
  cat "$OUT/config3_pmc_summary.txt"
  cp "$OUT/traffic_r04_config3.json" profiles/traffic_r04_config3.json
  line --spheres 1024 --no-cpu-baseline > "$OUT/bench_config3.json"
  line --spheres 1024 --no-cpu-baseline --dirlist 0 > "$OUT/bench_config3_grid.json"
  for n in 256 512; do line --spheres $n --no-cpu-baseline --no-extra-legs; line --spheres $n --no-cpu-baseline --no-extra-legs --dirlist 0; done > "$OUT/bench_256_512_lists_then_grid.jsonl"
  echo "config 3 done"
fi
if [[ $PART == *d* ]]; then
  trace k4_stats --no-extra-legs --no-cpu-baseline --width 3840 --height 2160
  pmc k4 0 1 --width 3840 --height 2160
  pmc k4 3 5 --width 3840 --height 2160
  python3 tools/pmc_summary.py "$OUT"/k4_pmc[0-9] --width 3840 --height 2160 --json "$OUT/traffic_r04_4k.json" > "$OUT/k4_pmc_summary.txt"
  for sky in 2048 8192; do
    pmc c5_$sky 0 1 --width 3840 --height 2160 --sky $sky
    python3 tools/pmc_summary.py "$OUT"/c5_${sky}_pmc[0-9] --width 3840 --height 2160 --sky-width $sky --json "$OUT/traffic_r04_config5_sky$sky.json" > "$OUT/c5_${sky}_pmc_summary.txt"
  done
  trace single_stats --no-extra-legs --no-cpu-baseline --batch 1
  pmc single 3 5 --batch 1
  python3 tools/pmc_summary.py "$OUT"/single_pmc[0-9] --frames-per-launch 1 > "$OUT/single_pmc_summary.txt"
  echo "4K / config 5 / single-frame kernel done"
fi
if [[ $PART == *b* ]]; then
  bash tools/other_configs.sh "$OUT/other_configs_bench.jsonl" 2> /dev/null
  echo "other configs done"
  python tools/config5.py 2>&1 | grep -v amdgpu.ids > "$OUT/config5_one_gpu.txt" || true
  python tools/batch_times.py 2>&1 | grep -v amdgpu.ids > "$OUT/batch_and_partition_times.txt" || true
  python tools/inflight_times.py 2>&1 | grep -v amdgpu.ids > "$OUT/inflight_times.txt" || true
  { python tools/multi_times.py; echo "--- dry launches (RT_OPT_FLAGS bit 4): the host's calls without the device's work"; python tools/multi_times.py --dry; } 2>&1 | grep -v amdgpu.ids > "$OUT/multi_times_one_gpu.txt" || true
  python tools/pcie_inclusive.py 2>&1 | grep -v amdgpu.ids > "$OUT/presentation.txt" || true
  { tools/microbench/host_calls; tools/microbench/d2h_rate; } > "$OUT/host_calls_and_d2h_rate.txt" 2>&1 || true
  echo "times done"
  for s in 64 1024; do
    echo "=== $s spheres, 8 frames per launch (make prof)"; RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 8 --spheres $s 2>&1 | grep -v amdgpu.ids
    echo "=== $s spheres, event counts (make events)"; RT355_LIB=$REPO/rsraytracingv3_amd/librt355_events.so python tools/phase_profile.py --batch 8 --spheres $s 2>&1 | grep "direction-mask scans\|events of that launch"
  done > "$OUT/phases.txt"
  echo "=== 64 spheres, one frame per launch (make prof)" >> "$OUT/phases.txt"
  RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 1 2>&1 | grep -v amdgpu.ids >> "$OUT/phases.txt"
  echo "=== 1024 spheres, the candidate-list scan's own steps (make prof-lists): 0 setup, 1 entry load + planes, 2 pair list, 3 trips, 4 decision, 5 merge; 7 everything else" >> "$OUT/phases.txt"
  RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof_lists.so python tools/phase_profile.py --batch 8 --spheres 1024 2>&1 | grep -v amdgpu.ids >> "$OUT/phases.txt"
  cat "$OUT/phases.txt"
  bash tools/dirlist_sweep.sh "$OUT/dirlist_sweep.txt" > /dev/null 2>&1 || true
  python tools/set_scene_time.py 2>&1 | grep -v amdgpu.ids > "$OUT/set_scene_time.txt" || true
  { python tools/word_boundary_times.py; python tools/outlier_times_large.py; } 2>&1 | grep -v amdgpu.ids > "$OUT/word_boundary_and_outliers.txt" || true
fi
