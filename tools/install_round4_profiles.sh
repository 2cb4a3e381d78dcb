#!/bin/bash
# Copies what tools/collect_round4.sh left under gpurun_out/prof_r04 into profiles/ under the names profiles/README.md indexes:
#   bash tools/install_round4_profiles.sh acdb     (the parts that were collected)
set -e
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r04
part=$1
if [[ $part == *a* ]]; then
cp $P/stats/bench_kernel_stats.csv profiles/r04_kernel_stats.csv; cp $P/stats.json profiles/r04_bench_under_rocprof.json
cp $P/stats_one_at_a_time/bench_kernel_stats.csv profiles/r04_one_at_a_time_kernel_stats.csv; cp $P/stats_one_at_a_time.json profiles/r04_bench_one_at_a_time_under_rocprof.json
cp $P/stats_all_legs/bench_kernel_stats.csv profiles/r04_all_legs_kernel_stats.csv; cp $P/stats_all_legs.json profiles/r04_all_legs_bench_under_rocprof.json
cp $P/pmc_summary.txt profiles/r04_pmc_summary.txt; cp $P/traffic_r04.json profiles/traffic_r04.json; cp $P/trace_union.txt profiles/r04_trace_union.txt
for n in driver_flags default one_at_a_time dirmask_off; do cp $P/bench_$n.json profiles/r04_bench_$n.json; done
fi
if [[ $part == *c* ]]; then
cp $P/config3_stats/bench_kernel_stats.csv profiles/r04_config3_kernel_stats.csv; cp $P/config3_stats.json profiles/r04_config3_bench_under_rocprof.json
cp $P/config3_stats_one_at_a_time/bench_kernel_stats.csv profiles/r04_config3_one_at_a_time_kernel_stats.csv; cp $P/config3_stats_one_at_a_time.json profiles/r04_config3_bench_one_at_a_time_under_rocprof.json
cp $P/config3_pmc_summary.txt profiles/r04_config3_pmc_summary.txt; cp $P/traffic_r04_config3.json profiles/traffic_r04_config3.json
cp $P/bench_config3.json profiles/r04_bench_config3.json; cp $P/bench_config3_grid.json profiles/r04_bench_config3_grid.json; cp $P/bench_256_512_lists_then_grid.jsonl profiles/r04_bench_256_512_lists_then_grid.jsonl
fi
if [[ $part == *d* ]]; then
cp $P/k4_stats/bench_kernel_stats.csv profiles/r04_4k_kernel_stats.csv; cp $P/k4_stats.json profiles/r04_4k_bench_under_rocprof.json; cp $P/k4_pmc_summary.txt profiles/r04_4k_pmc_summary.txt; cp $P/traffic_r04_4k.json profiles/traffic_r04_4k.json
cp $P/single_stats/bench_kernel_stats.csv profiles/r04_single_frame_kernel_stats.csv; cp $P/single_stats.json profiles/r04_single_frame_bench_under_rocprof.json; cp $P/single_pmc_summary.txt profiles/r04_single_frame_pmc_summary.txt
cp $P/traffic_r04_config5_sky2048.json profiles/; cp $P/traffic_r04_config5_sky8192.json profiles/
fi
if [[ $part == *b* ]]; then
for f in batch_and_partition_times config5_one_gpu dirlist_sweep host_calls_and_d2h_rate inflight_times multi_times_one_gpu presentation set_scene_time word_boundary_and_outliers; do [ -f $P/$f.txt ] || continue; cp $P/$f.txt profiles/r04_$f.txt; done
cp $P/phases.txt profiles/r04_phase_profile.txt; cp $P/other_configs_bench.jsonl profiles/r04_other_configs_bench.jsonl
fi
python3 tools/config5_fetch_summary.py $P > profiles/r04_config5_fetch.txt 2>/dev/null || true
sed -i "s#/tmp/code/arlohb__RSRayTracingV3/repo/##g; s#/root/repo/##g" profiles/r04_*pmc_summary.txt
