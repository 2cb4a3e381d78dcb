set -e
OUT=$(realpath -m gpurun_out/prof_r02b); REPO=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$REPO/bench.py" --no-extra-legs > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_all_legs" -o bench -- python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_all_legs_under_rocprof.json" 2> "$OUT/stats_all_legs.log"
cd "$REPO"; python bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_flags.json" 2>/dev/null
RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 8 2>&1 | grep -v amdgpu.ids > "$OUT/phases.txt"
RT355_LIB=$REPO/rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py --batch 1 2>&1 | grep -v amdgpu.ids >> "$OUT/phases.txt"
cat "$OUT/phases.txt"
