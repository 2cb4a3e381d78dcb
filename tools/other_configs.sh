#!/bin/bash
# bench.py lines of ONE GPU for the BASELINE configs other than the headline (config 2), and config 5's accumulation:
#   bash tools/other_configs.sh OUT.jsonl
OUT=$(realpath -m "$1")
: > "$OUT"
COMMON="--no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
python bench.py $COMMON --width 256 --height 256 --spheres 4 --bounces 1 --slots 80 --sky 64 >> "$OUT"     # config 1 (plumbing case)
python bench.py $COMMON --spheres 1024 >> "$OUT"                                                            # config 3 (candidate lists)
python bench.py $COMMON --spheres 1024 --dirlist 0 >> "$OUT"                                                # config 3 on the sphere grid (round 3)
python bench.py $COMMON --spheres 1024 --dirlist 0 --flags 2 >> "$OUT"                                      # config 3, lists and grid off (linear scan)
python bench.py $COMMON --width 3840 --height 2160 >> "$OUT"                                                # config 4's frame on one GPU
python bench.py $COMMON --width 3840 --height 2160 --sky 8192 --steps 64 >> "$OUT"                          # config 5's frame and skybox
python tools/config5.py 2>&1 | grep -v amdgpu.ids
