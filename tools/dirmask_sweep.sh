#!/bin/bash
# Direction-mask table A/B on one device: off, and on at several resolutions (headline workload, 8 frames per launch).
#   bash tools/dirmask_sweep.sh > gpurun_out/dirmask_sweep.txt
B="python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s, valu.frac', d['valu']['frac'], d['roofline']['kernel'], 'avg launch ms', d['roofline']['avg_launch_ms'])"; }
run --dirmask 0
run --dirmask 1
run --dirmask 0
run --dirmask 1
for k in 4 6 8 12 16; do for c in 512 4096 16384; do run --dirmask 1 --dirmask-k $k --dirmask-cells $c; done; done
run --dirmask 1 --spheres 32
run --dirmask 0 --spheres 32
run --dirmask 1 --spheres 16
run --dirmask 0 --spheres 16
run --dirmask 1 --batch 1
run --dirmask 0 --batch 1
