#!/bin/bash
# Direction-mask table A/B on one device: off / on, by sphere count, against the sphere grid, the reference's 80 slots.
#   bash tools/dirmask_sweep.sh > gpurun_out/dirmask_sweep.txt
B="python bench.py --no-cpu-baseline --no-extra-legs --min-seconds 0.6 --clock-warmup 1.0"
run() { echo -n "$* : "; $B "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], 'Mrays/s, valu.frac', d['valu']['frac'], d['roofline']['kernel'], 'avg launch ms', d['roofline']['avg_launch_ms'])"; }
run --dirmask 0
run --dirmask 1
run --dirmask 0
run --dirmask 1
run --dirmask 1 --dirmask-k 8
run --dirmask 1 --dirmask-k 12 --dirmask-cells 16384
run --dirmask 1 --dirmask-k 10
run --dirmask 1 --dirmask-k 16
for n in 12 16 20 24 28 32 40 48; do run --dirmask 1 --spheres $n; run --dirmask 0 --spheres $n; done
run --dirmask 1 --slots 80
run --dirmask 0 --slots 80
for n in 80 96 128; do run --spheres $n --dirmask 1 --grid-min 200; run --spheres $n --dirmask 0 --grid-min 200; run --spheres $n --grid-min 1; done
run --dirmask 1 --batch 1
run --dirmask 0 --batch 1
run --dirmask 1 --width 3840 --height 2160
run --dirmask 0 --width 3840 --height 2160
