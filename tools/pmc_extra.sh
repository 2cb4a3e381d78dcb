#!/bin/bash
# One more SQ pass: branches, instruction fetches, lane utilisation, occupancy.  bash tools/pmc_extra.sh OUT [bench flags]
OUT=$(realpath -m "$1"); shift
REPO=$(pwd); mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/x" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra-legs --steps 48 --warmup 16 --min-seconds 0.05 --clock-warmup 0.2 "$@" > "$OUT/x.json" 2> "$OUT/x.log"
cd "$REPO"
python3 tools/pmc_summary.py "$OUT/x" | grep render_kernel
