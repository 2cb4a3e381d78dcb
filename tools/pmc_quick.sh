#!/bin/bash
# Three SQ counter passes of the headline leg (issue utilisation, wait shares) — a quick look between full profile rounds.
#   bash tools/pmc_quick.sh gpurun_out/pmcq [extra bench flags]
set -e
OUT=$(realpath -m "$1"); shift
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc$i" -o pmc -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra-legs --steps 48 --warmup 16 --min-seconds 0.05 --clock-warmup 0.2 "$@" > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.log"
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT"/pmc[0-9] --json "$OUT/traffic.json" > "$OUT/pmc_summary.txt"
python3 - "$OUT/traffic.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["render_kernel"]
c = d["counters_per_launch"]
print("kernel", d["kernel"])
print("VALU wave-instructions per launch: %.1f M; issue util %.3f; cycles per VALU inst per SIMD %.2f" % (
    c["SQ_INSTS_VALU"] / 1e6, d.get("valu_issue_util", 0), 1024 * c["GRBM_GUI_ACTIVE"] / 8 / c["SQ_INSTS_VALU"]))
wc = c["SQ_WAVE_CYCLES"]
print("wave-cycles: issuing %.1f %%, waiting on an instruction %.1f %%, parked (waitcnt / barrier) %.1f %%" % (
    100 * c["SQ_ACTIVE_INST_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_WAIT_ANY"] / wc))
print("SALU %.1f M, LDS %.1f M (bank-conflict cycles %.1f M), VMEM rd %.2f M wr %.2f M, busy cycles %.2f M" % (
    c["SQ_INSTS_SALU"] / 1e6, c["SQ_INSTS_LDS"] / 1e6, c["SQ_LDS_BANK_CONFLICT"] / 1e6, c["SQ_INSTS_VMEM_RD"] / 1e6,
    c["SQ_INSTS_VMEM_WR"] / 1e6, c["GRBM_GUI_ACTIVE"] / 8 / 1e6))
PY
