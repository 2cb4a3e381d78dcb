#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the render and resolve kernels from rocprofv3 counter_collection CSVs.

  python tools/pmc_summary.py DIR [DIR ...] [--json profiles/traffic_rNN.json --frames-per-launch 8 ...]
Each DIR is the -d directory of one `rocprofv3 --pmc ...` pass (separate passes, never combined with trace
domains).  With --json the merged per-launch averages are written in the form bench.py reads:
  render_kernel.hbm_bytes_per_launch = FETCH_SIZE * 1024 (raw: the render kernel's reads are scattered 16-byte and
      4-byte gathers, which FETCH_SIZE counts exactly — profiles/r01_fetch_size_calibration.md) + WRITE_SIZE * 1024
  resolve_kernel.hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 (coalesced streams read half on gfx950 —
      MI355X_MICROARCH.md, HBM) + WRITE_SIZE * 1024
  render_kernel.valu_issue_util = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rsraytracingv3_amd._native import kernel_source_hash  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--json")
ap.add_argument("--frames-per-launch", type=int, default=8)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--objects", type=int, default=65)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--sky-width", type=int, default=2048)
args = ap.parse_args()
# a launch's render step may be two kernels: render_blocks, then tail_stream (RT_OPT_TAIL_SPILL); their per-launch averages are
# summed into "render_kernel" below, and kept apart under "render_parts"
KERNELS = {"render_kernel": "render_", "tail_kernel": "tail_stream", "resolve_kernel": "resolve_batch_kernel"}
merged = {k: {} for k in KERNELS}
names = {k: set() for k in KERNELS}
counts = {k: {} for k in KERNELS}
for d in args.dirs:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = {k: collections.defaultdict(list) for k in KERNELS}
        for r in csv.DictReader(open(f)):
            for key, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    m = re.search(r"((?:render_|tail_)\w+(?:<[\w, ]+>)?|resolve_batch_kernel)", r["Kernel_Name"])
                    names[key].add(m.group(1) if m else r["Kernel_Name"])
        for key in KERNELS:
            # launches of the warm-up may carry fewer frames; keep the modal half (the timed launches) by median
            for c, v in acc[key].items():
                v = sorted(v)
                core = v[len(v) // 4: len(v) - len(v) // 4] or v
                merged[key][c] = sum(core) / len(core)
                counts[key][c] = len(v)
            if acc[key]:
                print(d, key, " ".join("%s=%.4g (n=%d)" % (c, merged[key][c], counts[key][c]) for c in sorted(acc[key])))
if args.json:
    out = {
        "kernel_source_sha256": kernel_source_hash(),  # bench.py quotes these counters only for a tree with the same sources
        "workload_key": {"width": args.width, "height": args.height, "objects": args.objects, "bounces": args.bounces,
                         "sky_width": args.sky_width, "frames_per_launch": args.frames_per_launch, "n_gpus": 1},
        "note": "separate rocprofv3 --pmc passes (tools/profile_round.sh), interquartile mean over the kernel's dispatches. "
                "FETCH_SIZE / WRITE_SIZE are in KiB.  Render kernel: scattered gathers, FETCH_SIZE taken as is "
                "(profiles/r01_fetch_size_calibration.md); resolve kernel: coalesced streams, FETCH_SIZE doubled "
                "(MI355X_MICROARCH.md, HBM).  valu_issue_util = SQ_INSTS_VALU * 2 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8).",
    }
    if merged["tail_kernel"]:  # the launch's render step = both kernels
        parts = {}
        for key in ("render_kernel", "tail_kernel"):
            parts[", ".join(sorted(names[key]))] = {c: round(v, 2) for c, v in sorted(merged[key].items())}
        out["render_parts"] = parts
        for c, v in merged["tail_kernel"].items():
            merged["render_kernel"][c] = merged["render_kernel"].get(c, 0.0) + v
        names["render_kernel"] = {" + ".join(sorted(names["render_kernel"]) + sorted(names["tail_kernel"]))}
    for key in ("render_kernel", "resolve_kernel"):
        m = merged[key]
        fetch = m.get("FETCH_SIZE", 0.0) * 1024.0 * (2.0 if key == "resolve_kernel" else 1.0)
        write = m.get("WRITE_SIZE", 0.0) * 1024.0
        rec = {"kernel": ", ".join(sorted(names[key])), "counters_per_launch": {c: round(v, 2) for c, v in sorted(m.items())},
               "dispatches": counts[key].get("FETCH_SIZE"), "fetch_bytes": round(fetch, 1), "write_bytes": round(write, 1),
               "hbm_bytes_per_launch": round(fetch + write, 1),
               "hbm_bytes_per_frame": round((fetch + write) / args.frames_per_launch, 1)}
        if m.get("SQ_INSTS_VALU") and m.get("GRBM_GUI_ACTIVE"):
            rec["valu_issue_util"] = round(m["SQ_INSTS_VALU"] * 2.0 / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0), 4)
        out[key] = rec
    with open(args.json, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
