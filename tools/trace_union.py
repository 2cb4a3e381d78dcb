#!/usr/bin/env python3
"""Device time of overlapping kernels from a rocprofv3 kernel trace.

    python tools/trace_union.py gpurun_out/prof_r02/stats [--kernel render_blocks] [--last 200]

With several launches in flight (RT_OPT_BATCH_INFLIGHT / RT_OPT_INFLIGHT) the render kernels of consecutive
launches share the device: the `--stats` average of their durations counts that shared time once per kernel.
This reads the `*_kernel_trace.csv` of the same pass, takes the last N dispatches of the kernel (the timed
blocks of bench.py come last) and prints
  * the mean duration of one kernel (what `--stats` averages),
  * the union of their [start, end) intervals divided by their number: device time per launch, the figure
    bench.py reports as roofline.avg_launch_ms (there from HIP events: span of the block / launches),
  * how many of them were running at once on average (sum of durations / union).
"""
import argparse
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--kernel", default="render_blocks,tail_stream",
                    help="comma-separated name fragments: a launch's render step is render_blocks followed by tail_stream")
    ap.add_argument("--last", type=int, default=200)
    args = ap.parse_args()
    paths = glob.glob(os.path.join(args.directory, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        sys.exit("no *kernel_trace.csv under %s" % args.directory)
    spans = []
    kernels = [k for k in args.kernel.split(",") if k]
    launches = 0  # dispatches of the first kernel named = launches
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if any(k in row["Kernel_Name"] for k in kernels):
                    spans.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
                    launches += kernels[0] in row["Kernel_Name"]
    if not spans:
        sys.exit("no dispatch of a kernel named *%s*" % args.kernel)
    spans.sort()
    per_launch = max(1, round(len(spans) / max(launches, 1)))  # kernels per launch (2 with a tail_stream)
    spans = spans[-args.last * per_launch:]
    total = sum(e - s for s, e in spans)
    union, cur_s, cur_e = 0, spans[0][0], spans[0][1]
    for s, e in spans[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    n = len(spans) // per_launch
    print("%s: last %d launches (%d kernel(s) each)" % (" + ".join(kernels), n, per_launch))
    print("  mean kernel time per launch   %.4f ms   (sum of the durations / launches)" % (total / n * 1e-6))
    print("  device time per launch        %.4f ms   (union of the intervals / launches)" % (union / n * 1e-6))
    print("  kernels running at once       %.2f" % (total / union))
    print("  first start .. last end       %.4f ms per launch" % ((spans[-1][1] - spans[0][0]) / n * 1e-6))


if __name__ == "__main__":
    main()
