#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the render kernel from rocprofv3 counter_collection CSVs."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "render" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, " ".join("%s=%.4g" % (k, sum(v) / len(v)) for k, v in sorted(acc.items())))
