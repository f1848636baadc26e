#!/usr/bin/env python3
"""Summarises rocprofv3 counter_collection.csv files: per kernel, mean counter value per dispatch and mean duration."""
import collections
import csv
import glob
import sys


def main(root):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"] + f] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k in sorted(agg):
        d = list(dur[k].values())
        print("%-28s n=%-4d avg_ms=%.4f " % (k[-28:], len(d), sum(d) / len(d)) +
              " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
