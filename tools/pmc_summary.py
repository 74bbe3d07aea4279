#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv), per launch."""
import collections
import csv
import glob
import sys


def short(n):
    n = n.replace("void gklt::(anonymous namespace)::", "")
    return n.split("(")[0][:34]


def main(dirs):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                key = (short(r["Kernel_Name"]), r["Grid_Size"])
                out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in sorted(out):
        if "rocclr" in key[0]:
            continue
        print("%-36s grid %-8s" % key, "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(out[key].items())))


if __name__ == "__main__":
    main(sys.argv[1:])
