#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: duration per (kernel, grid) and a timeline slice."""
import collections
import csv
import glob
import sys


def short(n):
    n = n.replace("void gklt::(anonymous namespace)::", "").replace("gklt::(anonymous namespace)::", "")
    return n.split("(")[0][:40]


def main(d, timeline=0):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    g = collections.defaultdict(list)
    for r in rows:
        g[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"], r.get("Grid_Size_Z", "1"), r["LDS_Block_Size"])].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("%-42s %8s %5s %5s %6s %6s %8s %8s %8s" % ("kernel", "gridX", "gridY", "gridZ", "lds", "calls", "med_us", "min_us", "max_us"))
    for k, v in sorted(g.items()):
        v.sort()
        print("%-42s %8s %5s %5s %6s %6d %8.2f %8.2f %8.2f" % (k + (len(v), v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3)))
    if timeline:
        rs = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
        i0 = len(rs) * 2 // 3
        base = int(rs[i0]["Start_Timestamp"])
        for r in rs[i0:i0 + timeline]:
            print("%9.2f %8.2f  q%s %s %sx%s" % ((int(r["Start_Timestamp"]) - base) / 1e3,
                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], short(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Y"]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
