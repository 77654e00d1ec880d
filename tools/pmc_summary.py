#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc counter_collection.csv (tens of MB) to mean counter values per kernel symbol.
usage: pmc_summary.py <counter_collection.csv> <out.csv> [name filter substrings ...]"""
import collections
import csv
import re
import sys


def main():
    src, dst, filt = sys.argv[1], sys.argv[2], sys.argv[3:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    with open(src) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            n = re.sub(r"\(.*", "", n)
            if filt and not any(s in n for s in filt):
                continue
            a = agg[n][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    with open(dst, "w") as o:
        o.write("kernel,counter,mean_per_launch,launches\n")
        for k, v in agg.items():
            for c, (s, n) in v.items():
                o.write(f"\"{k}\",{c},{s / n:.0f},{n}\n")


if __name__ == "__main__":
    main()
