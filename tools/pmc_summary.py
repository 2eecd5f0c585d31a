#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, the LAST dispatch's counter values
(the timed steps come last) -- usage: pmc_summary.py file.csv [file.csv ...]"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    m = re.search(r"(k_\w+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(paths):
    for p in paths:
        per = OrderedDict()
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            d = per.setdefault(k, defaultdict(dict))
            d[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        print("==", p)
        for k, disp in per.items():
            last = disp[sorted(disp, key=int)[-1]]
            print("%-34s n=%-3d %s" % (k, len(disp), "  ".join("%s=%.4g" % kv for kv in sorted(last.items()))))


if __name__ == "__main__":
    main(sys.argv[1:])
