#!/usr/bin/env python3
"""Timeline of the last bench step from a rocprofv3 --kernel-trace csv: kernel, start offset, duration, gap before it (us).
usage: trace_gaps.py <kernel_trace.csv> [n_last_kernels]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps start at the k_collapse_local<2 launches
marks = [i for i, r in enumerate(rows) if "k_part_pass<64" in r["Kernel_Name"] or "k_collapse_local<2" in r["Kernel_Name"]]
step = rows[marks[-3]:marks[-2]]          # a timed step (the last launch belongs to the invariants pass)
t0 = int(step[0]["Start_Timestamp"]); prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"mg::|\(.*", "", r["Kernel_Name"])[:44]
    print("%-44s q=%s start %8.1f  dur %8.1f  gap %7.1f" % (name, r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((prev_end - t0) / 1e3))
