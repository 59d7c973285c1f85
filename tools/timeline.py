#!/usr/bin/env python3
"""Prints the kernel timeline of the LAST decode call in a rocprofv3 --kernel-trace CSV (start / end in ms relative to the call's first kernel).
    python tools/timeline.py path/to/*_kernel_trace.csv [ncalls_back]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "peel" in r["Kernel_Name"]]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    nm = re.sub(r"ldpc_amd::\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.3f} -> {(int(r['End_Timestamp']) - t0) / 1e6:8.3f} ms  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size', '?'):>8} lds {r.get('LDS_Block_Size', '?'):>7}  {nm}")
