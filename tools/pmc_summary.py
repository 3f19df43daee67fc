#!/usr/bin/env python3
"""Per-kernel means of the rocprofv3 counter CSVs that tools/gpu_pmc.sh leaves under gpurun_out/pmc_<tag>/p*/ -> <dir>/summary.txt
(the input of tools/pmc_traffic.py).  Also runnable here on the merged gpurun_out/ when a short GPU call ended before this step.
    python tools/pmc_summary.py gpurun_out/pmc_<tag>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
want = os.environ.get("PMC_KERNEL", "")     # substring of the kernel name to summarise (default: the attention kernels)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?")
        if want:
            if want not in k:
                continue
        elif "attn_" not in k or "profile" in k:
            continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:32s} n={len(v)} mean={sum(v) / len(v):.6g}\n")
print(open(out + "/summary.txt").read())
