#!/usr/bin/env python3
"""Print a short per-kernel summary of a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 8]:
    name = r["Name"].replace("mz::(anonymous namespace)::", "").replace("mz::impl_f16::", "").replace("mz::impl_bf16::", "").replace("void ", "")
    print(f"{name[:48]:48s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
