#!/bin/bash
# One rocprofv3 --pmc pass over tools/bench_net.py for an experiment library:
#   tools/pmc_one.sh <lib.so> <tag> <games> "<counters>"   -> gpurun_out/pmc1_<tag>.txt (per-dispatch averages, recurrent tower)
set -e
LIB=$1; TAG=$2; GAMES=$3; CTRS=$4
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG
mkdir -p $OUT
export MZ_LIB=$GRAFT_REPO_ROOT/$LIB
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_net.py --games $GAMES --iters 3 --precision f16 > $OUT.log 2>&1
python3 - <<PY > $OUT.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("mz::impl_f16::", "").replace("mz::impl_bf16::", "").replace("void ", "")[:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, v in sorted(agg.items()):
    if "k_tower" in k or "k_seg" in k or "k_bseg" in k:
        n = len(disp[k]); print(f"$TAG {k:46s} n={n} " + " ".join(f"{a}={b / n:.4g}" for a, b in sorted(v.items())))
PY
cat $OUT.txt
