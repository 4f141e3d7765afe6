#!/bin/bash
# PMC passes for the inference kernels (each pass its own run, within the per-block counter limits).
# Usage (on the GPU box): bash tools/pmc.sh <tag> [games]   -> gpurun_out/pmc_<tag>/summary.txt
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
GAMES=${2:-4096}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_net.py --games $GAMES --iters 3 > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
print("# per-dispatch averages over the recurrentInference launches of tools/bench_net.py --games $GAMES (rocprofv3 --pmc, one pass per counter group)")
for i in (1, 2, 3, 4, 5):
    fs = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("mz::(anonymous namespace)::", "").replace("void ", "")[:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        if any(t in k for t in ("k_seg", "k_bcast", "k_bseg", "k_tower", "k_heads")):
            n = len(disp[k])
            print(f"{k:46s} dispatches {n:3d}  " + "  ".join(f"{a}={b / n:.4g}" for a, b in sorted(v.items())))
PY
cat $OUT/summary.txt
