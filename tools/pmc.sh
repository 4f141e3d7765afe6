#!/bin/bash
# PMC passes for the inference kernels (each pass its own run, within the per-block counter limits).
# Usage (on the GPU box): bash tools/pmc.sh <tag> [games] [workload] [precision]
#   -> gpurun_out/pmc_<tag>/summary.txt and traffic_<workload>_<games>.json (copy both into profiles/)
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
GAMES=${2:-4096}
WL=${3:-go9}
PREC=${4:-f16}
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="GRBM_GUI_ACTIVE"
P6="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_net.py --workload $WL --games $GAMES --iters 3 --precision $PREC > $OUT/p$i.log 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections, json, re
print("# per-dispatch averages over the recurrentInference launches of tools/bench_net.py --workload $WL --games $GAMES --precision $PREC (rocprofv3 --pmc, one pass per counter group)")
traffic = {"fetch": {}, "write": {}}
for i in (1, 2, 3, 4, 5, 6):
    fs = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"mz::(\(anonymous namespace\)|impl_f16|impl_bf16)::", "", r["Kernel_Name"]).replace("void ", "")[:44]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    for k, v in sorted(agg.items()):
        if any(t in k for t in ("k_seg", "k_bcast", "k_bseg", "k_tower", "k_heads", "k_net_fp32")):
            n = len(disp[k])
            print(f"{k:46s} dispatches {n:3d}  " + "  ".join(f"{a}={b / n:.4g}" for a, b in sorted(v.items())))
            # the recurrent chain only (IN_OBS kernels belong to the one initialInference that sets the states up)
            # (the kernels whose 2nd template argument IN is 0 = IN_OBS belong to the one initialInference that sets the states up)
            rec = not re.search(r"Geo<[^>]*>, 0,", k)
            if "FETCH_SIZE" in v and rec: traffic["fetch"][k] = v["FETCH_SIZE"] / n
            if "WRITE_SIZE" in v and rec: traffic["write"][k] = v["WRITE_SIZE"] / n
# HBM bytes per recurrentInference launch chain.  gfx950: FETCH_SIZE (KB) tallies 128-B requests at 64 B for wide
# coalesced reads -> doubled; WRITE_SIZE (KB) is exact (MI355X_MICROARCH.md, HBM).
fetch_kb = sum(traffic["fetch"].values()); write_kb = sum(traffic["write"].values())
json.dump({"source": "tools/pmc.sh $1 $GAMES $WL $PREC: tools/bench_net.py, FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes",
           "workload": "$WL", "precision": "$PREC", "games": $GAMES,
           "fetch_size_kb_per_launch": traffic["fetch"], "write_size_kb_per_launch": traffic["write"],
           "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled; WRITE_SIZE exact",
           "bytes_per_sample": (2 * fetch_kb + write_kb) * 1024.0 / $GAMES},
          open("$OUT/traffic_${WL}_$GAMES.json", "w"), indent=1)
PY
cat $OUT/traffic_${WL}_$GAMES.json
cat $OUT/summary.txt
