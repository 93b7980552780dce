#!/bin/bash
# Two SQ counter passes over tools/gemm_one.py <kind>; prints per-kernel means.  Usage: tools/pmc_gemm.sh tn
K=${1:-tn}
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P -d $GRAFT_REPO_ROOT/gpurun_out/pmcg_${K}_$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py $K > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmcg_${K}_$i.err || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - $K <<'PY'
import csv, sys, collections
k = sys.argv[1]
for i in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"gpurun_out/pmcg_{k}_{i}/p_counter_collection.csv")):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in acc.items():
        print(kn)
        for c, v in cs.items():
            print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
