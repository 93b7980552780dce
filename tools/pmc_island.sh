#!/bin/bash
# Two SQ counter passes over the QM9-shaped layer step (bench.py --workload qm9 [args]); prints per-kernel means of
# the island kernels.  Usage: tools/pmc_island.sh [--layer egnn --dtype bf16]
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  cd $GRAFT_REPO_ROOT
  timeout -s KILL 120 rocprofv3 --pmc $P -d gpurun_out/pmci_$i -o p --output-format csv -- python3 bench.py --workload qm9 "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-model-step > /dev/null 2> gpurun_out/pmci_$i.err || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections
for i in (1, 2):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"gpurun_out/pmci_{i}/p_counter_collection.csv")):
        name = next((k for k in ("k_island_fwd", "k_island_bwd", "k_island_reduce") if k in r["Kernel_Name"]), None)
        if name:
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in acc.items():
        print(kn)
        for c, v in cs.items():
            print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
