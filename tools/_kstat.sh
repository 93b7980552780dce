#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/ks -o p --output-format csv -- python3 bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-model-step > /dev/null 2> gpurun_out/ks.err || exit 1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/ks/p_kernel_stats.csv")))
for r in rows[:12]:
    if "gemm" in r["Name"]: print("  ", r["Name"][:100], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3))
PY
