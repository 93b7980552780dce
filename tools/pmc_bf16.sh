#!/bin/bash
# HBM traffic per kernel of the bf16 headline step: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes.
# Output: gpurun_out/profiles/<round>_pmc_traffic_bf16.json            usage: tools/pmc_bf16.sh r03
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C -d gpurun_out/pmcb_$C -o p --output-format csv -- python3 bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-model-step > /dev/null 2> gpurun_out/pmcb_$C.err || exit 1
done
python3 tools/pmc_summary.py gpurun_out/profiles/${R}_pmc_traffic_bf16.json powerlaw=gpurun_out/pmcb_FETCH_SIZE/p_counter_collection.csv,gpurun_out/pmcb_WRITE_SIZE/p_counter_collection.csv || exit 1
python3 - gpurun_out/profiles/${R}_pmc_traffic_bf16.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
d["dtype"] = "bf16"
d["method"] = d["method"].replace("`bench.py --steps 3 --warmup 1`", "`bench.py --dtype bf16 --steps 3 --warmup 1`")
json.dump(d, open(sys.argv[1], "w"), indent=1)
for wl, ks in d["kernels"].items():
    for k, v in ks.items():
        if v["hbm_bytes_per_launch"] > 5e8: print(k[:70], v["launches"], round(v["hbm_bytes_per_launch"] / 1e9, 2), "GB")
PY
