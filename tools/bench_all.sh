#!/bin/bash
# One clean (unprofiled) bench line per configuration, eager and replayed from a HIP graph where the step is
# launch-bound.  Output: gpurun_out/profiles/<round>_bench_lines.jsonl        usage: tools/bench_all.sh r02
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
out=$O/${R}_bench_lines.jsonl
: > $out
run() { timeout -k 10 300 python3 bench.py "$@" --no-model-step 2>> gpurun_out/bench_all.err | tail -1 >> $out || exit 1; }
run --workload qm9 --steps 100 --warmup 10
run --workload qm9 --steps 300 --warmup 30 --backward-on-caller
run --workload qm9 --steps 200 --warmup 20 --hip-graph
run --workload qm9 --mode cat --steps 200 --warmup 20 --hip-graph
run --workload qm9 --layer egnn --dtype bf16 --steps 100 --warmup 10
run --workload qm9 --layer egnn --dtype bf16 --steps 300 --warmup 30 --backward-on-caller
run --workload qm9 --layer egnn --dtype bf16 --steps 200 --warmup 20 --hip-graph
run --workload flickr --steps 50 --warmup 10
run --workload flickr --first-layer --steps 50 --warmup 10
run --workload modelnet --steps 100 --warmup 10
run --workload modelnet --steps 300 --warmup 30 --backward-on-caller
run --workload modelnet --first-layer --steps 200 --warmup 20 --hip-graph
run --workload modelnet --steps 200 --warmup 20 --hip-graph
run --workload modelnet --layer egnn --steps 100 --warmup 10
run --workload modelnet --layer egnn --steps 300 --warmup 30 --backward-on-caller
run --workload modelnet --layer egnn --steps 200 --warmup 20 --hip-graph
run --dtype bf16 --steps 10 --warmup 3
python3 - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    c = d["config"]
    print(f'{c["workload"][:28]:28s} {c["layer"]:12s} {d["dtype"]:5s} {c.get("launch", ""):38s} {d["ms_per_step"]:9.4f} ms  {d["value"]/1e9:7.3f} G edges/s  verified={(d.get("verified") or {}).get("ok")}')
PY
