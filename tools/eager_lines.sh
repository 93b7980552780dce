#!/bin/bash
# Eager (no HIP graph) layer steps of the launch-bound workloads, with the backward on autograd's thread and on the
# calling thread; one JSON line each into gpurun_out/eager_lines.jsonl.
cd $GRAFT_REPO_ROOT
: > gpurun_out/eager_lines.jsonl
for W in "--workload qm9" "--workload qm9 --layer egnn --dtype bf16" "--workload modelnet" "--workload modelnet --layer egnn"; do
  for T in "" "--backward-on-caller"; do
    timeout -k 10 120 python bench.py $W $T --steps 300 --warmup 30 --no-cpu-baseline --no-model-step 2> /dev/null | tail -n 1 >> gpurun_out/eager_lines.jsonl || exit 1
  done
done
