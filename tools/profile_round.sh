#!/bin/bash
# Regenerates the per-round evidence on the GPU box: kernel-trace stats for the four workloads, the default
# bench line, and the FETCH_SIZE / WRITE_SIZE passes of the two big workloads.  Output: gpurun_out/profiles/<round>_*
# usage: tools/profile_round.sh r01
set -o pipefail
R=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in powerlaw flickr modelnet qm9; do
  echo "== $W kernel trace"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$W -o $W --output-format csv -- \
      python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-model-step > $O/${R}_bench_${W}_f32.json 2> gpurun_out/prof_$W.err || exit 1
  cp gpurun_out/prof_$W/${W}_kernel_stats.csv $O/${R}_bench_${W}_f32_kernel_stats.csv || exit 1
done
echo "== powerlaw bf16 kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_powerlaw_bf16 -o powerlaw --output-format csv -- \
    python3 bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-model-step > $O/${R}_bench_powerlaw_bf16.json 2> gpurun_out/prof_powerlaw_bf16.err || exit 1
cp gpurun_out/prof_powerlaw_bf16/powerlaw_kernel_stats.csv $O/${R}_bench_powerlaw_bf16_kernel_stats.csv || exit 1
echo "== qm9 / modelnet EGNN kernel traces"
for SPEC in "qm9 bf16" "modelnet f32"; do
  set -- $SPEC
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_egnn_$1 -o $1 --output-format csv -- \
      python3 bench.py --workload $1 --layer egnn --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-model-step > $O/${R}_bench_$1_egnn_$2.json 2> gpurun_out/prof_egnn_$1.err || exit 1
  cp gpurun_out/prof_egnn_$1/$1_kernel_stats.csv $O/${R}_bench_$1_egnn_$2_kernel_stats.csv || exit 1
done
SPECS=""
for W in powerlaw flickr; do
  for C in FETCH_SIZE WRITE_SIZE; do
    echo "== $W pmc $C"
    timeout -s KILL 300 rocprofv3 --pmc $C -d gpurun_out/pmc_${W}_$C -o p --output-format csv -- \
        python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-model-step > /dev/null 2> gpurun_out/pmc_${W}_$C.err || exit 1
  done
  SPECS="$SPECS $W=gpurun_out/pmc_${W}_FETCH_SIZE/p_counter_collection.csv,gpurun_out/pmc_${W}_WRITE_SIZE/p_counter_collection.csv"
done
python3 tools/pmc_summary.py $O/${R}_pmc_traffic.json $SPECS || exit 1
cp $O/${R}_pmc_traffic.json profiles/${R}_pmc_traffic.json   # so that the bench line below carries the fresh traffic figure
echo "== default bench"
timeout -k 10 500 python3 bench.py > $O/${R}_bench_default_run.json 2> gpurun_out/bench_default.err || exit 1
cat $O/${R}_bench_default_run.json | cut -c1-300
