set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/bench_all.sh r03 > gpurun_out/bench_all.log 2>&1 || { echo bench_all failed; tail -5 gpurun_out/bench_all.log; exit 1; }
( MPNN_AMD_LIB=$GRAFT_REPO_ROOT/tools/libmpnn_prof.so timeout -k 10 120 python3 tools/region_prof.py 32 && echo "--- EGNN layer" && MPNN_AMD_LIB=$GRAFT_REPO_ROOT/tools/libmpnn_prof.so timeout -k 10 120 python3 tools/region_prof.py 32 egnn ) 2>&1 | grep -v amdgpu.ids > $O/r03_region_phases.txt || exit 1
echo done
