set -e
timeout -k 10 500 python -u -m pytest tests/test_gpu_regions.py tests/test_gpu_egnn.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -5
for spec in "--workload modelnet --layer egnn" "--workload modelnet --first-layer" "--workload modelnet"; do
echo "== $spec graph"; timeout -k 10 120 python bench.py $spec --steps 300 --warmup 30 --hip-graph --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['verified']['ok'])"
done
timeout -k 10 300 python -u tools/train_bench.py 2>&1 | grep -i "modelnet"
