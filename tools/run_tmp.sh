set -o pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t2_tests.log 2>&1 || { tail -40 gpurun_out/t2_tests.log; exit 1; }
tail -3 gpurun_out/t2_tests.log
timeout -k 10 200 python -u bench.py --workload flickr --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/t2_bench_flickr.json 2> gpurun_out/t2_bench_flickr.err || { tail gpurun_out/t2_bench_flickr.err; exit 1; }
cut -c1-330 gpurun_out/t2_bench_flickr.json
timeout -k 10 300 python -u bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/t2_bench_default.json 2> gpurun_out/t2_bench_default.err || { tail gpurun_out/t2_bench_default.err; exit 1; }
cut -c1-330 gpurun_out/t2_bench_default.json
