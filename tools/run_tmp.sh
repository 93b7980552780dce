set -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_gpu_headline.py tests/test_gpu_egnn.py -m gpu -x -q -s --timeout 400 --timeout-method thread > gpurun_out/t3_tests.log 2>&1 || { tail -40 gpurun_out/t3_tests.log; exit 1; }
grep -E "checked|passed|through_flips" gpurun_out/t3_tests.log | cut -c1-400
