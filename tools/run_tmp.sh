set -o pipefail
timeout -k 10 700 python -u -m pytest tests/test_partition_gloo.py tests/test_cabi.py -x -q --timeout 600 --timeout-method thread > gpurun_out/t4_tests.log 2>&1 || { tail -60 gpurun_out/t4_tests.log; exit 1; }
tail -3 gpurun_out/t4_tests.log
