set -e
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_corr.py tests/test_gpu_corr_screened.py tests/test_gpu_digits.py tests/test_gpu_sequence.py tests/test_gpu_config3_sharded.py tests/test_gpu_config4.py -x -q > gpurun_out/r05/split_tests.txt 2>&1 || { tail -30 gpurun_out/r05/split_tests.txt; exit 1; }
tail -3 gpurun_out/r05/split_tests.txt
bash tools/r05_ab_bench.sh "split||--no-screened-step" "one_call||--no-screened-step --k1-one-call"
