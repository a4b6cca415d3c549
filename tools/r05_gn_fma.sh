set -e
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_ransac.py tests/test_gpu_config4.py tests/test_gpu_sequence.py tests/test_gpu_registration.py tests/test_gpu_golden.py tests/test_gpu_bench_size_parity.py tests/test_gpu_preprocess.py -x -q > gpurun_out/r05/gn_tests.txt 2>&1 || { tail -30 gpurun_out/r05/gn_tests.txt; exit 1; }
tail -3 gpurun_out/r05/gn_tests.txt
bash tools/r05_ab_bench.sh "gn_fma||--no-screened-step" "gn_old|gn_old|--no-screened-step"
