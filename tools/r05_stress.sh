set -e
mkdir -p gpurun_out/r05
timeout -k 10 300 python -m pytest tests/test_gpu_corr_screened.py -x -q > gpurun_out/r05/screened_tests.txt 2>&1 || { tail -30 gpurun_out/r05/screened_tests.txt; exit 1; }
tail -3 gpurun_out/r05/screened_tests.txt
rm -f gpurun_out/r05/stress.txt
for s in 1 2 3 4 5; do timeout -k 10 200 python tools/stress_corr_screened.py $s 120 >> gpurun_out/r05/stress.txt 2>&1 || { tail -5 gpurun_out/r05/stress.txt; exit 1; }; done
grep "^seed" gpurun_out/r05/stress.txt
timeout -k 10 300 python tools/r05_sparse_exp.py > gpurun_out/r05/sparse_exp.txt 2>&1 || { tail -5 gpurun_out/r05/sparse_exp.txt; exit 1; }
tail -12 gpurun_out/r05/sparse_exp.txt
