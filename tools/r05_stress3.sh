set -e
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/stress3.txt
for s in 4 5 6 7 8 9; do timeout -k 10 330 python tests/stress_gpu.py $s 60 >> gpurun_out/r05/stress3.txt 2>&1 || { tail -8 gpurun_out/r05/stress3.txt; exit 1; }; done
for s in 6 7 8 9 10; do timeout -k 10 200 python tools/stress_corr_screened.py $s 120 >> gpurun_out/r05/stress3.txt 2>&1 || { tail -5 gpurun_out/r05/stress3.txt; exit 1; }; done
grep "^seed" gpurun_out/r05/stress3.txt
