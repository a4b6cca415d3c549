set -e
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/stress2.txt
for s in 1 2 3; do timeout -k 10 330 python tests/stress_gpu.py $s 60 >> gpurun_out/r05/stress2.txt 2>&1 || { tail -8 gpurun_out/r05/stress2.txt; exit 1; }; done
grep "^seed" gpurun_out/r05/stress2.txt
