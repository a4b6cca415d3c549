#!/bin/bash
# round 4, call 9: the vote's distance-field bounds — tests, timing
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sequence.py tests/test_gpu_config3_sharded.py tests/test_gpu_errors.py -x -q -m gpu > gpurun_out/r04_m9_tests.txt 2>&1 || { tail -40 gpurun_out/r04_m9_tests.txt; exit 1; }
tail -3 gpurun_out/r04_m9_tests.txt
timeout -k 10 300 python tools/time_vote.py 256 5000 20000 > gpurun_out/r04_vote_bounds_256.txt 2>&1 && cat gpurun_out/r04_vote_bounds_256.txt
timeout -k 10 400 python tools/time_vote.py 512 5000 20000 > gpurun_out/r04_vote_bounds_512.txt 2>&1 && cat gpurun_out/r04_vote_bounds_512.txt
