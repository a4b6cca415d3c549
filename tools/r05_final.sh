#!/bin/bash
# round 5: the records kept under profiles/ — full GPU test suite, the bench line, the rocprofv3 kernel table of the same command,
# RCCL at world size 1, four gloo ranks sharing the GPU
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05/final; mkdir -p $o
timeout -k 10 900 python -m pytest tests -m gpu -q > $o/gpu_tests.txt 2>&1 || true
tail -5 $o/gpu_tests.txt
timeout -k 10 700 python bench.py --steps 20 > $o/bench_n1.json 2> $o/bench_n1.err || { tail -20 $o/bench_n1.err; exit 1; }
python tools/bench_brief.py < $o/bench_n1.json || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --steps 16 --no-cpu-baseline --no-parity-check --no-estimate-pose --no-f32-step --no-screened-step > $o/bench_under_rocprof.json 2> $o/bench_under_rocprof.err || { tail -20 $o/bench_under_rocprof.err; exit 1; }
f=$(ls $o/prof/*/*kernel_stats.csv | head -1); cp "$f" $o/kernel_stats.csv; rm -rf $o/prof
python tools/kstats.py $o/kernel_stats.csv 8 || true
ISR_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 400 python bench.py --steps 6 --no-cpu-baseline --no-estimate-pose --no-f32-step --no-screened-step > $o/bench_rccl_1rank.json 2> $o/bench_rccl_1rank.err || { tail -20 $o/bench_rccl_1rank.err; exit 1; }
python tools/bench_brief.py < $o/bench_rccl_1rank.json || true
ISR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 4 --images 64 --steps 3 --no-estimate-pose --no-f32-step --no-screened-step > $o/bench_gloo_4ranks.json 2> $o/bench_gloo_4ranks.err || { tail -30 $o/bench_gloo_4ranks.err; exit 1; }
python tools/bench_brief.py < $o/bench_gloo_4ranks.json || true
