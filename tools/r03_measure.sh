set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
set -o pipefail
python -m pytest tests -m gpu -q > gpurun_out/t_final.log 2>&1; tail -3 gpurun_out/t_final.log
python bench.py --steps 20 > gpurun_out/r03_bench_n1.json 2> gpurun_out/r03_bench_n1.err && tail -c 200 gpurun_out/r03_bench_n1.json
python bench.py --steps 10 --verify vote > gpurun_out/r03_bench_vote.json 2> gpurun_out/r03_bench_vote.err
python bench.py --steps 10 --keys 50000 --itr 4096 --no-cpu-baseline > gpurun_out/r03_bench_50k.json 2> gpurun_out/r03_bench_50k.err
ISR_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 4 --images 16 --no-cpu-baseline > gpurun_out/r03_bench_gloo2.json 2> gpurun_out/r03_bench_gloo2.err
ISR_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 6 --no-cpu-baseline > gpurun_out/r03_bench_rccl1.json 2> gpurun_out/r03_bench_rccl1.err
for b in 64 128; do python tools/time_ref_shape_batched.py --batch $b > gpurun_out/r03_ref_shape_b$b.log 2>&1; done
python tools/time_ref_shape_batched.py --distinct 256 --batch 256 > gpurun_out/r03_ref_shape_b256.log 2>&1
python tools/time_ref_shape_batched.py --batch 64 --streams 3 > gpurun_out/r03_ref_shape_b64_s3.log 2>&1
python tools/time_estimate_pose.py > gpurun_out/r03_ep_time.log 2>&1
python tools/time_estimate_pose.py --avg-queries 0 >> gpurun_out/r03_ep_time.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_bench -o b -- python3 $R/bench.py --steps 16 --no-cpu-baseline --no-parity-check > $R/gpurun_out/r03_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_ep -o ep -- python3 $R/tools/time_estimate_pose.py --reps 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_crops -o crops -- python3 $R/tools/time_ref_shape_batched.py --once --dtype bf16 --batch 64 > /dev/null 2>&1
ls $R/gpurun_out/r03_prof_bench $R/gpurun_out/r03_prof_ep $R/gpurun_out/r03_prof_crops
