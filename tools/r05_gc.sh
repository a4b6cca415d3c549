set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
for mode in off on; do
  rm -rf gpurun_out/r05/prof_gc
  ISR_BENCH_GC=$mode timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05/prof_gc -- python3 bench.py --steps 12 --no-cpu-baseline --no-parity-check --no-estimate-pose --no-f32-step --no-screened-step > gpurun_out/r05/gc_$mode.json 2> gpurun_out/r05/gc_$mode.err || { tail -20 gpurun_out/r05/gc_$mode.err; exit 1; }
  f=$(ls gpurun_out/r05/prof_gc/*/*kernel_trace.csv | head -1)
  echo "== cyclic GC $mode in the timed region"
  python3 tools/side_budget.py "$f" 12 | grep "duty\|all gaps"
  rm -rf gpurun_out/r05/prof_gc
done
bash tools/r05_ab_bench.sh "gc_off||--no-screened-step" "gc_on||--no-screened-step"
