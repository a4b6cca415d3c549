#!/bin/bash
# round 4, session 2: what bounds corr_recheck_kernel on a long list — kernel table of tools/time_corr_ties.py, tree against the timing-only
# build for three waves per SIMD (ab_tmp/rrw3.so)
# alt library: bash tools/build_ab_lib.sh rrw3 corr_argmax.hip -DISR_K1_RECHECK_WAVES=3   (rr_noexact: -DISR_ABL_RECHECK_NOEXACT, timing only)
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s2
for lib in tree rrw3; do
  [ $lib = tree ] && unset ISR_HIP_LIB || export ISR_HIP_LIB=$GRAFT_REPO_ROOT/ab_tmp/$lib.so
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2/prof_$lib -- python3 tools/time_corr_ties.py > gpurun_out/s2/ties_$lib.txt 2>&1 || { tail -5 gpurun_out/s2/ties_$lib.txt; exit 1; }
  echo "== $lib"; grep -E "^revolution" gpurun_out/s2/ties_$lib.txt
  f=$(ls gpurun_out/s2/prof_$lib/*/*kernel_stats.csv | head -1); python tools/kstats.py "$f" 6 | grep -E "corr_|total"
  rm -rf gpurun_out/s2/prof_$lib
done
