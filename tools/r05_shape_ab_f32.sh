#!/bin/bash
# the f16-plane and bf16-plane f32 routes of K1: corr_bf16_direct_kernel on 32x32x16 (tree) against the ISR_ABL_MFMA16 timing
# ablation (16x16x32), per-kernel durations from rocprofv3 (the ablation's wrong logits send a million queries to the recheck,
# whose time must stay out), alternated
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
out=gpurun_out/r05/k1_mfma_shape_ab_f32.txt
: > $out
for rep in 1 2; do
  for name in tree full16; do
    lib=""; [ "$name" != tree ] && lib="$GRAFT_REPO_ROOT/ab_tmp/$name.so"
    export ISR_HIP_LIB=$lib
    for cfg in "1228800 20000 64 0" "1228800 20000 64 2" "307200 20000 128 0"; do
      rm -rf gpurun_out/r05/prof_abl
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/prof_abl -- python3 tools/time_corr_f32.py $cfg > gpurun_out/r05/prof_abl.log 2>&1
      f=$(ls gpurun_out/r05/prof_abl/*/*kernel_stats.csv | head -1)
      python3 - "$f" "$name rep $rep: P N D route = $cfg" >> $out <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'corr_bf16_direct_kernel' in r['Name']:
        print(sys.argv[2], " corr_bf16_direct_kernel<" + r['Name'].split('<')[1].split('>')[0] + ">", r['Calls'], "launches  %.3f ms" % (float(r['AverageNs'])/1e6))
PY
    done
  done
done
rm -rf gpurun_out/r05/prof_abl
cat $out
