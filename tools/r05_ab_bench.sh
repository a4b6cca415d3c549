#!/bin/bash
# the bench step alternated over variants on one box: bash tools/r05_ab_bench.sh "<name>|<lib or empty>|<extra bench flags>" ...
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_bench.txt
: > $out
for rep in 1 2 3; do
  for v in "$@"; do
    IFS='|' read -r name lib flags <<< "$v"
    [ -n "$lib" ] && lib="$GRAFT_REPO_ROOT/ab_tmp/$lib.so"
    ISR_HIP_LIB=$lib python bench.py --steps 12 --no-cpu-baseline --no-parity-check --no-estimate-pose --no-f32-step $flags 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name rep $rep: %.1f images/s  %.2f ms/step  K1 in step %.2f ms  alone %.2f  final chamfer %.6f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['alone']['ms_per_launch'], d['final_chamfer']))" >> $out
  done
done
cat $out
