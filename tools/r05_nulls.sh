#!/bin/bash
# what the side work costs the step now (null experiments, timing only): the step without the verification (a13-a15), without the
# Gauss-Newton refit, without both; alternated three times on one box
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
out=gpurun_out/r05/nulls.txt
: > $out
VARS=("$@")
[ ${#VARS[@]} -eq 0 ] && VARS=("full|" "noverify|--ablate noverify" "norefit|--refine-iters 0" "neither|--ablate noverify --refine-iters 0")
F="--steps 12 --no-cpu-baseline --no-parity-check --no-estimate-pose --no-f32-step --no-screened-step"
for rep in 1 2 3; do
  for v in "${VARS[@]}"; do
    IFS='|' read -r name flags <<< "$v"
    python bench.py $F $flags 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name rep $rep: %.1f images/s  %.2f ms/step  K1 in step %.2f ms per launch  alone %.2f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['alone']['ms_per_launch']))" >> $out
  done
done
cat $out
