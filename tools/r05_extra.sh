#!/bin/bash
# round 5, final kernels: BASELINE configs[3] as written, the north star's 50 000-key variant, and K1 launched per 64 images
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05/extra; mkdir -p $o
F="--no-cpu-baseline --no-estimate-pose --no-f32-step --no-screened-step"
timeout -k 10 500 python bench.py --object revolution --keys 50000 --itr 4096 --confidence 1 --steps 5 $F > $o/bench_config4.json 2> $o/bench_config4.err || { tail -20 $o/bench_config4.err; exit 1; }
python tools/bench_brief.py < $o/bench_config4.json || true
timeout -k 10 500 python bench.py --keys 50000 --steps 10 $F > $o/bench_50k_keys.json 2> $o/bench_50k_keys.err || { tail -20 $o/bench_50k_keys.err; exit 1; }
python tools/bench_brief.py < $o/bench_50k_keys.json || true
: > $o/group_ab.txt
for rep in 1 2 3 4 5; do
  for g in 32 64; do
    python bench.py --steps 16 --group $g $F --no-parity-check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('group $g rep $rep: %.1f images/s  %.2f ms/step  K1 in step %.2f ms per launch  alone %.2f' % (d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['alone']['ms_per_launch']))" >> $o/group_ab.txt
  done
done
cat $o/group_ab.txt
