#!/bin/bash
# the bench step with K1 unscreened / screened at the bench's |k| = 5 and at |k| = 8 (SURVEY's first suggestion), one box
set -eo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
out=gpurun_out/r05/screen_bench.txt
: > $out
for tau in 5 8; do
  for k1 in log2 screened; do
    python bench.py --steps 10 --tau $tau --k1 $k1 --no-cpu-baseline --no-estimate-pose --no-f32-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']; p=d.get('parity_check') or {}
print('tau $tau --k1 $k1: %.1f images/s  %.2f ms/step  K1 %.2f ms per 32-image launch in the step (%.3f of the bf16 peak), %.2f alone; registered %s/64, final chamfer %.4f; parity: K1 rows %s/%s, pick equal %s, ICP rot %.1e rad; screen %s' % (d['value'], d['ms_per_step'], r['ms_per_launch'], r['frac'], r['alone']['ms_per_launch'], d['last_step'].get('registered_this_rank'), d['final_chamfer'], p.get('k1_idx_equal_rows'), p.get('k1_rows_checked'), p.get('pick_idx_equal'), p.get('icp_rot_rad', float('nan')), json.dumps(r.get('screen'))))" >> $out
  done
done
cat $out
