set -eo pipefail
cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r05/last; mkdir -p $o
true
true
ISR_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 2 --images 32 --steps 3 --no-estimate-pose --no-cpu-baseline > $o/bench_gloo_2ranks.json 2> $o/bench_gloo_2ranks.err || { tail -30 $o/bench_gloo_2ranks.err; exit 1; }
python tools/bench_brief.py < $o/bench_gloo_2ranks.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
