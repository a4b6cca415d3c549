#!/bin/bash
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT = the snapshot root)}"
# HBM traffic from PMC counters (separate passes, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not
# fit one pass) for K1 at the bench's launch sizes and for the brute-force NN shape of roofline_nn; writes
# profiles/k1_hbm_traffic.json and profiles/nn_hbm_traffic.json.   bash tools/pmc_all.sh <scratch dir>
set -u
out=${1:-gpurun_out/pmc_r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for tag in "k1_1 tools/time_corr.py 307200 20000 64" "k1_16 tools/time_corr.py 4915200 20000 64" "k1_32 tools/time_corr.py 9830400 20000 64" "nn tools/time_nn_brute.py 20000 32"; do
  set -- $tag; name=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/$name/fetch" -- python3 "$@" > "$out/$name.fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$out/$name/write" -- python3 "$@" > "$out/$name.write.log" 2>&1
done
python3 tools/pmc_to_json.py "$out"
