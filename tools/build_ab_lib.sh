#!/bin/bash
# An alternative libisr_hip.so for tools/ab_lib.sh: bash tools/build_ab_lib.sh <name> <source.hip> <extra hipcc flags ...>
# builds ab_tmp/<name>.so = the tree's objects with <source.hip> recompiled under the extra flags (run after the tree's build).
set -eo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"; pkg="$root/imagesequenceregistrationfor6dposeestimationlabeling_amd"
name="$1"; src="$2"; shift 2
mkdir -p "$root/ab_tmp"
extra=""; [ "$src" = corr_argmax.hip ] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form \
  -Wno-unused-function $extra "$@" -I"$root/include" -c "$pkg/csrc/$src" -o "$root/ab_tmp/$name.o"
objs=$(ls "$pkg"/csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/ab_tmp/$name.so" $objs "$root/ab_tmp/$name.o"
echo "$root/ab_tmp/$name.so"
