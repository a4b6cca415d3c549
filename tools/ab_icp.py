"""A/B on one box: the ICP loop with exact near-tie resolution (this tree) against the round-2 loop (f32 winners),
alternated — tools/ab/libisr_old_icp.so is this tree's library with nn_batched.hip of commit e96ce4a linked in.  The
library is not kept in the tree (tools/ab/ is git- and gpurun-ignored); rebuild it with
    mkdir -p tools/ab/src && git show e96ce4a:imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/nn_batched.hip > tools/ab/src/nn_batched.hip
    git show e96ce4a:imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/nn_grid.hpp > tools/ab/src/nn_grid.hpp
    cp imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/isr_common.hpp tools/ab/src/
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form \
          -Iinclude -c tools/ab/src/nn_batched.hip -o tools/ab/nn_old.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libisr_old_icp.so tools/ab/nn_old.o \
          $(ls imagesequenceregistrationfor6dposeestimationlabeling_amd/csrc/build/*.o | grep -v nn_batched)
and remove tools/ab/ from .gpurunignore for the run (it is listed there so that stale binaries do not travel)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
old = os.path.join(ROOT, "tools", "ab", "libisr_old_icp.so")
for rep in range(3):
    for name, lib in (("exact (this tree)", ""), ("f32 winners (round 2)", old)):
        env = dict(os.environ)
        if lib:
            env["ISR_HIP_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_icp.py")], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if "default" in l]
        print(f"{name:24s}", line[0] if line else out.stderr[-300:], flush=True)
