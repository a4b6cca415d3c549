"""Build libisr_hip.so in-tree with hipcc for gfx950.

`python -m imagesequenceregistrationfor6dposeestimationlabeling_amd.build` or
`build_hip()`; __graft_entry__.build() calls the latter.  hipcc cross-compiles without a GPU.
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
OBJ_DIR = CSRC / "build"
LIB_PATH = PKG_DIR / "libisr_hip.so"

ARCH = "gfx950"
# -ffp-contract=off: every fused multiply-add in the kernels is written explicitly, so the f32 /
# f64 op order is the one the CPU oracle restates and indices / masks compare bit for bit.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    # MFMA results land in VGPRs (gfx950's register file is unified): the softmax epilogue reads
    # every accumulator three times, and AGPR accumulators cost a v_accvgpr_read per read.
    "-mllvm", "-amdgpu-mfma-vgpr-form",
    "-Wall",
    "-Wno-unused-function",
]

# Per-source additions.  corr_argmax.hip: NaN-free logits are a precondition of the path (finite
# descriptors), and without the flag every fmaxf on a raw MFMA result costs an extra
# `v_max_f32 x, x, x` (sNaN quieting) in a loop that is bound by VALU issue.
EXTRA_FLAGS = {
    "corr_argmax.hip": ["-fno-honor-nans"],
}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libisr_hip.so cannot be built")


def sources() -> list[Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(out: Path, deps: list[Path]) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False, jobs: int = 4) -> Path:
    """Compile every csrc/*.hip for gfx950 and link libisr_hip.so.  Returns the library path."""
    hipcc = _hipcc()
    OBJ_DIR.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.hpp")) + [PKG_DIR.parent / "include" / "isr_hip.h"]
    srcs = sources()
    objs = [OBJ_DIR / (s.stem + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not (force or _stale(obj, [src] + headers)):
            return None
        cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        warns = list(ex.map(compile_one, zip(srcs, objs)))
    if verbose:
        for w in warns:
            if w:
                print(w, file=sys.stderr)
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH),
               *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    p = build_hip(force="--force" in sys.argv, verbose=True)
    print(p)
