"""A/B on one box: the ICP loop with exact near-tie resolution (this tree) against the round-2 loop (f32 winners),
alternated — tools/ab/libisr_old_icp.so is this tree's library with nn_batched.hip of commit e96ce4a linked in."""
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
