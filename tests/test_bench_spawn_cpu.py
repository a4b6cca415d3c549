"""CPU: `python bench.py --gpus N` started PLAINLY (not under torchrun) launches its N ranks as child
processes under torch.distributed.run and relays their result.  On a box without a HIP device every rank
must fail loudly ("needs a HIP device"), the launcher must return non-zero — and nothing may hang."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_plain_multi_gpu_invocation_spawns_ranks_and_fails_cleanly_without_a_gpu():
    import torch
    if torch.cuda.is_available():          # on a GPU box this path is exercised by the bench itself
        return
    env = dict(os.environ, ISR_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.stderr.count("needs a HIP device") >= 2, r.stderr[-2000:]
    assert not any(ln.lstrip().startswith("{") for ln in r.stdout.splitlines())      # no JSON line from a failed run


def test_child_command_of_the_eight_gpu_invocation():
    """`python bench.py --gpus 8 ...` launches `torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` — the driver's own line (the 8-rank run itself cannot be rehearsed on a
    one-GPU pool; the command that would start it can be checked)."""
    sys.path.insert(0, str(ROOT))
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.rank_command(8, 29511, argv)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == argv
