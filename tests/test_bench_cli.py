"""bench.py's launch contract, without a GPU: a WORLD_SIZE that disagrees with --gpus is an error (it used to run one GPU
and report n_gpus = 1 silently), and the multi-rank self-launch goes through torch.distributed.run."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode != 0
    assert "WORLD_SIZE=3" in (res.stderr + res.stdout) and "--gpus 2" in (res.stderr + res.stdout)


def test_self_launch_builds_a_torchrun_command(monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench

    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.launch_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
