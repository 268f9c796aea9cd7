"""bench.py's command line around --gpus N: the multi-GPU points of the scaling curve are the driver's to
run; what this repo owns is that the command cannot quietly fall back to one rank and still print a line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SRRG2_BENCH_SHARE_GPU")}
    env.update(over)
    return env


def test_gpus_n_without_a_launcher_starts_n_ranks_itself():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1", "--print-launch"],
                         capture_output=True, text=True, timeout=120, env=_env(), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the ranks get the same command line


def test_a_launcher_with_another_world_size_is_refused():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, timeout=120,
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True, timeout=120,
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bare_gpus_2_runs_two_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` with NO launcher: (i) on a one-GPU box it must fail -- never print an n_gpus = 1 line;
    (ii) under the dry-run hook (both ranks on device 0, gloo) it prints ONE line with n_gpus == 2 that says it was a dry run"""
    import torch

    args = [sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--total-alignments", "8",
            "--batch-points", "8000", "--no-one-gpu-reference"]
    if torch.cuda.device_count() < 2:
        out = subprocess.run(args, capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
        assert out.returncode != 0
        assert "HIP device" in (out.stderr + out.stdout)
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run(args, capture_output=True, text=True, timeout=600, env=_env(SRRG2_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["shared_gpu_dry_run"] is True and rec["config"]["alignments_per_step_by_rank"] == [4, 4]
