"""bench.py's launch contract (RetinaNet.py:105-116 becomes one process per GPU): `python bench.py --gpus N` alone starts its own N
ranks as a child torch.distributed.run before it touches torch or the GPU; under a launcher, WORLD_SIZE must equal --gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_mismatch_exits_nonzero_before_any_gpu_work():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2
    assert "--gpus 4 but WORLD_SIZE=2" in r.stderr
    r = subprocess.run([sys.executable, BENCH], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 2                      # the default --gpus 1 under a 2-rank launcher is a mismatch too


def test_self_launch_builds_the_driver_command(monkeypatch, tmp_path):
    """--gpus 3 without WORLD_SIZE: the parent must call `python -m torch.distributed.run --nproc-per-node 3 ... bench.py <same args>`
    as a child and return its code, without importing torch itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", [BENCH, "--gpus", "3", "--steps", "5", "--warmup", "1"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "3"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [BENCH, "--gpus", "3", "--steps", "5", "--warmup", "1"]


@pytest.mark.gpu
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """The rehearsal of the N > 1 launch on the one-GPU box: both ranks on device 0, gloo instead of RCCL (which refuses two ranks
    on one device).  The line must say n_gpus == 2 and count both ranks' images."""
    env = _env(RTN_BENCH_SHARE_GPU="1", RTN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert abs(j["value"] - 2 * 8 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
