"""CPU: `python bench.py --gpus 2` from a bare shell starts its own ranks (no torch.distributed.run around it), rendezvous over 127.0.0.1
and reports every rank -- exercised here with the gloo backend (the GPU runs use RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GSW_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and sorted(out["ranks_seen"]) == [0, 1]


def test_bench_refuses_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", GSW_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2


def _env_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GSW_BENCH_BACKEND"] = "gloo"
    env["GSW_DIST_BACKEND"] = "gloo"
    return env


def test_bench_preflight_two_ranks():
    """`bench.py --gpus 2 --preflight`: every rank runs device check -> rendezvous -> broadcast -> all_gather_into_tensor -> all_reduce, rank 0 prints ONE line"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=_env_gloo(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["preflight"] is True and out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["backend"] == "gloo"
    assert set(out["stage_ms"]) >= {"launcher", "devices", "rendezvous", "broadcast", "all_gather", "all_reduce"}


def test_extract_preflight_two_ranks():
    env = _env_gloo()
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-m", "gswm_amd.extract", "--gpus", "2", "--preflight", "--key_hex", "00" * 32, "--nonce_hex", "", "--original_message_hex", "00"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["preflight"] is True and out["ranks_seen"] == [0, 1]


def test_preflight_names_the_failure_class():
    """a rank whose rendezvous cannot complete (nobody listens on MASTER_PORT, world of two with one process) is ended by the watchdog with exit code 3
    and a line naming the stage; a broken launcher environment is reported as such"""
    env = dict(_env_gloo(), RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29599", GSW_PREFLIGHT_TIMEOUT_S="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert "[gswm preflight] rank 1: rendezvous" in r.stderr
    env = dict(_env_gloo(), RANK="5", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29598")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--preflight"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "launcher" in r.stderr
