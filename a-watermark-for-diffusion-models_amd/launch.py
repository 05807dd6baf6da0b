"""One process per GPU from a bare shell: `bench.py --gpus N` and `python -m gswm_amd.extract --gpus N` start their own ranks (RCCL / gloo
rendezvous over 127.0.0.1) when no launcher is around them.  Pure standard library: the parent never imports torch and never touches HIP,
and no process ever replaces itself -- the ranks are plain children, polled, and torn down by PID when one of them fails."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional

LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def under_launcher() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def _free_port() -> int:
    with socket.socket() as sk:
        sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _teardown(procs: List[subprocess.Popen], grace: float = 5.0) -> None:
    """Stop exactly the children we started (never by pattern): SIGTERM, a grace period, then SIGKILL."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_end = time.monotonic() + grace
    for p in procs:
        if p.poll() is None:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    for p in procs:
        if p.poll() is None:
            p.wait()


def spawn_ranks(cmd: List[str], nprocs: int, *, timeout: Optional[float] = None, poll: float = 0.2, env_extra: Optional[Dict[str, str]] = None) -> int:
    """Run `cmd` as `nprocs` ranks (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) and return 0 when all of them exit 0.
    The first rank that fails ends the job: its siblings -- which would otherwise sit in a rendezvous or a collective until the RCCL
    timeout -- are terminated and its exit code is returned; `timeout` seconds (None = no limit) bound the whole job (exit code 124)."""
    port = _free_port()
    base = {k: v for k, v in os.environ.items() if k not in LAUNCH_ENV}
    # dmabuf IPC: this image's host driver supports no legacy IPC handles -- without the setting RCCL / device-tensor sharing across processes fails with
    # `hipIpcGetMemHandle: invalid argument` (the build environment exports it for the same reason; setdefault: an explicit value of the caller wins).
    # `--preflight` (dist.preflight) is the check that it worked: RCCL init + a device broadcast + all_gather_into_tensor + all_reduce per rank.
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if env_extra:
        base.update(env_extra)
    procs: List[subprocess.Popen] = []
    try:
        for r in range(nprocs):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nprocs), LOCAL_WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen(cmd, env=env))
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                print(f"[gswm launch] a rank exited with code {bad[0]}: stopping the other ranks", file=sys.stderr)
                return abs(bad[0]) or 1
            if all(c == 0 for c in codes):
                return 0
            if deadline is not None and time.monotonic() > deadline:
                print(f"[gswm launch] {timeout:.0f} s limit reached: stopping all ranks", file=sys.stderr)
                return 124
            time.sleep(poll)
    finally:
        _teardown(procs)
