"""Multi-GPU plumbing: images are independent, so the batch is sharded across ranks (one process per
GPU) with NO collective on the data path.  Only two tiny control-plane exchanges exist:

  * broadcast of {key 32 B, nonce 16 B, message, seed, geometry} from the rank that owns the secrets (RCCL broadcast; the
    reference passes key_hex / nonce_hex on the command line of every process, extract.py:186-187)
  * all-gather of the recovered bitstrings (B_local x M/8 bytes per rank) and an all-reduce of the matched-bit count
    (the reference's "Average Bit Accuracy" roll-up, extract.py:157-163)

Backend-agnostic: "nccl" (= RCCL over xGMI on ROCm) uses device tensors, "gloo" (CPU tests) host tensors.
The in-kernel RNG is addressed by GLOBAL image index, so results do not depend on the number of GPUs.
"""
from __future__ import annotations

import struct
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

_HDR = struct.Struct("<32s16sQiiI")  # key, nonce, seed, height, width, message length in bytes
_MAX_MSG = 4096


def _on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _comm_device() -> torch.device:
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def rank_world() -> Tuple[int, int]:
    """(rank, world size) of the process group, (0, 1) without one."""
    return (dist.get_rank(), dist.get_world_size()) if _on() else (0, 1)


def gather_objects(obj) -> list:
    """All-gather one small picklable object per rank -> [object of rank 0, ..., object of rank world-1] on every rank (control plane only:
    the per-window outcome text of the directory harness).  Identity without a process group."""
    if not _on():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def shard_range(total: int, rank: Optional[int] = None, world: Optional[int] = None) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` images owned by `rank`; remainders go to the lowest ranks."""
    if rank is None:
        rank = dist.get_rank() if _on() else 0
    if world is None:
        world = dist.get_world_size() if _on() else 1
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_params(p: Dict) -> bytes:
    msg = p["message"]
    if len(msg) > _MAX_MSG:
        raise ValueError("message too long to broadcast")
    return _HDR.pack(p["key"], p["nonce"], int(p.get("seed", 0)), int(p.get("height", 512)), int(p.get("width", 512)), len(msg)) + msg


def unpack_params(buf: bytes) -> Dict:
    key, nonce, seed, h, w, ml = _HDR.unpack_from(buf)
    return {"key": key, "nonce": nonce, "seed": seed, "height": h, "width": w, "message": bytes(buf[_HDR.size:_HDR.size + ml])}


def broadcast_params(params: Optional[Dict], src: int = 0) -> Dict:
    """One fixed-size broadcast (header + message, <= 4.2 KiB) from `src`; identity without a process group."""
    if not _on():
        if params is None:
            raise ValueError("params required on a single process")
        return unpack_params(pack_params(params))
    dev = _comm_device()
    size = _HDR.size + _MAX_MSG
    if dist.get_rank() == src:
        raw = pack_params(params)
        t = torch.frombuffer(bytearray(raw + b"\0" * (size - len(raw))), dtype=torch.uint8).to(dev)
    else:
        t = torch.empty(size, dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    return unpack_params(t.cpu().numpy().tobytes())


def gather_bits(bits: torch.Tensor, async_op: bool = False):
    """All-gather the per-rank recovered messages [B_local, M/8] -> [world, B_local, M/8] (equal B_local on every rank)."""
    if not _on():
        return bits.unsqueeze(0) if not async_op else (bits.unsqueeze(0), None)
    dev = _comm_device()
    src = bits.contiguous().to(dev)
    out = torch.empty((dist.get_world_size(), *src.shape), dtype=src.dtype, device=dev)
    h = dist.all_gather_into_tensor(out.view(-1), src.view(-1), async_op=async_op)
    return (out, h) if async_op else out


def reduce_accuracy(matched_bits: torch.Tensor, total_bits: int) -> float:
    """Global bit accuracy = sum over ranks of matched bits / sum of compared bits."""
    m = matched_bits.sum().to(torch.int64).reshape(1)
    t = torch.tensor([total_bits], dtype=torch.int64, device=m.device)
    if _on():
        dev = _comm_device()
        m, t = m.to(dev), t.to(dev)
        dist.all_reduce(m)
        dist.all_reduce(t)
    return float(m.item()) / float(t.item())


# =====================================================================================================================
# Preflight of the multi-GPU control plane (`bench.py --gpus N --preflight`, `python -m gswm_amd.extract --gpus N --preflight`): every rank
# checks, in this order and each under a hard time limit, (1) the launcher environment, (2) that its GPU exists, (3) the rendezvous / RCCL
# init, (4) a device-tensor broadcast, (5) all_gather_into_tensor, (6) an all-reduce -- exactly the collectives the product path uses
# (broadcast_params, gather_bits, reduce_accuracy).  A stage that fails or hangs ends the rank with exit code 3 and ONE line naming the
# failure class; the launcher (launch.spawn_ranks) then stops the sibling ranks.  Nothing here replaces the process: a hang is ended by
# os._exit from a watchdog thread of the rank itself.
# =====================================================================================================================
PREFLIGHT_EXIT = 3
_PREFLIGHT_HINTS = {
    "launcher": "RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT missing or inconsistent: start the ranks with `--gpus N` (self-launch) or torch.distributed.run",
    "devices": "this rank's GPU is not visible: check HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES and that the node has --gpus devices",
    "rendezvous": "init_process_group did not complete: MASTER_ADDR must be 127.0.0.1 on one node; a stale process may hold MASTER_PORT; for RCCL across "
                  "processes this driver needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC; without it hipIpcGetMemHandle fails with 'invalid argument')",
    "broadcast": "the first device collective failed: RCCL could not move data between the GPUs (xGMI / IPC); see NCCL_DEBUG=INFO",
    "all_gather": "all_gather_into_tensor failed or returned wrong ranks",
    "all_reduce": "all_reduce failed or returned a wrong sum",
}


class PreflightError(RuntimeError):
    def __init__(self, stage: str, detail: str):
        super().__init__(f"{stage}: {detail} -- {_PREFLIGHT_HINTS.get(stage, '')}")
        self.stage = stage


def preflight(backend: str = "nccl", timeout_s: float = 120.0, init: bool = True) -> Dict:
    """Run the six stages on this rank (see above) -> {"rank", "world", "backend", "ranks_seen", "stage_ms"}; raises PreflightError.  With
    `timeout_s` > 0 a stage that hangs ends the process with PREFLIGHT_EXIT.  init=False: the process group already exists."""
    import os
    import sys
    import threading
    import time
    from datetime import timedelta

    stage_box = {"name": "launcher", "t0": time.monotonic()}
    rank_txt = os.environ.get("RANK", "?")
    done = threading.Event()

    def watchdog():
        while not done.wait(0.25):
            if timeout_s > 0 and time.monotonic() - stage_box["t0"] > timeout_s:
                print(f"[gswm preflight] rank {rank_txt}: {stage_box['name']}: no progress for {timeout_s:.0f} s -- "
                      f"{_PREFLIGHT_HINTS.get(stage_box['name'], '')}", file=sys.stderr, flush=True)
                os._exit(PREFLIGHT_EXIT)

    threading.Thread(target=watchdog, daemon=True).start()
    stage_ms: Dict[str, float] = {}

    def enter(name):
        now = time.monotonic()
        stage_ms[stage_box["name"]] = (now - stage_box["t0"]) * 1e3
        stage_box["name"], stage_box["t0"] = name, now

    try:
        try:
            world, rank, local = (int(os.environ[k]) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"))
            if not (0 <= rank < world) or not os.environ.get("MASTER_ADDR") or not os.environ.get("MASTER_PORT"):
                raise KeyError("MASTER_ADDR / MASTER_PORT")
        except (KeyError, ValueError) as exc:
            raise PreflightError("launcher", f"bad environment ({exc})") from None
        enter("devices")
        if backend == "nccl":
            n = torch.cuda.device_count()
            if local >= n:
                raise PreflightError("devices", f"LOCAL_RANK {local} but {n} visible device(s)")
            torch.cuda.set_device(local)
        enter("rendezvous")
        if init and not dist.is_initialized():
            try:
                kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
                dist.init_process_group(backend, timeout=timedelta(seconds=max(30.0, timeout_s)), **kw)
            except Exception as exc:  # noqa: BLE001
                raise PreflightError("rendezvous", f"{type(exc).__name__}: {exc}") from None
        dev = _comm_device()
        enter("broadcast")
        try:
            t = torch.arange(64, dtype=torch.int64, device=dev) * 3 + 1 if rank == 0 else torch.zeros(64, dtype=torch.int64, device=dev)
            dist.broadcast(t, src=0)
            ok = bool((t.cpu() == torch.arange(64, dtype=torch.int64) * 3 + 1).all())
        except Exception as exc:  # noqa: BLE001
            raise PreflightError("broadcast", f"{type(exc).__name__}: {exc}") from None
        if not ok:
            raise PreflightError("broadcast", "payload arrived corrupted")
        enter("all_gather")
        try:
            out = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(out, torch.tensor([rank], dtype=torch.int64, device=dev))
            seen = [int(v) for v in out.cpu().tolist()]
        except Exception as exc:  # noqa: BLE001
            raise PreflightError("all_gather", f"{type(exc).__name__}: {exc}") from None
        if seen != list(range(world)):
            raise PreflightError("all_gather", f"ranks seen {seen}, expected {list(range(world))}")
        enter("all_reduce")
        try:
            s = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
            dist.all_reduce(s)
            total = int(s.cpu().item())
        except Exception as exc:  # noqa: BLE001
            raise PreflightError("all_reduce", f"{type(exc).__name__}: {exc}") from None
        if total != world * (world + 1) // 2:
            raise PreflightError("all_reduce", f"sum {total}, expected {world * (world + 1) // 2}")
        enter("done")
    finally:
        done.set()
    return {"rank": rank, "world": world, "backend": backend, "ranks_seen": seen, "stage_ms": {k: round(v, 2) for k, v in stage_ms.items()}}


def preflight_main(backend: str, timeout_s: float = 120.0) -> int:
    """What `--preflight` runs on every rank: the report as ONE JSON line from rank 0 (exit 0), or one line naming the failure class (exit 3)."""
    import json
    import os
    import sys
    try:
        rep = preflight(backend, timeout_s)
    except PreflightError as exc:
        print(f"[gswm preflight] rank {os.environ.get('RANK', '?')}: {exc}", file=sys.stderr, flush=True)
        return PREFLIGHT_EXIT
    if rep["rank"] == 0:
        print(json.dumps({"preflight": True, "n_gpus": rep["world"], "backend": backend, "ranks_seen": rep["ranks_seen"], "stage_ms": rep["stage_ms"],
                          "collectives": ["broadcast", "all_gather_into_tensor", "all_reduce"]}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0
