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
