"""CPU, world_size 2, gloo: the control-plane exchanges of the multi-GPU path (parameter broadcast, gather of recovered
bitstrings, accuracy reduction) and the shard arithmetic that makes results independent of the GPU count."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import README_KEY, README_NONCE, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import gswm_amd
    from gswm_amd import dist as gdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        secrets = {"key": bytes.fromhex(README_KEY), "nonce": bytes.fromhex(README_NONCE), "message": b"lthero" + b"\0" * 26,
                   "seed": 2024, "height": 512, "width": 512}
        got = gdist.broadcast_params(secrets if rank == 0 else None, src=0)       # only rank 0 knows the secrets
        assert got == secrets
        total = 9
        lo, hi = gdist.shard_range(total)
        # each rank "recovers" its shard: fake bits = global image index in every byte, padded to equal shard size
        per = (total + world - 1) // world
        bits = torch.zeros((per, 32), dtype=torch.uint8)
        for i, g in enumerate(range(lo, hi)):
            bits[i] = g + 1
        allb = gdist.gather_bits(bits)
        assert tuple(allb.shape) == (world, per, 32)
        seen = sorted(int(v) for v in allb[:, :, 0].reshape(-1) if v)
        assert seen == list(range(1, total + 1))
        out, h = gdist.gather_bits(bits, async_op=True)
        h.wait()
        assert torch.equal(out, allb)
        # accuracy: rank r matched 250 + r of 256 bits on each of its images
        n_img = hi - lo
        acc = gdist.reduce_accuracy(torch.full((n_img,), 250 + rank, dtype=torch.int32), n_img * 256)
        want = sum((gdist.shard_range(total, r, world)[1] - gdist.shard_range(total, r, world)[0]) * (250 + r) for r in range(world)) / (total * 256)
        assert abs(acc - want) < 1e-12
        q.put((rank, "ok"))
    except Exception as e:  # noqa
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_control_plane_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
