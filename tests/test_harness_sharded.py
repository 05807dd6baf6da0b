"""CPU: the directory harness (extract.py:120-163 twin) streams its work in bounded windows, writes each directory's result.txt as soon as the
directory is complete, and -- under a process group (gloo here, RCCL on the GPUs) -- shards the images over the ranks with result files
byte-equal to the single-process run.  The device stages are stubbed (no GPU in this container): the "inverted latent" of an image is a
deterministic function of its pixels, so a wrong shard, order or gather shows up as a wrong bit string."""
import os
import re
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import README_KEY, README_NONCE, ROOT

MSG_HEX = (b"lthero" + b"\0" * 26).hex()


def _make_tree(root, n_dirs=3, per_dir=5):
    from PIL import Image
    rng = np.random.RandomState(7)
    for d in range(n_dirs):
        p = os.path.join(root, f"attack_{d}")
        os.makedirs(p)
        for i in range(per_dir + d):
            Image.fromarray(rng.randint(0, 256, (16, 16, 3), dtype=np.uint8)).save(os.path.join(p, f"img_{i}.png"))
        if d == 1:
            with open(os.path.join(p, "broken.png"), "wb") as f:       # a file that does not decode fails alone
                f.write(b"not a png")
    os.makedirs(os.path.join(root, "empty"))


def _args(root):
    return types.SimpleNamespace(model_id="no/such/checkpoint", images_directory_path=root, single_image_path="", key_hex=README_KEY, nonce_hex=README_NONCE,
                                 key=bytes.fromhex(README_KEY), nonce=bytes.fromhex(README_NONCE), original_message_hex=MSG_HEX, num_inference_steps=5,
                                 scheduler="DDIM", is_traverse_subdirectories=1, l=1, width=16, height=16, message_length=256, allow_synthetic_weights=True,
                                 batch_size=2, strict_kernels=0, gpus=1)


def _install_stubs(E, log=None):
    """device stages -> host stand-ins: latent = per-image pixel statistics, bits = 256 characters derived from them"""
    def invert(arrs, args, device="cuda"):
        if log is not None:
            log.append(len(arrs))
        for a in arrs:
            if int(a[0, 0, 0]) % 11 == 0:
                raise ValueError("stub: this image poisons its batch")      # exercises the redo-one-by-one path
        return torch.tensor([[float(a.astype(np.int64).sum() % 65521)] for a in arrs])

    def recover(latents, args):
        return [format(int(v.item()) * 2654435761 % (1 << 64), "064b") * 4 for v in latents]

    E.invert_decoded_images = invert
    E.recover_exactracted_message_batch = recover
    E.load_models = lambda *a, **k: None


def _snapshot(root):
    out = {}
    for here, _, files in os.walk(root):
        for f in files:
            if f == "result.txt":
                with open(os.path.join(here, f)) as fh:
                    out[os.path.relpath(os.path.join(here, f), root)] = re.sub(r"Time,[^\n]*", "Time,<stamp>", fh.read()).replace(root, "<root>")
    return out


def test_single_process_streams_in_windows_and_flushes_directories_as_they_complete(tmp_path, capsys):
    import gswm_amd
    from gswm_amd import extract as E
    import importlib
    E = importlib.reload(E)
    root = str(tmp_path / "tree")
    _make_tree(root)
    batches = []
    _install_stubs(E, batches)
    seen_after_window = []
    orig = E._Reporter.flush

    def spying_flush(self):
        orig(self)
        seen_after_window.append(sorted(_snapshot(root)))

    E._Reporter.flush = spying_flush
    E.process_directory(_args(root))
    snap = _snapshot(root)
    assert sorted(snap) == ["attack_0/result.txt", "attack_1/result.txt", "attack_2/result.txt", "result.txt"]
    assert max(b for b in batches) <= 2                                         # device batches respect --batch_size
    # the first directory's file exists long before the run ends (append-as-you-go), the last one only at the end
    first = next(i for i, s in enumerate(seen_after_window) if "attack_0/result.txt" in s)
    last = next(i for i, s in enumerate(seen_after_window) if "attack_2/result.txt" in s)
    assert first < last
    assert "Error processing" in snap["attack_1/result.txt"] and "broken.png" in snap["attack_1/result.txt"]
    assert "SYNTHETIC WEIGHTS" in snap["attack_0/result.txt"]
    text = capsys.readouterr().out
    assert text.index("=" * 20 + root + "=" * 20) < text.index("Bit Accuracy")       # the walk's banner precedes the directories it lists
    importlib.reload(E)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import gswm_amd
    from gswm_amd import extract as E
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts = []
        _install_stubs(E, counts)
        E.process_directory(_args(root))
        q.put((rank, "ok", sum(counts)))
    except Exception as e:  # noqa
        import traceback
        q.put((rank, traceback.format_exc(), 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_two_ranks_write_the_same_bytes_as_one_process(tmp_path):
    import gswm_amd
    from gswm_amd import extract as E
    import importlib
    E = importlib.reload(E)
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    _make_tree(one)
    _make_tree(two)
    _install_stubs(E)
    E.process_directory(_args(one))
    importlib.reload(E)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, two, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r[:2] for r in res) == [(0, "ok"), (1, "ok")], res
    done = sorted(r[2] for r in res)
    assert done[0] > 0 and abs(done[0] - done[1]) <= 6                          # both ranks did a share of the device work
    assert _snapshot(one) == _snapshot(two)


def test_dpms_scheduler_is_marked_parity_unpinned(tmp_path):
    import gswm_amd
    from gswm_amd import extract as E
    import importlib
    E = importlib.reload(E)
    root = str(tmp_path / "tree")
    _make_tree(root, n_dirs=1, per_dir=2)
    _install_stubs(E)
    a = _args(root)
    a.scheduler = "DPMs"
    E.process_directory(a)
    assert "PARITY UNPINNED" in _snapshot(root)["attack_0/result.txt"]
    a.scheduler = "DDIM"
    importlib.reload(E)


def test_strict_kernels_raise_by_default_and_the_opt_out_warns_and_counts():
    import gswm_amd
    from gswm_amd import unet as U, vae as V
    assert U.STRICT is True and V.STRICT is True                       # round 6: leaving the hand-written path raises unless the caller opts out
    with pytest.raises(RuntimeError, match="strict kernels"):
        U._note_fallback("linear K=100 N=7: library GEMM")
    U.STRICT = False
    try:
        U.FALLBACKS.clear()
        with pytest.warns(RuntimeWarning, match="library GEMM"):
            U._note_fallback("linear K=100 N=7: library GEMM")
        assert U.FALLBACKS == {"linear K=100 N=7: library GEMM": 1}
    finally:
        U.STRICT = True
        U.FALLBACKS.clear()
