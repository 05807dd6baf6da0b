#!/opt/conda/bin/python3.9
"""Golden vectors for the DDIM arithmetic (rows X2 / G1), produced by EXECUTING the reference's own shipped bytecode.

    cd /root/repo && /opt/conda/bin/python3.9 tests/golden/make_golden_ddim.py

The reference's extract.py delegates the scheduler arithmetic to diffusers 0.26.0, which is neither vendored nor installed here; what
the reference DOES ship is the closed form it used before, as CPython 3.8 bytecode:
/root/reference/__pycache__/inverse_stable_diffusion_gs.cpython-38.pyc -- module functions `backward_ddim`, `forward_ddim` and the
method `InversableStableDiffusionPipeline.backward_diffusion` (the sampling / inversion loop: timestep order, prev_timestep,
alpha lookup with final_alpha_cumprod, the alpha swap for reverse_process, classifier-free guidance, the per-step update).

How it is run: `marshal.loads` of the .pyc payload under CPython 3.9 (the only interpreter here that is within one minor version),
then `types.FunctionType` on the code objects.
  * backward_ddim / forward_ddim use only LOAD_FAST / LOAD_CONST / LOAD_GLOBAL / BINARY_* / CALL_FUNCTION / RETURN_VALUE, whose encoding
    is identical in 3.8 and 3.9: executed UNMODIFIED.
  * backward_diffusion additionally contains `COMPARE_OP 9` ("is not", three times: `x is not None`), which 3.9 encodes as `IS_OP 1`.
    Those instructions -- and nothing else -- are re-encoded (same stack effect, same semantics); every other opcode of the function
    is checked against an allow-list of opcodes that are unchanged between 3.8 and 3.9 (byte-offset jumps included).  The count of
    re-encoded instructions is stored in the fixture.
What the bytecode does NOT contain (they are arguments / attributes it reads from diffusers objects, supplied here as stubs and stored in
the fixture as INPUTS): the timestep list (`scheduler.timesteps`: DDIM 'leading' spacing with steps_offset 1, the published SD scheduler
config), the alphas_cumprod table (scaled-linear betas 0.00085 .. 0.012) and final_alpha_cumprod (= alphas_cumprod[0],
set_alpha_to_one False), init_noise_sigma = 1, scale_model_input = identity, and the UNet (an analytic eps function restated in the tests).

Output: tests/golden/ddim_bytecode.json -- data only (inputs + the values the reference's bytecode returned).
"""
import dis
import hashlib
import json
import marshal
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
PYC = "/root/reference/__pycache__/inverse_stable_diffusion_gs.cpython-38.pyc"
# The reference tree is untrusted input: the file is pinned by hash before it is unmarshalled, the code objects may only name what the
# loop is known to use, and they run with a builtins dict of two functions.  (The opcode allow-list below is about 3.8 == 3.9 encoding,
# not a sandbox.)  Only the JSON this script writes is consumed by the tests.
PYC_SHA256 = "1fa7486a416112cc66e8043e39e27a3c786fe4007b46644e4cffa428730716e1"
ALLOWED_NAMES = {"backward_ddim": set(), "forward_ddim": {"backward_ddim"},
                 "backward_diffusion": {"scheduler", "set_timesteps", "timesteps", "to", "device", "init_noise_sigma", "enumerate", "progress_bar",
                                        "reversed", "torch", "cat", "scale_model_input", "unet", "sample", "chunk", "config", "num_train_timesteps",
                                        "num_inference_steps", "alphas_cumprod", "final_alpha_cumprod", "backward_ddim"}}
SAFE_BUILTINS = {"enumerate": enumerate, "reversed": reversed}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddim_bytecode.json")

assert sys.version_info[:2] == (3, 9), "run under /opt/conda/bin/python3.9 (3.8 bytecode, see the docstring)"

# opcodes whose number, argument meaning and stack effect are the same in CPython 3.8 and 3.9
SAME_38_39 = {"LOAD_FAST", "STORE_FAST", "LOAD_CONST", "LOAD_GLOBAL", "LOAD_ATTR", "LOAD_METHOD", "CALL_METHOD", "CALL_FUNCTION",
              "CALL_FUNCTION_KW", "POP_TOP", "ROT_TWO", "BINARY_POWER", "BINARY_MULTIPLY", "BINARY_SUBTRACT", "BINARY_ADD",
              "BINARY_TRUE_DIVIDE", "BINARY_FLOOR_DIVIDE", "BINARY_MODULO", "BINARY_SUBSCR", "POP_JUMP_IF_FALSE", "POP_JUMP_IF_TRUE",
              "JUMP_FORWARD", "JUMP_ABSOLUTE", "GET_ITER", "FOR_ITER", "UNPACK_SEQUENCE", "BUILD_LIST", "EXTENDED_ARG", "RETURN_VALUE"}


def find_code(co, name):
    for k in co.co_consts:
        if isinstance(k, types.CodeType):
            if k.co_name == name:
                return k
            r = find_code(k, name)
            if r is not None:
                return r
    return None


def port_38_to_39(code):
    """Re-encode the 3.8-only comparison forms; refuse anything else that is not on the allow-list.  Returns (code, n_reencoded)."""
    raw = bytearray(code.co_code)
    n = 0
    for i in range(0, len(raw), 2):
        op, arg = raw[i], raw[i + 1]
        name = dis.opname[op]
        if name == "COMPARE_OP":
            if arg <= 5:
                continue                       # <, <=, ==, !=, >, >= : unchanged
            if arg in (8, 9):                  # 3.8: 'is' / 'is not'  ->  3.9: IS_OP 0 / 1
                raw[i], raw[i + 1] = dis.opmap["IS_OP"], arg - 8
                n += 1
                continue
            raise SystemExit(f"not executable: COMPARE_OP {arg} at {i}")
        if name not in SAME_38_39:
            raise SystemExit(f"not executable: opcode {name} at offset {i} is not on the 3.8 == 3.9 allow-list")
    return code.replace(co_code=bytes(raw)), n


def sd_alphas_cumprod(T=1000, b0=0.00085, b1=0.012):
    betas = np.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def eps_fn(x, t):          # the analytic stand-in for the UNet (restated verbatim in the tests)
    return 0.3 * np.tanh(x) + 0.05 * np.sin(3.0 * x + 0.01 * t)


def eps_text_fn(x, t):     # conditional branch of the classifier-free-guidance pair
    return eps_fn(x, t) + 0.1 * np.cos(2.0 * x - 0.003 * t)


class Arr(np.ndarray):
    """ndarray with the one tensor method the loop calls on the model output"""

    def chunk(self, n):
        return tuple(a.view(Arr) for a in np.split(np.asarray(self), n, axis=0))


class Timesteps(list):
    def to(self, device):
        return self


def main():
    raw = open(PYC, "rb").read()
    assert hashlib.sha256(raw).hexdigest() == PYC_SHA256, "the reference's .pyc is not the file this generator was written against"
    module = marshal.loads(raw[16:])
    co_bd, co_fd = find_code(module, "backward_ddim"), find_code(module, "forward_ddim")
    co_loop = find_code(module, "backward_diffusion")
    for co in (co_bd, co_fd, co_loop):
        extra = set(co.co_names) - ALLOWED_NAMES[co.co_name]
        assert not extra, f"{co.co_name} names {sorted(extra)}: not on the allow-list"
        assert not any(isinstance(k, types.CodeType) for k in co.co_consts), "nested code objects are not expected"
    for co in (co_bd, co_fd):
        _, n = port_38_to_39(co)
        assert n == 0
    co_loop39, n_reenc = port_38_to_39(co_loop)

    g_step = {"__builtins__": dict(SAFE_BUILTINS)}
    backward_ddim = types.FunctionType(co_bd, g_step, "backward_ddim")
    g_step["backward_ddim"] = backward_ddim
    forward_ddim = types.FunctionType(co_fd, g_step, "forward_ddim")

    out = {"source": "inverse_stable_diffusion_gs.cpython-38.pyc executed under CPython 3.9", "reencoded_is_not": n_reenc,
           "step": [], "loops": []}

    # ---- single-step vectors: (x_t, alpha_t, alpha_tm1, eps) -> x'
    ac = sd_alphas_cumprod()
    rng = np.random.RandomState(7)
    for (ta, tb) in [(981, 961), (1, -1), (961, 981), (501, 481), (21, 1), (999, 0), (1, 21), (481, 501)]:
        a_t = float(ac[ta])
        a_p = float(ac[tb]) if tb >= 0 else float(ac[0])
        x = rng.standard_normal(16)
        e = rng.standard_normal(16)
        y = backward_ddim(x, a_t, a_p, e)
        y2 = forward_ddim(x, a_t, a_p, e)
        assert np.array_equal(y, y2)                              # forward_ddim only forwards its arguments
        out["step"].append({"alpha_t": a_t, "alpha_tm1": a_p, "x_t": x.tolist(), "eps": e.tolist(), "out": np.asarray(y).tolist()})

    # ---- whole loops
    class Recorder:
        def __init__(self):
            self.calls = []

        def __call__(self, x_t, alpha_t, alpha_tm1, eps_xt):
            self.calls.append((float(alpha_t), float(alpha_tm1)))
            return backward_ddim(x_t, alpha_t, alpha_tm1, eps_xt)

    for S in (10, 20, 30, 50):
        for reverse in (False, True):
            for guidance in ((1.0, 7.5) if not reverse else (1.0,)):
                rec = Recorder()
                seen_t = []
                T = 1000
                ts = Timesteps(int(t) for t in ((np.arange(S) * (T // S)).round()[::-1].astype(np.int64) + 1))   # leading spacing, steps_offset 1

                class Sched:
                    timesteps = ts
                    init_noise_sigma = 1.0
                    alphas_cumprod = ac
                    final_alpha_cumprod = float(ac[0])
                    num_inference_steps = None
                    config = types.SimpleNamespace(num_train_timesteps=T)

                    def set_timesteps(self, n):
                        self.num_inference_steps = n

                    def scale_model_input(self, x, t):
                        return x

                def unet(x, t, encoder_hidden_states=None):
                    seen_t.append(int(t))
                    x = np.asarray(x)
                    if guidance > 1.0:
                        h = x.shape[0] // 2
                        s = np.concatenate([eps_fn(x[:h], t), eps_text_fn(x[h:], t)], axis=0)
                    else:
                        s = eps_fn(x, t)
                    return types.SimpleNamespace(sample=s.view(Arr))

                me = types.SimpleNamespace(scheduler=Sched(), unet=unet, device="cpu", progress_bar=lambda it: it)
                g_loop = {"__builtins__": dict(SAFE_BUILTINS), "backward_ddim": rec,
                          "torch": types.SimpleNamespace(cat=lambda xs: np.concatenate([np.asarray(a) for a in xs], axis=0))}
                loop = types.FunctionType(co_loop39, g_loop, "backward_diffusion")
                x0 = np.random.RandomState(100 + S).standard_normal((2, 24))
                # (the defaults of the original `def` live on the function object, not in the code object: every argument is passed)
                res = loop(me, use_old_emb_i=25, text_embeddings="ctx", old_text_embeddings=None, new_text_embeddings=None, latents=x0.copy(),
                           num_inference_steps=S, guidance_scale=guidance, callback=None, callback_steps=1, reverse_process=reverse)
                out["loops"].append({"steps": S, "reverse_process": reverse, "guidance_scale": guidance,
                                     "timesteps_input": list(ts), "model_t": seen_t,
                                     "alpha_t": [c[0] for c in rec.calls], "alpha_tm1": [c[1] for c in rec.calls],
                                     "x_in": x0.tolist(), "x_out": np.asarray(res).tolist()})
    out["alphas_cumprod_probe"] = {str(i): float(ac[i]) for i in (0, 1, 21, 481, 501, 961, 981, 999)}
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, "loops:", len(out["loops"]), "re-encoded 'is not':", n_reenc)


if __name__ == "__main__":
    main()
