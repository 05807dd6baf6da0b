"""GPU: HIP-graph replay of the eps model (graph.py) is bit-identical to the eager forward -- across steps (new x, new t), across contexts
(in-place refresh of the padded context and the cross-attention K / V^T the graph has baked in), across weight edits (recapture), and through
the DDIM loops (the reference's one-image-per-call regime, extract.py:112-117)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gswm_amd
    from gswm_amd import unet, graph, ddim, pipeline, codec
    return types.SimpleNamespace(unet=unet, graph=graph, ddim=ddim, pipeline=pipeline, codec=codec)


def _small_unet(G, seed=0):
    # product architecture at a quarter of the SD widths: every convolution / linear / attention (head_dim 64) on the hand-written kernels
    m = G.unet.UNet2DCondition(block_out_channels=(64, 128, 256, 256), cross_attention_dim=128, num_heads=(1, 2, 4, 4), head_dim=64)
    return G.unet.synthetic_init_(m, seed).cuda().half().eval()


def _inputs(rows, hw=32, ctx_dim=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, 4, hw, hw, generator=g).cuda().half()
    c = torch.randn(rows, 77, ctx_dim, generator=g).cuda().half()
    return x, c


@pytest.mark.parametrize("rows", [1, 2, 5])
def test_graph_replay_is_bit_identical_across_steps_and_contexts(G, rows):
    m = _small_unet(G)
    G.unet.FALLBACKS.clear()
    gm = G.graph.GraphedEpsModel(m, mode="always")
    x, c = _inputs(rows)
    with torch.no_grad():
        for t in (981, 500, 1):
            td = torch.full((), t, dtype=torch.int64, device="cuda")
            xs = x * (1.0 + 0.001 * t)
            want = m(xs, td, c).clone()
            got = gm(xs, td, c)
            assert torch.equal(got, want), (rows, t)
        assert gm.stats["captures"] == 1 and gm.stats["replays"] == 3 and gm.stats["context_refreshes"] == 0
        # another context tensor: static copy overwritten, padded copy and every layer's K / V^T refreshed in place
        _, c2 = _inputs(rows, seed=7)
        td = torch.full((), 321, dtype=torch.int64, device="cuda")
        assert torch.equal(gm(x, td, c2), m(x, td, c2))
        assert gm.stats["captures"] == 1 and gm.stats["context_refreshes"] == 1
        # the same context object edited in place (version counter) is a new context as well
        c2.mul_(0.5)
        assert torch.equal(gm(x, td, c2), m(x, td, c2))
        assert gm.stats["context_refreshes"] == 2
        # an expanded view of one row (what the harness passes for prompt ""): same context on every call, no refresh after the first
        c1 = c2[:1].contiguous()
        for _ in range(2):
            assert torch.equal(gm(x, td, c1.expand(rows, -1, -1)), m(x, td, c1.expand(rows, -1, -1).contiguous()))
        assert gm.stats["context_refreshes"] == (3 if rows > 1 else 2)      # one row: the "expanded" view IS c2 (same storage, geometry, version)
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS


def test_graph_recaptures_after_a_weight_edit(G):
    m = _small_unet(G)
    gm = G.graph.GraphedEpsModel(m, mode="always")
    x, c = _inputs(2)
    td = torch.full((), 41, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        y0 = gm(x, td, c).clone()
        m.conv_out.weight.mul_(2.0)                 # packed copies are keyed by the version counter; so are the graphs
        m.conv_out.bias.add_(0.25)
        _, c2 = _inputs(2, seed=3)                  # the weights are re-checked when the context changes (once per loop)
        y1 = gm(x, td, c2)
        assert gm.stats["captures"] == 2
        assert torch.equal(y1, m(x, td, c2)) and not torch.equal(y1, y0)


def test_auto_mode_graphs_small_batches_only(G):
    m = _small_unet(G)
    gm = G.graph.GraphedEpsModel(m, mode="auto", max_rows=2)
    td = torch.full((), 41, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        x, c = _inputs(2)
        gm(x, td, c)
        x3, c3 = _inputs(3)
        assert torch.equal(gm(x3, td, c3), m(x3, td, c3))
    assert gm.stats["captures"] == 1 and gm.stats["eager"] == 1
    assert G.graph.GraphedEpsModel(m, mode="never")._wants_graph(x) is False


def test_inversion_loop_through_the_graph_equals_eager(G, keys):
    """ddim_invert_extract with the graphed model: same latents, same bits as the eager loop (one image, like extract.py:112-117)."""
    key, nonce = keys
    m = _small_unet(G, seed=2)
    g = torch.Generator().manual_seed(5)
    x0 = (0.3 * torch.randn(1, 4, 32, 32, generator=g)).cuda().half()
    ctx = torch.randn(1, 77, 128, generator=g).cuda().half()
    sched = G.ddim.DDIMSchedule(num_inference_steps=10)
    gm = G.graph.GraphedEpsModel(m, mode="always")
    bits_e, flags_e, z_e = G.ddim.ddim_invert_extract(m, x0, ctx, sched, key, nonce, 256, return_latents=True)
    bits_g, flags_g, z_g = G.ddim.ddim_invert_extract(gm, x0, ctx, sched, key, nonce, 256, return_latents=True)
    assert torch.equal(z_e, z_g) and torch.equal(bits_e, bits_g) and torch.equal(flags_e, flags_g)
    assert gm.stats["captures"] == 1 and gm.stats["replays"] == 10


def test_pipeline_wraps_the_unet_and_full_size_forward_matches(G, keys):
    """The pipeline graphs the SD 2.1-shaped UNet at batch 1 (CFG: 2 rows): replay == eager on the full-size model."""
    key, nonce = keys
    m = G.unet.synthetic_init_(G.unet.UNet2DCondition(), 0).cuda().half().eval()
    pipe = G.pipeline.GaussianShadingPipeline(m, key, nonce, G.codec.pad_message("lthero", 32), num_inference_steps=2,
                                              ctx_uncond=torch.randn(1, 77, 1024, device="cuda", dtype=torch.float16))
    assert isinstance(pipe.eps_model, G.graph.GraphedEpsModel)
    x = torch.randn(2, 4, 64, 64, device="cuda", dtype=torch.float16)
    c = torch.randn(2, 77, 1024, device="cuda", dtype=torch.float16)
    td = torch.full((), 981, dtype=torch.int64, device="cuda")
    G.unet.FALLBACKS.clear()
    with torch.no_grad():
        want = m(x, td, c).clone()
        got = pipe.eps_model(x, td, c)
        assert torch.equal(got, want)
        got2 = pipe.eps_model(x * 0.5, td, c)
        assert torch.equal(got2, m(x * 0.5, td, c))
    assert pipe.eps_model.stats["captures"] == 1 and G.unet.FALLBACKS == {}


def test_graph_notices_weight_edits_under_an_unchanged_context(G):
    """a harness that keeps ONE context for its whole life: the parameters are re-checked every CHECK_EVERY replays, not only on a context change"""
    m = _small_unet(G)
    gm = G.graph.GraphedEpsModel(m, mode="always")
    x, c = _inputs(1)
    td = torch.full((), 41, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        y0 = gm(x, td, c)
        m.conv_out.bias.add_(0.5)
        ys = [gm(x, td, c) for _ in range(G.graph.CHECK_EVERY + 1)]
        want = m(x, td, c)
    assert gm.stats["captures"] == 2
    assert torch.equal(ys[-1], want) and not torch.equal(ys[-1], y0)


def test_graph_entries_are_keyed_by_the_module_switches(G):
    """an A/B toggle after a capture captures anew (and the old entry serves again when the switch comes back) instead of replaying the old sequence"""
    from gswm_amd import pf
    m = _small_unet(G)
    gm = G.graph.GraphedEpsModel(m, mode="always")
    x, c = _inputs(2)
    td = torch.full((), 41, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        y_a = gm(x, td, c)
        old = pf.GN_FUSED_MAX_WGS
        pf.GN_FUSED_MAX_WGS = 0
        try:
            y_b = gm(x, td, c)
            assert gm.stats["captures"] == 2
            assert torch.equal(y_b, m(x, td, c))
        finally:
            pf.GN_FUSED_MAX_WGS = old
        y_a2 = gm(x, td, c)
    assert gm.stats["captures"] == 2 and torch.equal(y_a, y_a2)


@pytest.mark.usefixtures("library_kernels_allowed")      # small / odd shapes off the hand-written path: strict mode (the default) would raise
def test_graph_results_are_not_aliased_and_entries_are_bounded(G):
    m = _small_unet(G)
    gm = G.graph.GraphedEpsModel(m, mode="always")
    x, c = _inputs(1)
    t1, t2 = (torch.full((), t, dtype=torch.int64, device="cuda") for t in (41, 901))
    with torch.no_grad():
        e1 = gm(x, t1, c)
        e2 = gm(x, t2, c)
        assert e1.data_ptr() != e2.data_ptr() and not torch.equal(e1, e2)
        assert torch.equal(e1, m(x, t1, c)) and torch.equal(e2, m(x, t2, c))
        raw = G.graph.GraphedEpsModel(m, mode="always", clone_output=False)
        assert raw(x, t1, c).data_ptr() == raw(x, t2, c).data_ptr()          # the static buffer itself: the in-repo loops' form
        for hw in (8, 16, 24, 32, 40, 48, 56, 64, 72, 80):
            xs, _ = _inputs(1, hw=hw)
            gm(xs, t1, c)
    assert len(gm._entries) <= G.graph.MAX_ENTRIES


@pytest.mark.parametrize("rows", [1, 3])
def test_cfg_dup_forward_equals_the_doubled_batch(G, rows):
    """classifier-free guidance with shared latents: forward(x [B], ctx [2B], cfg_dup=True) == forward(cat[x, x], ctx) -- the context-free prefix (conv_in,
    first resnet, first transformer up to its cross-attention queries) is computed once; eager and through the graph; per-image timesteps too"""
    m = _small_unet(G)
    G.unet.FALLBACKS.clear()
    x, _ = _inputs(rows)
    _, c2 = _inputs(2 * rows, seed=11)
    td = torch.full((), 301, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        want = m(torch.cat([x, x]), td, c2)
        got = m(x, td, c2, cfg_dup=True)
        assert got.shape == want.shape
        scale = want.float().abs().max().item()
        assert (got.float() - want.float()).abs().max().item() <= 2e-3 * scale          # (tilings / split-K choices differ with the row count: fp32 summation order)
        gm = G.graph.GraphedEpsModel(m, mode="always")
        assert gm.supports_cfg_dup
        for t in (301, 7):
            tdd = torch.full((), t, dtype=torch.int64, device="cuda")
            assert torch.equal(gm(x, tdd, c2, cfg_dup=True), m(x, tdd, c2, cfg_dup=True))
        assert gm.stats["captures"] == 1 and gm.stats["replays"] == 2
        tt = torch.arange(rows, device="cuda") * 37 + 5
        want_t = m(torch.cat([x, x]), torch.cat([tt, tt]), c2)
        got_t = m(x, tt, c2, cfg_dup=True)
        assert (got_t.float() - want_t.float()).abs().max().item() <= 2e-3 * scale
        # switched off: the doubled batch itself
        G.unet.CFG_SHARED_PREFIX = False
        try:
            assert torch.equal(m(x, td, c2, cfg_dup=True), want)
        finally:
            G.unet.CFG_SHARED_PREFIX = True
        with pytest.raises(ValueError):
            m(x, td, c2[:rows], cfg_dup=True)
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS


def test_sampling_loop_with_shared_prefix_equals_the_doubled_batch(G):
    """ddim_sample: the model that knows about the shared latents against a plain callable that gets the reference's doubled batch"""
    m = _small_unet(G, seed=4)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(2, 4, 32, 32, generator=g).cuda().half()
    cu = torch.randn(1, 77, 128, generator=g).cuda().half().expand(2, -1, -1).contiguous()
    ct = torch.randn(2, 77, 128, generator=g).cuda().half()
    sched = G.ddim.DDIMSchedule(num_inference_steps=6)
    a = G.ddim.ddim_sample(m, z, ct, sched, ctx_uncond=cu, guidance_scale=7.5)
    b = G.ddim.ddim_sample(lambda x, t, c: m(x, t, c), z, ct, sched, ctx_uncond=cu, guidance_scale=7.5)
    assert (a.float() - b.float()).abs().max().item() <= 1e-2 * b.float().abs().max().item()
