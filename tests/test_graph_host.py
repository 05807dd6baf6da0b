"""CPU: host logic of graph.GraphedEpsModel -- the pass-through for non-device tensors, mode parsing, context identity."""
import pytest
import torch

import gswm_amd  # noqa: F401
from gswm_amd import graph


class _Eps(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.calls = 0

    def prepare_context(self, ctx):
        pass

    def forward(self, x, t, ctx):
        self.calls += 1
        return x * self.w + t.float() * 0.0


def test_cpu_tensors_run_eagerly_and_attributes_pass_through():
    m = _Eps()
    gm = graph.GraphedEpsModel(m, mode="always")
    x = torch.randn(2, 4, 8, 8)
    y = gm(x, torch.tensor(5), torch.zeros(2, 1, 1))
    assert torch.equal(y, x) and m.calls == 1 and gm.stats == {"captures": 0, "replays": 0, "eager": 1, "context_refreshes": 0}
    assert gm.w is m.w and graph.graphed(gm) is gm and isinstance(graph.graphed(m), graph.GraphedEpsModel)


def test_mode_parsing(monkeypatch):
    m = _Eps()
    assert graph.GraphedEpsModel(m, mode="1").mode == "always" and graph.GraphedEpsModel(m, mode="off").mode == "never"
    monkeypatch.setenv("GSW_GRAPH", "never")
    assert graph.GraphedEpsModel(m).mode == "never"
    monkeypatch.delenv("GSW_GRAPH")
    assert graph.GraphedEpsModel(m).mode == "auto"
    with pytest.raises(ValueError):
        graph.GraphedEpsModel(m, mode="sometimes")


def test_context_identity_follows_base_version_and_view_geometry():
    c = torch.zeros(1, 77, 8)
    a, b = graph._ctx_identity(c.expand(4, -1, -1)), graph._ctx_identity(c.expand(4, -1, -1))
    assert graph._same_ctx(a, b)
    assert not graph._same_ctx(a, graph._ctx_identity(c.expand(2, -1, -1)))
    c.add_(1.0)
    assert not graph._same_ctx(a, graph._ctx_identity(c.expand(4, -1, -1)))
    assert not graph._same_ctx(graph._ctx_identity(c), graph._ctx_identity(c.clone()))
