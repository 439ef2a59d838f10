"""`op._dispatch.wanted`: needs_input_grad intersected with what the running backward pass will consume (the first pass of
the path-length regulariser / R1 — reference train.py:110-134 — asks for latents / images only)."""
import torch

from stylerenderer_amd.op._dispatch import mark_inputs, wanted

SEEN = []


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, const, w, flag=None):
        mark_inputs(ctx, x, const, w, flag)
        ctx.save_for_backward(x, w)
        return x * w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        need = wanted(ctx)
        SEEN.append(need)
        return (g * w if need[0] else None), None, (g * x if need[2] else None), None


def _operands():
    lat = torch.randn(5, requires_grad=True)
    w0 = torch.randn(5, requires_grad=True)
    return lat, w0, torch.randn(5)


def test_full_backward_wants_everything():
    lat, w0, c = _operands()
    _Mul.apply(lat * 1.0, c, w0 * 2.0).sum().backward()
    assert SEEN[-1] == (True, False, True)
    assert lat.grad is not None and w0.grad is not None


def test_grad_wrt_latent_only_prunes_the_weight_branch_and_keeps_values():
    lat, w0, c = _operands()
    y = _Mul.apply(lat * 1.0, c, w0 * 2.0)
    (g,) = torch.autograd.grad(y.sum(), lat, create_graph=True)
    assert SEEN[-1] == (True, False, False)
    assert torch.equal(g, (w0 * 2.0).detach())
    # the recorded first pass stays differentiable w.r.t. the weights (second pass: everything wanted again)
    (g * g).sum().backward()
    assert torch.allclose(w0.grad, 8.0 * w0.detach())


def test_leaves_named_as_inputs_count_as_wanted():
    lat, w0, c = _operands()
    torch.autograd.grad(_Mul.apply(lat, c, w0).sum(), [w0])          # the engine refuses the query for a captured leaf
    assert SEEN[-1] == (False, False, True)
    lat, w0, c = _operands()
    _Mul.apply(lat, c, w0).sum().backward(inputs=[lat])
    assert SEEN[-1] == (True, False, False)


def test_switch_off(monkeypatch):
    monkeypatch.setenv("SR_PRUNE_GRADS", "0")
    lat, w0, c = _operands()
    torch.autograd.grad(_Mul.apply(lat * 1.0, c, w0 * 2.0).sum(), lat)
    assert SEEN[-1] == (True, False, True)
