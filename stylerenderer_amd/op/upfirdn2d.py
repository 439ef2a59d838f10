"""upfirdn2d — host side of csrc/upfirdn2d.hip.

Interface parity with the reference (reference op/upfirdn2d.py):
  upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))                        :145-156
  gradient = the same operator with up<->down swapped, flipped taps and the padding of :111-114;
  double-backward = the forward operator on the grad-of-grad (:63-85).
`upfirdn2d_op(...)` is the tensor-level operator of the reference's pybind module
(op/upfirdn2d.cpp:24-26) on top of the C ABI `sr_upfirdn2d`.

Device tensors always run the HIP kernels.  CPU tensors take a plain PyTorch route, which is what
the reference does for them (`upfirdn2d_native`, op/upfirdn2d.py:159-200); it is written here as a
single strided transposed/regular convolution rather than the reference's view+pad sequence.
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F

from .. import _lib
from ._dispatch import DerivedCache, is_device_tensor, on_device_of, require_f32, stream_of


def _out_size(n, up, down, p0, p1, k):
    return (n * up + p0 + p1 - k) // down + 1


def upfirdn2d_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """input [major, in_h, in_w, 1] (or [major, in_h, in_w]) -> [major, out_h, out_w, 1]."""
    require_f32(input, "upfirdn2d")
    x = input.contiguous()
    k = kernel.contiguous()
    if k.device != x.device:
        k = k.to(x.device)
    require_f32(k, "upfirdn2d kernel")
    if x.dim() == 4 and x.size(3) != 1:
        raise RuntimeError("upfirdn2d: minor dimension must be 1")
    major, in_h, in_w = x.size(0), x.size(1), x.size(2)
    kh, kw = k.shape
    out_h = _out_size(in_h, up_y, down_y, pad_y0, pad_y1, kh)
    out_w = _out_size(in_w, up_x, down_x, pad_x0, pad_x1, kw)
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError("upfirdn2d: empty output (%d x %d)" % (out_h, out_w))
    out = torch.empty((major, out_h, out_w, 1), dtype=x.dtype, device=x.device)
    with on_device_of(x):
        rc = _lib.lib().sr_upfirdn2d(_lib.ptr(out), _lib.ptr(x), _lib.ptr(k), major, in_h, in_w, out_h,
                                     out_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                     pad_y0, pad_y1, stream_of(x))
    _lib.check(rc, "sr_upfirdn2d")
    return out


_FLIP_CACHE = DerivedCache(64)


def flipped(kernel):
    """torch.flip(kernel, [0, 1]), cached per tensor (address, version): the FIR kernels are module buffers that
    never change, and the gradient of every blur / resampling call needs the flipped taps (one launch each)."""
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), str(kernel.device), kernel.dtype)
    hit = _FLIP_CACHE.get(key)
    if hit is None:
        # holds `kernel`: the key is its address
        hit = _FLIP_CACHE.put(key, (torch.flip(kernel.detach(), [0, 1]).contiguous(), kernel))
    return hit[0]


class UpFirDn2dBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        up_x, up_y = up
        down_x, down_y = down
        g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1 = g_pad
        grad_output = grad_output.reshape(in_size[0] * in_size[1], out_size[0], out_size[1], 1)
        grad_input = upfirdn2d_op(grad_output, grad_kernel, down_x, down_y, up_x, up_y,
                                  g_pad_x0, g_pad_x1, g_pad_y0, g_pad_y1)
        grad_input = grad_input.view(in_size[0], in_size[1], in_size[2], in_size[3])
        ctx.save_for_backward(kernel)
        ctx.up, ctx.down, ctx.pad = up, down, pad
        ctx.in_size, ctx.out_size = in_size, out_size
        return grad_input

    @staticmethod
    def backward(ctx, gradgrad_input):
        (kernel,) = ctx.saved_tensors
        gg = gradgrad_input.reshape(ctx.in_size[0] * ctx.in_size[1], ctx.in_size[2], ctx.in_size[3], 1)
        out = upfirdn2d_op(gg, kernel, ctx.up[0], ctx.up[1], ctx.down[0], ctx.down[1], *ctx.pad)
        out = out.view(ctx.in_size[0], ctx.in_size[1], ctx.out_size[0], ctx.out_size[1])
        return out, None, None, None, None, None, None, None, None


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        pad_x0, pad_x1, pad_y0, pad_y1 = pad
        kernel_h, kernel_w = kernel.shape
        batch, channel, in_h, in_w = input.shape
        ctx.in_size = input.shape
        out_h = _out_size(in_h, up_y, down_y, pad_y0, pad_y1, kernel_h)
        out_w = _out_size(in_w, up_x, down_x, pad_x0, pad_x1, kernel_w)
        ctx.out_size = (out_h, out_w)
        ctx.up, ctx.down, ctx.pad = (up_x, up_y), (down_x, down_y), (pad_x0, pad_x1, pad_y0, pad_y1)
        ctx.g_pad = (kernel_w - pad_x0 - 1,
                     in_w * up_x - out_w * down_x + pad_x0 - up_x + 1,
                     kernel_h - pad_y0 - 1,
                     in_h * up_y - out_h * down_y + pad_y0 - up_y + 1)
        ctx.save_for_backward(kernel, flipped(kernel))
        # explicit plane count: `-1` is ambiguous for zero-size batches / channel counts (the reference's
        # view(-1, channel, ...) raises there, op/upfirdn2d.py:125)
        out = upfirdn2d_op(input.reshape(batch * channel, in_h, in_w, 1), kernel, up_x, up_y, down_x, down_y,
                           pad_x0, pad_x1, pad_y0, pad_y1)
        return out.view(batch, channel, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        grad_input = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, ctx.up, ctx.down,
                                             ctx.pad, ctx.g_pad, ctx.in_size, ctx.out_size)
        return grad_input, None, None, None, None


def _upfirdn2d_cpu(input, kernel, up, down, pad):
    """CPU tensors: zero-insert + pad/crop + correlation with the flipped kernel + decimate."""
    n, c, in_h, in_w = input.shape
    kh, kw = kernel.shape
    x = input.reshape(n * c, 1, in_h, in_w)
    if up > 1:
        z = x.new_zeros(n * c, 1, in_h * up, in_w * up)
        z[:, :, ::up, ::up] = x
        x = z
    p0, p1 = pad
    x = F.pad(x, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    x = x[:, :, max(-p0, 0): x.shape[2] - max(-p1, 0), max(-p0, 0): x.shape[3] - max(-p1, 0)]
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(x.dtype)
    y = F.conv2d(x, w)[:, :, ::down, ::down]
    return y.reshape(n, c, y.shape[2], y.shape[3])


class UpsampleAdd(Function):
    """upfirdn2d(skip, kernel, up = 2, pad) + addend in one kernel (ToRGB's skip connection, reference model.py:66-68).
    Linear in both inputs: the gradient of `skip` is the existing differentiable down-sampling operator, the gradient
    of `addend` is the incoming gradient itself, so gradients of any order stay on the same kernels."""

    @staticmethod
    def forward(ctx, skip, kernel, addend, pad):
        b, c, ih, iw = skip.shape
        p0, p1 = pad
        oh, ow = 2 * ih + p0 + p1 - 3, 2 * iw + p0 + p1 - 3
        x, a, k = skip.contiguous(), addend.contiguous(), kernel.contiguous()
        out = torch.empty((b, c, oh, ow), dtype=x.dtype, device=x.device)
        with on_device_of(x):
            rc = _lib.lib().sr_upsample2_add(_lib.ptr(out), _lib.ptr(x), _lib.ptr(k), _lib.ptr(a), b * c, ih, iw, oh,
                                             ow, p0, p1, stream_of(x))
        _lib.check(rc, "sr_upsample2_add")
        ctx.save_for_backward(kernel)
        ctx.cfg = (p0, p1, tuple(skip.shape), (oh, ow))
        return out

    @staticmethod
    def backward(ctx, g):
        (kernel,) = ctx.saved_tensors
        p0, p1, in_size, out_size = ctx.cfg
        g_skip = None
        if ctx.needs_input_grad[0]:
            g_pad = (4 - p0 - 1, in_size[3] * 2 - out_size[1] + p0 - 2 + 1, 4 - p0 - 1,
                     in_size[2] * 2 - out_size[0] + p0 - 2 + 1)
            g_skip = UpFirDn2dBackward.apply(g, kernel, flipped(kernel), (2, 2), (1, 1), (p0, p1, p0, p1), g_pad,
                                             in_size, out_size)
        return g_skip, None, (g if ctx.needs_input_grad[2] else None), None


def upsample2_add(skip, kernel, pad, addend):
    """addend + upfirdn2d(skip, kernel, up=2, pad=pad); one launch on device tensors with the 4x4 kernel."""
    ok = (is_device_tensor(skip) and skip.dtype == torch.float32 and addend.dtype == torch.float32 and skip.dim() == 4
          and tuple(kernel.shape) == (4, 4) and skip.numel() > 0
          and tuple(addend.shape) == (skip.shape[0], skip.shape[1], 2 * skip.shape[2] + pad[0] + pad[1] - 3,
                                      2 * skip.shape[3] + pad[0] + pad[1] - 3))
    if not ok:
        return addend + upfirdn2d(skip, kernel, up=2, down=1, pad=pad)
    k = kernel if kernel.device == skip.device else kernel.to(skip.device)
    return UpsampleAdd.apply(skip, k, addend, tuple(pad))


class SkipDown(Function):
    """x -> (x, upfirdn2d(x, kernel, down = 2, pad)): the fork at the top of a down-sampling ResBlock (reference
    layers.py:381-400: the input feeds conv1 AND the blur in front of the 1x1 stride-2 skip convolution).  As two
    consumers of one tensor, autograd adds the two input gradients with a full-size elementwise pass after the
    up-sampling FIR of the skip branch has written its own full-size tensor; as one node the backward is
    `upsample2_add`: FIR of the small gradient + the conv1 gradient in one kernel (two HBM passes instead of four).
    Built from differentiable operators, so gradients of any order (R1) stay on the same kernels."""

    @staticmethod
    def forward(ctx, x, kernel, pad):
        ctx.save_for_backward(kernel)
        ctx.pad, ctx.in_h = tuple(pad), x.shape[2]
        ctx.set_materialize_grads(False)
        return x.view_as(x), upfirdn2d(x, kernel, down=2, pad=pad)

    @staticmethod
    def backward(ctx, g_same, g_down):
        (kernel,) = ctx.saved_tensors
        if g_down is None:
            return g_same, None, None
        p0 = ctx.pad[0]
        g_pad = (kernel.shape[1] - p0 - 1, ctx.in_h - g_down.shape[2] * 2 + p0 - 1 + 1)
        fk = flipped(kernel)
        if g_same is None:
            return upfirdn2d(g_down, fk, up=2, down=1, pad=g_pad), None, None
        return upsample2_add(g_down, fk, g_pad, g_same), None, None


def skip_down(x, kernel, pad):
    return SkipDown.apply(x, kernel if kernel.device == x.device else kernel.to(x.device), tuple(pad))


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not is_device_tensor(input):
        return _upfirdn2d_cpu(input, kernel, up, down, pad)
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
