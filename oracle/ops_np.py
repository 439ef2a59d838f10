"""TEST INFRASTRUCTURE — numpy restatements of the reference's element-wise / FIR operators.

  fused_bias_act     <- kernel arithmetic, reference op/fused_bias_act_kernel.cu:15-42
  fused_leaky_relu*  <- autograd structure,  reference op/fused_act.py:20-72, 86-97
  upfirdn2d*         <- index algebra,       reference op/upfirdn2d.py:159-200 (upfirdn2d_native)
                        and the gradient parameterisation op/upfirdn2d.py:31-42, 103-114
  modulated_conv2d   <- reference layers.py:293-323 (float64 accumulation, tolerance oracle)

PINNING: tests/test_oracle_ops.py checks these against tests/golden/fused_act.npz,
tests/golden/upfirdn2d.npz and tests/golden/modconv.npz, which oracle/make_golden.py generated
by running the reference's own Python CPU branch (the .cu files need CUDA headers and cannot be
built here).  The dense contraction inside the reference is torch's conv2d (an unpinned
third-party dependency): parity there is a stated fp32 tolerance, not bitwise.

All float32 arithmetic below is done with separate multiply and add roundings (numpy never
fuses), in a fixed tap order, so that the HIP kernels compiled with -ffp-contract=off can be
compared bit for bit.  Only tests/, smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- fused bias act
def fused_bias_act(x, b=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    """out = f(x + b[channel]) * scale with channel = dim 1 of x (flat: (i / step_b) % size_b)."""
    x = np.asarray(x, F32)
    alpha, scale = F32(alpha), F32(scale)
    t = x
    if b is not None and np.size(b):
        shape = [1] * x.ndim
        shape[1] = -1
        t = x + np.asarray(b, F32).reshape(shape)
    code = act * 10 + grad
    if code == 30:
        y = np.where(t > 0, t, t * alpha)
    elif code == 31:
        r = np.asarray(ref, F32) if ref is not None and np.size(ref) else np.zeros_like(t)
        y = np.where(r > 0, t, t * alpha)
    elif code in (12, 32):
        y = np.zeros_like(t)
    else:  # 10, 11 and anything unknown: identity
        y = t
    return (y * scale).astype(F32)


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


def fused_leaky_relu_backward(grad_out, out, negative_slope=0.2, scale=2 ** 0.5):
    """(grad_input, grad_bias): grad_bias is summed in float64 and rounded once."""
    gi = fused_bias_act(grad_out, None, out, 3, 1, negative_slope, scale)
    axes = (0,) + tuple(range(2, gi.ndim))
    return gi, gi.astype(np.float64).sum(axes).astype(F32)


def fused_leaky_relu_double_backward(gg_input, gg_bias, out, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(gg_input, gg_bias, out, 3, 1, negative_slope, scale)


# --------------------------------------------------------------------------- upfirdn2d
def upfirdn2d_out_size(in_size, up, down, pad0, pad1, k):
    return (in_size * up + pad0 + pad1 - k) // down + 1


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x [N,C,H,W] float32, kernel [kh,kw]; same pad on both axes, as the reference's public
    signature (op/upfirdn2d.py:145)."""
    return upfirdn2d_full(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_full(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    x = np.asarray(x, F32)
    k = np.asarray(kernel, F32)
    n, c, ih, iw = x.shape
    kh, kw = k.shape
    # zero-insert
    u = np.zeros((n, c, ih * up_y, iw * up_x), F32)
    u[:, :, ::up_y, ::up_x] = x
    # pad (negative pad crops)
    u = np.pad(u, ((0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))))
    u = u[:, :, max(-py0, 0): u.shape[2] - max(-py1, 0), max(-px0, 0): u.shape[3] - max(-px1, 0)]
    oh_full = u.shape[2] - kh + 1
    ow_full = u.shape[3] - kw + 1
    oh = upfirdn2d_out_size(ih, up_y, down_y, py0, py1, kh)
    ow = upfirdn2d_out_size(iw, up_x, down_x, px0, px1, kw)
    if oh <= 0 or ow <= 0:
        return np.zeros((n, c, max(oh, 0), max(ow, 0)), F32)
    kf = k[::-1, ::-1]
    acc = np.zeros((n, c, oh, ow), F32)
    # correlation with the flipped kernel, taps visited ky-major then kx, fp32 mul then add
    for ky in range(kh):
        for kx in range(kw):
            win = u[:, :, ky: ky + oh_full: down_y, kx: kx + ow_full: down_x][:, :, :oh, :ow]
            acc = acc + win * kf[ky, kx]
    return acc


def upfirdn2d_grad_params(in_h, in_w, kh, kw, up, down, pad):
    """Parameters of the upfirdn2d call that computes d/d(input): reference
    op/upfirdn2d.py:103-114 and :31-42 (up and down trade places, kernel is flipped)."""
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    gx0 = kw - px0 - 1
    gy0 = kh - py0 - 1
    gx1 = in_w * up_x - out_w * down_x + px0 - up_x + 1
    gy1 = in_h * up_y - out_h * down_y + py0 - up_y + 1
    return dict(up_x=down_x, up_y=down_y, down_x=up_x, down_y=up_y,
                px0=gx0, px1=gx1, py0=gy0, py1=gy1), (out_h, out_w)


def upfirdn2d_backward(grad_out, kernel, in_shape, up=1, down=1, pad=(0, 0)):
    n, c, ih, iw = in_shape
    k = np.asarray(kernel, F32)
    prm, _ = upfirdn2d_grad_params(ih, iw, k.shape[0], k.shape[1], (up, up), (down, down),
                                   (pad[0], pad[1], pad[0], pad[1]))
    return upfirdn2d_full(grad_out, k[::-1, ::-1], **prm)


# --------------------------------------------------------------------------- modulated conv
def _conv2d_f64(x, w, stride=1, padding=0):
    """x [B,Ci,H,W], w [B,Co,Ci,k,k] (per-sample weights) -> [B,Co,Ho,Wo], float64."""
    b, ci, h, wd = x.shape
    co, k = w.shape[1], w.shape[-1]
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding), (padding, padding)))
    ho = (h + 2 * padding - k) // stride + 1
    wo = (wd + 2 * padding - k) // stride + 1
    out = np.zeros((b, co, ho, wo), np.float64)
    for ky in range(k):
        for kx in range(k):
            win = xp[:, :, ky: ky + stride * ho: stride, kx: kx + stride * wo: stride]
            out += np.einsum("bchw,boc->bohw", win, w[:, :, :, ky, kx])
    return out


def _conv_transpose2d_f64(x, w, stride=2):
    """x [B,Ci,H,W], w [B,Co,Ci,k,k] -> [B,Co,(H-1)*s+k,(W-1)*s+k]; out[s*y+ky] += x[y]*w[ky]."""
    b, ci, h, wd = x.shape
    co, k = w.shape[1], w.shape[-1]
    out = np.zeros((b, co, (h - 1) * stride + k, (wd - 1) * stride + k), np.float64)
    for ky in range(k):
        for kx in range(k):
            out[:, :, ky: ky + stride * h: stride, kx: kx + stride * wd: stride] += np.einsum(
                "bchw,boc->bohw", x, w[:, :, :, ky, kx])
    return out


def modulated_conv2d(x, weight, style, demodulate=True, upsample=False, blur_kernel=None,
                     blur_pad=(1, 1), eps=1e-8):
    """x [B,Ci,H,W]; weight [1,Co,Ci,k,k] (raw parameter); style [B,Ci] (already through the
    modulation EqualLinear).  float64 throughout; returns float32.
    upsample=True: transposed conv stride 2 then the FIR blur (kernel already x4)."""
    x = np.asarray(x, np.float64)
    wgt = np.asarray(weight, np.float64)
    s = np.asarray(style, np.float64)
    _, co, ci, k, _ = wgt.shape
    scale = 1.0 / np.sqrt(ci * k * k)
    w = scale * wgt * s[:, None, :, None, None]                          # [B,Co,Ci,k,k]
    if demodulate:
        w = w / np.sqrt((w * w).sum((2, 3, 4), keepdims=True) + eps)
    if upsample:
        out = _conv_transpose2d_f64(x, w, 2)
        kf = np.asarray(blur_kernel, np.float64)[::-1, ::-1]
        p0, p1 = blur_pad
        op = np.pad(out, ((0, 0), (0, 0), (p0, p1), (p0, p1)))
        oh, ow = op.shape[2] - kf.shape[0] + 1, op.shape[3] - kf.shape[1] + 1
        res = np.zeros(out.shape[:2] + (oh, ow), np.float64)
        for ky in range(kf.shape[0]):
            for kx in range(kf.shape[1]):
                res += op[:, :, ky: ky + oh, kx: kx + ow] * kf[ky, kx]
        return res.astype(F32)
    return _conv2d_f64(x, w, 1, k // 2).astype(F32)


def make_blur_kernel(taps=(1, 3, 3, 1), gain=1.0):
    k = np.asarray(taps, F32)
    k2 = (k[None, :] * k[:, None]).astype(F32)
    return (k2 / k2.sum() * F32(gain)).astype(F32)
