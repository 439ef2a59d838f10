"""Skinny products of all modulated layers of a pass, one launch each — host side of csrc/bank_mm.hip.

Two families of three autograd Functions, each family closed under differentiation (every derivative of a member is
another member), so the recorded backward of the path-length regulariser and of R1 (create_graph=True, reference
train.py:110-134) stays on the same three kernels at any order:

  modulation   `latent` X [B, R, K]; layer p reads row r_p with weight W_p [N_p, K] and bias b_p [N_p]
      ModNT(X, W.., b..)   S  = flat of blocks [B, N_p]:  S_p = alpha * X[:, r_p] @ W_p^T + bscale * b_p
      ModNN(G, W..)        gX [B, R, K]:  gX[:, r] = alpha * sum_{p: r_p = r} G_p @ W_p       (fixed order over p)
      ModTN(G, X)          gW_p = alpha * G_p^T @ X[:, r_p],  gb_p = bscale * sum_b G_p
      d ModNT = (ModNN, ModTN);  d ModNN = (ModNT without bias, ModTN);  d ModTN = (ModNT with the bias cotangents, ModNN)

  demodulation  A = flat of blocks [B, I_p] (the squared modulations), M_p = Wsq_p [I_p, J_p]
      DemNN(A, M..)        Q  = flat of blocks [B, J_p]:  Q_p = A_p @ M_p
      DemNT(G, M..)        gA = flat shaped like A:       gA_p = G_p @ M_p^T        (blocks no layer reads: zero)
      DemTN(A, G)          gM_p = A_p^T @ G_p
      d DemNN = (DemNT, DemTN);  d DemNT = (DemNN, DemTN);  d DemTN = (DemNT, DemNN)

A `Layout` fixes the block structure of a flat buffer; per-layer tensors are `flat.split(...)` views (one cat in the
backward, nothing per layer).  The weights are read where they lie — no stacking, no library GEMM (round 3's
op/style_bank.py stacked them for rocBLAS strided-batched products: ~25 copy launches per forward)."""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib
from ._dispatch import mark_inputs, on_device_of, stream_of, wanted


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _pa(ptrs):
    return _arr(ctypes.c_void_p, ptrs)


def _la(vals):
    return _arr(ctypes.c_int64, [int(v) for v in vals])


def _blocks(flat, batch, widths):
    """Per-problem [batch, width] views of a problem-major flat buffer."""
    return [t.view(batch, w) for t, w in zip(flat.split([batch * w for w in widths]), widths)]


def _offsets(batch, widths):
    offs, o = [], 0
    for w in widths:
        offs.append(o)
        o += batch * w
    return offs, o


def _dense(t):
    return t if t.is_contiguous() else t.contiguous()


def _flat_out(tensors_like, shapes, device, dtype):
    """One allocation, one contiguous view per shape."""
    sizes = [(a * b + 3) // 4 * 4 for a, b in shapes]
    store = torch.empty(sum(sizes), device=device, dtype=dtype)
    outs, o = [], 0
    for (a, b), n in zip(shapes, sizes):
        outs.append(store[o:o + a * b].view(a, b))
        o += n
    return outs


# ----------------------------------------------------------------------------------------------------------------------
# modulation family.  meta = (rows, widths, alpha, bscale): rows[p] = latent row of problem p, widths[p] = N_p
# ----------------------------------------------------------------------------------------------------------------------
def _mod_nt(meta, x, ws, bs):
    rows, widths, alpha, bscale = meta
    x = _dense(x)
    b, r, k = x.shape
    offs, total = _offsets(b, widths)
    out = torch.empty(total, device=x.device, dtype=x.dtype)
    ws = [_dense(w) for w in ws]
    bs = [_dense(t) for t in bs] if bs else None
    base, xp = out.data_ptr(), x.data_ptr()
    with on_device_of(x):
        rc = _lib.lib().sr_bank_nt(
            len(rows), _pa([base + 4 * o for o in offs]), _pa([xp + 4 * k * rr for rr in rows]),
            _pa([w.data_ptr() for w in ws]), _pa([t.data_ptr() for t in bs]) if bs else None,
            _la([r * k] * len(rows)), _la(widths), _la([k] * len(rows)), _la(widths), b, float(alpha), float(bscale),
            stream_of(x))
    _lib.check(rc, "sr_bank_nt")
    return out


def _mod_nn(meta, g, ws, shape):
    rows, widths, alpha, _ = meta
    g = _dense(g)
    b, r, k = shape
    offs, _ = _offsets(b, widths)
    used = sorted(set(rows))
    out = (torch.empty if len(used) == r else torch.zeros)(shape, device=g.device, dtype=g.dtype)
    ws = [_dense(w) for w in ws]
    gp, op = g.data_ptr(), out.data_ptr()
    terms = [[p for p, rr in enumerate(rows) if rr == row] for row in used]
    flat_terms = [p for ts in terms for p in ts]
    with on_device_of(g):
        rc = _lib.lib().sr_bank_nn(
            len(used), _pa([op + 4 * k * row for row in used]), _la([r * k] * len(used)), _la([k] * len(used)),
            _arr(ctypes.c_int, [len(ts) for ts in terms]), _pa([gp + 4 * offs[p] for p in flat_terms]),
            _pa([ws[p].data_ptr() for p in flat_terms]), _la([widths[p] for p in flat_terms]),
            _la([widths[p] for p in flat_terms]), b, float(alpha), stream_of(g))
    _lib.check(rc, "sr_bank_nn")
    return out


def _mod_tn(meta, g, x, want_bias):
    rows, widths, alpha, bscale = meta
    g, x = _dense(g), _dense(x)
    b, r, k = x.shape
    offs, _ = _offsets(b, widths)
    gws = _flat_out(None, [(n, k) for n in widths], g.device, g.dtype)
    gbs = _flat_out(None, [(1, n) for n in widths], g.device, g.dtype) if want_bias else None
    gp, xp = g.data_ptr(), x.data_ptr()
    with on_device_of(g):
        rc = _lib.lib().sr_bank_tn(
            len(rows), _pa([t.data_ptr() for t in gws]), _pa([t.data_ptr() for t in gbs]) if gbs else None,
            _pa([gp + 4 * o for o in offs]), _pa([xp + 4 * k * rr for rr in rows]), _la(widths), _la([r * k] * len(rows)),
            _la(widths), _la([k] * len(rows)), b, float(alpha), float(bscale), stream_of(g))
    _lib.check(rc, "sr_bank_tn")
    return gws, ([t.view(-1) for t in gbs] if gbs else None)


class ModNT(Function):
    @staticmethod
    def forward(ctx, meta, x, *wb):
        mark_inputs(ctx, meta, x, *wb)
        n = len(meta[0])
        ws, bs = wb[:n], wb[n:]
        ctx.meta, ctx.has_bias = meta, len(bs) > 0
        ctx.save_for_backward(x, *ws)
        return _mod_nt(meta, x, ws, bs)

    @staticmethod
    def backward(ctx, g):
        x, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        n = len(ws)
        need = wanted(ctx)
        gx = ModNN.apply(ctx.meta, g, tuple(x.shape), *ws) if need[1] else None
        want_w, want_b = any(need[2:2 + n]), ctx.has_bias and any(need[2 + n:])
        gws = gbs = None
        if want_w or want_b:
            res = ModTN.apply(ctx.meta, g, x, want_b)
            gws, gbs = res[:n], res[n:]
        out = [None, gx]
        out += [gws[p] if (gws is not None and need[2 + p]) else None for p in range(n)]
        if ctx.has_bias:
            out += [gbs[p] if (gbs and need[2 + n + p]) else None for p in range(n)]
        return tuple(out)


class ModNN(Function):
    @staticmethod
    def forward(ctx, meta, g, shape, *ws):
        mark_inputs(ctx, meta, g, shape, *ws)
        ctx.meta, ctx.shape = meta, shape
        ctx.save_for_backward(g, *ws)
        return _mod_nn(meta, g, ws, shape)

    @staticmethod
    def backward(ctx, go):
        g, ws = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        n = len(ws)
        need = wanted(ctx)
        rows, widths, alpha, _ = ctx.meta
        meta = (rows, widths, alpha, 0.0)
        dg = ModNT.apply(meta, go, *ws) if need[1] else None
        dws = ModTN.apply(meta, g, go, False) if any(need[3:]) else None
        return (None, dg, None) + tuple(dws[p] if (dws is not None and need[3 + p]) else None for p in range(n))


class ModTN(Function):
    @staticmethod
    def forward(ctx, meta, g, x, want_bias):
        mark_inputs(ctx, meta, g, x, want_bias)
        ctx.meta, ctx.want_bias = meta, bool(want_bias)
        ctx.save_for_backward(g, x)
        gws, gbs = _mod_tn(meta, g, x, want_bias)
        return tuple(gws) + (tuple(gbs) if gbs else ())

    @staticmethod
    def backward(ctx, *go):
        g, x = ctx.saved_tensors
        rows, widths, alpha, bscale = ctx.meta
        n = len(rows)
        k = x.shape[2]
        gow = [t if t is not None else x.new_zeros(widths[p], k) for p, t in enumerate(go[:n])]
        gob = ()
        if ctx.want_bias:
            gob = tuple(t if t is not None else x.new_zeros(widths[p]) for p, t in enumerate(go[n:2 * n]))
        need = wanted(ctx)
        dg = ModNT.apply((rows, widths, alpha, bscale if gob else 0.0), x, *gow, *gob) if need[1] else None
        dx = ModNN.apply((rows, widths, alpha, 0.0), g, tuple(x.shape), *gow) if need[2] else None
        return None, dg, dx, None


def modulation(x, rows, weights, biases, alpha, bscale):
    """Per-layer modulation vectors [B, N_p] = alpha * x[:, rows[p]] @ weights[p]^T + bscale * biases[p] as views of one
    flat buffer (which is returned too: the demodulation reads it whole)."""
    widths = tuple(int(w.shape[0]) for w in weights)
    meta = (tuple(int(r) for r in rows), widths, float(alpha), float(bscale))
    flat = ModNT.apply(meta, x, *weights, *(biases or ()))
    return flat, _blocks(flat, x.shape[0], widths)


def modulation_supported(x, weights, biases):
    k = x.shape[2]
    ok = (x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 3 and k % 4 == 0 and 0 < x.shape[0] <= 65535
          and all(w.dtype == torch.float32 and w.dim() == 2 and w.shape[1] == k and w.shape[0] % 4 == 0 for w in weights))
    if biases:
        ok = ok and len(biases) == len(weights) and all(t.dtype == torch.float32 for t in biases)
    return ok


# ----------------------------------------------------------------------------------------------------------------------
# demodulation family.  meta = (batch, a_offs, a_total, dims): problem p reads the [B, I_p] block of A at a_offs[p]
# (floats) and owns the p-th [B, J_p] block of Q;  dims[p] = (I_p, J_p)
# ----------------------------------------------------------------------------------------------------------------------
def _q_offsets(meta):
    batch, _, _, dims = meta
    return _offsets(batch, [j for _, j in dims])


def _dem_nn(meta, a, ms):
    batch, a_offs, _, dims = meta
    a = _dense(a)
    q_offs, q_total = _q_offsets(meta)
    out = torch.empty(q_total, device=a.device, dtype=a.dtype)
    ms = [_dense(m) for m in ms]
    ap, op = a.data_ptr(), out.data_ptr()
    n = len(dims)
    with on_device_of(a):
        rc = _lib.lib().sr_bank_nn(
            n, _pa([op + 4 * o for o in q_offs]), _la([j for _, j in dims]), _la([j for _, j in dims]),
            _arr(ctypes.c_int, [1] * n), _pa([ap + 4 * o for o in a_offs]), _pa([m.data_ptr() for m in ms]),
            _la([i for i, _ in dims]), _la([i for i, _ in dims]), batch, 1.0, stream_of(a))
    _lib.check(rc, "sr_bank_nn")
    return out


def _dem_nt(meta, g, ms):
    batch, a_offs, a_total, dims = meta
    g = _dense(g)
    q_offs, _ = _q_offsets(meta)
    covered = sum(batch * i for i, _ in dims)
    out = (torch.empty if covered == a_total else torch.zeros)(a_total, device=g.device, dtype=g.dtype)
    ms = [_dense(m) for m in ms]
    gp, op = g.data_ptr(), out.data_ptr()
    n = len(dims)
    with on_device_of(g):
        rc = _lib.lib().sr_bank_nt(
            n, _pa([op + 4 * o for o in a_offs]), _pa([gp + 4 * o for o in q_offs]), _pa([m.data_ptr() for m in ms]), None,
            _la([j for _, j in dims]), _la([i for i, _ in dims]), _la([j for _, j in dims]), _la([i for i, _ in dims]),
            batch, 1.0, 0.0, stream_of(g))
    _lib.check(rc, "sr_bank_nt")
    return out


def _dem_tn(meta, a, g):
    batch, a_offs, _, dims = meta
    a, g = _dense(a), _dense(g)
    q_offs, _ = _q_offsets(meta)
    outs = _flat_out(None, list(dims), a.device, a.dtype)
    ap, gp = a.data_ptr(), g.data_ptr()
    n = len(dims)
    with on_device_of(a):
        rc = _lib.lib().sr_bank_tn(
            n, _pa([t.data_ptr() for t in outs]), None, _pa([ap + 4 * o for o in a_offs]), _pa([gp + 4 * o for o in q_offs]),
            _la([i for i, _ in dims]), _la([j for _, j in dims]), _la([i for i, _ in dims]), _la([j for _, j in dims]),
            batch, 1.0, 0.0, stream_of(a))
    _lib.check(rc, "sr_bank_tn")
    return outs


class DemNN(Function):
    @staticmethod
    def forward(ctx, meta, a, *ms):
        mark_inputs(ctx, meta, a, *ms)
        ctx.meta = meta
        ctx.save_for_backward(a, *ms)
        return _dem_nn(meta, a, ms)

    @staticmethod
    def backward(ctx, g):
        a, ms = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        need = wanted(ctx)
        ga = DemNT.apply(ctx.meta, g, *ms) if need[1] else None
        gms = DemTN.apply(ctx.meta, a, g) if any(need[2:]) else None
        return (None, ga) + tuple(gms[p] if (gms is not None and need[2 + p]) else None for p in range(len(ms)))


class DemNT(Function):
    @staticmethod
    def forward(ctx, meta, g, *ms):
        mark_inputs(ctx, meta, g, *ms)
        ctx.meta = meta
        ctx.save_for_backward(g, *ms)
        return _dem_nt(meta, g, ms)

    @staticmethod
    def backward(ctx, go):
        g, ms = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        need = wanted(ctx)
        dg = DemNN.apply(ctx.meta, go, *ms) if need[1] else None
        dms = DemTN.apply(ctx.meta, go, g) if any(need[2:]) else None
        return (None, dg) + tuple(dms[p] if (dms is not None and need[2 + p]) else None for p in range(len(ms)))


class DemTN(Function):
    @staticmethod
    def forward(ctx, meta, a, g):
        mark_inputs(ctx, meta, a, g)
        ctx.meta = meta
        ctx.save_for_backward(a, g)
        return tuple(_dem_tn(meta, a, g))

    @staticmethod
    def backward(ctx, *go):
        a, g = ctx.saved_tensors
        dims = ctx.meta[3]
        gom = [t if t is not None else a.new_zeros(dims[p]) for p, t in enumerate(go)]
        need = wanted(ctx)
        da = DemNT.apply(ctx.meta, g, *gom) if need[1] else None
        dg = DemNN.apply(ctx.meta, a, *gom) if need[2] else None
        return None, da, dg


def demod_products(a_flat, batch, a_offs, mats):
    """q_p = A_p @ mats[p] for the [batch, I_p] blocks of `a_flat` at float offsets a_offs[p]: ([B, J_p] views, flat)."""
    dims = tuple((int(m.shape[0]), int(m.shape[1])) for m in mats)
    meta = (int(batch), tuple(int(o) for o in a_offs), int(a_flat.numel()), dims)
    flat = DemNN.apply(meta, a_flat, *mats)
    return flat, _blocks(flat, batch, [j for _, j in dims])


def demod_supported(mats):
    return all(m.device.type == "cuda" and m.dtype == torch.float32 and m.dim() == 2 and m.shape[1] % 4 == 0
               and m.shape[0] % 4 == 0 and m.data_ptr() % 16 == 0 for m in mats)
