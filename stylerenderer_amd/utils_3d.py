"""Mesh front-end of the rasterizer: the three helpers the reference's training loop calls between the
3DMM and GeneratorWithMap (reference train.py:229-230, 250-251, 305-306) — same names and argument
meaning as reference utils_3d.py / layers.py:

    normalize(vec, axis=-1, _type='L2', eps=1e-8)          layers.py:13-53 (Normalize)
    euler_mat(angle, _type='yxz')                          utils_3d.py:43-80
    random_apply_pose3D(p=[...], v=None)                   utils_3d.py:360-378
    mesh_point_normal(v, tri)                              utils_3d.py:379-404

`mesh_point_normal` on device tensors runs ONE gather kernel (csrc/mesh.hip, C ABI
sr_vertex_normals_f32) over a per-topology incidence list built once and cached, instead of three
sparse.mm scatters over index tensors rebuilt in Python on every call; it is deterministic and
differentiable (first and higher order through the defining tensor algebra).
"""
import torch
from torch.autograd import Function

from . import _lib
from .op._dispatch import DerivedCache, on_device_of, stream_of


def normalize(vec, axis=-1, _type="L2", eps=1e-8):
    eps = abs(eps)
    kind = _type.upper()
    if "L2" in kind:
        norm = torch.sqrt(torch.sum(vec * vec, axis, keepdim=True))
    elif "L1" in kind:
        norm = torch.sum(vec, axis, keepdim=True)              # (sic) the reference sums signed entries
    elif "LINF" in kind:
        norm = torch.max(torch.abs(vec), axis, keepdim=True)[0]
    else:
        raise ValueError("normalize: unknown type %r" % (_type,))
    # the reference's hand-written backward treats the clamp as pass-through: same here
    norm = norm + (torch.clamp(norm, min=eps) - norm).detach()
    return vec / norm


def euler_mat(angle, _type="yxz"):
    """Rotation matrices from Euler angles; axis i of `_type` uses angle[:, i]; later axes multiply
    from the left (T = R_i @ T)."""
    single = angle.dim() == 1
    a = angle.view(1, -1) if single else angle
    c, s = torch.cos(a), torch.sin(a)
    one = torch.ones(len(c), 1, dtype=c.dtype, device=c.device)
    zero = torch.zeros(len(c), 1, dtype=c.dtype, device=c.device)
    T = None
    for i in range(3):
        ci, si = c[:, i:i + 1], s[:, i:i + 1]
        ax = _type[i].lower()
        if ax == "x":
            rows = (one, zero, zero, zero, ci, -si, zero, si, ci)
        elif ax == "y":
            rows = (ci, zero, si, zero, one, zero, -si, zero, ci)
        elif ax == "z":
            rows = (ci, -si, zero, si, ci, zero, zero, zero, one)
        else:
            continue
        R = torch.cat(rows, -1).view(-1, 3, 3)
        T = R if T is None else torch.matmul(R, T)
    return T.view(3, 3) if single else T


def random_apply_pose3D(p=[.5, .1, .05, .1, .1, .1, .15], v=None):
    """p = sigmas of [yaw, pitch, roll, tx, ty, tz, log-scale]; returns the posed vertices
    v @ (scale * R) + t (row-vector convention of the reference), or one 3x4 transform when v is None."""
    batch = len(v) if v is not None and v.dim() >= 3 else 1
    if not isinstance(p, torch.Tensor):
        # sigmas created where the vertices live: the draw below then runs on that device (no host round trip)
        p = torch.tensor([float(x) for x in p], dtype=torch.float32, device=v.device if v is not None else None)
    p = torch.abs(p.reshape(-1)[:7])
    if len(p) < 7:
        p = torch.cat((p, torch.zeros(7 - len(p), dtype=p.dtype, device=p.device)))
    # ~ N(0, diag(p^2)) like the reference's torch.normal(mean=0, std=p[None].expand(batch, -1)) (utils_3d.py:368); drawn as
    # randn * p because torch.normal checks `std >= 0` on the host — a device read hipGraph capture refuses
    z = torch.randn(batch, 7, dtype=p.dtype, device=p.device) * p
    if v is not None and v.device.type == "cuda" and v.dtype == torch.float32 and z.device == v.device:
        # device path: the B scaled rotations in one launch (sr_pose_batch_fwd) and the vertices in one streaming pass
        # (sr_affine3_fwd) instead of ~30 element-wise launches and a 3-wide batched GEMM — this runs twice per training
        # iteration, inside the captured D and G phases
        zc = z.contiguous()
        lin = torch.empty((batch, 3, 3), dtype=zc.dtype, device=zc.device)
        with on_device_of(zc):
            _lib.check(_lib.lib().sr_pose_batch_fwd(_lib.ptr(lin), None, _lib.ptr(zc), batch, stream_of(zc)),
                       "sr_pose_batch_fwd")
        return affine3(v[..., :3].reshape(batch, -1, 3), lin, zc[:, 3:6])
    T = torch.cat((torch.exp(z[:, -1]).view(-1, 1, 1) * euler_mat(z[:, :3], "yxz"), z[:, 3:6].view(-1, 3, 1)), -1)
    if v is None:
        return T[0]
    T = T.to(v.device)
    return torch.matmul(v[..., :3].reshape(batch, -1, 3), T[:, :3, :3]) + T[:, :3, 3:].view(-1, 1, 3)


class _Affine3(Function):
    """out[b] = v[b | 0] @ M[b] (+ t[b]) on the device path (csrc/mesh.hip sr_affine3_fwd / _bwd): one streaming pass
    instead of a 3-wide library GEMM; gradients of M and t by fixed-order sums (deterministic), of v by the same pass
    with M^T.  First order only (poses are leaves of the loops that use it)."""

    @staticmethod
    def forward(ctx, v, m, t):
        vc, mc = v.contiguous(), m.contiguous()
        tc = t.contiguous() if t is not None else None
        b, nv = mc.shape[0], vc.shape[-2]
        out = torch.empty((b, nv, 3), dtype=vc.dtype, device=vc.device)
        share = vc.dim() == 2 or vc.shape[0] == 1
        with on_device_of(vc):
            rc = _lib.lib().sr_affine3_fwd(_lib.ptr(out), _lib.ptr(vc), _lib.ptr(mc), _lib.ptr(tc), b, nv,
                                           0 if share else nv * 3, stream_of(vc))
        _lib.check(rc, "sr_affine3_fwd")
        ctx.save_for_backward(vc, mc)
        ctx.share, ctx.has_t, ctx.v_shape = share, t is not None, tuple(v.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        vc, mc = ctx.saved_tensors
        g = g.contiguous()
        b, nv = mc.shape[0], vc.shape[-2]
        need_v, need_m, need_t = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_t and ctx.needs_input_grad[2]
        gm = torch.empty_like(mc) if need_m else None
        gt = torch.empty((b, 3), dtype=g.dtype, device=g.device) if need_t else None
        gv = None
        L = _lib.lib()
        with on_device_of(g):
            if need_m or need_t:
                _lib.check(L.sr_affine3_bwd(_lib.ptr(gm), _lib.ptr(gt), _lib.ptr(vc), _lib.ptr(g), b, nv,
                                            0 if ctx.share else nv * 3, stream_of(g)), "sr_affine3_bwd")
            if need_v:
                gv = torch.empty_like(g)
                mt = mc.transpose(1, 2).contiguous()
                _lib.check(L.sr_affine3_fwd(_lib.ptr(gv), _lib.ptr(g), _lib.ptr(mt), None, b, nv, nv * 3,
                                            stream_of(g)), "sr_affine3_fwd")
                if ctx.share:
                    gv = gv.sum(0, keepdim=True)
                gv = gv.reshape(ctx.v_shape)
        return gv, gm, gt


class _PoseMatrices(Function):
    """(lin, rot) [1, 3, 3] from pose [7] on the device path (csrc/mesh.hip sr_pose_fwd / _bwd), see `pose_matrices`."""

    @staticmethod
    def forward(ctx, pose):
        pc = pose.contiguous()
        lin = torch.empty((1, 3, 3), dtype=pc.dtype, device=pc.device)
        rot = torch.empty_like(lin)
        with on_device_of(pc):
            _lib.check(_lib.lib().sr_pose_fwd(_lib.ptr(lin), _lib.ptr(rot), _lib.ptr(pc), stream_of(pc)), "sr_pose_fwd")
        ctx.save_for_backward(pc)
        return lin, rot

    @staticmethod
    def backward(ctx, glin, grot):
        (pc,) = ctx.saved_tensors
        gp = torch.empty_like(pc)
        gl = glin.contiguous() if glin is not None else None
        gr = grot.contiguous() if grot is not None else None
        with on_device_of(pc):
            _lib.check(_lib.lib().sr_pose_bwd(_lib.ptr(gp), _lib.ptr(gl), _lib.ptr(gr), _lib.ptr(pc), stream_of(pc)),
                       "sr_pose_bwd")
        return gp


def pose_matrices(pose):
    """pose [7] = (yaw, pitch, roll, tx, ty, tz, log-scale) -> (lin, rot), each [1, 3, 3]: rot = euler_mat(pose[:3],
    "yxz"), lin = exp(pose[6]) * rot.  Device fp32: one launch forward, one backward (first order); otherwise the
    tensor algebra it stands for."""
    if pose.device.type == "cuda" and pose.dtype == torch.float32 and pose.numel() == 7:
        return _PoseMatrices.apply(pose)
    rot = euler_mat(pose[:3].view(1, 3), "yxz")
    return torch.exp(pose[6]) * rot, rot


def affine3(v, m, t=None):
    """v [B or 1, nv, 3] @ m [B, 3, 3] + t [B, 3] (row-vector convention of the reference's pose functions)."""
    if v.device.type == "cuda" and v.dtype == torch.float32 and m.dim() == 3:
        return _Affine3.apply(v, m, t)
    out = torch.matmul(v, m)
    return out + t.view(-1, 1, 3) if t is not None else out


# ---- vertex normals -------------------------------------------------------------------------------
_ADJ_CACHE = DerivedCache(16)


def incidence_lists(tri, nv):
    """CSR incidence of a [nf, 3] triangle list: (adj_off [nv+1] int32, adj [3 nf] int32) with the entries
    of vertex i = ascending corner-major indices k*nf + f such that tri[f, k] == i.  Cached per tensor."""
    key = (tri.data_ptr(), tuple(tri.shape), tri._version, str(tri.device), int(nv))
    hit = _ADJ_CACHE.get(key)
    if hit is not None:
        return hit
    flat = tri.t().reshape(-1)                                   # corner-major: index = k*nf + f
    if flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= nv):
        raise RuntimeError("mesh_point_normal: triangle index out of range [0, %d)" % nv)
    order = torch.sort(flat, stable=True)[1].to(torch.int32)
    counts = torch.bincount(flat, minlength=nv)
    off = torch.zeros(nv + 1, dtype=torch.int32, device=tri.device)
    off[1:] = torch.cumsum(counts, 0).to(torch.int32)
    # keep `tri` alive: the key holds its address
    return _ADJ_CACHE.put(key, (off, order.contiguous(), tri))


def _normals_composite(v, tri):
    """The defining tensor algebra (reference utils_3d.py:380-404 with index_add in place of sparse.mm)."""
    va, vb, vc = v[:, tri[:, 0]], v[:, tri[:, 1]], v[:, tri[:, 2]]
    fn = torch.cross(vb - va, vc - va, dim=2)
    vn = torch.zeros_like(v[:, :, :3])
    for j in range(3):
        vn = vn + torch.zeros_like(vn).index_add_(1, tri[:, j], fn)
    return normalize(vn)


class _VertexNormals(Function):
    @staticmethod
    def forward(ctx, v, tri):
        vc = v.contiguous()
        b, nv, _ = vc.shape
        off, adj, _ = incidence_lists(tri, nv)
        tric = tri.contiguous()
        out = torch.empty_like(vc)
        with on_device_of(vc):
            rc = _lib.lib().sr_vertex_normals_f32(_lib.ptr(out), None, _lib.ptr(vc), _lib.ptr(tric), _lib.ptr(off),
                                                  _lib.ptr(adj), b, nv, tric.size(0), 1e-8, stream_of(vc))
        _lib.check(rc, "sr_vertex_normals_f32")
        ctx.save_for_backward(v, tri)
        return out

    @staticmethod
    def backward(ctx, g):
        v, tri = ctx.saved_tensors
        higher = torch.is_grad_enabled()                         # create_graph=True upstream
        with torch.enable_grad():
            vv = v if (higher and v.requires_grad) else v.detach().requires_grad_(True)
            out = _normals_composite(vv, tri)
            (gv,) = torch.autograd.grad(out, vv, g, create_graph=higher)
        return gv, None


def mesh_point_normal(v, tri):
    """Area-weighted, normalised vertex normals [B, nv, 3] of vertices v [B, nv, >=3] and triangles
    tri [nf, 3] (int64)."""
    if v.dim() != 3 or tri.dim() != 2 or tri.size(1) != 3:
        raise ValueError("mesh_point_normal: expected v [B, nv, 3+] and tri [nf, 3]")
    if v.dtype not in (torch.float32, torch.float64):
        raise ValueError("Not supported type")
    if v.device.type == "cuda" and v.dtype == torch.float32:
        return _VertexNormals.apply(v[:, :, :3], tri)
    return _normals_composite(v[:, :, :3], tri)


# ---- ADA augmentation (reference utils_3d.py:155-188, 189-349 cam=None branch, 350-359) ------------
def _axis_angle_matrix(axis_angle):
    """[B, 3] rotation vectors -> [B, 3, 3] (Rodrigues' formula in closed form)."""
    theta = torch.sqrt((axis_angle * axis_angle).sum(1, keepdim=True)).clamp_min(1e-12)
    k = axis_angle / theta
    kx, ky, kz = k[:, 0], k[:, 1], k[:, 2]
    zero = torch.zeros_like(kx)
    K = torch.stack([zero, -kz, ky, kz, zero, -kx, -ky, kx, zero], 1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=k.dtype, device=k.device).unsqueeze(0)
    s, c = torch.sin(theta).view(-1, 1, 1), torch.cos(theta).view(-1, 1, 1)
    return eye + s * K + (1 - c) * torch.matmul(K, K)


def random_apply_color(p=[.2, .3, 0, .15, .5], img=None):
    """Random brightness / contrast / luma flip / hue rotation / saturation as one 3x4 colour matrix per sample
    (p = sigmas of [brightness, log-contrast], luma-flip probability, sigmas of [hue, log-saturation])."""
    batch = len(img) if img is not None and img.dim() >= 4 else 1
    p = torch.abs(torch.as_tensor(p, dtype=torch.float32).reshape(-1)[:5])
    p = torch.cat((p, torch.zeros(5 - len(p))))
    z = torch.cat([torch.normal(mean=0, std=p[:2].unsqueeze(0).expand(batch, -1)), torch.rand(batch, 1),
                   torch.normal(mean=0, std=p[3:].unsqueeze(0).expand(batch, -1))], 1)
    bright, contrast = z[:, 0], torch.exp(z[:, 1])
    luma = (z[:, 2] < p[2]).to(z.dtype)
    hue, sat = z[:, 3:4], torch.exp(z[:, 4]).view(-1, 1, 1)
    eye = torch.eye(3).unsqueeze(0)
    ones = torch.ones(3, 3).unsqueeze(0)
    C = torch.cat([contrast.view(-1, 1, 1) * eye.expand(batch, -1, -1),
                   (contrast * bright).view(-1, 1, 1).expand(-1, 3, 1)], 2)               # [B, 3, 4]
    C = torch.matmul(eye - luma.view(-1, 1, 1) * 2. / 3, C)
    C = torch.matmul(_axis_angle_matrix(hue.expand(-1, 3) / (3 ** 0.5)), C)
    C = torch.matmul(eye * sat + ones * (1 - sat) / 3., C)
    if img is None:
        return C[0]
    shape = img.shape
    x = img.reshape(batch, -1, shape[-1] * shape[-2])
    C = C.to(dtype=img.dtype, device=img.device)
    return (torch.matmul(C[:, :, :3], x) + C[:, :, 3:4]).view(shape)


def random_apply_pose2D_img(p=[.1, .1, .05, .15, 0, .5], img=None, pad=None):
    """Random translation / in-plane rotation / zoom / horizontal flip of a batch of images by bilinear resampling.
    pad=None: the zoom is raised per sample until the rotated, shifted frame stays inside the source (no border
    shows), the behaviour `augment` relies on; 'zeros' / 'border' / 'reflection' skip that and pad instead."""
    if img is None or img.dim() < 4:
        raise ValueError("random_apply_pose2D_img: a batch of images [B, C, H, W] is required")
    batch, hi, wi = img.shape[0], int(img.shape[-2]), int(img.shape[-1])
    ho, wo = hi, wi
    p = torch.abs(torch.as_tensor(p, dtype=torch.float32).reshape(-1)[:6])
    p = torch.cat((p, torch.zeros(6 - len(p))))
    z = torch.cat([torch.normal(mean=0, std=p[:3].unsqueeze(0).expand(batch, -1)),
                   torch.normal(mean=p[4:5].unsqueeze(0).expand(batch, 1), std=p[3:4].unsqueeze(0).expand(batch, -1)),
                   torch.rand(batch, 1)], 1)
    flip = z[:, 4:5] < p[-1]
    f = torch.exp(z[:, 3:4])
    s, c = torch.sin(z[:, 2:3]), torch.cos(z[:, 2:3])
    tx, ty = z[:, 0:1], z[:, 1:2]
    yy, xx = torch.meshgrid(torch.linspace(0, ho, ho), torch.linspace(0, wo, wo), indexing="ij")
    half = max(wo, ho) / 2.
    x = ((xx.reshape(1, -1) - wo / 2.) / half).expand(batch, -1)
    y = ((ho / 2. - yy.reshape(1, -1)) / half).expand(batch, -1)
    x = torch.where(flip.expand(-1, ho * wo), -x, x) - tx
    y = y - ty
    mode = "zeros"
    if pad is None:
        corner = [0, wo - 1, wo * (ho - 1), ho * wo - 1]
        cx, cy = x[:, corner], y[:, corner]
        rx = (c * cx + s * cy) * max(wo, ho) / float(wi)
        ry = (-s * cx + c * cy) * max(wo, ho) / float(hi)
        fmax = torch.max(torch.abs(torch.cat((rx, ry), 1)), 1, keepdim=True)[0]
        f = torch.where(f < fmax, fmax, f)
    else:
        mode = {"z": "zeros", "b": "border", "r": "reflection"}[str(pad).lower()[0]]
    x, y = x / f, y / f
    x, y = c * x + s * y, -s * x + c * y
    gx = (x * max(wo, ho) / float(wi)).view(-1, ho, wo, 1)
    gy = (-y * max(wo, ho) / float(hi)).view(-1, ho, wo, 1)
    grid = torch.cat((gx, gy), -1).to(dtype=img.dtype, device=img.device)
    # the reference's grid runs linspace(-1, 1) over pixel CENTRES, i.e. it was written for grid_sample's
    # pre-1.3 default (align_corners=True); with today's default the identity transform would blur the image
    return torch.nn.functional.grid_sample(img, grid, mode="bilinear", padding_mode=mode, align_corners=True)


def augment(img, augment_ratio=.5):
    """Per sample, with probability `augment_ratio`: random 2-D pose then random colour (reference utils_3d.py:350-359)."""
    shape = img.shape
    while img.dim() < 4:
        img = img.unsqueeze(0)
    aug = random_apply_color(img=random_apply_pose2D_img(img=img, pad=None))
    pick = torch.rand(img.shape[0], 1, 1, 1, dtype=img.dtype, device=img.device)
    return torch.where(pick.expand_as(img) < augment_ratio, aug, img).view(shape)
