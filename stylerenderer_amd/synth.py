"""Synthetic inputs for tests and benchmarks (numpy only, closed-form, no RNG across machines
unless a seed is passed).  The licensed Basel Face Model the reference loads
(reference face_model.py:342) is absent, so SURVEY.md §8(d) prescribes a formula mesh of the
same size class (nv ~ 25k, nf ~ 50k) for the rasterizer configuration."""
import numpy as np


def uv_ellipsoid(n_lat, n_lon, radii=(0.8, 0.95, 0.6)):
    """Closed UV ellipsoid.  Returns (v [nv,3] float32, tri [nf,3] int64).

    Faces are wound counter-clockwise when seen from outside, which is what the
    rasterizer keeps (reference op/rasterize.h:56 culls det > eps, i.e. clockwise in its
    y-down screen space).  nv = (n_lat-1)*n_lon + 2, nf = 2*n_lon*(n_lat-1).
    """
    lat = np.pi * np.arange(1, n_lat) / n_lat                       # exclude poles
    lon = 2 * np.pi * np.arange(n_lon) / n_lon
    st, ct = np.sin(lat)[:, None], np.cos(lat)[:, None]
    ring = np.stack([st * np.cos(lon)[None], ct * np.ones_like(lon)[None],
                     st * np.sin(lon)[None]], -1).reshape(-1, 3)
    v = np.concatenate([[[0.0, 1.0, 0.0]], ring, [[0.0, -1.0, 0.0]]], 0)
    v = (v * np.asarray(radii)[None]).astype(np.float32)
    top, bot = 0, v.shape[0] - 1

    def vid(i, j):
        return 1 + i * n_lon + (j % n_lon)

    tris = []
    for j in range(n_lon):
        tris.append((top, vid(0, j + 1), vid(0, j)))
        tris.append((bot, vid(n_lat - 2, j), vid(n_lat - 2, j + 1)))
    for i in range(n_lat - 2):
        for j in range(n_lon):
            a, b, c, d = vid(i, j), vid(i, j + 1), vid(i + 1, j), vid(i + 1, j + 1)
            tris.append((a, b, c))
            tris.append((b, d, c))
    return v, np.asarray(tris, np.int64)


def face_sized_mesh():
    """~25k vertices / ~50k triangles (BFM size class): nv = 24 962, nf = 49 920."""
    return uv_ellipsoid(130, 192)


def vertex_normals(v, tri):
    """Area-weighted vertex normals [.., nv, 3] (inputs to `rasterize` as `tex`, the role
    reference utils_3d.py:379-404 plays)."""
    v = np.asarray(v, np.float64)
    batched = v.ndim == 3
    vb = v if batched else v[None]
    n = np.zeros_like(vb)
    p0, p1, p2 = vb[:, tri[:, 0]], vb[:, tri[:, 1]], vb[:, tri[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0)
    for k in range(3):
        for s in range(vb.shape[0]):
            np.add.at(n[s], tri[:, k], fn[s])
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    n = n.astype(np.float32)
    return n if batched else n[0]


def random_poses(v, batch, seed=1234, yaw_sigma=0.5, scale_sigma=0.15):
    """Per-sample yaw / pitch rotation + isotropic scale of a mesh: [batch, nv, 3] float32."""
    rng = np.random.RandomState(seed)
    out = np.empty((batch,) + v.shape, np.float32)
    for s in range(batch):
        yaw, pitch = rng.randn() * yaw_sigma, rng.randn() * 0.3 * yaw_sigma
        sc = float(np.exp(rng.randn() * scale_sigma))
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        out[s] = (v.astype(np.float64) @ (rx @ ry).T * sc).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------
# Deterministic, RNG-free tensor fills (integer hash -> float32), identical on every machine.
def _mix64(idx, key):
    with np.errstate(over="ignore"):
        h = idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(
            (int(key) * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return h


def det_uniform(shape, key):
    """float32 in [-1, 1), a pure function of (flat index, key)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = _mix64(np.arange(n, dtype=np.uint64), key)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)       # 24 bits -> [0,1)
    return (u * 2.0 - 1.0).astype(np.float32).reshape(shape)


def det_normal(shape, key):
    """Approximately N(0,1) float32: scaled sum of four independent det_uniform streams."""
    acc = np.zeros(shape, np.float64)
    for j in range(4):
        acc += det_uniform(shape, key * 4 + j + 1000003)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def sample_index(numel, count=256):
    """Evenly spaced flat indices (all of them when numel <= count): the gradient samples of the golden fixtures."""
    if numel <= count:
        return np.arange(numel, dtype=np.int64)
    return np.linspace(0, numel - 1, count).astype(np.int64)


def name_key(name):
    import zlib

    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def fill_state_dict(state_dict, salt=0):
    """Overwrite every floating tensor of a torch state_dict in place with a value that
    depends only on (key name, shape, salt).  Weights ~ N(0,1) (mapping-network weights keep
    the reference's 1/lr_mul = 100 gain, reference layers.py:226), biases and noise strengths get
    small non-zero values so that every term of the forward pass is exercised; FIR kernels and
    other buffers named '*.kernel' are left untouched."""
    import torch

    for name, t in state_dict.items():
        if not torch.is_floating_point(t) or name.endswith(".kernel"):
            continue
        key = name_key(name) + salt
        val = det_normal(tuple(t.shape), key)
        if name.endswith("modulation.bias"):
            val = 1.0 + 0.1 * val
        elif name.endswith("bias") or name.endswith("noise.weight"):
            val = 0.1 * val
        elif name.startswith("style.") and name.endswith("weight"):
            val = 100.0 * val
        t.copy_(torch.from_numpy(np.ascontiguousarray(val)).to(t.dtype))
    return state_dict
