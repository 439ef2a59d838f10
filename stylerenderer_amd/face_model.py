"""3D morphable model in front of the rasterizer — same class name, constructor arguments,
parameter names (`fc.weight`, `fc.bias`, `sigma`) and methods as reference face_model.py:4-74, so
its checkpoints load unchanged.  Vertices = fc(coefficients).view(B, nv, 3): one [B, d] x [d, 3 nv]
GEMM (d = shape + expression dims), a library call on the device."""
import numpy as np
import torch
from torch import nn


def _as_basis(w, rows, dim):
    """Accepts [dim, 3 nv] or [3 nv, dim] (or anything reshapeable to [-1, last]) and returns [dim', 3 nv']."""
    w = np.array(w, np.float32)
    w = w.reshape((-1, w.shape[-1]))
    if w.shape[0] == rows and w.shape[1] >= dim:
        w = w.T
    return w


class LinearMorphableModel(nn.Module):
    def __init__(self, vertices_num, shape_dim=0, expression_dim=0, vertices_mean=None, w_shape_numpy=None,
                 w_expression_numpy=None, sigma_shape=1, sigma_expression=.01, learnable=False):
        super().__init__()
        vertices_num = max(int(vertices_num), 1)
        shape_dim = max(int(shape_dim), 0)
        expression_dim = max(int(expression_dim), 0)
        d = shape_dim + expression_dim
        # random model when no data is given (reference face_model.py:16-19)
        v = (np.random.rand(vertices_num * 3).astype(np.float32) * 2 - 1) * np.sqrt(d)
        w = (np.random.rand(d, v.shape[0]).astype(np.float32) * 2 - 1) * np.sqrt(d)
        if vertices_mean is not None:
            m = np.array(vertices_mean, np.float32)
            if m.shape[0] == 3:
                m = m.reshape(3, -1).T
            elif m.ndim > 1:
                m = m.reshape(-1, m.shape[-1])
            else:
                m = m.reshape(-1, 3)
            n = min(vertices_num, m.shape[0])
            v[:3 * n] = m[:n, :3].reshape(-1)
        if w_shape_numpy is not None:
            ws = _as_basis(w_shape_numpy, w.shape[1], shape_dim)
            k, n = min(shape_dim, ws.shape[0]), min(vertices_num, ws.shape[1] // 3)
            w[:k, :3 * n] = ws[:k, :3 * n]
        if w_expression_numpy is not None and expression_dim > 0:
            we = _as_basis(w_expression_numpy, w.shape[1], expression_dim)
            k, n = min(expression_dim, we.shape[0]), min(vertices_num, we.shape[1] // 3)
            w[shape_dim:shape_dim + k, :3 * n] = we[:k, :3 * n]

        def sigmas(src, count):
            src = [] if src is None else list(np.reshape(src, -1))
            return [abs(src[i]) if len(src) > i else (abs(src[-1]) if src else 1) for i in range(count)]

        self.dim = [shape_dim, expression_dim, vertices_num * 3]
        self.fc = nn.Linear(d, vertices_num * 3, bias=True)
        self.sigma = nn.Parameter(torch.Tensor(sigmas(sigma_shape, shape_dim) + sigmas(sigma_expression, expression_dim)),
                                  requires_grad=False)
        with torch.no_grad():
            self.fc.weight.copy_(torch.from_numpy(w.T).float())
            self.fc.bias.copy_(torch.from_numpy(v).float())
        if not learnable:
            self.fc.weight.requires_grad = False
            self.fc.bias.requires_grad = False

    def random_input(self, batch_size=1):
        # = torch.normal(mean=0, std=sigma[None].expand(batch, -1)) (reference face_model.py:69-70: same distribution, another draw) without its host-side
        # check of `std >= 0`, which reads the device and is refused inside a hipGraph capture (graph_train samples the
        # meshes inside the captured D / G phases)
        return torch.randn(batch_size, self.sigma.shape[0], device=self.sigma.device, dtype=self.sigma.dtype) * self.sigma

    def forward(self, x):
        return torch.reshape(self.fc(x), (-1, self.dim[2] // 3, 3))

    def regulation(self, x):
        return ((x / self.sigma[np.newaxis, :]) ** 2).sum()


def load_bfm(file_name="/data/BaselFaceModel.mat"):
    """Basel Face Model -> (LinearMorphableModel, tri int64 [nf, 3]) with the reference's `.mat` contract
    (reference face_model.py:342-362): keys `v` [3, nv] (mean shape), `w_shape` [3 nv, ds], `w_exp` [3 nv, de],
    optional `sigma_shape` / `sigma_exp` (folded into the bases), `tri` as a 1x1 MATLAB cell of 1-based indices.
    Coordinates are centred and scaled by 1e-5 like the reference.  `file_name` may be the path of the (licensed,
    not distributed) file or an already loaded dict."""
    if isinstance(file_name, str):
        import scipy.io as sio

        data = sio.loadmat(file_name)
    else:
        data = file_name
    v = (data["v"] - data["v"].mean(1).reshape(-1, 1)).T * 1e-5
    w_shape = data["w_shape"] * 1e-5
    w_exp = data["w_exp"] * 1e-5
    if "sigma_shape" in data.keys():
        w_shape = w_shape.dot(np.diag(np.reshape(data["sigma_shape"], -1)))
    if "sigma_exp" in data.keys():
        w_exp = w_exp.dot(np.diag(np.reshape(data["sigma_exp"], -1)))
    tri = np.asarray(data["tri"][0, 0]).astype(np.int64)
    tri = tri - tri.min()
    if tri.shape[0] == 3 and tri.shape[1] != 3:
        tri = tri.T
    model = LinearMorphableModel(len(v), w_shape.shape[1], w_exp.shape[1], v, w_shape, w_exp)
    return model, torch.from_numpy(np.ascontiguousarray(tri))
