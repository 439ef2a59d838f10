"""Shared helpers of the operator wrappers."""
import contextlib

import torch

from .. import _lib


def is_device_tensor(t):
    return t.device.type == "cuda"


def require_f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError("%s: expected a float32 tensor, got %s" % (name, t.dtype))


@contextlib.contextmanager
def on_device_of(t):
    """Makes the tensor's GPU current for the duration of a native launch (one process per GPU
    normally makes this a no-op)."""
    idx = t.device.index
    if idx is None or idx == torch.cuda.current_device():
        yield
    else:
        with torch.cuda.device(idx):
            yield


def stream_of(t):
    return _lib.current_stream(t.device)
