"""Dataset contract of the reference: reference dataset.py:56-92 (reader) and
prepare_data.py:95-124 (writer).

Key/value layout of the multi-resolution store:
    b'length'                      -> ASCII decimal number of images
    b'<resolution>-<index, zero padded to max(5, digits(length))>'   -> one encoded image (JPEG bytes)
The tensor contract handed to the training step (reference train.py:557-560): random horizontal flip,
ToTensor, Normalize(0.5, 0.5) -> float32 [3, R, R] in [-1, 1].

The reference reads an LMDB environment; `lmdb` is not installable here, so the reader takes any store
with the same get(key) -> bytes contract: an LMDB environment when the module is importable, or a
`DictStore` (a directory of <key> files / an in-memory dict) produced by `write_store` below.  The image
decode path needs no torchvision: PIL when present, else raw .npy payloads.
"""
import io
import math
import os

import numpy as np
import torch


def index_bits(length):
    """Zero padding of the per-image keys: max(floor(log10(length)) + 1, 5) (prepare_data.py:100)."""
    return max(int(math.floor(math.log(max(length, 1)) / math.log(10))) + 1, 5)


def make_key(resolution, index, length):
    return ("%d-%%0%dd" % (resolution, index_bits(length)) % index).encode("utf-8")


class DictStore:
    """Minimal key/value store with the reader side of the LMDB API the dataset uses."""

    def __init__(self, source):
        self.source = source                      # dict (in memory) or a directory path

    def get(self, key):
        if isinstance(self.source, dict):
            return self.source.get(key)
        path = os.path.join(self.source, key.decode("utf-8"))
        if not os.path.isfile(path):
            return None
        with open(path, "rb") as f:
            return f.read()

    def keys(self):
        if isinstance(self.source, dict):
            return list(self.source.keys())
        return [n.encode("utf-8") for n in sorted(os.listdir(self.source))]


class _LmdbStore:
    def __init__(self, path):
        import lmdb

        self.env = lmdb.open(path, max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
        if not self.env:
            raise IOError("Cannot open lmdb dataset", path)

    def get(self, key):
        with self.env.begin(write=False) as txn:
            return txn.get(key)

    def keys(self):
        with self.env.begin(write=False) as txn:
            return list(txn.cursor().iternext(values=False))


def open_store(path_or_store):
    if hasattr(path_or_store, "get") and hasattr(path_or_store, "keys") and not isinstance(path_or_store, (str, bytes)):
        return path_or_store if not isinstance(path_or_store, dict) else DictStore(path_or_store)
    if os.path.isfile(os.path.join(path_or_store, "data.mdb")):
        return _LmdbStore(path_or_store)
    if os.path.isdir(path_or_store):
        return DictStore(path_or_store)
    raise IOError("Cannot open dataset", path_or_store)


def encode_image(img_u8, fmt="JPEG"):
    """uint8 [H, W, 3] -> bytes (JPEG like prepare_data.save_img; 'NPY' = lossless raw payload)."""
    if fmt.upper() == "NPY":
        buf = io.BytesIO()
        np.save(buf, np.ascontiguousarray(img_u8))
        return buf.getvalue()
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(img_u8).save(buf, format=fmt)
    return buf.getvalue()


def decode_image(payload):
    """bytes -> uint8 [H, W, 3] (RGB)."""
    if payload[:6] == b"\x93NUMPY":
        return np.load(io.BytesIO(payload))
    from PIL import Image

    return np.asarray(Image.open(io.BytesIO(payload)).convert("RGB"))


def write_store(target, images, resolutions, fmt="JPEG"):
    """prepare_data.prepare(): `images` = iterable of uint8 [H, W, 3] arrays already at each resolution
    (dict resolution -> array) ; `target` = dict or directory.  Returns the number of images written."""
    images = list(images)
    total = len(images)
    put = (target.__setitem__ if isinstance(target, dict)
           else lambda k, v: open(os.path.join(target, k.decode("utf-8")), "wb").write(v))
    if not isinstance(target, dict):
        os.makedirs(target, exist_ok=True)
    for i, per_res in enumerate(images):
        for r in resolutions:
            put(make_key(r, i, total), encode_image(per_res[r], fmt))
    put(b"length", str(total).encode("utf-8"))
    return total


class MultiResolutionDataset(torch.utils.data.Dataset):
    """reference dataset.py:56-92.  `transform` receives the decoded uint8 HWC image; the default is the
    training transform of train.py:557-560 without the flip (use `train_transform` for that)."""

    def __init__(self, path, transform=None, resolution=256):
        super().__init__()
        self.store = open_store(path)
        raw = self.store.get(b"length")
        if raw is None:
            raise IOError("Cannot open dataset: no 'length' key", path)
        self.length = int(raw.decode("utf-8"))
        res = set()
        for key in self.store.keys():
            try:
                res.add(int(key.decode("utf-8").split("-")[0]))
            except ValueError:
                pass
        if resolution not in res:
            raise KeyError("No specified resolution", sorted(res))
        self.resolution = resolution
        self.transform = transform if transform is not None else to_unit_tensor

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        payload = self.store.get(make_key(self.resolution, index, self.length))
        if payload is None:
            raise KeyError(index)
        return self.transform(decode_image(payload))


def to_unit_tensor(img_u8):
    """ToTensor + Normalize((.5,.5,.5), (.5,.5,.5)): uint8 HWC -> float32 CHW in [-1, 1]."""
    t = torch.from_numpy(np.array(img_u8, copy=True)).permute(2, 0, 1).to(torch.float32).div_(255.0)
    return t.sub_(0.5).div_(0.5)


def train_transform(rng=None):
    """RandomHorizontalFlip -> ToTensor -> Normalize (reference train.py:557-560)."""
    rng = rng or np.random

    def apply(img_u8):
        if rng.rand() < 0.5:
            img_u8 = img_u8[:, ::-1]
        return to_unit_tensor(img_u8)

    return apply


def sample_data(loader):
    """reference train.py:180-183: endless iterator over a DataLoader."""
    while True:
        for batch in loader:
            yield batch
