"""CPU: checkpoint / dataset contract of the reference (train.py:411-420, 537-556,
generate.py:61-69, dataset.py:56-92, prepare_data.py:95-124)."""
import os

import numpy as np
import pytest
import torch

from stylerenderer_amd import checkpoint, dataset, model, train

SIZE, LATENT, NMLP = 8, 32, 2


def make_trainer(seed=1):
    return train.Trainer(size=SIZE, latent=LATENT, n_mlp=NMLP, device="cpu", seed=seed)


def test_checkpoint_keys_and_name():
    tr = make_trainer()
    st = tr.state_dict()
    assert set(checkpoint.CKPT_KEYS) <= set(st)               # the reference's seven keys
    assert checkpoint.checkpoint_name(10000) == "010000.pt"
    assert checkpoint.checkpoint_name(1234567, 2000000) == "1234567.pt"
    assert checkpoint.start_iter_from_name("/x/checkpoint/020000.pt") == 20000
    assert checkpoint.start_iter_from_name("/x/stylegan2-ffhq-config-f.pt") == 0
    # g_optim is written in the reference's indexing: one slot per generator parameter
    assert len(st["g_optim"]["param_groups"][0]["params"]) == sum(1 for _ in tr.generator.parameters())


def test_save_resume_continues_bit_identically(tmp_path):
    data = train.SyntheticImages(8, SIZE, "cpu", seed=5)
    batches = [data.batch(4) for _ in range(4)]

    def run(tr, bs, start):
        logs = []
        for j, b in enumerate(bs):
            tr.np_rng = np.random.RandomState(100 + start + j)
            np.random.seed(300 + start + j)         # style-mixing inject_index uses numpy's global stream
            torch.manual_seed(200 + start + j)
            logs.append(tr.step(b))
        return logs

    a = make_trainer()
    run(a, batches[:2], 0)
    path = checkpoint.save_checkpoint(str(tmp_path / checkpoint.checkpoint_name(2)), a)
    logs_a = run(a, batches[2:], 2)

    b = make_trainer(seed=9)                                  # different init: everything must come from the file
    ck = checkpoint.load_checkpoint(path, b)
    assert b.iteration == 2 and set(checkpoint.CKPT_KEYS) <= set(ck)
    logs_b = run(b, batches[2:], 2)
    assert logs_a == logs_b
    for (n, p), (_, q) in zip(a.generator.named_parameters(), b.generator.named_parameters()):
        assert torch.equal(p, q), n
    for (n, p), (_, q) in zip(a.g_ema.named_parameters(), b.g_ema.named_parameters()):
        assert torch.equal(p, q), n
    for (n, p), (_, q) in zip(a.discriminator.named_parameters(), b.discriminator.named_parameters()):
        assert torch.equal(p, q), n


def test_reference_style_checkpoint_loads(tmp_path):
    """A checkpoint as the reference writes it: optimiser over ALL generator parameters (incl. the dead
    ToRGB tail), no extension keys, iteration only in the file name."""
    g = model.Generator(SIZE, LATENT, NMLP)
    d = model.Discriminator(SIZE)
    g_optim = torch.optim.Adam(g.parameters(), lr=0.0016, betas=(0.0, 0.99 ** 0.8))
    d_optim = torch.optim.Adam(d.parameters(), lr=0.0019, betas=(0.0, 0.99 ** (16 / 17)))
    img, _ = g([torch.randn(2, LATENT)])
    (img.sum() + 0 * sum(p.sum() for p in g.parameters())).backward()       # every parameter gets Adam state
    g_optim.step()
    d(img.detach()).sum().backward()
    d_optim.step()
    path = str(tmp_path / "030000.pt")
    torch.save({"g": g.state_dict(), "d": d.state_dict(), "g_ema": g.state_dict(), "g_optim": g_optim.state_dict(),
                "d_optim": d_optim.state_dict(), "args": {"size": SIZE}, "ada_aug_p": 0.25}, path)
    tr = make_trainer()
    checkpoint.load_checkpoint(path, tr)
    assert tr.iteration == 30000 and tr.ada_aug_p == 0.25
    names = [n for n, _ in tr.generator.named_parameters()]
    used = [n for n in names if n not in tr.frozen]
    st = tr.g_optim.state_dict()["state"]
    ref = g_optim.state_dict()["state"]
    for i, n in enumerate(used):
        assert torch.equal(st[i]["exp_avg"], ref[names.index(n)]["exp_avg"]), n
    gen = checkpoint.load_generator(path, SIZE, LATENT, NMLP)
    assert not gen.training
    for (n, p), (_, q) in zip(gen.named_parameters(), g.named_parameters()):
        assert torch.equal(p, q), n
    tr.step(train.SyntheticImages(8, SIZE, "cpu").batch(4))                # and training continues


def test_reference_checkpoint_built_from_the_contract_fixture_loads_at_256(golden, tmp_path):
    """The checkpoint is built from the REFERENCE's names / shapes / parameter order (tests/golden/state_dict_contract.npz:
    model.py:86-123, 188-223, 296-336; optimiser over `generator.parameters()` incl. the dead ToRGB tail, train.py:529-536),
    not from this repo's classes."""
    gold = golden("state_dict_contract")

    def shapes(tag):
        return [tuple(int(x) for x in row[1:1 + int(row[0])]) for row in gold[tag + "_shapes"]]

    def state(tag, base):
        return {str(n): torch.full(sh, base + 1e-3 * i) for i, (n, sh) in enumerate(zip(gold[tag + "_names"], shapes(tag)))}

    def optim_state(tag, lr):
        names = [str(n) for n in gold[tag + "_names"]]
        by_name = dict(zip(names, shapes(tag)))
        pnames = [str(n) for n in gold[tag + "_param_names"]]
        st = {i: {"step": torch.tensor(7.0), "exp_avg": torch.full(by_name[n], 1e-4 * (i + 1)),
                  "exp_avg_sq": torch.full(by_name[n], 1e-6 * (i + 1))} for i, n in enumerate(pnames)}
        return {"state": st, "param_groups": [{"lr": lr, "betas": (0.0, 0.99), "eps": 1e-8, "weight_decay": 0,
                                               "amsgrad": False, "params": list(range(len(pnames)))}]}

    path = str(tmp_path / "120000.pt")
    torch.save({"g": state("gm", 0.5), "d": state("d", -0.5), "g_ema": state("gm", 0.25),
                "g_optim": optim_state("gm", 0.0016), "d_optim": optim_state("d", 0.0019),
                "args": {"size": 256}, "ada_aug_p": 0.0}, path)
    tr = train.Trainer(size=256, use_mesh=True, device="cpu")
    checkpoint.load_checkpoint(path, tr, strict=True)                       # strict: every key, no extras
    assert tr.iteration == 120000
    names = [str(n) for n in gold["gm_names"]]
    k = names.index("convs.3.conv.weight")
    assert float(tr.generator.state_dict()["convs.3.conv.weight"].flatten()[0]) == pytest.approx(0.5 + 1e-3 * k)
    assert float(tr.g_ema.state_dict()["convs.3.conv.weight"].flatten()[0]) == pytest.approx(0.25 + 1e-3 * k)
    pnames = [str(n) for n in gold["gm_param_names"]]
    used = [n for n, _ in tr.generator.named_parameters() if n not in tr.frozen]
    assert len(used) < len(pnames)                                           # the dead tail is not optimised here
    st = tr.g_optim.state_dict()["state"]
    for i in (0, len(used) // 2, len(used) - 1):
        assert float(st[i]["exp_avg"].flatten()[0]) == pytest.approx(1e-4 * (pnames.index(used[i]) + 1)), used[i]
    # plain Generator checkpoints (generate.py:61-69) load through the same names
    gpath = str(tmp_path / "g.pt")
    torch.save({"g_ema": state("g", 0.125)}, gpath)
    gen = checkpoint.load_generator(gpath, 256)
    assert float(gen.state_dict()["to_rgbs.11.conv.weight"].flatten()[0]) == pytest.approx(
        0.125 + 1e-3 * [str(n) for n in gold["g_names"]].index("to_rgbs.11.conv.weight"))


def test_multi_resolution_store_roundtrip(tmp_path):
    rng = np.random.RandomState(0)
    imgs = [{16: rng.randint(0, 256, (16, 16, 3)).astype(np.uint8),
             8: rng.randint(0, 256, (8, 8, 3)).astype(np.uint8)} for _ in range(3)]
    store = {}
    assert dataset.write_store(store, imgs, (8, 16), fmt="NPY") == 3
    assert store[b"length"] == b"3" and b"16-00002" in store and b"8-00000" in store      # reference key format
    ds = dataset.MultiResolutionDataset(store, resolution=16)
    assert len(ds) == 3
    x = ds[1]
    assert x.shape == (3, 16, 16) and x.dtype == torch.float32 and -1 <= float(x.min()) and float(x.max()) <= 1
    want = torch.from_numpy(imgs[1][16]).permute(2, 0, 1).float() / 255 * 2 - 1
    assert torch.allclose(x, want, atol=1e-6)
    with pytest.raises(KeyError):
        dataset.MultiResolutionDataset(store, resolution=32)
    # directory-backed store with JPEG payloads (what prepare_data.py writes into LMDB values)
    d = str(tmp_path / "ds")
    dataset.write_store(d, imgs, (16,), fmt="JPEG")
    ds2 = dataset.MultiResolutionDataset(d, transform=dataset.train_transform(np.random.RandomState(1)), resolution=16)
    loader = torch.utils.data.DataLoader(ds2, batch_size=2, drop_last=True)
    batch = next(dataset.sample_data(loader))
    assert batch.shape == (2, 3, 16, 16) and torch.isfinite(batch).all()
    assert os.path.isfile(os.path.join(d, "length"))


def test_generate_cli_writes_images_from_a_checkpoint(tmp_path):
    """reference generate.py:14-75: load 'g_ema', optional truncation, %06d.png files."""
    from stylerenderer_amd import generate

    g = model.Generator(8, 512, 8)
    path = str(tmp_path / "000010.pt")
    torch.save({"g_ema": g.state_dict()}, path)
    out = str(tmp_path / "sample")
    n = generate.main(["--size", "8", "--pics", "2", "--sample", "3", "--truncation", "0.7", "--truncation_mean", "16",
                       "--ckpt", path, "--output", out, "--seed", "1", "--gpu", "-1"])
    assert n == 2 and sorted(os.listdir(out)) == ["000000.png", "000001.png"]
    from PIL import Image

    im = np.asarray(Image.open(os.path.join(out, "000000.png")))
    assert im.shape == (3 * (8 + 2) + 2, 8 + 2 + 2, 3)                     # three samples, one per row, 2 px padding
    grid = generate.to_uint8_grid(torch.tensor([[[[-1.0, 1.0]], [[0.0, 0.0]], [[3.0, -3.0]]]]), padding=0)
    assert grid.tolist() == [[[0, 128, 255], [255, 128, 0]]]


def test_single_image_grid_is_unpadded_like_make_grid():
    """torchvision.utils.make_grid returns a one-image batch without the 2 px border, so the reference's
    default `--sample 1` writes size x size files."""
    from stylerenderer_amd import generate

    one = generate.to_uint8_grid(torch.zeros(1, 3, 8, 8))
    assert one.shape == (8, 8, 3) and int(one[0, 0, 0]) == 128
    assert generate.to_uint8_grid(torch.zeros(2, 3, 8, 8)).shape == (2 * (8 + 2) + 2, 8 + 2 + 2, 3)


def test_lmdb_branch_runs_against_an_lmdb_shaped_module(tmp_path, monkeypatch):
    """dataset.py's LMDB branch executed (never run — `lmdb` is not installable here): a stand-in module
    with the reader API the reference uses (dataset.py:59-67 `lmdb.open(path, max_readers=32, readonly=True, lock=False,
    readahead=False, meminit=False)`, `env.begin(write=False)` as a context manager, `txn.get(key)`,
    `txn.cursor().iternext(values=False)`) serves a store written with the reference's key format
    `"%d-%0Nd"` / `length` (prepare_data.py:99-116), N = max(5, digits)."""
    import sys
    import types

    from stylerenderer_amd import dataset

    rng = np.random.RandomState(0)
    n = 12
    imgs = [{r: rng.randint(0, 256, (r, r, 3), dtype=np.uint8) for r in (8, 16)} for _ in range(n)]
    table = {}
    dataset.write_store(table, imgs, (8, 16), fmt="NPY")
    assert b"8-00000" in table and b"16-00011" in table and table[b"length"] == b"12"
    opened = {}

    class Txn:
        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

        def get(self, key):
            assert isinstance(key, bytes)
            return table.get(key)

        def cursor(self):
            return types.SimpleNamespace(iternext=lambda values=True: iter(sorted(table)))

    class Env:
        def begin(self, write=False):
            assert write is False
            return Txn()

    def lmdb_open(path, **kw):
        opened.update(path=path, **kw)
        return Env()

    monkeypatch.setitem(sys.modules, "lmdb", types.SimpleNamespace(open=lmdb_open))
    root = tmp_path / "faces_lmdb"
    root.mkdir()
    (root / "data.mdb").write_bytes(b"")                       # what makes open_store take the LMDB branch
    ds = dataset.MultiResolutionDataset(str(root), resolution=16)
    assert isinstance(ds.store, dataset._LmdbStore)
    assert opened == dict(path=str(root), max_readers=32, readonly=True, lock=False, readahead=False, meminit=False)
    assert len(ds) == n
    x = ds[11]
    assert x.shape == (3, 16, 16) and x.dtype == torch.float32
    want = torch.from_numpy(imgs[11][16].copy()).permute(2, 0, 1).float().div(255).sub(0.5).div(0.5)
    assert torch.equal(x, want)
    with pytest.raises(KeyError):
        dataset.MultiResolutionDataset(str(root), resolution=32)
    # six-digit stores: the index width grows with the length, like prepare_data.py:100
    assert dataset.make_key(256, 7, 123456) == b"256-000007" and dataset.make_key(256, 7, 99999) == b"256-00007"
