"""PCN-family dataset readers on a tiny on-disk fixture tree built by the test itself."""
import json
import os

import numpy as np
import pytest
import torch


def _write_pcd(path, pts, binary=False):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
            f"COUNT 1 1 1\nWIDTH {len(pts)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\n")
    with open(path, "wb") as fh:
        if binary:
            fh.write((head + "DATA binary\n").encode())
            fh.write(pts.astype("<f4").tobytes())
        else:
            fh.write((head + "DATA ascii\n").encode())
            for p in pts:
                fh.write(("%.8g %.8g %.8g\n" % tuple(p)).encode())


@pytest.fixture()
def tree(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    cats = [{"taxonomy_id": "02691156", "train": ["a1", "a2"], "val": ["a3"], "test": ["a3"]},
            {"taxonomy_id": "02958343", "train": ["c1"], "val": ["c2"], "test": ["c2"]}]
    (tmp_path / "PCN.json").write_text(json.dumps(cats))
    clouds = {}
    for dc in cats:
        for subset in ("train", "val", "test"):
            for s in dc[subset]:
                gt = rng.random((16384, 3)).astype(np.float32) - 0.5
                _write_pcd(str(tmp_path / subset / "complete" / dc["taxonomy_id"] / f"{s}.pcd"), gt, binary=True)
                clouds[(subset, s)] = gt
                for i in range(8 if subset == "train" else 1):
                    n = int(rng.integers(700, 3000))
                    _write_pcd(str(tmp_path / subset / "partial" / dc["taxonomy_id"] / s / f"{i:02d}.pcd"), gt[:n])
                for v in range(24):
                    d = tmp_path / subset / "image" / dc["taxonomy_id"] / s
                    d.mkdir(parents=True, exist_ok=True)
                    Image.fromarray((rng.random((137, 137, 3)) * 255).astype(np.uint8)).save(d / f"{v:02d}.png")
    return tmp_path, clouds


def _cfg(root, subset, name, with_img=False):
    from imgadapointr_amd.utils.config import EasyDict
    c = EasyDict(NAME=name, CATEGORY_FILE_PATH=str(root / "PCN.json"), N_POINTS=16384, N_RENDERINGS=8,
                 PARTIAL_POINTS_PATH=str(root) + "/%s/partial/%s/%s/%02d.pcd",
                 COMPLETE_POINTS_PATH=str(root) + "/%s/complete/%s/%s.pcd", CARS=False)
    if with_img:
        c.IMG_PATH = str(root) + "/%s/image/%s/%s/%02d.png"
    return c, EasyDict(subset=subset)


def test_pcd_reader_ascii_and_binary(tmp_path):
    from imgadapointr_amd.utils.io_module import IO
    pts = (np.random.default_rng(1).random((100, 3)).astype(np.float32) - 0.5)
    _write_pcd(str(tmp_path / "a.pcd"), pts)
    _write_pcd(str(tmp_path / "b.pcd"), pts, binary=True)
    assert np.allclose(IO.get(str(tmp_path / "a.pcd")), pts, atol=1e-7)
    assert np.array_equal(IO.get(str(tmp_path / "b.pcd")).astype(np.float32), pts)
    np.save(tmp_path / "c.npy", pts)
    assert np.array_equal(IO.get(str(tmp_path / "c.npy")), pts)
    with pytest.raises(Exception):
        IO.get(str(tmp_path / "x.xyz"))


def test_pcn_dataset_items(tree):
    from imgadapointr_amd.datasets import build_dataset_from_cfg
    root, clouds = tree
    base, others = _cfg(root, "train", "PCN")
    ds = build_dataset_from_cfg(base, others)
    assert len(ds) == 3
    tax, mid, (partial, gt) = ds[0]
    assert tax == "02691156" and mid == "a1"
    assert partial.shape == (2048, 3) and gt.shape == (16384, 3) and partial.dtype == torch.float32
    # the mirror augmentation only flips signs of x / z: |gt| is the stored cloud
    assert np.allclose(np.abs(gt.numpy()), np.abs(clouds[("train", "a1")]), atol=1e-6)
    base, others = _cfg(root, "val", "PCN")
    ds = build_dataset_from_cfg(base, others)
    tax, mid, (partial, gt) = ds[1]
    assert np.allclose(gt.numpy(), clouds[("val", "c2")], atol=1e-6)     # no augmentation in val
    nz = int((partial.abs().sum(-1) > 0).sum())
    assert nz == 2048 or bool((partial[nz:] == 0).all())                 # zero padding at the tail


def test_img_and_seg_datasets(tree):
    from imgadapointr_amd.datasets import build_dataset_from_cfg
    root, _ = tree
    base, others = _cfg(root, "val", "ImgPCN", with_img=True)
    tax, mid, (partial, gt, img) = build_dataset_from_cfg(base, others)[0]
    assert img.shape == (3, 224, 224) and 0.0 <= float(img.min()) and float(img.max()) <= 1.0
    base, others = _cfg(root, "val", "SegImgPCN", with_img=True)
    ds = build_dataset_from_cfg(base, others)
    _, _, (partial, gt, img, cls_vec) = ds[0]
    assert cls_vec.shape == (16,) and cls_vec[0] == 1 and cls_vec.sum() == 1      # airplane -> class 0
    _, _, (_, _, _, cls_vec) = ds[1]
    assert cls_vec.sum() == 0                                                     # car: not in SEG_LIST


def test_transforms():
    from imgadapointr_amd.datasets import data_transforms as T
    np.random.seed(0)
    pts = np.random.rand(100, 3).astype(np.float32)
    out = T.RandomSamplePoints({"n_points": 128})(pts.copy())
    assert out.shape == (128, 3) and (out[100:] == 0).all()
    assert {tuple(r) for r in out[:100]} == {tuple(r) for r in pts}
    assert T.UpSamplePoints({"n_points": 350})(pts.copy()).shape == (350, 3)
    assert T.UpSamplePoints({"n_points": 40})(pts.copy()).shape == (40, 3)
    m = T.RandomMirrorPoints()
    assert np.allclose(m(pts.copy(), 0.1), pts * [-1, 1, -1])
    assert np.allclose(m(pts.copy(), 0.3), pts * [-1, 1, 1])
    assert np.allclose(m(pts.copy(), 0.6), pts * [1, 1, -1])
    assert np.allclose(m(pts.copy(), 0.9), pts)


def _lzf_compress(data):
    """A small greedy liblzf-format encoder for the test (literal runs <= 32, back references found by a 3-byte hash)."""
    out, lit, i, n, table = bytearray(), bytearray(), 0, len(data), {}

    def flush():
        for s in range(0, len(lit), 32):
            chunk = lit[s:s + 32]
            out.append(len(chunk) - 1)
            out.extend(chunk)
        lit.clear()

    while i < n:
        key = bytes(data[i:i + 3])
        ref = table.get(key) if len(key) == 3 else None
        if len(key) == 3:
            table[key] = i
        if ref is not None and 0 < i - ref <= 8192:
            length = 3
            while i + length < n and length < 264 and data[ref + length] == data[i + length]:
                length += 1
            flush()
            off, l2 = i - ref - 1, length - 2
            if l2 < 7:
                out.append((l2 << 5) | (off >> 8))
            else:
                out.append((7 << 5) | (off >> 8))
                out.append(l2 - 7)
            out.append(off & 255)
            i += length
        else:
            lit.append(data[i])
            i += 1
    flush()
    return bytes(out)


def test_pcd_reader_binary_compressed(tmp_path):
    """PCL's DATA binary_compressed: LZF over a field-major block, with an extra non-xyz field in between."""
    from imgadapointr_amd.utils.io_module import IO, lzf_decompress
    # the decoder on its own: literals, a plain back reference and an overlapping (run-length) one
    assert lzf_decompress(bytes([2]) + b"abc" + bytes([(1 << 5) | 0, 2]), 6) == b"abcabc"
    assert lzf_decompress(bytes([0]) + b"a" + bytes([(7 << 5) | 0, 1, 0]), 11) == b"a" * 11
    with pytest.raises(ValueError):
        lzf_decompress(bytes([(1 << 5) | 0, 5]), 3)
    rng = np.random.default_rng(3)
    pts = np.round(rng.random((700, 3)) * 64).astype(np.float32) / 64 - 0.5      # repetitive bytes: real back references
    rgb = rng.integers(0, 2 ** 24, size=700).astype("<u4")
    raw = pts[:, 0].astype("<f4").tobytes() + rgb.tobytes() + pts[:, 1].astype("<f4").tobytes() + pts[:, 2].astype("<f4").tobytes()
    comp = _lzf_compress(raw)
    assert len(comp) < len(raw) and lzf_decompress(comp, len(raw)) == raw
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x rgb y z\nSIZE 4 4 4 4\nTYPE F U F F\n"
            "COUNT 1 1 1 1\nWIDTH 700\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 700\nDATA binary_compressed\n")
    with open(tmp_path / "c.pcd", "wb") as fh:
        fh.write(head.encode())
        fh.write(np.asarray([len(comp), len(raw)], dtype="<u4").tobytes())
        fh.write(comp)
    assert np.array_equal(IO.get(str(tmp_path / "c.pcd")).astype(np.float32), pts)


def test_readers_and_transforms_match_the_reference_on_the_demo_scans(tmp_path):
    """tests/golden/demo_data.npz: the reference's PCN / PCNv2 dataset classes (reader -> RandomSamplePoints /
    UpSamplePoints -> RandomMirrorPoints -> ToTensor) run over a tree made of the 8 real scans of its demo/samples,
    seeded RNGs (generator: tests/golden/make_golden_round3.py).  The same tree is rebuilt here from the fixture's
    raw clouds; items must agree bit for bit: file parsing, RNG draw order, padding, mirroring."""
    import hashlib
    import random
    from golden_util import build_demo_tree, demo_tree_spec
    from imgadapointr_amd.datasets import build_dataset_from_cfg
    from imgadapointr_amd.utils.config import EasyDict
    from imgadapointr_amd.utils.io_module import IO
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_data.npz"))
    spec = demo_tree_spec()
    raws = {t: g[f"raw/{t}"] for t in spec["taxonomies"]}
    assert sorted(len(r) for r in raws.values())[0] >= 700          # real scans: 708 .. 1850 points
    root = str(tmp_path)
    build_demo_tree(root, raws)
    for t in spec["taxonomies"]:                                      # the reader returns the scan's float32 values
        got = IO.get(os.path.join(root, "val", "partial", t, "v0", "00.pcd"))
        assert got.dtype == np.float64 and np.array_equal(got.astype(np.float32), raws[t])
    n_items = 0
    for name in ("PCN", "PCNv2"):
        for subset in ("train", "val"):
            base = EasyDict(NAME=name, CATEGORY_FILE_PATH=os.path.join(root, "PCN.json"), N_POINTS=16384,
                            N_RENDERINGS=8, CARS=False, PARTIAL_POINTS_PATH=root + "/%s/partial/%s/%s/%02d.pcd",
                            COMPLETE_POINTS_PATH=root + "/%s/complete/%s/%s/PCN/%s.pcd")
            ds = build_dataset_from_cfg(base, EasyDict(subset=subset))
            random.seed(spec["seed"])
            np.random.seed(spec["seed"])
            ids = [str(x) for x in g[f"{name}/{subset}/ids"]]
            assert len(ds) == len(ids) == 8
            for i in range(len(ds)):
                tax, mid, (partial, gt) = ds[i]
                assert f"{tax}/{mid}" == ids[i]
                want = g[f"{name}/{subset}/{i}/partial"]
                assert partial.dtype == torch.float32 and tuple(partial.shape) == want.shape == (2048, 3)
                assert np.array_equal(partial.numpy(), want), (name, subset, i)
                if name == "PCN":                                    # zero padding of the short scans survives
                    assert int((np.abs(want).sum(-1) == 0).sum()) == max(0, 2048 - (len(raws[tax]) - 0))  \
                        or subset == "train"
                assert np.array_equal(gt[:64].numpy(), g[f"{name}/{subset}/{i}/gt_head"])
                sha = np.frombuffer(hashlib.sha1(np.ascontiguousarray(gt.numpy()).tobytes()).digest(), dtype=np.uint8)
                assert np.array_equal(sha, g[f"{name}/{subset}/{i}/gt_sha1"])
                n_items += 1
    assert n_items == 32


def test_image_datasets_match_the_reference_on_the_demo_tree(tmp_path):
    """tests/golden/demo_img_data.npz: the reference's ImgPCN / SegImgPCN classes (datasets/ImgPCNDataset.py,
    SegImgPCNDataset.py over BasePCNDataset.py:36-113) run over the demo tree plus 24 seeded renders per object: which
    view survives the sampling loop (:72-75 - ImgPCN draws 1 view, SegImgPCN 6 and keeps the last), the PIL load +
    Resize(224) + ToTensor result bit for bit (square and non-square renders), the class vector, the point branch."""
    import hashlib
    import random
    from golden_util import build_demo_images, build_demo_tree, demo_tree_spec
    from imgadapointr_amd.datasets import build_dataset_from_cfg
    from imgadapointr_amd.utils.config import EasyDict
    here = os.path.join(os.path.dirname(__file__), "golden")
    g, raw = np.load(os.path.join(here, "demo_img_data.npz")), np.load(os.path.join(here, "demo_data.npz"))
    spec = demo_tree_spec()
    root = str(tmp_path)
    build_demo_tree(root, {t: raw[f"raw/{t}"] for t in spec["taxonomies"]})
    build_demo_images(root)
    sha = lambda a: np.frombuffer(hashlib.sha1(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)   # noqa: E731
    n_items, shapes = 0, set()
    for name in ("ImgPCN", "SegImgPCN"):
        for subset in ("train", "val"):
            base = EasyDict(NAME=name, CATEGORY_FILE_PATH=os.path.join(root, "PCN.json"), N_POINTS=16384, N_RENDERINGS=8,
                            CARS=False, PARTIAL_POINTS_PATH=root + "/%s/partial/%s/%s/%02d.pcd",
                            COMPLETE_POINTS_PATH=root + "/%s/complete/%s/%s/PCN/%s.pcd",
                            IMG_PATH=root + "/%s/complete/%s/%s/ShapeNetCoreV1/projections/%02d.png")
            random.seed(spec["seed"] + 1)
            np.random.seed(spec["seed"] + 1)
            ds = build_dataset_from_cfg(base, EasyDict(subset=subset))
            views = [int(os.path.basename(f["img_path"])[:2]) for f in ds.file_list]
            assert views == g[f"{name}/{subset}/views"].tolist(), (name, subset)
            assert [str(x) for x in g[f"{name}/{subset}/ids"]] == [f"{f['taxonomy_id']}/{f['model_id']}" for f in ds.file_list]
            for i in range(len(ds)):
                tax, mid, item = ds[i]
                partial, gt, img = item[:3]
                assert np.array_equal(sha(partial.numpy()), g[f"{name}/{subset}/{i}/partial_sha1"]), (name, subset, i)
                assert list(img.shape) == g[f"{name}/{subset}/{i}/img_shape"].tolist()
                assert img.dtype == torch.float32 and np.array_equal(img[:, :8, :8].numpy(), g[f"{name}/{subset}/{i}/img_head"])
                assert np.array_equal(sha(img.numpy()), g[f"{name}/{subset}/{i}/img_sha1"]), (name, subset, i)
                shapes.add(tuple(img.shape))
                if name == "SegImgPCN":
                    assert np.array_equal(item[3].numpy(), g[f"{name}/{subset}/{i}/cls_vec"])
                else:
                    assert len(item) == 3
                n_items += 1
    assert n_items == 32 and (3, 224, 224) in shapes and (3, 224, 261) in shapes     # 160 x 137 -> 261 x 224 (truncated)
