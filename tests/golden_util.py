"""Helpers shared by tests/golden/make_golden.py and the tests: deterministic weights and the
synthetic inputs of SURVEY §8d.  Pure numpy/torch; knows nothing about the reference."""
import zlib

import numpy as np
import torch


def seeded_state_dict(sd, tag, std=0.02):
    """A state_dict with the keys/shapes/dtypes of `sd`, filled as a pure function of
    (tag, key, shape).  1-D '*.weight' (norm scales) are 1 + 0.1 n with mixed small deviations,
    running_var is positive, integer buffers are zero, everything else is std * n."""
    out = {}
    for key in sd:
        ref = sd[key]
        g = torch.Generator().manual_seed(zlib.crc32(f"{tag}/{key}".encode()) & 0x7FFFFFFF)
        if not ref.is_floating_point():
            out[key] = torch.zeros_like(ref)
            continue
        n = torch.randn(ref.shape, generator=g, dtype=torch.float32)
        if key.endswith("running_var"):
            t = 1.0 + 0.1 * n.abs()
        elif key.endswith("running_mean"):
            t = 0.05 * n
        elif ref.dim() == 1 and key.endswith("weight"):
            t = 1.0 + 0.1 * n
        elif "query_ranking" in key:
            # wide scores: the top-512-of-768 query selection (argsort) must not hinge on 1e-7 noise
            # (std 0.3 gives pre-sigmoid scores of order 1: spread out, not saturated)
            t = (0.3 if key.endswith("weight") else 0.1) * n
        else:
            t = std * n
        out[key] = t.to(ref.dtype)
    return out


def synthetic_batch(B, n_partial, n_gt, seed=0, structured=False):
    """(partial [B,n_partial,3], gt [B,n_gt,3]) float32.

    Plain: uniform in [-0.5, 0.5)^3.  Structured: points on sphere shells of random radius, the
    last 25 % of the partial rows zeroed (the zero padding RandomSamplePoints produces,
    datasets/data_transforms.py:55-67 of the reference)."""
    rng = np.random.default_rng(seed)
    if not structured:
        return (rng.random((B, n_partial, 3), dtype=np.float32) - 0.5,
                rng.random((B, n_gt, 3), dtype=np.float32) - 0.5)

    def shell(n):
        v = rng.standard_normal((B, n, 3)).astype(np.float32)
        v /= np.linalg.norm(v, axis=-1, keepdims=True) + 1e-9
        return (v * (0.25 + 0.2 * rng.random((B, 1, 1), dtype=np.float32))).astype(np.float32)

    partial, gt = shell(n_partial), shell(n_gt)
    partial[:, n_partial - n_partial // 4:, :] = 0.0
    return partial, gt


def demo_partial(pcd_path, n=2048):
    """ASCII .pcd -> [n,3] float32: the points in file order, zero-padded (or truncated) to n."""
    pts = []
    with open(pcd_path) as fh:
        data = False
        for line in fh:
            if data:
                vals = line.split()
                if len(vals) >= 3:
                    pts.append([float(vals[0]), float(vals[1]), float(vals[2])])
            elif line.startswith("DATA"):
                assert "ascii" in line
                data = True
    a = np.asarray(pts, dtype=np.float32)[:n]
    out = np.zeros((n, 3), np.float32)
    out[:len(a)] = a
    return out


def synthetic_image(B, seed=0):
    """[B,3,224,224] float32 in [0,1): the value range torchvision's ToTensor produces."""
    return np.random.default_rng(seed).random((B, 3, 224, 224), dtype=np.float32)


# ------------------------------------------------------------------ round 3: train-step cases and the demo-scan tree
# One full train step at B = 2 of the three fused model families.  `ref_yaml` is the path inside the reference tree
# (used by tests/golden/make_golden_round3.py only), `yaml` this repository's counterpart.
TRAINSTEP_CASES = {
    "img": dict(ref_yaml="cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml",
                yaml="cfgs/ImgPCN_models/ImgResNetEncAdaPoinTrVariableLoss.yaml", tag="imgadapointr",
                takes_img=True, has_img=True, has_cls=False, frozen_segmentator=False, seed=411, epoch=7),
    "seg": dict(ref_yaml="cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml",
                yaml="cfgs/SegImgPCN_models/ImgEncSegDecAdaPoinTrVariableLoss.yaml", tag="imgsegadapointr",
                takes_img=True, has_img=True, has_cls=True, frozen_segmentator=True, seed=412, epoch=12),
    "segenc": dict(ref_yaml="cfgs/SegImgPCN_models/SegEncAdaPoinTr.yaml",
                   yaml="cfgs/SegImgPCN_models/SegEncAdaPoinTr.yaml", tag="segencadapointr",
                   takes_img=True, has_img=False, has_cls=True, frozen_segmentator=True, seed=413, epoch=3),
}


def trainstep_state_dict(sd, case):
    """Seeded weights of a train-step case.  On top of `seeded_state_dict`: the segmentator gets std 0.1 — with 0.02 its
    activations are so small that its train-mode BatchNorms (batch statistics over B = 2) divide by sqrt(eps) and
    amplify 1e-7 input differences to 0.1 at the output, which no implementation pair can reproduce — and the last
    layer of coarse_pred gets std 0.08 so that the predicted coarse cloud is spread like a shape, not collapsed into a
    cluster (degenerate neighbourhoods for the normals, near-tied query-ranking scores)."""
    spec = TRAINSTEP_CASES[case]
    out = seeded_state_dict(sd, spec["tag"])
    wide = seeded_state_dict({k: v for k, v in sd.items() if ".segmentator." in k}, spec["tag"], std=0.1)
    out.update(wide)
    out.update(seeded_state_dict({k: v for k, v in sd.items() if k.endswith("coarse_pred.2.weight")}, spec["tag"], std=0.08))
    return out


def trainstep_inputs(case, B=2):
    """(partial [B,2048,3], gt [B,16384,3], img [B,3,224,224] | None, cls_vec [B,16] | None) of a train-step case:
    sphere-shell clouds with a zero-padded tail (SURVEY 8d's structured set), uniform images, one-hot classes 4 / 8."""
    spec = TRAINSTEP_CASES[case]
    partial, gt = synthetic_batch(B, 2048, 16384, seed=spec["seed"], structured=True)
    img = synthetic_image(B, seed=spec["seed"] + 1000) if spec["has_img"] else None
    cls = None
    if spec["has_cls"]:
        cls = np.zeros((B, 16), np.float32)
        for b in range(B):
            cls[b, (4, 8, 0, 15)[b % 4]] = 1.0
    return partial, gt, img, cls


def write_ascii_pcd(path, pts):
    """[n,3] float32 -> ASCII .pcd with the header layout of the reference's demo scans; %.9g round-trips float32."""
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    pts = np.asarray(pts, dtype=np.float32)
    with open(path, "w") as fh:
        fh.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
                 f"COUNT 1 1 1\nWIDTH {len(pts)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\nDATA ascii\n")
        for p in pts:
            fh.write("%.9g %.9g %.9g\n" % (p[0], p[1], p[2]))


def demo_tree_spec():
    return {"taxonomies": ["02691156", "02933112", "02958343", "03001627", "03636649", "04256520", "04379243",
                           "04530566"], "seed": 20263}


def build_demo_tree(root, raws):
    """A PCN-layout dataset tree under `root` from the 8 demo scans (`raws`: taxonomy -> [n,3] float32).
    Per taxonomy: train model 't0' with 8 partial renderings (rendering i drops the last 37 i points of the scan),
    val/test model 'v0' with the whole scan; complete clouds are seeded uniform 16384-point clouds, stored in the
    ImgPCN layout `<subset>/complete/<tax>/<model>/PCN/<model>.pcd` (the 4-placeholder pattern the reference's
    BasePCNDataset.py:70 formats; its own 3-placeholder cfgs/dataset_configs/PCN.yaml raises TypeError there)."""
    import json
    import os
    spec = demo_tree_spec()
    cats = []
    for ti, tax in enumerate(spec["taxonomies"]):
        raw = np.asarray(raws[tax], dtype=np.float32)
        cats.append({"taxonomy_id": tax, "taxonomy_name": tax, "train": ["t0"], "val": ["v0"], "test": ["v0"]})
        rng = np.random.default_rng(1000 + ti)
        for subset, mid, nr in (("train", "t0", 8), ("val", "v0", 1), ("test", "v0", 1)):
            gt = rng.random((16384, 3), dtype=np.float32) - 0.5
            write_ascii_pcd(os.path.join(root, subset, "complete", tax, mid, "PCN", f"{mid}.pcd"), gt)
            for i in range(nr):
                write_ascii_pcd(os.path.join(root, subset, "partial", tax, mid, f"{i:02d}.pcd"),
                                raw[:len(raw) - 37 * i])
    with open(os.path.join(root, "PCN.json"), "w") as fh:
        json.dump(cats, fh)


def build_demo_images(root, views=24):
    """24 seeded RGB renders per object of the demo tree in the ImgPCN layout
    `<subset>/complete/<tax>/<model>/ShapeNetCoreV1/projections/<view>.png`: smooth colour fields (so that resampling
    matters) of 137 x 137 pixels, every third object 160 x 137 (a non-square image exercises Resize(224)'s aspect rule)."""
    import os
    from PIL import Image
    spec = demo_tree_spec()
    for ti, tax in enumerate(spec["taxonomies"]):
        for subset, mid in (("train", "t0"), ("val", "v0"), ("test", "v0")):
            w, h = (160, 137) if ti % 3 == 2 else (137, 137)
            d = os.path.join(root, subset, "complete", tax, mid, "ShapeNetCoreV1", "projections")
            os.makedirs(d, exist_ok=True)
            yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
            for v in range(views):
                rng = np.random.default_rng(7000 + 100 * ti + v + (0 if subset == "train" else 50))
                f = rng.random((3, 3)) * 0.2
                img = np.stack([127.5 + 127.5 * np.sin(f[c, 0] * xx + f[c, 1] * yy + 6.0 * f[c, 2] + v) for c in range(3)], -1)
                Image.fromarray(img.astype(np.uint8), "RGB").save(os.path.join(d, f"{v:02d}.png"))


def metrics_cases():
    """Six (taxonomy, model, partial, gt, coarse, dense) validation samples for the Metrics / validate() fixture: ground
    truth on a sphere shell, predictions = ground truth moved by noise of the F-Score threshold's size (so precision and
    recall are neither 0 nor 1), one exact prediction, one far-off prediction."""
    cases = []
    taxa = ["02691156", "02691156", "03001627", "04379243", "04379243", "04379243"]
    for i, tax in enumerate(taxa):
        rng = np.random.default_rng(900 + i)
        v = rng.standard_normal((2048, 3))
        gt = (v / np.linalg.norm(v, axis=1, keepdims=True) * (0.3 + 0.02 * i)).astype(np.float32)
        scale = [0.004, 0.008, 0.0, 0.02, 0.006, 0.1][i]
        dense = (gt + rng.standard_normal((2048, 3)).astype(np.float32) * np.float32(scale)).astype(np.float32)
        dense = dense[rng.permutation(2048)]
        coarse = (gt[:512] + rng.standard_normal((512, 3)).astype(np.float32) * np.float32(0.01)).astype(np.float32)
        partial = gt[gt[:, 0] > 0][:1024]
        partial = np.concatenate([partial, np.zeros((2048 - len(partial), 3), np.float32)])
        cases.append(dict(taxonomy=tax, model=f"m{i}", partial=partial, gt=gt, coarse=coarse, dense=dense))
    return cases


def inference_input(raw):
    """The demo scan moved by a seeded 1e-3 jitter (float32), the input cloud of the inference fixture."""
    rng = np.random.default_rng(4242)
    return (np.asarray(raw, np.float32) + rng.standard_normal(raw.shape).astype(np.float32) * np.float32(1e-3)).astype(np.float32)
