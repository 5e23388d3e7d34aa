"""CPU tests of the snippet loader: the oracle's restatement of data_loader.py against
known answers, and the host half of sfm_amd.DataLoader (file lists, intrinsics files, JPEG decode into batch
buffers) on a synthetic dataset written in the reference's on-disk format.  PARITY UNPINNED by the reference
(no TensorFlow, no fixtures): the TF-1 conversions are ASSUMPTIONS listed in oracle/ref_loader.py."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_loader as R
from sfm_amd.data_loader import AugmentDraws, DataLoader
from helpers import write_snippet_dataset


def _frames(H, W, n_f):
    """side-by-side snippet whose frame f is the constant 10 * (f + 1) + channel"""
    seq = np.zeros((H, n_f * W, 3), np.uint8)
    for f in range(n_f):
        for c in range(3):
            seq[:, f * W:(f + 1) * W, c] = 10 * (f + 1) + c
    return seq


@pytest.mark.parametrize("n_src", [2, 4])
def test_unpack_order_known_answer(n_src):
    """target = centre frame; sources = frames before it, then after it, stacked along the channels (:296-324)"""
    H, W = 4, 6
    tgt, src = R.unpack_image_sequence(_frames(H, W, n_src + 1), H, W, n_src)
    assert (tgt[..., 0] == 10 * (n_src // 2 + 1)).all()
    others = [f for f in range(n_src + 1) if f != n_src // 2]
    for i, f in enumerate(others):
        for c in range(3):
            assert (src[..., 3 * i + c] == 10 * (f + 1) + c).all()
    bt, bs = R.batch_unpack_image_sequence(np.stack([_frames(H, W, n_src + 1)] * 2), H, W, n_src)
    assert (bt[1] == tgt).all() and (bs[1] == src).all()


def test_conversions_known_answers():
    u = np.arange(256, dtype=np.uint8)
    assert (R.to_uint8_saturate(R.to_float(u)) == u).all()          # the uint8 round trip is the identity
    assert R.to_uint8_saturate(np.float32([0.5]))[0] == 127            # 127.75 truncates
    assert (R.to_uint8_saturate(np.float32([-0.2, 1.7])) == [0, 255]).all()  # saturate
    assert R.preprocess_image(np.uint8([0, 255])).tolist() == [-1.0, 1.0]
    assert np.float32(1.0 / 255) == np.float32(1) / np.float32(255)    # the kernel's constant


def test_augmentation_known_answers():
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (2, 5, 7, 9), dtype=np.uint8)
    assert (R.data_augmentation(im, 1.0, (1, 1, 1), False) == im).all()
    assert (R.data_augmentation(im, 1.0, (1, 1, 1), True) == im[:, :, ::-1]).all()   # mirror along the width
    dark = R.data_augmentation(im, 0.5, (1, 1, 1), False)
    assert (dark == np.floor(im.astype(np.float64) / 255 * 0.5 * 255.5 + 1e-4)).all()
    col = R.data_augmentation(im, 1.0, (2.0, 1.0, 0.0), False)
    assert (col[..., 2::3] == 0).all() and (col[..., 1::3] == im[..., 1::3]).all()
    assert (col[..., 0::3] >= im[..., 0::3]).all()
    assert (R.adjust_contrast(im, 1.0) == im).all()                                  # factor 1 is the identity
    flat = R.adjust_contrast(im, 0.0).astype(np.float64)                             # factor 0 -> the channel mean
    assert np.abs(flat - flat.mean(axis=(1, 2), keepdims=True)).max() == 0
    assert np.abs(flat[0, 0, 0] - im[0].reshape(-1, 9).mean(0)).max() <= 1.0


def test_multi_scale_intrinsics():
    k = np.float32([[[241.7, 0, 204.2], [0, 246.3, 59.0], [0, 0, 1]]])
    ms = R.get_multi_scale_intrinsics(k, 4)
    assert ms.shape == (1, 4, 3, 3)
    for s in range(4):
        assert ms[0, s, 0, 0] == np.float32(241.7) / np.float32(2 ** s) and ms[0, s, 1, 2] == np.float32(59.0) / np.float32(2 ** s)
        assert ms[0, s, 2, 2] == 1 and ms[0, s, 0, 1] == 0 and ms[0, s, 2, 0] == 0


def test_fetch_batch_val_is_plain_unpack():
    rng = np.random.default_rng(1)
    seq = rng.integers(0, 256, (3, 6, 3 * 8, 3), dtype=np.uint8)
    k = np.tile(np.eye(3, dtype=np.float32), (3, 1, 1))
    tgt, src, ms = R.fetch_batch(seq, k, 6, 8, 2, 2, augment=False)
    assert np.array_equal(tgt, R.preprocess_image(seq[:, :, 8:16]))
    assert np.array_equal(src[..., :3], R.preprocess_image(seq[:, :, 0:8]))
    assert np.array_equal(src[..., 3:], R.preprocess_image(seq[:, :, 16:24]))
    t2, s2, _ = R.fetch_batch(seq, k, 6, 8, 2, 2, augment=True)           # neutral draws change nothing
    assert np.array_equal(t2, tgt) and np.array_equal(s2, src)


# ---------------------------------------------------------------- host half of the mirror
def test_file_list_and_cam_files(tmp_path):
    root = str(tmp_path)
    seqs, cams = write_snippet_dataset(root, n=5, H=8, W=12, n_src=2, seed=3)
    dl = DataLoader(root, 2, 8, 12, 2, 4)
    mine, ref = dl.format_file_list(root, "train"), R.format_file_list(root, "train")
    assert mine == ref and len(mine["image_file_list"]) == 5
    assert all(os.path.exists(p) for p in mine["image_file_list"] + mine["cam_file_list"])
    for p, k in zip(mine["cam_file_list"], cams):
        got = dl.read_cam_file(p)
        assert np.array_equal(got, R.decode_cam_line(open(p).read()))
        assert np.allclose(got, k, atol=1e-6 * 1000) and got[2, 2] == 1 and got[0, 1] == 0   # '%f' keeps 6 decimals


def test_host_batches_decode_and_order(tmp_path):
    from PIL import Image
    root = str(tmp_path)
    write_snippet_dataset(root, n=5, H=8, W=12, n_src=2, seed=4)
    dl = DataLoader(root, 2, 8, 12, 2, 4, device="cpu", seed=7, num_workers=2)
    files = dl.format_file_list(root, "train")
    decoded = {p: np.asarray(Image.open(p).convert("RGB")) for p in files["image_file_list"]}
    got = [(img.clone().numpy(), cam.clone().numpy()) for img, cam, _ in dl._host_batches("train", True, 2)]
    assert len(got) == 4                                                # 2 epochs x (5 // 2) batches
    seen = []
    for img, cam in got:
        assert img.shape == (2, 8, 36, 3) and img.dtype == np.uint8 and cam.shape == (2, 3, 3)
        for j in range(2):
            hit = [p for p, a in decoded.items() if np.array_equal(a, img[j])]
            assert len(hit) == 1
            assert np.array_equal(cam[j], dl.read_cam_file(hit[0][:-4] + "_cam.txt"))   # image and intrinsics stay paired
            seen.append(hit[0])
    assert len(set(seen[:4])) == 4 and len(set(seen[4:])) == 4          # no repeats inside an epoch
    again = [img.clone().numpy() for img, _, _ in DataLoader(root, 2, 8, 12, 2, 4, device="cpu", seed=7)._host_batches("train", True, 2)]
    assert all(np.array_equal(a, b[0]) for a, b in zip(again, got))     # same seed, same order


def test_wrong_size_and_cpu_rejected(tmp_path):
    root = str(tmp_path)
    write_snippet_dataset(root, n=2, H=8, W=12, n_src=2, seed=5)
    with pytest.raises(ValueError):
        list(DataLoader(root, 2, 8, 16, 2, 4, device="cpu")._host_batches("train", False, 1))
    with pytest.raises(ValueError):
        list(DataLoader(root, 4, 8, 12, 2, 4, device="cpu")._host_batches("train", False, 1))
    with pytest.raises(TypeError):
        DataLoader(root, 2, 8, 12, 2, 4, device="cpu").fetch_val_batch(torch.zeros(2, 8, 36, 3, dtype=torch.uint8), torch.eye(3).expand(2, 3, 3))
    with pytest.raises(TypeError):
        next(DataLoader(root, 2, 8, 12, 2, 4, device="cpu").load_train_batch())


def test_draws_ranges():
    import random
    rng = random.Random(0)
    for _ in range(200):
        b, c, s, k = AugmentDraws.draw(rng)
        assert 1 / 1.2 <= b <= 1 / 0.8 and all(0.8 <= x <= 1.2 for x in c) and 0.5 <= k <= 1.5 and isinstance(s, bool)


def test_golden_loader():
    """the committed vectors of tests/golden/make_golden.py::loader_fixture still come out of the oracle"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_snippets.npz"))
    for tag in ("n2", "n4"):
        B, H, W, N = (int(v) for v in g[f"{tag}_dims"])
        d = [float(v) for v in g[f"{tag}_draws"]]
        vt, vs, km = R.fetch_batch(g[f"{tag}_seq"], g[f"{tag}_K"], H, W, N, 4, augment=False)
        at, as_, _ = R.fetch_batch(g[f"{tag}_seq"], g[f"{tag}_K"], H, W, N, 4, True, d[0], d[1:4], bool(d[4]), d[5])
        for got, name in ((vt, "val_tgt"), (vs, "val_src"), (km, "K_ms"), (at, "aug_tgt"), (as_, "aug_src")):
            assert np.array_equal(got, g[f"{tag}_{name}"]), f"{tag}_{name}"
