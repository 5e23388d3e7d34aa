"""GPU parity of the snippet unpack / augmentation kernels (csrc/snippet.hip) against
oracle/ref_loader.py: byte / integer work, so the bar is BIT-EXACT (the fp32 outputs are affine images of
uint8 values).  Calls go through the C ABI via the sfm_amd.data_loader mirror."""
import numpy as np
import pytest
import torch

import sfm_amd
from oracle import ref_loader as R
from sfm_amd.data_loader import AugmentDraws, DataLoader, snippet_unpack
from helpers import write_snippet_dataset

pytestmark = pytest.mark.gpu


def _case(B, H, W, n_src, seed):
    rng = np.random.default_rng(seed)
    low = rng.uniform(0, 255, (B, H // 3 + 2, (n_src + 1) * W // 3 + 2, 3))
    seq = np.kron(low, np.ones((1, 3, 3, 1)))[:, :H, :(n_src + 1) * W] + rng.uniform(-20, 20, (B, H, (n_src + 1) * W, 3))
    seq = np.clip(seq, 0, 255).astype(np.uint8)
    k = np.tile(np.float32([[241.7, 0, 204.2], [0, 246.3, 59.0], [0, 0, 1]]), (B, 1, 1)) + rng.uniform(0, 1, (B, 1, 1)).astype(np.float32)
    k[:, 2] = [0, 0, 1]; k[:, 0, 1] = 0; k[:, 1, 0] = 0
    return seq, k


SHAPES = [(2, 8, 12, 2), (3, 13, 17, 2), (1, 37, 53, 4), (8, 128, 416, 2), (2, 64, 129, 4)]
DRAWS = [None, AugmentDraws(1.0, (1, 1, 1), False, 1.0), AugmentDraws(1.2, (0.83, 1.1, 1.19), True, 1.43),
         AugmentDraws(0.85, (1.2, 0.8, 1.0), False, 0.52), AugmentDraws(1.25, (1.2, 1.2, 1.2), True, 1.5)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("di", range(len(DRAWS)))
def test_unpack_bit_exact(dev, shape, di):
    B, H, W, n_src = shape
    seq, k = _case(B, H, W, n_src, seed=di * 10 + B)
    d = DRAWS[di]
    tgt, src, kms = snippet_unpack(torch.from_numpy(seq).to(dev), torch.from_numpy(k).to(dev), H, W, n_src, 4, d)
    if d is None:
        rt, rs, rk = R.fetch_batch(seq, k, H, W, n_src, 4, augment=False)
    else:
        rt, rs, rk = R.fetch_batch(seq, k, H, W, n_src, 4, True, d[0], d[1], d[2], d[3])
    assert tgt.shape == (B, H, W, 3) and src.shape == (B, H, W, 3 * n_src) and kms.shape == (B, 4, 3, 3)
    assert np.array_equal(tgt.cpu().numpy(), rt)
    assert np.array_equal(src.cpu().numpy(), rs)
    assert np.array_equal(kms.cpu().numpy(), rk)


def test_full_size_properties(dev):
    """c3-sized batch (16 snippets of 1248x384 per GPU): val path = plain slices; neutral draws = val path;
    mirror twice = identity; output range."""
    B, H, W, n_src = 16, 384, 1248, 2
    g = torch.Generator(device="cpu").manual_seed(8964)
    seq = torch.randint(0, 256, (B, H, 3 * W, 3), dtype=torch.uint8, generator=g).to(dev)
    k = torch.eye(3).expand(B, 3, 3).contiguous().to(dev)
    tgt, src, _ = snippet_unpack(seq, k, H, W, n_src, 4)
    f = lambda u: u.float() * np.float32(1.0 / 255) * 2.0 - 1.0
    assert torch.equal(tgt, f(seq[:, :, W:2 * W])) and torch.equal(src[..., :3], f(seq[:, :, :W])) and torch.equal(src[..., 3:], f(seq[:, :, 2 * W:]))
    t1, s1, _ = snippet_unpack(seq, k, H, W, n_src, 4, AugmentDraws(1.0, (1, 1, 1), False, 1.0))
    assert torch.equal(t1, tgt) and torch.equal(s1, src)
    t2, s2, _ = snippet_unpack(seq, k, H, W, n_src, 4, AugmentDraws(1.0, (1, 1, 1), True, 1.0))
    assert torch.equal(t2.flip(2), tgt) and torch.equal(s2.flip(2), src)
    t3, s3, _ = snippet_unpack(seq, k, H, W, n_src, 4, AugmentDraws(1.25, (1.2, 0.8, 1.1), False, 1.5))
    assert float(t3.min()) >= -1 and float(t3.max()) <= 1 and float(s3.min()) >= -1 and float(s3.max()) <= 1


def test_rejections(dev):
    seq = torch.zeros(2, 8, 36, 3, dtype=torch.uint8, device=dev)
    k = torch.eye(3, device=dev).expand(2, 3, 3).contiguous()
    with pytest.raises(ValueError):
        snippet_unpack(seq, k, 8, 12, 3, 4)          # odd source count
    with pytest.raises(ValueError):
        snippet_unpack(seq, k, 8, 13, 2, 4)          # size mismatch
    with pytest.raises(TypeError):
        snippet_unpack(seq.float(), k, 8, 12, 2, 4)  # not uint8
    with pytest.raises(sfm_amd.SfmAmdError):
        snippet_unpack(seq, k, 8, 12, 2, 17)         # too many scales: rejected by the C ABI


def test_dataloader_end_to_end(dev, tmp_path):
    """files on disk -> device batches; each batch equals the oracle applied to the same decoded snippets + draws."""
    import random
    from PIL import Image
    root = str(tmp_path)
    H, W, n_src, B = 16, 24, 2, 2
    write_snippet_dataset(root, n=7, H=H, W=W, n_src=n_src, seed=11)
    dl = DataLoader(root, B, H, W, n_src, 4, device=dev, seed=5, num_workers=2)
    host = [(i.clone().numpy(), c.clone().numpy()) for i, c, _ in DataLoader(root, B, H, W, n_src, 4, device="cpu", seed=5)._host_batches("train", True, 2)]
    it = dl.load_train_batch(epochs=2)
    assert dl.steps_per_epoch == 3
    rng = random.Random(6)
    n = 0
    for (tgt, src, kms), (seq, cam) in zip(it, host):
        d = AugmentDraws.draw(rng)
        rt, rs, rk = R.fetch_batch(seq, cam, H, W, n_src, 4, True, d[0], d[1], d[2], d[3])
        assert np.array_equal(tgt.cpu().numpy(), rt) and np.array_equal(src.cpu().numpy(), rs) and np.array_equal(kms.cpu().numpy(), rk)
        n += 1
    assert n == 6
    val = list(dl.load_val_batch(epochs=1))
    assert len(val) == 3 and val[0][0].shape == (B, H, W, 3)
    # the mirror's per-snippet helpers
    one = torch.from_numpy(np.asarray(Image.open(dl.format_file_list(root, "train")["image_file_list"][0]).convert("RGB"))).to(dev)
    t, s = dl.unpack_image_sequence(one, H, W, n_src)
    rt, rs = R.unpack_image_sequence(one.cpu().numpy(), H, W, n_src)
    assert np.array_equal(t.cpu().numpy(), R.preprocess_image(rt)) and np.array_equal(s.cpu().numpy(), R.preprocess_image(rs))
    k = torch.from_numpy(host[0][1]).to(dev)
    assert np.array_equal(dl.get_multi_scale_intrinsics(k, 4).cpu().numpy(), R.get_multi_scale_intrinsics(host[0][1], 4))


def test_loader_feeds_the_loss(dev, tmp_path):
    """the loader's tensors go straight into the fused loss (layout contract of SURVEY 8b)"""
    root = str(tmp_path)
    H, W, B = 32, 64, 2
    write_snippet_dataset(root, n=2, H=H, W=W, n_src=2, seed=2)
    tgt, src, kms = next(DataLoader(root, B, H, W, 2, 4, device=dev, seed=1).load_train_batch(epochs=1))
    inp = sfm_amd.synth.make_inputs(B, H, W, S=4, N=2, seed=0)
    depth = [d.to(dev) for d in inp["depth"]]
    trans = [t.to(dev) for t in inp["trans"]]
    out = sfm_amd.multi_scale_loss(tgt, src, kms[:, 0].contiguous(), depth, inp["pose"].to(dev), trans)
    loss = out[0] if isinstance(out, (tuple, list)) else out
    assert torch.isfinite(loss).all()


def test_golden_loader(dev):
    """the HIP path against the committed vectors (tests/golden/loader_snippets.npz): bit-exact"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loader_snippets.npz"))
    for tag in ("n2", "n4"):
        B, H, W, N = (int(v) for v in g[f"{tag}_dims"])
        d = [float(v) for v in g[f"{tag}_draws"]]
        seq, k = torch.from_numpy(g[f"{tag}_seq"]).to(dev), torch.from_numpy(g[f"{tag}_K"]).to(dev)
        vt, vs, km = snippet_unpack(seq, k, H, W, N, 4)
        at, as_, _ = snippet_unpack(seq, k, H, W, N, 4, AugmentDraws(d[0], d[1:4], bool(d[4]), d[5]))
        for got, name in ((vt, "val_tgt"), (vs, "val_src"), (km, "K_ms"), (at, "aug_tgt"), (as_, "aug_src")):
            assert np.array_equal(got.cpu().numpy(), g[f"{tag}_{name}"]), f"{tag}_{name}"
