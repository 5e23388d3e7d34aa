"""configs[4] (BASELINE.json): the decoder of disp_aspp_u_dense (nets.py:666-692) - batch norm + ELU + 1x1 stem, five
disp_bottleneck_dense levels, three disparity heads - against the oracle's op-for-op restatement, alone and chained
into the photometric loss (decoder -> normalize_disp -> multi_scale_loss -> backward through the decoder)."""
import copy

import pytest
import torch

import sfm_amd
from sfm_amd import aspp as A
from sfm_amd import synth
from oracle import ref_graph as R
from helpers import TOL_GRAD, TOL_OUT, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,Cin", [(2, 9, 13, 32), (1, 16, 24, 64), (2, 7, 5, 128), (1, 33, 70, 8), (1, 5, 6, 256)])
def test_disp_head(dev, B, H, W, Cin):
    g = torch.Generator().manual_seed(5 + Cin)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, 1, generator=g) / (3 * Cin ** 0.5) * 3.0
    b = torch.randn(1, generator=g) * 0.3
    gy = torch.randn(B, H, W, 1, generator=g)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    yo = R.disp_head(xo, wo, bo)
    yo.backward(gy)
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    yg = sfm_amd.disp_head(xg, wg, bg)
    yg.backward(gy.to(dev))
    assert rel_err(yg, yo) <= TOL_OUT
    assert rel_err(xg.grad, xo.grad) <= TOL_GRAD
    assert rel_err(wg.grad, wo.grad) <= TOL_GRAD
    assert rel_err(bg.grad, bo.grad) <= TOL_GRAD
    # reproducible: fixed-order reductions, no atomics
    xg2, wg2, bg2 = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    sfm_amd.disp_head(xg2, wg2, bg2).backward(gy.to(dev))
    assert torch.equal(wg2.grad, wg.grad) and torch.equal(bg2.grad, bg.grad) and torch.equal(xg2.grad, xg.grad)


def test_disp_head_rejects_unsupported_channel_counts(dev):
    x = torch.zeros(1, 4, 4, 48, device=dev)
    with pytest.raises(ValueError, match="power of two"):
        sfm_amd.disp_head(x, torch.zeros(3, 3, 48, 1, device=dev), torch.zeros(1, device=dev))
    with pytest.raises(ValueError, match="w must be"):
        sfm_amd.disp_head(torch.zeros(1, 4, 4, 32, device=dev), torch.zeros(3, 3, 16, 1, device=dev), torch.zeros(1, device=dev))


def _leaves(params):
    """every tensor of a decoder parameter dict, in a fixed order, as (path, tensor)"""
    out = []
    for k in sorted(params):
        for kk in sorted(params[k]):
            v = params[k][kk]
            if isinstance(v, dict):
                for k3 in sorted(v):
                    out.append((f"{k}.{kk}.{k3}", v[k3]))
            else:
                out.append((f"{k}.{kk}", v))
    return out


def _with_grad(params, device):
    p = copy.deepcopy(params)
    for k in p:
        for kk in p[k]:
            v = p[k][kk]
            if isinstance(v, dict):
                for k3 in v:
                    v[k3] = v[k3].detach().to(device).requires_grad_(True)
            else:
                p[k][kk] = v.detach().to(device).requires_grad_(True)
    return p


def _compare_decoder(dev, B, h, w, width_div, tol_out, tol_grad, seed=3):
    params = A.make_decoder_params("cpu", seed, width_div)
    ends = A.make_end_points(B, h, w, "cpu", seed, width_div)
    g = torch.Generator().manual_seed(seed + 9)
    po, pg = _with_grad(params, "cpu"), _with_grad(params, dev)
    eo = {k: v.clone().requires_grad_(True) for k, v in ends.items()}
    eg = {k: v.to(dev).requires_grad_(True) for k, v in ends.items()}
    do = R.disp_aspp_u_dense_decoder(eo, po)
    dg = sfm_amd.disp_aspp_u_dense_decoder(eg, pg)
    up = [torch.randn(t.shape, generator=g) for t in do]   # random upstream weights on the three maps
    sum((a * b).sum() for a, b in zip(do, up)).backward()
    sum((a * b.to(dev)).sum() for a, b in zip(dg, up)).backward()
    worst_out = max(rel_err(a, b) for a, b in zip(dg, do))
    assert worst_out <= tol_out, worst_out
    worst = {}
    # A beta in front of a linear operator and another batch norm ('stem', 'out' of the levels that feed the next one)
    # has a mathematically ZERO gradient - the next batch norm removes any constant - so what both sides hold there is
    # cancellation noise: errors are taken relative to at least 1e-3 of the largest parameter gradient.
    floor = 1e-3 * max(float(b.grad.abs().max()) for _, b in _leaves(po))
    for (name, a), (_, b) in zip(_leaves(pg), _leaves(po)):
        assert a.grad is not None, name
        worst[name] = rel_err(a.grad, b.grad, floor=floor)
    for k in eo:
        worst["end_points." + k] = rel_err(eg[k].grad, eo[k].grad)
    bad = {k: v for k, v in worst.items() if v > tol_grad}
    assert not bad, bad
    return worst_out, max(worst.values())


def test_decoder_small(dev):
    """every operator of the decoder at reduced width (channels / 8, floor 32) on a 64 x 96 image: forward and the
    gradient of every parameter and end-point"""
    wo, wg = _compare_decoder(dev, 2, 2, 3, 8, 1e-4, 1e-3)
    print(f"decoder small: worst output rel err {wo:.2e}, worst gradient rel err {wg:.2e}")


@pytest.mark.timeout(600)
def test_decoder_full_size(dev):
    """configs[4] at full size: B=8, 128 x 416 (end-points 4x13x2048 ... 64x208x64), full channel widths.  fp32 sums over
    K up to 9216 in a different order than the CPU convolution, five levels of batch norm deep: 3e-4 / 3e-3"""
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    wo, wg = _compare_decoder(dev, 8, 4, 13, 1, 3e-4, 3e-3)
    print(f"decoder 8x128x416: worst output rel err {wo:.2e}, worst gradient rel err {wg:.2e}")


def _chain(decoder, normalize, loss_fn, ends, params, inp):
    disp = decoder(ends, params)                       # [disp4, disp3, disp2] = scales 0, 1, 2
    dn, depth = normalize(disp)                        # SfMLearner_uni.py:40, 47
    return loss_fn(inp["tgt"], inp["src"], inp["K"], depth, inp["pose"], inp["trans"], pred_disp=dn, smooth_weight=0.01)


def test_decoder_chained_into_the_loss(dev):
    """one step of configs[4]: decoder forward -> disp / mean(disp), 1 / disp -> multi_scale_loss (S = 3, the decoder's
    three maps, as SfMLearner_x.py:261 runs it) -> backward through loss and decoder; every decoder gradient against
    autograd of the oracle chain"""
    B, h, w, wd = 2, 2, 3, 8
    H, W, S, N = 32 * h, 32 * w, 3, 2
    params = A.make_decoder_params("cpu", 11, wd)
    ends = A.make_end_points(B, h, w, "cpu", 11, wd)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=12, full_res=False)
    po, pg = _with_grad(params, "cpu"), _with_grad(params, dev)
    lo = _chain(R.disp_aspp_u_dense_decoder, R.normalize_disp, R.multi_scale_loss, ends, po, inp)
    lo.backward()
    gi = synth.to_device(inp, dev)
    lg = _chain(sfm_amd.disp_aspp_u_dense_decoder, sfm_amd.normalize_disp, sfm_amd.multi_scale_loss,
                {k: v.to(dev) for k, v in ends.items()}, pg, gi)
    lg.backward()
    assert rel_err(lg, lo) <= 1e-4
    bad = {}
    floor = 1e-3 * max(float(b.grad.abs().max()) for _, b in _leaves(po))   # (zero-gradient betas: see _compare_decoder)
    for (name, a), (_, b) in zip(_leaves(pg), _leaves(po)):
        e = rel_err(a.grad, b.grad, floor=floor)
        if e > 2e-3:
            bad[name] = e
    assert not bad, bad


def _to(params, device):
    return {k: ({kk: (({k3: v3.to(device) for k3, v3 in vv.items()}) if isinstance(vv, dict) else vv.to(device))
                 for kk, vv in v.items()}) for k, v in params.items()}


@pytest.mark.parametrize("B,h,w,width_div,tol", [(2, 2, 3, 8, 1e-4), (8, 4, 13, 1, 3e-4)], ids=["small", "configs[4] size"])
def test_decoder_inference_mode(dev, B, h, w, width_div, tol):
    """is_training=False (test_kitti_depth_con.py:19-65): every batch norm on its moving statistics, folded with the ELU
    into the convolution that precedes it and written as a channel slice of the level's concat buffer - against the
    oracle's op-for-op inference graph; no autograd graph is built; the moving statistics stay untouched"""
    params = A.make_decoder_params("cpu", 21, width_div, moving_stats=True)
    ends = A.make_end_points(B, h, w, "cpu", 21, width_div)
    with torch.no_grad():
        do = R.disp_aspp_u_dense_decoder(ends, params, is_training=False)
    pg = _to(params, dev)
    before = pg["concat"]["up"]["moving_mean"].clone()
    dg = sfm_amd.disp_aspp_u_dense_decoder({k: v.to(dev) for k, v in ends.items()}, pg, is_training=False)
    assert all(not t.requires_grad and t.grad_fn is None for t in dg)
    assert torch.equal(pg["concat"]["up"]["moving_mean"], before)
    worst = max(rel_err(a, b) for a, b in zip(dg, do))
    print(f"decoder inference {B}x{32 * h}x{32 * w}: worst output rel err {worst:.2e}")
    assert worst <= tol
    # and it differs from the training-mode graph on the same inputs (the statistics really are the moving ones)
    dt = sfm_amd.disp_aspp_u_dense_decoder({k: v.to(dev) for k, v in ends.items()}, pg)
    assert max(rel_err(a, b) for a, b in zip(dt, dg)) > 1e-3


def test_inference_mode_needs_moving_statistics(dev):
    params = _to(A.make_decoder_params("cpu", 22, 8), dev)
    ends = {k: v.to(dev) for k, v in A.make_end_points(1, 2, 3, "cpu", 22, 8).items()}
    with pytest.raises(ValueError, match="moving_mean"):
        sfm_amd.disp_aspp_u_dense_decoder(ends, params, is_training=False)
