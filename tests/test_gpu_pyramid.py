"""GPU parity of the pyramid kernels (csrc/pyramid.hip) against the CPU oracle:
disp normalisation (SfMLearner_uni.py:40,47), tf.image.resize_bilinear (:91,113) and the smoothness /
|t| regulariser (:94-101, 205-219), forward and backward, through the C ABI."""
import pytest
import torch

import sfm_amd
from sfm_amd import pyramid as P
from oracle import ref_graph as R
from helpers import TOL_GRAD, TOL_OUT, rel_err

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).contiguous()


@pytest.mark.parametrize("B,h,w,C,H,W", [
    (2, 8, 12, 1, 16, 24),     # x2
    (1, 16, 52, 6, 128, 416),  # x8, the coarsest KITTI scale, two sources' translations
    (3, 5, 7, 3, 13, 17),      # ragged, non-integer ratio
    (2, 16, 24, 6, 16, 24),    # identity
    (1, 9, 11, 12, 37, 53),    # four sources
])
def test_resize_bilinear(dev, B, h, w, C, H, W):
    x = _rand((B, h, w, C), 1)
    go = _rand((B, H, W, C), 2)
    xo = x.clone().requires_grad_(True)
    yo = R.resize_bilinear_tf1(xo, H, W)
    yo.backward(go)
    xg = x.to(dev).requires_grad_(True)
    yg = sfm_amd.resize_bilinear(xg, H, W)
    yg.backward(go.to(dev))
    assert torch.equal(yg.detach().cpu(), yo.detach()), "same fp32 op order: the upsample must be bit-exact"
    assert rel_err(xg.grad, xo.grad) <= TOL_OUT


def test_pyramid_upsample_stacks_all_scales(dev):
    B, H, W, C = 2, 32, 64, 6
    maps = [_rand((B, H >> s, W >> s, C), 10 + s) for s in range(4)]
    out = P.pyramid_upsample([m.to(dev) for m in maps], H, W)
    assert out.shape == (4, B, H, W, C)
    for s in range(4):
        assert torch.equal(out[s].cpu(), R.resize_bilinear_tf1(maps[s], H, W))
    go = _rand((4, B, H, W, C), 20)
    gin = P.pyramid_upsample_grad(go.to(dev), [(m.shape[1], m.shape[2]) for m in maps])
    for s in range(4):
        xo = maps[s].clone().requires_grad_(True)
        R.resize_bilinear_tf1(xo, H, W).backward(go[s])
        assert rel_err(gin[s], xo.grad) <= TOL_OUT


def test_resize_rejects_downsampling_and_cpu(dev):
    with pytest.raises(ValueError):
        sfm_amd.resize_bilinear(torch.zeros(1, 8, 8, 1, device=dev), 4, 4)
    with pytest.raises(TypeError):
        sfm_amd.resize_bilinear(torch.zeros(1, 4, 4, 1), 8, 8)


@pytest.mark.parametrize("B,h,w,C", [(2, 16, 24, 1), (1, 3, 3, 6), (3, 13, 17, 6), (1, 128, 416, 1), (2, 5, 4, 12)])
def test_compute_smooth_loss(dev, B, h, w, C):
    x = _rand((B, h, w, C), 3)
    xo = x.clone().requires_grad_(True)
    lo = R.compute_smooth_loss(xo)
    lo.backward()
    xg = x.to(dev).requires_grad_(True)
    lg = sfm_amd.compute_smooth_loss(xg)
    (2.0 * lg).backward()
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(xg.grad, 2.0 * xo.grad) <= TOL_OUT


def test_smooth_regulariser_all_maps_one_launch(dev):
    """smooth_loss of the scale loop (SfMLearner_uni.py:94-101) over S disparity and S translation maps,
    data-parallel denominators included (global batch 2x the local one)."""
    B, H, W, S, N = 2, 32, 64, 4, 2
    disp = [_rand((B, H >> s, W >> s, 1), 30 + s).abs() + 0.1 for s in range(S)]
    exp = [_rand((B, H >> s, W >> s, 3 * N), 40 + s, 0.01) for s in range(S)]
    do = [d.clone().requires_grad_(True) for d in disp]
    eo = [e.clone().requires_grad_(True) for e in exp]
    sw, er = 0.01, 0.01
    lo = sum(sw / 2 ** s * R.compute_smooth_loss(do[s]) + er / 2 ** s * (R.compute_smooth_loss(eo[s]) + eo[s].abs().mean())
             for s in range(S))
    lo.backward()
    ws, wa = P.smooth_weights(S, sw, er)
    for gb in (B, 2 * B):
        lg, grads = P.smooth_terms([t.to(dev) for t in disp + exp], ws, wa, gb, True)
        k = B / gb
        assert rel_err(lg, k * lo) <= TOL_OUT
        for s in range(S):
            assert rel_err(grads[s], k * do[s].grad) <= TOL_OUT
            assert rel_err(grads[S + s], k * eo[s].grad) <= TOL_OUT


def test_smooth_sign_at_exact_zero(dev):
    """abs'(0) = 0 in TF and torch alike: a constant map has zero loss and zero gradient."""
    x = torch.full((1, 6, 7, 3), 0.25, device=dev).requires_grad_(True)
    l = sfm_amd.compute_smooth_loss(x)
    l.backward()
    assert float(l.detach()) == 0.0 and float(x.grad.abs().max()) == 0.0


def test_smooth_signs_with_negative_zero_samples(dev):
    """the kernel takes the sign of a difference from the float's bits: -0.0 samples (and the -0.0 differences they could
    produce next to +0.0 ones) must count as zero, as sign() does in the reference"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 9, 11, 3, generator=g)
    x[torch.rand(x.shape, generator=g) < 0.3] = 0.0
    x[torch.rand(x.shape, generator=g) < 0.3] = -0.0
    xo = x.clone().requires_grad_(True)
    lo = R.compute_smooth_loss(xo) + xo.abs().mean()
    lo.backward()
    xg = x.to(dev).requires_grad_(True)
    lg = sfm_amd.compute_smooth_loss(xg) + xg.abs().mean()
    lg.backward()
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(xg.grad, xo.grad) <= TOL_GRAD


@pytest.mark.parametrize("B,sizes", [
    (2, [(16, 24), (8, 12)]), (3, [(128, 416), (64, 208), (32, 104), (16, 52)]),   # one launch per direction
    (1, [(5, 7)]), (2, [(16, 24), (7, 9)]),          # an item that is no multiple of 4 elements: the chunked launch pair
    (1, [(192, 400), (96, 200)])])                   # an item beyond 64 Ki elements: the chunked launch pair
def test_normalize_disp(dev, B, sizes):
    raw = [10.0 * torch.sigmoid(_rand((B, h, w, 1), 50 + k)) + 0.01 for k, (h, w) in enumerate(sizes)]
    gu = [_rand(r.shape, 60 + k) for k, r in enumerate(raw)]
    gd = [_rand(r.shape, 70 + k) for k, r in enumerate(raw)]
    ro = [r.clone().requires_grad_(True) for r in raw]
    dno, dpo = R.normalize_disp(ro)
    sum((a * g).sum() for a, g in zip(dno, gu)).add(sum((a * g).sum() for a, g in zip(dpo, gd))).backward()
    rg = [r.to(dev).requires_grad_(True) for r in raw]
    dng, dpg = sfm_amd.normalize_disp(rg)
    sum((a * g.to(dev)).sum() for a, g in zip(dng, gu)).add(sum((a * g.to(dev)).sum() for a, g in zip(dpg, gd))).backward()
    for k in range(len(sizes)):
        assert rel_err(dng[k], dno[k]) <= TOL_OUT
        assert rel_err(dpg[k], dpo[k]) <= TOL_OUT
        assert rel_err(rg[k].grad, ro[k].grad) <= TOL_GRAD


@pytest.mark.parametrize("use", ["disp_norm", "depth"])
def test_normalize_disp_one_output_used(dev, use):
    """only one of the two outputs reaches the loss: the other's gradient is absent (a null source in the kernels)"""
    raw = [10.0 * torch.sigmoid(_rand((2, h, w, 1), 90 + k)) + 0.01 for k, (h, w) in enumerate([(32, 104), (16, 52)])]
    g = [_rand(r.shape, 95 + k) for k, r in enumerate(raw)]
    ro = [r.clone().requires_grad_(True) for r in raw]
    rg = [r.to(dev).requires_grad_(True) for r in raw]
    k = 0 if use == "disp_norm" else 1
    sum((a * w).sum() for a, w in zip(R.normalize_disp(ro)[k], g)).backward()
    sum((a * w.to(dev)).sum() for a, w in zip(sfm_amd.normalize_disp(rg)[k], g)).backward()
    for a, b_ in zip(rg, ro):
        assert rel_err(a.grad, b_.grad) <= TOL_GRAD


def test_total_loss_from_raw_network_outputs(dev):
    """disp -> normalise -> depth -> pyramid upsample -> warps + photometric loss + smoothness, end to end
    from the raw disparity pyramid (SfMLearner_uni.py:40-176), gradients back to the raw disparities."""
    from sfm_amd import synth
    B, H, W, S, N = 2, 32, 64, 3, 2
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=77, full_res=False)
    raw = [10.0 * torch.sigmoid(_rand(d.shape, 80 + k)) + 0.01 for k, d in enumerate(inp["disp"])]

    ro = [r.clone().requires_grad_(True) for r in raw]
    to = [t.clone().requires_grad_(True) for t in inp["trans"]]
    po = inp["pose"].clone().requires_grad_(True)
    dno, dpo = R.normalize_disp(ro)
    lo = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], dpo, po, to, pred_disp=dno, smooth_weight=0.01)
    lo.backward()

    g = synth.to_device(inp, dev)
    rg = [r.to(dev).requires_grad_(True) for r in raw]
    tg = [t.requires_grad_(True) for t in g["trans"]]
    pg = g["pose"].requires_grad_(True)
    dng, dpg = sfm_amd.normalize_disp(rg)
    lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], dpg, pg, tg, pred_disp=dng, smooth_weight=0.01)
    lg.backward()
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(pg.grad, po.grad) <= TOL_GRAD
    for s in range(S):
        assert rel_err(rg[s].grad, ro[s].grad) <= TOL_GRAD, f"raw disparity scale {s}"
        assert rel_err(tg[s].grad, to[s].grad) <= TOL_GRAD, f"translations scale {s}"


@pytest.mark.parametrize("B,H,W,S,N,sizes", [
    (2, 32, 64, 4, 2, None),                                   # H >> s: exact halvings
    (1, 37, 53, 3, 2, [(37, 53), (19, 27), (10, 14)]),          # ragged: no ratio is a power of two
    (2, 24, 40, 3, 4, [(12, 20), (24, 40), (3, 5)]),            # seq = 5; a full-size level in the middle of the list
])
def test_fused_resize_equals_materialised_resize(dev, B, H, W, S, N, sizes):
    """the warp kernels reading the pyramid at its native sizes (sfm_loss_forward_pyr /
    sfm_loss_backward_pyr: resize_bilinear evaluated while loading, its adjoint scattered by the backward kernel) against
    the same step on maps resized first (sfm_pyramid_upsample_fwd / _bwd).  Forward: one arithmetic, so the loss and
    d_pose (which never passes through the resize) agree bit for bit; the map gradients are summed in another order
    (LDS footprints + float atomics instead of two gather passes): 1e-4 like every gradient."""
    import sfm_amd
    from sfm_amd import synth
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=71, pose_scale=0.03, full_res=False)
    g = synth.to_device(inp, dev)
    if sizes is not None:
        gen = torch.Generator().manual_seed(72)
        g["depth"] = [(0.5 + torch.rand(B, h, w, 1, generator=gen)).to(dev) for h, w in sizes]
        g["trans"] = [(0.02 * torch.randn(B, h, w, 3 * N, generator=gen)).to(dev) for h, w in sizes]

    def run(fuse):
        depth = [d.detach().clone().requires_grad_(True) for d in g["depth"]]
        trans = [t.detach().clone().requires_grad_(True) for t in g["trans"]]
        pose = g["pose"].detach().clone().requires_grad_(True)
        loss = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], depth, pose, trans, fuse_resize=fuse)
        loss.backward()
        return loss.detach(), [d.grad for d in depth], pose.grad, [t.grad for t in trans]

    la, da, pa, ta = run(True)
    lb, db, pb, tb = run(False)
    assert torch.equal(la, lb)
    assert torch.equal(pa, pb)
    for s in range(S):
        assert da[s].shape == db[s].shape and ta[s].shape == tb[s].shape
        assert rel_err(da[s], db[s]) <= 1e-5, f"d_depth scale {s}"     # same values, another summation order
        assert rel_err(ta[s], tb[s]) <= 1e-5, f"d_trans scale {s}"


def test_pyramid_entry_points_reject_bad_tables(dev):
    from sfm_amd import _lib
    from sfm_amd._lib import ptr
    from sfm_amd.pyramid import _ints, _ptrs
    lib = _lib.load()
    B, H, W, S, N = 1, 16, 24, 2, 2
    t = torch.zeros(B, H, W, 3, device=dev)
    src = torch.zeros(B, H, W, 6, device=dev)
    ws = torch.zeros(lib.sfm_loss_workspace_floats(B, H, W, S, N), device=dev)
    params = torch.zeros(N * B, _lib.PARAM_STRIDE, device=dev)
    part = torch.zeros(S * (3 * N + 1), dtype=torch.float64, device=dev)
    d = [torch.ones(B, H, W, device=dev), torch.ones(B, 8, 12, device=dev)]
    tr = [torch.zeros(B, H, W, 6, device=dev), torch.zeros(B, 8, 12, 6, device=dev)]
    args = lambda hs, ws_: (ptr(t), ptr(src), _ptrs(d), _ptrs(tr), _ints(hs), _ints(ws_), ptr(params), ptr(ws), B, H, W, S, N,
                            B * H * W * 3, 0, ptr(part), None)
    assert lib.sfm_loss_forward_pyr(*args([H, 8], [W, 12])) == 0
    torch.cuda.synchronize()
    assert lib.sfm_loss_forward_pyr(*args([H, 2 * H], [W, 12])) == -1       # a level larger than the image
    assert lib.sfm_loss_forward_pyr(*args([H, 0], [W, 12])) == -1
