"""GPU parity of the ASPP dilated 3x3 convolutions (csrc/aspp.hip, fp32-input MFMA implicit GEMM)
against the oracle (oracle.ref_graph.dilated_conv3x3 = nets.py:457 restated with torch-CPU conv2d)
and torch autograd for the two gradients.  fp32 in / fp32 accumulate: tolerance 1e-5 relative
(summation order only; K up to 9216)."""
import pytest
import torch

import sfm_amd
from oracle import ref_graph as R
from helpers import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5

CASES = [  # (B, H, W, Cin, Cout, rate): every tile configuration, ragged M, rates beyond the image
    (1, 5, 7, 32, 32, 1),
    (2, 9, 13, 64, 32, 2),      # last-level shape class (64 -> 32), split weight gradient
    (1, 8, 26, 128, 64, 3),     # 128 -> 64
    (2, 8, 26, 256, 128, 4),    # 128-wide tiles, few tiles: every stream-K range cuts one
    (1, 6, 6, 96, 160, 12),     # odd multiples of 32, rate larger than the image
    (1, 16, 52, 64, 64, 6),     # 9-tap row-segment weight gradient, ragged last segment
    (2, 12, 40, 128, 32, 3),    # the same with two input-channel blocks
    (1, 9, 70, 64, 32, 16),     # its largest rate
    # round 2 (stream-K ranges, loader waves, LDS-DMA): shapes that stress the loaders' coordinate tracking and the slabs
    (1, 7, 9, 128, 128, 2),     # M = 63: one ragged tile, K-steps of the weight gradient end inside a 32-pixel step
    (3, 11, 27, 128, 64, 2),    # M = 891: ragged against 128 / 256 rows, every range cuts a tile
    (2, 40, 5, 32, 128, 1),     # W < 8: a DMA instruction's 8 rows wrap over image rows; Cin = 32 (one channel step per tap)
    (1, 3, 3, 256, 256, 3),     # 9 pixels, rate = image size: only the centre tap is ever valid
    (5, 6, 50, 160, 96, 5),     # five images in one tile range, odd channel multiples
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,rate", CASES)
def test_dilated_conv_forward_and_gradients(dev, B, H, W, Cin, Cout, rate):
    g = torch.Generator().manual_seed(100 + Cin + Cout + rate)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / (3.0 * Cin ** 0.5)
    go = torch.randn(B, H, W, Cout, generator=g)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = R.dilated_conv3x3(xo, wo, rate)
    yo.backward(go)
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yg = sfm_amd.dilated_conv3x3(xg, wg, rate)
    yg.backward(go.to(dev))
    assert rel_err(yg, yo) <= TOL
    assert rel_err(xg.grad, xo.grad) <= TOL
    assert rel_err(wg.grad, wo.grad) <= TOL


def test_identity_tap_and_shift_known_answer(dev):
    """w = identity on the centre tap returns the input; identity on tap (0, 0) returns the input shifted
    by (-rate, -rate) with zero fill - catches transposed / mirrored tile maps (asymmetric)."""
    B, H, W, C, rate = 1, 12, 20, 64, 3
    x = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(7)).to(dev)
    eye = torch.eye(C)
    w = torch.zeros(3, 3, C, C); w[1, 1] = eye
    assert torch.equal(sfm_amd.dilated_conv3x3(x, w.to(dev), rate), x)
    w = torch.zeros(3, 3, C, C); w[0, 2] = eye   # tap dy = -1, dx = +1
    y = sfm_amd.dilated_conv3x3(x, w.to(dev), rate)
    ref = torch.zeros_like(x)
    ref[:, rate:, :W - rate] = x[:, :H - rate, rate:]
    assert torch.equal(y, ref)


def test_bitwise_reproducible(dev):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 26, 256, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(3, 3, 256, 128, generator=g) / 48).to(dev).requires_grad_(True)
    outs = []
    for _ in range(2):
        x.grad = w.grad = None
        y = sfm_amd.dilated_conv3x3(x, w, 2)
        y.square().sum().backward()
        outs.append((y.detach().clone(), x.grad.clone(), w.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("h,w_,Cin,Cout,rate", [(8, 26, 1024, 512, 2), (128, 416, 64, 32, 12)],
                         ids=["concat 8x26 1024->512 r2", "concat4 128x416 64->32 r12"])
def test_full_size_layers_of_config5(dev, h, w_, Cin, Cout, rate):
    """The smallest-M and the largest-M layer of BASELINE.json configs[4] (B=8): forward and weight gradient
    (K = B*h*w up to 425 984 per tap) against the oracle."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, h, w_, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / (3.0 * Cin ** 0.5)
    torch.set_num_threads(16)
    go = torch.randn(8, h, w_, Cout, generator=g)
    wo = w.clone().requires_grad_(True)
    yo = R.dilated_conv3x3(x, wo, rate)
    yo.backward(go)
    wg = w.to(dev).requires_grad_(True)
    yg = sfm_amd.dilated_conv3x3(x.to(dev), wg, rate)
    yg.backward(go.to(dev))
    assert rel_err(yg, yo) <= TOL
    assert rel_err(wg.grad, wo.grad) <= TOL


# ---------------------------------------------------------------- batch norm + ELU
@pytest.mark.parametrize("shape", [(2, 8, 26, 512), (1, 16, 52, 256), (8, 128, 416, 32), (3, 7, 5, 64), (2, 5, 3, 1024),
                                   (1, 1, 1, 32)])
def test_batch_norm_elu_matches_oracle(dev, shape):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(*shape, generator=g) * 1.5 + 0.7      # off-centre: the variance must not cancel
    C = shape[-1]
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    gy = torch.randn(*shape, generator=g)
    mm_o, mv_o = torch.zeros(C), torch.ones(C)
    xo, go, bo = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yo = R.batch_norm_elu(xo, go, bo, mm_o, mv_o)
    yo.backward(gy)
    mm_g, mv_g = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    xg, gg, bg = (t.to(dev).requires_grad_(True) for t in (x, gamma, beta))
    yg = sfm_amd.batch_norm_elu(xg, gg, bg, mm_g, mv_g)
    yg.backward(gy.to(dev))
    M = x.numel() // C
    if M > 1:
        assert rel_err(yg, yo) <= 1e-5
        assert rel_err(xg.grad, xo.grad) <= 1e-4
        assert rel_err(gg.grad, go.grad) <= 1e-4
        assert rel_err(mv_g, mv_o) <= 1e-5
    assert rel_err(bg.grad, bo.grad) <= 1e-4
    assert rel_err(mm_g, mm_o) <= 1e-5


@pytest.mark.parametrize("C,split", [(64, (3, 5)), (32, (1, 7)), (256, (4, 4))])
def test_sync_batch_norm_two_rehearsed_ranks_equal_the_global_batch(dev, C, split):
    """synchronised batch norm (data parallel): two "ranks" with unequal, differently centred shards -
    their gathered (n, mean, M2) triples and summed backward sums reproduce the oracle's batch norm of the WHOLE batch"""
    from sfm_amd import aspp as A
    from sfm_amd._lib import call, ptr, stream_of
    from sfm_amd import _lib
    g = torch.Generator().manual_seed(15)
    B = sum(split)
    x = torch.randn(B, 9, 14, C, generator=g) * 1.5 + 0.7
    x[split[0]:] += 2.0                                     # the second shard's own mean is far from the first's
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    gy = torch.randn(B, 9, 14, C, generator=g)
    mm_o, mv_o = torch.zeros(C), torch.ones(C)
    xo, go, bo = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    R.batch_norm_elu(xo, go, bo, mm_o, mv_o).backward(gy)

    shards = [x[:split[0]].contiguous().to(dev), x[split[0]:].contiguous().to(dev)]
    gys = [gy[:split[0]].contiguous().to(dev), gy[split[0]:].contiguous().to(dev)]
    stats_all = torch.empty(2, 3, C, dtype=torch.float64, device=dev)
    for r, xs in enumerate(shards):
        M = xs.numel() // C
        ws = torch.empty(_lib.load().sfm_bn_elu_workspace_floats(M, C) // 2 + 1, dtype=torch.float64, device=dev)
        call("sfm_bn_local_stats", ptr(xs), M, C, ptr(stats_all[r]), ptr(ws), stream_of(xs))
    ys, leaves, mms, saved = [], [], [], []
    for r, xs in enumerate(shards):
        xr, gr, br = xs.requires_grad_(True), gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
        mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y, mean, invstd = A._SyncBatchNormElu.apply(xr, gr, br, mm, mv, A.BN_DECAY, A.BN_EPSILON, True, None, stats_all, None)
        ys.append(y); leaves.append((xr, gr, br)); mms.append((mm, mv)); saved.append((mean, invstd))
    assert rel_err(torch.cat(ys), R.batch_norm_elu(x, gamma, beta)) <= 1e-5
    for mm, mv in mms:                                      # every rank holds the global moving statistics
        assert rel_err(mm, mm_o) <= 1e-5 and rel_err(mv, mv_o) <= 1e-5
    assert torch.equal(saved[0][0], saved[1][0]) and torch.equal(saved[0][1], saved[1][1])
    # the all-reduce of the backward sums, rehearsed: every shard's local sums first, then their total for each backward
    local = []
    for r, xs in enumerate(shards):
        M = xs.numel() // C
        ws = torch.empty(_lib.load().sfm_bn_elu_workspace_floats(M, C) // 2 + 1, dtype=torch.float64, device=dev)
        gg, gb = torch.empty(C, device=dev), torch.empty(C, device=dev)
        call("sfm_bn_elu_bwd_sums", ptr(xs), ptr(ys[r].detach()), ptr(gys[r]), ptr(saved[r][0]), ptr(saved[r][1]), M, C, 1, ptr(gg),
             ptr(gb), ptr(ws), stream_of(xs))
        local.append(torch.stack([gg, gb]))
    total = local[0] + local[1]
    A._SyncBatchNormElu.rehearsal_sums = staticmethod(lambda both: total.clone())
    try:
        for r in range(2):
            ys[r].backward(gys[r])
    finally:
        A._SyncBatchNormElu.rehearsal_sums = None
    assert rel_err(torch.cat([leaves[0][0].grad, leaves[1][0].grad]), xo.grad) <= 1e-4
    assert rel_err(leaves[0][1].grad + leaves[1][1].grad, go.grad) <= 1e-4     # parameter gradients: summed over ranks
    assert rel_err(leaves[0][2].grad + leaves[1][2].grad, bo.grad) <= 1e-4


def test_conv2d_bn_elu_is_one_slim_conv2d(dev):
    """slim.conv2d(x, 64, [3, 3], rate=3) under the ASPP arg-scope = dilated conv -> batch norm -> ELU"""
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 12, 20, 128, generator=g)
    w = torch.randn(3, 3, 128, 64, generator=g) / (3 * 128 ** 0.5)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = R.batch_norm_elu(R.dilated_conv3x3(xo, wo, 3), gamma, beta)
    yo.square().sum().backward()
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yg = sfm_amd.conv2d_bn_elu(xg, wg, gamma.to(dev), beta.to(dev), 3)
    yg.square().sum().backward()
    assert rel_err(yg, yo) <= 1e-5
    assert rel_err(xg.grad, xo.grad) <= 1e-4
    assert rel_err(wg.grad, wo.grad) <= 1e-4


def test_batch_norm_elu_rejects_bad_shapes(dev):
    with pytest.raises(ValueError):
        sfm_amd.batch_norm_elu(torch.zeros(1, 2, 2, 6, device=dev), torch.ones(6, device=dev), torch.zeros(6, device=dev))
    with pytest.raises(TypeError):
        sfm_amd.batch_norm_elu(torch.zeros(1, 2, 2, 32), torch.ones(32), torch.zeros(32))


# ---------------------------------------------------------------- the rest of disp_bottleneck_dense
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 26, 64, 32), (1, 5, 7, 160, 96), (2, 16, 20, 256, 128), (1, 9, 9, 32, 512)])
def test_conv1x1(dev, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(1, 1, Cin, Cout, generator=g) / Cin ** 0.5
    gy = torch.randn(B, H, W, Cout, generator=g)
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yo = R.conv1x1(xo, wo)
    yo.backward(gy)
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yg = sfm_amd.conv1x1(xg, wg)
    yg.backward(gy.to(dev))
    assert rel_err(yg, yo) <= TOL and rel_err(xg.grad, xo.grad) <= TOL and rel_err(wg.grad, wo.grad) <= TOL


@pytest.mark.parametrize("shape,out", [((2, 4, 13, 64), (8, 26)), ((1, 8, 26, 1024), (16, 52)), ((3, 5, 7, 32), (10, 14))])
def test_resize_feature_maps(dev, shape, out):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(*shape, generator=g)
    gy = torch.randn(shape[0], out[0], out[1], shape[3], generator=g)
    xo = x.clone().requires_grad_(True)
    yo = R.resize_bilinear_tf1(xo, *out)
    yo.backward(gy)
    xg = x.to(dev).requires_grad_(True)
    yg = sfm_amd.resize_bilinear(xg, *out)
    yg.backward(gy.to(dev))
    assert torch.equal(yg.detach().cpu(), yo.detach())
    assert rel_err(xg.grad, xo.grad) <= 1e-5


def _block_params(g, cin, cout, skip_c, use_skip, use_skip2, skip2_c):
    def conv(k, ci, co):
        return dict(w=torch.randn(k, k, ci, co, generator=g) / (k * ci ** 0.5), gamma=torch.rand(co, generator=g) + 0.5,
                    beta=torch.randn(co, generator=g) * 0.1)
    p = dict(up=conv(1, cin, cin), sup=conv(1, cin, cout), aspp_up=conv(3, cin, cout), aspp_up2=conv(3, 2 * cout, cout),
             out=conv(1, (5 if use_skip2 else 4) * cout, cout))
    if use_skip:
        p["skip_bn"] = dict(gamma=torch.rand(skip_c, generator=g) + 0.5, beta=torch.randn(skip_c, generator=g) * 0.1)
        p["skip"] = conv(1, skip_c, cout)
    else:
        p["skip"] = conv(3, 2 * cout, cout)
    if use_skip2:
        p["skip2"] = conv(1, skip2_c, cout)
    return p


@pytest.mark.parametrize("cin,cout,rates,use_skip,use_skip2", [
    (128, 64, (2, 4), True, False),    # "concat" / "concat1" form (nets.py:675-676)
    (64, 32, (3, 6), True, True),      # "concat2" / "concat3" form, with the second skip (:677, 681)
    (64, 32, (3, 6), False, True),     # "concat4": no skip block, the rate-12 convolution instead (:686, 463)
])
def test_disp_bottleneck_dense_block(dev, cin, cout, rates, use_skip, use_skip2):
    g = torch.Generator().manual_seed(9)
    B, h, w = 2, 6, 10
    concat = torch.randn(B, h, w, cin, generator=g)
    skip_feat = torch.randn(B, 2 * h, 2 * w, 96, generator=g) if use_skip else None
    skip2 = torch.randn(B, h, w, cin, generator=g) if use_skip2 else None
    p = _block_params(g, cin, cout, 96, use_skip, use_skip2, cin)
    # random upstream weights: sum(y^2) of a batch-normalised y is constant up to eps, its gradient pure cancellation
    wy, wu = torch.randn(B, 2 * h, 2 * w, cout, generator=g), torch.randn(B, 2 * h, 2 * w, cin, generator=g)
    leaves_o = [concat.clone().requires_grad_(True)] + [q["w"].clone().requires_grad_(True) for q in p.values() if "w" in q]
    po = {k: dict(v) for k, v in p.items()}
    for q, wl in zip([q for q in po.values() if "w" in q], leaves_o[1:]):
        q["w"] = wl
    yo, uo = R.disp_bottleneck_dense(po, leaves_o[0], skip_feat, rates, use_skip, use_skip2, skip2)
    ((yo * wy).sum() + (uo * wu).sum()).backward()

    pg = {k: {kk: vv.to(dev) for kk, vv in v.items()} for k, v in p.items()}
    leaves_g = [concat.to(dev).requires_grad_(True)]
    for q in [q for q in pg.values() if "w" in q]:
        q["w"] = q["w"].requires_grad_(True)
        leaves_g.append(q["w"])
    yg, ug = sfm_amd.disp_bottleneck_dense(pg, leaves_g[0], None if skip_feat is None else skip_feat.to(dev), rates,
                                           use_skip, use_skip2, None if skip2 is None else skip2.to(dev))
    ((yg * wy.to(dev)).sum() + (ug * wu.to(dev)).sum()).backward()
    assert rel_err(yg, yo) <= 2e-5 and rel_err(ug, uo) <= 2e-5
    for a, b in zip(leaves_g, leaves_o):
        assert rel_err(a.grad, b.grad) <= 2e-4


@pytest.mark.parametrize("cin,cout,rates,use_skip,use_skip2", [
    (128, 64, (2, 4), True, False), (64, 32, (3, 6), True, True), (64, 32, (3, 6), False, True)])
def test_concat_free_level_equals_the_composed_operators(dev, cin, cout, rates, use_skip, use_skip2):
    """the level as one autograd node (batch norms writing channel slices of one buffer, convolutions reading slices,
    gradients of an activation's several readers summed while loading) against the same operators composed through
    autograd with torch.cat: outputs, and the gradient of EVERY input and parameter (gamma / beta included), the upnet
    output used downstream in one run and unused in the other"""
    g = torch.Generator().manual_seed(21)
    B, h, w = 2, 5, 9
    base = dict(concat=torch.randn(B, h, w, cin, generator=g),
                skip_feat=torch.randn(B, 2 * h, 2 * w, 96, generator=g) if use_skip else None,
                skip2=torch.randn(B, h - 1, w - 2, cin, generator=g) if use_skip2 else None)   # (any size that upsamples)
    p = _block_params(g, cin, cout, 96, use_skip, use_skip2, cin)
    wy, wu = torch.randn(B, 2 * h, 2 * w, cout, generator=g).to(dev), torch.randn(B, 2 * h, 2 * w, cin, generator=g).to(dev)
    for use_upnet in (True, False):
        runs = []
        for fused in (True, False):
            inp = {k: (None if v is None else v.to(dev).requires_grad_(True)) for k, v in base.items()}
            pg = {k: {kk: vv.to(dev).requires_grad_(True) for kk, vv in v.items()} for k, v in p.items()}
            for q in pg.values():   # moving statistics, updated in place by both forms
                c = q["gamma"].shape[0]
                q["moving_mean"], q["moving_var"] = torch.zeros(c, device=dev), torch.ones(c, device=dev)
            y, u = sfm_amd.disp_bottleneck_dense(pg, inp["concat"], inp["skip_feat"], rates, use_skip, use_skip2, inp["skip2"],
                                                 fused_level=fused)
            ((y * wy).sum() + ((u * wu).sum() if use_upnet else 0.0)).backward()
            leaves = {f"in.{k}": v for k, v in inp.items() if v is not None}
            leaves.update({f"{k}.{kk}": vv for k, v in pg.items() for kk, vv in v.items()})
            runs.append((y.detach(), u.detach(), leaves))
        (yf, uf, lf), (yc, uc, lc) = runs
        assert torch.equal(yf, yc) and torch.equal(uf, uc)          # the same kernels in the same order
        for name in lf:
            if name.endswith("moving_mean") or name.endswith("moving_var"):
                assert torch.equal(lf[name], lc[name]), name
                continue
            assert lf[name].grad is not None, name
            assert rel_err(lf[name].grad, lc[name].grad) <= 2e-6, (name, use_upnet)


@pytest.mark.parametrize("B,H,W,Cin,Cout,rate,act", [
    (2, 8, 26, 256, 128, 2, True),     # small M: tiles cut by several stream-K ranges, epilogue in the fix-up kernel
    (2, 32, 52, 64, 32, 3, True),      # 256x32 tiles, epilogue in the MFMA kernel's store loop
    (1, 16, 40, 128, 64, 6, False),    # batch norm alone (activation_fn=None)
])
def test_conv_bn_elu_inference_epilogue(dev, B, H, W, Cin, Cout, rate, act):
    """slim.conv2d under is_training=False: batch norm on the moving statistics + ELU fused into the convolution's store"""
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(3, 3, Cin, Cout, generator=g) / (3 * Cin ** 0.5)
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.3
    mm, mv = torch.randn(Cout, generator=g) * 0.2, torch.rand(Cout, generator=g) + 0.3
    ref = R.batch_norm_infer(R.dilated_conv3x3(x, w, rate), gamma, beta, mm, mv, 1e-5, act)
    out = sfm_amd.conv2d_bn_elu(x.to(dev), w.to(dev), gamma.to(dev), beta.to(dev), rate, mm.to(dev), mv.to(dev),
                                is_training=False, activation=act)
    assert rel_err(out, ref) <= TOL
    # the plain convolution next to it is untouched by the epilogue plumbing
    assert rel_err(sfm_amd.dilated_conv3x3(x.to(dev), w.to(dev), rate), R.dilated_conv3x3(x, w, rate)) <= TOL


def test_grad_buckets_receive_the_weight_gradients_in_place(dev):
    """sfm_amd.GradBuckets: the weight-gradient kernels write straight into the flat all-reduce buffers (no cat before /
    no copy after the collective); the values equal the ordinary gradients bit for bit."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 12, 20, 32, generator=g).to(dev)
    w3 = (torch.randn(3, 3, 32, 64, generator=g) * 0.05).to(dev).requires_grad_(True)
    w1 = (torch.randn(1, 1, 64, 32, generator=g) * 0.1).to(dev).requires_grad_(True)
    gamma = torch.ones(32, device=dev).requires_grad_(True)
    beta = torch.zeros(32, device=dev).requires_grad_(True)

    def loss():
        y = sfm_amd.dilated_conv3x3(x, w3, 2)
        y = sfm_amd.batch_norm_elu(sfm_amd.conv1x1(y, w1), gamma, beta)
        return (y * y).sum()

    loss().backward()
    plain = [t.grad.clone() for t in (w3, w1, gamma, beta)]
    for t in (w3, w1, gamma, beta):
        t.grad = None
    gb = sfm_amd.GradBuckets([beta, gamma, w1, w3], bucket_bytes=64 << 10)   # backward order; several buckets
    try:
        loss().backward()
        for t, p in zip((w3, w1, gamma, beta), plain):
            assert torch.equal(gb.grad(t), p)
            assert torch.equal(t.grad, p)
        assert w3.grad.data_ptr() == gb.grad(w3).data_ptr()                    # autograd kept the view: no copy
        gb.allreduce()()                                                       # one rank: a no-op
        assert torch.equal(gb.grad(w3), plain[0])
    finally:
        gb.release()
    for t in (w3, w1, gamma, beta):
        t.grad = None
    loss().backward()
    assert w3.grad.data_ptr() != gb.grad(w3).data_ptr()                        # released: ordinary tensors again


@pytest.mark.parametrize("B,H,W,Cin,Cout,rate", [(2, 8, 26, 256, 128, 2), (1, 16, 52, 128, 64, 3), (1, 12, 40, 64, 32, 3)])
def test_convolutions_are_bitwise_reproducible(dev, B, H, W, Cin, Cout, rate):
    """stream-K slabs are added in workgroup order and the weight-gradient fix-up in a fixed group order: no atomics, so two
    runs of the same call give the same bits (forward, backward data, backward weight, 3x3 and 1x1)"""
    g = torch.Generator().manual_seed(401)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w3 = (torch.randn(3, 3, Cin, Cout, generator=g) / (3.0 * Cin ** 0.5)).to(dev)
    w1 = (torch.randn(1, 1, Cin, Cout, generator=g) / Cin ** 0.5).to(dev)
    go = torch.randn(B, H, W, Cout, generator=g).to(dev)

    def run():
        xs, a, b = x.clone().requires_grad_(True), w3.clone().requires_grad_(True), w1.clone().requires_grad_(True)
        y3, y1 = sfm_amd.dilated_conv3x3(xs, a, rate), sfm_amd.conv1x1(xs, b)
        (y3 * go).sum().backward(retain_graph=True)
        g3 = (xs.grad.clone(), a.grad.clone())
        xs.grad = None
        (y1 * go).sum().backward()
        return y3.detach(), y1.detach(), g3[0], g3[1], xs.grad, b.grad

    first, second = run(), run()
    for u, v in zip(first, second):
        assert torch.equal(u, v)
