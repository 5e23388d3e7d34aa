"""The data-parallel forward (LossEngine.forward_partials with more than one rank: three calls, the mask sums
all-reduced beside the gather-side stencil) against the single-call forward, on ONE GPU with a world-size-1 RCCL group
reported as two ranks: same kernels, so the results must be bit-identical; the collectives run through
torch.distributed's nccl backend as they do at N > 1.  Also the synchronised batch norm over the same group."""
import os

import pytest
import torch
import torch.distributed as dist

import sfm_amd
from sfm_amd import synth
from sfm_amd.loss import LossEngine
from oracle import ref_graph as R
from helpers import TOL_GRAD, TOL_OUT, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1(dev):
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield
    dist.destroy_process_group()


def _stacked(g):
    depth = torch.stack([d.squeeze(3) for d in g["depth"]], 0).contiguous()
    trans = torch.stack(g["trans"], 0).contiguous()
    return depth, trans


@pytest.mark.parametrize("B,H,W,S,N", [(2, 32, 64, 3, 2), (1, 37, 53, 2, 4)])
def test_three_call_forward_equals_single_call(dev, nccl_world1, monkeypatch, B, H, W, S, N):
    import sfm_amd.loss as L
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=91)
    g = synth.to_device(inp, dev)
    depth, trans = _stacked(g)
    eng = LossEngine(B, H, W, S, N, dev)
    ne = B * H * W * 3
    p = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
    ref = eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, p, ne, 1.0, True)
    monkeypatch.setattr(L, "_world_size", lambda pg=None: 2)
    for _ in range(2):  # twice: the workspace invariants must survive the split sequence
        q = eng.forward_partials(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, ne, True)
        dist.all_reduce(q)
        out = eng.backward(g["tgt"], g["src"], g["K"], depth, g["pose"], trans, q, ne, 1.0, True)
        assert torch.equal(q, p)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
    torch.cuda.synchronize()
    assert eng.overflow_state() == (0, 0.0)


@pytest.mark.parametrize("fuse_resize", [True, False])
def test_multi_scale_loss_on_the_data_parallel_path_matches_the_oracle(dev, nccl_world1, monkeypatch, fuse_resize):
    """with a (1-rank) group and the world size reported as 2, multi_scale_loss goes through the three-call forward
    (pyramid and materialised resize)"""
    import sfm_amd.loss as L
    monkeypatch.setattr(L, "_world_size", lambda pg=None: 2)
    inp = synth.make_inputs(2, 32, 64, S=3, N=2, seed=92, full_res=False)
    do = [d.clone().requires_grad_(True) for d in inp["depth"]]
    to = [t.clone().requires_grad_(True) for t in inp["trans"]]
    po = inp["pose"].clone().requires_grad_(True)
    lo = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], do, po, to, pred_disp=inp["disp"], smooth_weight=0.01)
    lo.backward()
    g = synth.to_device(inp, dev)
    dg = [d.requires_grad_(True) for d in g["depth"]]
    tg = [t.requires_grad_(True) for t in g["trans"]]
    pg = g["pose"].requires_grad_(True)
    with pytest.raises(ValueError, match="global_batch_size"):
        sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], dg, pg, tg, pred_disp=g["disp"], smooth_weight=0.01)
    lg = sfm_amd.multi_scale_loss(g["tgt"], g["src"], g["K"], dg, pg, tg, pred_disp=g["disp"], smooth_weight=0.01,
                                  global_batch_size=2, fuse_resize=fuse_resize)
    lg.backward()
    assert rel_err(lg, lo) <= TOL_OUT
    assert rel_err(pg.grad, po.grad) <= TOL_GRAD
    for s in range(3):
        assert rel_err(tg[s].grad, to[s].grad) <= TOL_GRAD
        assert rel_err(dg[s].grad, do[s].grad) <= TOL_GRAD


def test_sync_batch_norm_over_a_one_rank_rccl_group_equals_plain_batch_norm(dev, nccl_world1):
    """batch_norm_elu(process_group=...): the all-gather / all-reduce path itself, on a 1-rank RCCL group"""
    import torch.distributed as dist
    g = torch.Generator().manual_seed(31)
    C = 64
    x = (torch.randn(4, 10, 12, C, generator=g) * 1.3 + 0.4).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    gy = torch.randn(4, 10, 12, C, generator=g).to(dev)
    outs = []
    for pg in (None, dist.group.WORLD):
        xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y = sfm_amd.batch_norm_elu(xr, gr, br, mm, mv, process_group=pg)
        y.backward(gy)
        outs.append((y.detach(), xr.grad, gr.grad, br.grad, mm, mv))
    for a, b in zip(*outs):
        assert (a - b).abs().max() <= 1e-5 * max(1.0, float(b.abs().max()))


def test_decoder_level_with_synchronised_batch_norm_on_a_one_rank_group(dev, nccl_world1):
    """disp_bottleneck_dense(process_group=...): every batch norm of the block on the gathered statistics; with one rank
    that is the plain block up to the order of the fp64 merges"""
    import torch.distributed as dist
    g = torch.Generator().manual_seed(77)
    cin, cout, B, h, w = 64, 32, 2, 6, 10
    def conv(k, ci, co):
        return dict(w=(torch.randn(k, k, ci, co, generator=g) / (k * ci ** 0.5)).to(dev), gamma=(torch.rand(co, generator=g) + 0.5).to(dev),
                    beta=(torch.randn(co, generator=g) * 0.1).to(dev))
    p = dict(up=conv(1, cin, cin), sup=conv(1, cin, cout), aspp_up=conv(3, cin, cout), aspp_up2=conv(3, 2 * cout, cout),
             out=conv(1, 4 * cout, cout), skip=conv(3, 2 * cout, cout))
    concat = torch.randn(B, h, w, cin, generator=g).to(dev)
    wy = torch.randn(B, 2 * h, 2 * w, cout, generator=g).to(dev)
    res = []
    for pg in (None, dist.group.WORLD):
        c = concat.clone().requires_grad_(True)
        y, u = sfm_amd.disp_bottleneck_dense(p, c, None, (3, 6), False, False, None, process_group=pg)
        (y * wy).sum().backward()
        res.append((y.detach(), u.detach(), c.grad))
    for a, b in zip(*res):
        assert rel_err(a, b) <= 2e-5
