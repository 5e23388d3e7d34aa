"""The captured-graph form of total_loss (sfm_amd.GraphedTotalLoss) must give the loss and the pose gradient of the eager
call sequence it captured bit for bit, call after call with fresh inputs; the map gradients pass through the float
atomics of the fused resize adjoint (cells on tile borders have up to four adders), so they agree to the last bits
only: 1e-6 of the tensor's maximum."""
import pytest
import torch

import sfm_amd
from sfm_amd import synth

pytestmark = pytest.mark.gpu


def test_graphed_total_loss_matches_eager(dev):
    B, H, W, S, N = 2, 32, 64, 3, 2
    inp = synth.to_device(synth.make_inputs(B, H, W, S=S, N=N, seed=5, full_res=False), dev)
    raw0 = [(d * 3.0) for d in inp["disp"]]
    g = sfm_amd.GraphedTotalLoss(inp["tgt"], inp["src"], inp["K"], raw0, inp["pose"], inp["trans"], smooth_weight=0.01)
    for trial in range(3):
        gen = torch.Generator(device="cpu").manual_seed(100 + trial)
        raw = [(r * (1.0 + 0.1 * torch.rand(r.shape, generator=gen).to(dev))).requires_grad_(True) for r in raw0]
        trans = [(t + 0.002 * torch.randn(t.shape, generator=gen).to(dev)).requires_grad_(True) for t in inp["trans"]]
        pose = (inp["pose"] + 0.002 * torch.randn(inp["pose"].shape, generator=gen).to(dev)).requires_grad_(True)
        dn, dp = sfm_amd.normalize_disp(raw)
        loss = sfm_amd.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], dp, pose, trans, pred_disp=dn, smooth_weight=0.01)
        grads = torch.autograd.grad(loss, raw + trans + [pose])
        gl, g_raw, g_pose, g_trans = g([r.detach() for r in raw], pose.detach(), [t.detach() for t in trans])
        assert torch.equal(gl, loss.detach())
        assert torch.equal(g_pose, grads[-1])
        for a, b in zip(g_raw + g_trans, grads[:-1]):
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
