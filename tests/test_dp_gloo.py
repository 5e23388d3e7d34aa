"""Data-parallel protocol of the path on CPU (gloo, world_size 2): each rank holds a slice of
the batch of snippets, contributes S*(3N+1) fp64 partial sums, the host code of the product
(sfm_amd.loss.allreduce_partials / global_batch) sums them, and the loss finalised from the
global sums equals the single-device loss (tf_ssim's global mean is
taken before its clip, so the sums - not per-rank losses - are what is reduced).

The per-rank kernel work is stood in for by the oracle here (no GPU in this tier); on the GPU
the same two host functions sit between sfm_loss_forward and sfm_loss_backward."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_graph as R
from sfm_amd import loss as L
from sfm_amd import synth

B, H, W, S, N = 3, 12, 16, 2, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inp = synth.make_inputs(B, H, W, S=S, N=N, seed=55, pose_scale=0.03)
        lo, hi = (0, 2) if rank == 0 else (2, 3)   # uneven shards: 2 + 1 snippets
        sl = lambda t: t[lo:hi].contiguous()
        part = R.loss_partials(sl(inp["tgt"]), sl(inp["src"]), sl(inp["K"]), [sl(d) for d in inp["depth"]],
                               sl(inp["pose"]), [sl(t) for t in inp["trans"]])
        gb = L.global_batch(hi - lo)
        part = L.allreduce_partials(part)
        loss = R.loss_from_partials(part, S, N, gb * H * W * 3)
        if rank == 0:
            torch.save({"loss": loss, "gb": gb, "partials": part}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_partials_reproduce_the_global_loss(tmp_path):
    out_path = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=True)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=55, pose_scale=0.03)
    ref = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    assert got["gb"] == B
    assert abs(float(got["loss"]) - float(ref)) <= 2e-6 * abs(float(ref))
    whole = R.loss_partials(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    assert torch.allclose(got["partials"], whole, rtol=1e-9)


def test_per_rank_losses_do_not_average_to_the_global_loss():
    """why the sums are reduced before the clip: averaging per-shard losses is NOT the reference's loss
    once the masked-SSIM term is non-linear in the batch (documents the design choice)."""
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=55, pose_scale=0.03)
    ref = float(R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"]))
    acc = 0.0
    for lo, hi in ((0, 2), (2, 3)):
        sl = lambda t: t[lo:hi]
        acc += (hi - lo) / B * float(R.multi_scale_loss(sl(inp["tgt"]), sl(inp["src"]), sl(inp["K"]),
                                                        [sl(d) for d in inp["depth"]], sl(inp["pose"]),
                                                        [sl(t) for t in inp["trans"]]))
    assert abs(acc - ref) > 1e-7 * abs(ref)


def test_single_process_helpers_are_identity():
    p = torch.arange(4, dtype=torch.float64)
    assert torch.equal(L.allreduce_partials(p.clone()), p)
    assert L.global_batch(5) == 5


def _grad_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sfm_amd.aspp import allreduce_weight_grads
        g = torch.Generator().manual_seed(100 + rank)
        shapes = [(3, 3, 8, 4), (4,), (3, 3, 4, 4), (1, 1, 16, 8), (8,)]   # filters and BN vectors of different sizes
        grads = [torch.randn(s, generator=g) for s in shapes]
        mine = [t.clone() for t in grads]
        wait = allreduce_weight_grads(grads, bucket_bytes=600)              # forces several buckets
        wait()
        torch.save({"reduced": grads, "mine": mine}, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_weight_gradient_allreduce(tmp_path):
    """config 5 under data parallelism: the ASPP weight gradients are averaged over the ranks in buckets"""
    out_path = str(tmp_path / "wg.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    r0 = torch.load(out_path + ".0", weights_only=True)
    r1 = torch.load(out_path + ".1", weights_only=True)
    for a, b, m0, m1 in zip(r0["reduced"], r1["reduced"], r0["mine"], r1["mine"]):
        assert torch.equal(a, b)                                   # every rank ends with the same gradients
        assert torch.allclose(a, (m0 + m1) / 2, rtol=1e-6, atol=1e-7)
