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
    """config 5 under data parallelism: the ASPP weight gradients are SUMMED over the ranks in buckets (every rank's
    loss is the global one, so its parameter gradient is a partial sum of the global gradient)"""
    out_path = str(tmp_path / "wg.pt")
    mp.spawn(_grad_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    r0 = torch.load(out_path + ".0", weights_only=True)
    r1 = torch.load(out_path + ".1", weights_only=True)
    for a, b, m0, m1 in zip(r0["reduced"], r1["reduced"], r0["mine"], r1["mine"]):
        assert torch.equal(a, b)                                   # every rank ends with the same gradients
        assert torch.allclose(a, m0 + m1, rtol=1e-6, atol=1e-7)


def _param_grad_worker(rank, world, port, out_path):
    """The oracle loss with GLOBAL denominators on this rank's shard, differentiated with respect to a shared
    'network' parameter theta (depth = base * exp(theta)); the per-rank gradients are summed the way
    allreduce_weight_grads / GradBuckets do."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sfm_amd.aspp import allreduce_weight_grads
        inp = synth.make_inputs(B, H, W, S=S, N=N, seed=56, pose_scale=0.03)
        lo, hi = (0, 2) if rank == 0 else (2, 3)
        sl = lambda t: t[lo:hi].contiguous()
        theta = torch.zeros(S, dtype=torch.float32, requires_grad=True)
        depth = [sl(d) * torch.exp(theta[s]) for s, d in enumerate(inp["depth"])]
        part = R.loss_partials(sl(inp["tgt"]), sl(inp["src"]), sl(inp["K"]), depth, sl(inp["pose"]),
                               [sl(t) for t in inp["trans"]])
        gb = L.global_batch(hi - lo)
        # the loss every rank differentiates: its own sums enter with gradient, the other rank's as constants
        tot = L.allreduce_partials(part.detach().clone())
        loss = R.loss_from_partials(part + (tot - part.detach()), S, N, gb * H * W * 3)
        (g,) = torch.autograd.grad(loss, theta)
        grads = [g.clone()]
        allreduce_weight_grads(grads)()
        if rank == 0:
            torch.save({"g": grads[0], "loss": loss.detach()}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_summed_parameter_gradients_equal_the_single_process_gradient(tmp_path):
    out_path = str(tmp_path / "pg.pt")
    mp.spawn(_param_grad_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=True)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=56, pose_scale=0.03)
    theta = torch.zeros(S, dtype=torch.float32, requires_grad=True)
    depth = [d * torch.exp(theta[s]) for s, d in enumerate(inp["depth"])]
    ref = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], depth, inp["pose"], inp["trans"])
    (g,) = torch.autograd.grad(ref, theta)
    assert abs(float(got["loss"]) - float(ref)) <= 2e-6 * abs(float(ref))
    assert float((got["g"] - g).abs().max()) <= 1e-4 * float(g.abs().max())


class _OracleEngine(L.LossEngine):
    """LossEngine with its four kernel calls replaced by the oracle (no GPU in this tier): what is under test is the
    HOST sequence of step_dp - which buffer carries what through which all-reduce, and when the gather-side slots and
    the tail are read - with uneven shards."""

    def __init__(self, B_, S_, N_):
        self.dims = (B_, H, W, S_, N_)
        self.device = torch.device("cpu")
        self.n_partials = S_ * (3 * N_ + 1)
        self.calls = []

    def _parts(self, tgt, src, K, depth, pose, trans):
        S_ = self.dims[3]
        return R.loss_partials(tgt, src, K, [depth[s][..., None] for s in range(S_)], pose, [trans[s] for s in range(S_)])

    def forward_splat(self, tgt, src, K, depth, pose, trans, n_elem_global, need_grad, buf):
        S_, N_ = self.dims[3], self.dims[4]
        self.calls.append("splat")
        p = self._parts(tgt, src, K, depth, pose, trans).reshape(S_, 3 * N_ + 1)
        view = buf[:self.n_partials].view(S_, 3 * N_ + 1)
        view[:, :3 * N_] = p[:, :3 * N_]          # like sfm_loss_forward_splat: the min-reprojection slots stay untouched

    def forward_gather(self, tgt, src, K, depth, pose, trans, n_elem_global, need_grad, buf, e_sums):
        S_, N_ = self.dims[3], self.dims[4]
        self.calls.append("gather")
        e_sums.copy_(self._parts(tgt, src, K, depth, pose, trans).reshape(S_, 3 * N_ + 1)[:, 3 * N_])

    def backward(self, tgt, src, K, depth, pose, trans, partials, n_elem_global, gloss=1.0, need_grad=True):
        self.calls.append("backward")
        self.kv_inputs = partials[:self.n_partials].clone()    # what the gradients would be computed from
        return torch.zeros((), dtype=torch.float32), None, None, None

    def loss_value(self, buf, e_sums, n_elem_global, loss):
        S_, N_ = self.dims[3], self.dims[4]
        self.calls.append("value")
        p = buf[:self.n_partials].clone().view(S_, 3 * N_ + 1)
        p[:, 3 * N_] = e_sums
        loss.copy_(R.loss_from_partials(p.reshape(-1), S_, N_, n_elem_global).float())


def _step_dp_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inp = synth.make_inputs(B, H, W, S=S, N=N, seed=57, pose_scale=0.03)
        lo, hi = (0, 2) if rank == 0 else (2, 3)
        sl = lambda t: t[lo:hi].contiguous()
        depth = torch.stack([sl(d).squeeze(3) for d in inp["depth"]], 0)
        trans = torch.stack([sl(t) for t in inp["trans"]], 0)
        eng = _OracleEngine(hi - lo, S, N)
        tail = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)     # this rank's share of the regulariser
        loss, _, _, _, tail_out = eng.step_dp(sl(inp["tgt"]), sl(inp["src"]), sl(inp["K"]), depth, sl(inp["pose"]), trans,
                                              B * H * W * 3, need_grad=False, tail=tail)
        if rank == 0:
            torch.save({"loss": loss, "tail": tail_out.clone(), "calls": eng.calls, "kv": eng.kv_inputs}, out_path)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_overlapped_step_dp_host_sequence_with_uneven_shards(tmp_path):
    """LossEngine.step_dp on two gloo ranks (2 + 1 snippets), the oracle standing in for the kernels: the splat-side
    sums + tail ride in all-reduce #1, the min-reprojection sums in #2, the backward sees GLOBAL splat-side sums, and
    the loss assembled from the two reduced buffers is the single-process loss."""
    out_path = str(tmp_path / "sd.pt")
    mp.spawn(_step_dp_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=True)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=57, pose_scale=0.03)
    ref = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"])
    assert got["calls"] == ["splat", "gather", "backward", "value"]
    assert abs(float(got["loss"]) - float(ref)) <= 2e-6 * abs(float(ref))
    assert abs(float(got["tail"][0]) - 0.75) < 1e-12                       # 0.25 + 0.5 summed over the ranks
    whole = R.loss_partials(inp["tgt"], inp["src"], inp["K"], inp["depth"], inp["pose"], inp["trans"]).reshape(S, 3 * N + 1)
    kv = got["kv"].reshape(S, 3 * N + 1)
    assert torch.allclose(kv[:, :3 * N], whole[:, :3 * N], rtol=1e-9)      # global sums reached the backward
