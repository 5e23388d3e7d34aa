"""Two data-parallel ranks on the ONE GPU of the test box, through the product's own entry point: each rank calls
sfm_amd.multi_scale_loss (fused, the pyramid read by the warp kernels) on its shard of the batch with a gloo process
group (collectives staged through host memory - RCCL refuses two ranks on one device), i.e. the real host sequence of
_FusedTotalLoss: warps -> all-reduce of the mask sums beside the gather-side stencil -> splat-side stencil ->
all-reduce of the partial sums and the regulariser's share -> backward.  Uneven shards (2 + 1 snippets).  Checked
against the oracle on the whole batch: the loss on every rank, and every per-sample gradient."""
import os
import socket

import pytest
import torch

from oracle import ref_graph as R
from sfm_amd import synth
from helpers import TOL_GRAD, TOL_OUT, rel_err

pytestmark = pytest.mark.gpu
B, H, W, S, N = 3, 32, 64, 3, 2


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    import sfm_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        inp = synth.make_inputs(B, H, W, S=S, N=N, seed=93, full_res=False)
        lo, hi = (0, 2) if rank == 0 else (2, 3)
        sl = lambda t: t[lo:hi].contiguous().to(dev)  # noqa: E731
        depth = [sl(d).requires_grad_(True) for d in inp["depth"]]
        trans = [sl(t).requires_grad_(True) for t in inp["trans"]]
        pose = sl(inp["pose"]).requires_grad_(True)
        res = {}
        for fuse in (True, False):
            for t in depth + trans + [pose]:
                t.grad = None
            loss = sfm_amd.multi_scale_loss(sl(inp["tgt"]), sl(inp["src"]), sl(inp["K"]), depth, pose, trans,
                                            pred_disp=[sl(d) for d in inp["disp"]], smooth_weight=0.01,
                                            global_batch_size=B, fuse_resize=fuse)
            loss.backward()
            res[fuse] = {"loss": loss.detach().cpu(), "pose": pose.grad.cpu(), "depth": [d.grad.cpu() for d in depth],
                         "trans": [t.grad.cpu() for t in trans]}
        torch.save(res, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_reproduce_the_single_process_loss_and_gradients(dev, tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "dp2.pt")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    inp = synth.make_inputs(B, H, W, S=S, N=N, seed=93, full_res=False)
    do = [d.clone().requires_grad_(True) for d in inp["depth"]]
    to = [t.clone().requires_grad_(True) for t in inp["trans"]]
    po = inp["pose"].clone().requires_grad_(True)
    lo = R.multi_scale_loss(inp["tgt"], inp["src"], inp["K"], do, po, to, pred_disp=inp["disp"], smooth_weight=0.01)
    lo.backward()
    for rank, (a, b) in enumerate(((0, 2), (2, 3))):
        got = torch.load(out_path + f".{rank}", weights_only=True)
        for fuse in (True, False):
            g = got[fuse]
            assert rel_err(g["loss"], lo) <= TOL_OUT, (rank, fuse)
            assert rel_err(g["pose"], po.grad[a:b]) <= TOL_GRAD, (rank, fuse)
            for s_ in range(S):
                assert rel_err(g["depth"][s_], do[s_].grad[a:b]) <= TOL_GRAD, (rank, fuse, s_)
                assert rel_err(g["trans"][s_], to[s_].grad[a:b]) <= TOL_GRAD, (rank, fuse, s_)
