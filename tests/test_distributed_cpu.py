"""CPU, world_size 2, gloo: the multi-view sharding + single all-reduce of train_step (the N>1
path of bench.py) with a stand-in pipeline, against the single-process result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


class FakeCtx:
    device = torch.device("cpu")


class FakePipe:
    """Duck-types SplatPipeline: deterministic per-view image and gradient on the CPU."""

    def __init__(self, N, result_size):
        self.N, self.result_size, self.ctx = N, result_size, FakeCtx()
        self.calls = []

    def forward(self, view):
        W, H = self.result_size
        img = torch.full((W, H, 3), float(view), dtype=torch.float32)
        self.calls.append(view)
        return img, {"view": view}

    def backward_rows(self, state, dL_dimage):
        state["dl_sum"] = dL_dimage.sum()

    def backward_params(self, states, grads, accumulate=False):
        if not accumulate:
            grads.flat.zero_()
        for st in states:
            grads.flat += torch.arange(grads.flat.numel(), dtype=torch.float32) * (st["view"] + 1) + st["dl_sum"]
        return grads


def _worker(rank, world, port, n_views, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gsplat_amd as gs
    pipe = FakePipe(7, [4, 3])
    grads = gs.FlatGrads(7, torch.device("cpu"))
    targets = [torch.zeros(4, 3, 3) for _ in range(n_views)]
    loss, n_local = gs.train_step(pipe, list(range(n_views)), targets, grads)
    torch.save({"flat": grads.flat.clone(), "calls": pipe.calls, "n_local": n_local, "loss": loss},
               os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_views", [8, 5])
def test_two_ranks_equal_one_rank(tmp_path, n_views):
    import gsplat_amd as gs
    # single process reference
    pipe = FakePipe(7, [4, 3])
    ref = gs.FlatGrads(7, torch.device("cpu"))
    targets = [torch.zeros(4, 3, 3) for _ in range(n_views)]
    loss_ref, n_ref = gs.train_step(pipe, list(range(n_views)), targets, ref)
    assert n_ref == n_views and pipe.calls == list(range(n_views))
    mp.spawn(_worker, args=(2, _free_port(), n_views, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    assert r0["calls"] == [v for v in range(n_views) if v % 2 == 0]
    assert r1["calls"] == [v for v in range(n_views) if v % 2 == 1]
    assert r0["n_local"] + r1["n_local"] == n_views
    # after the all-reduce both ranks hold the full-batch gradient
    assert torch.allclose(r0["flat"], ref.flat, rtol=1e-6)
    assert torch.equal(r0["flat"], r1["flat"])
    assert abs((r0["loss"] + r1["loss"]).item() - loss_ref.item()) < 1e-4


def test_flat_grads_views_alias_one_buffer():
    import gsplat_amd as gs
    g = gs.FlatGrads(5, torch.device("cpu"))
    offs, total = gs.flat_offsets(5)
    assert g.flat.numel() == total == 300 and offs == [0, 16, 32, 52, 60]     # groups start on 16-byte boundaries
    assert gs.flat_offsets(8) == ([0, 24, 48, 80, 88], 8 * 59)
    g.means.fill_(1); g.log_scales.fill_(2); g.quats.fill_(3); g.opacity.fill_(4); g.sh.fill_(5)
    assert g.flat.sum().item() == 5 * (3 * 1 + 3 * 2 + 4 * 3 + 4 + 48 * 5)
    g.zero_()
    assert g.flat.abs().sum().item() == 0
