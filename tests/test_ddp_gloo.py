"""CPU, world_size 2 over gloo: the gradient-bucket all-reduce that bench.py / the trainer use over RCCL.

The collective layer is independent of the kernels, so it is exercised here with a small stock-torch model:
bucket views, arrival-counted launches during backward, averaging, unused parameters, phase-specific buckets.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 4, 1), nn.Flatten(),
                                  nn.Linear(4 * 6 * 6, 5))
        self.body[0].weight.data = self.body[0].weight.data.contiguous(memory_format=torch.channels_last)
        self.unused = nn.Linear(3, 3)  # never touched by forward: its bucket must still be reduced (zeros)

    def forward(self, x):
        return self.body(x)


def _model():
    torch.manual_seed(0)
    return _Net()


def _worker(rank, world, port, ret, use_arena):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heatnet_amd import ddp
    m = _model()
    if rank == 1:  # perturb, then check the broadcast restores rank 0's values
        for p in m.parameters():
            p.data.add_(1.0)
    ddp.broadcast_module_state(m)
    ref = _model()
    same = all(torch.equal(a, b) for a, b in zip(m.parameters(), ref.parameters()))

    if use_arena:  # gradients live in the optimizer's flat arena; buckets are slices of it
        from heatnet_amd import optim
        arena = optim.ParamArena(list(m.parameters()))
        buckets = ddp.GradBuckets(arenas=[arena], bucket_bytes=256)
        assert all(b["flat"].data_ptr() >= arena.grad.data_ptr() for b in buckets.buckets)
    else:
        buckets = ddp.GradBuckets(list(m.parameters()), bucket_bytes=256)  # several small buckets
    assert len(buckets.buckets) >= 2
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(100 + rank)
    errs = []
    for it in range(2):
        x = torch.randn(4, 3, 6, 6, generator=g)
        y = torch.randn(4, 5, generator=g)
        # expected: mean over ranks of the local gradients
        ref.load_state_dict(m.state_dict())
        ref.zero_grad()
        ((ref(x) - y) ** 2).mean().backward()
        expected = []
        for p in ref.parameters():
            gl = p.grad.clone() if p.grad is not None else torch.zeros_like(p)
            dist.all_reduce(gl)
            expected.append(gl / world)
        opt.zero_grad(set_to_none=False)
        buckets.begin_step()
        ((m(x) - y) ** 2).mean().backward()
        buckets.finish_step()
        for p, e in zip(m.parameters(), expected):
            errs.append(float((p.grad - e).abs().max()))
        assert m.body[0].weight.grad.is_contiguous(memory_format=torch.channels_last)
        opt.step()
    # parameters stay identical across ranks after the synchronised steps
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    ret[rank] = (same, max(errs), float((other[0] - other[1]).abs().max()))
    dist.destroy_process_group()


@pytest.mark.parametrize("use_arena", [False, True])
def test_grad_buckets_two_ranks(use_arena):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, use_arena), nprocs=world, join=True)
    for rank in range(world):
        same, err, drift = ret[rank]
        assert same
        assert err < 1e-6, err
        assert drift == 0.0


def test_grad_buckets_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from heatnet_amd import ddp
    m = _model()
    b = ddp.GradBuckets(list(m.parameters()))
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    opt.zero_grad(set_to_none=False)
    b.begin_step()
    m(torch.randn(2, 3, 6, 6)).sum().backward()
    b.finish_step()
    assert all(p.grad is not None for p in m.parameters())
    with pytest.raises(RuntimeError, match="left its bucket"):
        opt.zero_grad(set_to_none=True)
        b.begin_step()


# ---- phase switches (conv_segnet.setPhase flips requires_grad between two parameter sets) --------------------------------

def _phase_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heatnet_amd import ddp, optim
    m = _model()
    seg, critic = list(m.body[0].parameters()) + list(m.body[2].parameters()), list(m.body[4].parameters())
    for p in m.unused.parameters():
        p.requires_grad = False

    def set_phase(phase):
        for p in seg:
            p.requires_grad = phase == "seg"
        for p in critic:
            p.requires_grad = phase == "critic"

    set_phase("critic")
    arenas = [optim.ParamArena(seg), optim.ParamArena(critic)]
    buckets = ddp.GradBuckets(arenas=arenas, bucket_bytes=256)  # built once, while `seg` is frozen
    g = torch.Generator().manual_seed(100 + rank)
    errs, frozen_moved = [], []
    for phase in ("critic", "seg", "critic", "seg"):
        set_phase(phase)
        x = torch.randn(4, 3, 6, 6, generator=g)
        y = torch.randn(4, 5, generator=g)
        active = arenas[0] if phase == "seg" else arenas[1]
        frozen = arenas[1] if phase == "seg" else arenas[0]
        ref = _model()
        ref.load_state_dict(m.state_dict())
        ((ref(x) - y) ** 2).mean().backward()
        ref_params = (list(ref.body[0].parameters()) + list(ref.body[2].parameters())) if phase == "seg" \
            else list(ref.body[4].parameters())
        expected = []
        for p in ref_params:
            gl = p.grad.clone()
            dist.all_reduce(gl)
            expected.append(gl / world)
        before = frozen.grad.clone()
        for a in arenas:
            if a.active():
                a.zero_grad()
        buckets.begin_step()
        ((m(x) - y) ** 2).mean().backward()
        buckets.finish_step()
        for p, e in zip(active.params, expected):
            errs.append(float((active.grad_view(p) - e).abs().max()))
        frozen_moved.append(float((frozen.grad - before).abs().max()))
        with torch.no_grad():
            active.flat.sub_(0.1 * active.grad)
    flat = torch.cat([a.flat for a in arenas])
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    ret[rank] = (max(errs), max(frozen_moved), float((other[0] - other[1]).abs().max()))
    dist.destroy_process_group()


def test_grad_buckets_follow_phase_switches():
    """Buckets are built once for every arena; each step exchanges exactly the arena that is trainable in that phase
    (after setPhase the newly active arena was never all-reduced and the frozen one's stale gradients were)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_phase_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        err, moved, drift = ret[rank]
        assert err < 1e-6, err
        assert moved == 0.0      # a frozen arena's gradient buffer is neither reduced nor scaled
        assert drift == 0.0      # replicas stay identical through the switches


def test_param_buckets_refuse_a_changed_trainable_set():
    sys.path.insert(0, ROOT)
    from heatnet_amd import ddp
    m = _model()
    for p in m.unused.parameters():
        p.requires_grad = False
    b = ddp.GradBuckets(list(m.parameters()))
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    opt.zero_grad(set_to_none=False)
    b.begin_step()
    m(torch.randn(2, 3, 6, 6)).sum().backward()
    b.finish_step()
    for p in m.unused.parameters():
        p.requires_grad = True  # a phase switch the fixed bucket set does not cover
    with pytest.raises(RuntimeError, match="became trainable"):
        b.begin_step()
    for p in m.unused.parameters():
        p.requires_grad = False
    m.body[0].weight.requires_grad = False
    with pytest.raises(RuntimeError, match="was frozen"):
        b.begin_step()


# ---- what backward hides: bucket sizes, launches inside backward, exposed communication -------------------------------------

def _overlap_worker(rank, world, port, ret, use_arena):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heatnet_amd import ddp, optim
    torch.manual_seed(0)
    m = nn.Sequential(nn.Linear(16, 64), nn.ReLU(), nn.Linear(64, 256), nn.ReLU(), nn.Linear(256, 256), nn.ReLU(),
                      nn.Linear(256, 8))  # every parameter is used: every bucket fills inside backward
    kw = dict(bucket_bytes=64 << 10, tail_bytes=2 << 10)
    if use_arena:
        arena = optim.ParamArena(list(m.parameters()))
        b = ddp.GradBuckets(arenas=[arena], **kw)
    else:
        b = ddp.GradBuckets(list(m.parameters()), **kw)
    first = id(next(m.parameters()))
    tail = [bk for bk in b.buckets if first in [id(p) for p in bk["params"]]][0]
    sizes = [bk["flat"].numel() * 4 for bk in b.buckets]
    opt = torch.optim.SGD(m.parameters(), lr=0.01)
    in_bwd = []
    for _ in range(3):
        if use_arena:
            arena.zero_grad()
        else:
            opt.zero_grad(set_to_none=False)
        b.begin_step()
        m(torch.randn(4, 16)).pow(2).mean().backward()
        launched_when_backward_returned = b._step_launched
        b.finish_step()
        in_bwd.append((launched_when_backward_returned, b.launched_in_backward))
    ret[rank] = (sizes, tail["flat"].numel() * 4, in_bwd, b.exposed_ms(), b.world)
    dist.destroy_process_group()


@pytest.mark.parametrize("use_arena", [False, True])
def test_all_buckets_but_none_wait_for_finish_step_and_the_last_arriving_is_small(use_arena):
    """The bucket that holds the first-registered parameters (reached last by backward) is cut at `tail_bytes`; with every
    parameter in use, every bucket's all-reduce is issued from the arrival hooks BEFORE backward returns (finish_step() has
    nothing left to launch); exposed_ms() reports the wait; the group size is what the process group says."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), ret, use_arena), nprocs=world, join=True)
    for rank in range(world):
        sizes, tail, in_bwd, exposed, seen = ret[rank]
        assert len(sizes) >= 3
        assert tail == min(sizes) and tail <= (2 << 10) + 64 * 16 * 4 + 1024   # tail_bytes + one parameter (+ slot padding)
        assert max(sizes) >= (64 << 10)
        for before_finish, recorded in in_bwd:
            assert before_finish == recorded == len(sizes)
        assert exposed >= 0.0 and seen == world
