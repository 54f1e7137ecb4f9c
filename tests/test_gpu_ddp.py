"""GPU, world_size 2 over gloo with both ranks on the one MI355X: the data-parallel step as bench.py runs it -- HIP model,
flat arenas, gradient slots, ddp.GradBuckets hooks, fused RMSprop -- checked against a single-process computation of the
same thing (mean of the two ranks' gradients).  RCCL needs one GPU per rank: it is exercised over ONE rank (`nccl` backend, collectives are
the identity) -- initialisation, stream and event ordering, not the reduction itself."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build():
    sys.path.insert(0, ROOT)
    import torch.nn as nn
    from heatnet_amd import ops
    from heatnet_amd.layers import HnBatchNorm2d, HnConv2d, conv_bn_act

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.b1 = HnConv2d(32, 64, 3, padding=1, bias=False), HnBatchNorm2d(64)
            self.c2, self.b2 = HnConv2d(64, 32, 1, bias=True), HnBatchNorm2d(32)
            self.head = HnConv2d(32, 5, 1, bias=True)

        def forward(self, x):
            z = conv_bn_act(x, self.c1, self.b1, ops.ACT_RELU)
            z = conv_bn_act(z, self.c2, self.b2, ops.ACT_RELU)
            return self.head(z)

    torch.manual_seed(0)
    return Net()


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(2, 32, 12, 16, generator=g), torch.randint(0, 5, (2, 12, 16), generator=g)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heatnet_amd import ddp, optim, train_step
    dev = torch.device("cuda:0")
    net = _build().to(dev).train()
    ddp.broadcast_module_state(net)
    arenas = optim.arenas_for(net)
    opt = optim.FlatRMSprop(arenas, lr=1e-3)
    buckets = ddp.GradBuckets(arenas=arenas, bucket_bytes=32 << 10)  # several buckets
    x, lab = _data(rank)
    x = x.to(dev).contiguous(memory_format=torch.channels_last)
    for _ in range(2):
        opt.zero_grad()
        buckets.begin_step()
        loss = train_step.segmentation_loss(net(x), lab.to(dev))
        loss.backward()
        buckets.finish_step()
        grad = arenas[0].grad.detach().cpu().clone()
        opt.step()
    ret[rank] = (grad, arenas[0].flat.detach().cpu().clone(), len(buckets.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    (g0, w0, nb), (g1, w1, _) = ret[0], ret[1]
    assert nb >= 2
    assert torch.equal(g0, g1) and torch.equal(w0, w1)   # replicas stay identical: same reduced gradients, same update
    # single process: per-rank gradients computed separately (BatchNorm statistics are per replica), averaged, same update
    sys.path.insert(0, ROOT)
    from heatnet_amd import optim, train_step
    dev = torch.device("cuda:0")
    net = _build().to(dev).train()
    arenas = optim.arenas_for(net)
    opt = optim.FlatRMSprop(arenas, lr=1e-3)
    for _ in range(2):
        acc = torch.zeros_like(arenas[0].grad)
        for rank in range(2):
            opt.zero_grad()
            x, lab = _data(rank)
            train_step.segmentation_loss(net(x.to(dev).contiguous(memory_format=torch.channels_last)), lab.to(dev)).backward()
            acc += arenas[0].grad
            # the second replica must not see the first one's BatchNorm running-statistics update order differently:
            # running buffers are not part of this comparison (per-replica in the reference as well)
        opt.zero_grad()
        for p in arenas[0].params:  # re-attach the slots, then install the averaged gradient
            p.grad = arenas[0].grad_view(p)
        arenas[0].grad.copy_(acc / 2)
        last = arenas[0].grad.detach().cpu().clone()
        opt.step()
    scale = float(last.abs().max())
    assert float((g0 - last).abs().max()) <= 2e-5 * scale, float((g0 - last).abs().max()) / scale
    assert float((w0 - arenas[0].flat.detach().cpu()).abs().max()) <= 1e-4


# ---- RCCL itself, as far as one GPU allows: a process group of ONE rank on the `nccl` backend ---------------------------

def _rccl_worker(port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # as bench.py initialises it
    from heatnet_amd import ddp, ops, optim, train_step
    ops.set_deterministic(True)  # bitwise comparison below
    x, lab = _data(0)
    x = x.to(dev).contiguous(memory_format=torch.channels_last)
    out = {}
    for exchange in (False, True):
        net = _build().to(dev).train()
        ddp.broadcast_module_state(net)
        arenas = optim.arenas_for(net)
        opt = optim.FlatRMSprop(arenas, lr=1e-3)
        buckets = ddp.GradBuckets(arenas=arenas, bucket_bytes=32 << 10, always_exchange=exchange)
        for _ in range(3):
            opt.zero_grad()
            buckets.begin_step()
            train_step.segmentation_loss(net(x), lab.to(dev)).backward()
            buckets.finish_step()
            grad = arenas[0].grad.detach().clone()
            opt.step()
        torch.cuda.synchronize()
        out[exchange] = (grad.cpu(), arenas[0].flat.detach().cpu().clone(), buckets.launched, len(buckets.buckets))
        buckets.remove()
    t = torch.tensor([3.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks reduction and barrier on the same backend
    dist.barrier()
    ret["out"] = out
    ret["max"] = float(t.item())
    dist.destroy_process_group()


def test_rccl_backend_one_rank_buckets_leave_gradients_unchanged():
    """The collective path on the real backend (`nccl` = RCCL): process group with `device_id`, bucketed async all-reduces
    issued from autograd hooks on the communication stream behind an event, the compute stream waiting for them before the
    optimizer.  Over one rank an all-reduce is the identity, so gradients and weights after three steps must be those of a
    run that never exchanges (up to the run-to-run spread of the f64 statistics atomics, ~1e-7 relative)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), ret))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    out = ret["out"]
    (g0, w0, n0, _), (g1, w1, n1, nb) = out[False], out[True]
    assert n0 == 0 and nb >= 2 and n1 == 3 * nb  # every bucket exchanged once per step
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-5 * scale, float((g0 - g1).abs().max()) / scale
    assert float((w0 - w1).abs().max()) <= 1e-4, float((w0 - w1).abs().max())
    assert ret["max"] == 3.0


# ---- confusion-maximisation pattern under data parallelism (BASELINE configs[3]) ----------------------------------------
# The segmentation net is applied TWICE per iteration (day and night batches: the second use of every layer adds into its
# gradient slot in place and reports no gradient of its own, ops.grad_slot) and the trainable half of the model changes
# with the phase (conv_segnet.setPhase, cm/models/conf_segnet.py:86-104; cm/train_trgb_segnet_conf.py:577-592).

def _build_conf():
    sys.path.insert(0, ROOT)
    import torch.nn as nn
    from heatnet_amd import ops
    from heatnet_amd.layers import HnBatchNorm2d, HnConv2d, conv_bn_act

    class Seg(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.b1 = HnConv2d(32, 64, 3, padding=1, bias=False), HnBatchNorm2d(64)
            self.c2, self.b2 = HnConv2d(64, 32, 1, bias=True), HnBatchNorm2d(32)
            self.head = HnConv2d(32, 5, 1, bias=True)

        def forward(self, x):
            z1 = conv_bn_act(x, self.c1, self.b1, ops.ACT_RELU)
            z2 = conv_bn_act(z1, self.c2, self.b2, ops.ACT_RELU)
            out = self.head(z2)
            return out, [out, z1], None

    class Critic(nn.Module):
        def __init__(self, cin):
            super().__init__()
            self.conv1 = HnConv2d(cin, 32, 4, stride=2, padding=1)
            self.classifier = HnConv2d(32, 1, 4, stride=2, padding=1)

        def forward(self, x):
            x = self.classifier(self.conv1(x, act=ops.ACT_LEAKY, slope=0.2))
            return ops.bilinear(x, (4 * x.size(2), 4 * x.size(3)))

    class Conf(nn.Module):  # the slice of conv_segnet the trainer touches
        def __init__(self):
            super().__init__()
            self.trgb_segnet = Seg()
            self.critics = nn.ModuleList([Critic(5), Critic(64)])

        def setPhase(self, phase):
            seg = phase == "train_seg"
            for p in self.critics.parameters():
                p.requires_grad = not seg
            for p in self.trgb_segnet.parameters():
                p.requires_grad = seg

        def forward(self, a, b):
            la, fa, _ = self.trgb_segnet(*a)
            lb, fb, _ = self.trgb_segnet(*b)
            return {"critics_a": [c(fa[i]) for i, c in enumerate(self.critics)],
                    "critics_b": [c(fb[i]) for i, c in enumerate(self.critics)],
                    "pred_label_a": la, "pred_label_b": lb}

    torch.manual_seed(0)
    return Conf()


_PHASES = ["train_critic", "train_seg", "train_seg", "train_critic"]


def _conf_data(rank, dev):
    g = torch.Generator().manual_seed(200 + rank)
    mk = lambda: torch.randn(2, 32, 16, 16, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    return mk(), mk(), torch.randint(0, 5, (2, 16, 16), generator=g).to(dev)


def _conf_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from heatnet_amd import ddp, optim, train_step
    dev = torch.device("cuda:0")
    model = _build_conf().to(dev).train()
    ddp.broadcast_module_state(model)
    model.setPhase(_PHASES[0])
    arenas = optim.arenas_for(model)                 # [segmentation net, critics]
    opt = optim.FlatRMSprop(arenas, lr=1e-3)
    buckets = ddp.GradBuckets(arenas=arenas, bucket_bytes=16 << 10)  # built ONCE, for both phases
    day, night, lab = _conf_data(rank, dev)
    out = []
    for phase in _PHASES:
        model.setPhase(phase)
        train_step.train_step(model, opt, [day], [night], lab, phase, adv="MSE", grad_sync=buckets)
        out.append([a.flat.detach().cpu().clone() for a in arenas])
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_confmax_phases_two_ranks_match_single_process():
    """Layer reuse (day + night) + phase switches + bucket hooks: both ranks stay bit-identical and follow the
    single-process trajectory that averages the two ranks' gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_conf_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = ret[0], ret[1]
    for s0, s1 in zip(r0, r1):
        for a0, a1 in zip(s0, s1):
            assert torch.equal(a0, a1)
    sys.path.insert(0, ROOT)
    from heatnet_amd import optim, train_step
    dev = torch.device("cuda:0")
    model = _build_conf().to(dev).train()
    model.setPhase(_PHASES[0])
    arenas = optim.arenas_for(model)
    opt = optim.FlatRMSprop(arenas, lr=1e-3)
    data = [_conf_data(r, dev) for r in range(2)]

    class Capture:  # grad_sync stand-in: snapshot this "rank"'s gradients instead of exchanging them
        def __init__(self):
            self.grads = []

        def begin_step(self):
            pass

        def finish_step(self):
            self.grads.append([a.grad.detach().clone() for a in arenas])

    class NoStep:  # the optimizer interface train_step drives, without the update
        def zero_grad(self, set_to_none=False):
            opt.zero_grad()

        def step(self):
            pass

    for it, phase in enumerate(_PHASES):
        model.setPhase(phase)
        cap = Capture()
        for day, night, lab in data:
            train_step.train_step(model, NoStep(), [day], [night], lab, phase, adv="MSE", grad_sync=cap)
        active = 0 if phase == "train_seg" else 1
        opt.zero_grad()
        for p in arenas[active].params:
            p.grad = arenas[active].grad_view(p)
            p._hn_slot_used = True
        arenas[active].grad.copy_((cap.grads[0][active] + cap.grads[1][active]) / 2)
        opt.step()
        for k in range(2):
            ref = arenas[k].flat.detach().cpu()
            err = float((r0[it][k] - ref).abs().max())
            assert err <= 1e-4, (it, phase, k, err)
        # the frozen arena's weights did not move in this iteration
        if it > 0:
            assert torch.equal(r0[it][1 - active], r0[it - 1][1 - active])


def test_foreign_zero_grad_does_not_accumulate_stale_gradients():
    """nn.Module.zero_grad() (set_to_none) instead of the arena optimizer's: the next backward must start from an empty
    slot, not add to the previous step's contents (ops.grad_slot keyed only on `_hn_slot_used`)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    sys.path.insert(0, ROOT)
    from heatnet_amd import optim, train_step
    dev = torch.device("cuda:0")
    net = _build().to(dev).train()
    arenas = optim.arenas_for(net)
    x, lab = _data(0)
    x = x.to(dev).contiguous(memory_format=torch.channels_last)
    grads = []
    for _ in range(2):
        net.zero_grad()  # torch >= 2: p.grad = None, arena not cleared
        train_step.segmentation_loss(net(x), lab.to(dev)).backward()
        grads.append(arenas[0].grad.detach().clone())
        arenas[0].check()
    scale = float(grads[0].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale  # same inputs, same weights: same gradient, not 2x
    # dropping gradients AFTER backward and then stepping is refused
    opt = optim.FlatRMSprop(arenas, lr=1e-3)
    net.zero_grad()
    with pytest.raises(RuntimeError, match="dropped after backward"):
        opt.step()


# ---- the reference's own wrapper: torch DistributedDataParallel around the built network (scripts/main.py:119-123) ------

def _torch_ddp_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _seeded import seeded_inputs, seeded_state, set_dropout
    from heatnet_amd import ddp, ops, train_step
    from models.confusion_maximization.models import build_net  # the reference's import path (overlay package)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    # no float atomics: the two exchanges are compared on gradients that do not depend on atomic arrival order (the stem
    # gradients of this random-weight network move by ~1e-2 between two ordinary runs, profiles/r01_grad_error_run_spread.txt)
    ops.set_deterministic(True)

    def fresh():
        net = build_net.build_network(None, "resnet50", in_channels=4, late_fusion=True)  # ends with .cuda(), as the reference
        net.load_state_dict(seeded_state(net, seed=1))
        set_dropout(net, 0.0)
        return net.train()

    rgb, ir, lab = [t.to(dev) for t in seeded_inputs(2, 64, 96, seed=30 + rank)]
    # (a) exactly what scripts/main.py:123 does
    wrapped = torch.nn.parallel.DistributedDataParallel(fresh(), device_ids=[0])
    out = wrapped(rgb, ir)[0]
    train_step.segmentation_loss(out, lab).backward()
    g_ddp = {k: p.grad.detach().cpu().clone() for k, p in wrapped.module.named_parameters()}
    bn_ddp = wrapped.module.state_dict()["feats.bn1.running_mean"].cpu().clone()
    # (b) the library's own exchange on an identical replica
    net = fresh()
    buckets = ddp.GradBuckets(list(net.parameters()), bucket_bytes=8 << 20)
    buckets.begin_step()
    train_step.segmentation_loss(net(rgb, ir)[0], lab).backward()
    buckets.finish_step()
    g_own = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
    ret[rank] = (g_ddp, g_own, bn_ddp)
    dist.barrier()
    dist.destroy_process_group()


def test_torch_ddp_wrapper_around_built_network_matches_grad_buckets():
    """`DistributedDataParallel(build_network(...))` as scripts/main.py:119-123 wraps it (world size 2, gloo, both ranks on
    the one GPU): the averaged gradients equal those of heatnet_amd.ddp.GradBuckets on the same replicas, on both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_torch_ddp_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
        assert p.exitcode == 0
    (d0, o0, _), (d1, o1, _) = ret[0], ret[1]
    assert sorted(d0) == sorted(o0) and len(d0) > 200
    worst = 0.0
    for k in d0:
        scale = float(o0[k].abs().max()) + 1e-12
        assert float((d0[k] - d1[k]).abs().max()) <= 1e-6 * scale + 1e-12, k       # both ranks hold the same average
        worst = max(worst, float((d0[k] - o0[k]).abs().max()) / scale)
        assert float((d0[k] - o0[k]).abs().max()) <= 2e-3 * scale + 1e-9, k       # == the library's own exchange
