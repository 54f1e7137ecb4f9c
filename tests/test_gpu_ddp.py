"""GPU, world_size 2 over gloo with both ranks on the one MI355X: the data-parallel step as bench.py runs it -- HIP model,
flat arenas, gradient slots, ddp.GradBuckets hooks, fused RMSprop -- checked against a single-process computation of the
same thing (mean of the two ranks' gradients).  RCCL itself cannot be exercised with one GPU; everything above it is."""
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
