"""debug: test_fused_conv_bn_act_matches_unfused's failing case with every dgrad tile forced in turn"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch.nn as nn, torch.nn.functional as F
from heatnet_amd import layers, ops
dev = torch.device("cuda:0")
cin, cout, k, stride, dil = 128, 128, 3, 1, 2
n, h, w_ = 2, 72, 80
pad = dil * (k - 1) // 2
orig_pick = ops._pick
forced = {}
def pick(kind, d, cands, launch):
    for pre, v in forced.items():
        if kind.startswith(pre):
            for c in cands: launch(c)
            print("   ", kind, "candidates", cands, "forced", v, flush=True)
            return v
    return orig_pick(kind, d, cands, launch)
ops._pick = pick
def run(planes):
    prev = ops.PLANES; ops.PLANES = planes
    try:
        torch.manual_seed(5)
        conv = layers.HnConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False).to(dev)
        bn = layers.HnBatchNorm2d(cout).to(dev)
        g = torch.Generator().manual_seed(7)
        with torch.no_grad():
            conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) / np.sqrt(cin * k * k)).to(dev))
            bn.weight.copy_((torch.rand(cout, generator=g) + 0.5).to(dev))
            bn.bias.copy_((torch.randn(cout, generator=g) * 0.3).to(dev))
        x = F.relu(torch.randn(n, cin, h, w_, generator=g)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        z = layers.conv_bn_act(x * 1.0, conv, bn, ops.ACT_RELU, residual=None, prelu_weight=None, drop_mask=None, out="planes", mask_max=1.0)
        zf = ops.planes_to_tensor(z) if ops.is_planes(z) else z.detach()
        gz = (torch.randn(n, cout, h, w_, generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
        z.backward(gz)
        return dict(z=zf.float().cpu(), dx=x.grad.float().cpu(), dw=conv.weight.grad.float().cpu())
    finally:
        ops.PLANES = prev
ops.set_conv_math("f16x2")
b = run(False)
for v in (None, 0, 1, 2, 3, 4, 5, 6):
    forced.clear()
    if v is not None: forced["dgrad.planes"] = v
    a = run(True)
    print("tile", v, {k_: float((a[k_] - b[k_]).abs().max() / b[k_].abs().max()) for k_ in b}, flush=True)
    if v is not None:
        e = (a["dx"] - b["dx"]).abs()
        idx = (e > 1e-4 * b["dx"].abs().max()).nonzero()
        print("   bad elements", len(idx), "first", idx[:3].tolist(), "last", idx[-3:].tolist())
