"""f16x2 on a TRAINED network (GPU box).  VERDICT r2 item 3c: the head-room evidence of round 2 came from random-weight networks
on noise with random labels (near-uniform logits, flat cross-entropy gradients).  This script

  1. trains the dual-stream PSPNet on a LEARNABLE synthetic task -- smooth random RGB / thermal fields, the label a
     quantised function of the input -- with the reference's optimizer setting (RMSprop, lr 1e-4) until the cross-entropy is
     below 0.3 (confident pixels beside wrong ones: the dY tensors then span many orders of magnitude);
  2. runs the range census (hn_range_census) over every f16x2 convolution operand of one further training iteration:
     fraction of the non-zero elements more than 2^18 below their tensor's magnitude bound, activations and gradients;
  3. compares the parameter gradients of one iteration with an fp64 run of the CPU oracle on the same trained state, under
     the fp32 MFMA, bf16x3 and f16x2 (plane tensors on and off).

usage: python tools/trained_state_f16x2.py [max_steps] [batch h w]      -> stdout (commit under profiles/)"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import set_dropout  # noqa: E402
from heatnet_amd import ops, optim, train_step  # noqa: E402
from heatnet_amd.conf_segnet import weights_init_normal  # noqa: E402
from heatnet_amd.pspnet import PSPNet  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def smooth_batch(n, h, w, gen, dev):
    """Low-frequency random fields in [-1, 1] and labels that are a function of them (13 classes)."""
    def field(c):
        low = torch.randn(n, c, 6, 12, generator=gen)
        return torch.tanh(F.interpolate(low, size=(h, w), mode="bicubic", align_corners=False))
    rgb, ir = field(3), field(1)
    score = 0.5 + 0.5 * torch.tanh(1.5 * ir[:, 0] + 0.75 * rgb[:, 0] - 0.5 * rgb[:, 2])
    lab = (score * 13).floor().clamp_(0, 12).long()
    return rgb.to(dev), ir.to(dev), lab.to(dev)


def main():
    max_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (4, 320, 704)
    dev = torch.device("cuda:0")
    ops.set_conv_math("f16x2")
    ops.set_range_census(False)
    torch.manual_seed(0)
    net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    net.apply(weights_init_normal)
    net = net.to(dev).train()
    arenas = optim.arenas_for(net)
    opt = optim.FlatRMSprop(arenas, lr=1e-4)
    gen = torch.Generator().manual_seed(11)
    pool = [smooth_batch(n, h, w, gen, dev) for _ in range(int(os.environ.get("POOL", "8")))]  # a small fixed training set
    t0 = time.time()
    ema, step = None, 0
    print("training: PSPNet (ResNet-50, late fusion), bs %d, %dx%d, RMSprop lr 1e-4, f16x2 (plane tensors %s), dropout on"
          % (n, h, w, "on" if ops.PLANES else "off"))
    while step < max_steps:
        rgb, ir, lab = pool[step % len(pool)]
        loss = float(train_step.segmentation_step(net, opt, rgb, ir, lab))
        ema = loss if ema is None else 0.95 * ema + 0.05 * loss
        step += 1
        if step % 100 == 0 or step == 1:
            print("  step %5d  CE %.4f  (ema %.4f)  %.0f s" % (step, loss, ema, time.time() - t0), flush=True)
        if not math.isfinite(loss):
            print("  non-finite loss at step %d" % step)
            break
        if ema < 0.3 and step >= 200:
            break
    print("trained: %d steps, CE ema %.4f, last %.4f" % (step, ema, loss))

    # ---- 2. range census of one more iteration on a batch of the training set (dropout on, as in training) ----
    rgb, ir, lab = pool[0]
    ops._census.clear()
    ops.set_range_census(True)
    for a in arenas:
        a.zero_grad()
    with ops.frozen_weights():
        ops.prepare_weights()
        logits = net(rgb, ir)[0]
    ce = train_step.segmentation_loss(logits, lab)
    ce.backward()
    ops.set_range_census(False)
    prob = torch.softmax(logits.detach().float(), 1)
    pmax = prob.max(1)[0]
    acc = float((logits.argmax(1) == lab).float().mean())
    print("\ncensus batch: CE %.4f, pixel accuracy %.3f; softmax max: %.1f%% of pixels > 0.99, %.1f%% < 0.5"
          % (float(ce.detach()), acc, 100 * float((pmax > 0.99).float().mean()), 100 * float((pmax < 0.5).float().mean())))
    dl = (prob - F.one_hot(lab, 13).permute(0, 3, 1, 2).float()).abs()
    nzd = dl[dl > 0]
    print("|dCE/dlogit| (x N pixels): max %.3e, median %.3e, 1st percentile %.3e -> spans 2^%.1f"
          % (float(nzd.max()), float(nzd.median()), float(nzd.kthvalue(max(1, nzd.numel() // 100))[0]),
             math.log2(float(nzd.max()) / float(nzd.kthvalue(max(1, nzd.numel() // 100))[0]))))
    rows = ops.range_report()
    rows.sort(key=lambda r: -max(r[2], r[4]))
    print("\nrange census, %d f16x2 layers (fraction of an operand's NON-ZERO elements more than 2^18 below the tensor's bound)" % len(rows))
    print("%-24s %14s %12s %14s %12s" % ("weight shape", "activ. nonzero", "frac lossy", "grad nonzero", "frac lossy"))
    for geom, nza, fa, nzg, fg, _ in rows[:16]:
        print("%-24s %14d %12.3e %14d %12.3e" % (str(geom), nza, fa, nzg, fg))
    fa_all = sorted(r[2] for r in rows)
    fg_all = sorted(r[4] for r in rows)
    q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
    print("activation operands: median %.2e, 90%% %.2e, max %.2e;  gradient operands: median %.2e, 90%% %.2e, max %.2e"
          % (q(fa_all, .5), q(fa_all, .9), fa_all[-1], q(fg_all, .5), q(fg_all, .9), fg_all[-1]))
    print("layers above a 5%% guard threshold: %d of %d" % (sum(1 for r in rows if max(r[2], r[4]) > 0.05), len(rows)))

    # ---- 3. gradients against an fp64 run of the oracle on the trained state (smaller geometry: the fp64 CPU run) ----
    gn, gh, gw = 2, 192, 384  # (> 2048 pixels on the stride-8 maps: the fused plane-tensor node covers every bottleneck)
    rgb, ir, lab = smooth_batch(gn, gh, gw, gen, torch.device("cpu"))
    state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    ref.load_state_dict(state)
    set_dropout(ref, 0.0)
    ref = ref.double().train()
    F.cross_entropy(ref(rgb.double(), ir.double())[0], lab).backward()
    g64 = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    ref32 = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    ref32.load_state_dict(state)
    set_dropout(ref32, 0.0)
    ref32.train()
    F.cross_entropy(ref32(rgb, ir)[0], lab).backward()
    runs = {"cpu fp32 oracle": {k: p.grad.detach().double() for k, p in ref32.named_parameters()}}
    for name, math_, planes in (("hip f32", "f32", True), ("hip bf16x3", "bf16x3", True), ("hip f16x2 planes", "f16x2", True),
                                ("hip f16x2 no planes", "f16x2", False)):
        ops.set_conv_math(math_)
        prev = ops.PLANES
        ops.PLANES = planes
        try:
            m = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
            m.load_state_dict(state)
            set_dropout(m, 0.0)
            m = m.to(dev).train()
            lg = m(rgb.to(dev), ir.to(dev))[0]
            train_step.segmentation_loss(lg, lab.to(dev)).backward()
            runs[name] = {k: p.grad.detach().cpu().double() for k, p in m.named_parameters()}
        finally:
            ops.PLANES = prev
    ops.set_conv_math("f16x2")
    print("\nparameter gradients of one iteration at %dx%dx%d on the TRAINED state vs the fp64 oracle "
          "(per parameter tensor: |g - g64| / |g64|, cosine)" % (gn, gh, gw))
    # a conv bias in front of a batch-statistics BatchNorm has an exactly-zero true gradient (rounding noise in fp64, exact
    # zeros here): parameters at that noise level are left out
    floor = 1e-6 * max(float(g.norm()) for g in g64.values())
    print("%-22s %12s %12s %12s %14s   worst parameter" % ("run", "median rel", "90% rel", "max rel", "min cosine"))
    for name, gs in runs.items():
        rel, cos = [], []
        for k, g in gs.items():
            r = g64[k]
            if float(r.norm()) < floor:
                continue
            rel.append((float((g - r).norm() / r.norm()), k))
            cos.append(float((g * r).sum() / (g.norm() * r.norm() + 1e-300)))
        rel.sort()
        vals = [v for v, _ in rel]
        print("%-22s %12.3e %12.3e %12.3e %14.6f   %s" % (name, q(vals, .5), q(vals, .9), vals[-1], min(cos), rel[-1][1]))


if __name__ == "__main__":
    main()
