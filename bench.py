#!/usr/bin/env python3
"""Headline benchmark: training images/s of the HeatNet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload seg|confmax_seg|confmax_critic]

A step = one pass of the hot path over one synthetic batch resident in HBM:
  seg (default, BASELINE.json configs[1]): dual-stream RGB+thermal PSPNet (ResNet-50, late fusion), bs=8/GPU,
      320x704, forward + per-pixel CE + backward + RMSprop update.
  confmax_seg / confmax_critic (configs[2]): one train_seg / train_critic iteration of the confusion-maximisation
      trainer (day + night passes, six critics); an "image" is then one day+night pair.
N>1: one process per GPU, full replica each, gradients all-reduced by RCCL in flat buckets overlapped with backward (weak
scaling).  Either launch it as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, or just run
`python bench.py --gpus N`: without a torchrun environment this process starts the N workers itself (as a child
`torch.distributed.run`, before anything here touches the GPU), relays their output and exits with their code.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernels -- the implicit-GEMM convolutions -- as
algorithmic conv FLOPs (2*M*N*K of every conv forward / data-gradient / weight-gradient launch) over the HIP-event time
of exactly those launches, measured in an instrumented pass after the timed region.
`cpu_baseline` times the CPU oracle (oracle/heatnet_ref.py, a port) on this host, on a bounded sample.

Conv arithmetic (HEATNET_CONV_MATH): the default "f16x2" computes the fp32 convolution on the
fp16 matrix cores -- every operand tensor scaled by a power of two and split into two fp16 terms inside the kernel, three
products accumulated in fp32 -- with an error against fp64 at or below the fp32 MFMA's (tests/test_gpu_kernels.py::
test_conv2d_bf16x3_is_fp32_accurate, profiles/r02_f16x2_accuracy.txt); its roofline peak is the dense fp16 MFMA peak / 3.
"bf16x3" (three bf16 terms, six products, peak / 6) and "f32" (v_mfma_f32_32x32x2_f32) run the same step; at N=1 both are
timed too, in the same process (`alt_conv_math` = the fp32 MFMA, `alt_conv_maths` = both), so all numbers come from one run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The N > 1 path keeps four HIP streams busy per process (compute, weight gradients, gradient exchange, RCCL's own).  With
# the runtime's default of four hardware queues the weight-gradient stream stopped overlapping the compute stream as soon as
# the process group existed: 42.6 ms per step through the N > 1 path on one rank against 40.2 with eight queues (39.9 / 39.8
# without a process group; profiles/r03_hw_queues.txt).  Read when the HIP runtime initialises: set before torch loads it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# MI355X_MICROARCH.md, dense peaks: v_mfma_f32_32x32x2_f32 157.3 TFLOP/s; bf16 MFMA ~2500 TFLOP/s, of which one fp32
# product costs six MFMAs under bf16x3
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0 / 6.0, "bf16": 2500.0, "f16x2": 2500.0 / 3.0}  # (fp16 MFMA peak = bf16's)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured by a streaming copy)
KERNEL_LABEL = {"f32": "conv_igemm_kernel / conv_wgrad_kernel (v_mfma_f32_32x32x2_f32)",
                "bf16x3": "conv_igemm_kernel<..,MATH_X3> / conv_wgrad_x3_kernel (6 x v_mfma_f32_16x16x32_bf16 per fp32 "
                          "product; peak = dense bf16 MFMA peak / 6)",
                "bf16": "conv_igemm2_kernel<..,1,..> / conv_wgrad_x3_kernel<..,1> (one v_mfma_f32_16x16x32_bf16 per product)",
                "f16x2": "conv_igemm4_kernel / conv_wgrad4_kernel on plane tensors (all LDS-DMA), conv_igemm{,3}_kernel<..,MATH_H2> / "
                         "conv_wgrad_x3_kernel<..,2> elsewhere (3 x v_mfma_f32_16x16x32_f16 per fp32 product; peak = dense fp16 "
                         "MFMA peak / 3)"}
CONV_MATH_NOTE = {"f32": "f32: v_mfma_f32_32x32x2_f32 on the fp32 operands",
                  "bf16x3": "bf16x3: fp32 operands split exactly into 3 bf16 terms in-kernel, 6 bf16 MFMAs per product, fp32 "
                            "accumulate; error vs fp64 equal to the fp32 MFMA's (profiles/r01_bf16x3_probe.txt)",
                  "f16x2": "f16x2: fp32 RESULTS from two fp16 terms per operand -- every operand tensor scaled by one power of two "
                           "(from a bound of its magnitude) and split into hi + lo fp16, by the pass that produces it (plane "
                           "tensors) or in the kernel; 3 fp16 MFMAs per product, fp32 accumulate; error vs fp64 at or below the "
                           "fp32 MFMA's (profiles/r02_f16x2_accuracy.txt); elements > 2^18 below their tensor's maximum lose "
                           "bits (HEATNET_CONV_MATH=bf16x3 has no such range); Cin % 32 != 0 layers on the fp32 MFMA",
                  "bf16": "bf16 (REDUCED precision, BASELINE configs[4]): conv operands rounded to nearest bf16 in the "
                          "loaders, one bf16 MFMA per product, fp32 accumulate; activations in HBM, BatchNorm statistics, "
                          "master weights and optimizer fp32; stems / 13-class head (Cin % 32 != 0) on the fp32 MFMA"}
H, W, BS = 320, 704, 8
# conv FLOPs of one step in the REFERENCE's op graph (per image/pair x 8)
REFERENCE_CONV_TFLOP = {"seg": 8 * 1.0738, "seg_bf16": 8 * 1.0738, "confmax_seg": 8 * 2.2532, "confmax_critic": 8 * 0.8484}


def synthetic_batch(rank, dev, n=BS, h=H, w=W):
    g = torch.Generator().manual_seed(1234 + rank)
    mk = lambda c: torch.randn(n, c, h, w, generator=g).clamp_(-1, 1)
    batch = {"rgb_day": mk(3), "ir_day": torch.rand(n, 1, h, w, generator=g) * 2 - 1,
             "rgb_night": mk(3), "ir_night": torch.rand(n, 1, h, w, generator=g) * 2 - 1,
             "label_day": torch.randint(0, 13, (n, h, w), generator=g)}
    return {k: v.to(dev) for k, v in batch.items()}


def build(workload, dev):
    from heatnet_amd import conf_segnet
    from heatnet_amd.conf_segnet import weights_init_normal
    from heatnet_amd.pspnet import PSPNet
    torch.manual_seed(0)
    if workload in ("seg", "seg_bf16"):
        model = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
        model.apply(weights_init_normal)
    elif workload == "seg_resnext":
        from heatnet_amd.trgb_segnet import net_resnext50
        model = net_resnext50(classes=13, input_channels=4, late_fusion=True)
        model.apply(weights_init_normal)
    else:
        model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet",
                                        late_fusion=True)
        model.setPhase("train_seg" if workload == "confmax_seg" else "train_critic")
    return model.to(dev).train()


def make_step(workload, model, opt, batch, buckets):
    from heatnet_amd import train_step
    if workload in ("seg", "seg_bf16", "seg_resnext"):
        return lambda: train_step.segmentation_step(model, opt, batch["rgb_day"], batch["ir_day"], batch["label_day"],
                                                    grad_sync=buckets)
    state = "train_seg" if workload == "confmax_seg" else "train_critic"
    return lambda: train_step.train_step(model, opt, [batch["rgb_day"], batch["ir_day"]],
                                         [batch["rgb_night"], batch["ir_night"]], batch["label_day"], state,
                                         adv="MSE", grad_sync=buckets)


def measure_conv_roofline(step_fn, steps, workload, scale=1.0, geom=(BS, H, W)):
    """Instrumented pass: HIP events (torch's current stream = the launch stream) around every conv-kernel launch."""
    from heatnet_amd import ops
    records = []
    ops.set_conv_profiler(records)
    try:
        for _ in range(steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        ops.set_conv_profiler(None)
    per_kind = {}
    flops = ms = 0.0
    table = {}
    nbytes = 0.0
    inst = {}
    for kind, fl, e0, e1, shape, by, kname in records:
        nbytes += by
        ie = inst.setdefault(kname, [0, 0.0])
        ie[0] += 1
        ie[1] += by
        t = e0.elapsed_time(e1)
        row = table.setdefault((kind,) + shape, [0, 0.0, 0.0])
        row[0] += 1
        row[1] += fl
        row[2] += t
        k = per_kind.setdefault(kind, [0, 0.0, 0.0])
        k[0] += 1
        k[1] += fl
        k[2] += t
        flops += fl
        ms += t
    launches = len(records)
    dump = os.environ.get("HN_CONV_TABLE")
    if dump:  # per-shape table for kernel tuning: kind, geometry, launches/step, ms/step, TFLOP/s
        with open(dump, "w") as f:
            f.write("kind N Hin Win Cin Hout Wout Cout K stride dil launches ms_per_step tflops\n")
            for key, (cnt, fl, t) in sorted(table.items(), key=lambda kv: -kv[1][2]):
                f.write(" ".join(str(v) for v in key) + " %d %.3f %.1f\n" % (cnt // steps, t / steps, fl / t / 1e9))
    detail = {k: {"launches_per_step": v[0] // steps, "tflop_per_step": v[1] / steps / 1e12,
                  "ms_per_step": v[2] / steps, "avg_launch_us": 1e3 * v[2] / v[0], "tflops": v[1] / v[2] / 1e9}
              for k, v in per_kind.items()}
    achieved = flops / ms / 1e9 if ms > 0 else 0.0
    ref_tflop = REFERENCE_CONV_TFLOP[workload] * scale if (workload in REFERENCE_CONV_TFLOP and abs(scale - round(scale)) < 1e-9) else None
    math = ops.CONV_MATH
    peak = round(PEAK_TFLOPS[math], 1)
    traffic_all, traffic_per_kernel, traffic_source = pmc_traffic(math, workload, *geom)
    # algorithmic (compulsory) bytes per launch of each kernel instantiation, next to its measured traffic
    per_inst = {}
    for name, (cnt, by) in inst.items():
        per_inst[name] = {"launches_per_step": cnt // steps, "algorithmic_bytes_per_launch": round(by / cnt),
                          "traffic": (traffic_per_kernel or {}).get(name)}
    # the same launches priced against bytes: compulsory traffic (each of a convolution's three tensors touched once)
    # over the same time.  With one MFMA per product (bf16) the small-K layers are HBM-bound and the byte view is the
    # tighter of the two; `bound` names the roof the kernels sit closer to.
    gbs = nbytes / ms / 1e6 if ms > 0 else 0.0
    hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    mfma = {"achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4)}
    bound = "hbm" if hbm["frac"] > mfma["frac"] else "mfma"
    head = hbm if bound == "hbm" else mfma
    return {"bound": bound, "achieved": head["achieved"], "peak": head["peak"], "unit": head["unit"],
            "frac": head["frac"], "mfma_view": mfma, "hbm_view": hbm,
            "traffic": traffic_all, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": round(nbytes / max(launches, 1)),
            "by_instantiation": per_inst,
            "kernel": KERNEL_LABEL[math], "conv_math": math,
            "launches_per_step": launches // steps,
            # executed = 2*M*N*K of the launches actually made (the pyramid bottleneck is applied per level at the
            # pooled resolution); reference = the same step in the reference's op graph
            # (1073.8 GFLOP/image).  `achieved`/`frac` use the EXECUTED count.
            "conv_tflop_per_step": round(flops / steps / 1e12, 3),
            "reference_op_tflop_per_step": ref_tflop,
            "achieved_vs_reference_op_count": round(ref_tflop * 1e3 / (ms / steps), 2) if ref_tflop else None,
            "conv_ms_per_step": round(ms / steps, 3), "avg_launch_us": round(1e3 * ms / max(launches, 1), 2),
            "by_kind": detail}


def pmc_traffic(math, workload, batch, height, width):
    """HBM bytes per conv-kernel launch from PMC passes of THIS benchmark (rocprofv3 --pmc FETCH_SIZE and --pmc
    WRITE_SIZE, separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md; tools/pmc_aggregate.py).  Counters cannot be
    collected from inside this process, so the figures come from profiles/pmc_traffic_<math>.json -- and ONLY when that
    file's `_meta` says it was measured on the same workload, geometry, conv arithmetic and kernel generation
    (hn_version) as this run; otherwise they are null.  `source` names the file and the commit it was measured at.
    Returns (mean over ALL conv-kernel launches, per-instantiation {kernel: bytes per launch}, source)."""
    from heatnet_amd import _lib
    name = "pmc_traffic_%s.json" % math
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            data = json.load(f)
        meta = data.pop("_meta")
        want = {"workload": workload, "batch": batch, "height": height, "width": width, "conv_math": math,
                "hn_version": _lib.load().hn_version()}
        if any(meta.get(k) != v for k, v in want.items()):
            return None, None, None
        conv = {k: v for k, v in data.items() if k.startswith(("conv_igemm", "conv_wgrad", "stem_", "narrow_"))}
        calls = sum(v["calls"] for v in conv.values())
        mean = sum(v["calls"] * v["hbm_bytes_per_launch"] for v in conv.values()) / calls
        per = {k: round(v["hbm_bytes_per_launch"]) for k, v in conv.items()}
        return round(mean), per, "profiles/%s @ %s" % (name, meta.get("commit", "?"))
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None, None, None


def host_cpu():
    """(model name, physical cores) of this host from /proc/cpuinfo; falls back to the logical count."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    n = len(cores) or (os.cpu_count() or 1)
    try:  # never more threads than this process may run on (cgroup / affinity limits of the box)
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return model, n


def cpu_baseline(workload):
    """The oracle on this host's cores: bs=1, 320x704, 1 warm-up + 2 timed steps (about 15-30 s of CPU work)."""
    from oracle import heatnet_ref as R
    cpu_model, cores = host_cpu()
    torch.set_num_threads(cores)  # one thread per physical core
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(99)
    rgb, ir = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1), torch.rand(1, 1, H, W, generator=g) * 2 - 1
    lab = torch.randint(0, 13, (1, H, W), generator=g)
    if workload in ("seg", "seg_bf16"):
        model = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4).train()
        opt = torch.optim.RMSprop(model.parameters(), lr=1e-4)

        def step():
            opt.zero_grad()
            loss = torch.nn.functional.cross_entropy(model(rgb, ir)[0], lab)
            loss.backward()
            opt.step()
    else:
        model = R.RefConfSegnet().train()
        state = "train_seg" if workload == "confmax_seg" else "train_critic"
        model.setPhase(state)
        opt = torch.optim.RMSprop(model.parameters(), lr=1e-4)
        rgb2, ir2 = torch.randn(1, 3, H, W, generator=g).clamp_(-1, 1), torch.rand(1, 1, H, W, generator=g) * 2 - 1
        step = lambda: R.train_step(model, opt, [rgb, ir], [rgb2, ir2], lab, state)
    step()
    t0 = time.time()
    n = 2
    for _ in range(n):
        step()
    dt = (time.time() - t0) / n
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
            "sample": "oracle/heatnet_ref.py, %s step at bs=1 320x704 fp32, %d threads (one per physical core), 1 warm-up "
                      "+ %d timed steps, %.1f s/step" % (workload, cores, n, dt)}


def self_launch(n):
    """`python bench.py --gpus N` from a bare shell: start the N workers (one process per GPU) as a child
    `torch.distributed.run`, which sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for them.  This parent never
    initialises the GPU (it only spawns -- no exec from a process that touched HIP); the workers' stdout (rank 0's JSON
    line) and stderr pass straight through, and a worker failure is this process's exit code."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="seg", choices=["seg", "seg_bf16", "confmax_seg", "confmax_critic", "seg_resnext"])
    ap.add_argument("--height", type=int, default=H, help="input height (default: the BASELINE metric's 320)")
    ap.add_argument("--width", type=int, default=W, help="input width (default 704)")
    ap.add_argument("--batch", type=int, default=BS, help="images per GPU (default 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timing under the other conv arithmetic")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the extra timings of the bf16 and confusion-maximisation workloads (default run, N=1, seg only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # nothing has touched the GPU yet in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal knobs for a 1-GPU box (never set by the driver): every rank on one device, gloo instead of RCCL
    if os.environ.get("HN_BENCH_SHARE_GPU"):
        local = 0
    backend = os.environ.get("HN_BENCH_BACKEND", "nccl")
    # HN_BENCH_FORCE_DIST=1: the N=1 run goes through the whole N>1 path on RCCL (process group of one rank, bucketed
    # all-reduces on the communication stream, barriers, max over ranks) -- the rehearsal a one-GPU box allows
    force_dist = world == 1 and os.environ.get("HN_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            sock.close()
        os.environ.update(RANK="0", WORLD_SIZE="1")
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if os.environ.get("HN_BENCH_FAIL_RANK") == str(rank) and world > 1:
        raise RuntimeError("HN_BENCH_FAIL_RANK: injected worker failure (tests/test_gpu_bench_contract.py)")
    from heatnet_amd import ddp, optim, ops
    if args.workload == "seg_bf16":
        ops.set_conv_math("bf16")  # reduced-precision arithmetic: a separate workload, never the headline
    model = build(args.workload, dev)
    if distributed:
        ddp.broadcast_module_state(model)
    arenas = optim.arenas_for(model)          # flat parameter / gradient storage per training phase
    opt = optim.FlatRMSprop(arenas, lr=1e-4)  # one fused kernel launch per step (cm/train_trgb_segnet_conf.py:270)
    buckets = ddp.GradBuckets(arenas=arenas, always_exchange=force_dist) if distributed else None
    batch = synthetic_batch(rank, dev, args.batch, args.height, args.width)
    step = make_step(args.workload, model, opt, batch, buckets)

    step()  # priming step, always untimed: first-use costs (kernel-variant timing for geometries missing from the tune
    #         cache, LDS attribute calls, allocator growth) must not land in the timed region even with --warmup 0
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
        buckets.reset_exposed()
    torch.cuda.synchronize()
    for k_ in ops.bn_reduce_counts:
        ops.bn_reduce_counts[k_] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # how the BatchNorm backward reductions of a step ran: in a data-gradient epilogue / as a pass of their own
    bn_passes = {k_ + "_per_step": round(v_ / max(args.steps, 1), 2) for k_, v_ in ops.bn_reduce_counts.items()}
    comm = None
    if distributed:
        # per rank: time the compute stream idled waiting for gradient collectives (what backward did not hide)
        t = torch.tensor([elapsed, buckets.exposed_ms(args.steps)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        comm = {"backend": backend, "world_size": dist.get_world_size(),  # as the process group (RCCL) sees it
                "comm_exposed_ms": round(float(t[1].item()), 3), "buckets": len(buckets.buckets),
                "bucket_mib": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in buckets.buckets if b["armed"]],
                "launched_in_backward": buckets.launched_in_backward,
                "note": "comm_exposed_ms = max over ranks of the mean time per step the compute stream waited for the bucketed "
                        "all-reduces in finish_step() (events on the compute stream)"}

    scale = (args.height * args.width * args.batch) / float(H * W * BS)
    roof = None
    if not args.no_roofline:
        # every rank runs the instrumented steps (a step contains the gradient all-reduce); only rank 0 records events
        nprof = max(1, min(3, args.steps))
        if rank == 0:
            roof = measure_conv_roofline(step, nprof, args.workload, scale=scale,
                                         geom=(args.batch, args.height, args.width))
        else:
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    alt = None
    alts = {}
    if world == 1 and not args.no_alt and args.workload != "seg_bf16":
        # the same step under the other fp32-result conv arithmetics, timed the same way (N=1 only: like cpu_baseline,
        # reference figures printed beside the headline, not part of it): the fp32 MFMA first (`alt_conv_math`), then
        # whichever of bf16x3 / f16x2 is not the headline's
        main_math = ops.CONV_MATH
        for other in [m for m in ("f32", "bf16x3", "f16x2") if m != main_math]:
            ops.set_conv_math(other)
            try:
                for _ in range(max(2, args.warmup)):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                alt_elapsed = time.perf_counter() - t1
                alt_roof = None
                if not args.no_roofline:
                    r = measure_conv_roofline(step, max(1, min(3, args.steps)), args.workload, scale=scale,
                                              geom=(args.batch, args.height, args.width))
                    alt_roof = dict({k: r[k] for k in ("bound", "conv_ms_per_step", "kernel")}, **r["mfma_view"])
                alts[other] = {"conv_math": other, "value": round(args.batch * args.steps / alt_elapsed, 3), "unit": "images/s",
                               "ms_per_step": round(1e3 * alt_elapsed / args.steps, 3), "steps": args.steps,
                               "roofline": alt_roof}
            finally:
                ops.set_conv_math(main_math)
        alt = alts.get("f32") or next(iter(alts.values()))
    others = None
    if world == 1 and args.workload == "seg" and not args.no_other_workloads:
        # the other BASELINE configurations, timed in the same process so that one default run records them all (N=1
        # only; reference figures beside the headline, never part of `value`): configs[4] = the same step under the
        # mixed-precision bf16 arithmetic, configs[2] = the two confusion-maximisation iterations (an image = a pair)
        others = {}
        n_other = max(2, min(args.steps, 10))

        def timed(fn, images):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_other):
                fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            return {"value": round(images * n_other / dt, 3), "ms_per_step": round(1e3 * dt / n_other, 3), "steps": n_other}

        main_math = ops.set_conv_math("bf16")
        try:
            others["seg_bf16"] = dict(timed(step, args.batch), unit="images/s", dtype="bf16",
                                      workload="configs[4] arithmetic: the configs[1] step on the mixed-precision bf16 conv path")
        finally:
            ops.set_conv_math(main_math)
        del step, opt, arenas, model
        torch.cuda.empty_cache()
        for wl in ("confmax_seg", "confmax_critic"):
            if min(args.height, args.width) < 256:
                break  # the critics' five stride-2 convolutions need >= 32 pixels at 1/8 scale
            m2 = build(wl, dev)
            a2 = optim.arenas_for(m2)
            o2 = optim.FlatRMSprop(a2, lr=1e-4)
            s2 = make_step(wl, m2, o2, batch, None)
            s2()  # priming (kernel-variant lookups, allocator growth)
            others[wl] = dict(timed(s2, args.batch), unit="pairs/s", dtype="f32", arithmetic=ops.CONV_MATH,
                              workload="configs[2]: confusion-maximisation %s iteration (day+night passes, 6 critics)"
                                       % ("train_seg" if wl == "confmax_seg" else "train_critic"))
            if not args.no_roofline:  # the same roofline block as the headline's, for this workload's conv launches
                r2 = measure_conv_roofline(s2, 2, wl, scale=scale, geom=(args.batch, args.height, args.width))
                others[wl]["roofline"] = {k: r2[k] for k in ("bound", "achieved", "peak", "unit", "frac", "conv_ms_per_step",
                                                             "conv_tflop_per_step", "reference_op_tflop_per_step",
                                                             "achieved_vs_reference_op_count", "launches_per_step",
                                                             "traffic", "by_kind")}
            del s2, o2, a2, m2
            torch.cuda.empty_cache()
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload != "seg_resnext":
        base = cpu_baseline(args.workload)

    if rank == 0:
        names = {"seg": "configs[1]: HeatNet RGB+thermal dual-stream PSPNet (ResNet-50, late fusion), forward + CE + "
                        "backward + RMSprop",
                 "seg_bf16": "configs[4] (one GPU's share): the configs[1] step with the mixed-precision bf16 conv MFMA path",
                 "confmax_seg": "configs[2]: confusion-maximisation train_seg iteration (day+night passes, 6 critics)",
                 "confmax_critic": "configs[2]: confusion-maximisation train_critic iteration (day+night passes, 6 critics)",
                 "seg_resnext": "optional branch (SURVEY 8f N4): the configs[1] step on the reference's arch='custom' network "
                                "(ResNeXt-50, 64 groups, + ASPP, late fusion); no CPU baseline"}
        out = {"metric": "training images/s at %dx%d RGB+IR, bs=%d/GPU" % (args.height, args.width, args.batch), "value": round(world * args.batch * args.steps / elapsed, 3),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16" if ops.CONV_MATH == "bf16" else "f32", "arithmetic": ops.CONV_MATH,
               "data": "synthetic", "distributed": comm, "bn_backward_reductions": bn_passes,
               "config": {"workload": names[args.workload], "global_batch": world * args.batch, "height": args.height,
                          "width": args.width,
                          "parallelism": "dp%d" % world, "optimizer": "RMSprop lr=1e-4 (fused, flat arena)", "weights": "random init",
                          "conv_math": CONV_MATH_NOTE[ops.CONV_MATH]},
               "roofline": roof, "cpu_baseline": base, "alt_conv_math": alt,
               "alt_conv_maths": list(alts.values()) or None, "other_workloads": others}
        if force_dist:
            out["rehearsal"] = {"backend": backend, "world_size": 1, "collectives_issued": buckets.launched,
                                "note": "HN_BENCH_FORCE_DIST=1: the N>1 path (process group, bucketed all-reduce on the "
                                        "communication stream, barriers) on one rank"}
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
