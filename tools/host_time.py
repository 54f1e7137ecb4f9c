"""How long the host takes to ENQUEUE one training step, against the step's wall time: python tools/host_time.py [workload]
(the GPU runs behind the host; if the two numbers meet, the step is launch-bound and kernel work no longer decides it)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "seg"
    from heatnet_amd import ops, optim
    if workload == "seg_bf16":
        ops.set_conv_math("bf16")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    buckets = None
    dist_mode = os.environ.get("HN_HOST_TIME_DIST") == "1"  # the N > 1 path on one rank (process group, buckets, hooks)
    if dist_mode:
        import socket
        import torch.distributed as dist
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sock.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        sock.close()
        dist.init_process_group("nccl", device_id=dev)
    model = bench.build(workload, dev)
    arenas = optim.arenas_for(model)
    opt = optim.FlatRMSprop(arenas, lr=1e-4)
    if dist_mode:
        from heatnet_amd import ddp
        buckets = ddp.GradBuckets(arenas=arenas, always_exchange=True)
    batch = bench.synthetic_batch(0, dev)
    step = bench.make_step(workload, model, opt, batch, buckets)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n, host = 20, 0.0
    t_all = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        host += time.perf_counter() - t0
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    if dist_mode:
        workload += " (N > 1 path)"
    print("%s: host enqueue %.2f ms per step (loop %.2f), wall %.2f ms per step" % (workload, 1e3 * host / n, 1e3 * t_enq / n, 1e3 * wall / n))


if __name__ == "__main__":
    main()
