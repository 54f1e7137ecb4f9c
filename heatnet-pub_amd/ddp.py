"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI overlapped with backward.

Replaces what the reference gets from `nn.DataParallel` / `DistributedDataParallel`
(scripts/main.py:103-123, models/confusion_maximization/train_trgb_segnet_conf.py:234): every rank holds a full
replica, BatchNorm statistics stay per-replica (the reference never uses SyncBN), and the mean-reduced loss makes
"average the gradients over ranks" equal to the reference's single large batch.

Design for MI355X: gradients live as views into a few large flat fp32 buckets that are allocated once per
training phase (train_seg: the 54 M segnet parameters, train_critic: the 21 M critic parameters), filled in
reverse-registration order (~ the order backward produces them).  A post-accumulate-grad hook counts arrivals;
when a bucket is complete its all-reduce is issued on a dedicated communication stream behind an event, so the
collective runs while backward continues.  xGMI is point-to-point, so few large messages (default 64 MiB) beat
many small ones.  No data-path collective other than this one exists.
"""
import torch
import torch.distributed as dist


class GradBuckets:
    """`params`: the buckets allocate their own flat gradient storage.  `arenas` (optim.ParamArena): the buckets are
    contiguous slices of the arenas' flat gradient buffers -- the optimizer then reads the reduced gradients in place.

    Training phases (conv_segnet.setPhase flips `requires_grad` between the segmentation net and the critics every
    50 / 500 iterations, cm/train_trgb_segnet_conf.py:577-592): buckets and hooks are built ONCE for every arena, and
    `begin_step()` arms exactly the buckets whose arena takes part in the current phase -- nothing has to be rebuilt
    at a phase switch, and a frozen arena's stale gradients are never exchanged.  With plain `params` there is one
    fixed set: `begin_step()` raises if the set of trainable parameters no longer matches it.

    `always_exchange`: issue the collectives at world size 1 as well (an all-reduce over one rank leaves the gradients
    unchanged) -- the rehearsal of the RCCL path, its communication stream and its event ordering on a one-GPU box."""

    def __init__(self, params=None, bucket_bytes=64 << 20, process_group=None, average=True, arenas=None,
                 always_exchange=False, tail_bytes=8 << 20):
        """`tail_bytes`: size of the bucket that holds the FIRST-registered parameters (the stems, layer1): backward reaches
        them last, so their all-reduce is the one nothing can hide behind -- it is kept short, the others large."""
        tail_bytes = min(tail_bytes, bucket_bytes)
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.exchange = self.world > 1 or (always_exchange and dist.is_initialized())
        self.launched = 0  # collectives issued so far (diagnostics / tests)
        self.launched_in_backward = 0  # ... of the current step by the time finish_step() was entered
        self._step_launched = 0
        self._exposed = []  # per step: (start, end) events on the compute stream around the wait for the collectives
        self._exposed_cpu = []  # CPU tensors: wall seconds spent waiting in finish_step()
        self.buckets = []  # dicts: flat, params, pending, work
        self.arenas = list(arenas) if arenas is not None else None
        if arenas is not None:
            self.params = [p for a in arenas for p in a.params]
        else:
            self._all_params = list(params)
            self.params = [p for p in self._all_params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBuckets: no trainable parameters")
        self.device = self.params[0].device
        self.on_gpu = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        if arenas is not None:
            for a in arenas:
                start, cur = 0, []
                ends = a.offsets[1:] + [a.numel]  # parameter slots are padded to 256 B; padding stays zero
                for p, end in zip(a.params, ends):
                    cur.append(p)
                    if (end - start) * 4 >= (tail_bytes if start == 0 else bucket_bytes):
                        self._adopt(a.grad[start:end], cur, a)
                        start, cur = end, []
                if cur:
                    self._adopt(a.grad[start:a.numel], cur, a)
        else:
            # cut in registration order (first bucket short, see tail_bytes), fill in reverse: backward reaches the
            # last-registered parameters first
            groups, cur, cur_bytes = [], [], 0
            for p in self.params:
                cur.append(p)
                cur_bytes += p.numel() * 4
                if cur_bytes >= (tail_bytes if not groups else bucket_bytes):
                    groups.append(cur)
                    cur, cur_bytes = [], 0
            if cur:
                groups.append(cur)
            for g in reversed(groups):
                self._seal(list(reversed(g)))
        self._hooks = []
        self._index = {}
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._index[id(p)] = bi
                # hooks can only be registered on parameters that require grad: flip a frozen phase's parameters for
                # the registration (as optim.ParamArena does)
                frozen = not p.requires_grad
                if frozen:
                    p.requires_grad_(True)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                if frozen:
                    p.requires_grad_(False)
        self._active = False

    def _adopt(self, flat, plist, arena):
        self.buckets.append({"flat": flat, "params": list(plist), "pending": 0, "work": None, "arena": arena,
                             "armed": False})

    def _seal(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, device=self.device, dtype=torch.float32)
        off = 0
        for p in plist:
            # the view keeps the parameter's own memory layout (channels-last conv weights stay channels-last)
            view = flat[off:off + p.numel()].as_strided(p.size(), p.stride())
            p.grad = view
            off += p.numel()
        self.buckets.append({"flat": flat, "params": list(plist), "pending": 0, "work": None, "arena": None,
                             "armed": False})

    # -- per-step protocol -------------------------------------------------------------------------------
    def begin_step(self):
        """Call after optimizer.zero_grad(set_to_none=False): arms the arrival counters of the buckets that take part in
        the current training phase."""
        if self.arenas is None:
            for p in self._all_params:
                if p.requires_grad and id(p) not in self._index:
                    raise RuntimeError("GradBuckets: a parameter became trainable after the buckets were built (build "
                                       "them from optim arenas, or rebuild them at a phase switch)")
        any_armed = False
        for b in self.buckets:
            a = b["arena"]
            b["work"] = None
            if a is not None:
                state = [p.requires_grad for p in b["params"]]
                if any(state) != all(state) or state[0] != a.active():
                    raise RuntimeError("GradBuckets: requires_grad differs inside one arena (phases are per arena)")
                b["armed"] = a.active()
            else:
                if not all(p.requires_grad for p in b["params"]):
                    raise RuntimeError("GradBuckets: a bucketed parameter was frozen (build the buckets from optim "
                                       "arenas, or rebuild them at a phase switch)")
                b["armed"] = True
            if not b["armed"]:
                b["pending"] = -1
                continue
            any_armed = True
            b["pending"] = len(b["params"])
            for p in b["params"]:
                if p.grad is None and hasattr(p, "_hn_grad_slot"):
                    continue  # arena parameter after zero_grad(): backward re-attaches a view of its slot
                if p.grad is None or p.grad.data_ptr() < b["flat"].data_ptr() or \
                        p.grad.data_ptr() >= b["flat"].data_ptr() + b["flat"].numel() * 4:
                    raise RuntimeError("GradBuckets: a gradient left its bucket (use zero_grad(set_to_none=False))")
        if not any_armed:
            raise RuntimeError("GradBuckets: no trainable parameters in this phase")
        self._active = True
        self._step_launched = 0

    def _on_grad(self, p):
        if not self._active:
            return
        b = self.buckets[self._index[id(p)]]
        if not b["armed"]:
            raise RuntimeError("GradBuckets: a gradient arrived for a parameter of a frozen phase")
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def _launch(self, b):
        self._step_launched += 1  # (counted at world size 1 too: the hook protocol is what tests inspect)
        if not self.exchange:
            return
        self.launched += 1
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                from . import ops
                if ops.side_stream_used.get(self.device):  # weight gradients may still be running on the side stream
                    self.comm_stream.wait_stream(ops._side_stream(self.device))
                main = ops.dual_stream_used.get(self.device)
                if main is not None:
                    # the bucket's gradients come from BOTH encoder streams (ops.DUAL_STREAMS); this hook runs on one of them
                    self.comm_stream.wait_stream(ops.dual_stream(self.device))
                    self.comm_stream.wait_stream(main)
                if self.average:
                    b["flat"].mul_(1.0 / self.world)
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if self.average:
                b["flat"].mul_(1.0 / self.world)
            b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish_step(self):
        """Call after backward: launches buckets that never filled (unused parameters keep zero gradients) and makes
        the compute stream wait for every collective before the optimizer reads the gradients."""
        self._active = False
        self.launched_in_backward = self._step_launched
        for b in self.buckets:
            if b["armed"] and b["pending"] != 0 and b["work"] is None:
                b["pending"] = 0
                self._launch(b)
        # exposed communication = what the compute stream idles here: events around the wait (GPU), wall time (CPU)
        ev0 = ev1 = None
        t0 = None
        if self.exchange:
            if self.on_gpu:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
            else:
                import time
                t0 = time.perf_counter()
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
        if self.on_gpu and self.exchange:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            ev1.record(torch.cuda.current_stream(self.device))
            self._exposed.append((ev0, ev1))
            del self._exposed[:-256]
        elif t0 is not None:
            import time
            self._exposed_cpu.append(time.perf_counter() - t0)
            del self._exposed_cpu[:-256]

    def exposed_ms(self, last=None):
        """Mean time per step the compute stream spent waiting for gradient collectives in finish_step() (the communication
        backward did not hide), over the last `last` steps (default: all recorded, at most 256).  Synchronises."""
        if self.on_gpu:
            pairs = self._exposed[-last:] if last else self._exposed
            if not pairs:
                return 0.0
            torch.cuda.synchronize(self.device)
            return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        vals = self._exposed_cpu[-last:] if last else self._exposed_cpu
        return 1e3 * sum(vals) / len(vals) if vals else 0.0

    def reset_exposed(self):
        self._exposed, self._exposed_cpu = [], []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_module_state(module, src=0, process_group=None):
    """Rank `src`'s parameters and buffers to every rank (what DDP does at construction, and what keeps the
    DataParallel convention that replica 0's BatchNorm running statistics are the ones that persist)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.is_contiguous():
                dist.broadcast(t, src, group=process_group)
            else:
                tmp = t.contiguous()
                dist.broadcast(tmp, src, group=process_group)
                t.copy_(tmp)
    from . import ops
    ops.weights_changed()  # in case the caller holds a frozen_weights() scope open
