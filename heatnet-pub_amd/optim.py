"""Flat parameter arenas and the fused optimizer updates of the HeatNet trainers.

The reference steps `torch.optim.RMSprop(all parameters, lr=1e-4)` (cm/train_trgb_segnet_conf.py:270) or
`torch.optim.Adam` (scripts/main.py:159) tensor by tensor.  Here the parameters of one training phase live in ONE flat
fp32 buffer (each `Parameter` is a view that keeps its own strides, so conv weights stay channels-last), their
gradients in a second flat buffer of the same layout, and an optimizer step is a single HBM-bound kernel launch
(`hn_rmsprop_step` / `hn_adam_step`) over the whole range; `zero_grad` is one memset.  The same flat gradient buffer
is what `ddp.GradBuckets` all-reduces, so there is no gather/scatter of gradients either.

Semantics follow torch >= 2.0 as the reference runs today: a parameter group whose parameters do not require grad in
the current phase (`conv_segnet.setPhase`) is skipped entirely -- its `square_avg` does not decay.
"""
import torch

from . import ops


def _dense(p):
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


class ParamArena:
    """Re-homes `params` into one flat buffer (+ one flat gradient buffer); order is kept."""

    def __init__(self, params):
        self.params = list(params)
        if not self.params:
            raise ValueError("ParamArena: empty parameter list")
        dev = self.params[0].device
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32 or not _dense(p):
                raise ValueError("ParamArena: parameters must be dense fp32 tensors on one device")
        # every parameter starts on a 256-B boundary: the conv kernels take their 16-B vector / buffer-load paths only
        # for aligned weight pointers (a 1-element PReLU slope would otherwise misalign everything behind it)
        align = 64
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.flat = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                view = self.flat[off:off + n].as_strided(p.size(), p.stride())
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[off:off + n].as_strided(p.size(), p.stride())
                # ops.grad_slot(): after a zero_grad() the backward kernels write the parameter's first gradient of the
                # step straight into this slot (no temporary, no accumulation add)
                p._hn_grad_slot = (self.grad, off)
                p._hn_slot_used = False
                # gradients produced by anything else (a sliced weight's SliceBackward, stock torch ops) are adopted by
                # autograd as free-standing tensors: move them into the slot as soon as they are accumulated (this hook
                # is registered before any ddp.GradBuckets hook, so buckets only ever see slot-resident gradients)
                frozen = not p.requires_grad  # a phase switch (conv_segnet.setPhase) may enable it later
                if frozen:
                    p.requires_grad_(True)
                p.register_post_accumulate_grad_hook(self._adopt)
                if frozen:
                    p.requires_grad_(False)

    def active(self):
        """True when this arena's parameters take part in the current phase."""
        return self.params[0].requires_grad

    def grad_view(self, p):
        """A view of `p`'s gradient slot (what `p.grad` is while the parameter holds a gradient)."""
        flat, off = p._hn_grad_slot
        return flat[off:off + p.numel()].as_strided(p.size(), p.stride())

    def _adopt(self, p):
        g = p.grad
        flat, off = p._hn_grad_slot
        if g is None or g.data_ptr() == flat.data_ptr() + 4 * off:
            return
        view = flat[off:off + p.numel()].as_strided(p.size(), p.stride())
        view.copy_(g)
        p.grad = view
        p._hn_slot_used = True

    def zero_grad(self):
        """One memset; parameters drop their `.grad` (the next backward re-attaches views of the slots, see
        ops.grad_slot) -- a parameter that receives no gradient in a step therefore reads as zero gradient."""
        self.grad.zero_()
        for p in self.params:
            p.grad = None
            p._hn_slot_used = False

    def check(self):
        """Every gradient that exists must live in its arena slot."""
        base = self.grad.data_ptr()
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is not None and g.data_ptr() != base + 4 * off:
                raise RuntimeError("ParamArena: a gradient left the arena (a foreign zero_grad(set_to_none=True), or a "
                                   "gradient tensor autograd could not adopt in place)")
            if g is None and p._hn_slot_used:
                # a backward kernel wrote this slot, but the parameter's `.grad` was dropped afterwards (e.g.
                # nn.Module.zero_grad() between backward and step): stepping now would apply a gradient the caller
                # believes cleared
                raise RuntimeError("ParamArena: a parameter's gradient was dropped after backward wrote its arena slot "
                                   "(clear gradients with the optimizer's zero_grad(), before the forward pass)")


class _FlatOptimizer(torch.optim.Optimizer):
    """Common part: one param group per arena (so LR schedulers and state_dict-style access keep working).

    `state_dict()` / `load_state_dict()` speak torch.optim's format -- per-parameter entries `step` plus the buffers
    named in `_BUFFERS` -- so that optimizer states move both ways between this class and the reference's
    `torch.optim.RMSprop` / `Adam` checkpoints (cm/train_trgb_segnet_conf.py:276-283,648-654).  Parameters are matched by
    their position in the concatenation of all param groups, whatever the grouping of the file."""

    _BUFFERS = ()

    def __init__(self, arenas, defaults):
        if isinstance(arenas, ParamArena):
            arenas = [arenas]
        self.arenas = list(arenas)
        groups = [{"params": a.params} for a in self.arenas]
        super().__init__(groups, defaults)  # param_groups[i] <-> arenas[i]
        self.steps = [0 for _ in self.arenas]

    def zero_grad(self, set_to_none=False):
        """One memset per arena; `set_to_none` is accepted for API compatibility and ignored (gradients are detached
        from the parameters either way and re-attached, in place, by the next backward)."""
        for a in self.arenas:
            if a.active():
                a.zero_grad()

    def _flat_buffers(self, name):
        return getattr(self, name)  # list of flat tensors, one per arena

    def state_dict(self):
        state, groups, idx = {}, [], 0
        for gi, (g, a) in enumerate(zip(self.param_groups, self.arenas)):
            ids = []
            for p, off in zip(a.params, a.offsets):
                if self.steps[gi] > 0:  # torch creates a parameter's state at its first step
                    entry = {"step": torch.tensor(float(self.steps[gi]))}
                    for name in self._BUFFERS:
                        flat = self._flat_buffers(name)[gi]
                        entry[name] = flat[off:off + p.numel()].as_strided(p.size(), p.stride()).clone()
                    state[idx] = entry
                ids.append(idx)
                idx += 1
            groups.append(dict({k: v for k, v in g.items() if k != "params"}, params=ids))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        saved_ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        own = [(gi, p, off) for gi, a in enumerate(self.arenas) for p, off in zip(a.params, a.offsets)]
        if len(saved_ids) != len(own):
            raise ValueError("optimizer state has %d parameters, this optimizer %d" % (len(saved_ids), len(own)))
        steps = [0 for _ in self.arenas]
        with torch.no_grad():
            for name in self._BUFFERS:
                for flat in self._flat_buffers(name):
                    flat.zero_()
            for sid, (gi, p, off) in zip(saved_ids, own):
                entry = state_dict["state"].get(sid)
                if entry is None:
                    continue
                steps[gi] = max(steps[gi], int(float(entry.get("step", 0))))
                for name in self._BUFFERS:
                    t = entry[name]
                    if tuple(t.shape) != tuple(p.shape):
                        raise ValueError("optimizer state %s of parameter %d has shape %s, expected %s"
                                         % (name, sid, tuple(t.shape), tuple(p.shape)))
                    flat = self._flat_buffers(name)[gi]
                    flat[off:off + p.numel()].as_strided(p.size(), p.stride()).copy_(t)
        self.steps = steps
        # hyper-parameters: a single saved group (the reference's optimizer) applies to every arena
        saved_groups = state_dict["param_groups"]
        for gi, g in enumerate(self.param_groups):
            src = saved_groups[gi] if len(saved_groups) == len(self.param_groups) else saved_groups[0]
            for k, v in src.items():
                if k != "params" and k in g:
                    g[k] = v


class FlatRMSprop(_FlatOptimizer):
    """torch.optim.RMSprop(lr, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False) on arenas."""

    _BUFFERS = ("square_avg",)

    def __init__(self, arenas, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(arenas, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay, momentum=0, centered=False))
        self.square_avg = [torch.zeros_like(a.flat) for a in self.arenas]

    @torch.no_grad()
    def step(self, closure=None):
        for i, (g, a, sq) in enumerate(zip(self.param_groups, self.arenas, self.square_avg)):
            if not a.active():
                continue
            self.steps[i] += 1
            a.check()
            ops.rmsprop_step_(a.flat, a.grad, sq, g["lr"], g["alpha"], g["eps"], g["weight_decay"])
        ops.weights_changed()  # the update goes through raw pointers (no version bump): drop prepared weight operands


class FlatAdam(_FlatOptimizer):
    """torch.optim.Adam(lr, betas, eps, weight_decay) on arenas (no amsgrad)."""

    _BUFFERS = ("exp_avg", "exp_avg_sq")

    def __init__(self, arenas, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(arenas, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self.exp_avg = [torch.zeros_like(a.flat) for a in self.arenas]
        self.exp_avg_sq = [torch.zeros_like(a.flat) for a in self.arenas]

    @torch.no_grad()
    def step(self, closure=None):
        for i, (g, a) in enumerate(zip(self.param_groups, self.arenas)):
            if not a.active():
                continue
            a.check()
            self.steps[i] += 1
            ops.adam_step_(a.flat, a.grad, self.exp_avg[i], self.exp_avg_sq[i], self.steps[i], g["lr"], g["betas"][0],
                           g["betas"][1], g["eps"], g["weight_decay"])
        ops.weights_changed()  # see FlatRMSprop.step


def arenas_for(model):
    """One arena per training phase of a `conv_segnet` (segmentation net, critics), or one for a plain network."""
    if hasattr(model, "trgb_segnet"):
        groups = [list(model.trgb_segnet.parameters())]
        if hasattr(model, "critics"):
            groups.append(list(model.critics.parameters()))
        if hasattr(model, "downscale_nets"):  # feedback_seg: never frozen by setPhase (cm/models/conf_segnet.py:86-104)
            groups.append(list(model.downscale_nets.parameters()))
        return [ParamArena(g) for g in groups]
    return [ParamArena(list(model.parameters()))]
