"""optim.step() of the BPR train loop on the HIP library: `Adam` is torch.optim.Adam (same constructor, same state_dict --
`step`, `exp_avg`, `exp_avg_sq` per parameter, so ngacf_amd/checkpoint.py and the reference's .pkl files interchange) whose
step() is ONE ngacf_adam_step launch over all parameters instead of torch's multi-tensor kernels (49 us -> see DESIGN.md
section 3.6 at Gowalla scale).  Reference: `optim = Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)`
run_Gowalla.py:114, `optim.step()` train_eval_Gowalla.py:138."""
import ctypes

import torch

from . import _lib, ops


class Adam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) for fp32 CUDA parameters; the step counters live on the device
    (`capturable`), so step() can be captured in a hipGraph.  amsgrad / maximize / tensor lr / sparse gradients: not on this
    path (ValueError)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kw):
        for k in ("amsgrad", "maximize", "differentiable"):
            if kw.pop(k, False):
                raise ValueError("ngacf_amd.optim.Adam: %s is not supported" % k)
        for k in ("foreach", "fused", "capturable"):
            kw.pop(k, None)
        if kw:
            raise ValueError("ngacf_amd.optim.Adam: unknown arguments %s" % sorted(kw))
        if torch.is_tensor(lr):
            raise ValueError("ngacf_amd.optim.Adam: lr must be a float")
        # capturable=True makes torch keep `step` as a device tensor (also when a state_dict is loaded)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True, foreach=False, fused=False)

    def load_state_dict(self, state_dict):
        """Accepts the state of any torch.optim.Adam (the reference's checkpoints included); the step counters move to the device."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group["capturable"], group["foreach"], group["fused"] = True, False, False
        for p, st in self.state.items():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].to(device=p.device, dtype=torch.float32)

    supports_grad_pairs = True

    def step_counters(self):
        """The device step counters of every parameter that takes gradients (state created if need be): a caller that
        increments them itself -- ops.tick, together with its other per-step counters -- passes steps_incremented=True to step()."""
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.requires_grad:
                    out.append(self._state_of(p)["step"])
        return out

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if st["step"].device != p.device or st["step"].dtype != torch.float32:      # a state loaded from a host-step Adam
            st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
        return st

    @torch.no_grad()
    def step(self, closure=None, grad_pairs=None, steps_incremented=False):
        """grad_pairs: {parameter: (g_a, g_b)} -- the parameter's gradient is g_a + g_b (g_b may be None), summed inside the
        step's kernel; parameters not in it use .grad as usual.  (GraphedBPRStep hands in the gradients of its two
        propagations this way instead of letting autograd accumulate them.)"""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, ms, vs, steps = [], [], [], [], []
            g2s = []
            for p in group["params"]:
                ga, gb = grad_pairs[p] if grad_pairs is not None and p in grad_pairs else (p.grad, None)
                if ga is None and gb is not None:
                    ga, gb = gb, None
                if ga is None:
                    if steps_incremented and p.requires_grad:
                        raise RuntimeError("a parameter whose step counter was incremented up front got no gradient")
                    continue
                if ga.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                    raise ValueError("ngacf_amd.optim.Adam: contiguous fp32 CUDA parameters with dense gradients only")
                st = self._state_of(p)
                ps.append(p); gs.append(ga.contiguous()); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"]); steps.append(st["step"])
                g2s.append(None if gb is None else gb.contiguous())
            if not ps:
                continue
            if not steps_incremented:
                torch._foreach_add_(steps, 1.0)
            b1, b2 = group["betas"]
            for lo in range(0, len(ps), _lib.ADAM_MAX_TENSORS):
                hi = lo + _lib.ADAM_MAX_TENSORS
                ops.adam_step(ps[lo:hi], gs[lo:hi], ms[lo:hi], vs[lo:hi], steps[lo:hi], group["lr"], b1, b2, group["eps"], group["weight_decay"],
                              grads2=g2s[lo:hi])
        return loss
