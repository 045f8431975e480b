"""`FusedAdam`: torch.optim.Adam semantics (no weight decay, no amsgrad) with one fused HIP launch for all parameters
(csrc/optim.hip).  Drop-in for `torch.optim.Adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)` of the
reference (main_nerf.py:211); param_groups / lr schedulers / state_dict work as for any torch optimizer.
Parameters that are not CUDA fp32 fall back to the same update written in torch ops."""
import ctypes
import math

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._group_of = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_now()
        return loss

    @torch.no_grad()
    def step_grid_table(self, p, offsets, level_dim, extra=()):
        """Adam on a hash-grid table whose backward left its gradient as record lists (grid_encode_backward(defer=True),
        csrc/gridencoder.hip: k_grid_tile_adam): one pass over the table sums each 64-KiB tile's records in LDS and
        updates the tile's rows of p / m / v from there -- the gradient table is neither written nor read nor cleared
        for the binned levels.  p.grad (dense, zero-initialised, kept across steps) receives what was not binned and
        comes back cleared.  `extra`: up to 8 small fp32 parameters with dense gradients (the MLP weights), updated by
        the same launch.  Same update as step_now on the summed gradient."""
        extra = [q for q in extra if q.grad is not None]
        if len(extra) > 8 or any(not (q.is_cuda and q.dtype == torch.float32 and q.is_contiguous()
                                      and q.grad.is_contiguous() and q.grad.dtype == torch.float32) for q in extra):
            if L.lib().enerf_amp_armed():                   # step_now neither unscales nor skips
                raise RuntimeError("FusedAdam.step_grid_table: under the device-side GradScaler (enerf_amp_begin) every extra "
                                   "tensor must ride in the table launch: at most 8 contiguous fp32 CUDA tensors")
            self.step_now(only=extra)                       # (more / other tensors than the launch carries)
            extra = []
        group, st, args = self.grid_table_args(p, extra, [q.grad for q in extra])
        b1, b2 = group["betas"]
        L.check(L.lib().enerf_grid_adam_from_records_ex(
            p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), offsets.data_ptr(),
            offsets.numel() - 1, int(level_dim), float(group["lr"]), b1, b2, float(group["eps"]), st["step"], *args,
            L.stream_handle()), "grid_adam_from_records")

    def grid_table_args(self, p, extra, extra_grads):
        """Bookkeeping of one step_grid_table: the step counts of the table `p` and of the `extra` small tensors advance,
        and -> (the table's param group, its state, the (n_small, p, g, m, v, n, lr, step) host arrays of
        enerf_grid_adam_from_records_ex).  `extra_grads`: the tensors the small gradients live in (they need not be the
        parameters' .grad yet: the native step hands out the buffers its backward is about to write)."""
        def state_of(q):
            st = self.state[q]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(q, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(q, memory_format=torch.preserve_format)
            st["step"] += 1
            return st

        groups = self._group_of
        if groups is None:
            groups = self._group_of = {id(q): g for g in self.param_groups for q in g["params"]}
        group = groups[id(p)]
        st = state_of(p)
        n = len(extra)
        if n:
            sts = [state_of(q) for q in extra]
            vp, u32, fl = ctypes.c_void_p * n, ctypes.c_uint32 * n, ctypes.c_float * n
            args = (n, vp(*[q.data_ptr() for q in extra]), vp(*[g.data_ptr() for g in extra_grads]),
                    vp(*[t["exp_avg"].data_ptr() for t in sts]), vp(*[t["exp_avg_sq"].data_ptr() for t in sts]),
                    u32(*[q.numel() for q in extra]), fl(*[float(groups[id(q)]["lr"]) for q in extra]),
                    u32(*[t["step"] for t in sts]))
        else:
            args = (0, None, None, None, None, None, None, None)
        return group, st, args

    def grid_table_plan(self, p, extra, extra_grads):
        """grid_table_args for a loop that steps the same tensors every time: the host arrays are built ONCE; the returned
        callable advances the step counts, refreshes the learning rates (a scheduler may have moved them) and -> (lr,
        beta1, beta2, eps, table step) -- the arrays behind `.arrays` then hold this step's values."""
        group, st, args = self.grid_table_args(p, extra, extra_grads)
        for q in [p] + list(extra):                       # (the planning call advanced them once: undo)
            self.state[q]["step"] -= 1
        groups = self._group_of
        sts = [self.state[q] for q in extra]
        egroups = [groups[id(q)] for q in extra]
        n, lr_arr, step_arr = args[0], args[6], args[7]

        def advance():
            st["step"] += 1
            for k in range(n):
                t = sts[k]
                t["step"] += 1
                step_arr[k] = t["step"]
                lr_arr[k] = float(egroups[k]["lr"])
            b1, b2 = group["betas"]
            return float(group["lr"]), b1, b2, float(group["eps"]), st["step"]
        advance.arrays = args
        advance.state = st
        return advance

    @torch.no_grad()
    def step_now(self, only=None, ranges=None, zero_grads=False, advance=None):
        """The update itself.  `step()` is wrapped by torch.optim.Optimizer with profiler / hook plumbing that costs
        ~40 us per call; a training loop that needs neither can call this directly.

        only   : restrict the update to these parameters (a data-parallel loop updates each bucket as soon as its
                 all-reduce has landed).  Every parameter must be stepped exactly once per optimisation step.
        ranges : {parameter: (lo, hi)} -- update only elements [lo, hi) of that parameter; the step count advances on
                 the range starting at 0 (a loop that walks all ranges), or on every call with advance=True (a rank
                 that only ever updates its own slice).
        zero_grads : the fused kernel also clears the gradients it has just read (same pass, no extra launch): a
                 loop that keeps its gradient buffers can skip the next step's zero-fill."""
        batches = {}          # (beta1, beta2, eps) -> parameters the fused kernel takes, all in one launch
        only = None if only is None else {id(p) for p in only}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None or (only is not None and id(p) not in only):
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                lo, hi = (0, p.numel()) if ranges is None or p not in ranges else ranges[p]
                if lo == 0 if advance is None else advance:
                    st["step"] += 1
                g = p.grad
                if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() \
                        and g.dtype == torch.float32 and p.numel() > 0:
                    if (lo, hi) != (0, p.numel()):
                        if lo % 4 or hi <= lo:
                            raise ValueError("FusedAdam ranges must start on a multiple of 4 elements")
                        batches.setdefault((b1, b2, eps), []).append((p.view(-1)[lo:hi], g.view(-1)[lo:hi], dict(
                            step=st["step"], exp_avg=st["exp_avg"].view(-1)[lo:hi],
                            exp_avg_sq=st["exp_avg_sq"].view(-1)[lo:hi]), lr))
                        continue
                    batches.setdefault((b1, b2, eps), []).append((p, g, st, lr))
                else:
                    # host tensors / other dtypes: the same update in torch ops, on the same element range
                    t = st["step"]
                    pv, gv = p.view(-1)[lo:hi], g.reshape(-1)[lo:hi]
                    m, v = st["exp_avg"].view(-1)[lo:hi], st["exp_avg_sq"].view(-1)[lo:hi]
                    m.lerp_(gv, 1 - b1)
                    v.mul_(b2).addcmul_(gv, gv, value=1 - b2)
                    denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
                    pv.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
                    if zero_grads:
                        g.view(-1)[lo:hi].zero_()
        for (b1, b2, eps), items in batches.items():
            for k in range(0, len(items), 16):
                chunk = items[k:k + 16]
                n = len(chunk)
                vp, sz, fl, u32 = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_float * n, ctypes.c_uint32 * n
                L.check(L.lib().enerf_adam_step_multi(
                    n, vp(*[p.data_ptr() for p, _, _, _ in chunk]), vp(*[g.data_ptr() for _, g, _, _ in chunk]),
                    vp(*[st["exp_avg"].data_ptr() for _, _, st, _ in chunk]),
                    vp(*[st["exp_avg_sq"].data_ptr() for _, _, st, _ in chunk]),
                    sz(*[p.numel() for p, _, _, _ in chunk]), fl(*[lr for _, _, _, lr in chunk]),
                    u32(*[st["step"] for _, _, st, _ in chunk]), b1, b2, eps, int(bool(zero_grads)),
                    L.stream_handle()), "adam_step_multi")
