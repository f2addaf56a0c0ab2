"""Fused Adam for flat network parameters (``ndq_adam_step``, include/ndq.h).

The reference's default optimiser is ``torch.optim.Adam(params)`` stepped once per epoch (solvers.py:182,331-341).
:class:`FusedAdam` IS a ``torch.optim.Adam`` (same hyper-parameters, same ``state_dict`` layout: per-parameter
``step`` / ``exp_avg`` / ``exp_avg_sq``) whose state tensors are views of flat buffers once a solver has bound its
:class:`~neurodiffeq_amd.networks.FlatParams`; ``step()`` then costs one kernel launch per network instead of a
foreach chain.  Parameters that are not bound fall back to the stock implementation."""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._bound = []            # [(FlatParams, exp_avg_flat, exp_avg_sq_flat, group)]
        self._bound_ids = set()
        self._steps = {}
        self._dirty_steps = False

    # pickling / deepcopy (checkpoints: callbacks.py:129-155, solvers_utils.py:281-398): torch serialises defaults,
    # state and param_groups only, so the per-parameter ``step`` tensors are brought up to date first and the
    # binding to a solver's flat buffers -- device pointers -- is dropped; the next solver binds afresh and adopts the
    # moments / step counts from ``state``.
    def __getstate__(self):
        self._sync_step_tensors()
        return super().__getstate__()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._bound, self._bound_ids, self._steps, self._dirty_steps = [], set(), {}, False

    def load_state_dict(self, state_dict):
        """Loaded moments / step counts live in fresh tensors: drop the binding, the solver binds again before its next
        native epoch and ``bind`` adopts them."""
        super().load_state_dict(state_dict)
        self._bound, self._bound_ids, self._steps, self._dirty_steps = [], set(), {}, False

    def bind(self, flat_params):
        """Adopt the flat buffers of ``flat_params`` (list of FlatParams) for every parameter this optimiser owns."""
        if [id(fp) for fp, *_ in self._bound] == [id(fp) for fp in flat_params]:
            return
        self._sync_step_tensors()              # a re-bind (system rebuilt mid-training) adopts the CURRENT step counts
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        self._bound, self._bound_ids = [], set()
        for fp in flat_params:
            groups = {id(group_of.get(id(p))) for p in fp.params}
            if len(groups) != 1 or None in [group_of.get(id(p)) for p in fp.params]:
                continue                        # not (entirely) ours, or split across groups: leave to torch
            group = group_of[id(fp.params[0])]
            m = torch.zeros_like(fp.grad)
            v = torch.zeros_like(fp.grad)
            steps = set()
            for p, off in zip(fp.params, fp._offsets):
                st = self.state[p]
                mv, vv = m[off:off + p.numel()].view(p.shape), v[off:off + p.numel()].view(p.shape)
                if "exp_avg" in st:
                    mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                    steps.add(int(st["step"]))
                else:
                    steps.add(0)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
            if len(steps) != 1:
                continue
            self._steps[id(fp)] = steps.pop()
            self._bound.append((fp, m, v, group))
            self._bound_ids.update(id(p) for p in fp.params)

    def bound(self, fp):
        return any(f is fp and not g.get("amsgrad") and not g.get("maximize") for f, _, _, g in self._bound)

    def fast_slot(self, fp):
        """(exp_avg, exp_avg_sq, group, step AFTER the next update) for the solver's native epoch path, which performs
        the Adam update on the device itself (ndq_epoch_tail); None if ``fp`` is not bound / not eligible."""
        for f, m, v, group in self._bound:
            if f is fp and not group.get("amsgrad") and not group.get("maximize"):
                self._steps[id(fp)] += 1
                self._dirty_steps = True
                return m, v, group, self._steps[id(fp)]
        return None

    def fast_slots(self, fp, k):
        """``fast_slot`` for ``k`` consecutive updates performed by one native call (ndq_fused_fit_run): the step count
        returned is the one AFTER the first of them; the counter advances by ``k``."""
        for f, m, v, group in self._bound:
            if f is fp and not group.get("amsgrad") and not group.get("maximize"):
                first = self._steps[id(fp)] + 1
                self._steps[id(fp)] += k
                self._dirty_steps = True
                return m, v, group, first
        return None

    def _sync_step_tensors(self):
        if getattr(self, "_dirty_steps", False):
            for fp, _, _, _ in self._bound:
                for p in fp.params:
                    self.state[p]["step"].fill_(float(self._steps[id(fp)]))
            self._dirty_steps = False

    def state_dict(self):
        self._sync_step_tensors()
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        self._sync_step_tensors()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        fused = set()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream) if self._bound else None
        for fp, m, v, group in self._bound:
            if not (fp._is_flat() and fp.grads_attached()) or group.get("amsgrad") or group.get("maximize"):
                continue
            self._steps[id(fp)] += 1
            step = self._steps[id(fp)]
            b1, b2 = group["betas"]
            f64 = fp.grad.dtype == torch.float64          # fp64 systems: the same kernel in double (libndq64.so)
            adam = _lib.lib64().ndq64_adam_step if f64 else _lib.lib().ndq_adam_step
            rc = adam(fp.flat.data_ptr(), fp.grad.data_ptr(), m.data_ptr(), v.data_ptr(), fp.numel,
                      float(group["lr"]), b1, b2, group["eps"], group["weight_decay"], step, stream)
            _lib.check(rc, "ndq_adam_step")
            for p in fp.params:
                self.state[p]["step"] += 1
            fused.update(id(p) for p in fp.params)
        if len(fused) < sum(len(g["params"]) for g in self.param_groups):
            # anything not handled above: stock Adam on the remaining parameters
            saved = [g["params"] for g in self.param_groups]
            try:
                for g in self.param_groups:
                    g["params"] = [p for p in g["params"] if id(p) not in fused]
                super().step()
            finally:
                for g, ps in zip(self.param_groups, saved):
                    g["params"] = ps
        return loss
