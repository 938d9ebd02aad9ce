"""Adam as ONE HIP launch over all parameter tensors (reference: utils/misc.py get_optimizer -> torch.optim.Adam,
lr 1e-4, betas (0.99, 0.999), train.py:127).

torch's capturable Adam on this build falls back to one `div_` kernel per parameter and bias-correction term
(1,268 launches, 5.3 ms per step at 634 tensors).  Here the (tensor, chunk) work list is flattened once; the step count
and the learning rate are device scalars, so the launch is HIP-graph replayable and a scheduler can still change the
rate between steps (`param_groups[0]['lr']` is copied to the device by `sync_hyper()`).
"""
import ctypes

import torch

from . import _capi, _lib

CHUNK = 4096


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.99, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        assert len(self.param_groups) == 1, "one parameter group (as the reference's get_optimizer)"
        self.params = list(self.param_groups[0]["params"])
        self._built = False

    # ------------------------------------------------------------------------------------------------ state
    def _build(self):
        """Called at the first step: only parameters that received a gradient take part (90 tensors never
        do), exactly like torch.optim.Adam, which skips `grad is None`."""
        ps = [p for p in self.params if p.grad is not None]
        dev = ps[0].device
        self.active = ps
        self.exp_avg = [torch.zeros_like(p) for p in ps]
        self.exp_avg_sq = [torch.zeros_like(p) for p in ps]
        self.step_t = torch.zeros(1, device=dev, dtype=torch.float32)
        self.lr_t = torch.full((1,), float(self.param_groups[0]["lr"]), device=dev, dtype=torch.float32)
        i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)
        self.p_ptr = i64([p.data_ptr() for p in ps])
        self.m_ptr = i64([t.data_ptr() for t in self.exp_avg])
        self.v_ptr = i64([t.data_ptr() for t in self.exp_avg_sq])
        self.sizes = i64([p.numel() for p in ps])
        ct, co = [], []
        for i, p in enumerate(ps):
            for off in range(0, p.numel(), CHUNK):
                ct.append(i)
                co.append(off)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=dev)
        self.chunk_off = i64(co)
        self.g_ptr = None
        self._grad_ids = None
        self._pinned = {}                  # grad-address tuple -> device table baked into a captured optimizer graph
        self._scratch_table = None         # the eager steps' table (overwritten in place)
        self._norm_partial = None
        self._built = True
        self.generation = getattr(self, "generation", 0) + 1     # the step engine drops its captures when this changes

    def _grad_table(self, pin=False):
        """Device table of the CURRENT .grad addresses (they change when grads are re-created; static under replay).
        pin=True (the step engine, right before it captures the optimizer graph): the table gets its own tensor that stays
        alive until `unpin` - a captured launch has the address of the table it was captured with baked in.  Otherwise
        (eager steps: gradients are re-created every step and the allocator may hand back new addresses each time) ONE
        scratch table is overwritten in stream order, so nothing accumulates over a long run."""
        ids = tuple(p.grad.data_ptr() for p in self.active)
        if ids != self._grad_ids:
            hit = self._pinned.get(ids)
            if hit is None:
                host = torch.tensor(ids, dtype=torch.int64)
                if pin:
                    hit = self._pinned[ids] = host.to(self.p_ptr.device)
                else:
                    if self._scratch_table is None:
                        self._scratch_table = torch.empty(len(ids), dtype=torch.int64, device=self.p_ptr.device)
                    self._scratch_table.copy_(host)
                    hit = self._scratch_table
            self.g_ptr = hit
            self._grad_ids = ids
        elif pin and ids not in self._pinned:       # the current table is the scratch one: give the capture its own
            self._pinned[ids] = self.g_ptr = torch.tensor(ids, dtype=torch.int64, device=self.p_ptr.device)
        return self.g_ptr

    def unpin(self, table):
        """Forget ONE pinned address table (the capture that used it is gone).  None = a capture that never pinned one:
        nothing to forget (it must not wipe the tables of the live captures - `unpin_all` is the explicit form)."""
        if table is None:
            return
        for k in [k for k, v in self._pinned.items() if v is table]:
            del self._pinned[k]
        self._grad_ids = None

    def unpin_all(self):
        """Forget every pinned address table (all captures are being dropped)."""
        self._pinned.clear()
        self._grad_ids = None

    def sync_hyper(self):
        """Push the (possibly scheduler-modified) learning rate to the device scalar the kernel reads."""
        if self._built:
            self.lr_t.fill_(float(self.param_groups[0]["lr"]))

    @torch.no_grad()
    def step(self, closure=None):
        if not self._built:
            self._build()
        for p in self.active:
            assert p.grad is not None and p.grad.is_contiguous() and p.is_contiguous()
        g_ptr = self._grad_table()
        b1, b2 = self.param_groups[0]["betas"]
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        code = _lib.lib().singa_adam_step(vp(self.p_ptr), vp(g_ptr), vp(self.m_ptr), vp(self.v_ptr), vp(self.sizes),
                                          vp(self.chunk_tensor), vp(self.chunk_off), self.chunk_tensor.numel(), CHUNK,
                                          vp(self.step_t), vp(self.lr_t), b1, b2, self.param_groups[0]["eps"],
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _capi.check(_lib.lib(), code, "singa_adam_step")

    def grad_norm(self):
        """Total 2-norm of the gradients the step will consume (torch.nn.utils.clip_grad_norm_'s return value,
        train.py:126) as a device scalar: one chunked sum-of-squares launch + a fixed-order finish, HIP-graph replayable
        and identical eager / replayed."""
        if not self._built:
            self._build()
        g_ptr = self._grad_table()
        if self._norm_partial is None:
            self._norm_partial = torch.empty(self.chunk_tensor.numel(), device=self.p_ptr.device, dtype=torch.float32)
            self._norm_out = torch.zeros(1, device=self.p_ptr.device, dtype=torch.float32)
        vp = lambda t: ctypes.c_void_p(t.data_ptr())
        code = _lib.lib().singa_grad_norm(vp(g_ptr), vp(self.sizes), vp(self.chunk_tensor), vp(self.chunk_off),
                                          self.chunk_tensor.numel(), CHUNK, vp(self._norm_partial), vp(self._norm_out),
                                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        _capi.check(_lib.lib(), code, "singa_grad_norm")
        return self._norm_out[0]

    # ------------------------------------------------------------------------------------------------ snapshots
    def snapshot(self):
        """Copies of everything a step changes (parameters, both moments, the step count), for TrainStep's warm-up."""
        snap = {"params": [p.detach().clone() for p in self.params], "built": self._built}
        if self._built:
            snap.update(step=self.step_t.clone(), exp_avg=[t.clone() for t in self.exp_avg],
                        exp_avg_sq=[t.clone() for t in self.exp_avg_sq])
        return snap

    @torch.no_grad()
    def restore(self, snap):
        for p, q in zip(self.params, snap["params"]):
            p.copy_(q)
        if not self._built:
            return
        if snap["built"]:
            self.step_t.copy_(snap["step"])
            torch._foreach_copy_(self.exp_avg, snap["exp_avg"])
            torch._foreach_copy_(self.exp_avg_sq, snap["exp_avg_sq"])
        else:                       # the optimizer was built by the warm-up itself: back to its initial state
            self.step_t.zero_()
            torch._foreach_zero_(self.exp_avg)
            torch._foreach_zero_(self.exp_avg_sq)

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim.Adam's own layout (what the reference's checkpoints hold, train.py:244-252 / gen.py:106-110):
        {'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [0..n-1]}]}; only
        parameters that have taken a step have a state entry (torch skips `grad is None` the same way)."""
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        state = {}
        if self._built:
            idx = {id(p): i for i, p in enumerate(self.params)}
            step = self.step_t.detach().cpu().reshape(()).clone()
            if float(step) > 0:
                for p, m, v in zip(self.active, self.exp_avg, self.exp_avg_sq):
                    state[idx[id(p)]] = {"step": step.clone(), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if not (isinstance(sd, dict) and "param_groups" in sd and "state" in sd and len(sd["param_groups"]) == 1):
            raise ValueError("Adam.load_state_dict: expected torch.optim.Adam's state_dict layout "
                             "{'state': {...}, 'param_groups': [one group]}")
        g = sd["param_groups"][0]
        if len(g.get("params", [])) != len(self.params):
            raise ValueError(f"Adam.load_state_dict: the checkpoint's group has {len(g.get('params', []))} parameters, "
                             f"this optimizer has {len(self.params)}")
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise ValueError("Adam.load_state_dict: weight_decay / amsgrad / maximize are not supported (the reference uses none)")
        for k in ("lr", "betas", "eps"):                       # hyper-parameters only; never the 'params' index list
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]
        state = {int(i): st for i, st in sd["state"].items()}
        if not state:
            self._built = False
            self.generation = getattr(self, "generation", 0) + 1
            return
        steps = {float(st["step"]) for st in state.values()}
        if len(steps) != 1:
            raise ValueError("Adam.load_state_dict: parameters with different step counts are not supported")
        same = self._built and {id(p) for p in self.active} == {id(self.params[i]) for i in state}
        if not same:
            # (re)build over exactly the parameters the checkpoint has a state for; the step engine notices the new
            # `generation` and drops captures that still point at the old moment buffers.  NB torch.optim.Adam would create
            # the state of a parameter lazily when it first receives a gradient; here the participating set is fixed by
            # the checkpoint (the 90 gradient-free tensors never change within a run)
            for i in state:
                if self.params[i].grad is None:
                    self.params[i].grad = torch.zeros_like(self.params[i])
            for i, p in enumerate(self.params):
                if i not in state and p.grad is not None:
                    p.grad = None
            self._build()
        self.step_t.fill_(steps.pop())
        idx = {id(p): i for i, p in enumerate(self.params)}
        with torch.no_grad():
            for p, m, v in zip(self.active, self.exp_avg, self.exp_avg_sq):
                st = state[idx[id(p)]]
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
        self.sync_hyper()
