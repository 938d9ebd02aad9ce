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
        self._built = True

    def _grad_table(self):
        """Device table of the CURRENT .grad addresses (they change when grads are re-created; static under replay)."""
        ids = tuple(p.grad.data_ptr() for p in self.active)
        if ids != self._grad_ids:
            self.g_ptr = torch.tensor(ids, dtype=torch.int64, device=self.p_ptr.device)
            self._grad_ids = ids
        return self.g_ptr

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

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        if not self._built:
            return {"built": False, "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}
        idx = {id(p): i for i, p in enumerate(self.params)}
        return {"built": True, "step": float(self.step_t), "active": [idx[id(p)] for p in self.active],
                "exp_avg": [t.clone() for t in self.exp_avg], "exp_avg_sq": [t.clone() for t in self.exp_avg_sq],
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        self.param_groups[0].update(sd["param_groups"][0])
        if not sd.get("built"):
            return
        for i in sd["active"]:
            if self.params[i].grad is None:
                self.params[i].grad = torch.zeros_like(self.params[i])
        self._build()
        self.step_t.fill_(sd["step"])
        for a, b in zip(self.exp_avg, sd["exp_avg"]):
            a.copy_(b)
        for a, b in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            a.copy_(b)
