import os, sys, ctypes, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__; __graft_entry__.build()
from singa_amd import _lib, ops
torch.manual_seed(0)
B, heads, T, S = 32, 4, 201, 230
BH = B * heads
q, k, v = torch.randn(BH, T, 32, device="cuda"), torch.randn(BH, S, 32, device="cuda"), torch.randn(BH, S, 64, device="cuda")
mask = (torch.rand(B, 1, S, device="cuda") < 0.2).expand(B, T, S)
scale = 1 / math.sqrt(32)
def ref():
    p = ops.masked_softmax(ops.bmm_small(q, k.transpose(1, 2)), mask, scale, heads)
    return ops.bmm_small(p, v)
want = ref()
ctx, lse = torch.full((BH, T, 64), float("nan"), device="cuda"), torch.empty(BH, T, 2, device="cuda")
lib = _lib.lib(); p = lambda t: ctypes.c_void_p(t.data_ptr()); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    rc = lib.singa_attn_fwd(p(q), p(k), p(v), p(mask), mask.stride(0), 0, p(ctx), p(lse), BH, T, S, heads, 32, 64, 0, 0, 0, 0, scale, st); assert rc == 0
run(); torch.cuda.synchronize()
print("ctx max abs err", float((ctx - want).abs().max()), "ref max", float(want.abs().max()), "nan", int(torch.isnan(ctx).sum()))
sc = (torch.bmm(q, k.transpose(1, 2)) * scale).view(B, heads, T, S).masked_fill(mask.unsqueeze(1), -1e9).view(BH, T, S)
print("lse max abs err", float((lse[..., 0] - torch.log(lse[..., 1]) - torch.logsumexp(sc, -1)).abs().max()))
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print(f"library path {t(ref):.1f} us, fused {t(run):.1f} us")
# ---- backward
go = torch.randn(BH, T, 64, device="cuda")
qg, kg, vg = (t_.clone().requires_grad_(True) for t_ in (q, k, v))
pr = ops.masked_softmax(ops.bmm_small(qg, kg.transpose(1, 2)), mask, scale, heads)
ref_g = torch.autograd.grad(ops.bmm_small(pr, vg), (qg, kg, vg), go)
gq, gk, gv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
dsum = torch.empty(BH, T, device="cuda")
def runb():
    rc = lib.singa_attn_bwd(p(q), p(k), p(v), p(mask), mask.stride(0), 0, p(ctx), p(lse), p(go), p(gq), p(gk), p(gv), p(dsum),
                            BH, T, S, heads, 32, 64, 0, 0, 0, 0, scale, st); assert rc == 0
run(); runb(); torch.cuda.synchronize()
for nm, a, b in zip("qkv", (gq, gk, gv), ref_g):
    print(f"g_{nm}: rel err {float((a - b).norm() / b.norm()):.2e} nan {int(torch.isnan(a).sum())}")
def refb():
    pr = ops.masked_softmax(ops.bmm_small(qg, kg.transpose(1, 2)), mask, scale, heads)
    torch.autograd.grad(ops.bmm_small(pr, vg), (qg, kg, vg), go)
print(f"library fwd+bwd {t(refb):.1f} us, fused backward alone {t(runb):.1f} us")
