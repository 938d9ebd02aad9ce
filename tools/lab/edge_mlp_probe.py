import os, sys, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__; __graft_entry__.build()
from singa_amd import _lib
E = int(sys.argv[1]) if len(sys.argv) > 1 else 374578
torch.manual_seed(0)
attr = torch.randn(E, 64, device="cuda")
nets = [(torch.nn.Linear(64, H, device="cuda"), torch.nn.Linear(H, H, device="cuda")) for H in (32, 64)]
with torch.no_grad():
    want = [l2(torch.nn.functional.softplus(l1(attr)) - math.log(2.0)) for l1, l2 in nets]
    ws = []
    for l1, l2 in nets:
        ws += [l1.weight.t().contiguous(), l1.bias.contiguous(), l2.weight.t().contiguous(), l2.bias.contiguous()]
wk, wv = torch.full((E, 32), float("nan"), device="cuda"), torch.full((E, 64), float("nan"), device="cuda")
p = lambda t: ctypes.c_void_p(t.data_ptr())
lib = _lib.lib(); _lib.ensure_init(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    rc = lib.singa_edge_mlp_fwd(p(attr), *[p(t) for t in ws], p(wk), p(wv), E, 64, 32, 64, st)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
for got, ref, n in ((wk, want[0], "k"), (wv, want[1], "v")):
    print(n, "max abs err", float((got - ref).abs().max()), "ref max", float(ref.abs().max()), "nan", int(torch.isnan(got).sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): run()
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"fused forward {e0.elapsed_time(e1) / 20 * 1e3:.1f} us for E={E}")
# ---- backward
for (l1, l2), H in zip(nets, (32, 64)):
    g = torch.randn(E, H, device="cuda")
    out = l2(torch.nn.functional.softplus(l1(attr)) - math.log(2.0))
    ref = torch.autograd.grad(out, (l1.weight, l1.bias, l2.weight, l2.bias), g)
    n = lib.singa_edge_mlp_bwd_nparts(E, H)
    S = H // 32
    psz = 32 * 64 + 32 + H * 32 + H
    part = torch.full((n, S, psz), float("nan"), device="cuda")
    w1t, w2 = l1.weight.detach().t().contiguous(), l2.weight.detach().contiguous()
    def runb():
        rc = lib.singa_edge_mlp_bwd(p(attr), p(g), p(w1t), p(l1.bias.detach()), p(w2), p(part), E, 64, H, st)
        assert rc == 0, rc
    runb(); torch.cuda.synchronize()
    tot = part.double().sum(0)
    o1, o2, o3 = 32 * 64, 32 * 64 + 32, 32 * 64 + 32 + H * 32
    got = (tot[:, :o1].reshape(H, 64), tot[:, o1:o2].reshape(H), tot[:, o2:o3].reshape(S, H, 32).permute(1, 0, 2).reshape(H, H), tot[0, o3:])
    for nm, a, b in zip(("dW1", "db1", "dW2", "db2"), got, ref):
        print(f"H={H} {nm}: rel err {float((a - b.double()).norm() / b.double().norm()):.2e}  nan {int(torch.isnan(a).sum())}")
    for _ in range(3): runb()
    e0.record()
    for _ in range(10): runb()
    e1.record(); torch.cuda.synchronize()
    print(f"H={H} fused backward {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
