"""Times the per-edge MLP kernels (k15c: forward for both nets, backward per net) on E kNN edges through the C ABI.
    python tools/lab/edge_mlp_probe.py [E]
Flops per edge: forward 2*(64*32 + 32*32) + 2*(64*64 + 64*64) = 22,528; backward H=32: 12,288; H=64: 32,768."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import _lib, ops

if os.environ.get("SINGA_PROBE_LIB"):          # lab only: time another build of the library (A/B on one box)
    _lib.LIB_PATH = os.environ["SINGA_PROBE_LIB"]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2_900_000
dev = "cuda"
torch.manual_seed(0)
attr = torch.randn(E, 64, device=dev)
nets = [(torch.nn.Linear(64, H, device=dev), torch.nn.Linear(H, H, device=dev)) for H in (32, 64)]
gk, gv = torch.randn(E, 32, device=dev), torch.randn(E, 64, device=dev)
lib = _lib.lib()
_lib.ensure_init(0)
p = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
wk, wv = torch.empty(E, 32, device=dev), torch.empty(E, 64, device=dev)
fw = [t.detach().contiguous() for l1, l2 in nets for t in (l1.weight, l1.bias, l2.weight, l2.bias)]


def fwd():
    assert lib.singa_edge_mlp_fwd(p(attr), *[p(t) for t in fw], p(wk), p(wv), E, 64, 32, 64, st) == 0


def bwd(H):
    g = gk if H == 32 else gv
    w1, b1, w2 = (fw[0], fw[1], fw[2]) if H == 32 else (fw[4], fw[5], fw[6])
    part = torch.empty(lib.singa_edge_mlp_bwd_nparts(E, H), H * 64 + H + H * H + H, device=dev)
    return lambda: lib.singa_edge_mlp_bwd(p(attr), p(g), p(w1), p(b1), p(w2), p(part), E, 64, H, st)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, f, flop in (("fwd", fwd, 22528), ("bwd32", bwd(32), 12288), ("bwd64", bwd(64), 32768)):
    us = timeit(f)
    print(f"{name:6s} E={E}: {us:8.1f} us  {E * flop / us / 1e6:6.1f} TF/s  ({E * flop / us / 1e6 / 157.3:.2f} of the f32 MFMA peak)")
