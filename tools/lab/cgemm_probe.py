"""Lab: the 3M complex SO(2) convolution (ops.so2_conv3m, k7c) against the block-weight form (ops.so2_linear3, k7): results,
gradients and time per pass at the conv1 / conv2 shapes of the bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from singa_amd import ops

dev = "cuda"
torch.manual_seed(0)


def run(tag, E, ins, outs, reps=10):
    n0, n1, n2 = ins
    X = torch.randn(E, sum(ins), device=dev, requires_grad=True)
    w0 = (torch.randn(outs[0], n0, device=dev) / n0 ** 0.5).requires_grad_(True)
    b0 = torch.randn(outs[0], device=dev, requires_grad=True)
    w1 = (torch.randn(outs[1], n1 // 2, device=dev) / n1 ** 0.5).requires_grad_(True)
    w2 = (torch.randn(outs[2], n2 // 2, device=dev) / n2 ** 0.5).requires_grad_(True)
    G = [torch.randn(E, o, device=dev) for o in outs]

    def old():
        ys = ops.so2_linear3(X, w0, b0, ops.block_weight(w1), ops.block_weight(w2), n0, n1)
        return ys

    def new():
        return ops.so2_conv3m(X, w0, b0, w1, w2, n0, n1)

    res = {}
    for name, fn in (("old", old), ("new", new)):
        for p in (X, w0, b0, w1, w2):
            p.grad = None
        ys = fn()
        torch.autograd.backward(list(ys), G)
        res[name] = [y.detach().double() for y in ys] + [p.grad.double().clone() for p in (X, w0, b0, w1, w2)]
    # float64 reference
    Xd, W0, B0, W1, W2 = (t.detach().double() for t in (X, w0, b0, w1, w2))

    def cplx(xb, w):
        K, N = xb.shape[1] // 2, w.shape[0] // 2
        xr, xi, wr, wi = xb[:, :K], xb[:, K:], w[:N], w[N:]
        return torch.cat([xr @ wr.t() - xi @ wi.t(), xr @ wi.t() + xi @ wr.t()], 1)

    Xr = Xd.clone().requires_grad_(True)
    W0r, B0r, W1r, W2r = (t.clone().requires_grad_(True) for t in (W0, B0, W1, W2))
    ys = [Xr[:, :n0] @ W0r.t() + B0r, cplx(Xr[:, n0:n0 + n1], W1r), cplx(Xr[:, n0 + n1:], W2r)]
    torch.autograd.backward(ys, [g.double() for g in G])
    ref = [y.detach() for y in ys] + [Xr.grad, W0r.grad, B0r.grad, W1r.grad, W2r.grad]
    names = ["y0", "y1", "y2", "gX", "gw0", "gb0", "gw1", "gw2"]
    line = f"{tag} E={E}:"
    for i, n in enumerate(names):
        eo = float((res["old"][i] - ref[i]).norm() / ref[i].norm())
        en = float((res["new"][i] - ref[i]).norm() / ref[i].norm())
        line += f"  {n} {eo:.1e}/{en:.1e}"
    print(line + "   (old / new vs float64)", flush=True)
    for name, fn in (("old", old), ("new", new)):
        for p in (X, w0, b0, w1, w2):
            p.grad = None
        torch.cuda.synchronize()
        ops.profile_start()
        torch.autograd.backward(list(fn()), G)
        torch.cuda.synchronize()
        print(f"    {name} launches: " + "  ".join(f"{t} {1e3 * ms:.0f}us({nn} tiles)" for t, ms, ne, nn in ops.profile_collect()), flush=True)
        for phase in ("fwd", "fwd+bwd"):
            ts = []
            for _ in range(reps):
                for p in (X, w0, b0, w1, w2):
                    p.grad = None
                torch.cuda.synchronize()
                t = time.perf_counter()
                if phase == "fwd":
                    with torch.no_grad():
                        fn()
                else:
                    torch.autograd.backward(list(fn()), G)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t)
            ts.sort()
            print(f"    {name} {phase}: {1e3 * ts[len(ts) // 2]:.3f} ms", flush=True)


CASES = {"conv2": ("conv2 L4", (640, 1024, 768), (640, 1024, 768)), "conv1": ("conv1 L4", (160, 256, 192), (1024, 1024, 768)),
         "conv2_L2": ("conv2 L2", (384, 512, 256), (384, 512, 256)), "conv1_L6": ("conv1 L6", (224, 384, 320), (896 + 384, 1536, 1280))}
if len(sys.argv) > 1:                    # one case (for a profiler run): name E
    t, i, o = CASES[sys.argv[1]]
    run(t, int(sys.argv[2]), i, o)
else:
    for E in (101632, 13337):
        run(*((CASES["conv2"][0], E) + CASES["conv2"][1:]))
        run(*((CASES["conv1"][0], E) + CASES["conv1"][1:]))
    run("conv2 L2", 5003, *CASES["conv2_L2"][1:])
    run("conv1 L6", 4099, *CASES["conv1_L6"][1:])
