"""Lab: the SO(2)-convolution GEMMs (config 3: E = 99,302 edges, L = 4) forward / dX / dW with the product library and with a
lab build of it (extra -D flags given as arguments, e.g. -DSINGA_GEMM_PIPE=0), each in its own child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--run":
    import torch
    from singa_amd import _lib
    if os.environ.get("SINGA_LAB_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["SINGA_LAB_LIB"])
    from singa_amd import ops
    dev, E = "cuda", 99302

    def t_us(fn, n=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e3

    for name, blocks in (("conv2", [(640, 560), (1024, 896), (768, 672)]), ("conv1", [(160, 992), (256, 1024), (192, 768)])):
        nin, nout = sum(b[0] for b in blocks), sum(b[1] for b in blocks)
        X = torch.randn(E, nin, device=dev)
        G = torch.randn(E, nout, device=dev)
        ws = [torch.randn(o, i, device=dev) * 0.05 for i, o in blocks]
        H, GX = torch.empty(E, nout, device=dev), torch.empty(E, nin, device=dev)
        flops = 2 * E * sum(i * o for i, o in blocks)
        S = ops._splits_for(E)
        part = torch.empty(S, sum(i * o for i, o in blocks), device=dev)

        def nt():
            items, ai, ci = [], 0, 0
            for w, (k, o) in zip(ws, blocks):
                items.append(dict(a=X.data_ptr() + 4 * ai, lda=nin, b=w.data_ptr(), ldb=k, c=H.data_ptr() + 4 * ci, ldc=nout, I=E, J=o, R=k))
                ai, ci = ai + k, ci + o
            ops._gemm(items, True, True)

        def nn():
            items, ai, ci = [], 0, 0
            for w, (k, o) in zip(ws, blocks):
                items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=w.data_ptr(), ldb=k, c=GX.data_ptr() + 4 * ai, ldc=nin, I=E, J=k, R=o))
                ai, ci = ai + k, ci + o
            ops._gemm(items, True, False)

        def tn():
            items, ai, ci, off = [], 0, 0, 0
            tot = part.shape[1]
            for w, (k, o) in zip(ws, blocks):
                items.append(dict(a=G.data_ptr() + 4 * ci, lda=nout, b=X.data_ptr() + 4 * ai, ldb=nin, c=part.data_ptr() + 4 * off, ldc=k,
                                  I=o, J=k, R=E, c_split_stride=tot))
                ai, ci, off = ai + k, ci + o, off + o * k
            ops._gemm(items, False, False, S)

        line = f"{os.environ.get('SINGA_LAB_TAG', 'product'):28s} {name}:"
        for tag, fn in (("NT", nt), ("NN", nn), ("TN", tn)):
            t = t_us(fn)
            line += f"  {tag} {t:7.1f} us {flops / t / 1e6:6.1f} TF/s"
        print(line, flush=True)
    sys.exit(0)

variants = [("product", None)] + [(a, a) for a in sys.argv[1:]]
for tag, flag in variants:
    env = dict(os.environ, SINGA_LAB_TAG=tag)
    if flag:
        out = os.path.join(ROOT, "tools", "lab", "build", "lib_" + "".join(c if c.isalnum() else "_" for c in flag) + ".so")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + flag.split() +
                              ["-o", out, os.path.join(ROOT, "singa_amd", "csrc", "singa_hip.hip")], stderr=subprocess.DEVNULL)
        env["SINGA_LAB_LIB"] = out
    subprocess.call([sys.executable, os.path.abspath(__file__), "--run"], env=env)
