import torch, time, sys
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); a=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-a)/n*1e6
shapes = [("node 6400x256 @ 256x256", (6400,256),(256,256)), ("node 6400x1024 @ 1024x256", (6400,1024),(1024,256)),
          ("dW 256x6400 @ 6400x256", (256,6400),(6400,256)), ("dW 1024x6400 @ 6400x256", (1024,6400),(6400,256)),
          ("dW 256x6432 @ 6432x1024", (256,6432),(6432,1024)), ("lig 960x256 @ 256x256", (960,256),(256,256)),
          ("emb 6400x144 @ 144x256", (6400,144),(144,256)), ("dW 116x6400 @ 6400x256", (116,6400),(6400,256)),
          ("edge 28800x64 @ 64x64", (28800,64),(64,64)), ("dW 64x28800 @ 28800x64", (64,28800),(28800,64)),
          ("dW 16x54400 @ 54400x48", (16,54400),(54400,48)), ("edge 54400x48 @ 48x16", (54400,48),(48,16)),
          ("dW lig 256x960 @ 960x256", (256,960),(960,256)), ("dW 736x54400 @ 54400x96", (736,54400),(54400,96)),
          ("fwd 54400x96 @ 96x736", (54400,96),(96,736)), ("fwd 54400x384 @ 384x336", (54400,384),(384,336)),
          ("fwd 54400x256 @ 256x448", (54400,256),(256,448)), ("dX 54400x736 @ 736x96", (54400,736),(736,96)),
          ("dW 64x374578 @ 374578x64", (64,374578),(374578,64)), ("fwd 374578x64 @ 64x64", (374578,64),(64,64)),
          ("dec 6432x256 @ 256x1024", (6432,256),(256,1024))]
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print("==", lib)
    for name, sa, sb in shapes:
        a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
        at = torch.randn(sa[1], sa[0], device="cuda").t()   # transposed-storage A (as autograd dW sees it)
        us = t(lambda: a @ b); ust = t(lambda: at @ b)
        fl = 2*sa[0]*sa[1]*sb[1]
        print(f"{name:34s} {us:8.1f} us {fl/us*1e-6:7.2f} TF | A^T-storage {ust:8.1f} us")
