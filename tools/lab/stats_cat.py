"""Category sums of a rocprofv3 kernel_stats.csv: python tools/lab/stats_cat.py <dir> <steps_in_run>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cat = collections.defaultdict(lambda: [0, 0.0])
def which(n):
    if n.startswith("Cijk_"): return "library GEMM"
    if "colsum" in n: return "singa colsum"
    if "adam" in n: return "singa adam"
    if "(anonymous namespace)::" in n and "at::native" not in n: return "singa HIP kernels"
    if "layer_norm" in n or "LayerNorm" in n or "GammaBeta" in n or "cuComputeGradInput" in n: return "torch layer norm"
    if "copyBuffer" in n or "direct_copy" in n or "fillBuffer" in n: return "copies"
    if "FillFunctor" in n: return "fills"
    if "elementwise" in n: return "torch elementwise"
    if "softmax" in n: return "torch softmax"
    if "sort" in n.lower() or "rocprim" in n or "topk" in n.lower(): return "sort/scan"
    if "index" in n or "gather" in n or "scatter" in n: return "index/gather"
    if "CatArray" in n: return "cat"
    return "other"
for r in csv.DictReader(open(f)):
    c = cat[which(r["Name"])]
    c[0] += int(r["Calls"]); c[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in cat.values())
for k, (c, t) in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:22s} {c / steps:8.0f} launches/step {t / 1e6 / steps:8.2f} ms/step {t / tot * 100:5.1f}%")
print(f"{'total':22s} {sum(v[0] for v in cat.values()) / steps:8.0f} launches/step {tot / 1e6 / steps:8.2f} ms/step")
