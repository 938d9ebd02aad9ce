"""Ablation timing of the k15c backward kernel: extracts the edge-MLP section of singa_hip.hip into a standalone HIP
program, applies one textual edit per variant (results become wrong - only the time matters), builds and runs each.
    python tools/lab/emlp_ablate.py [E]        (on the GPU box)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "singa_amd", "csrc", "singa_hip.hip")).read()
a = src.index("// one net on one 32-edge tile.  asel[s] = attr[edge][32 * half + s]")
b = src.index("// ------------------------------------------------------------------------------------------------ masked, scaled softmax")
body = src[a:b]
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2_900_000
HDR = r'''#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define SINGA_RCP(x) __builtin_amdgcn_rcpf(x)
__device__ __forceinline__ float ssp_fast(float x) { return fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))) - 0.69314718055994530942f; }
'''
MAIN = r'''
int main(int argc, char** argv) {
    const int E = atoi(argv[1]);
    float *attr, *g, *w1, *b1, *w2, *part, *wk, *wv;
    hipMalloc(&attr, (size_t)E * 64 * 4); hipMalloc(&g, (size_t)E * 64 * 4);
    hipMalloc(&wk, (size_t)E * 32 * 4); hipMalloc(&wv, (size_t)E * 64 * 4);
    hipMalloc(&w1, 64 * 64 * 4); hipMalloc(&b1, 64 * 4); hipMalloc(&w2, 64 * 64 * 4); hipMalloc(&part, (size_t)512 * 8400 * 4);
    std::vector<float> h((size_t)E * 64);
    for (auto& v : h) v = (float)(rand() % 20001) * 1e-4f - 1.0f;
    hipMemcpy(attr, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w1, h.data(), 64 * 64 * 4, hipMemcpyHostToDevice); hipMemcpy(w2, h.data() + 5000, 64 * 64 * 4, hipMemcpyHostToDevice);
    hipMemcpy(b1, h.data() + 9000, 64 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 3; ++which) {
        float ms = 0;
        for (int it = 0; it < 23; ++it) {
            if (it == 3) hipEventRecord(e0, 0);
            if (which == 0) hipLaunchKernelGGL((edge_mlp_mfma_bwd_kernel<64>), dim3(256, 2), dim3(256), 0, 0, attr, g, w1, b1, w2, part, E);
            if (which == 1) hipLaunchKernelGGL((edge_mlp_mfma_bwd_kernel<32>), dim3(512, 1), dim3(256), 0, 0, attr, g, w1, b1, w2, part, E);
            if (which == 2) hipLaunchKernelGGL(edge_mlp_mfma_fwd_kernel, dim3(768), dim3(256), 0, 0, attr, w1, b1, w2, b1, w1, b1, w2, b1, wk, wv, E);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf(" %s %8.1f us", which == 0 ? "bwd64" : which == 1 ? "bwd32" : "fwd", ms / 20 * 1e3);
    }
    printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
'''


def sub(text, old, new, count=1):
    assert old in text, old
    return text.replace(old, new, count)


def no_act(t):
    t = sub(t, "const float tq = __expf(-fabsf(p)), u = 1.f + tq;", "const float tq = p, u = p;")
    t = sub(t, "const float sg = (p >= 0.f ? 1.f : tq) * SINGA_RCP(u);", "const float sg = tq;")
    return sub(t, "pacc[r] = ok ? fmaxf(p, 0.f) + __logf(u) - 0.69314718055994530942f : 0.f;", "pacc[r] = p + u;")


def no_prefetch(t):       # bwd only: the in-loop fetch of the next tile (the first tile's rows are reused)
    return sub(t, "if (tile + stride < tilesN) fetch(tile + stride);   // the attr registers are free now; 96 MFMAs of cover", "")


def no_fences(t):
    i0 = t.index("template <int H>\n__global__ void __launch_bounds__(256, 2) edge_mlp_mfma_bwd_kernel")
    return t[:i0] + t[i0:].replace("__builtin_amdgcn_sched_barrier(0);", "")


def no_db(t):
    t = sub(t, "db2p[a] += gv;", "")
    i0 = t.index("        if (lane < 32) {\n            float c = 0.f;")
    i1 = t.index("            db1p += c;\n        }\n", i0) + len("            db1p += c;\n        }\n")
    return t[:i0] + t[i1:]


def coalesced_fetch(t):     # same bytes per tile, lane-linear addresses (1 KB contiguous per load instruction): wrong operands, right traffic
    t = sub(t, "nasel4[m] = *reinterpret_cast<const float4*>(attr + er * 64 + 32 * half + 4 * m);",
            "nasel4[m] = *reinterpret_cast<const float4*>(attr + first_row(tl) * 64 + (m * 64 + lane) * 4 + 0 * er);")
    return sub(t, "ngrow4[m] = *reinterpret_cast<const float4*>(g_out + er * H + HH * half + 4 * m);",
               "ngrow4[m] = *reinterpret_cast<const float4*>(g_out + first_row(tl) * H + (m * 64 + lane) * 4);")


def coalesced_fetch_fwd(t):
    return sub(t, "for (int m = 0; m < 8; ++m) nxt[m] = *reinterpret_cast<const float4*>(row + 4 * m);",
               "for (int m = 0; m < 8; ++m) nxt[m] = *reinterpret_cast<const float4*>(attr + (tl * 32 < E - 32 ? tl * 32 : E - 32) * 64 + (m * 64 + lane) * 4 + 0 * (row - attr));")


def linear_stores_fwd(t):     # forward: same bytes stored, lane-linear addresses (1 KB contiguous per store instruction)
    t = sub(t, "*reinterpret_cast<float4*>(wk + e * 32 + 8 * blk + 4 * half) =", "*reinterpret_cast<float4*>(wk + tile * 1024 + (blk * 64 + lane) * 4) =")
    return sub(t, "*reinterpret_cast<float4*>(wv + e * 64 + 32 * u + 8 * blk + 4 * half) =", "*reinterpret_cast<float4*>(wv + tile * 2048 + ((u * 4 + blk) * 64 + lane) * 4) =")


def no_stores_fwd(t):
    t = sub(t, "*reinterpret_cast<float4*>(wk + e * 32 + 8 * blk + 4 * half) =", "if (out[0][4 * blk] == 123.456f) *reinterpret_cast<float4*>(wk + e * 32 + 8 * blk + 4 * half) =")
    return sub(t, "*reinterpret_cast<float4*>(wv + e * 64 + 32 * u + 8 * blk + 4 * half) =", "if (out[u][4 * blk] == 123.456f) *reinterpret_cast<float4*>(wv + e * 64 + 32 * u + 8 * blk + 4 * half) =")


def no_act_fwd(t):
    return sub(t, "hacc[t][r] = ssp_fast(hacc[t][r]);", "hacc[t][r] = hacc[t][r] + 1.0f;")


VARIANTS = [("full", lambda t: t), ("fwd: lane-linear stores", linear_stores_fwd), ("fwd: no stores", no_stores_fwd),
            ("fwd: no activation math", no_act_fwd), ("fwd: no stores, no activation", lambda t: no_act_fwd(no_stores_fwd(t))), ("lane-linear row loads (bwd and fwd)", lambda t: coalesced_fetch_fwd(coalesced_fetch(t))), ("no activation math", no_act), ("no prefetch of the next tile", no_prefetch),
            ("no act + no prefetch", lambda t: no_prefetch(no_act(t))), ("no sched_barrier fences", no_fences),
            ("no bias-gradient sums", no_db)]
os.makedirs("/tmp/emlp_abl", exist_ok=True)
for k, (name, f) in enumerate(VARIANTS):
    path = f"/tmp/emlp_abl/v{k}.hip"
    open(path, "w").write(HDR + f(body) + MAIN)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-w", "-o", f"/tmp/emlp_abl/v{k}", path])
    out = subprocess.run([f"/tmp/emlp_abl/v{k}", str(E)], capture_output=True, text=True, timeout=120)
    print(f"{name:45s} {out.stdout.strip()} {out.stderr.strip()[-200:]}", flush=True)
