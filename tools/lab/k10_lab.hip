// Kernel lab (not shipped): times variants of the scatter-TP forward kernel on the bench shape to find what bounds it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/k10_lab tools/lab/k10_lab.hip && /tmp/k10_lab
#include "../../singa_amd/csrc/singa_hip.hip"

#include <vector>
#include <random>

namespace lab {

// V1: same traffic, no rotation (acc += v): is the kernel bound by the Wigner work / readlanes?
template <int L, int M, int U>
__global__ void stream_only(Segs msg, const float* __restrict__ alpha, const int* __restrict__ row_ptr,
                            float* __restrict__ out, int N, int CH, int vh) {
    using I = SO3Idx<L, M>;
    const int c = threadIdx.x < CH ? threadIdx.x : CH - 1;
    const int heads = CH / vh;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        float acc[I::K];
#pragma unroll
        for (int k = 0; k < I::K; ++k) acc[k] = 0.f;
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        for (int e0 = beg; e0 < end; e0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = e0 + u < end;
                const int e = ok ? e0 + u : end - 1;
                const float a = ok ? alpha[(long long)e * heads + c / vh] : 0.f;
                const float* b0 = msg.p[0] + (long long)e * msg.ld[0] + c;
                const float* b1 = msg.p[1] + (long long)e * msg.ld[1] + c;
                const float* b2 = msg.p[2] + (long long)e * msg.ld[2] + c;
#pragma unroll
                for (int r = 0; r < msg.rows[0] && r < 3; ++r) acc[r] += b0[r * CH] * a;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[3 + r] += b1[r * CH] * a;
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[7 + r] += b2[r * CH] * a;
            }
        }
        if (threadIdx.x < CH) {
            float* o = out + (long long)n * I::K * CH + c;
#pragma unroll
            for (int k = 0; k < I::K; ++k) o[(long long)k * CH] = acc[k];
        }
    }
}

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) b[i] = a[i];
}
__global__ void read4(const float4* __restrict__ a, float* __restrict__ sink, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, st = (long long)gridDim.x * blockDim.x;
    float s = 0.f;
    for (; i < n; i += st) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) sink[0] = s;
}

}  // namespace lab

template <class F>
float time_us(F f, int reps = 30) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

template <class F>
float time_cold_us(F launch, void* flush, size_t flush_bytes, int reps = 8) {
    // each timed dispatch runs right after a 1 GiB memset: L2 and the 256 MiB Infinity Cache hold none of its inputs
    float tot = 0.f;
    for (int i = 0; i < reps; ++i) {
        (void)hipMemsetAsync(flush, i, flush_bytes, 0);
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        launch(a, b);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        tot += ms;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    return tot * 1e3f / reps;
}

int main() {
    constexpr int L = 2, CH = 112, heads = 7;
    using I = SO3Idx<L, 2>;
    const int N = 6400, E = 54400;
    std::mt19937 rng(1);
    std::vector<int> deg(N, E / N);
    for (int i = 0; i < E - (E / N) * N; ++i) deg[i]++;
    std::vector<int> rp(N + 1, 0);
    for (int i = 0; i < N; ++i) rp[i + 1] = rp[i] + deg[i];
    if (FILE* f = fopen("tools/lab/row_ptr_cfg2_pp.bin", "rb")) {   // real degree distribution of the bench batch
        size_t got = fread(rp.data(), 4, N + 1, f);
        fclose(f);
        printf("loaded real row_ptr (%zu entries, E=%d)\n", got, rp[N]);
    }
    float *y0, *y1, *y2, *alpha, *wr, *out; int* drp;
    size_t n0 = (size_t)E * 3 * CH, n1 = (size_t)E * 4 * CH, n2 = (size_t)E * 2 * CH;
    (void)hipMalloc(&y0, n0 * 4); (void)hipMalloc(&y1, n1 * 4); (void)hipMalloc(&y2, n2 * 4);
    (void)hipMalloc(&alpha, (size_t)E * heads * 4); (void)hipMalloc(&wr, (size_t)E * I::WSZ * 4);
    (void)hipMalloc(&out, (size_t)N * I::K * CH * 4); (void)hipMalloc(&drp, (N + 1) * 4);
    (void)hipMemset(y0, 0x3c, n0 * 4); (void)hipMemset(y1, 0x3c, n1 * 4); (void)hipMemset(y2, 0x3c, n2 * 4);
    (void)hipMemset(alpha, 0x3c, (size_t)E * heads * 4); (void)hipMemset(wr, 0x3c, (size_t)E * I::WSZ * 4);
    (void)hipMemcpy(drp, rp.data(), (N + 1) * 4, hipMemcpyHostToDevice);
    Segs s; s.p[0] = y0; s.p[1] = y1; s.p[2] = y2; s.ld[0] = 3 * CH; s.ld[1] = 4 * CH; s.ld[2] = 2 * CH;
    s.rows[0] = 3; s.rows[1] = 4; s.rows[2] = 2;
    double bytes = (double)E * (I::KR * CH * 4 + heads * 4 + I::WSZ * 4) + (double)N * I::K * CH * 4 + (N + 1) * 4;
    auto rep = [&](const char* name, float us) { printf("%-44s %8.2f us  %7.1f GB/s (algorithmic)\n", name, us, bytes / us * 1e-3); };
    rep("k10 U=4 grid=N wg=128", time_us([&] { hipLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 4>), dim3(N), dim3(128), 0, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }));
    rep("k10 U=1 grid=N wg=128", time_us([&] { hipLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 1>), dim3(N), dim3(128), 0, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }));
    rep("k10 U=2 grid=N wg=128", time_us([&] { hipLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 2>), dim3(N), dim3(128), 0, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }));
    for (int g : {1024, 2048, 3072, 4096})
        { char nm[64]; snprintf(nm, 64, "k10 U=4 grid=%d wg=128", g);
          rep(nm, time_us([&] { hipLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 4>), dim3(g), dim3(128), 0, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); })); }
    rep("stream-only U=4 grid=N", time_us([&] { hipLaunchKernelGGL((lab::stream_only<L, 2, 4>), dim3(N), dim3(128), 0, 0, s, alpha, drp, out, N, CH, CH / heads); }));
    rep("stream-only U=1 grid=N", time_us([&] { hipLaunchKernelGGL((lab::stream_only<L, 2, 1>), dim3(N), dim3(128), 0, 0, s, alpha, drp, out, N, CH, CH / heads); }));
    void* flush; size_t fb = (size_t)1 << 30; (void)hipMalloc(&flush, fb);
    rep("COLD k10 U=4 grid=N", time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 4>), dim3(N), dim3(128), 0, 0, a, b, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }, flush, fb));
    rep("COLD k10 U=2 grid=N", time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 2>), dim3(N), dim3(128), 0, 0, a, b, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }, flush, fb));
    rep("COLD k10 U=1 grid=N", time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((rotate_back_scatter_kernel<L, 2, false, 1>), dim3(N), dim3(128), 0, 0, a, b, 0, s, alpha, wr, drp, out, N, CH, CH / heads, 1.f); }, flush, fb));
    rep("COLD stream-only U=4", time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL((lab::stream_only<L, 2, 4>), dim3(N), dim3(128), 0, 0, a, b, 0, s, alpha, drp, out, N, CH, CH / heads); }, flush, fb));
    {
        float us = time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(lab::read4, dim3(2048), dim3(256), 0, 0, a, b, 0, (const float4*)y1, out, (long long)n1 / 4); }, flush, fb);
        printf("%-44s %8.2f us  %7.1f GB/s (rd)\n", "COLD float4 read 97 MB", us, (double)n1 * 4 / us * 1e-3);
        us = time_cold_us([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(lab::copy4, dim3(2048), dim3(256), 0, 0, a, b, 0, (const float4*)y1, (float4*)y0, (long long)n0 / 4); }, flush, fb);
        printf("%-44s %8.2f us  %7.1f GB/s (rd+wr)\n", "COLD float4 copy 73 MB", us, 2.0 * n0 * 4 / us * 1e-3);
    }
    long long nf4 = (long long)(n1) / 4;  // 97 MB buffer
    float us = time_us([&] { hipLaunchKernelGGL(lab::copy4, dim3(2048), dim3(256), 0, 0, (const float4*)y1, (float4*)y0, (long long)n0 / 4); });
    printf("%-44s %8.2f us  %7.1f GB/s (rd+wr)\n", "float4 copy 73 MB -> 73 MB", us, 2.0 * n0 * 4 / us * 1e-3);
    us = time_us([&] { hipLaunchKernelGGL(lab::read4, dim3(2048), dim3(256), 0, 0, (const float4*)y1, out, nf4); });
    printf("%-44s %8.2f us  %7.1f GB/s (rd)\n", "float4 read 97 MB", us, (double)n1 * 4 / us * 1e-3);
    return 0;
}
