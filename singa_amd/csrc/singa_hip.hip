// libsinga_hip.so — hand-written gfx950 (MI355X / CDNA4) kernels for the SINGA equivariant message-passing
// hot path, behind the C ABI declared in include/singa_hip.h.
//
// Design notes (DESIGN.md has the long form):
//  * 64-wide wavefronts; one lane = one channel of one edge / node, so every global access of a wave is one
//    contiguous run of channels (coalesced 64-448 B rows).
//  * Edges are sorted by destination.  Aggregation kernels give one workgroup to one destination node and walk
//    its edge segment with register accumulators: no atomics, no [E,K,CH] intermediate, bit-reproducible sums.
//  * Per-edge Wigner rows are wave-uniform (edge index derives from blockIdx), so they are fetched with scalar
//    loads and used as SGPR operands of v_fma: no LDS traffic and no per-lane operand fetch for the rotation.
//  * Loops over (l, m, j) are fully unrolled against the constexpr tables of so3_index.h; accumulators live in
//    statically indexed VGPRs.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/singa_hip.h"
#include "../../include/singa_hip_lab.h"
#include "so3_index.h"

namespace {

thread_local char g_err[256] = "ok";

int fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return SINGA_OK;
}

// Optional per-dispatch timing of the scatter-TP forward kernel: when enabled, its launches go through
// hipExtLaunchKernelGGL with a start and a stop event attached to the dispatch itself (the timestamps rocprofv3 reads),
// on the caller's stream.  singa_prof_collect() reads them back after the caller has synchronised.
struct ProfRec {
    hipEvent_t a, b;
    int E, N, tag;
};
constexpr int PROF_CAP = 8192;
ProfRec g_prof[PROF_CAP];
int g_prof_n = 0;
bool g_prof_on = false;
// singa_prof_stamps(buf, cap): "graph mode" - instead of events (external event-record nodes are refused under stream capture by
// this ROCm build) a one-thread kernel in front of and behind every tagged launch writes the 100 MHz wall clock into the caller's
// buffer: plain kernel nodes, so they are captured with the step and re-run by every replay.  Record 0 is a calibration pair with
// nothing in between (one dependent-launch gap + the stamp kernel's own run time), subtracted when the records are read.
unsigned long long* g_stamp_buf = nullptr;
int g_stamp_cap = 0;
__global__ void stamp_kernel(unsigned long long* dst) { *dst = (unsigned long long)wall_clock64(); }
int g_prof_edges_hint = 0;  // E of the next profiled launch (the kernel itself only needs row_ptr)

// (Tried in round 4 and dropped: non-temporal loads of the message rows / stores of the node rows in the scatter kernel k10 -
// 320 instead of 270 us per config-3 launch.)
// SINGA_KEEP_VGPR(x): an empty asm that pins x to its own vector register at that point (tests/emul defines it away)
#ifndef SINGA_KEEP_VGPR
#define SINGA_KEEP_VGPR(x) __asm__ volatile("" : "+v"(x))
#endif

// SINGA_LAUNCH(tag, E, N, kernel, grid, block, stream, args...): a plain launch, or - while profiling is enabled - the same
// launch with a start/stop event pair attached to the dispatch and a (tag, E, N) record for singa_prof_collect_tagged.
#define SINGA_LAUNCH(tag_, E_, N_, kern, grid, block, st, ...)                                                   \
    do {                                                                                                          \
        if (g_prof_on && g_prof_n < PROF_CAP) {                                                                   \
            ProfRec& r_ = g_prof[g_prof_n++];                                                                     \
            (void)hipEventCreate(&r_.a);                                                                          \
            (void)hipEventCreate(&r_.b);                                                                          \
            r_.E = (E_);                                                                                          \
            r_.N = (N_);                                                                                          \
            r_.tag = (tag_);                                                                                      \
            hipExtLaunchKernelGGL(kern, grid, block, 0, (st), r_.a, r_.b, 0, __VA_ARGS__);                        \
        } else if (g_stamp_buf && 2 * (g_prof_n + 2) <= g_stamp_cap && g_prof_n + 1 < PROF_CAP) {                 \
            if (g_prof_n == 0) {                                                                                  \
                g_prof[0].a = g_prof[0].b = nullptr;                                                              \
                g_prof[0].E = g_prof[0].N = g_prof[0].tag = 0;                                                    \
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (st), g_stamp_buf);                         \
                hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (st), g_stamp_buf + 1);                     \
                g_prof_n = 1;                                                                                     \
            }                                                                                                     \
            ProfRec& r_ = g_prof[g_prof_n];                                                                       \
            r_.a = r_.b = nullptr;                                                                                \
            r_.E = (E_);                                                                                          \
            r_.N = (N_);                                                                                          \
            r_.tag = (tag_);                                                                                      \
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (st), g_stamp_buf + 2 * g_prof_n);              \
            hipLaunchKernelGGL(kern, grid, block, 0, (st), __VA_ARGS__);                                          \
            hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (st), g_stamp_buf + 2 * g_prof_n + 1);          \
            ++g_prof_n;                                                                                           \
        } else                                                                                                    \
            hipLaunchKernelGGL(kern, grid, block, 0, (st), __VA_ARGS__);                                          \
    } while (0)

// Workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own 4 MiB L2.  The kNN kernels below take
// one centre node per workgroup and gather the rows of its neighbours - atoms of the same molecule, i.e. rows a few
// hundred indices away.  With the plain node = blockIdx mapping every XCD walks the whole batch and each molecule's rows
// are pulled into all eight L2s from the Infinity Cache; handing XCD x the x-th contiguous eighth of the nodes keeps a
// molecule's rows in ONE L2.  The grid is a multiple of 8 (kn_grid); ids past the end are skipped.
__device__ __forceinline__ int xcd_range_id(int vb, int n8) { return (vb & 7) * (n8 >> 3) + (vb >> 3); }
constexpr int MAX_J = 2300;  // sum_{l<=11} (2l+1)^2
__device__ float g_J[MAX_J];
int g_lmax_init = -1;

__host__ __device__ constexpr int j_off(int l) {
    int s = 0;
    for (int i = 0; i < l; ++i) s += (2 * i + 1) * (2 * i + 1);
    return s;
}

struct Segs {
    const float* p[3];
    long long ld[3];
    int rows[3];
};
struct SegsMut {
    float* p[3];
    long long ld[3];
    int rows[3];
};

// v_rcp_f32 (1 ulp) for the sigmoids: `1.0f / x` and `__frcp_rn` are the correctly rounded reciprocal, i.e. the
// ten-instruction IEEE division sequence (div_scale / rcp / fma x 4 / div_fmas / div_fixup) - 28 % of the S2 kernel's
// instructions before this
#ifndef SINGA_RCP
#define SINGA_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
__device__ __forceinline__ float silu(float u) { return u * SINGA_RCP(1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_grad(float u) {
    float s = SINGA_RCP(1.0f + __expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// Per-edge Wigner rows are the same for every lane of a wavefront.  Fetching them with scalar loads serialises on the
// ~100-SGPR budget (the compiler issues them piecemeal, one s_waitcnt round trip per piece: measured 40 us block
// lifetimes).  Instead each wave reads the record ONCE with ordinary coalesced vector loads (lane i holds W[i], W[i+64],
// ...) - these overlap with the message loads - and every use broadcasts one element to an SGPR with v_readlane.
#ifndef SINGA_LANE_BCAST  // tests/emul overrides this with a plain memory read (sequential threads have no lanes)
#define SINGA_LANE_BCAST(reg, lane, ptr, idx) \
    __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (reg)), (lane)))
#endif

template <int WSZ>
struct WRows {
    static constexpr int NW = (WSZ + 63) / 64;
    float v[NW];
    const float* p;
    __device__ __forceinline__ void load(const float* W, int lane) {
        p = W;
#pragma unroll
        for (int k = 0; k < NW; ++k) v[k] = (lane + 64 * k < WSZ) ? W[lane + 64 * k] : 0.f;
    }
};
#define W_AT(w, idx) SINGA_LANE_BCAST((w).v[(idx) / 64], (idx) % 64, (w).p, (idx))

// ------------------------------------------------------------------------------------------------ k1: edge frames
// init_edge_rot_mat (EF:2286-2351), one thread per edge, the reference's operations in the reference's order (including
// the second normalisation of z): x = v / |v|; helper r = (rand - 0.5) / |rand - 0.5|, replaced by one of its two
// 90-degree copies (-r1, r0, r2) / (r0, -r2, r1) when it is the more parallel one; z = x X r, y = x X z; rows of the frame
// = (z, x, -y).  The reference's two guards need min |v| and max |x . r| over the edges: both are order-independent, so
// they are folded with wave butterflies + one integer atomic per wave on the float bit patterns (non-negative floats order
// like their bits; a NaN has the largest pattern, so it survives the max exactly as torch.max propagates it).
// stats[0] = min(stats[0], min |v|), stats[1] = max(stats[1], max |cos|): the caller initialises (inf, 0).
#ifndef SINGA_WAVE_MIN_MAX  // tests/emul: every (sequential) thread is its own wave
#define SINGA_WAVE_MIN_MAX(lo, hi)                             \
    _Pragma("unroll") for (int o_ = 32; o_ > 0; o_ >>= 1) {    \
        int a_ = __shfl_xor((lo), o_, 64), b_ = __shfl_xor((hi), o_, 64); \
        (lo) = a_ < (lo) ? a_ : (lo);                          \
        (hi) = b_ > (hi) ? b_ : (hi);                          \
    }
#define SINGA_WAVE_LEADER ((threadIdx.x & 63) == 0)
#endif
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float norm3(float3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ float absdot3(float3 a, float3 b) { return fabsf(a.x * b.x + a.y * b.y + a.z * b.z); }

__global__ void __launch_bounds__(256) edge_frames_kernel(const float* __restrict__ vec, const float* __restrict__ rnd,
                                                          float* __restrict__ rot, int* __restrict__ stats, int E) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    int dbits = 0x7f800000, cbits = 0;  // +inf, 0
    if (e < E) {
        float3 v = make_float3(vec[3 * (long long)e], vec[3 * (long long)e + 1], vec[3 * (long long)e + 2]);
        float d = norm3(v);
        float3 x = make_float3(v.x / d, v.y / d, v.z / d);
        float3 r = make_float3(rnd[3 * (long long)e] - 0.5f, rnd[3 * (long long)e + 1] - 0.5f, rnd[3 * (long long)e + 2] - 0.5f);
        float rn = norm3(r);
        r = make_float3(r.x / rn, r.y / rn, r.z / rn);
        float3 rb = make_float3(-r.y, r.x, r.z), rc = make_float3(r.x, -r.z, r.y);
        float db = absdot3(rb, x), dc = absdot3(rc, x), d0 = absdot3(r, x);
        if (d0 > db) r = rb;
        d0 = absdot3(r, x);
        if (d0 > dc) r = rc;
        d0 = absdot3(r, x);
        float3 z = cross3(x, r);
        float zn = norm3(z);
        z = make_float3(z.x / zn, z.y / zn, z.z / zn);
        zn = norm3(z);
        z = make_float3(z.x / zn, z.y / zn, z.z / zn);
        float3 y = cross3(x, z);
        float yn = norm3(y);
        y = make_float3(y.x / yn, y.y / yn, y.z / yn);
        float* R = rot + 9 * (long long)e;
        R[0] = z.x, R[1] = z.y, R[2] = z.z;
        R[3] = x.x, R[4] = x.y, R[5] = x.z;
        R[6] = -y.x, R[7] = -y.y, R[8] = -y.z;
        dbits = __builtin_bit_cast(int, d) & 0x7fffffff;   // (a NaN length never wins the min; the NaN dot below reports it)
        cbits = __builtin_bit_cast(int, d0) & 0x7fffffff;
    }
    SINGA_WAVE_MIN_MAX(dbits, cbits);
    if (SINGA_WAVE_LEADER) {
        atomicMin(stats, dbits);
        atomicMax(stats + 1, cbits);
    }
}

// ------------------------------------------------------------------------------------------------ k2: Wigner rows
// One thread per (edge, reduced row).  D_l = Za J Zb J Zc with Z(t) = diag cos(f t) + antidiag sin(f t),
// f = l, l-1, .., -l (EF:2207-2229); angles from the 3x3 frame as e3nn's xyz_to_angles / angles_to_matrix do
// (EF:508-517).  Only rows |m| <= M of each block are produced.
template <int L, int M>
__global__ void wigner_rows_kernel(const float* __restrict__ rot, float* __restrict__ wr, int E) {
    using I = SO3Idx<L, M>;
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    int e = tid / I::KR;
    int r = tid - e * I::KR;
    if (e >= E) return;
    const float* R = rot + (long long)e * 9;
    // x = R * (0,1,0) = second column of R
    float x0 = R[1], x1 = R[4], x2 = R[7];
    float nrm = sqrtf(x0 * x0 + x1 * x1 + x2 * x2);
    nrm = fmaxf(nrm, 1e-12f);
    x0 = fminf(fmaxf(x0 / nrm, -1.f), 1.f);
    x1 = fminf(fmaxf(x1 / nrm, -1.f), 1.f);
    x2 = fminf(fmaxf(x2 / nrm, -1.f), 1.f);
    float beta = acosf(x1), alpha = atan2f(x0, x2);
    float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta);
    // first row of (Ry(alpha) Rx(beta))^T R  ->  gamma = atan2(R'[0][2], R'[0][0])
    // (Ry Rx) column 0 = (ca, 0, -sa)
    float r00 = ca * R[0] - sa * R[6];
    float r02 = ca * R[2] - sa * R[8];
    float gamma = atan2f(r02, r00);
    (void)cb;
    (void)sb;
    // locate (l, mi) of reduced row r
    int l = 0, base = 0;
    while (base + I::nr(l) <= r) {
        base += I::nr(l);
        ++l;
    }
    int mi = r - base;
    int n = 2 * l + 1;
    int i = l - I::mm(l) + mi;  // row inside the (2l+1)x(2l+1) block
    const float* J = g_J + j_off(l);
    float fi = (float)(l - i);
    float ci = cosf(fi * alpha), si = sinf(fi * alpha);
    float B[2 * L + 1], Cc[2 * L + 1];
    // A[i,k] = cos(f_i a) J[i,k] + sin(f_i a) J[n-1-i,k]   (centre row: cos term only)
    // B[i,k] = A[i,k] cos(f_k b) + A[i,n-1-k] sin(f_{n-1-k} b)
    for (int k = 0; k < n; ++k) {
        int kk = n - 1 - k;
        float a_k = ci * J[i * n + k] + ((i != n - 1 - i) ? si * J[(n - 1 - i) * n + k] : 0.f);
        float a_kk = ci * J[i * n + kk] + ((i != n - 1 - i) ? si * J[(n - 1 - i) * n + kk] : 0.f);
        float fk = (float)(l - k), fkk = (float)(l - kk);
        B[k] = a_k * cosf(fk * beta) + ((k != kk) ? a_kk * sinf(fkk * beta) : 0.f);
    }
    for (int k = 0; k < n; ++k) {
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += B[j] * J[j * n + k];
        Cc[k] = s;
    }
    float* out = wr + (long long)e * I::WSZ + I::w_off(l) + mi * n;
    for (int k = 0; k < n; ++k) {
        int kk = n - 1 - k;
        float fk = (float)(l - k), fkk = (float)(l - kk);
        out[k] = Cc[k] * cosf(fk * gamma) + ((k != kk) ? Cc[kk] * sinf(fkk * gamma) : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ k3-k6: gather+rotate
// One wavefront per edge (grid-stride).  Lanes 0..2C-1: channel c of [x_src | x_dst].  MODE 0: forward (out = rotated *
// rad).  MODE 1: backward w.r.t. rad (g_rad = sum over the +-m rows of g_out * rotated).
template <int L, int M, int C, int MODE, bool RAD>
__global__ void __launch_bounds__(64) gather_rotate_kernel(const float* __restrict__ x_src, const float* __restrict__ x_dst,
                                     const int* __restrict__ src, const int* __restrict__ dst,
                                     const float* __restrict__ wr, const float* __restrict__ rad,
                                     const float* __restrict__ g_out, float* __restrict__ out, int E) {
    using I = SO3Idx<L, M>;
    const int lane = threadIdx.x;
    const bool act = lane < 2 * C;       // lanes >= 2C shadow lanes 0..2C-1 (same loads, no stores): every lane of the
    const int ln = lane & (2 * C - 1);   // wave must stay live for the Wigner-record fetch and its lane broadcasts
    // (edge = blockIdx + k * gridDim: contiguous runs of edges per workgroup with contiguous run ranges per XCD were tried
    // for L2 locality of the node rows and were SLOWER, 131 vs 118 us at L = 4 - the rows sit in the Infinity Cache either way)
    for (int e = blockIdx.x; e < E; e += gridDim.x) {
        WRows<I::WSZ> W;
        W.load(wr + (long long)e * I::WSZ, lane);
        const int node = ln < C ? src[e] : dst[e];
        const float* xin = (ln < C ? x_src : x_dst) + (long long)node * I::K * C + (ln & (C - 1));
        const long long eo = (long long)e * I::KR * 2 * C + ln;
        const long long ro = (long long)e * I::RAD_ROWS * 2 * C + ln;
        // every load of the edge is issued before the first use: the node rows (K of them), the radial weights (forward)
        // or the incoming gradient rows (MODE 1).  RAD is a template flag - a run-time `rad ? .. : 1` per row made the
        // compiler interleave load / wait / branch row by row (148 waits for 47 loads: 0.38 of the HBM roofline).
        float xv[I::K], rv[I::RAD_ROWS], gv[MODE == 1 ? I::KR : 1];
#pragma unroll
        for (int k = 0; k < I::K; ++k) xv[k] = xin[k * C];
        if (MODE == 0 && RAD) {
#pragma unroll
            for (int i = 0; i < I::RAD_ROWS; ++i) rv[i] = rad[ro + i * 2 * C];
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < I::KR; ++r) gv[r] = g_out[eo + r * 2 * C];
#pragma unroll
            for (int i = 0; i < I::RAD_ROWS; ++i) rv[i] = 0.f;
        }
#pragma unroll
        for (int l = 0; l <= L; ++l) {
#pragma unroll
            for (int mi = 0; mi < I::nr(l); ++mi) {
                const int m = mi - I::mm(l);
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 2 * l + 1; ++j) acc = fmaf(W_AT(W, I::w_off(l) + mi * (2 * l + 1) + j), xv[l * l + j], acc);
                if (MODE == 0) {
                    if (RAD) acc *= rv[I::rad_row(l, m)];
                    if (act) out[eo + I::mpos(l, m) * 2 * C] = acc;
                } else {
                    rv[I::rad_row(l, m)] = fmaf(gv[I::mpos(l, m)], acc, rv[I::rad_row(l, m)]);
                }
            }
        }
        if (MODE == 1 && act) {
#pragma unroll
            for (int i = 0; i < I::RAD_ROWS; ++i) out[ro + i * 2 * C] = rv[i];
        }
    }
}

// Backward of gather+rotate w.r.t. the node tensors.  A wavefront takes FOUR nodes: lane group q = lane / 16 walks the edge
// segment of node 4 w + q, lane % 16 = channel, so all 64 lanes work (the first version gave a whole wavefront to one node
// and kept 16 of its 64 lanes busy: 0.12 of the HBM roofline).  The four groups are on four different edges, so the Wigner
// record is no longer wave-uniform: every lane reads its edge's record itself, 16 bytes at a time (the 16 lanes of a group
// read the same addresses - one L1 transaction), instead of the readlane broadcasts of the edge-parallel kernels.
// SIDE 0: destination nodes, edges row_ptr[n]..row_ptr[n+1] (already contiguous).  SIDE 1: source nodes, edge ids
// eperm[col_ptr[n]..col_ptr[n+1]].  gx[n, l^2+j, c] = sum_e sum_r W_e[r][j] * g_out[e, r, side*C + c] * rad[e, r, ..].
// Records are WSZ floats long, WSZ a multiple of 4 (so3_index.h): every float4 load is aligned and stays inside the record.
struct __attribute__((aligned(16))) F4U {
    float v[4];
};
template <int L, int M, int C, int SIDE, bool RAD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((L > 4 ? 1 : 4), (L > 4 ? 2 : 8)))) gather_rotate_bwd_node_kernel(const float* __restrict__ g_out, const float* __restrict__ wr,
                                              const float* __restrict__ rad, const int* __restrict__ ptr,
                                              const int* __restrict__ eperm, float* __restrict__ gx, int N) {
    using I = SO3Idx<L, M>;
    static_assert(C == 16, "lane group = 16 channels");
    const int lane = threadIdx.x & 63, grp = lane >> 4, c = lane & 15;
    const int co = SIDE == 0 ? C + c : c;                             // dst half of the 2C channels is [C, 2C)
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
    for (int n0 = wave * 4; n0 < N; n0 += nwaves * 4) {
        const int n = n0 + grp;
        float acc[I::K];
#pragma unroll
        for (int k = 0; k < I::K; ++k) acc[k] = 0.f;
        const int beg = n < N ? ptr[n] : 0, end = n < N ? ptr[n + 1] : 0;
        int len = 0;                                                   // longest of the wavefront's four segments
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nq = n0 + q;
            const int lq = nq < N ? ptr[nq + 1] - ptr[nq] : 0;
            len = lq > len ? lq : len;
        }
        for (int i = 0; i < len; ++i) {
            const bool on = beg + i < end;
            const int slot = on ? beg + i : (end > beg ? end - 1 : 0);  // an idle group re-reads a valid edge with weight 0
            const int e = (end > beg) ? (SIDE == 0 ? slot : eperm[slot]) : 0;
            const float live = on ? 1.f : 0.f;
            const float* wp = wr + (long long)e * I::WSZ;
            const long long eo = (long long)e * I::KR * 2 * C + co;
            const long long ro = (long long)e * I::RAD_ROWS * 2 * C + co;
#pragma unroll
            for (int l = 0; l <= L; ++l) {
                // this degree's block of the record, [w_off(l), w_off(l+1)), fetched as the float4 chunks that cover it.
                // L = 6: the scheduler may not hoist the loads of degrees 4..6 above the arithmetic of degrees 0..3 (all
                // 235 + 47 loads in flight at once needed 350 registers and spilled)
                if (L > 4 && l == 4) __builtin_amdgcn_sched_barrier(0);
                const int lo4 = I::w_off(l) / 4, hi4 = (I::w_off(l + 1) + 3) / 4;
                float w[(2 * L + 1) * (2 * M + 1) + 6];
#pragma unroll
                for (int q = lo4; q < hi4; ++q) {
                    const F4U t = *reinterpret_cast<const F4U*>(wp + 4 * q);
                    w[4 * (q - lo4)] = t.v[0]; w[4 * (q - lo4) + 1] = t.v[1]; w[4 * (q - lo4) + 2] = t.v[2]; w[4 * (q - lo4) + 3] = t.v[3];
                }
                const int sh = I::w_off(l) - 4 * lo4;
                // all loads of the block first (RAD is a template flag: a run-time `if (rad)` per row made the compiler
                // serialise the row loads - load, wait, branch, load, wait: ~40 memory round trips per edge)
                float gv[2 * M + 1];
#pragma unroll
                for (int mi = 0; mi < I::nr(l); ++mi) {
                    const int m = mi - I::mm(l);
                    gv[mi] = g_out[eo + I::mpos(l, m) * 2 * C] * live;
                    if (RAD) gv[mi] *= rad[ro + I::rad_row(l, m) * 2 * C];
                }
#pragma unroll
                for (int mi = 0; mi < I::nr(l); ++mi) {
#pragma unroll
                    for (int j = 0; j < 2 * l + 1; ++j)
                        acc[l * l + j] = fmaf(w[sh + mi * (2 * l + 1) + j], gv[mi], acc[l * l + j]);
                }
            }
        }
        if (n < N) {
            float* o = gx + (long long)n * I::K * C + c;
#pragma unroll
            for (int k = 0; k < I::K; ++k) o[k * C] = acc[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------ k10: rotate back + scatter
__host__ __device__ constexpr float rescale_of(int l, int M) {
    // sqrt((2l+1)/(2M+1)) for l > M, M = 2: sqrt(7/5), sqrt(9/5), sqrt(11/5), sqrt(13/5), ... (EF:1545)
    return l <= M ? 1.0f
                  : (l == 3 ? 1.1832159566199232f
                            : (l == 4 ? 1.3416407864998738f
                                      : (l == 5 ? 1.4832396974191326f : (l == 6 ? 1.6124515496597098f : 0.f))));
}

// One workgroup per destination node; thread = channel (blockDim = 64 or 128 >= CH).  Walks the node's edge
// segment U edges at a time: all U*rows message loads of a chunk are issued before the first FMA (memory-level
// parallelism; the segment walk is otherwise a chain of dependent HBM round trips), then per edge
// m[r] = alpha * msg[e, r, c]; acc[l^2+j] += W_e[r][j] * m[r].  M0: only the m = 0 rows exist.
template <int L, int M, bool M0, int U, int CH, int VH>
__global__ void __launch_bounds__(128) rotate_back_scatter_kernel(Segs msg, const float* __restrict__ alpha, const float* __restrict__ wr,
                                           const int* __restrict__ row_ptr, float* __restrict__ out, int N,
                                           float out_scale) {
    using I = SO3Idx<L, M>;
    constexpr int NR = M0 ? L + 1 : I::KR;
    constexpr int heads = CH / VH;
    constexpr int r0 = L + 1, r01 = L + 1 + 2 * L;   // rows of the m = 0 and m = +-1 segments (checked by the host)
    const int act = threadIdx.x < CH;
    const int c = act ? threadIdx.x : CH - 1;   // surplus lanes shadow the last channel (they only fetch Wigner rows)
    const int lane = threadIdx.x & 63;

    // fetch one edge: Wigner record + its NR message rows (scaled by alpha; weight 0 for an out-of-range slot)
    auto fetch = [&](int e, bool ok, WRows<I::WSZ>& W, float (&v)[NR]) {
        W.load(wr + (long long)e * I::WSZ, lane);
        const float a = ok ? (alpha ? alpha[(long long)e * heads + c / VH] : 1.f) : 0.f;
        const float* b0 = msg.p[0] + (long long)e * msg.ld[0] + c;
        const float* b1 = M0 ? b0 : msg.p[1] + (long long)e * msg.ld[1] + c;
        const float* b2 = M0 ? b0 : msg.p[2] + (long long)e * msg.ld[2] + c;
#pragma unroll
        for (int l = 0; l <= L; ++l) {
#pragma unroll
            for (int mi = 0; mi < I::nr(l); ++mi) {
                const int m = mi - I::mm(l);
                if (M0 && m != 0) continue;
                const int q = I::mpos(l, m);        // compile-time: segment and offset are immediates
                const float x = q < r0 ? b0[q * CH] : (q < r01 ? b1[(q - r0) * CH] : b2[(q - r01) * CH]);
                v[M0 ? l : I::kr_off(l) + mi] = x * a;
            }
        }
    };
    auto rotate_add = [&](const WRows<I::WSZ>& W, const float (&v)[NR], float (&acc)[I::K]) {
#pragma unroll
        for (int l = 0; l <= L; ++l) {
#pragma unroll
            for (int mi = 0; mi < I::nr(l); ++mi) {
                const int m = mi - I::mm(l);
                if (M0 && m != 0) continue;
                const float x = v[M0 ? l : I::kr_off(l) + mi];
#pragma unroll
                for (int j = 0; j < 2 * l + 1; ++j)
                    acc[l * l + j] = fmaf(W_AT(W, I::w_off(l) + mi * (2 * l + 1) + j), x, acc[l * l + j]);
            }
        }
    };

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        float acc[I::K];
#pragma unroll
        for (int k = 0; k < I::K; ++k) acc[k] = 0.f;
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        if (U > 0) {
            // chunked walk: all loads of U edges are in flight before the first FMA
            for (int e0 = beg; e0 < end; e0 += (U > 0 ? U : 1)) {
                float v[U > 0 ? U : 1][NR];
                WRows<I::WSZ> W[U > 0 ? U : 1];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = e0 + u < end;
                    fetch(ok ? e0 + u : end - 1, ok, W[u], v[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) rotate_add(W[u], v[u], acc);
            }
        } else if (beg < end) {
            // software pipeline (large L: the rotation of one edge is ~500 instructions): the next edge's loads are
            // issued before the current edge is rotated, so HBM latency hides under the FMAs
            float vc[NR], vn[NR];
            WRows<I::WSZ> Wc, Wn;
            fetch(beg, true, Wc, vc);
            for (int e = beg; e < end; ++e) {
                const bool more = e + 1 < end;
                fetch(more ? e + 1 : e, more, Wn, vn);
                rotate_add(Wc, vc, acc);
#pragma unroll
                for (int i = 0; i < NR; ++i) vc[i] = vn[i];
                Wc = Wn;
            }
        }
        if (act) {
            float* o = out + (long long)n * I::K * CH + c;
#pragma unroll
            for (int l = 0; l <= L; ++l) {
#pragma unroll
                for (int j = 0; j < 2 * l + 1; ++j)
                    o[(long long)(l * l + j) * CH] = acc[l * l + j] * (rescale_of(l, M) * out_scale);
            }
        }
    }
}

// Backward: g[k] = g_out[n,k,c] * rescale * out_scale once per node; per edge t[r] = sum_j W_e[r][j] g[l^2+j];
// g_msg[e,r,c] = alpha * t[r]; g_alpha_part[e,c] = sum_r msg[e,r,c] * t[r].
template <int L, int M, bool M0, int CH, int VH>
__global__ void __launch_bounds__(128) rotate_back_scatter_bwd_kernel(const float* __restrict__ g_out, Segs msg, SegsMut gmsg,
                                               const float* __restrict__ alpha, const float* __restrict__ wr,
                                               const int* __restrict__ row_ptr, float* __restrict__ g_alpha_part, int N,
                                               float out_scale) {
    using I = SO3Idx<L, M>;
    constexpr int NR = M0 ? L + 1 : I::KR;
    constexpr int heads = CH / VH;
    constexpr int r0 = L + 1, r01 = L + 1 + 2 * L;
    const bool act = threadIdx.x < CH;
    const int c = act ? threadIdx.x : CH - 1;   // surplus lanes shadow the last channel and never store
    const int lane = threadIdx.x & 63;

    // per-edge inputs: Wigner record, alpha, and (for d/d alpha) the forward message rows
    auto fetch = [&](int e, WRows<I::WSZ>& W, float& a, float (&mv)[NR]) {
        W.load(wr + (long long)e * I::WSZ, lane);
        a = alpha ? alpha[(long long)e * heads + c / VH] : 1.f;
        if (alpha) {
            const float* b0 = msg.p[0] + (long long)e * msg.ld[0] + c;
            const float* b1 = M0 ? b0 : msg.p[1] + (long long)e * msg.ld[1] + c;
            const float* b2 = M0 ? b0 : msg.p[2] + (long long)e * msg.ld[2] + c;
#pragma unroll
            for (int l = 0; l <= L; ++l) {
#pragma unroll
                for (int mi = 0; mi < I::nr(l); ++mi) {
                    const int m = mi - I::mm(l);
                    if (M0 && m != 0) continue;
                    const int q = I::mpos(l, m);
                    mv[M0 ? l : I::kr_off(l) + mi] = q < r0 ? b0[q * CH] : (q < r01 ? b1[(q - r0) * CH] : b2[(q - r01) * CH]);
                }
            }
        }
    };

    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        float g[I::K];
        const float* gi = g_out + (long long)n * I::K * CH + c;
#pragma unroll
        for (int l = 0; l <= L; ++l) {
#pragma unroll
            for (int j = 0; j < 2 * l + 1; ++j) g[l * l + j] = gi[(long long)(l * l + j) * CH] * (rescale_of(l, M) * out_scale);
        }
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        if (beg >= end) continue;
        WRows<I::WSZ> Wc, Wn;
        float ac, an, mc[NR], mn[NR];
        fetch(beg, Wc, ac, mc);
        for (int e = beg; e < end; ++e) {
            fetch(e + 1 < end ? e + 1 : e, Wn, an, mn);      // next edge's loads fly while this edge is rotated
            float* o0 = gmsg.p[0] + (long long)e * gmsg.ld[0] + c;
            float* o1 = M0 ? o0 : gmsg.p[1] + (long long)e * gmsg.ld[1] + c;
            float* o2 = M0 ? o0 : gmsg.p[2] + (long long)e * gmsg.ld[2] + c;
            float part = 0.f;
#pragma unroll
            for (int l = 0; l <= L; ++l) {
#pragma unroll
                for (int mi = 0; mi < I::nr(l); ++mi) {
                    const int m = mi - I::mm(l);
                    if (M0 && m != 0) continue;
                    const int q = I::mpos(l, m);
                    float t = 0.f;
#pragma unroll
                    for (int j = 0; j < 2 * l + 1; ++j) t = fmaf(W_AT(Wc, I::w_off(l) + mi * (2 * l + 1) + j), g[l * l + j], t);
                    if (act) {
                        if (q < r0) o0[q * CH] = ac * t;
                        else if (q < r01) o1[(q - r0) * CH] = ac * t;
                        else o2[(q - r01) * CH] = ac * t;
                    }
                    if (alpha) part = fmaf(mc[M0 ? l : I::kr_off(l) + mi], t, part);
                }
            }
            if (alpha && act) g_alpha_part[(long long)e * CH + c] = part;
            Wc = Wn;
            ac = an;
#pragma unroll
            for (int i = 0; i < NR; ++i) mc[i] = mn[i];
        }
    }
}


// ------------------------------------------------------------------------------------------------ k9a: attention logits
// x0_alpha.view(E, heads, 32) -> LayerNorm(32) -> SmoothLeakyReLU(0.2) -> dot with alpha_dot[heads, 32]  (EF:1175-1178).
// 32 lanes = the 32 alpha channels of one (edge, head); a wavefront works on two edges at a time and loops over the heads.
// Statistics and the dot product are 5-step butterflies inside each 32-lane half.
__device__ __forceinline__ float half_sum(float v) {
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float smooth_leaky(float a) { return 0.6f * a + 0.4f * a * (2.0f * SINGA_RCP(1.0f + __expf(-a)) - 1.0f); }
__device__ __forceinline__ float smooth_leaky_grad(float a) {
    const float sg = SINGA_RCP(1.0f + __expf(-a));
    return 0.2f + 0.8f * sg + 0.8f * a * sg * (1.0f - sg);
}

template <int HEADS>
__global__ void __launch_bounds__(256) alpha_logits_fwd_kernel(const float* __restrict__ h0, long long ld,
                                                               const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                               const float* __restrict__ dot, float* __restrict__ logits, int E,
                                                               float eps) {
    const int k = threadIdx.x & 31;
    const long long slot = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const long long nslots = ((long long)gridDim.x * blockDim.x) >> 5;
    const float w = ln_w[k], b = ln_b[k];
    for (long long e = slot; e < E; e += nslots) {
        const float* x = h0 + e * ld + k;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            const float v = x[h * 32];
            const float mu = half_sum(v) * (1.0f / 32.0f);
            const float d = v - mu;
            const float rstd = rsqrtf(half_sum(d * d) * (1.0f / 32.0f) + eps);
            const float a = d * rstd * w + b;
            const float lg = half_sum(smooth_leaky(a) * dot[h * 32 + k]);
            if (k == 0) logits[e * HEADS + h] = lg;
        }
    }
}

// Backward (recompute): g_x [E, HEADS*32] and per-slot partial parameter gradients part[slot][(2 + HEADS) * 32] =
// [d ln_w | d ln_b | d alpha_dot[h, :]], reduced afterwards by singa_colsum.
template <int HEADS>
__global__ void __launch_bounds__(256) alpha_logits_bwd_kernel(const float* __restrict__ h0, long long ld,
                                                               const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                               const float* __restrict__ dot, const float* __restrict__ g_logits,
                                                               float* __restrict__ g_x, long long ld_gx,
                                                               float* __restrict__ part, int E, float eps) {
    const int k = threadIdx.x & 31;
    const long long slot = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const long long nslots = ((long long)gridDim.x * blockDim.x) >> 5;
    const float w = ln_w[k], b = ln_b[k];
    float gw = 0.f, gb = 0.f, gd[HEADS];
#pragma unroll
    for (int h = 0; h < HEADS; ++h) gd[h] = 0.f;
    for (long long e = slot; e < E; e += nslots) {
        const float* x = h0 + e * ld + k;
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            const float v = x[h * 32];
            const float mu = half_sum(v) * (1.0f / 32.0f);
            const float d = v - mu;
            const float rstd = rsqrtf(half_sum(d * d) * (1.0f / 32.0f) + eps);
            const float xh = d * rstd;
            const float a = xh * w + b;
            const float gl = g_logits[e * HEADS + h];
            gd[h] = fmaf(gl, smooth_leaky(a), gd[h]);
            const float da = gl * dot[h * 32 + k] * smooth_leaky_grad(a);
            gw = fmaf(da, xh, gw);
            gb += da;
            const float dxh = da * w;
            const float m1 = half_sum(dxh) * (1.0f / 32.0f);
            const float m2 = half_sum(dxh * xh) * (1.0f / 32.0f);
            g_x[e * ld_gx + h * 32 + k] = rstd * (dxh - m1 - xh * m2);
        }
    }
    float* pp = part + slot * ((2 + HEADS) * 32) + k;
    pp[0] = gw;
    pp[32] = gb;
#pragma unroll
    for (int h = 0; h < HEADS; ++h) pp[(2 + h) * 32] = gd[h];
}

// ------------------------------------------------------------------------------------------------ k9: segment softmax
__global__ void segment_softmax_fwd_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr,
                                           float* __restrict__ y, int N, int H, float eps) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= N * H) return;
    int n = tid / H, h = tid - n * H;
    int beg = row_ptr[n], end = row_ptr[n + 1];
    if (beg >= end) return;
    float mx = -INFINITY;
    for (int e = beg; e < end; ++e) mx = fmaxf(mx, x[(long long)e * H + h]);
    float s = 0.f;
    for (int e = beg; e < end; ++e) s += expf(x[(long long)e * H + h] - mx);
    float inv = 1.0f / (s + eps);
    for (int e = beg; e < end; ++e) y[(long long)e * H + h] = expf(x[(long long)e * H + h] - mx) * inv;
}

// y = ex / (S + eps);  dx = y * (gy - sum_seg(gy * y))   (exact for eps = 0; for eps = 1e-16 the ratio S/(S+eps)
// differs from 1 by < 1e-16 since S >= 1, below fp32 resolution).
__global__ void segment_softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                           const int* __restrict__ row_ptr, float* __restrict__ gx, int N, int H) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= N * H) return;
    int n = tid / H, h = tid - n * H;
    int beg = row_ptr[n], end = row_ptr[n + 1];
    float dot = 0.f;
    for (int e = beg; e < end; ++e) dot = fmaf(gy[(long long)e * H + h], y[(long long)e * H + h], dot);
    for (int e = beg; e < end; ++e) {
        long long i = (long long)e * H + h;
        gx[i] = y[i] * (gy[i] - dot);
    }
}

// Dense segments (the kNN graphs of the CProMG encoders: ~58 edges per node, 4 heads): one wavefront per node, one edge
// per lane, the four heads of an edge as one float4 - the per-thread walk above is 3 x 58 dependent loads long.
__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void __launch_bounds__(256) segment_softmax4_fwd_kernel(const float* __restrict__ x, const int* __restrict__ row_ptr,
                                                                   float* __restrict__ y, int N, float eps) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int beg = row_ptr[n], end = row_ptr[n + 1];
    if (beg >= end) return;
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    for (int e = beg + lane; e < end; e += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + (long long)e * 4);
        m0 = fmaxf(m0, v.x), m1 = fmaxf(m1, v.y), m2 = fmaxf(m2, v.z), m3 = fmaxf(m3, v.w);
    }
    m0 = wmax64(m0), m1 = wmax64(m1), m2 = wmax64(m2), m3 = wmax64(m3);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int e = beg + lane; e < end; e += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + (long long)e * 4);
        s0 += expf(v.x - m0), s1 += expf(v.y - m1), s2 += expf(v.z - m2), s3 += expf(v.w - m3);
    }
    s0 = 1.f / (wsum64(s0) + eps), s1 = 1.f / (wsum64(s1) + eps), s2 = 1.f / (wsum64(s2) + eps), s3 = 1.f / (wsum64(s3) + eps);
    for (int e = beg + lane; e < end; e += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + (long long)e * 4);
        *reinterpret_cast<float4*>(y + (long long)e * 4) =
            make_float4(expf(v.x - m0) * s0, expf(v.y - m1) * s1, expf(v.z - m2) * s2, expf(v.w - m3) * s3);
    }
}

__global__ void __launch_bounds__(256) segment_softmax4_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy,
                                                                   const int* __restrict__ row_ptr, float* __restrict__ gx, int N) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int beg = row_ptr[n], end = row_ptr[n + 1];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int e = beg + lane; e < end; e += 64) {
        const float4 a = *reinterpret_cast<const float4*>(y + (long long)e * 4);
        const float4 g = *reinterpret_cast<const float4*>(gy + (long long)e * 4);
        d0 = fmaf(g.x, a.x, d0), d1 = fmaf(g.y, a.y, d1), d2 = fmaf(g.z, a.z, d2), d3 = fmaf(g.w, a.w, d3);
    }
    d0 = wsum64(d0), d1 = wsum64(d1), d2 = wsum64(d2), d3 = wsum64(d3);
    for (int e = beg + lane; e < end; e += 64) {
        const float4 a = *reinterpret_cast<const float4*>(y + (long long)e * 4);
        const float4 g = *reinterpret_cast<const float4*>(gy + (long long)e * 4);
        *reinterpret_cast<float4*>(gx + (long long)e * 4) =
            make_float4(a.x * (g.x - d0), a.y * (g.y - d1), a.z * (g.z - d2), a.w * (g.w - d3));
    }
}

// ------------------------------------------------------------------------------------------------ k15: weighted segment sum
// out[n, t] = sum_e w[e, t / F] * v[e, t];  one workgroup per node, thread t in [0, H*F).
__global__ void segment_wsum_fwd_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                        const int* __restrict__ row_ptr, float* __restrict__ out, int N, int H, int F) {
    const int t = threadIdx.x;
    const int HF = H * F;
    if (t >= HF) return;
    const int h = t / F;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int beg = row_ptr[n], end = row_ptr[n + 1];
        float acc = 0.f;
        for (int e = beg; e < end; ++e) acc = fmaf(w[(long long)e * H + h], v[(long long)e * HF + t], acc);
        out[(long long)n * HF + t] = acc;
    }
}

// gv[e,t] = w[e,h] * g[n,t];  gw[e,h] = sum_{t in head h} v[e,t] * g[n,t]  (F = 64: one wavefront per head, reduced
// with DPP/shuffle butterflies).
__global__ void segment_wsum_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ w,
                                        const float* __restrict__ v, const int* __restrict__ row_ptr,
                                        float* __restrict__ gw, float* __restrict__ gv, int N, int H, int F) {
    const int t = threadIdx.x;
    const int HF = H * F;
    const int h = t / F;
    const bool act = t < HF;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int beg = row_ptr[n], end = row_ptr[n + 1];
        float g = act ? g_out[(long long)n * HF + t] : 0.f;
        for (int e = beg; e < end; ++e) {
            float p = 0.f;
            if (act) {
                gv[(long long)e * HF + t] = w[(long long)e * H + h] * g;
                p = v[(long long)e * HF + t] * g;
            }
            // reduce p over the F lanes of this head (F is a power of two <= 64 and divides 64)
            for (int o = F >> 1; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            if (act && (t % F) == 0) gw[(long long)e * H + h] = p;
        }
    }
}


#define SINGA_XCD_NODE_LOOP(n, N)                                                       \
    for (int vb_ = blockIdx.x, n8_ = ((N) + 7) / 8 * 8; vb_ < n8_; vb_ += gridDim.x)     \
        if (const int n = xcd_range_id(vb_, n8_); n < (N))

// ------------------------------------------------------------------------------------------------ k15b: fused graph attention
// CProMG MultiHeadAttention (CP:59-74) with weight_k_lin / weight_v_lin hoisted to node level by linearity:
//   qk[e,h]   = scale * sum_d qp[row,h,d] * wk[e,d] * hk[col,h,d] + cterm[row,h]
//   out[n,h,f] = sum_e alpha[e,h] * wv[e,f] * hv[col_e,h,f]
// so that no [E,H,D] / [E,H,F] tensor is ever materialised.  Edges are sorted by row (the centre node).
__device__ __forceinline__ float group_sum(float v, int width) {  // butterfly over `width` consecutive lanes
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wavefront per centre node; lane = (slot, d) with D lanes per slot and 64/D edge slots; loop over heads.
// four per-head sums over the 32 lanes of an edge slot, folded: two exchange steps leave head g with lane group g
// (8 lanes each), three more finish the sum - 5 cross-lane steps instead of 20
__device__ __forceinline__ float quad_reduce4_half(float t0, float t1, float t2, float t3, int lane) {
    const bool b4 = lane & 16, b3 = lane & 8;
    const float a = (b4 ? t2 : t0) + __shfl_xor(b4 ? t0 : t2, 16, 64);
    const float b = (b4 ? t3 : t1) + __shfl_xor(b4 ? t1 : t3, 16, 64);
    float v = (b3 ? b : a) + __shfl_xor(b3 ? a : b, 8, 64);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;                                            // lanes of group g = (lane % 32) / 8: the complete sum of head g
}

template <int D, int H>
__global__ void __launch_bounds__(64) edge_logits_fwd_kernel(const float* __restrict__ qp, const float* __restrict__ wk,
                                                             const float* __restrict__ hk, const float* __restrict__ cterm,
                                                             const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                             float* __restrict__ qk, int N, float scale) {
    static_assert(D == 32 && H == 4, "lane mapping and the folded reduction are written for 32 key channels, 4 heads");
    constexpr int SL = 64 / D;
    const int lane = threadIdx.x, d = lane % D, slot = lane / D, grp = d >> 3;
    SINGA_XCD_NODE_LOOP(n, N) {
        float q[H];
#pragma unroll
        for (int h = 0; h < H; ++h) q[h] = qp[((long long)n * H + h) * D + d] * scale;
        const float ct = cterm[(long long)n * H + grp];
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        for (int e0 = beg; e0 < end; e0 += 2 * SL) {       // two rounds of SL edges in flight
            const int ea = e0 + slot, eb = e0 + SL + slot;
            const bool oka = ea < end, okb = eb < end;
            const int ja = oka ? col[ea] : 0, jb = okb ? col[eb] : 0;
            const float wa = oka ? wk[(long long)ea * D + d] : 0.f, wb = okb ? wk[(long long)eb * D + d] : 0.f;
            float ta[H], tb[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                ta[h] = q[h] * wa * hk[((long long)ja * H + h) * D + d];
                tb[h] = q[h] * wb * hk[((long long)jb * H + h) * D + d];
            }
            const float pa = quad_reduce4_half(ta[0], ta[1], ta[2], ta[3], lane);
            const float pb = quad_reduce4_half(tb[0], tb[1], tb[2], tb[3], lane);
            if ((d & 7) == 0) {
                if (oka) qk[(long long)ea * H + grp] = pa + ct;
                if (okb) qk[(long long)eb * H + grp] = pb + ct;
            }
        }
    }
}

// grads w.r.t. qp (and cterm) and wk: same walk.  g_qp[n,h,d] = scale * sum_e g[e,h] wk[e,d] hk[col,h,d];
// g_wk[e,d] = scale * sum_h g[e,h] qp[n,h,d] hk[col,h,d];  g_cterm[n,h] = sum_e g[e,h].
template <int D, int H>
__global__ void __launch_bounds__(64) edge_logits_bwd_row_kernel(const float* __restrict__ g, const float* __restrict__ qp,
                                                                 const float* __restrict__ wk, const float* __restrict__ hk,
                                                                 const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                                 float* __restrict__ g_qp, float* __restrict__ g_wk,
                                                                 float* __restrict__ g_cterm, int N, float scale) {
    constexpr int SL = 64 / D;
    const int lane = threadIdx.x, d = lane % D, slot = lane / D;
    SINGA_XCD_NODE_LOOP(n, N) {
        float q[H], aq[H], ac[H];
#pragma unroll
        for (int h = 0; h < H; ++h) {
            q[h] = qp[((long long)n * H + h) * D + d] * scale;
            aq[h] = 0.f;
            ac[h] = 0.f;
        }
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        for (int e0 = beg; e0 < end; e0 += SL) {
            const int e = e0 + slot;
            if (e < end) {
                const int j = col[e];
                const float w = wk[(long long)e * D + d];
                float gw = 0.f;
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const float ge = g[(long long)e * H + h];
                    const float k = hk[((long long)j * H + h) * D + d];
                    aq[h] = fmaf(ge * w, k, aq[h]);
                    gw = fmaf(ge * q[h], k, gw);
                    ac[h] += ge;
                }
                g_wk[(long long)e * D + d] = gw;
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float a = aq[h], c = ac[h];
            for (int o = D; o < 64; o <<= 1) {  // combine the edge slots
                a += __shfl_xor(a, o, 64);
                c += __shfl_xor(c, o, 64);
            }
            if (slot == 0) g_qp[((long long)n * H + h) * D + d] = a * scale;
            if (lane == 0) g_cterm[(long long)n * H + h] = c;
        }
    }
}

// grad w.r.t. hk, by neighbour node j over the edges that have col == j (eperm = edge ids sorted by col):
// g_hk[j,h,d] = scale * sum_e g[e,h] qp[row_e,h,d] wk[e,d].
template <int D, int H>
__global__ void __launch_bounds__(64) edge_logits_bwd_col_kernel(const float* __restrict__ g, const float* __restrict__ qp,
                                                                 const float* __restrict__ wk, const int* __restrict__ col_ptr,
                                                                 const int* __restrict__ eperm, const int* __restrict__ row,
                                                                 float* __restrict__ g_hk, int N, float scale) {
    constexpr int SL = 64 / D;
    const int lane = threadIdx.x, d = lane % D, slot = lane / D;
    SINGA_XCD_NODE_LOOP(j, N) {
        float acc[H];
#pragma unroll
        for (int h = 0; h < H; ++h) acc[h] = 0.f;
        const int beg = col_ptr[j], end = col_ptr[j + 1];
        for (int i0 = beg; i0 < end; i0 += SL) {
            const int i = i0 + slot;
            if (i < end) {
                const int e = eperm[i];
                const int n = row[e];
                const float w = wk[(long long)e * D + d];
#pragma unroll
                for (int h = 0; h < H; ++h)
                    acc[h] = fmaf(g[(long long)e * H + h] * w, qp[((long long)n * H + h) * D + d], acc[h]);
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            float a = acc[h];
            for (int o = D; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
            if (slot == 0) g_hk[((long long)j * H + h) * D + d] = a * scale;
        }
    }
}

// out[n,h,f] = sum_e alpha[e,h] * wv[e,f] * hv[col_e,h,f]; one wavefront per centre node, lane = f (F = 64).
template <int H>
__global__ void __launch_bounds__(64) gather_wsum_fwd_kernel(const float* __restrict__ alpha, const float* __restrict__ wv,
                                                             const float* __restrict__ hv, const int* __restrict__ row_ptr,
                                                             const int* __restrict__ col, float* __restrict__ out, int N) {
    constexpr int F = 64;
    const int f = threadIdx.x;
    SINGA_XCD_NODE_LOOP(n, N) {
        float acc[H];
#pragma unroll
        for (int h = 0; h < H; ++h) acc[h] = 0.f;
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        int e = beg;
        for (; e + 1 < end; e += 2) {                      // two edges in flight: both gathers are issued before either FMA chain
            const int j0 = col[e], j1 = col[e + 1];
            const float w0 = wv[(long long)e * F + f], w1 = wv[(long long)(e + 1) * F + f];
            float v0[H], v1[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                v0[h] = hv[((long long)j0 * H + h) * F + f];
                v1[h] = hv[((long long)j1 * H + h) * F + f];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                acc[h] = fmaf(alpha[(long long)e * H + h] * w0, v0[h], acc[h]);
                acc[h] = fmaf(alpha[(long long)(e + 1) * H + h] * w1, v1[h], acc[h]);
            }
        }
        if (e < end) {
            const int j = col[e];
            const float w = wv[(long long)e * F + f];
#pragma unroll
            for (int h = 0; h < H; ++h)
                acc[h] = fmaf(alpha[(long long)e * H + h] * w, hv[((long long)j * H + h) * F + f], acc[h]);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) out[((long long)n * H + h) * F + f] = acc[h];
    }
}

// g_alpha[e,h] = sum_f g[n,h,f] wv[e,f] hv[col,h,f];  g_wv[e,f] = sum_h g[n,h,f] alpha[e,h] hv[col,h,f].
// The four per-head sums over the 64 lanes are reduced together: two exchange steps fold the four values into the four
// lane quarters (quarter q ends up owning head q), four more steps finish the sum inside each quarter - 6 cross-lane
// steps per edge instead of 24.  Two edges are in flight per iteration, so the second edge's dependent gather
// (col -> hv row) is issued while the first is reduced.
__device__ __forceinline__ float quad_reduce4(float t0, float t1, float t2, float t3, int lane) {
    const bool hi32 = lane & 32, hi16 = lane & 16;
    // fold over lane bit 5: the lower half keeps heads 0,1, the upper half heads 2,3
    const float a = (hi32 ? t2 : t0) + __shfl_xor(hi32 ? t0 : t2, 32, 64);
    const float b = (hi32 ? t3 : t1) + __shfl_xor(hi32 ? t1 : t3, 32, 64);
    // fold over lane bit 4: quarters own heads 0,1,2,3 in lane order
    float v = (hi16 ? b : a) + __shfl_xor(hi16 ? a : b, 16, 64);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;                                            // lanes of quarter q: the complete sum of head q
}

template <int H>
__global__ void __launch_bounds__(64) gather_wsum_bwd_row_kernel(const float* __restrict__ g, const float* __restrict__ alpha,
                                                                 const float* __restrict__ wv, const float* __restrict__ hv,
                                                                 const int* __restrict__ row_ptr, const int* __restrict__ col,
                                                                 float* __restrict__ g_alpha, float* __restrict__ g_wv, int N) {
    static_assert(H == 4, "the four-way folded reduction is written for four heads");
    constexpr int F = 64;
    const int f = threadIdx.x, quarter = f >> 4;
    SINGA_XCD_NODE_LOOP(n, N) {
        float gn[H];
#pragma unroll
        for (int h = 0; h < H; ++h) gn[h] = g[((long long)n * H + h) * F + f];
        const int beg = row_ptr[n], end = row_ptr[n + 1];
        for (int e = beg; e < end; e += 2) {
            const bool two = e + 1 < end;
            const int e1 = two ? e + 1 : e;
            const int j0 = col[e], j1 = col[e1];
            const float w0 = wv[(long long)e * F + f], w1 = wv[(long long)e1 * F + f];
            float t0[H], t1[H], gw0 = 0.f, gw1 = 0.f;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                t0[h] = gn[h] * hv[((long long)j0 * H + h) * F + f];
                t1[h] = gn[h] * hv[((long long)j1 * H + h) * F + f];
            }
#pragma unroll
            for (int h = 0; h < H; ++h) {
                gw0 = fmaf(t0[h], alpha[(long long)e * H + h], gw0);
                gw1 = fmaf(t1[h], alpha[(long long)e1 * H + h], gw1);
            }
            const float p0 = quad_reduce4(t0[0] * w0, t0[1] * w0, t0[2] * w0, t0[3] * w0, f);
            const float p1 = quad_reduce4(t1[0] * w1, t1[1] * w1, t1[2] * w1, t1[3] * w1, f);
            if ((f & 15) == 0) {
                g_alpha[(long long)e * H + quarter] = p0;
                if (two) g_alpha[(long long)e1 * H + quarter] = p1;
            }
            g_wv[(long long)e * F + f] = gw0;
            if (two) g_wv[(long long)e1 * F + f] = gw1;
        }
    }
}

// g_hv[j,h,f] = sum_{e: col_e = j} alpha[e,h] wv[e,f] g[row_e,h,f].
template <int H>
__global__ void __launch_bounds__(64) gather_wsum_bwd_col_kernel(const float* __restrict__ g, const float* __restrict__ alpha,
                                                                 const float* __restrict__ wv, const int* __restrict__ col_ptr,
                                                                 const int* __restrict__ eperm, const int* __restrict__ row,
                                                                 float* __restrict__ g_hv, int N) {
    constexpr int F = 64;
    const int f = threadIdx.x;
    SINGA_XCD_NODE_LOOP(j, N) {
        float acc[H];
#pragma unroll
        for (int h = 0; h < H; ++h) acc[h] = 0.f;
        const int beg = col_ptr[j], end = col_ptr[j + 1];
        for (int i = beg; i < end; ++i) {
            const int e = eperm[i];
            const int n = row[e];
            const float w = wv[(long long)e * F + f];
#pragma unroll
            for (int h = 0; h < H; ++h)
                acc[h] = fmaf(alpha[(long long)e * H + h] * w, g[((long long)n * H + h) * F + f], acc[h]);
        }
#pragma unroll
        for (int h = 0; h < H; ++h) g_hv[((long long)j * H + h) * F + f] = acc[h];
    }
}


// ------------------------------------------------------------------------------------------------ narrow LayerNorm + SiLU
// The radial MLPs (EF:1634-1657) normalise rows of only C = 16 channels; a row-per-wave LayerNorm kernel spends its time
// on cross-lane reductions over 16 values.  Here one thread owns one row (64 B: consecutive lanes read consecutive rows),
// LayerNorm (biased variance, eps inside the root) and the SiLU that always follows it are one pass, and the backward
// recomputes the statistics from x.  d gamma / d beta are accumulated per thread over its rows and reduced by
// singa_colsum over the [threads, 2C] partials.
template <int C>
__device__ __forceinline__ void ln_row(const float* __restrict__ x, long long r, float (&xh)[C], float& rstd, float eps) {
    float mean = 0.f;
#pragma unroll
    for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
        xh[c] = v.x, xh[c + 1] = v.y, xh[c + 2] = v.z, xh[c + 3] = v.w;
        mean += (v.x + v.y) + (v.z + v.w);
    }
    mean *= 1.f / C;
    float var = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        xh[c] -= mean;
        var = fmaf(xh[c], xh[c], var);
    }
    rstd = 1.f / sqrtf(var * (1.f / C) + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) xh[c] *= rstd;
}

template <int C>
__global__ void __launch_bounds__(64) ln_silu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, long long M,
                                                         float eps) {
    const long long stride = (long long)gridDim.x * 64;
    for (long long r = (long long)blockIdx.x * 64 + threadIdx.x; r < M; r += stride) {
        float xh[C], rstd;
        ln_row<C>(x, r, xh, rstd, eps);
        float o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = fmaf(xh[c], gamma[c], beta[c]);
            o[c] = y / (1.f + expf(-y));
        }
#pragma unroll
        for (int c = 0; c < C; c += 4) *reinterpret_cast<float4*>(out + r * C + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    }
}

template <int C>
__global__ void __launch_bounds__(64) ln_silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ g_out,
                                                         float* __restrict__ g_x, float* __restrict__ part, long long M,
                                                         float eps) {
    const long long tid = (long long)blockIdx.x * 64 + threadIdx.x, stride = (long long)gridDim.x * 64;
    float ag[C], ab[C];
#pragma unroll
    for (int c = 0; c < C; ++c) ag[c] = ab[c] = 0.f;
    for (long long r = tid; r < M; r += stride) {
        float xh[C], rstd;
        ln_row<C>(x, r, xh, rstd, eps);
        float gh[C], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(g_out + r * C + c);
            gh[c] = v.x, gh[c + 1] = v.y, gh[c + 2] = v.z, gh[c + 3] = v.w;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float y = fmaf(xh[c], gamma[c], beta[c]);
            const float sg = 1.f / (1.f + expf(-y));
            const float gy = gh[c] * sg * (1.f + y * (1.f - sg));          // d SiLU
            ag[c] = fmaf(gy, xh[c], ag[c]);
            ab[c] += gy;
            gh[c] = gy * gamma[c];                                         // d x_hat
            m1 += gh[c];
            m2 = fmaf(gh[c], xh[c], m2);
        }
        m1 *= 1.f / C, m2 *= 1.f / C;
        float o[C];
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = rstd * (gh[c] - m1 - xh[c] * m2);
#pragma unroll
        for (int c = 0; c < C; c += 4) *reinterpret_cast<float4*>(g_x + r * C + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        part[tid * 2 * C + c] = ag[c];
        part[tid * 2 * C + C + c] = ab[c];
    }
}


// ------------------------------------------------------------------------------------------------ bias + shifted softplus
// y = softplus(u + b) - ln 2 on [M, n] rows (CP:41-48: the activation between the two Linears of the edge MLPs, with the
// first Linear's bias folded in so that its GEMM needs no epilogue) and gu = g * sigmoid(u + b).  float4 per thread.
__global__ void __launch_bounds__(256) bias_ssp_fwd_kernel(const float* __restrict__ u, const float* __restrict__ b,
                                                           float* __restrict__ y, long long total4, int n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)((i * 4) % n);
    const float4 v = *reinterpret_cast<const float4*>(u + i * 4);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    // softplus(x) = max(x, 0) + log(1 + exp(-|x|)): one fast exp and one fast log, absolute error < 1e-7
    auto f = [](float x) { return fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))) - 0.69314718055994530942f; };
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(f(v.x + bb.x), f(v.y + bb.y), f(v.z + bb.z), f(v.w + bb.w));
}

__global__ void __launch_bounds__(256) bias_ssp_bwd_kernel(const float* __restrict__ u, const float* __restrict__ b,
                                                           const float* __restrict__ g, float* __restrict__ gu,
                                                           long long total4, int n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)((i * 4) % n);
    const float4 v = *reinterpret_cast<const float4*>(u + i * 4);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    const float4 gg = *reinterpret_cast<const float4*>(g + i * 4);
    auto s = [](float x) { return SINGA_RCP(1.f + __expf(-x)); };
    *reinterpret_cast<float4*>(gu + i * 4) = make_float4(gg.x * s(v.x + bb.x), gg.y * s(v.y + bb.y), gg.z * s(v.z + bb.z),
                                                         gg.w * s(v.w + bb.w));
}


// ------------------------------------------------------------------------------------------------ LayerNorm(256) + residual
// y = LayerNorm(a + r) over rows of C = 256 channels (r optional): every residual LayerNorm of the CProMG transformer
// (CP:78, 105, 158, 176, 191, 264).  One wavefront per row, four consecutive channels per lane; the sum a + r is never
// written.  The backward recomputes the statistics from a (+ r), returns the gradient of the sum (it is the gradient of
// both summands) and per-wave partial sums [waves][2C] = [d gamma | d beta] for singa_colsum.
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void ln256_row(const float* __restrict__ a, const float* __restrict__ r, long long row, int lane,
                                          float (&xh)[4], float& rstd, float eps) {
    float4 v = *reinterpret_cast<const float4*>(a + row * 256 + lane * 4);
    if (r) {
        const float4 w = *reinterpret_cast<const float4*>(r + row * 256 + lane * 4);
        v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
    }
    const float mean = wave_sum64((v.x + v.y) + (v.z + v.w)) * (1.f / 256);
    xh[0] = v.x - mean, xh[1] = v.y - mean, xh[2] = v.z - mean, xh[3] = v.w - mean;
    const float var = wave_sum64((xh[0] * xh[0] + xh[1] * xh[1]) + (xh[2] * xh[2] + xh[3] * xh[3])) * (1.f / 256);
    rstd = 1.f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) xh[i] *= rstd;
}

__global__ void __launch_bounds__(256) ln256_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, long long M, float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const float4 gm = *reinterpret_cast<const float4*>(gamma + lane * 4), bt = *reinterpret_cast<const float4*>(beta + lane * 4);
    for (long long row = wave; row < M; row += nwaves) {
        float xh[4], rstd;
        ln256_row(a, r, row, lane, xh, rstd, eps);
        *reinterpret_cast<float4*>(y + row * 256 + lane * 4) =
            make_float4(fmaf(xh[0], gm.x, bt.x), fmaf(xh[1], gm.y, bt.y), fmaf(xh[2], gm.z, bt.z), fmaf(xh[3], gm.w, bt.w));
    }
}

__global__ void __launch_bounds__(256) ln256_bwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                        const float* __restrict__ gamma, const float* __restrict__ g,
                                                        float* __restrict__ gs, float* __restrict__ part, long long M,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const float4 gm4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const float gm[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long row = wave; row < M; row += nwaves) {
        float xh[4], rstd;
        ln256_row(a, r, row, lane, xh, rstd, eps);
        const float4 g4 = *reinterpret_cast<const float4*>(g + row * 256 + lane * 4);
        float gh[4] = {g4.x, g4.y, g4.z, g4.w};
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ag[i] = fmaf(gh[i], xh[i], ag[i]);
            ab[i] += gh[i];
            gh[i] *= gm[i];
            s1 += gh[i];
            s2 = fmaf(gh[i], xh[i], s2);
        }
        const float m1 = wave_sum64(s1) * (1.f / 256), m2 = wave_sum64(s2) * (1.f / 256);
        *reinterpret_cast<float4*>(gs + row * 256 + lane * 4) =
            make_float4(rstd * (gh[0] - m1 - xh[0] * m2), rstd * (gh[1] - m1 - xh[1] * m2), rstd * (gh[2] - m1 - xh[2] * m2),
                        rstd * (gh[3] - m1 - xh[3] * m2));
    }
    *reinterpret_cast<float4*>(part + wave * 512 + lane * 4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
    *reinterpret_cast<float4*>(part + wave * 512 + 256 + lane * 4) = make_float4(ab[0], ab[1], ab[2], ab[3]);
}


// ------------------------------------------------------------------------------------------------ incremental decoder
// One new position of the CProMG decoder (CP:134-191, 346-383) for every live row of a beam search (rows = proteins x
// beams, a few dozen): the three sub-blocks of a decoder layer as three kernels, one workgroup of 1024 threads per row.
// With ~20 rows every library GEMM of the step is a 20-row product that pays a full launch for microseconds of work
// (~200 launches per token).  Here the input row sits in LDS and the weights are read once per workgroup as TRANSPOSED
// matrices wT[in][out]: thread (col = t % 256, part = t / 256) accumulates its output channel(s) over a quarter of the
// input channels (64 lanes read 64 consecutive outputs; sixteen wavefronts keep enough loads in flight for a
// latency-bound 20-workgroup launch), the four partial sums meet in LDS in a fixed order.
// Hidden 256, 4 heads, 32 key / 64 value channels per head, FFN 1024 - the shipped configuration.
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sum over the first four wavefronts (threads 0..255) of a 1024-thread workgroup; every thread calls it
__device__ __forceinline__ float block_sum_first256(float v, float* red) {
    if (threadIdx.x < 256) {
        v = wave_sum64(v);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    }
    __syncthreads();
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
}

// LayerNorm(256) of the row held one channel per thread by threads 0..255 (o); result valid in those threads
__device__ __forceinline__ float block_layer_norm256(float o, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* red, float eps) {
    const int c = threadIdx.x & 255;
    const float mean = block_sum_first256(o, red) * (1.f / 256);
    const float d = o - mean;
    const float var = block_sum_first256(d * d, red) * (1.f / 256);
    return d * (1.f / sqrtf(var + eps)) * gamma[c] + beta[c];
}

// partial[part][j*256 + col] = sum over k in this thread's quarter of in[k] * wT[k*N + j*256 + col], j < NJ (N = 256*NJ)
template <int NJ>
__device__ __forceinline__ void gemv_quarter(const float* in, int K, const float* __restrict__ wT, float* partial) {
    const int col = threadIdx.x & 255, part = threadIdx.x >> 8, kq = K / 4;
    constexpr int N = 256 * NJ;
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    const float* w = wT + (long long)part * kq * N + col;
    const float* x = in + part * kq;
#pragma unroll 8
    for (int k = 0; k < kq; ++k) {
        const float xv = x[k];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = fmaf(xv, w[(long long)k * N + j * 256], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) partial[part * N + j * 256 + col] = acc[j];
}

__device__ __forceinline__ float sum4(const float* partial, int N, int c) {
    return (partial[c] + partial[N + c]) + (partial[2 * N + c] + partial[3 * N + c]);
}

// self-attention sub-block: q,k,v projections of the new position, k/v appended to the caches at `pos`, attention of
// each head (wavefront h of the first four) over positions 0..pos, output projection + residual + LayerNorm.
__global__ void __launch_bounds__(1024) dec_self_attn_kernel(const float* __restrict__ x, const float* __restrict__ wqkv_t,
                                                             const float* __restrict__ bqkv, const float* __restrict__ wo_t,
                                                             const float* __restrict__ bo, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ kc,
                                                             float* __restrict__ vc, const long long* __restrict__ pos_ptr, int P,
                                                             float* __restrict__ y, float eps) {
    __shared__ float xs[256], qkv[512], ps[4][256], ctx[256], partial[4 * 512], red[4];
    const int r = blockIdx.x, t = threadIdx.x, c = t & 255, part = t >> 8, h = c >> 6, lane = c & 63;
    const int pos = (int)pos_ptr[0];
    if (t < 256) xs[t] = x[(long long)r * 256 + t];
    __syncthreads();
    gemv_quarter<2>(xs, 256, wqkv_t, partial);
    __syncthreads();
    if (t < 512) {
        const float v = bqkv[t] + sum4(partial, 512, t);              // [0,128): q   [128,256): k   [256,512): v
        qkv[t] = v;
        if (t >= 256) vc[(((long long)r * 4 + (t - 256) / 64) * P + pos) * 64 + (t - 256) % 64] = v;
        else if (t >= 128) kc[(((long long)r * 4 + (t - 128) / 32) * P + pos) * 32 + (t - 128) % 32] = v;
    }
    __syncthreads();
    if (t < 256) {
        const float* q = qkv + h * 32;
        const float* krow = kc + ((long long)r * 4 + h) * P * 32;
        float sc[4], mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int tt = lane + 64 * m;
            float sv = -INFINITY;
            if (tt <= pos) {
                const float* kr = tt == pos ? qkv + 128 + h * 32 : krow + (long long)tt * 32;
                float acc = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) acc = fmaf(q[d], kr[d], acc);
                sv = acc * 0.17677669529663687f;       // 1 / sqrt(32)
            }
            sc[m] = sv;
            mx = fmaxf(mx, sv);
        }
        mx = wave_max64(mx);
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            sc[m] = lane + 64 * m <= pos ? __expf(sc[m] - mx) : 0.f;
            sum += sc[m];
        }
        sum = 1.f / wave_sum64(sum);
#pragma unroll
        for (int m = 0; m < 4; ++m) ps[h][lane + 64 * m] = sc[m] * sum;
    }
    __syncthreads();
    {   // context: quarter `part` of the cached positions per thread, the four partial sums meet in LDS
        const float* vrow = vc + ((long long)r * 4 + h) * P * 64;
        float a = part == 0 ? ps[h][pos] * qkv[256 + c] : 0.f;
#pragma unroll 4
        for (int tt = part; tt < pos; tt += 4) a = fmaf(ps[h][tt], vrow[(long long)tt * 64 + lane], a);
        partial[part * 256 + c] = a;
    }
    __syncthreads();
    if (t < 256) ctx[t] = sum4(partial, 256, t);
    __syncthreads();
    gemv_quarter<1>(ctx, 256, wo_t, partial);
    __syncthreads();
    const float o = t < 256 ? bo[t] + xs[t] + sum4(partial, 256, t) : 0.f;
    const float out = block_layer_norm256(o, gamma, beta, red, eps);
    if (t < 256) y[(long long)r * 256 + t] = out;
}

// encoder-decoder attention sub-block: keys ck[B][4][32][S] and values cv[B][4][S][64] were projected once per protein;
// row r belongs to protein r / beams; pad[B][S] != 0 marks padding positions (score -1e9, as masked_fill in CP:150).
constexpr int DEC_MAX_S = 1024;

__global__ void __launch_bounds__(1024) dec_cross_attn_kernel(const float* __restrict__ y, const float* __restrict__ wq_t,
                                                              const float* __restrict__ bq, const float* __restrict__ ck,
                                                              const float* __restrict__ cv, const unsigned char* __restrict__ pad,
                                                              const float* __restrict__ wo_t, const float* __restrict__ bo,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int beams, int S, float* __restrict__ z, float eps) {
    __shared__ float ys[256], qs[128], ps[4][DEC_MAX_S], ctx[256], partial[4 * 256], red[4], inv[4];
    const int r = blockIdx.x, t = threadIdx.x, c = t & 255, part = t >> 8, h = c >> 6, lane = c & 63;
    const long long b = r / beams;
    if (t < 256) ys[t] = y[(long long)r * 256 + t];
    __syncthreads();
    {   // q projection: 128 outputs, eight slices of 32 input channels
        const int col = t & 127, slice = t >> 7;
        float acc = 0.f;
#pragma unroll 8
        for (int i = slice * 32; i < slice * 32 + 32; ++i) acc = fmaf(ys[i], wq_t[i * 128 + col], acc);
        partial[slice * 128 + col] = acc;
    }
    __syncthreads();
    if (t < 128) {
        float acc = bq[t];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) acc += partial[s8 * 128 + t];
        qs[t] = acc;
    }
    __syncthreads();
    if (t < 256) {
        const float* kh = ck + (b * 4 + h) * 32 * (long long)S;
        float mx = -INFINITY;
        for (int s = lane; s < S; s += 64) {
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) acc = fmaf(qs[h * 32 + d], kh[(long long)d * S + s], acc);
            acc = pad[b * S + s] ? -1e9f : acc * 0.17677669529663687f;
            ps[h][s] = acc;
            mx = fmaxf(mx, acc);
        }
        mx = wave_max64(mx);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) {
            const float e = __expf(ps[h][s] - mx);
            ps[h][s] = e;
            sum += e;
        }
        sum = wave_sum64(sum);
        if (lane == 0) inv[h] = 1.f / sum;
    }
    __syncthreads();
    {
        const float* vh = cv + (b * 4 + h) * (long long)S * 64;
        float a = 0.f;
#pragma unroll 4
        for (int s = part; s < S; s += 4) a = fmaf(ps[h][s], vh[(long long)s * 64 + lane], a);
        partial[part * 256 + c] = a;
    }
    __syncthreads();
    if (t < 256) ctx[t] = sum4(partial, 256, t) * inv[h];
    __syncthreads();
    gemv_quarter<1>(ctx, 256, wo_t, partial);
    __syncthreads();
    const float o = t < 256 ? bo[t] + ys[t] + sum4(partial, 256, t) : 0.f;
    const float out = block_layer_norm256(o, gamma, beta, red, eps);
    if (t < 256) z[(long long)r * 256 + t] = out;
}

// position-wise feed-forward sub-block: 256 -> 1024 (ReLU) -> 256, residual, LayerNorm.
__global__ void __launch_bounds__(1024) dec_ffn_kernel(const float* __restrict__ z, const float* __restrict__ w1_t,
                                                       const float* __restrict__ b1, const float* __restrict__ w2_t,
                                                       const float* __restrict__ b2, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out, float eps) {
    __shared__ float zs[256], hs[1024], partial[4 * 1024], red[4];
    const int r = blockIdx.x, t = threadIdx.x;
    if (t < 256) zs[t] = z[(long long)r * 256 + t];
    __syncthreads();
    gemv_quarter<4>(zs, 256, w1_t, partial);
    __syncthreads();
    hs[t] = fmaxf(b1[t] + sum4(partial, 1024, t), 0.f);
    __syncthreads();
    gemv_quarter<1>(hs, 1024, w2_t, partial);
    __syncthreads();
    const float o = t < 256 ? b2[t] + zs[t] + sum4(partial, 256, t) : 0.f;
    const float res = block_layer_norm256(o, gamma, beta, red, eps);
    if (t < 256) out[(long long)r * 256 + t] = res;
}


// ------------------------------------------------------------------------------------------------ CProMG edge MLPs on MFMA
// W_k = L2k(ssp(L1k(attr))), W_v = L2v(ssp(L1v(attr))) for every kNN edge (CP:41-48, 58, 68): two chained GEMMs per net
// whose [E, hidden] intermediates cost ~2.4 GB of HBM traffic per encoder layer as library calls.  Here a wavefront takes
// 32 edges, the attribute rows are read once, and both GEMMs run on the f32 MFMA (v_mfma_f32_32x32x2_f32) without the
// hidden activations ever leaving registers:
//   GEMM 1 is computed TRANSPOSED, pre^T[hidden x edge] = W1 . attr^T: the accumulator then has the EDGE on the lane and
//   16 hidden units in its registers (unit 8(r/4) + 4(lane/32) + r%4 in register r), which is exactly the B-operand shape
//   of GEMM 2, out^T[out x edge] = W2 . h^T, if GEMM 2 walks its k index in that same permuted order - so step s of GEMM 2
//   takes register s of the activated accumulator as B and W2[:, 8(s/4) + 4(lane/32) + s%4] as A (read from LDS).
// Weights sit in LDS transposed (w1t[in][hidden], w2t[hidden][out]) so that the 32 lanes of an A operand read consecutive
// addresses.  Biases initialise the accumulators.
#ifndef SINGA_FLOATX16              // tests/emul supplies a plain struct so that the file still compiles with g++
typedef float floatx16 __attribute__((ext_vector_type(16)));
#else
typedef SINGA_FLOATX16 floatx16;
#endif

__device__ __forceinline__ float ssp_fast(float x) {
    return fmaxf(x, 0.f) + __logf(1.f + __expf(-fabsf(x))) - 0.69314718055994530942f;
}

// one net on one 32-edge tile.  asel[s] = attr[edge][32 * half + s] (GEMM 1 pairs input channel s of the lower lane half
// with channel 32 + s of the upper half in k-step s, so each half reads one contiguous 128-byte half row).  The weights sit
// in LDS as nn.Linear stores them, [out][in], with a row pitch of in + 4 floats: lane i then finds the A operands of FOUR
// consecutive k-steps in one 16-byte read (row i, four consecutive input channels; pitches 68 and 36 are conflict-free for
// ds_read_b128's 16-lane groups), and the read for the next four steps is issued ahead of the current four MFMAs.
// T = H / 32 hidden tiles.
constexpr int EMLP_P1 = 68;                                    // pitch of a [.][64] weight image
template <int H> struct EmlpP2 { static constexpr int v = H + 4; };   // pitch of a [.][H] weight image

template <int H>
__device__ __forceinline__ void edge_mlp_tile(const float (&asel)[32], const float* __restrict__ w1, const float* __restrict__ b1,
                                              const float* __restrict__ w2, const float* __restrict__ b2, int lane,
                                              floatx16 (&out)[H / 32]) {
    constexpr int T = H / 32, P2 = EmlpP2<H>::v;
    const int i = lane & 31, half = lane >> 5;
    floatx16 hacc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[t][r] = b1[32 * t + 8 * (r >> 2) + 4 * half + (r & 3)];
    {
        const float* a1 = w1 + i * EMLP_P1 + 32 * half;
        float4 fa[2][T];
#pragma unroll
        for (int t = 0; t < T; ++t) fa[0][t] = *reinterpret_cast<const float4*>(a1 + 32 * t * EMLP_P1);
        __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (q + 1 < 8) {
#pragma unroll
                for (int t = 0; t < T; ++t) fa[(q + 1) & 1][t] = *reinterpret_cast<const float4*>(a1 + 32 * t * EMLP_P1 + 4 * (q + 1));
                __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float4 f = fa[q & 1][t];
                    const float av = j == 0 ? f.x : j == 1 ? f.y : j == 2 ? f.z : f.w;
                    hacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, asel[4 * q + j], hacc[t], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * T, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[t][r] = ssp_fast(hacc[t][r]);
#pragma unroll
    for (int u = 0; u < T; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[u][r] = b2[32 * u + 8 * (r >> 2) + 4 * half + (r & 3)];
    {
        // k-step (t, s) takes hidden unit 32 t + 8 (s / 4) + 4 half + s % 4 (the unit register s of this lane half holds):
        // steps 4 g .. 4 g + 3 of tile t are four consecutive columns of W2
        const float* a2 = w2 + i * P2 + 4 * half;
        float4 fb[2][T];
#pragma unroll
        for (int u = 0; u < T; ++u) fb[0][u] = *reinterpret_cast<const float4*>(a2 + 32 * u * P2);
        __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
#pragma unroll
        for (int g = 0; g < 4 * T; ++g) {
            if (g + 1 < 4 * T) {
#pragma unroll
                for (int u = 0; u < T; ++u)
                    fb[(g + 1) & 1][u] = *reinterpret_cast<const float4*>(a2 + 32 * u * P2 + 32 * ((g + 1) >> 2) + 8 * ((g + 1) & 3));
                __builtin_amdgcn_sched_group_barrier(0x100, T, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < T; ++u) {
                    const float4 f = fb[g & 1][u];
                    const float av = j == 0 ? f.x : j == 1 ? f.y : j == 2 ? f.z : f.w;
                    out[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, hacc[g >> 2][4 * (g & 3) + j], out[u], 0, 0, 0);
                }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * T, 0);
        }
    }
}

// stage a weight matrix w[rows][cols] (as stored) into an LDS image of row pitch cols + 4, in two phases so that a
// workgroup's loads of all its matrices are in flight together (one load - wait - write per element costs a memory round
// trip each: ~25 us per workgroup for the 44 elements a thread stages in the forward kernel)
template <int ROWS, int COLS>
__device__ __forceinline__ void emlp_stage_load(float (&v)[ROWS * COLS / 256], const float* __restrict__ w) {
#pragma unroll
    for (int k = 0; k < ROWS * COLS / 256; ++k) v[k] = w[threadIdx.x + 256 * k];
}
template <int ROWS, int COLS>
__device__ __forceinline__ void emlp_stage_store(float* __restrict__ img, const float (&v)[ROWS * COLS / 256]) {
#pragma unroll
    for (int k = 0; k < ROWS * COLS / 256; ++k) {
        const int t = threadIdx.x + 256 * k;
        img[(t / COLS) * (COLS + 4) + (t % COLS)] = v[k];
    }
}

__global__ void __launch_bounds__(256, 2) edge_mlp_mfma_fwd_kernel(const float* __restrict__ attr, const float* __restrict__ w1tk,
                                                                const float* __restrict__ b1k, const float* __restrict__ w2tk,
                                                                const float* __restrict__ b2k, const float* __restrict__ w1tv,
                                                                const float* __restrict__ b1v, const float* __restrict__ w2tv,
                                                                const float* __restrict__ b2v, float* __restrict__ wk,
                                                                float* __restrict__ wv, int E) {
    __shared__ __attribute__((aligned(16))) float lw1k[32 * EMLP_P1], lw2k[32 * 36], lw1v[64 * EMLP_P1], lw2v[64 * 68];
    __shared__ float lb[32 + 32 + 64 + 64];
    {                                                      // the parameters arrive in nn.Linear's own [out][in] layout
        float s1k[8], s2k[4], s1v[16], s2v[16];
        emlp_stage_load<32, 64>(s1k, w1tk);
        emlp_stage_load<32, 32>(s2k, w2tk);
        emlp_stage_load<64, 64>(s1v, w1tv);
        emlp_stage_load<64, 64>(s2v, w2tv);
        const int tb = threadIdx.x & 63;
        const float bias = threadIdx.x < 64 ? (tb < 32 ? b1k[tb] : b2k[tb - 32]) : threadIdx.x < 128 ? b1v[tb] : threadIdx.x < 192 ? b2v[tb] : 0.f;
        emlp_stage_store<32, 64>(lw1k, s1k);
        emlp_stage_store<32, 32>(lw2k, s2k);
        emlp_stage_store<64, 64>(lw1v, s1v);
        emlp_stage_store<64, 64>(lw2v, s2v);
        if (threadIdx.x < 192) lb[threadIdx.x] = bias;     // [b1k | b2k | b1v | b2v]
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
    const long long tiles = ((long long)E + 31) / 32, stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 nxt[8];
    auto fetch = [&](long long tl) __attribute__((always_inline)) {      // this lane's half row of its edge of tile tl
        long long ee = tl * 32 + i;
        const float* row = attr + (ee < E ? ee : (long long)E - 1) * 64 + 32 * half;
#pragma unroll
        for (int m = 0; m < 8; ++m) nxt[m] = *reinterpret_cast<const float4*>(row + 4 * m);
    };
    if (tile < tiles) fetch(tile);
    for (; tile < tiles; tile += stride) {
        const long long e = tile * 32 + i;
        const bool ok = e < E;
        float asel[32];
#pragma unroll
        for (int m = 0; m < 8; ++m) asel[4 * m] = nxt[m].x, asel[4 * m + 1] = nxt[m].y, asel[4 * m + 2] = nxt[m].z, asel[4 * m + 3] = nxt[m].w;
        if (tile + stride < tiles) fetch(tile + stride);   // the next tile's rows travel while this tile's 176 MFMAs run
        {
            floatx16 out[1];
            edge_mlp_tile<32>(asel, lw1k, lb, lw2k, lb + 32, lane, out);
            if (ok) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    *reinterpret_cast<float4*>(wk + e * 32 + 8 * blk + 4 * half) =
                        make_float4(out[0][4 * blk], out[0][4 * blk + 1], out[0][4 * blk + 2], out[0][4 * blk + 3]);
            }
        }
        {
            floatx16 out[2];
            edge_mlp_tile<64>(asel, lw1v, lb + 64, lw2v, lb + 128, lane, out);
            if (ok) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk)
                        *reinterpret_cast<float4*>(wv + e * 64 + 32 * u + 8 * blk + 4 * half) =
                            make_float4(out[u][4 * blk], out[u][4 * blk + 1], out[u][4 * blk + 2], out[u][4 * blk + 3]);
            }
        }
    }
}


// Backward of one net (H hidden = H output units).  Per 32-edge tile, all on the MFMA:
//   pre^T = W1 . attr^T (recomputed, as forward), gh^T = W2^T . g_out^T  -> both [hidden x edge] accumulators (edge on the
//   lane), so h = ssp(pre), g_pre = gh * sigmoid(pre) are register-wise.  The weight-gradient products run over the EDGE as
//   k index (k-step s takes edge s in the lower lanes, 16 + s in the upper):
//     dW2[o][j] += g_out[e][o] h[e][j]       A = g_out rows, column on the lane;  B = h^T, four k-steps per 16-byte read
//     dW1[j][c] += g_pre[e][j] attr[e][c]    A = g_pre^T likewise;                B = attr rows, column on the lane
//   db2 / db1 are column sums of g_out / g_pre.
// Every operand of those products comes from a wave-private LDS region that is filled from REGISTERS: each lane holds its
// own edge's attr / g_out half rows anyway (the B operands of the two recompute chains), so it writes them out as the row
// images [edge][64 + 4] / [edge][H + 4] (16-byte writes, conflict-free with that pitch) and the products read them back with
// the column on the lane (4-byte reads of consecutive addresses, base + immediate).  The first version took these operands
// from global memory - 64 more 4-byte loads per tile and wavefront of rows that were cache hits, but ablation (tools/lab/
// emlp_ablate.py) put 1/3 of the kernel's time on them and on the exposed row loads.  The region holds {h^T, g_out rows}
// for the dW2 products and {g_pre^T, attr rows} for the dW1 products, one after the other (LDS operations of one wavefront
// complete in order): 13.3 KB per wavefront, so that two workgroups still share a CU.  The attr rows go out first (the
// pre chain is their last use in registers), which frees those registers for the rows of tile t + 1: loaded right after
// the pre chain of tile t, with the remaining 96 MFMAs of the tile as cover.  A ragged
// last tile is moved back to end at row E - 1 and masks the rows the tile before it has taken, rows before row 0 (fewer
// than 32 edges in all) are clamped and masked - all loads are in range without a per-row branch.
// A workgroup handles ONE tile of 32 hidden units (blockIdx.y): the 64-unit value net runs as two such slices, which
// halves the accumulator registers (two wavefronts per SIMD instead of one) at no extra matrix work.
// The gradients accumulate in registers over all tiles of a wavefront; the four wavefronts of a workgroup are summed
// through LDS and every workgroup writes one partial row part[block][slice][32*64 | 32 | H*32 | H] =
// [dW1 rows of the slice | db1 of the slice | dW2 columns of the slice, [out][32] | db2], reduced afterwards with
// singa_colsum.  No gradient w.r.t. attr (it carries none, CP:295-298).
template <int H>
__global__ void __launch_bounds__(256, 2) edge_mlp_mfma_bwd_kernel(const float* __restrict__ attr, const float* __restrict__ g_out,
                                                                   const float* __restrict__ w1t, const float* __restrict__ b1,
                                                                   const float* __restrict__ w2, float* __restrict__ part, int E) {
    constexpr int TO = H / 32, LD = 36, HH = H / 2, P2 = H + 4;
    constexpr int PSZ = 32 * 64 + 32 + H * 32 + H;
    constexpr int REG = 32 * LD + 32 * EMLP_P1;         // a wavefront's region: [32][LD] transposed tile + [32][68] row image
    __shared__ __attribute__((aligned(16))) float lw1[32 * EMLP_P1], lw2[32 * P2], tiles[4 * REG];
    __shared__ float lb1[32];
    static_assert(4 * REG >= PSZ, "the tile images double as the reduction buffer");
    const int ht = blockIdx.y;                          // hidden units [32 ht, 32 ht + 32)
    {
        float s1[8], s2[H / 8];
        emlp_stage_load<32, 64>(s1, w1t + (long long)32 * ht * 64);                      // rows 32 ht .. of W1 [H][64]
#pragma unroll
        for (int k = 0; k < H / 8; ++k) {                                                // columns 32 ht .. of W2 [H][H]
            const int t = threadIdx.x + 256 * k;
            s2[k] = w2[(t >> 5) * H + 32 * ht + (t & 31)];
        }
        const float bias = b1[32 * ht + (threadIdx.x & 31)];
        emlp_stage_store<32, 64>(lw1, s1);
#pragma unroll
        for (int k = 0; k < H / 8; ++k) {                                                // ... as rows of W2^T
            const int t = threadIdx.x + 256 * k;
            lw2[(t & 31) * P2 + (t >> 5)] = s2[k];
        }
        if (threadIdx.x < 32) lb1[threadIdx.x] = bias;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
    float* tt = tiles + wave * REG;                     // h^T, then g_pre^T: [32 hidden][LD], edge along the row
    float* rimg = tt + 32 * LD;                         // g_out rows [32 edges][P2], then attr rows [32 edges][68]
    floatx16 dw2[TO], dw1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int a = 0; a < TO; ++a) dw2[a][r] = 0.f;
        dw1[0][r] = 0.f, dw1[1][r] = 0.f;
    }
    float db1p = 0.f, db2p[TO];
#pragma unroll
    for (int a = 0; a < TO; ++a) db2p[a] = 0.f;
    const long long tilesN = ((long long)E + 31) / 32, stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    float4 asel4[8], grow4[HH / 4], nasel4[8], ngrow4[HH / 4];
    auto first_row = [&](long long tl) __attribute__((always_inline)) { return tl * 32 + 32 <= E ? tl * 32 : (long long)E - 32; };
    auto fetch = [&](long long tl) __attribute__((always_inline)) {      // this lane's half rows of its edge of tile tl
        const long long ee = first_row(tl) + i, er = ee > 0 ? ee : 0;
#pragma unroll
        for (int m = 0; m < 8; ++m) nasel4[m] = *reinterpret_cast<const float4*>(attr + er * 64 + 32 * half + 4 * m);
#pragma unroll
        for (int m = 0; m < HH / 4; ++m) ngrow4[m] = *reinterpret_cast<const float4*>(g_out + er * H + HH * half + 4 * m);
    };
    if (tile < tilesN) fetch(tile);
    for (; tile < tilesN; tile += stride) {
        const long long e0 = tile * 32;
        const bool ok = first_row(tile) + i >= e0;          // false: a row the previous tile has taken (or a row before row 0)
#pragma unroll
        for (int m = 0; m < 8; ++m) asel4[m] = nasel4[m];
#pragma unroll
        for (int m = 0; m < HH / 4; ++m) grow4[m] = ok ? ngrow4[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        // region = {g_pre^T (after the activation), attr rows (now)}
#pragma unroll
        for (int m = 0; m < 8; ++m) *reinterpret_cast<float4*>(rimg + i * EMLP_P1 + 32 * half + 4 * m) = asel4[m];
        floatx16 pacc, gacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            pacc[r] = lb1[8 * (r >> 2) + 4 * half + (r & 3)];
            gacc[r] = 0.f;
        }
        {
            const float* a1 = lw1 + i * EMLP_P1 + 32 * half;
            float4 fa[2];
            fa[0] = *reinterpret_cast<const float4*>(a1);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q + 1 < 8) {
                    fa[(q + 1) & 1] = *reinterpret_cast<const float4*>(a1 + 4 * (q + 1));
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                const float4 f = fa[q & 1], b = asel4[q];
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, b.x, pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, b.y, pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, b.z, pacc, 0, 0, 0);
                pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, b.w, pacc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tile + stride < tilesN) fetch(tile + stride);   // the attr registers are free now; 96 MFMAs of cover
        __builtin_amdgcn_sched_barrier(0);
        {
            const float* a2 = lw2 + i * P2 + HH * half;     // k-step s: output unit s (lower lanes) / HH + s (upper lanes)
            float4 fa[2];
            fa[0] = *reinterpret_cast<const float4*>(a2);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
            for (int q = 0; q < HH / 4; ++q) {
                if (q + 1 < HH / 4) {
                    fa[(q + 1) & 1] = *reinterpret_cast<const float4*>(a2 + 4 * (q + 1));
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                const float4 f = fa[q & 1], b = grow4[q];
                gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, b.x, gacc, 0, 0, 0);
                gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, b.y, gacc, 0, 0, 0);
                gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, b.z, gacc, 0, 0, 0);
                gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, b.w, gacc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        // ---- dW1: g_pre^T joins the attr rows; h stays in the registers of pre
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = 8 * (r >> 2) + 4 * half + (r & 3);
            // h = ssp(p) = max(p, 0) + log(1 + t) - ln 2 and sigmoid(p) = (p >= 0 ? 1 : t) / (1 + t) share t = exp(-|p|)
            const float p = pacc[r];
            const float tq = __expf(-fabsf(p)), u = 1.f + tq;
            const float sg = (p >= 0.f ? 1.f : tq) * SINGA_RCP(u);
            tt[j * LD + i] = ok ? gacc[r] * sg : 0.f;
            pacc[r] = ok ? fmaxf(p, 0.f) + __logf(u) - 0.69314718055994530942f : 0.f;
        }
        if (lane < 32) {
            float c = 0.f;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 v = *reinterpret_cast<const float4*>(tt + lane * LD + 4 * m);
                c += (v.x + v.y) + (v.z + v.w);
            }
            db1p += c;
        }
        {
            const float* prow = tt + i * LD + 16 * half;
            const float* arows = rimg + 16 * half * EMLP_P1 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 pa = *reinterpret_cast<const float4*>(prow + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = 4 * q + j;
                    const float pv = j == 0 ? pa.x : j == 1 ? pa.y : j == 2 ? pa.z : pa.w;
                    dw1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, arows[s * EMLP_P1], dw1[0], 0, 0, 0);
                    dw1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv, arows[s * EMLP_P1 + 32], dw1[1], 0, 0, 0);
                }
            }
        }
        // ---- dW2: region = {h^T, g_out rows (zero rows for masked edges)}
#pragma unroll
        for (int r = 0; r < 16; ++r) tt[(8 * (r >> 2) + 4 * half + (r & 3)) * LD + i] = pacc[r];
#pragma unroll
        for (int m = 0; m < HH / 4; ++m) *reinterpret_cast<float4*>(rimg + i * P2 + HH * half + 4 * m) = grow4[m];
        {
            const float* hrow = tt + i * LD + 16 * half;
            const float* grows = rimg + 16 * half * P2 + i;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hb = *reinterpret_cast<const float4*>(hrow + 4 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int s = 4 * q + j;
                    const float hv = j == 0 ? hb.x : j == 1 ? hb.y : j == 2 ? hb.z : hb.w;
#pragma unroll
                    for (int a = 0; a < TO; ++a) {
                        const float gv = grows[s * P2 + 32 * a];
                        db2p[a] += gv;
                        dw2[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, hv, dw2[a], 0, 0, 0);
                    }
                }
            }
        }
    }
    // workgroup reduction through LDS (the tile images are free now), then one partial row per workgroup and slice
    __syncthreads();
    float* red = tiles;
    constexpr int O_B1 = 32 * 64, O_W2 = O_B1 + 32, O_B2 = O_W2 + H * 32;
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 8 * (r >> 2) + 4 * half + (r & 3);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float* q = red + row * 64 + 32 * b + i;                  // dW1[hidden row of the slice][input channel]
                    *q = w == 0 ? dw1[b][r] : *q + dw1[b][r];
                }
#pragma unroll
                for (int a = 0; a < TO; ++a) {
                    float* q = red + O_W2 + (32 * a + row) * 32 + i;         // dW2[output unit][hidden unit of the slice]
                    *q = w == 0 ? dw2[a][r] : *q + dw2[a][r];
                }
            }
#pragma unroll
            for (int a = 0; a < TO; ++a) {
                const float d2 = db2p[a] + __shfl_xor(db2p[a], 32, 64);
                if (half == 0) {
                    float* q = red + O_B2 + 32 * a + i;
                    *q = w == 0 ? d2 : *q + d2;
                }
            }
            if (lane < 32) {
                float* q = red + O_B1 + lane;
                *q = w == 0 ? db1p : *q + db1p;
            }
        }
        __syncthreads();
    }
    // One partial row per workgroup COLUMN (blockIdx.x), the slices' pieces placed where the parameters' rows are:
    // [dW1 [H][64] | db1 [H] | dW2 [H][H] | db2 [H]] - a column sum over the rows is then the four gradients as they are
    // stored (the slice's 32 hidden units are rows 32 ht .. of W1 / b1 and COLUMNS 32 ht .. of W2; db2 is the same in every
    // slice: slice 0 writes it).
    constexpr int SL = H / 32, ROW = H * 64 + H + H * H + H;
    float* dst = part + (long long)blockIdx.x * ROW;
    for (int t = threadIdx.x; t < PSZ; t += 256) {
        int off;
        if (t < O_B1) off = ht * 32 * 64 + t;
        else if (t < O_W2) off = SL * 32 * 64 + ht * 32 + (t - O_B1);
        else if (t < O_B2) off = SL * 32 * 64 + SL * 32 + ((t - O_W2) >> 5) * H + ht * 32 + ((t - O_W2) & 31);
        else off = ht == 0 ? SL * 32 * 64 + SL * 32 + H * H + (t - O_B2) : -1;
        if (off >= 0) dst[off] = red[t];
    }
}


// ------------------------------------------------------------------------------------------------ masked, scaled softmax
// P = softmax(mask ? -1e9 : scale * s) over the last axis of s[BH, T, S] (ScaledDotProduct(De)Attention, CP:111-115,
// 140-146: divide by sqrt(d), masked_fill(-1e9), softmax) and dS = scale * P * (dP - sum_j dP_j P_j), zero where masked.
// One wavefront per (bh, t) row; mask[B, T, S] bytes with explicit strides (an expanded padding mask has stride 0 over t).
__global__ void __launch_bounds__(256) masked_softmax_fwd_kernel(const float* __restrict__ s, const unsigned char* __restrict__ mask,
                                                                 long long msb, long long mst, float* __restrict__ p, long long rows,
                                                                 int T, int S, int heads, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long bh = row / T, t = row - bh * T;
    const unsigned char* m = mask + (bh / heads) * msb + t * mst;
    const float* x = s + row * S;
    float mx = -INFINITY;
    for (int j = lane; j < S; j += 64) mx = fmaxf(mx, m[j] ? -1e9f : x[j] * scale);
    mx = wmax64(mx);
    float sum = 0.f;
    for (int j = lane; j < S; j += 64) sum += expf((m[j] ? -1e9f : x[j] * scale) - mx);
    sum = 1.f / wsum64(sum);
    for (int j = lane; j < S; j += 64) p[row * S + j] = expf((m[j] ? -1e9f : x[j] * scale) - mx) * sum;
}

__global__ void __launch_bounds__(256) masked_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ gp,
                                                                 const unsigned char* __restrict__ mask, long long msb, long long mst,
                                                                 float* __restrict__ gs, long long rows, int T, int S, int heads,
                                                                 float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long bh = row / T, t = row - bh * T;
    const unsigned char* m = mask + (bh / heads) * msb + t * mst;
    float dot = 0.f;
    for (int j = lane; j < S; j += 64) dot = fmaf(gp[row * S + j], p[row * S + j], dot);
    dot = wsum64(dot);
    for (int j = lane; j < S; j += 64) gs[row * S + j] = m[j] ? 0.f : scale * p[row * S + j] * (gp[row * S + j] - dot);
}


// ------------------------------------------------------------------------------------------------ dense attention on MFMA
// ctx = softmax(mask ? -1e9 : q k^T / sqrt(32)) v for q[BH,T,32], k[BH,S,32], v[BH,S,64] (ScaledDotProduct(De)Attention,
// CP:107-117, 136-148), flash-style on the f32 MFMA with the accumulator-as-operand chaining of k15c: a wavefront owns 32
// queries of one (batch, head); per tile of 32 keys it forms the TRANSPOSED scores s^T[key x query] = K . Q^T, so the
// query sits on the lane and 16 keys in its registers - the online softmax is register-wise (one cross-half exchange for
// the running maximum) and the probabilities are the B operand of ctx^T[value x query] += V^T . P^T, whose A operand
// v[key][value channel] is read straight from global memory in the accumulator's key order.  Nothing but ctx and, per
// query, the row maximum and the reciprocal row sum (for the backward) is written.
// Where the rows of one (batch, head) sit.  Head-major: q[B*heads, T, 32] (k: [.., S, 32], v / ctx: [.., 64]).  Token-major
// (tm): q[B, T, heads, 32] etc. - the layout in which the projections W_Q / W_K / W_V produce them and `linear` consumes
// the context (CP:96-117), so no head transposes (four copies forward, five backward per attention) are needed.
struct AttnLay {
    long long qb, kb, vb, ob;      // float offset of row 0 in q / k / v / ctx-like tensors
    int qs, ks, vs, os;            // floats between consecutive rows
};
struct AttnPitch {
    int q, k, v;                   // token-major only: floats between consecutive tokens of q / k / v (and of their gradients);
};                                 // heads * D when dense, larger when they are column blocks of one fused projection output
__device__ __forceinline__ AttnLay attn_lay(int bh, int heads, int T, int S, int tm, AttnPitch ld) {
    AttnLay a;
    if (tm) {
        const int b = bh / heads, h = bh - b * heads;
        a.qb = (long long)b * T * ld.q + h * 32; a.kb = (long long)b * S * ld.k + h * 32;
        a.vb = (long long)b * S * ld.v + h * 64; a.ob = ((long long)b * T * heads + h) * 64;
        a.qs = ld.q; a.ks = ld.k; a.vs = ld.v; a.os = heads * 64;
    } else {
        a.qb = (long long)bh * T * 32; a.kb = (long long)bh * S * 32; a.vb = (long long)bh * S * 64; a.ob = (long long)bh * T * 64;
        a.qs = 32; a.ks = 32; a.vs = 64; a.os = 64;
    }
    return a;
}

// The 16 mask bytes a lane needs for one 32-key tile sit in four runs of four consecutive keys (k0 + 8 g + 4 half + 0..3, the
// key order of the score accumulator): each run is ONE 4-byte load (any byte alignment - the target allows it), clamped so
// that it never reaches past the row (a ragged last tile reads the row's last four bytes; attn_mask_bit shifts when the word
// is USED - shifting at fetch time would make the prefetch wait for its own loads).  Runs that start at or past S are never
// looked at.  Rows shorter than 4 bytes: byte loads.
__device__ __forceinline__ void attn_mask_fetch(const unsigned char* __restrict__ mrow, int k0, int half, int S, unsigned (&w)[4]) {
    if (S >= 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kr = k0 + 8 * g + 4 * half, ks = kr < S - 4 ? kr : S - 4;
            __builtin_memcpy(&w[g], mrow + ks, 4);
        }
    } else {
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            unsigned v = 0;
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int kr = k0 + 8 * g + 4 * half + j;
                v |= (unsigned)mrow[kr < S ? kr : S - 1] << (8 * j);
            }
            w[g] = v;
        }
    }
}
// mask byte of accumulator register r (key k0 + 8 (r / 4) + 4 half + r % 4 < S) from the words attn_mask_fetch(k0) loaded
__device__ __forceinline__ bool attn_mask_bit(const unsigned (&w)[4], int k0, int half, int S, int r) {
    const int kr = k0 + 8 * (r >> 2) + 4 * half;
    const int sh = 8 * ((S >= 4 && kr > S - 4 ? kr - (S - 4) : 0) + (r & 3));
    return (sh < 32) & (((w[r >> 2] >> (sh & 31)) & 0xffu) != 0u);      // no short circuit: no branch per key
}

__global__ void __launch_bounds__(256) attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                       long long msb, long long mst, float* __restrict__ ctx,
                                                       float* __restrict__ lse, int BH, int T, int S, int heads, int tm, float scale, AttnPitch ld) {
    const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
    const int qtiles = (T + 31) / 32;
    // XCD b % 8 takes a contiguous eighth of the (batch x head, tile) list: the keys / values of one (batch, head) are
    // then read into ONE L2 instead of all eight (grid: a multiple of 8 workgroups)
    const long long w = (long long)xcd_range_id((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
    if (w >= (long long)BH * qtiles) return;
    const int bh = (int)(w / qtiles), qt = (int)(w - (long long)bh * qtiles);
    const int tq = qt * 32 + i, tqc = tq < T ? tq : T - 1;
    const AttnLay A = attn_lay(bh, heads, T, S, tm, ld);
    float qreg[16];
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
        const float4 t4 = *reinterpret_cast<const float4*>(q + A.qb + (long long)tqc * A.qs + 16 * half + 4 * m4);
        qreg[4 * m4] = t4.x, qreg[4 * m4 + 1] = t4.y, qreg[4 * m4 + 2] = t4.z, qreg[4 * m4 + 3] = t4.w;
    }
    const unsigned char* mrow = mask + (long long)(bh / heads) * msb + (long long)tqc * mst;
    const float* kb = k + A.kb;
    const float* vb = v + A.vb;
    floatx16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = 0.f, o1[r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    float4 knext[4];
    unsigned mnext[4];
    // the next tile's key rows AND this query's 16 mask bytes of that tile (key order of the score accumulator): read inside
    // the softmax loop they were 16 branch - load - vmcnt(0) round trips per tile, each also waiting for the prefetched rows
    auto fetch_k = [&](int k0) __attribute__((always_inline)) {
        const int ka = k0 + i < S ? k0 + i : S - 1;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) knext[m4] = *reinterpret_cast<const float4*>(kb + (long long)ka * A.ks + 16 * half + 4 * m4);
        attn_mask_fetch(mrow, k0, half, S, mnext);
    };
    fetch_k(0);
    for (int k0 = 0; k0 < S; k0 += 32) {
        float kreg[16], vreg[32];
        unsigned mb[4];
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4)
            kreg[4 * m4] = knext[m4].x, kreg[4 * m4 + 1] = knext[m4].y, kreg[4 * m4 + 2] = knext[m4].z, kreg[4 * m4 + 3] = knext[m4].w;
#pragma unroll
        for (int g = 0; g < 4; ++g) mb[g] = mnext[g];
        // this tile's value operands and the next tile's key rows travel while the score MFMAs and the softmax run
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int ks = k0 + 8 * (s >> 2) + 4 * half + (s & 3);
            const float* vr = vb + (long long)(ks < S ? ks : S - 1) * A.vs + i;
            vreg[2 * s] = vr[0], vreg[2 * s + 1] = vr[32];
        }
        fetch_k(k0 + 32 < S ? k0 + 32 : k0);      // unconditional (the last tile re-reads itself): behind a branch the
                                                   // compiler's vmcnt counts assume the shorter path and wait for the prefetch
        floatx16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], qreg[s], sc, 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            const bool bit = attn_mask_bit(mb, k0, half, S, r);
            const float vin = bit ? -1e9f : sc[r] * scale;
            const float val = kr < S ? vin : -INFINITY;
            sc[r] = val;
            mx = fmaxf(mx, val);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float alpha = __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __expf(sc[r] - mnew);
            psum += sc[r];
        }
        lrun = lrun * alpha + psum;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) o0[r] *= alpha, o1[r] *= alpha;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[2 * s], sc[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[2 * s + 1], sc[s], o1, 0, 0, 0);
        }
    }
    lrun += __shfl_xor(lrun, 32, 64);
    if (tq < T) {
        const float inv = 1.f / lrun;
        float* dst = ctx + A.ob + (long long)tq * A.os + 4 * half;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            *reinterpret_cast<float4*>(dst + 8 * blk) =
                make_float4(o0[4 * blk] * inv, o0[4 * blk + 1] * inv, o0[4 * blk + 2] * inv, o0[4 * blk + 3] * inv);
            *reinterpret_cast<float4*>(dst + 32 + 8 * blk) =
                make_float4(o1[4 * blk] * inv, o1[4 * blk + 1] * inv, o1[4 * blk + 2] * inv, o1[4 * blk + 3] * inv);
        }
        if (half == 0) {                                  // (row maximum, 1 / row sum): kept apart - for a fully masked row the
            lse[((long long)bh * T + tq) * 2] = mrun;     // maximum is -1e9 and log(sum) would vanish in its rounding
            lse[((long long)bh * T + tq) * 2 + 1] = inv;
        }
    }
}


// Backward of k19, two passes that both recompute the scores from q, k and the stored log-sum-exp:
//   pass A (a wavefront = 32 queries, lane = query): s^T, dP^T[key x query] = V . dO^T in the same layout, so
//     dS^T = P^T (dP^T - D) scale is register-wise (D = rowsum(dO * ctx), computed here and stored for pass B), and
//     dQ^T[dk x query] += K^T . dS^T takes dS^T as B operand;
//   pass B (a wavefront = 32 keys, lane = key): s[query x key] = Q . K^T and dP = dO . V^T with the key on the lane and 16
//     queries in registers, then dV^T[dv x key] += dO^T . P and dK^T[dk x key] += Q^T . dS with P / dS as B operands.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                          long long msb, long long mst, const float* __restrict__ ctx,
                                                          const float* __restrict__ lse, const float* __restrict__ go,
                                                          float* __restrict__ gq, float* __restrict__ dsum, int BH, int T, int S,
                                                          int heads, int tm, float scale, AttnPitch ld) {
    const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
    const int qtiles = (T + 31) / 32;
    // XCD b % 8 takes a contiguous eighth of the (batch x head, tile) list: the keys / values of one (batch, head) are
    // then read into ONE L2 instead of all eight (grid: a multiple of 8 workgroups)
    const long long w = (long long)xcd_range_id((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
    if (w >= (long long)BH * qtiles) return;
    const int bh = (int)(w / qtiles), qt = (int)(w - (long long)bh * qtiles);
    const int tq = qt * 32 + i, tqc = tq < T ? tq : T - 1;
    const long long qrow = (long long)bh * T + tqc;                  // lse / dsum stay [B*heads, T]
    const AttnLay A = attn_lay(bh, heads, T, S, tm, ld);
    float qreg[16], goreg[32];
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
        const float4 t4 = *reinterpret_cast<const float4*>(q + A.qb + (long long)tqc * A.qs + 16 * half + 4 * m4);
        qreg[4 * m4] = t4.x, qreg[4 * m4 + 1] = t4.y, qreg[4 * m4 + 2] = t4.z, qreg[4 * m4 + 3] = t4.w;
    }
    float dpart = 0.f;
#pragma unroll
    for (int m4 = 0; m4 < 8; ++m4) {
        const float4 g4 = *reinterpret_cast<const float4*>(go + A.ob + (long long)tqc * A.os + 32 * half + 4 * m4);
        const float4 c4 = *reinterpret_cast<const float4*>(ctx + A.ob + (long long)tqc * A.os + 32 * half + 4 * m4);
        goreg[4 * m4] = g4.x, goreg[4 * m4 + 1] = g4.y, goreg[4 * m4 + 2] = g4.z, goreg[4 * m4 + 3] = g4.w;
        dpart += (g4.x * c4.x + g4.y * c4.y) + (g4.z * c4.z + g4.w * c4.w);
    }
    const float dq_i = dpart + __shfl_xor(dpart, 32, 64);          // D of this lane's query
    const float m_i = lse[qrow * 2], linv_i = lse[qrow * 2 + 1];
    if (half == 0 && tq < T) dsum[qrow] = dq_i;
    const unsigned char* mrow = mask + (long long)(bh / heads) * msb + (long long)tqc * mst;
    const float* kb = k + A.kb;
    const float* vb = v + A.vb;
    floatx16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    float4 kn[4], vn[8];
    unsigned mnext[4];
    auto fetch = [&](int k0) __attribute__((always_inline)) {    // key / value rows and the 16 mask bytes of the next tile (see attn_fwd_kernel)
        const int ka = k0 + i < S ? k0 + i : S - 1;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) kn[m4] = *reinterpret_cast<const float4*>(kb + (long long)ka * A.ks + 16 * half + 4 * m4);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) vn[m4] = *reinterpret_cast<const float4*>(vb + (long long)ka * A.vs + 32 * half + 4 * m4);
        attn_mask_fetch(mrow, k0, half, S, mnext);
    };
    fetch(0);
    for (int k0 = 0; k0 < S; k0 += 32) {
        float kreg[16], vreg[32], kd[16];
        unsigned mb[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) mb[g] = mnext[g];
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) kreg[4 * m4] = kn[m4].x, kreg[4 * m4 + 1] = kn[m4].y, kreg[4 * m4 + 2] = kn[m4].z, kreg[4 * m4 + 3] = kn[m4].w;
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) vreg[4 * m4] = vn[m4].x, vreg[4 * m4 + 1] = vn[m4].y, vreg[4 * m4 + 2] = vn[m4].z, vreg[4 * m4 + 3] = vn[m4].w;
#pragma unroll
        for (int s = 0; s < 16; ++s) {                  // A operands of the dQ product, issued ahead of the 48 MFMAs before it
            const int ks = k0 + 8 * (s >> 2) + 4 * half + (s & 3);
            kd[s] = kb[(long long)(ks < S ? ks : S - 1) * A.ks + i];
        }
        fetch(k0 + 32 < S ? k0 + 32 : k0);        // unconditional, as in attn_fwd_kernel
        floatx16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s], qreg[s], sc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 32; ++s) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[s], goreg[s], dp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + 8 * (r >> 2) + 4 * half + (r & 3);
            const bool in = kr < S;
            const bool msk = in & attn_mask_bit(mb, k0, half, S, r);
            const float val = msk ? -1e9f : sc[r] * scale;
            const float pr = __expf((in ? val : -INFINITY) - m_i) * linv_i;     // exp(-inf) = 0 past the row end, no branch
            sc[r] = msk ? 0.f : pr * (dp[r] - dq_i) * scale;        // dS^T
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) dq = __builtin_amdgcn_mfma_f32_32x32x2f32(kd[s], sc[s], dq, 0, 0, 0);
    }
    if (tq < T) {
        float* dst = gq + A.qb + (long long)tq * A.qs + 4 * half;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
            *reinterpret_cast<float4*>(dst + 8 * blk) = make_float4(dq[4 * blk], dq[4 * blk + 1], dq[4 * blk + 2], dq[4 * blk + 3]);
    }
}

// (one wavefront per SIMD by design: asked for two, the compiler parks the prefetched rows in scratch and waits for them)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                           long long msb, long long mst, const float* __restrict__ lse,
                                                           const float* __restrict__ dsum, const float* __restrict__ go,
                                                           float* __restrict__ gk, float* __restrict__ gv, int BH, int T, int S,
                                                           int heads, int tm, float scale, AttnPitch ld) {
    const int lane = threadIdx.x & 63, i = lane & 31, half = lane >> 5;
    const int ktiles = (S + 31) / 32;
    // XCD b % 8 takes a contiguous eighth of the (batch x head, tile) list: the keys / values of one (batch, head) are
    // then read into ONE L2 instead of all eight (grid: a multiple of 8 workgroups)
    const long long w = (long long)xcd_range_id((int)blockIdx.x, (int)gridDim.x) * 4 + (threadIdx.x >> 6);
    if (w >= (long long)BH * ktiles) return;
    const int bh = (int)(w / ktiles), kt = (int)(w - (long long)bh * ktiles);
    const int key = kt * 32 + i, keyc = key < S ? key : S - 1;
    const bool kin = key < S;
    const AttnLay A = attn_lay(bh, heads, T, S, tm, ld);
    float kreg[16], vreg[32];
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
        const float4 t4 = *reinterpret_cast<const float4*>(k + A.kb + (long long)keyc * A.ks + 16 * half + 4 * m4);
        kreg[4 * m4] = t4.x, kreg[4 * m4 + 1] = t4.y, kreg[4 * m4 + 2] = t4.z, kreg[4 * m4 + 3] = t4.w;
    }
#pragma unroll
    for (int m4 = 0; m4 < 8; ++m4) {
        const float4 t4 = *reinterpret_cast<const float4*>(v + A.vb + (long long)keyc * A.vs + 32 * half + 4 * m4);
        vreg[4 * m4] = t4.x, vreg[4 * m4 + 1] = t4.y, vreg[4 * m4 + 2] = t4.z, vreg[4 * m4 + 3] = t4.w;
    }
    const unsigned char* mb = mask + (long long)(bh / heads) * msb + keyc;
    const float* qb = q + A.qb;
    const float* gob = go + A.ob;
    floatx16 dk, dv0, dv1;
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[r] = 0.f, dv0[r] = 0.f, dv1[r] = 0.f;
    float4 qn0, qn1, qn2, qn3, gn0, gn1, gn2, gn3, gn4, gn5, gn6, gn7;        // (scalars, not arrays: the compiler moved arrays to LDS / scratch)
    // The tile's 32 query rows (q: 32 floats, dO: 64 floats) travel global -> registers (one tile ahead, coalesced float4
    // rows) -> a wave-private LDS image [query][100], from which BOTH operand layouts are read right in front of their MFMAs:
    // [query on the lane] for s^T = K Q^T and dP^T = V dO^T, [query in the register index] for dV^T += dO^T P and
    // dK^T += Q^T dS.  (Round 2 fetched the second layout with 48 more global loads per tile and held both in registers:
    // 352 registers = one wavefront per SIMD, the compiler sank those loads to right in front of their MFMAs - vmcnt(47..0)
    // one by one - and a branch around the prefetch cost a vmcnt(0) per tile: MFMA-busy 0.27.)  The per-query softmax
    // statistics (row maximum, 1 / row sum, D = rowsum(dO . O)) ride along: one query per lane, read back per register row.
    constexpr int QS = 100;                                        // row pitch (floats): 16-byte aligned rows, q at 0, dO at 32
    __shared__ __attribute__((aligned(16))) float qimg[4][32 * QS];
    __shared__ float sstat[4][96];
    float* lw = qimg[threadIdx.x >> 6];
    float* ss = sstat[threadIdx.x >> 6];
    float sn0 = 0.f, sn1 = 0.f, sn2 = 0.f;
    unsigned char mraw[16];
#define SINGA_DKV_FETCH(T0)                                                                                                     \
    do {                                                                                                                        \
        const int ta_ = (T0) + i < T ? (T0) + i : T - 1; /* query row this lane supplies */                                     \
        const float* qr_ = qb + (long long)ta_ * A.qs + 16 * half;                                                               \
        const float* gr_ = gob + (long long)ta_ * A.os + 32 * half;                                                              \
        qn0 = *reinterpret_cast<const float4*>(qr_), qn1 = *reinterpret_cast<const float4*>(qr_ + 4);                            \
        qn2 = *reinterpret_cast<const float4*>(qr_ + 8), qn3 = *reinterpret_cast<const float4*>(qr_ + 12);                       \
        gn0 = *reinterpret_cast<const float4*>(gr_), gn1 = *reinterpret_cast<const float4*>(gr_ + 4);                            \
        gn2 = *reinterpret_cast<const float4*>(gr_ + 8), gn3 = *reinterpret_cast<const float4*>(gr_ + 12);                       \
        gn4 = *reinterpret_cast<const float4*>(gr_ + 16), gn5 = *reinterpret_cast<const float4*>(gr_ + 20);                      \
        gn6 = *reinterpret_cast<const float4*>(gr_ + 24), gn7 = *reinterpret_cast<const float4*>(gr_ + 28);                      \
        const float2 st_ = *reinterpret_cast<const float2*>(lse + ((long long)bh * T + ta_) * 2);                                \
        sn0 = st_.x, sn1 = st_.y, sn2 = dsum[(long long)bh * T + ta_];                                                           \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) { /* mask bytes of the tile's 16 register rows (a padding mask: 16 x */ \
            const int tr_ = (T0) + 8 * (r_ >> 2) + 4 * half + (r_ & 3);             /* the same byte), fetched with the rows */  \
            mraw[r_] = mb[(long long)(tr_ < T ? tr_ : T - 1) * mst];                                                             \
        }                                                                                                                        \
    } while (0)
    SINGA_DKV_FETCH(0);
    for (int t0 = 0; t0 < T; t0 += 32) {
        if (half == 0) ss[i] = sn0, ss[32 + i] = sn1, ss[64 + i] = sn2;
        {
            float* qw = lw + i * QS + 16 * half;
            float* gw = lw + i * QS + 32 + 32 * half;
            *reinterpret_cast<float4*>(qw) = qn0, *reinterpret_cast<float4*>(qw + 4) = qn1;
            *reinterpret_cast<float4*>(qw + 8) = qn2, *reinterpret_cast<float4*>(qw + 12) = qn3;
            *reinterpret_cast<float4*>(gw) = gn0, *reinterpret_cast<float4*>(gw + 4) = gn1;
            *reinterpret_cast<float4*>(gw + 8) = gn2, *reinterpret_cast<float4*>(gw + 12) = gn3;
            *reinterpret_cast<float4*>(gw + 16) = gn4, *reinterpret_cast<float4*>(gw + 20) = gn5;
            *reinterpret_cast<float4*>(gw + 24) = gn6, *reinterpret_cast<float4*>(gw + 28) = gn7;
        }
        unsigned mbits = 0;                             // this tile's mask bits (fetched with its rows, one tile ago)
#pragma unroll
        for (int r = 0; r < 16; ++r) mbits |= (mraw[r] != 0 ? 1u : 0u) << r;
        const int tnext = t0 + 32 < T ? t0 + 32 : t0;  // unconditional (the last tile re-reads its own rows): no branch, no vmcnt(0)
        SINGA_DKV_FETCH(tnext);
        floatx16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            const float4 a4 = *reinterpret_cast<const float4*>(lw + i * QS + 16 * half + 4 * m4);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, kreg[4 * m4], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, kreg[4 * m4 + 1], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, kreg[4 * m4 + 2], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, kreg[4 * m4 + 3], sc, 0, 0, 0);
        }
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) {
            const float4 a4 = *reinterpret_cast<const float4*>(lw + i * QS + 32 + 32 * half + 4 * m4);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, vreg[4 * m4], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, vreg[4 * m4 + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, vreg[4 * m4 + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, vreg[4 * m4 + 3], dp, 0, 0, 0);
        }
        floatx16 pr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tl = 8 * (r >> 2) + 4 * half + (r & 3);          // query of register r inside the tile
            const bool in = t0 + tl < T && kin;
            const bool msk = in && ((mbits >> r) & 1u);
            const float val = msk ? -1e9f : sc[r] * scale;
            const float p = in ? __expf(val - ss[tl]) * ss[32 + tl] : 0.f;
            pr[r] = p;
            sc[r] = msk ? 0.f : p * (dp[r] - ss[64 + tl]) * scale;          // dS
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* row = lw + (8 * (s >> 2) + 4 * half + (s & 3)) * QS;      // query 8 (s / 4) + 4 half + s % 4 of the tile
            dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32 + i], pr[s], dv0, 0, 0, 0);
            dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(row[64 + i], pr[s], dv1, 0, 0, 0);
            dk = __builtin_amdgcn_mfma_f32_32x32x2f32(row[i], sc[s], dk, 0, 0, 0);
        }
    }
#undef SINGA_DKV_FETCH
    if (kin) {
        float* dkd = gk + A.kb + (long long)key * A.ks + 4 * half;
        float* dvd = gv + A.vb + (long long)key * A.vs + 4 * half;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            *reinterpret_cast<float4*>(dkd + 8 * blk) = make_float4(dk[4 * blk], dk[4 * blk + 1], dk[4 * blk + 2], dk[4 * blk + 3]);
            *reinterpret_cast<float4*>(dvd + 8 * blk) = make_float4(dv0[4 * blk], dv0[4 * blk + 1], dv0[4 * blk + 2], dv0[4 * blk + 3]);
            *reinterpret_cast<float4*>(dvd + 32 + 8 * blk) =
                make_float4(dv1[4 * blk], dv1[4 * blk + 1], dv1[4 * blk + 2], dv1[4 * blk + 3]);
        }
    }
}


// ------------------------------------------------------------------------------------------------ column sums
// out[j] = sum_i x[i*ld + j]: bias / broadcast gradients.  A fixed-shape reduction tree: every pass lets one thread add up
// to COLSUM_R rows of one column (consecutive threads = consecutive columns, so loads coalesce), passes repeat until one
// row is left.  No atomics, no LDS, no global semaphores: identical results under eager launch and HIP-graph replay.
constexpr int COLSUM_R = 128;

// rows per thread in the first pass: as few as keep the number of partial rows <= COLSUM_R (so the second pass is the
// last one), but at least 16 - a [6400, 256] gradient then runs 100 x 256 threads instead of 50 x 256 and each walks
// 64 rows instead of 128 (the kernel is latency-bound at these sizes: 12 us for 6.5 MB before, in-graph)
static inline int colsum_first_r(long long M) {
    if (M > (long long)COLSUM_R * COLSUM_R) return COLSUM_R;
    int r = 16;
    while ((M + r - 1) / r > COLSUM_R) r <<= 1;
    return r;
}

__global__ void colsum_pass_kernel(const float* __restrict__ x, long long ld, long long M, int n, float* __restrict__ out,
                                   int R) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long slabs = (M + R - 1) / R;
    if (t >= slabs * n) return;
    const long long slab = t / n;
    const int j = (int)(t - slab * n);
    const long long r0 = slab * R;
    const long long r1 = r0 + R < M ? r0 + R : M;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    long long i = r0;
    for (; i + 7 < r1; i += 8) {
        a0 += x[i * ld + j];
        a1 += x[(i + 1) * ld + j];
        a2 += x[(i + 2) * ld + j];
        a3 += x[(i + 3) * ld + j];
        a4 += x[(i + 4) * ld + j];
        a5 += x[(i + 5) * ld + j];
        a6 += x[(i + 6) * ld + j];
        a7 += x[(i + 7) * ld + j];
    }
    for (; i < r1; ++i) a0 += x[i * ld + j];
    out[slab * n + j] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}


// Many column sums in two launches (the parameter-gradient reductions of one backward pass, singa_colsum_multi): the job
// table travels in the kernel arguments, so a captured launch carries it without a host-to-device copy.  Job k reduces
// x_k [M, n] over its rows; column j is ADDED to dst[j - col0] of the segment that holds it (a job feeds several
// parameter gradients when its columns are [d gamma | d beta | ...]).  Pass 1: one thread per (slab of R rows, column),
// as colsum_pass_kernel; jobs with a single slab finish there.  Pass 2: one thread per column adds the <= COLSUM_R slab
// partials in slab order.  Each gradient element is touched by exactly one thread per launch: fixed summation order.
constexpr int COLSUM_MJ = 36;     // jobs per launch
constexpr int COLSUM_MS = 72;     // destination segments per launch

struct ColsumJob {
    const float* x;
    float* work;          // [slabs, n] partials (unused when slabs == 1)
    long long ld;
    int M, n, R, slabs;
    int blk1, blk2;       // first block of this job in pass 1 / pass 2
    int seg0, nseg;
    int vec, pad_;        // vec: four columns per thread (float4 loads): n, ld, every segment start % 4 == 0, 16-byte aligned bases
};
struct ColsumSeg {
    float* dst;
    int col0;
    int pad_;
};
struct ColsumBatch {
    ColsumJob job[COLSUM_MJ];
    ColsumSeg seg[COLSUM_MS];
    int n_jobs;
};
static_assert(sizeof(ColsumBatch) <= 3968, "the job table must fit the kernel-argument segment");

__device__ inline void colsum_multi_emit(const ColsumBatch& b, const ColsumJob& jb, int j, float v) {
    int s = jb.seg0;
    for (int k = 1; k < jb.nseg; ++k)
        if (j >= b.seg[jb.seg0 + k].col0) s = jb.seg0 + k;
    float* d = b.seg[s].dst + (j - b.seg[s].col0);
    *d += v;
}

__device__ inline void colsum_multi_emit4(const ColsumBatch& b, const ColsumJob& jb, int j, float4 v) {
    int s = jb.seg0;
    for (int k = 1; k < jb.nseg; ++k)
        if (j >= b.seg[jb.seg0 + k].col0) s = jb.seg0 + k;
    float4* d = reinterpret_cast<float4*>(b.seg[s].dst + (j - b.seg[s].col0));
    float4 o = *d;
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *d = o;
}

__global__ void __launch_bounds__(256) colsum_multi_pass1_kernel(const ColsumBatch b) {
    int k = 0;
    for (int q = 1; q < b.n_jobs; ++q)
        if ((int)blockIdx.x >= b.job[q].blk1) k = q;
    const ColsumJob& jb = b.job[k];
    const long long t = (long long)((int)blockIdx.x - jb.blk1) * 256 + threadIdx.x;
    const float* __restrict__ x = jb.x;
    const long long ld = jb.ld;
    if (jb.vec) {                       // the partial slabs of the weight-gradient GEMMs: 16 bytes per lane
        const int n4 = jb.n >> 2;
        if (t >= (long long)jb.slabs * n4) return;
        const int slab = (int)(t / n4);
        const int j = 4 * (int)(t - (long long)slab * n4);
        const long long r0 = (long long)slab * jb.R;
        const long long r1 = r0 + jb.R < jb.M ? r0 + jb.R : jb.M;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        long long i = r0;
        for (; i + 3 < r1; i += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(x + i * ld + j), v1 = *reinterpret_cast<const float4*>(x + (i + 1) * ld + j);
            const float4 v2 = *reinterpret_cast<const float4*>(x + (i + 2) * ld + j), v3 = *reinterpret_cast<const float4*>(x + (i + 3) * ld + j);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; i < r1; ++i) {
            const float4 v0 = *reinterpret_cast<const float4*>(x + i * ld + j);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
        const float4 v = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                                     (a0.w + a1.w) + (a2.w + a3.w));
        if (jb.slabs == 1) colsum_multi_emit4(b, jb, j, v);
        else *reinterpret_cast<float4*>(jb.work + (long long)slab * jb.n + j) = v;
        return;
    }
    if (t >= (long long)jb.slabs * jb.n) return;
    const int slab = (int)(t / jb.n);
    const int j = (int)(t - (long long)slab * jb.n);
    const long long r0 = (long long)slab * jb.R;
    const long long r1 = r0 + jb.R < jb.M ? r0 + jb.R : jb.M;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    long long i = r0;
    for (; i + 7 < r1; i += 8) {
        a0 += x[i * ld + j];
        a1 += x[(i + 1) * ld + j];
        a2 += x[(i + 2) * ld + j];
        a3 += x[(i + 3) * ld + j];
        a4 += x[(i + 4) * ld + j];
        a5 += x[(i + 5) * ld + j];
        a6 += x[(i + 6) * ld + j];
        a7 += x[(i + 7) * ld + j];
    }
    for (; i < r1; ++i) a0 += x[i * ld + j];
    const float v = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (jb.slabs == 1)
        colsum_multi_emit(b, jb, j, v);
    else
        jb.work[(long long)slab * jb.n + j] = v;
}

__global__ void __launch_bounds__(256) colsum_multi_pass2_kernel(const ColsumBatch b) {
    int k = 0;
    for (int q = 1; q < b.n_jobs; ++q)
        if ((int)blockIdx.x >= b.job[q].blk2) k = q;
    const ColsumJob& jb = b.job[k];
    if (jb.slabs == 1) return;
    const float* __restrict__ w = jb.work;
    if (jb.vec) {
        const int j = 4 * (((int)blockIdx.x - jb.blk2) * 256 + threadIdx.x);
        if (j >= jb.n) return;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        int s = 0;
        for (; s + 1 < jb.slabs; s += 2) {
            const float4 v0 = *reinterpret_cast<const float4*>(w + (long long)s * jb.n + j);
            const float4 v1 = *reinterpret_cast<const float4*>(w + (long long)(s + 1) * jb.n + j);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
        }
        for (; s < jb.slabs; ++s) {
            const float4 v0 = *reinterpret_cast<const float4*>(w + (long long)s * jb.n + j);
            a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
        colsum_multi_emit4(b, jb, j, make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w));
        return;
    }
    const int j = ((int)blockIdx.x - jb.blk2) * 256 + threadIdx.x;
    if (j >= jb.n) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < jb.slabs; s += 4) {
        a0 += w[(long long)s * jb.n + j];
        a1 += w[(long long)(s + 1) * jb.n + j];
        a2 += w[(long long)(s + 2) * jb.n + j];
        a3 += w[(long long)(s + 3) * jb.n + j];
    }
    for (; s < jb.slabs; ++s) a0 += w[(long long)s * jb.n + j];
    colsum_multi_emit(b, jb, j, (a0 + a1) + (a2 + a3));
}

// rows per slab of one job: a power of two >= 16 that leaves at most COLSUM_R slabs
// (wide jobs of at most COLSUM_R rows - the partial slabs of the split weight-gradient GEMMs - take ONE slab: a thread adds
// up all rows of its columns, nothing is written to and read back from the workspace, and no second pass runs)
static inline int colsum_multi_r(long long M, int n) {
    if (M <= COLSUM_R && n >= 2048) return COLSUM_R;
    long long r = 16;
    while ((M + r - 1) / r > COLSUM_R) r <<= 1;
    return (int)r;
}

// ------------------------------------------------------------------------------------------------ fused Adam
// One launch updates every parameter tensor (torch.optim.Adam semantics, no weight decay / amsgrad): block b handles
// chunk b of the flattened (tensor, offset) table.  The step count and the learning rate live in device memory so the
// launch can be replayed from a HIP graph; a second one-thread kernel advances the step.
__global__ void __launch_bounds__(256) adam_kernel(float* const* __restrict__ p, const float* const* __restrict__ g,
                                                   float* const* __restrict__ m, float* const* __restrict__ v,
                                                   const long long* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                   const long long* __restrict__ chunk_off, int chunk,
                                                   const float* __restrict__ step, const float* __restrict__ lr, float b1,
                                                   float b2, float eps) {
    const int t = chunk_tensor[blockIdx.x];
    const long long off = chunk_off[blockIdx.x];
    const long long n = sizes[t];
    const float k = step[0] + 1.0f;
    const float bc1 = 1.0f - powf(b1, k), bc2s = sqrtf(1.0f - powf(b2, k));
    const float step_size = lr[0] / bc1;
    float* pp = p[t];
    const float* gg = g[t];
    float* mm = m[t];
    float* vv = v[t];
    const long long end = off + chunk < n ? off + chunk : n;
    for (long long i = off + threadIdx.x; i < end; i += blockDim.x) {
        const float gi = gg[i];
        const float mi = mm[i] + (gi - mm[i]) * (1.0f - b1);
        const float vi = b2 * vv[i] + (1.0f - b2) * gi * gi;
        mm[i] = mi;
        vv[i] = vi;
        pp[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
    }
}

__global__ void adam_advance_kernel(float* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1.0f;
}

// Total gradient norm over the same (tensor, chunk) table (clip_grad_norm_ of train.py:126): block b writes the sum of
// squares of its chunk, a one-block second pass adds the partials in a fixed order (double accumulators) and takes the
// root.  No cross-launch state, fixed summation order: the value is the same eager and replayed (torch's multi-block
// reductions were not, on this build - see singa_colsum).
__global__ void __launch_bounds__(256) grad_sumsq_kernel(const float* const* __restrict__ g, const long long* __restrict__ sizes,
                                                         const int* __restrict__ chunk_tensor,
                                                         const long long* __restrict__ chunk_off, int chunk,
                                                         float* __restrict__ partial) {
    __shared__ float red[256];
    const int t = chunk_tensor[blockIdx.x];
    const long long off = chunk_off[blockIdx.x];
    const long long n = sizes[t];
    const float* gg = g[t];
    const long long end = off + chunk < n ? off + chunk : n;
    float acc = 0.f;
    for (long long i = off + threadIdx.x; i < end; i += blockDim.x) acc = fmaf(gg[i], gg[i], acc);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) grad_norm_finish_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sqrt(red[0]);
}

// ------------------------------------------------------------------------------------------------ k8: S2 activation
// thread = (edge-or-node e, channel c).  x rows in registers; loop over the G grid points with the two grid-matrix
// rows as wave-uniform scalars: u = to[g,:].x, s = SiLU(u), y += from[g,:] * s.  Row 0 of the result is SiLU(gate).
// KIN rows, C channels (compile time).  EDGE: the rows come in the three per-m segments of an SO(2) convolution
// ((L+1) | 2L | 2(L-1) rows, KIN = 5L-1); otherwise one contiguous [KIN, C] record per row of the batch.
template <int KIN, int C, bool EDGE>
__global__ void __launch_bounds__(256) s2act_fwd_kernel(Segs x, const float* __restrict__ gate, long long ldg,
                                                        const float* __restrict__ to_grid,
                                                        const float* __restrict__ from_grid, float* __restrict__ out,
                                                        long long EC, int G) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= EC) return;
    long long e = tid / C;
    int c = (int)(tid - e * C);
    constexpr int LL = (KIN + 1) / 5;
    constexpr int r0 = EDGE ? LL + 1 : KIN, r01 = EDGE ? 3 * LL + 1 : KIN;
    const float* b0 = x.p[0] + e * x.ld[0] + c;
    const float* b1 = EDGE ? x.p[1] + e * x.ld[1] + c : b0;
    const float* b2 = EDGE ? x.p[2] + e * x.ld[2] + c : b0;
    float xv[KIN], yv[KIN];
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        xv[i] = i < r0 ? b0[i * C] : (i < r01 ? b1[(i - r0) * C] : b2[(i - r01) * C]);
        yv[i] = 0.f;
    }
    for (int g = 0; g < G; ++g) {
        const float* tg = to_grid + g * KIN;
        const float* fg = from_grid + g * KIN;
        float u = 0.f;
#pragma unroll
        for (int i = 0; i < KIN; ++i) u = fmaf(tg[i], xv[i], u);
        float s = silu(u);
#pragma unroll
        for (int i = 1; i < KIN; ++i) yv[i] = fmaf(fg[i], s, yv[i]);
    }
    float* o = out + e * KIN * C + c;
    o[0] = silu(gate[e * ldg + c]);
#pragma unroll
    for (int i = 1; i < KIN; ++i) o[(long long)i * C] = yv[i];
}

// Backward (recompute u): v = sum_{i>=1} from[g,i] gy[i];  w = v * SiLU'(u);  gx[i] += to[g,i] * w;
// g_gate = gy[0] * SiLU'(gate).
template <int KIN, int C, bool EDGE>
__global__ void __launch_bounds__(256) s2act_bwd_kernel(Segs x, const float* __restrict__ gate, long long ldg,
                                                        const float* __restrict__ to_grid,
                                                        const float* __restrict__ from_grid,
                                                        const float* __restrict__ g_out, float* __restrict__ gx,
                                                        float* __restrict__ g_gate, long long EC, int G) {
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= EC) return;
    long long e = tid / C;
    int c = (int)(tid - e * C);
    constexpr int LL = (KIN + 1) / 5;
    constexpr int r0 = EDGE ? LL + 1 : KIN, r01 = EDGE ? 3 * LL + 1 : KIN;
    const float* b0 = x.p[0] + e * x.ld[0] + c;
    const float* b1 = EDGE ? x.p[1] + e * x.ld[1] + c : b0;
    const float* b2 = EDGE ? x.p[2] + e * x.ld[2] + c : b0;
    const float* gi = g_out + e * KIN * C + c;
    float xv[KIN], gy[KIN], ga[KIN];
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        xv[i] = i < r0 ? b0[i * C] : (i < r01 ? b1[(i - r0) * C] : b2[(i - r01) * C]);
        gy[i] = gi[i * C];
        ga[i] = 0.f;
    }
    for (int g = 0; g < G; ++g) {
        const float* tg = to_grid + g * KIN;
        const float* fg = from_grid + g * KIN;
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int i = 0; i < KIN; ++i) u = fmaf(tg[i], xv[i], u);
#pragma unroll
        for (int i = 1; i < KIN; ++i) v = fmaf(fg[i], gy[i], v);
        float w = v * silu_grad(u);
#pragma unroll
        for (int i = 0; i < KIN; ++i) ga[i] = fmaf(tg[i], w, ga[i]);
    }
    float* o = gx + e * KIN * C + c;
#pragma unroll
    for (int i = 0; i < KIN; ++i) o[(long long)i * C] = ga[i];
    g_gate[e * C + c] = gy[0] * silu_grad(gate[e * ldg + c]);
}


// ------------------------------------------------------------------------------------------------ k8 (separable form)
// The grid matrices factor: to_grid[(b,a), i] = P[b,i] * A[a, mc(i)], from_grid[(b,a), i] = Q[b,i] * A[a, mc(i)] (Legendre
// transform in beta, Fourier transform in alpha).  Per beta ring: v[m] = sum_l P[b,(l,m)] x_(l,m); for each alpha:
// u = sum_m A[a,m] v[m], s = SiLU(u), w[m] += A[a,m] s; then y_(l,m) += Q[b,(l,m)] w[m].  ~3x fewer FMAs than the dense
// [G, KIN] products (L = 6 FFN grid: 6.8k instead of 20.6k per channel).  EDGE: m-primary rows of an SO(2) convolution
// (mmax = 2, three segments); otherwise the full l-primary [K, C] record of a node.
template <int L, bool EDGE>
struct S2Sep {
    static constexpr int KIN = EDGE ? 5 * L - 1 : (L + 1) * (L + 1);
    static constexpr int MM = EDGE ? 2 : L;
    static constexpr int NM = 2 * MM + 1;
    static constexpr int RB = 2 * (L + 1);
    static constexpr int RA = EDGE ? (L == 2 ? 7 : 5) : 2 * L + 3;
    static constexpr int HA = (RA - 1) / 2;          // RA is odd: alpha_a and alpha_{RA-a} are mirror images
    static constexpr int mc(int i) {
        if (EDGE) return i < L + 1 ? 2 : (i < 2 * L + 1 ? 3 : (i < 3 * L + 1 ? 1 : (i < 4 * L ? 4 : 0)));
        int l = 0;
        while ((l + 1) * (l + 1) <= i) ++l;
        return i - l * l - l + L;
    }
    // parity of l + |m| of row i: P~_l^|m|(-z) = (-1)^(l+|m|) P~_l^|m|(z), i.e. ring RB-1-b carries the table row of
    // ring b with this sign (beta_{RB-1-b} = pi - beta_b; the quadrature weights are symmetric too)
    static constexpr int par(int i) {
        if (EDGE) {
            int l = i < L + 1 ? i : (i < 2 * L + 1 ? 1 + i - (L + 1) : (i < 3 * L + 1 ? 1 + i - (2 * L + 1) : (i < 4 * L ? 2 + i - (3 * L + 1) : 2 + i - 4 * L)));
            int m = i < L + 1 ? 0 : (i < 3 * L + 1 ? 1 : 2);
            return (l + m) & 1;
        }
        int l = 0;
        while ((l + 1) * (l + 1) <= i) ++l;
        int m = i - l * l - l;
        return (l + (m < 0 ? -m : m)) & 1;
    }
};

// sqrt(2) cos / sin (2 pi k a / RA) as compile-time constants (the alpha-harmonics of the grid, EF:562-587 via e3nn's
// `sha`): a local constexpr table, indexed by unrolled loop counters, folds into instruction literals - no table loads,
// no scalar-register pressure (the run-time [RA, 2M+1] table of the first version cost 440 SGPR spill moves per ring).
constexpr double cx_sin_cos(int j, int n, bool want_cos) {      // sin or cos of 2 pi j / n, |j| reduced to (-n/2, n/2]
    j %= n;
    if (j < 0) j += n;
    if (2 * j > n) j -= n;
    const double x = 6.283185307179586476925286766559 * (double)j / (double)n;
    double term = want_cos ? 1.0 : x, sum = term;
    for (int t = 1; t < 30; ++t) {
        const int d = want_cos ? (2 * t - 1) * (2 * t) : (2 * t) * (2 * t + 1);
        term = -term * x * x / (double)d;
        sum += term;
    }
    return sum;
}
template <int RA, int MM>
struct FourTab {
    float c[MM + 1][(RA - 1) / 2 + 1];
    float s[MM + 1][(RA - 1) / 2 + 1];
};
template <int RA, int MM>
constexpr FourTab<RA, MM> make_four_tab() {
    FourTab<RA, MM> t{};
    for (int k = 0; k <= MM; ++k)
        for (int a = 0; a <= (RA - 1) / 2; ++a) {
            t.c[k][a] = (float)(1.4142135623730950488016887242097 * cx_sin_cos(k * a, RA, true));
            t.s[k][a] = (float)(1.4142135623730950488016887242097 * cx_sin_cos(k * a, RA, false));
        }
    return t;
}
__device__ __forceinline__ float silu_fast(float u) { return u * SINGA_RCP(1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_grad_fast(float u) {
    const float sg = SINGA_RCP(1.0f + __expf(-u));
    return sg * (1.0f + u * (1.0f - sg));
}

// u[a] = sum_m A[a, m] v[m] over the RA points of one ring (A[a, +k] = sqrt2 cos(k alpha_a), A[a, -k] = sqrt2 sin(k alpha_a),
// A[a, 0] = 1; v[MM + m]) through the even / odd split: u[a] = E[a] + O[a], u[RA - a] = E[a] - O[a].
template <int RA, int MM>
__device__ __forceinline__ void ring_to_grid(const float (&v)[2 * MM + 1], float (&u)[RA]) {
    constexpr auto T = make_four_tab<RA, MM>();
    constexpr int HA = (RA - 1) / 2;
    float e0 = v[MM];
#pragma unroll
    for (int k = 1; k <= MM; ++k) e0 = fmaf(T.c[k][0], v[MM + k], e0);
    u[0] = e0;
#pragma unroll
    for (int a = 1; a <= HA; ++a) {
        float ev = v[MM], od = 0.f;
#pragma unroll
        for (int k = 1; k <= MM; ++k) {
            ev = fmaf(T.c[k][a], v[MM + k], ev);
            od = fmaf(T.s[k][a], v[MM - k], od);
        }
        u[a] = ev + od;
        u[RA - a] = ev - od;
    }
}
// w[m] += sum_a A[a, m] s[a] (the transpose of ring_to_grid)
template <int RA, int MM>
__device__ __forceinline__ void ring_from_grid(const float (&sv)[RA], float (&w)[2 * MM + 1]) {
    constexpr auto T = make_four_tab<RA, MM>();
    constexpr int HA = (RA - 1) / 2;
    float sp[HA + 1], sm[HA + 1];
    float tot = sv[0];
#pragma unroll
    for (int a = 1; a <= HA; ++a) {
        sp[a] = sv[a] + sv[RA - a];
        sm[a] = sv[a] - sv[RA - a];
        tot += sp[a];
    }
    w[MM] = tot;
#pragma unroll
    for (int k = 1; k <= MM; ++k) {
        float wc = T.c[k][0] * sv[0], ws = 0.f;
#pragma unroll
        for (int a = 1; a <= HA; ++a) {
            wc = fmaf(T.c[k][a], sp[a], wc);
            ws = fmaf(T.s[k][a], sm[a], ws);
        }
        w[MM + k] = wc;
        w[MM - k] = ws;
    }
}

// thread = (row e, channel c), the KIN coefficient rows in registers.  Rings are handled in mirror pairs (b, RB-1-b): one
// Legendre pass with the table row of ring b gives both rings' Fourier coefficients (even / odd parts), and both rings'
// results are folded back with one pass.  Per ring the Fourier transform over alpha and its transpose use the even / odd
// split above.  SiLU fused; row 0 of the result := SiLU(gate).  P, Q: [RB, KIN] Legendre tables (to-grid / from-grid) in
// the row order of x, read as wave-uniform scalars.
template <int L, bool EDGE, int C>
__global__ void __launch_bounds__(256) s2act_sep_fwd_kernel(Segs x, const float* __restrict__ gate, long long ldg,
                                                            const float* __restrict__ P, const float* __restrict__ Q,
                                                            const float* __restrict__ A, float* __restrict__ out,
                                                            long long EC) {
    using S = S2Sep<L, EDGE>;
    constexpr int KIN = S::KIN, NM = S::NM, RA = S::RA, MM = S::MM;
    (void)A;
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= EC) return;
    long long e = tid / C;
    int c = (int)(tid - e * C);
    constexpr int r0 = EDGE ? L + 1 : KIN, r01 = EDGE ? 3 * L + 1 : KIN;
    const float* b0 = x.p[0] + e * x.ld[0] + c;
    const float* b1 = EDGE ? x.p[1] + e * x.ld[1] + c : b0;
    const float* b2 = EDGE ? x.p[2] + e * x.ld[2] + c : b0;
    float xv[KIN], yv[KIN];
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        xv[i] = i < r0 ? b0[i * C] : (i < r01 ? b1[(i - r0) * C] : b2[(i - r01) * C]);
        yv[i] = 0.f;
    }
    for (int b = 0; b < S::RB / 2; ++b) {
        const float* Pb = P + b * KIN;
        const float* Qb = Q + b * KIN;
        float ve[NM], vo[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = 0.f; vo[m] = 0.f; }
#pragma unroll
        for (int i = 0; i < KIN; ++i) {
            if (S::par(i)) vo[S::mc(i)] = fmaf(Pb[i], xv[i], vo[S::mc(i)]);
            else ve[S::mc(i)] = fmaf(Pb[i], xv[i], ve[S::mc(i)]);
        }
        float v1[NM], v2[NM], w1[NM], w2[NM], u[RA];
#pragma unroll
        for (int m = 0; m < NM; ++m) { v1[m] = ve[m] + vo[m]; v2[m] = ve[m] - vo[m]; }
        ring_to_grid<RA, MM>(v1, u);
#pragma unroll
        for (int a = 0; a < RA; ++a) u[a] = silu_fast(u[a]);
        ring_from_grid<RA, MM>(u, w1);
        ring_to_grid<RA, MM>(v2, u);
#pragma unroll
        for (int a = 0; a < RA; ++a) u[a] = silu_fast(u[a]);
        ring_from_grid<RA, MM>(u, w2);
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = w1[m] + w2[m]; vo[m] = w1[m] - w2[m]; }
#pragma unroll
        for (int i = 1; i < KIN; ++i) yv[i] = fmaf(Qb[i], S::par(i) ? vo[S::mc(i)] : ve[S::mc(i)], yv[i]);
    }
    float* o = out + e * KIN * C + c;
    o[0] = silu_fast(gate[e * ldg + c]);
#pragma unroll
    for (int i = 1; i < KIN; ++i) o[i * C] = yv[i];
}

// Two channels per thread on packed f32 arithmetic (v_pk_fma_f32: two FMAs per lane and instruction).  The one-channel
// kernel above issues its ~700 FMAs per ring pair one by one and sits at the scalar-FMA rate of the vector unit (edge grid:
// 0.55 TB/s of traffic, ~72 TFLOP/s); the Legendre / Fourier coefficients are wave-uniform scalars, so every FMA pairs the
// same coefficient with the two channels' values.  Same operations per channel in the same order: bit-identical results.
// (The backward kernel keeps one channel per thread.  Its two-channel twin was built and measured: 255 registers + 76 bytes of
// scratch at two wavefronts per SIMD for the L = 4 node grid, 2674 vs 2171 us at 49 k nodes; only the L = 2 grid gained, 9 %.)
#ifndef SINGA_V2F
typedef float v2f __attribute__((ext_vector_type(2)));
#else
typedef SINGA_V2F v2f;
#endif
__device__ __forceinline__ v2f silu_fast2(v2f u) {
    v2f r;
    r[0] = silu_fast(u[0]);
    r[1] = silu_fast(u[1]);
    return r;
}
template <int RA, int MM>
__device__ __forceinline__ void ring_to_grid2(const v2f (&v)[2 * MM + 1], v2f (&u)[RA]) {
    constexpr auto T = make_four_tab<RA, MM>();
    constexpr int HA = (RA - 1) / 2;
    v2f e0 = v[MM];
#pragma unroll
    for (int k = 1; k <= MM; ++k) e0 = T.c[k][0] * v[MM + k] + e0;
    u[0] = e0;
#pragma unroll
    for (int a = 1; a <= HA; ++a) {
        v2f ev = v[MM], od = {0.f, 0.f};
#pragma unroll
        for (int k = 1; k <= MM; ++k) {
            ev = T.c[k][a] * v[MM + k] + ev;
            od = T.s[k][a] * v[MM - k] + od;
        }
        u[a] = ev + od;
        u[RA - a] = ev - od;
    }
}
template <int RA, int MM>
__device__ __forceinline__ void ring_from_grid2(const v2f (&sv)[RA], v2f (&w)[2 * MM + 1]) {
    constexpr auto T = make_four_tab<RA, MM>();
    constexpr int HA = (RA - 1) / 2;
    v2f sp[HA + 1], sm[HA + 1];
    v2f tot = sv[0];
#pragma unroll
    for (int a = 1; a <= HA; ++a) {
        sp[a] = sv[a] + sv[RA - a];
        sm[a] = sv[a] - sv[RA - a];
        tot += sp[a];
    }
    w[MM] = tot;
#pragma unroll
    for (int k = 1; k <= MM; ++k) {
        v2f wc = T.c[k][0] * sv[0], ws = {0.f, 0.f};
#pragma unroll
        for (int a = 1; a <= HA; ++a) {
            wc = T.c[k][a] * sp[a] + wc;
            ws = T.s[k][a] * sm[a] + ws;
        }
        w[MM + k] = wc;
        w[MM - k] = ws;
    }
}
template <int L, bool EDGE, int C>
__global__ void __launch_bounds__(256) s2act_sep_fwd2_kernel(Segs x, const float* __restrict__ gate, long long ldg,
                                                             const float* __restrict__ P, const float* __restrict__ Q,
                                                             float* __restrict__ out, long long EC2) {
    using S = S2Sep<L, EDGE>;
    constexpr int KIN = S::KIN, NM = S::NM, RA = S::RA, MM = S::MM, C2 = C / 2;
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= EC2) return;
    long long e = tid / C2;
    int c = 2 * (int)(tid - e * C2);
    constexpr int r0 = EDGE ? L + 1 : KIN, r01 = EDGE ? 3 * L + 1 : KIN;
    const float* b0 = x.p[0] + e * x.ld[0] + c;
    const float* b1 = EDGE ? x.p[1] + e * x.ld[1] + c : b0;
    const float* b2 = EDGE ? x.p[2] + e * x.ld[2] + c : b0;
    v2f xv[KIN], yv[KIN];
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        xv[i] = *reinterpret_cast<const v2f*>(i < r0 ? b0 + i * C : (i < r01 ? b1 + (i - r0) * C : b2 + (i - r01) * C));
        yv[i] = v2f{0.f, 0.f};
    }
    for (int b = 0; b < S::RB / 2; ++b) {
        const float* Pb = P + b * KIN;
        const float* Qb = Q + b * KIN;
        v2f ve[NM], vo[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = v2f{0.f, 0.f}; vo[m] = v2f{0.f, 0.f}; }
#pragma unroll
        for (int i = 0; i < KIN; ++i) {
            if (S::par(i)) vo[S::mc(i)] = Pb[i] * xv[i] + vo[S::mc(i)];
            else ve[S::mc(i)] = Pb[i] * xv[i] + ve[S::mc(i)];
        }
        v2f v1[NM], v2[NM], w1[NM], w2[NM], u[RA];
#pragma unroll
        for (int m = 0; m < NM; ++m) { v1[m] = ve[m] + vo[m]; v2[m] = ve[m] - vo[m]; }
        ring_to_grid2<RA, MM>(v1, u);
#pragma unroll
        for (int a = 0; a < RA; ++a) u[a] = silu_fast2(u[a]);
        ring_from_grid2<RA, MM>(u, w1);
        ring_to_grid2<RA, MM>(v2, u);
#pragma unroll
        for (int a = 0; a < RA; ++a) u[a] = silu_fast2(u[a]);
        ring_from_grid2<RA, MM>(u, w2);
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = w1[m] + w2[m]; vo[m] = w1[m] - w2[m]; }
#pragma unroll
        for (int i = 1; i < KIN; ++i) yv[i] = Qb[i] * (S::par(i) ? vo[S::mc(i)] : ve[S::mc(i)]) + yv[i];
    }
    float* o = out + e * KIN * C + c;
    const v2f g2 = *reinterpret_cast<const v2f*>(gate + e * ldg + c);
    *reinterpret_cast<v2f*>(o) = silu_fast2(g2);
#pragma unroll
    for (int i = 1; i < KIN; ++i) *reinterpret_cast<v2f*>(o + i * C) = yv[i];
}

// Backward (recompute), same pairing: per ring v = P x and gq = Q^T gy (rows i >= 1); per alpha u = A v,
// t = SiLU'(u) * (A gq); acc = A^T t; gx_i += P[b, i] acc[m(i)].  g_gate = gy_0 * SiLU'(gate).
// FFN (node grid only): the gradient arriving at the activation's output is not read but FORMED here - it is the input gradient
// of the SO3_LinearV2 that follows the activation in the feed-forward block (EF:262, 655-671: 512 -> 16 channels), gy[i][c] =
// sum_u gs[n, i, u] * W2[l(i)][u][c], a 16-long contraction per coefficient row.  `g_out` is then the SMALL gradient gs [N, K, 16]
// and W2 the linear's weight [L+1][16][C] (c contiguous).  A workgroup holds 256 channels of ONE node: the node's K x 16 rows of
// gs are staged in LDS once and read back as broadcasts (every lane the same address: conflict-free), each degree's 16 weights of
// the thread's channel sit in registers while its 2l+1 rows are formed.  This removes the [N, K, 512] gradient tensor's write
// (the k11s expand launch) and its re-read here: 2 x 2.5 GB per layer pass at config 3.
template <int L, bool EDGE, int C, bool FFN = false>
__global__ void __launch_bounds__(256, (FFN && L == 4) ? 3 : 1) s2act_sep_bwd_kernel(Segs x, const float* __restrict__ gate, long long ldg,
                                                            const float* __restrict__ P, const float* __restrict__ Q,
                                                            const float* __restrict__ A, const float* __restrict__ g_out,
                                                            SegsMut gx, float* __restrict__ g_gate, long long ldgg,
                                                            long long EC, const float* __restrict__ W2) {
    using S = S2Sep<L, EDGE>;
    constexpr int KIN = S::KIN, NM = S::NM, RA = S::RA, MM = S::MM;
    // (round 4, measured and left off: the two rings of a mirror pair packed into v_pk_fma_f32 halves - 2.45 instead of 2.41 ms on
    // the feed-forward grid, 565 instead of 520 us on the attention grid: the ~110 exp + rcp pairs per thread, quarter rate,
    // and the moves that build the pairs cost what the halved FMA count saves)
#ifdef SINGA_S2_BWD_PACK_RINGS
    constexpr bool PACK_RINGS = true;
#else
    constexpr bool PACK_RINGS = false;
#endif
    (void)A;
    static_assert(!FFN || (!EDGE && C % 256 == 0), "the fused form is the node grid's, one node per workgroup");
    long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!FFN && tid >= EC) return;               // (FFN: EC is a multiple of the workgroup size - no partial workgroup, checked by the host)
    long long e = tid / C;
    int c = (int)(tid - e * C);
    constexpr int r0 = EDGE ? L + 1 : KIN, r01 = EDGE ? 3 * L + 1 : KIN;
    const float* b0 = x.p[0] + e * x.ld[0] + c;
    const float* b1 = EDGE ? x.p[1] + e * x.ld[1] + c : b0;
    const float* b2 = EDGE ? x.p[2] + e * x.ld[2] + c : b0;
    float xv[KIN], gy[KIN], ga[KIN];
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        xv[i] = i < r0 ? b0[i * C] : (i < r01 ? b1[(i - r0) * C] : b2[(i - r01) * C]);
        ga[i] = 0.f;
    }
    if constexpr (FFN) {
        constexpr int U = 16;
        __shared__ __attribute__((aligned(16))) float gs[KIN * U];
        const float4* gp = reinterpret_cast<const float4*>(g_out + e * (KIN * U));
        for (int t = threadIdx.x; t < KIN * U / 4; t += 256) reinterpret_cast<float4*>(gs)[t] = gp[t];
        __syncthreads();
        const float* wc = W2 + c;
        float wn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) wn[u] = wc[u * C];
#pragma unroll
        for (int l = 0; l <= L; ++l) {
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = wn[u];
            if (l < L) {                       // the next degree's weights travel behind this degree's FMAs
#pragma unroll
                for (int u = 0; u < U; ++u) wn[u] = wc[((l + 1) * U + u) * C];
            }
#pragma unroll
            for (int i = l * l; i < (l + 1) * (l + 1); ++i) {
                float acc = 0.f;
#pragma unroll
                for (int q = 0; q < U / 4; ++q) {
                    const float4 g4 = *reinterpret_cast<const float4*>(gs + i * U + 4 * q);
                    acc = fmaf(g4.x, w[4 * q], acc);
                    acc = fmaf(g4.y, w[4 * q + 1], acc);
                    acc = fmaf(g4.z, w[4 * q + 2], acc);
                    acc = fmaf(g4.w, w[4 * q + 3], acc);
                }
                // L = 6: one row at a time.  Left alone, the SLP vectoriser pairs rows into v_pk_fma_f32 and keeps more weights
                // and LDS words live at once: fine inside the 168-register budget of L = 4 (three wavefronts per SIMD, see
                // __launch_bounds__), scratch spills at L = 6
                if constexpr (L > 4) SINGA_KEEP_VGPR(acc);
                gy[i] = acc;
            }
        }
    } else {
        const float* gi = g_out + e * KIN * C + c;
#pragma unroll
        for (int i = 0; i < KIN; ++i) gy[i] = gi[i * C];
    }
    for (int b = 0; b < S::RB / 2; ++b) {
        const float* Pb = P + b * KIN;
        const float* Qb = Q + b * KIN;
        float ve[NM], vo[NM], ge[NM], go[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = 0.f; vo[m] = 0.f; ge[m] = 0.f; go[m] = 0.f; }
#pragma unroll
        for (int i = 0; i < KIN; ++i) {
            if (S::par(i)) vo[S::mc(i)] = fmaf(Pb[i], xv[i], vo[S::mc(i)]);
            else ve[S::mc(i)] = fmaf(Pb[i], xv[i], ve[S::mc(i)]);
        }
#pragma unroll
        for (int i = 1; i < KIN; ++i) {
            if (S::par(i)) go[S::mc(i)] = fmaf(Qb[i], gy[i], go[S::mc(i)]);
            else ge[S::mc(i)] = fmaf(Qb[i], gy[i], ge[S::mc(i)]);
        }
        float acc1[NM], acc2[NM];
        if constexpr (PACK_RINGS) {
            // the two rings of the mirror pair go through the SAME Fourier steps with the same (compile-time) coefficients: packed
            // into the two halves of v_pk_fma_f32 operands the three transforms of a pair cost the instructions of one ring
            // (the same operations per ring in the same order: bit-identical to the scalar form below)
            v2f v[NM], q[NM], u[RA], t[RA], acc[NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                v[m] = v2f{ve[m] + vo[m], ve[m] - vo[m]};
                q[m] = v2f{ge[m] + go[m], ge[m] - go[m]};
            }
            ring_to_grid2<RA, MM>(v, u);
            ring_to_grid2<RA, MM>(q, t);
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                t[a][0] *= silu_grad_fast(u[a][0]);
                t[a][1] *= silu_grad_fast(u[a][1]);
            }
            ring_from_grid2<RA, MM>(t, acc);
#pragma unroll
            for (int m = 0; m < NM; ++m) { acc1[m] = acc[m][0]; acc2[m] = acc[m][1]; }
        } else {
            float v[NM], q[NM], u[RA], t[RA];
#pragma unroll
            for (int m = 0; m < NM; ++m) { v[m] = ve[m] + vo[m]; q[m] = ge[m] + go[m]; }
            ring_to_grid<RA, MM>(v, u);
            ring_to_grid<RA, MM>(q, t);
#pragma unroll
            for (int a = 0; a < RA; ++a) t[a] *= silu_grad_fast(u[a]);
            ring_from_grid<RA, MM>(t, acc1);
#pragma unroll
            for (int m = 0; m < NM; ++m) { v[m] = ve[m] - vo[m]; q[m] = ge[m] - go[m]; }
            ring_to_grid<RA, MM>(v, u);
            ring_to_grid<RA, MM>(q, t);
#pragma unroll
            for (int a = 0; a < RA; ++a) t[a] *= silu_grad_fast(u[a]);
            ring_from_grid<RA, MM>(t, acc2);
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) { ve[m] = acc1[m] + acc2[m]; vo[m] = acc1[m] - acc2[m]; }
#pragma unroll
        for (int i = 0; i < KIN; ++i) ga[i] = fmaf(Pb[i], S::par(i) ? vo[S::mc(i)] : ve[S::mc(i)], ga[i]);
    }
    // the gradient goes out in the segments the caller names (e.g. straight into the column blocks of the gradient of the
    // SO(2) convolution's three outputs - no concatenation afterwards)
    float* o0 = gx.p[0] + e * gx.ld[0] + c;
    float* o1 = EDGE ? gx.p[1] + e * gx.ld[1] + c : o0;
    float* o2 = EDGE ? gx.p[2] + e * gx.ld[2] + c : o0;
#pragma unroll
    for (int i = 0; i < KIN; ++i) {
        if (i < r0) o0[i * C] = ga[i];
        else if (i < r01) o1[(i - r0) * C] = ga[i];
        else o2[(i - r01) * C] = ga[i];
    }
    g_gate[e * ldgg + c] = gy[0] * silu_grad_fast(gate[e * ldg + c]);
}

// ------------------------------------------------------------------------------------------------ k11s: skinny SO3 linears
// The feed-forward block's SO3_LinearV2 pair maps 16 <-> 512 channels per coefficient row (EF:232-262, 655-671).  With a
// 16-long contraction the MFMA tile kernel (k11) is only a vehicle for moving the [N, K, 512] tensor: 0.78 ms to WRITE it
// (3.3 TB/s) and 0.9-1.0 ms to reduce it into a [16, 512] weight gradient (its four 128-column tiles fetch every 2 KB row in
// four 512-byte pieces at different times).  Two VALU kernels take those two shapes: thread = one of the 512 channels, so
// every access to the big tensor is a whole 2 KB row per node row (consecutive lanes = consecutive channels); the 16-wide
// side of a node ([K, 16]) is loaded once per wavefront and read as lane broadcasts, fetched one node ahead.
//   expand:  big[n, k, c]      = sum_u small[n, k, u] * W[l(k)][c][u]  (+ bias[c] on k = 0)         (forward 16 -> 512; dX 512 -> 16)
//   reduce:  part[b][l][u][c]  = sum_{n in block b} sum_{k in l} small[n, k, u] * big[n, k, c]        (both weight gradients)
// W is addressed with strides (w_l, w_c, w_u) so that one kernel serves weight[l][c][u] and weight[l][u][c].
// C = 512: two workgroups of 256 threads per run of nodes (blockIdx & 1 = channel half); C = 112 (the attention's output
// projection, EF:1201-1204): one workgroup of 128 threads, 112 of them active.
template <int C>
struct SkinnyCfg {
    static constexpr int HALVES = C > 256 ? 2 : 1;
    static constexpr int BLOCK = (C / HALVES + 63) / 64 * 64;
};

template <int L, int C>
__global__ void __launch_bounds__(SkinnyCfg<C>::BLOCK) so3_skinny_expand_kernel(const float* __restrict__ small, const float* __restrict__ W,
                                                                long long w_l, long long w_c, long long w_u,
                                                                const float* __restrict__ bias, float* __restrict__ big, int N,
                                                                int npb) {
    constexpr int K = (L + 1) * (L + 1), HV = SkinnyCfg<C>::HALVES, BL = SkinnyCfg<C>::BLOCK;
    const int lane = threadIdx.x & 63;
    const int ct = (int)(blockIdx.x % HV) * BL + threadIdx.x;
    const bool act = ct < C;
    const int c = act ? ct : C - 1;                      // surplus lanes shadow the last channel (loads only, no stores)
    const int n0 = (int)(blockIdx.x / HV) * npb;
    const int n1 = n0 + npb < N ? n0 + npb : N;
    if (n0 >= n1) return;
    float w[L + 1][16];
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int u = 0; u < 16; ++u) w[l][u] = W[l * w_l + c * w_c + u * w_u];
    const float bc = bias ? bias[c] : 0.f;
    // the node's [K, 16] rows: one coalesced load per wavefront (lane i holds elements i, i + 64, ..), every use is a lane
    // broadcast into a scalar register (as the Wigner records of k4 / k10); the next node's rows are fetched a node ahead.
    // (Scalar loads of the rows: 1.25 ms - every use waits for ALL outstanding scalar loads.  Through LDS, 100 broadcast
    // ds_read_b128 per wavefront and node: 0.72 ms, LDS-bandwidth bound, and a barrier per node.  This form: 0.81 ms.)
    WRows<K * 16> X, Xn;
    Xn.load(small + (long long)n0 * K * 16, lane);
    for (int n = n0; n < n1; ++n) {
        X = Xn;
        Xn.load(small + (long long)(n + 1 < n1 ? n + 1 : n) * K * 16, lane);
        float* o = big + (long long)n * K * C + c;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int l = 0;
            while ((l + 1) * (l + 1) <= k) ++l;
            float a = k == 0 ? bc : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) a = fmaf(W_AT(X, k * 16 + u), w[l][u], a);
            if (act) o[k * C] = a;
        }
    }
}

// out_cu: partial rows are [l][c][u] (the layout of weight[l][c][u]); otherwise [l][u][c].  bias_row: one more row [C] per
// partial = sum over the block's nodes of big[n, 0, c] (the bias gradient of the 16 -> 512 map).  part: [blocks/2][PSZ].
template <int L, int C>
__global__ void __launch_bounds__(SkinnyCfg<C>::BLOCK) so3_skinny_reduce_kernel(const float* __restrict__ small, const float* __restrict__ big,
                                                                float* __restrict__ part, int N, int npb, int out_cu,
                                                                int bias_row) {
    constexpr int K = (L + 1) * (L + 1), HV = SkinnyCfg<C>::HALVES, BL = SkinnyCfg<C>::BLOCK;
    const int lane = threadIdx.x & 63;
    const int ct = (int)(blockIdx.x % HV) * BL + threadIdx.x;
    const bool act = ct < C;
    const int c = act ? ct : C - 1;
    const int n0 = (int)(blockIdx.x / HV) * npb;
    const int n1 = n0 + npb < N ? n0 + npb : N;
    float acc[L + 1][16], accb = 0.f;
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[l][u] = 0.f;
    if (n0 < n1) {
        WRows<K * 16> X, Xn;
        float bv[K], bn[K];
        Xn.load(small + (long long)n0 * K * 16, lane);
#pragma unroll
        for (int k = 0; k < K; ++k) bn[k] = big[((long long)n0 * K + k) * C + c];
        for (int n = n0; n < n1; ++n) {
            X = Xn;
#pragma unroll
            for (int k = 0; k < K; ++k) bv[k] = bn[k];
            const int nn = n + 1 < n1 ? n + 1 : n;                   // the next node's rows travel during this node's FMAs
            Xn.load(small + (long long)nn * K * 16, lane);
#pragma unroll
            for (int k = 0; k < K; ++k) bn[k] = big[((long long)nn * K + k) * C + c];
            accb += bv[0];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int l = 0;
                while ((l + 1) * (l + 1) <= k) ++l;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    acc[l][u] = fmaf(W_AT(X, k * 16 + u), bv[k], acc[l][u]);
                    // one VGPR per accumulator: without this the SLP vectoriser pairs them into v_pk_fma_f32, parks the pairs
                    // in AGPRs and shuffles (512 registers + scratch for 80 accumulators)
                    SINGA_KEEP_VGPR(acc[l][u]);
                }
            }
        }
    }
    constexpr int WSZ_ = (L + 1) * 16 * C;
    float* p = part + (long long)(blockIdx.x / HV) * (WSZ_ + (bias_row ? C : 0));
    if (!act) return;
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (out_cu) p[(l * C + c) * 16 + u] = acc[l][u];
            else p[(l * 16 + u) * C + c] = acc[l][u];
        }
    if (bias_row) p[WSZ_ + c] = accb;
}

// ---- k11s on the matrix cores.  The VALU kernels above spend one v_readlane per FMA (the 16-wide operand is a lane
// broadcast), i.e. ~800 vector instructions per thread and node for 100 bytes of the big tensor: 0.73 / 0.86 ms per launch at
// config 3 against 0.48 ms for moving the 2.5 GB.  v_mfma_f32_16x16x4_f32 does the same 16-long contraction with the big
// tensor's channels on the lanes and needs 32 cycles per KB of it per SIMD - a quarter of what HBM can feed - so both kernels
// become plain streaming kernels.  Channel slab of a wavefront: 64 channels c0 + 4 i + t (i = lane & 15 = MFMA column, t = 0..3
// = one of four accumulator tiles): the four tiles of a lane are four CONSECUTIVE channels, so every access to the big tensor
// is a float4 per lane and 256 contiguous bytes per row and 16-lane group.
//   expand: tile rows = 16 consecutive nodes at one coefficient k, contraction u = 4 g + j (g = lane >> 4, j = MFMA step):
//           A = small[node i][k][4 g .. 4 g + 3] (one float4 load), B = W[l(k)][channel][u] (registers, per degree).
//   reduce: tile rows = the 16 values of u, contraction = 4 consecutive nodes at one coefficient k:
//           A = small[node g][k][u = i], B = big[node g][k][4 channels] (one float4 load), accumulators per degree.
#ifndef SINGA_FLOATX4
typedef float floatx4 __attribute__((ext_vector_type(4)));
#else
typedef SINGA_FLOATX4 floatx4;
#endif

template <int L, int C>
__global__ void __launch_bounds__(SkinnyCfg<C>::BLOCK) so3_skinny_expand_mfma_kernel(const float* __restrict__ small, const float* __restrict__ W,
                                                                     long long w_l, long long w_c, long long w_u,
                                                                     const float* __restrict__ bias, float* __restrict__ big, int N,
                                                                     int npb) {
    constexpr int K = (L + 1) * (L + 1), HV = SkinnyCfg<C>::HALVES, BL = SkinnyCfg<C>::BLOCK;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int c0 = ((int)(blockIdx.x % HV) * (BL / 64) + (int)(threadIdx.x >> 6)) * 64 + 4 * i;   // this lane's four channels
    const bool act = c0 + 3 < C;                                   // (C is a multiple of 4: a lane's float4 is inside or outside)
    const int cc = act ? c0 : C - 4;
    const int n0 = (int)(blockIdx.x / HV) * npb;
    const int n1 = n0 + npb < N ? n0 + npb : N;
    if (n0 >= n1) return;
    float w[L + 1][4][4];                                          // [degree][tile t = channel cc + t][step j: u = 4 g + j]
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[l][t][j] = W[l * w_l + (cc + t) * w_c + (4 * g + j) * w_u];
    float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bc = *reinterpret_cast<const float4*>(bias + cc);
    for (int nb = n0; nb < n1; nb += 16) {
        const int na = nb + i < n1 ? nb + i : n1 - 1;              // the node whose rows this lane feeds as A
        const float* arow = small + (long long)na * K * 16 + 4 * g;
        float4 a = *reinterpret_cast<const float4*>(arow);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float4 an = a;
            if (k + 1 < K) an = *reinterpret_cast<const float4*>(arow + (k + 1) * 16);
            int l = 0;
            while ((l + 1) * (l + 1) <= k) ++l;
            const float av[4] = {a.x, a.y, a.z, a.w};
            floatx4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float b0 = k == 0 ? (t == 0 ? bc.x : (t == 1 ? bc.y : (t == 2 ? bc.z : bc.w))) : 0.f;
                acc[t] = floatx4{b0, b0, b0, b0};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], w[l][t][j], acc[t], 0, 0, 0);
            // accumulator register r = tile row 4 g + r = node nb + 4 g + r, column i = channels cc .. cc + 3 over the tiles
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + 4 * g + r;
                if (act && n < n1)
                    *reinterpret_cast<float4*>(big + ((long long)n * K + k) * C + cc) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            }
            a = an;
        }
    }
}

template <int L>
struct SkinnyChunk {
    static constexpr int K = (L + 1) * (L + 1);
    static constexpr int D = L == 2 ? 3 : (L == 4 ? 5 : 7);        // coefficient rows per prefetch chunk (K = D * D)
};

template <int L, int C>
__global__ void __launch_bounds__(SkinnyCfg<C>::BLOCK) so3_skinny_reduce_mfma_kernel(const float* __restrict__ small, const float* __restrict__ big,
                                                                     float* __restrict__ part, int N, int npb, int out_cu,
                                                                     int bias_row) {
    constexpr int K = (L + 1) * (L + 1), HV = SkinnyCfg<C>::HALVES, BL = SkinnyCfg<C>::BLOCK, D = SkinnyChunk<L>::D, NCH = K / D;
    static_assert(NCH * D == K, "chunking");
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int c0 = ((int)(blockIdx.x % HV) * (BL / 64) + (int)(threadIdx.x >> 6)) * 64 + 4 * i;
    const bool act = c0 + 3 < C;
    const int cc = act ? c0 : C - 4;
    const int n0 = (int)(blockIdx.x / HV) * npb;
    const int n1 = n0 + npb < N ? n0 + npb : N;
    floatx4 acc[L + 1][4];
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[l][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 < n1) {
        float4 bcur[D], bnxt[D];
        float acur[D], anxt[D];
        // chunk (q, ch): coefficients ch * D .. ch * D + D - 1 of the node quad q (nodes q + g); rows past the run are fed as
        // zeros on the A side (their B rows are read from the last valid node: finite values times zero)
        auto load = [&](float4* b, float* a, int q, int ch) {
            const int n = q + g;
            const bool ok = n < n1;
            const long long nn = ok ? n : n1 - 1;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int k = ch * D + d;
                b[d] = *reinterpret_cast<const float4*>(big + (nn * K + k) * C + cc);
                const float v = small[(nn * K + k) * 16 + i];
                a[d] = ok ? v : 0.f;
            }
        };
        load(bnxt, anxt, n0, 0);
        for (int q = n0; q < n1; q += 4) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
                for (int d = 0; d < D; ++d) bcur[d] = bnxt[d], acur[d] = anxt[d];
                if (ch + 1 < NCH) load(bnxt, anxt, q, ch + 1);
                else load(bnxt, anxt, q + 4 < n1 ? q + 4 : q, 0);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const int k = ch * D + d;
                    int l = 0;
                    while ((l + 1) * (l + 1) <= k) ++l;
                    const float bv[4] = {bcur[d].x, bcur[d].y, bcur[d].z, bcur[d].w};
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[l][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[d], bv[t], acc[l][t], 0, 0, 0);
                    if (k == 0 && q + g < n1) {
                        accb.x += bv[0]; accb.y += bv[1]; accb.z += bv[2]; accb.w += bv[3];
                    }
                }
            }
        }
    }
    constexpr int WSZ_ = (L + 1) * 16 * C;
    float* p = part + (long long)(blockIdx.x / HV) * (WSZ_ + (bias_row ? C : 0));
    if (bias_row) {                                                // the four lane groups hold the sums of their own nodes
        accb.x += __shfl_xor(accb.x, 16, 64); accb.y += __shfl_xor(accb.y, 16, 64); accb.z += __shfl_xor(accb.z, 16, 64); accb.w += __shfl_xor(accb.w, 16, 64);
        accb.x += __shfl_xor(accb.x, 32, 64); accb.y += __shfl_xor(accb.y, 32, 64); accb.z += __shfl_xor(accb.z, 32, 64); accb.w += __shfl_xor(accb.w, 32, 64);
    }
    if (!act) return;
    // accumulator register r of tile t: row u = 4 g + r, column i = channel cc + t
#pragma unroll
    for (int l = 0; l <= L; ++l) {
        if (out_cu) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                *reinterpret_cast<float4*>(p + ((long long)l * C + cc + t) * 16 + 4 * g) = make_float4(acc[l][t][0], acc[l][t][1], acc[l][t][2], acc[l][t][3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<float4*>(p + ((long long)l * 16 + 4 * g + r) * C + cc) = make_float4(acc[l][0][r], acc[l][1][r], acc[l][2][r], acc[l][3][r]);
        }
    }
    if (bias_row && g == 0) *reinterpret_cast<float4*>(p + WSZ_ + cc) = accb;
}

static inline int so3_skinny_npb(int N, int target_blocks) {
    int npb = (N + target_blocks - 1) / target_blocks;
    return npb < 8 ? 8 : npb;
}

// node runs of the reduction: as many as run concurrently in ONE round (every workgroup walks its nodes sequentially, so
// a partly filled second round would cost a full round's time)
template <int C>
static int so3_skinny_reduce_runs(int lmax) {
    static int cached[8] = {0};
    if (lmax < 0 || lmax > 7) return 512;
    if (cached[lmax]) return cached[lmax];
    int per_cu = 0, cus = 256;
    hipError_t e = hipErrorInvalidValue;
    constexpr int BL = SkinnyCfg<C>::BLOCK;
    if (lmax == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, so3_skinny_reduce_kernel<2, C>, BL, 0);
    if (lmax == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, so3_skinny_reduce_kernel<4, C>, BL, 0);
    if (lmax == 6) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, so3_skinny_reduce_kernel<6, C>, BL, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 2;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    cached[lmax] = per_cu * cus / SkinnyCfg<C>::HALVES;
    return cached[lmax];
}
static inline bool so3_skinny_channels_ok(int C) { return C == 512 || C == 112; }

// ------------------------------------------------------------------------------------------------ k12: equivariant RMS norm
// FOUR nodes per wavefront, C = 16: lane group q = lane >> 4 owns node 4 w + q, lane & 15 = channel; a lane holds its channel's K
// coefficient rows in registers (K independent loads in flight), the degree of a row is a compile-time constant of the
// unrolled loop, the sums over a node are K serial FMAs per lane + a 16-lane butterfly, and the per-wave partial parameter
// gradients are per-lane registers folded over the four groups at the end.  (Rounds 1-3: one node per wavefront, element idx =
// lane + 64 t with idx / C and a degree search per element at run time, 7 loads in flight, two 64-lane reductions per node and an
// LDS round trip for the per-degree sums: 68 / 91 us per launch on the 79 MB node tensors of config 3 = 2.3 / 2.6 TB/s.)
// forward:  x~ = x with its l = 0 row centred over the channels; r = (mean_c sum_k b_k x~^2 + eps)^-1/2, b_k = 1 / ((2l+1)(L+1));
//           y = x~ r w[l][c] (+ bias[c] on the l = 0 row)                                                   (EF:2155-2192, Q3)
// backward: gx~ = r w g - r^3 S b_k x~ / C with S = sum(g w x~); the l = 0 row then loses its channel mean (centering); ADD: + g_add,
//           the gradient of the residual branch that bypasses the norm.  Per-wave partials gw_part[wave][l][c] = sum g x~ r,
//           gb_part[wave][c] = sum g[0][c].
__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
__host__ __device__ constexpr int deg_of_c(int k) {
    int l = 0;
    while ((l + 1) * (l + 1) <= k) ++l;
    return l;
}
__host__ __device__ constexpr float bal_of_c(int k, int L) { return 1.0f / (float)((2 * deg_of_c(k) + 1) * (L + 1)); }

template <int L>
__global__ void __launch_bounds__(64) rmsnorm_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N, float eps) {
    constexpr int C = 16, K = (L + 1) * (L + 1), KC = K * C;
    const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
    float w[L + 1];
#pragma unroll
    for (int l = 0; l <= L; ++l) w[l] = weight[l * C + c];
    const float bc = bias[c];
    for (int n0 = 4 * (int)blockIdx.x; n0 < N; n0 += 4 * (int)gridDim.x) {
        const int n = n0 + q;
        const bool ok = n < N;
        const float* xi = x + (long long)(ok ? n : N - 1) * KC + c;
        float v[K];
#pragma clang loop unroll(full)
        for (int k = 0; k < K; ++k) v[k] = xi[k * C];
        v[0] -= group16_sum(v[0]) * (1.0f / C);
        float ss = 0.f;
#pragma clang loop unroll(full)
        for (int k = 0; k < K; ++k) ss = fmaf(v[k] * v[k], bal_of_c(k, L), ss);
        const float r = rsqrtf(group16_sum(ss) * (1.0f / C) + eps);
        if (ok) {
            float* yo = y + (long long)n * KC + c;
#pragma clang loop unroll(full)
            for (int k = 0; k < K; ++k) {
                const float rw = r * w[deg_of_c(k)];
                yo[k * C] = k == 0 ? fmaf(v[k], rw, bc) : v[k] * rw;
            }
        }
    }
}

template <int L, bool ADD>
__global__ void __launch_bounds__(64) rmsnorm_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                          const float* __restrict__ gy, const float* __restrict__ g_add,
                                                          float* __restrict__ gx, float* __restrict__ gw_part,
                                                          float* __restrict__ gb_part, int N, float eps) {
    constexpr int C = 16, K = (L + 1) * (L + 1), KC = K * C;
    const int lane = threadIdx.x, q = lane >> 4, c = lane & 15;
    float w[L + 1], gwp[L + 1];
#pragma unroll
    for (int l = 0; l <= L; ++l) { w[l] = weight[l * C + c]; gwp[l] = 0.f; }
    float gbp = 0.f;
    for (int n0 = 4 * (int)blockIdx.x; n0 < N; n0 += 4 * (int)gridDim.x) {
        const int n = n0 + q;
        const bool ok = n < N;
        const long long base = (long long)(ok ? n : N - 1) * KC + c;
        float v[K], g[K], ga[ADD ? K : 1];
#pragma clang loop unroll(full)
        for (int k = 0; k < K; ++k) {
            v[k] = x[base + k * C];
            g[k] = gy[base + k * C];
            if constexpr (ADD) ga[k] = g_add[base + k * C];
        }
        const float live = ok ? 1.f : 0.f;
        v[0] -= group16_sum(v[0]) * (1.0f / C);
        float ss = 0.f, gl[L + 1];
#pragma unroll
        for (int l = 0; l <= L; ++l) gl[l] = 0.f;
#pragma clang loop unroll(full)
        for (int k = 0; k < K; ++k) {
            ss = fmaf(v[k] * v[k], bal_of_c(k, L), ss);
            gl[deg_of_c(k)] = fmaf(g[k], v[k], gl[deg_of_c(k)]);      // sum of g x~ over the degree's rows
        }
        float S = 0.f;
#pragma unroll
        for (int l = 0; l <= L; ++l) S = fmaf(gl[l], w[l], S);
        gbp = fmaf(g[0], live, gbp);
        const float r = rsqrtf(group16_sum(ss) * (1.0f / C) + eps);
        S = group16_sum(S);
        const float r3s = r * r * r * S * (1.0f / C);
#pragma unroll
        for (int l = 0; l <= L; ++l) gwp[l] = fmaf(gl[l] * live, r, gwp[l]);
        float* go = gx + base;
        float d0 = fmaf(r * w[0], g[0], -r3s * bal_of_c(0, L) * v[0]);
        d0 -= group16_sum(d0) * (1.0f / C);
        if (ok) {
            go[0] = ADD ? d0 + ga[0] : d0;
#pragma clang loop unroll(full)
            for (int k = 1; k < K; ++k) {
                const float d = fmaf(r * w[deg_of_c(k)], g[k], -r3s * bal_of_c(k, L) * v[k]);
                go[k * C] = ADD ? d + ga[k] : d;
            }
        }
    }
    // fold the four lane groups' partials; one row of partials per wavefront
#pragma unroll
    for (int l = 0; l <= L; ++l) {
        gwp[l] += __shfl_xor(gwp[l], 16, 64);
        gwp[l] += __shfl_xor(gwp[l], 32, 64);
    }
    gbp += __shfl_xor(gbp, 16, 64);
    gbp += __shfl_xor(gbp, 32, 64);
    if (lane < C) {
        float* wp = gw_part + (long long)blockIdx.x * (L + 1) * C;
#pragma unroll
        for (int l = 0; l <= L; ++l) wp[l * C + c] = gwp[l];
        gb_part[(long long)blockIdx.x * C + c] = gbp;
    }
}


// ------------------------------------------------------------------------------------------------ n1: kNN edge attributes
// The CProMG encoders' edge features in their final layout (CP:295-298 after to_undirected): for every centre node i, whose
// undirected kNN edges are e in [ptr[i], ptr[i+1]) of the row-sorted list, the Gaussian-smeared lengths
// exp(coeff * (len[e] - offset[g])^2) NEGATED (get_laplacian's off-diagonal weights for 2-D edge weights, Q12) at row e + i
// of `out`, and their sum (the degree row of the appended self loop) at row ptr[i+1] + i.  One wavefront per node, lane =
// one of the 64 Gaussians: the [E + N, 64] tensor (768 MB at config 3) is written once - the torch form (smearing in five
// elementwise passes, negation, concatenations, a segmented sum, a gather into the sorted order) touched it eight times.
// Edges >= n_real (the inert padding edges of a padded batch) carry zeros.
__global__ void __launch_bounds__(64) knn_edge_attr_kernel(const float* __restrict__ len, const int* __restrict__ ptr, long long n_real,
                                                           const float* __restrict__ offset, float coeff, float* __restrict__ out, int N) {
    const int g = threadIdx.x;
    const float off = offset[g];
    SINGA_XCD_NODE_LOOP(i, N) {
        const int beg = ptr[i], end = ptr[i + 1];
        float deg = 0.f;
        int e = beg;
        for (; e + 3 < end; e += 4) {                     // four edges in flight
            float d[4], v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d[q] = (e + q < n_real ? len[e + q] : 0.f) - off;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sq = d[q] * d[q];
                v[q] = e + q < n_real ? expf(coeff * sq) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                out[((long long)(e + q) + i) * 64 + g] = -v[q];
                deg += v[q];
            }
        }
        for (; e < end; ++e) {
            const float dd = (e < n_real ? len[e] : 0.f) - off;
            const float sq = dd * dd;
            const float v = e < n_real ? expf(coeff * sq) : 0.f;
            out[((long long)e + i) * 64 + g] = -v;
            deg += v;
        }
        out[((long long)end + i) * 64 + g] = deg;
    }
}

// ------------------------------------------------------------------------------------------------ k7 / k11: f32 MFMA GEMM
// C[i, j] (+= bias[j]) = sum_r Aop[i, r] * Bop[r, j] on v_mfma_f32_32x32x2_f32 (exact f32: a k-ordered fmaf chain), for up to
// SINGA_GEMM_MAX independent problems per launch (the m = 0, 1, 2 blocks of an SO(2) convolution, EF:807-875; the
// degrees l of an SO3_LinearV2, EF:655-671).  Each operand is stored either "reduction-contiguous" ([i][r], e.g. an
// activation matrix X[E, K] or an nn.Linear weight W[out, in]) or "output-contiguous" ([r][i]):
//     forward   Y = X W^T     A = X  (RC)   B = W   (RC)
//     d input   dX = dY W     A = dY (RC)   B = W   (output-contiguous)
//     d weight  dW = dY^T X   A = dY (OC)   B = X   (OC), reduction over the edges, split over workgroups (partials)
// Workgroup = 256 threads; macro tile 128 x 128 (2 x 2 wavefronts of 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers)
// or one of the smaller tiles listed at the kernel; K step 32, two LDS buffers, two register sets (the loads of step s + 2
// are in flight during step s); LDS images and fragment reads as described in front of the kernel.  Tiles are numbered row
// tile by row tile and, inside, problem by problem; every XCD (blockIdx % 8) takes a contiguous range of tile ids, so its
// workgroups share the A rows of a row tile in that XCD's L2 and every XCD sees the same mix of problems.
// Rows of A and C may be grouped (row i -> (i / group) * ld_group + (i % group) * ld): the (2l+1) coefficient rows of a
// degree inside [N, K, C] node tensors.  Epilogue options per problem: bias[j]; ReLU; zeroing where a second tensor of C's
// layout is not positive (the gradient of a ReLU whose output was kept).
struct GemmProb {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* mask;      // same layout as C (plain rows): C is zeroed where mask <= 0
    const float* addend;    // same layout as C (plain or grouped rows): added to the product (after the bias, before the ReLU)
    float* asum;            // (0, 0) form only: per split, sum over the reduction rows of A -> asum[split * asum_stride + i]
    long long asum_stride;
    long long lda, ldb, ldc, a_gld, b_gld, c_gld, c_split;
    int I, J, R, a_group, b_group, c_group, tiles_j, tile_begin, relu;
};
struct GemmBatch {
    GemmProb p[SINGA_GEMM_MAX];
    int n, tiles_total, splits, tj_total;      // tj_total > 0: interleaved order (tiles_total = row tiles x column tiles of all problems)
    long long r_chunk;
};

// CFG 0: macro tile 128 x 128 (2 x 2 wavefronts of 64 x 64); CFG 1: 128 x 32 (4 x 1 wavefronts of 32 x 32) for outputs with
// few columns, CFG 2: 32 x 128 (1 x 4 wavefronts) for outputs with few rows (the 16-channel sides of SO3_LinearV2); CFG 3:
// 64 x 64 (2 x 2 wavefronts of 32 x 32) for launches whose 128 x 128 tiles would not fill the 256 CUs (the CProMG
// transformer's 256-wide projections on a few thousand rows).
//
// LDS images, K step 32.  A reduction-contiguous operand keeps its global layout, [row][r] with a pitch of 36 floats: global
// float4 -> ds_write_b128 with no transposition, and a lane's fragment for FOUR k-steps is ONE ds_read_b128 (row = lane % 32,
// r = 8 t + 4 (lane / 32) .. + 3; 9 sixteen-byte slots per row: the 16 lanes of a read group land in 16 different slots).
// The MFMA takes k from lanes 0-31 and k' from lanes 32-63 of its operand registers; WHICH reduction indices those are is
// free as long as both operands agree: MFMA (t, s) pairs r = 8 t + s with r = 8 t + 4 + s.  An output-contiguous operand is
// stored [r][col] (pitch + 4) and read with one ds_read_b32 per k-step at row 8 t + s + 4 (lane / 32) - the same pairing.
// Epilogue: an accumulator holds 4 consecutive ROWS of one column per register quad, so a direct store is 64 dword stores
// per lane; instead each wavefront passes its 32-row blocks through LDS (the K-loop buffers are free by then) and writes
// float4 rows - 4x fewer store instructions, full 256-byte row segments.
#ifndef SINGA_GEMM_PIPE
#define SINGA_GEMM_PIPE 1
#endif
#if SINGA_GEMM_PIPE   // the 128 x 128 tile takes the interleaved issue order (`pipeline` below) instead of the three fenced phases
#define SINGA_GEMM_FENCE() do { if constexpr (CFG != 0 || !decltype(all_c)::value) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SINGA_GEMM_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <bool A_RC, bool B_RC, int CFG>
__global__ void __launch_bounds__(256, 2) gemm_f32_kernel(GemmBatch gb) {
    constexpr int BM = CFG == 2 ? 32 : (CFG == 3 ? 64 : 128), BN = CFG == 1 ? 32 : (CFG == 3 ? 64 : 128), BK = 32, PR = BK + 4;
    constexpr int MT = CFG == 0 ? 2 : 1, NT = CFG == 0 ? 2 : 1;   // 32 x 32 MFMA tiles per wavefront
    constexpr int NA = BM / 32, NB = BN / 32;                      // float4 loads per thread and K step
    constexpr int LDA = BM + 4, LDB = BN + 4;                      // pitches of the [r][col] images
    constexpr int SZA = A_RC ? BM * PR : BK * LDA, SZB = B_RC ? BN * PR : BK * LDB;
    constexpr int PE = 32 * NT + 4;                                // epilogue staging pitch (floats)
    constexpr int SMEM = 2 * SZA + 2 * SZB > 4 * 32 * PE ? 2 * SZA + 2 * SZB : 4 * 32 * PE;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SZA;
    // XCD-aware numbering: block b runs on XCD b % 8; give every XCD a contiguous range of tile ids
    const int nblk = gb.tiles_total * gb.splits;
    const int per = (nblk + 7) >> 3;
    const int id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (id >= nblk) return;
    const int split = id / gb.tiles_total;
    const int tile = id - split * gb.tiles_total;
    // tiles are ordered row tile by row tile, and inside a row tile problem by problem: every XCD's contiguous range then
    // holds the same mix of problems (their reduction lengths differ: problem-by-problem ranges left some XCDs with only
    // the long ones and the launch waited for them)
    // (problems with equal row counts only; otherwise problem after problem: tj_total = 0)
    const int ti = gb.tj_total ? tile / gb.tj_total : 0, trem = gb.tj_total ? tile - ti * gb.tj_total : tile;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < SINGA_GEMM_MAX; ++q)
        if (q < gb.n && trem >= gb.p[q].tile_begin) pi = q;
    const GemmProb& P = gb.p[pi];
    const int local = trem - P.tile_begin;
    const int i0 = (gb.tj_total ? ti : local / P.tiles_j) * BM, j0 = (gb.tj_total ? local : local % P.tiles_j) * BN;
    const long long r_begin = (long long)split * gb.r_chunk;
    const long long r_end = (r_begin + gb.r_chunk < P.R) ? r_begin + gb.r_chunk : P.R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wrow = CFG == 0 ? (wave >> 1) * 64 : (CFG == 1 ? wave * 32 : (CFG == 3 ? (wave >> 1) * 32 : 0));   // the wavefront's block
    const int wcol = CFG == 0 ? (wave & 1) * 64 : (CFG == 1 ? 0 : (CFG == 3 ? (wave & 1) * 32 : wave * 32));
    const int l31 = lane & 31, half = lane >> 5;
    const int kq = tid & 7, rr = tid >> 3;                          // reduction-contiguous staging: float4 kq of row rr + 32 j

    // ---- per-thread global row pointers (loop invariant for reduction-contiguous operands; nullptr = outside the matrix)
    const float* arow[NA];
    const float* brow[NB];
    if (A_RC) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int i = i0 + rr + 32 * j;
            const long long off = P.a_group == (1 << 30) ? (long long)i * P.lda
                                                         : (long long)(i / P.a_group) * P.a_gld + (long long)(i % P.a_group) * P.lda;
            arow[j] = i < P.I ? P.A + off + 4 * kq : nullptr;
        }
    }
    if (B_RC) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int jj = j0 + rr + 32 * j;
            brow[j] = jj < P.J ? P.B + (long long)jj * P.ldb + 4 * kq : nullptr;
        }
    }
    // two register sets: the loads of K step s+2 are issued at the start of step s and written to LDS at the end of step
    // s+1, so every load has two steps (~8,000 MFMA cycles) to arrive - one step did not cover an HBM miss under load
    float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    // `full`: the tile lies inside the matrices and every K step is complete - loads without predicates (8 exec-mask
    // round trips per step less, and the compiler batches them)
    const bool full = i0 + BM <= P.I && j0 + BN <= P.J && r_end > r_begin && (r_end - r_begin) % BK == 0;
    // Ragged tiles / K steps and grouped rows: every load is still issued unconditionally - from a clamped, valid address -
    // and a per-register-set bit mask says which registers hold data; the zeros are selected when the registers are
    // written to LDS.  (Predicated loads put branches into the K loop, and the compiler then waits for ALL outstanding
    // loads - vmcnt(0) - before each LDS write: the two-step prefetch collapsed to none, 5.7 us per K step on the
    // grouped-row gradients of SO3_LinearV2.)
    auto load_rc = [&](auto all_c, float4* reg, unsigned& mask, const float* const* rows, const float* dummy, int n, long long r0) {
        if constexpr (decltype(all_c)::value) {
#pragma unroll
            for (int j = 0; j < n; ++j) reg[j] = *reinterpret_cast<const float4*>(rows[j] + r0);
            mask = ~0u;
        } else {
            const bool in = r0 + 4 * kq < r_end;
            mask = 0;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const bool ok = in && rows[j] != nullptr;
                const float* src = ok ? rows[j] + r0 : dummy + r_begin;      // dummy: the first row of the matrix
                reg[j] = *reinterpret_cast<const float4*>(src);
                mask |= (ok ? 1u : 0u) << j;
            }
        }
    };
    auto load_oc = [&](auto all_c, float4* reg, unsigned& mask, const float* base, long long ld, int grp, long long gld, int o0,
                       int lim, int n, int width4, long long r0) {
        const int c4 = tid % width4, rq = tid / width4, rows = 256 / width4;
        const int i = o0 + 4 * c4;
        if constexpr (decltype(all_c)::value) {           // host side: `full` tiles of ungrouped operands only
            const float* b = base + (r0 + rq) * ld + i;
#pragma unroll
            for (int j = 0; j < n; ++j) reg[j] = *reinterpret_cast<const float4*>(b + (long long)(rows * j) * ld);
            mask = ~0u;
        } else {
            const int ic = i < lim ? i : o0;
            mask = 0;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const long long r = r0 + rq + rows * j;
                const bool ok = r < r_end && i < lim;
                // R is an int: 32-bit unsigned division (a 64-bit one is ~150 instructions, per row and K step); an
                // ungrouped operand has grp = 2^30 > R: quotient 0, the same formula - no branch in the K loop
                const unsigned ru = (unsigned)(r < r_end ? r : r_begin), gq = ru / (unsigned)grp;
                const long long row = (long long)gq * gld + (long long)(ru - gq * (unsigned)grp) * ld;
                reg[j] = *reinterpret_cast<const float4*>(base + row + ic);
                mask |= (ok ? 1u : 0u) << j;
            }
        }
    };
    auto store_rc = [&](float* S, const float4* reg, unsigned mask, int n) {
#pragma unroll
        for (int j = 0; j < n; ++j)
            *reinterpret_cast<float4*>(S + (rr + 32 * j) * PR + 4 * kq) = (mask >> j) & 1u ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_oc = [&](float* S, const float4* reg, unsigned mask, int pitch, int n, int width4) {
        const int c4 = tid % width4, rq = tid / width4, rows = 256 / width4;
#pragma unroll
        for (int j = 0; j < n; ++j)
            *reinterpret_cast<float4*>(S + (rq + rows * j) * pitch + 4 * c4) = (mask >> j) & 1u ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_ab = [&](auto all_c, float4* ra, float4* rb, unsigned& ma, unsigned& mb, long long r0) __attribute__((always_inline)) {
        if (A_RC) load_rc(all_c, ra, ma, arow, P.A, NA, r0);
        else load_oc(all_c, ra, ma, P.A, P.lda, P.a_group, P.a_gld, i0, P.I, NA, BM / 4, r0);
        if (B_RC) load_rc(all_c, rb, mb, brow, P.B, NB, r0);
        else load_oc(all_c, rb, mb, P.B, P.ldb, P.b_group, P.b_gld, j0, P.J, NB, BN / 4, r0);
    };
    // (0, 0) form with `asum`: the column sums of A over the reduction rows - the bias gradient sum_m g[m, n] next to the weight
    // gradient g^T x - are taken from the registers on their way to LDS by the workgroups of the problem's first column tile:
    // four columns per thread, reduced over the workgroup in front of the epilogue.  (The separate column-sum pass it
    // replaces re-read every gradient tensor of the transformer's Linears: ~6 GB per step.)
    const bool do_asum = !A_RC && P.asum != nullptr && j0 == 0;
    const float asum_on = do_asum ? 1.f : 0.f;
    float4 asum4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_ab = [&](const float4* ra, const float4* rb, unsigned ma, unsigned mb, int buf) __attribute__((always_inline)) {
        if constexpr (!A_RC) {           // branch-free (a branch here would split the scheduled K step): weight 0 when not wanted
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const float wj = ((ma >> j) & 1u) ? asum_on : 0.f;
                asum4.x = fmaf(wj, ra[j].x, asum4.x); asum4.y = fmaf(wj, ra[j].y, asum4.y);
                asum4.z = fmaf(wj, ra[j].z, asum4.z); asum4.w = fmaf(wj, ra[j].w, asum4.w);
            }
        }
        if (A_RC) store_rc(As0 + buf * SZA, ra, ma, NA);
        else store_oc(As0 + buf * SZA, ra, ma, LDA, NA, BM / 4);
        if (B_RC) store_rc(Bs0 + buf * SZB, rb, mb, NB);
        else store_oc(Bs0 + buf * SZB, rb, mb, LDB, NB, BN / 4);
    };
    // fragments of k-group t (8 reduction indices = 4 MFMA k-steps) for the wavefront's MT / NT 32-wide blocks
    const int ia = wrow + l31, jb = wcol + l31;
    auto frag = [&](const float* S, bool rc, int pitch, int col, int t, float (&f)[4]) __attribute__((always_inline)) {
        if (rc) {
            const float4 v = *reinterpret_cast<const float4*>(S + col * PR + 8 * t + 4 * half);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = S[(8 * t + q + 4 * half) * pitch + col];
        }
    };

    floatx16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // nt8: k-groups of 8 reduction indices that hold data in this step (4, fewer in a ragged last step: a reduction of
    // 16 - the 16-channel side of SO3_LinearV2 - then costs 32 MFMAs per wavefront instead of 64 on zero padding)
    // 32-wide blocks of this wavefront's tile that lie inside the matrices: a ragged tile (the last column tile of a 560- or
    // 160-wide output, the last row tile) issues no MFMAs for blocks that only hold padding - they would be a quarter of the
    // work of the 160 / 192-wide conv1 outputs and an eighth of conv2's
    unsigned blkv = 0;                                             // bit a * NT + b: block (a, b) holds at least one output
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
            blkv |= ((i0 + wrow + 32 * a < P.I) && (j0 + wcol + 32 * b < P.J) ? 1u : 0u) << (a * NT + b);
    blkv = __builtin_amdgcn_readfirstlane(blkv);
    auto compute = [&](auto edge_c, int buf, int nt8) __attribute__((always_inline)) {
        const float* Sa = As0 + buf * SZA;
        const float* Sb = Bs0 + buf * SZB;
        float fa[2][MT][4], fb[2][NT][4];
#pragma unroll
        for (int a = 0; a < MT; ++a) frag(Sa, A_RC, LDA, ia + 32 * a, 0, fa[0][a]);
#pragma unroll
        for (int b = 0; b < NT; ++b) frag(Sb, B_RC, LDB, jb + 32 * b, 0, fb[0][b]);
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            if (t >= nt8) break;
            const int cur = t & 1;
            if (t + 1 < BK / 8) {            // the next group's fragments travel behind this group's MFMAs
#pragma unroll
                for (int a = 0; a < MT; ++a) frag(Sa, A_RC, LDA, ia + 32 * a, t + 1, fa[cur ^ 1][a]);
#pragma unroll
                for (int b = 0; b < NT; ++b) frag(Sb, B_RC, LDB, jb + 32 * b, t + 1, fb[cur ^ 1][b]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        if constexpr (decltype(edge_c)::value) {
                            if ((blkv >> (a * NT + b)) & 1u)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][a][q], fb[cur][b][q], acc[a][b], 0, 0, 0);
                        } else {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][a][q], fb[cur][b][q], acc[a][b], 0, 0, 0);
                        }
                    }
        }
    };

    const long long nsteps = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;
    const int last8 = nsteps > 0 ? (int)((r_end - r_begin - (nsteps - 1) * BK + 7) / 8) : 0;   // k-groups of the last step
    // Step st computes LDS buffer st & 1; register set (st + 1) & 1 holds step st + 1 (written to LDS at the end of this
    // step), the other set receives step st + 2.  The steady-state loop has NO branch around its loads and the scheduler
    // may not move anything across the marks: only then does the compiler wait with a COUNT (the older set's loads) before
    // the LDS writes instead of vmcnt(0), i.e. only then are two steps of loads really in flight.  (With `if (st + 2 <
    // nsteps) load` inside one loop it issued vmcnt(0) at every join and hoisted the LDS writes above the new loads.)
    // Issue order of one steady-state K step of the 128 x 128 tile (SINGA_GEMM_PIPE): the step's 8 global loads, its LDS
    // fragment reads and its 8 LDS writes are spread BETWEEN the 64 MFMAs instead of standing in front of and behind them.
    // An MFMA occupies the matrix core for 64 cycles while the wavefront goes on issuing: with the three phases fenced off
    // (load | 64 MFMAs | store) a wavefront issued no MFMA during ~15 % of a step and the pipe only stayed busy when the
    // CU's second workgroup happened to be in its MFMA phase (MFMA-busy 0.75).  Groups (llvm.amdgcn.sched.group.barrier):
    // k-group 0: 2 MFMA + 1 load, 8 times, with the fragment reads of k-group 1 among them; k-groups 1, 2: 16 MFMAs with
    // the next group's fragment reads; k-group 3: 2 MFMA + 1 LDS write, 8 times.
    auto pipeline = [&](auto interior_c) __attribute__((always_inline)) {
#if SINGA_GEMM_PIPE
        if constexpr (CFG == 0 && decltype(interior_c)::value) {
            constexpr int DSR = (A_RC ? MT : 4 * MT) + (B_RC ? NT : 4 * NT);      // fragment reads per k-group
            constexpr int MF = 0x008, VM = 0x020, DR = 0x100, DW = 0x200;
            __builtin_amdgcn_sched_group_barrier(DR, DSR, 0);                      // k-group 0's fragments first
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(MF, 2, 0);
                __builtin_amdgcn_sched_group_barrier(VM, 1, 0);
                if (i % 2 == 1) __builtin_amdgcn_sched_group_barrier(DR, DSR / 4, 0);
            }
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(MF, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(DR, DSR / 4, 0);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(MF, 2, 0);
                __builtin_amdgcn_sched_group_barrier(DW, 1, 0);
            }
        }
#endif
    };
    auto k_loop = [&](auto all_c) __attribute__((always_inline)) {
        constexpr std::integral_constant<bool, !decltype(all_c)::value> edge_c{};      // ragged tiles skip padding blocks
        unsigned ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;
        if (nsteps > 0) {
            load_ab(all_c, ra0, rb0, ma0, mb0, r_begin);
            if (nsteps > 1) load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + BK);
            store_ab(ra0, rb0, ma0, mb0, 0);
        }
        __syncthreads();
        long long st = 0;
#if !defined(SINGA_GEMM_LAB_NOLOAD) && !defined(SINGA_GEMM_LAB_NOSYNC)
        // all prologue loads land before the loop: otherwise the compiler's wait insertion merges "a prologue load into
        // register X may be pending" into the loop header and waits for vmcnt(0) there on EVERY iteration
        if (nsteps > 3) __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) only
        for (; st + 3 < nsteps; st += 2) {
            load_ab(all_c, ra0, rb0, ma0, mb0, r_begin + (st + 2) * BK);
            SINGA_GEMM_FENCE();
            compute(edge_c, 0, 4);
            SINGA_GEMM_FENCE();
            store_ab(ra1, rb1, ma1, mb1, 1);
            pipeline(all_c);
            __syncthreads();
            load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + (st + 3) * BK);
            SINGA_GEMM_FENCE();
            compute(edge_c, 1, 4);
            SINGA_GEMM_FENCE();
            store_ab(ra0, rb0, ma0, mb0, 0);
            pipeline(all_c);
            __syncthreads();
        }
#endif
        for (; st < nsteps; st += 2) {         // the last (up to three) steps, and the tools/lab variants
#if defined(SINGA_GEMM_LAB_NOLOAD) || defined(SINGA_GEMM_LAB_NOSYNC)   // tools/lab only: which part of a step costs what
            compute(edge_c, 0, 4);
#ifndef SINGA_GEMM_LAB_NOSYNC
            if (st + 1 < nsteps) store_ab(ra1, rb1, ma1, mb1, 1);
            __syncthreads();
#endif
            if (st + 1 >= nsteps) break;
            compute(edge_c, 1, 4);
#ifndef SINGA_GEMM_LAB_NOSYNC
            if (st + 2 < nsteps) store_ab(ra0, rb0, ma0, mb0, 0);
            __syncthreads();
#endif
#else
            if (st + 2 < nsteps) load_ab(all_c, ra0, rb0, ma0, mb0, r_begin + (st + 2) * BK);
            compute(edge_c, 0, st + 1 == nsteps ? last8 : 4);
            if (st + 1 < nsteps) store_ab(ra1, rb1, ma1, mb1, 1);
            __syncthreads();
            if (st + 1 >= nsteps) break;
            if (st + 3 < nsteps) load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + (st + 3) * BK);
            compute(edge_c, 1, st + 2 == nsteps ? last8 : 4);
            if (st + 2 < nsteps) store_ab(ra0, rb0, ma0, mb0, 0);
            __syncthreads();
#endif
        }
    };
    // `full` tiles of ungrouped operands take the unpredicated loads
    if (full && P.a_group == (1 << 30) && P.b_group == (1 << 30)) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    if (!A_RC && do_asum) {          // workgroup-uniform: reduce the threads' column sums over the row groups, one slab row per split
        constexpr int W4 = BM / 4, RG = 256 / W4;
        float* red = smem;                                          // [RG][BM] (the K-loop buffers are free behind its last barrier)
        *reinterpret_cast<float4*>(red + (tid / W4) * BM + 4 * (tid % W4)) = asum4;
        __syncthreads();
        if (tid < BM) {
            float t = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < RG; ++g2) t += red[g2 * BM + tid];
            if (i0 + tid < P.I) P.asum[(long long)split * P.asum_stride + i0 + tid] = t;
        }
        __syncthreads();
    }
    // ---- epilogue.  Accumulator register q of a 32 x 32 tile holds row (q & 3) + 8 (q >> 2) + 4 half, column lane & 31.
    // Every wavefront owns 32 x PE floats of LDS (all K-loop reads are behind the barrier above); one 32-row block at a time:
    // registers -> [row][col] image -> float4 rows -> global, bias added on the way.
    float* Cb = P.C + (long long)split * P.c_split;
    float* stage = smem + wave * (32 * PE);
    constexpr int C4 = 8 * NT;                                       // float4 per staged row
    constexpr int ROWS_PER_PASS = 64 / C4, PASSES = 32 / ROWS_PER_PASS;
    const int c4 = lane % C4, rsub = lane / C4;
    const int jcol = j0 + wcol + 4 * c4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.bias && jcol < P.J) bias4 = *reinterpret_cast<const float4*>(P.bias + jcol);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) stage[((q & 3) + 8 * (q >> 2) + 4 * half) * PE + 32 * b + l31] = acc[a][b][q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int rloc = rsub + ROWS_PER_PASS * ps;
            const int i = i0 + wrow + 32 * a + rloc;
            float4 v = *reinterpret_cast<const float4*>(stage + rloc * PE + 4 * c4);
            if (i < P.I && jcol < P.J) {
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                const long long row = P.c_group == (1 << 30) ? (long long)i * P.ldc
                                                             : (long long)(i / P.c_group) * P.c_gld + (long long)(i % P.c_group) * P.ldc;
                if (P.addend) {
                    const float4 ad = *reinterpret_cast<const float4*>(P.addend + row + jcol);
                    v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                }
                if (P.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (P.mask) {
                    const float4 m = *reinterpret_cast<const float4*>(P.mask + row + jcol);
                    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(Cb + row + jcol) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ k7c: complex f32 MFMA GEMM (3M)
// The order-m > 0 blocks of an SO(2) convolution (EF:677-729) are COMPLEX products: with x = x_+m + i x_-m and W = Wr + i Wi
//     out_re = x_re Wr^T - x_im Wi^T,   out_im = x_re Wi^T + x_im Wr^T.
// As one real GEMM on the block weight [[Wr, -Wi], [Wi, Wr]] (what k7 did) that is four real products; here three:
//     P1 = (a + b) c,  P2 = a (d - c),  P3 = b (c + d):   re = P1 - P3,  im = P1 + P2        (a + i b)(c + i d)
// - a quarter of the MFMA work of 80 % of the convolution's flops.  The forward pass, d input and d weight are the same product
// with different operand forms (sigma = the sign on the B side's imaginary part):
//     forward   (a, b) = x_re, x_im [E, K]      (c, d) = Wr, Wi [N, K]           sigma = +1      A RC, B RC
//     d input   (a, b) = g_re, g_im [E, N]      (c, d) = Wr, Wi ([r = n][j = k])  sigma = -1      A RC, B OC
//     d weight  (a, b) = g_re, g_im ([r = e][i = n])   (c, d) = x_re, x_im ([r = e][j = k])  sigma = -1   A OC, B OC, split over e
// The imaginary part of every operand lies a fixed number of elements behind the real part (a_im, b_im, c_im).  Both parts of
// both operands are staged in LDS as they are; the sums a + b, sigma d - c, c + sigma d are formed on the fragments in
// registers (8 + 8 VALU operations per 24 MFMAs).  Workgroup = 256 threads = 2 x 2 wavefronts, tile 128 rows x 64 complex
// columns, K step 16 (complex), three accumulator sets of 2 MFMA tiles (96 registers), LDS images / two-step register prefetch /
// XCD-contiguous tile ranges / staged float4 epilogue as in k7.
struct CGemmProb {
    const float* A;
    const float* B;
    float* C;
    long long lda, ldb, ldc, a_im, b_im, c_im, c_split;
    int I, J, R, tiles_j, tile_begin;
    float sigma;
};
struct CGemmBatch {
    CGemmProb p[SINGA_CGEMM_MAX];
    int n, tiles_total, splits, tj_total;
    long long r_chunk;
};

#ifndef SINGA_CGEMM_PIPE
#define SINGA_CGEMM_PIPE 1
#endif
#if SINGA_CGEMM_PIPE
#define SINGA_CGEMM_FENCE() do { if constexpr (!decltype(all_c)::value) __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SINGA_CGEMM_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <bool A_RC, bool B_RC>
__global__ void __launch_bounds__(256, 2) cgemm3m_f32_kernel(CGemmBatch gb) {
    constexpr int BM = 128, BN = 64, BK = 16, PR = BK + 4, LDA = BM + 4, LDB = BN + 4;
    constexpr int NA = 4, NB = 2;                                   // float4 loads per thread and K step (both parts)
    constexpr int SZA1 = A_RC ? BM * PR : BK * LDA, SZB1 = B_RC ? BN * PR : BK * LDB;      // one part's image
    constexpr int SZA = 2 * SZA1, SZB = 2 * SZB1;
    constexpr int PE = 64 + 4;                                      // epilogue staging pitch: 32 real + 32 imaginary columns
    constexpr int SMEM = 2 * SZA + 2 * SZB > 4 * 32 * PE ? 2 * SZA + 2 * SZB : 4 * 32 * PE;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * SZA;
    const int nblk = gb.tiles_total * gb.splits;
    const int per = (nblk + 7) >> 3;
    const int id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (id >= nblk) return;
    const int split = id / gb.tiles_total;
    const int tile = id - split * gb.tiles_total;
    const int ti = gb.tj_total ? tile / gb.tj_total : 0, trem = gb.tj_total ? tile - ti * gb.tj_total : tile;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < SINGA_CGEMM_MAX; ++q)
        if (q < gb.n && trem >= gb.p[q].tile_begin) pi = q;
    const CGemmProb& P = gb.p[pi];
    const int local = trem - P.tile_begin;
    const int i0 = (gb.tj_total ? ti : local / P.tiles_j) * BM, j0 = (gb.tj_total ? local : local % P.tiles_j) * BN;
    const long long r_begin = (long long)split * gb.r_chunk;
    const long long r_end = (r_begin + gb.r_chunk < P.R) ? r_begin + gb.r_chunk : P.R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 32;
    const int l31 = lane & 31, half = lane >> 5;
    const float sigma = P.sigma;

    // ---- staging maps.  Reduction-contiguous operand: float4 kq (of 4) of part pt of row rr + 32 j.  Output-contiguous operand:
    // float4 c4 of reduction row rq (+ rows per pass) of part j / (n / 2).
    const int kq = tid & 3, pt = (tid >> 2) & 1, rr = tid >> 3;
    const float* arow[NA];
    const float* brow[NB];
    if (A_RC) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int i = i0 + rr + 32 * j;
            arow[j] = i < P.I ? P.A + (long long)i * P.lda + (long long)pt * P.a_im + 4 * kq : nullptr;
        }
    }
    if (B_RC) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int jj = j0 + rr + 32 * j;
            brow[j] = jj < P.J ? P.B + (long long)jj * P.ldb + (long long)pt * P.b_im + 4 * kq : nullptr;
        }
    }
    float4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    const bool full = i0 + BM <= P.I && j0 + BN <= P.J && r_end > r_begin && (r_end - r_begin) % BK == 0;
    auto load_rc = [&](auto all_c, float4* reg, unsigned& mask, const float* const* rows, const float* dummy, int n, long long r0) {
        if constexpr (decltype(all_c)::value) {
#pragma unroll
            for (int j = 0; j < n; ++j) reg[j] = *reinterpret_cast<const float4*>(rows[j] + r0);
            mask = ~0u;
        } else {
            const bool in = r0 + 4 * kq < r_end;
            mask = 0;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const bool ok = in && rows[j] != nullptr;
                const float* src = ok ? rows[j] + r0 : dummy + r_begin;
                reg[j] = *reinterpret_cast<const float4*>(src);
                mask |= (ok ? 1u : 0u) << j;
            }
        }
    };
    // output-contiguous: `width4` float4 per reduction row and part, 256 / width4 reduction rows per pass, n / 2 passes per part
    auto load_oc = [&](auto all_c, float4* reg, unsigned& mask, const float* base, long long ld, long long im, int o0, int lim, int n,
                       int width4, long long r0) {
        const int c4 = tid % width4, rq = tid / width4, rows = 256 / width4;
        const int i = o0 + 4 * c4;
        if constexpr (decltype(all_c)::value) {
            const float* b = base + (r0 + rq) * ld + i;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const int part = j / (n / 2), ps = j % (n / 2);
                reg[j] = *reinterpret_cast<const float4*>(b + (long long)part * im + (long long)(rows * ps) * ld);
            }
            mask = ~0u;
        } else {
            const int ic = i < lim ? i : o0;
            mask = 0;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const int part = j / (n / 2), ps = j % (n / 2);
                const long long r = r0 + rq + rows * ps;
                const bool ok = r < r_end && i < lim;
                const long long rc = r < r_end ? r : r_begin;
                reg[j] = *reinterpret_cast<const float4*>(base + rc * ld + (long long)part * im + ic);
                mask |= (ok ? 1u : 0u) << j;
            }
        }
    };
    auto store_rc = [&](float* S, int sz1, const float4* reg, unsigned mask, int n) {
#pragma unroll
        for (int j = 0; j < n; ++j)
            *reinterpret_cast<float4*>(S + pt * sz1 + (rr + 32 * j) * PR + 4 * kq) =
                (mask >> j) & 1u ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_oc = [&](float* S, int sz1, const float4* reg, unsigned mask, int pitch, int n, int width4) {
        const int c4 = tid % width4, rq = tid / width4, rows = 256 / width4;
#pragma unroll
        for (int j = 0; j < n; ++j) {
            const int part = j / (n / 2), ps = j % (n / 2);
            *reinterpret_cast<float4*>(S + part * sz1 + (rq + rows * ps) * pitch + 4 * c4) =
                (mask >> j) & 1u ? reg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto load_ab = [&](auto all_c, float4* ra, float4* rb, unsigned& ma, unsigned& mb, long long r0) __attribute__((always_inline)) {
        if (A_RC) load_rc(all_c, ra, ma, arow, P.A, NA, r0);
        else load_oc(all_c, ra, ma, P.A, P.lda, P.a_im, i0, P.I, NA, BM / 4, r0);
        if (B_RC) load_rc(all_c, rb, mb, brow, P.B, NB, r0);
        else load_oc(all_c, rb, mb, P.B, P.ldb, P.b_im, j0, P.J, NB, BN / 4, r0);
    };
    auto store_ab = [&](const float4* ra, const float4* rb, unsigned ma, unsigned mb, int buf) __attribute__((always_inline)) {
        if (A_RC) store_rc(As0 + buf * SZA, SZA1, ra, ma, NA);
        else store_oc(As0 + buf * SZA, SZA1, ra, ma, LDA, NA, BM / 4);
        if (B_RC) store_rc(Bs0 + buf * SZB, SZB1, rb, mb, NB);
        else store_oc(Bs0 + buf * SZB, SZB1, rb, mb, LDB, NB, BN / 4);
    };
    const int ia = wrow + l31, jb = wcol + l31;
    auto frag = [&](const float* S, bool rc, int pitch, int col, int t, float (&f)[4]) __attribute__((always_inline)) {
        if (rc) {
            const float4 v = *reinterpret_cast<const float4*>(S + col * PR + 8 * t + 4 * half);
            f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = S[(8 * t + q + 4 * half) * pitch + col];
        }
    };
    floatx16 acc1[2], acc2[2], acc3[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[a][r] = acc2[a][r] = acc3[a][r] = 0.f;
    const bool row1 = __builtin_amdgcn_readfirstlane((int)(i0 + wrow + 32 < P.I)) != 0;      // the wavefront's second 32-row block holds rows
    const bool colv = __builtin_amdgcn_readfirstlane((int)(j0 + wcol < P.J)) != 0;           // ... its 32-column block holds columns
    auto compute = [&](auto all_c, int buf) __attribute__((always_inline)) {
        if constexpr (!decltype(all_c)::value) {
            if (!colv) return;                   // (a 96-wide result: the last tile's second column block is padding only)
        }
        const float* Sa = As0 + buf * SZA;
        const float* Sb = Bs0 + buf * SZB;
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            float ar[2][4], ai[2][4], br[4], bi[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                frag(Sa, A_RC, LDA, ia + 32 * a, t, ar[a]);
                frag(Sa + SZA1, A_RC, LDA, ia + 32 * a, t, ai[a]);
            }
            frag(Sb, B_RC, LDB, jb, t, br);
            frag(Sb + SZB1, B_RC, LDB, jb, t, bi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float ds = sigma * bi[q];
                const float b2 = ds - br[q], b3 = ds + br[q];
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[0][q] + ai[0][q], br[q], acc1[0], 0, 0, 0);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[0][q], b2, acc2[0], 0, 0, 0);
                acc3[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[0][q], b3, acc3[0], 0, 0, 0);
                if (decltype(all_c)::value || row1) {
                    acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[1][q] + ai[1][q], br[q], acc1[1], 0, 0, 0);
                    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[1][q], b2, acc2[1], 0, 0, 0);
                    acc3[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[1][q], b3, acc3[1], 0, 0, 0);
                }
            }
        }
    };
    const long long nsteps = r_end > r_begin ? (r_end - r_begin + BK - 1) / BK : 0;
    // Issue order of one steady-state K step of a full tile (as k7's `pipeline`): the 6 global loads and the 6 LDS writes are
    // spread between the 48 MFMAs, the second k-group's fragment reads travel behind the first group's MFMAs.
    auto pipeline = [&](auto interior_c) __attribute__((always_inline)) {
#if SINGA_CGEMM_PIPE
        if constexpr (decltype(interior_c)::value) {
            constexpr int DSR = (A_RC ? 4 : 16) + (B_RC ? 2 : 8);                  // fragment reads per k-group
            constexpr int MF = 0x008, VM = 0x020, DR = 0x100, DW = 0x200;
            __builtin_amdgcn_sched_group_barrier(DR, DSR, 0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(MF, 4, 0);
                __builtin_amdgcn_sched_group_barrier(VM, 1, 0);
                __builtin_amdgcn_sched_group_barrier(DR, (DSR + 5) / 6, 0);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                __builtin_amdgcn_sched_group_barrier(MF, 4, 0);
                __builtin_amdgcn_sched_group_barrier(DW, 1, 0);
            }
        }
#endif
    };
    auto k_loop = [&](auto all_c) __attribute__((always_inline)) {
        unsigned ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;
        if (nsteps > 0) {
            load_ab(all_c, ra0, rb0, ma0, mb0, r_begin);
            if (nsteps > 1) load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + BK);
            store_ab(ra0, rb0, ma0, mb0, 0);
        }
        __syncthreads();
        long long st = 0;
        if (nsteps > 3) __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) only (see k7)
        for (; st + 3 < nsteps; st += 2) {
            load_ab(all_c, ra0, rb0, ma0, mb0, r_begin + (st + 2) * BK);
            SINGA_CGEMM_FENCE();
            compute(all_c, 0);
            SINGA_CGEMM_FENCE();
            store_ab(ra1, rb1, ma1, mb1, 1);
            pipeline(all_c);
            __syncthreads();
            load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + (st + 3) * BK);
            SINGA_CGEMM_FENCE();
            compute(all_c, 1);
            SINGA_CGEMM_FENCE();
            store_ab(ra0, rb0, ma0, mb0, 0);
            pipeline(all_c);
            __syncthreads();
        }
        for (; st < nsteps; st += 2) {
            if (st + 2 < nsteps) load_ab(all_c, ra0, rb0, ma0, mb0, r_begin + (st + 2) * BK);
            compute(all_c, 0);
            if (st + 1 < nsteps) store_ab(ra1, rb1, ma1, mb1, 1);
            __syncthreads();
            if (st + 1 >= nsteps) break;
            if (st + 3 < nsteps) load_ab(all_c, ra1, rb1, ma1, mb1, r_begin + (st + 3) * BK);
            compute(all_c, 1);
            if (st + 2 < nsteps) store_ab(ra0, rb0, ma0, mb0, 0);
            __syncthreads();
        }
    };
    if (full) k_loop(std::true_type{});
    else k_loop(std::false_type{});
    // ---- epilogue: re = P1 - P3, im = P1 + P2; per wavefront one 32-row block at a time through LDS ([row][32 re | 32 im]),
    // then float4 rows: 8 float4 of the real half and 8 of the imaginary half (c_im elements behind) per row
    float* Cb = P.C + (long long)split * P.c_split;
    float* stage = smem + wave * (32 * PE);
    const int c4 = lane & 15, rsub = lane >> 4;
    const int part = c4 >> 3, jcol = j0 + wcol + 4 * (c4 & 7);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rw = (q & 3) + 8 * (q >> 2) + 4 * half;
            stage[rw * PE + l31] = acc1[a][q] - acc3[a][q];
            stage[rw * PE + 32 + l31] = acc1[a][q] + acc2[a][q];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int rloc = rsub + 4 * ps;
            const int i = i0 + wrow + 32 * a + rloc;
            const float4 v = *reinterpret_cast<const float4*>(stage + rloc * PE + 4 * c4);
            if (i < P.I && jcol < P.J)
                *reinterpret_cast<float4*>(Cb + (long long)i * P.ldc + (long long)part * P.c_im + jcol) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ small weight / node transforms
// SO2_m_Convolution's Linear acts on [x_+m | x_-m] through the "complex" recombination (out_r = fc_r(x_+) - fc_i(x_-),
// out_i = fc_r(x_-) + fc_i(x_+), EF:721-729); folded into ONE block weight B = [[Wr, -Wi], [Wi, Wr]] (Wr = w[:h], Wi = w[h:])
// the convolution is a single GEMM.  Built by torch ops (two slices, a negation, three concatenations) this cost ~13
// launches per module and step, forward and backward; here it is one launch each way.
__global__ void __launch_bounds__(256) block_weight_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int h, int k) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4LL * h * k) return;
    const int i = (int)(t / (2 * k)), j = (int)(t - (long long)i * 2 * k);
    float v;
    if (i < h) v = j < k ? w[(long long)i * k + j] : -w[(long long)(h + i) * k + (j - k)];
    else v = j < k ? w[(long long)i * k + j] : w[(long long)(i - h) * k + (j - k)];
    out[t] = v;
}
// gw[:h] = G[:h, :k] + G[h:, k:],  gw[h:] = G[h:, :k] - G[:h, k:]   (accumulate != 0: added to gw)
__global__ void __launch_bounds__(256) block_weight_bwd_kernel(const float* __restrict__ G, float* __restrict__ gw, int h, int k, int accumulate) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2LL * h * k) return;
    const int i = (int)(t / k), j = (int)(t - (long long)i * k);
    float v;
    if (i < h) v = G[(long long)i * 2 * k + j] + G[(long long)(h + i) * 2 * k + k + j];
    else v = G[(long long)i * 2 * k + j] - G[(long long)(i - h) * 2 * k + k + j];
    gw[t] = accumulate ? gw[t] + v : v;
}

// out[m] = scale * sum_d x[m, d] b[d] for D = 32 (the hoisted bias term q . b of the graph attention's logits, CP:61-65):
// 8 lanes per row, one float4 each.  Backward: gx[m, d] = scale g[m] b[d]; per-workgroup partial sums of the bias gradient
// scale g[m] x[m, d] -> part[blocks][32] (added up by the step's shared column-sum launch).
__global__ void __launch_bounds__(256) rowdot32_fwd_kernel(const float* __restrict__ x, const float* __restrict__ b, float* __restrict__ out,
                                                           long long M, float scale) {
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int q = threadIdx.x & 7;
    float s = 0.f;
    if (row < M) {
        const float4 xv = *reinterpret_cast<const float4*>(x + row * 32 + 4 * q);
        const float4 bv = *reinterpret_cast<const float4*>(b + 4 * q);
        s = xv.x * bv.x + xv.y * bv.y + xv.z * bv.z + xv.w * bv.w;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (row < M && q == 0) out[row] = s * scale;
}
// rows per workgroup of the backward kernel (2,048 at first: 91 workgroups for the 186 k (node, head) rows of config 3, each
// streaming 512 KB in 64 dependent passes - 42 us for 48 MB; with 256 rows the partial buffer is 8x longer and still small)
constexpr int ROWDOT_ROWS = 256;
__global__ void __launch_bounds__(256) rowdot32_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ b,
                                                           float* __restrict__ gx, float* __restrict__ part, long long M, float scale) {
    __shared__ float red[32][33];
    const int q = threadIdx.x & 7, r = threadIdx.x >> 3;            // 32 rows per pass, 8 lanes each
    const float4 bv = *reinterpret_cast<const float4*>(b + 4 * q);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long r0 = (long long)blockIdx.x * ROWDOT_ROWS;
    for (int p = 0; p < ROWDOT_ROWS / 32; ++p) {
        const long long row = r0 + p * 32 + r;
        if (row < M) {
            const float gs = g[row] * scale;
            const float4 xv = *reinterpret_cast<const float4*>(x + row * 32 + 4 * q);
            *reinterpret_cast<float4*>(gx + row * 32 + 4 * q) = make_float4(gs * bv.x, gs * bv.y, gs * bv.z, gs * bv.w);
            acc.x = fmaf(gs, xv.x, acc.x); acc.y = fmaf(gs, xv.y, acc.y); acc.z = fmaf(gs, xv.z, acc.z); acc.w = fmaf(gs, xv.w, acc.w);
        }
    }
    red[r][4 * q] = acc.x, red[r][4 * q + 1] = acc.y, red[r][4 * q + 2] = acc.z, red[r][4 * q + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int i = 0; i < 32; ++i) s += red[i][threadIdx.x];
        part[(long long)blockIdx.x * 32 + threadIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------ n2: Laplacian eigenvectors
// dgl.lap_pe / `lap_pe` of the reference (model/CProMG.py:562-571, called inside forward at model/GAN.py:71,77): the k
// eigenvectors after the smallest of the normalised Laplacian I - D^-1/2 A D^-1/2 of every graph of the batch, from the
// graph's edge list.  One workgroup of 1024 threads per graph, fp64, dense symmetric storage (n <= 896 atoms: a few MB,
// cache resident), but only the diagonal blocks of the graph's CONNECTED COMPONENTS are ever touched: a bonded pocket graph
// has dozens of components (a 45-fold zero eigenvalue is typical), and all of the O(n^3) work below is per component.
//   0. component labels by minimum-label propagation over the edges with pointer jumping (LDS); positions = atoms sorted by
//      (component, index); everything below works in positions, `perm` maps them back.
//   1. the Laplacian's component blocks are built in place: zero, raw adjacency, in-degrees (clipped at 1), scaling and
//      symmetrisation.
//   2. Householder tridiagonalisation per component.  Full symmetric storage, so that "column r" is read as row r; the
//      rank-2 update of step j and the matrix-vector product of step j + 1 are ONE pass over the trailing block (the row that
//      defines reflector j + 1 is updated first): 16 bytes of traffic per trailing element and step.  Reflector j stays
//      in row j right of the diagonal, its factor in beta[j].  Off-diagonal 0 between components.
//   3. the k + 1 smallest eigenvalues of the (block) tridiagonal matrix by multi-section on the Sturm count (64 shifts per
//      eigenvalue and round),
//   4. their eigenvectors by inverse iteration (tridiagonal LU with partial pivoting, one lane per vector) with Gram-Schmidt
//      inside clusters of close eigenvalues - any orthonormal basis of a repeated eigenvalue's invariant subspace is as
//      good as the reference's (dgl draws random signs on top) -
//   5. back-transformation through the reflectors, sign convention (entry of largest magnitude positive), fp32 output.
#ifndef SINGA_EMUL      // (workgroup-cooperative: not part of the sequential CPU emulation build of tests/emul)
// ---- the sparse route of lap_pe_kernel (see "1b." there): Chebyshev-filtered subspace iteration, block of 16 vectors
constexpr int LAP_FSI_PB = 16, LAP_FSI_ELLW = 32, LAP_FSI_DEG = 24, LAP_FSI_MAXIT = 40;
constexpr int LAP_FSI_PER_LD = 3 * LAP_FSI_PB + LAP_FSI_ELLW + LAP_FSI_ELLW / 2;     // doubles of global scratch per atom
constexpr int LAP_WORK_PER_LD = 3 + 4 * 9 + LAP_FSI_PER_LD;
template <int NW, int KV>
__device__ __forceinline__ bool lap_fsi(const double* __restrict__ A, int ld, int n, int sl, const int* perm, const int* cbeg,
                                        const int* cend, int* ell_n, int* flag, double* Xl, double* sc, double* wk, int m, int tid,
                                        int lane, int wave) {
    constexpr int PB = LAP_FSI_PB, W = LAP_FSI_ELLW;
    double* Yg = wk;                                    // [n][PB] the filter's previous block (global, L2 resident)
    double* Wg = Yg + (long long)ld * PB;               // [n][PB] L X
    double* Ng = Wg + (long long)ld * PB;               // [n][PB] the filter's next block, before it moves into LDS
    double* ell_v = Ng + (long long)ld * PB;            // [n][W] the Laplacian's rows: values ...
    int* ell_i = reinterpret_cast<int*>(ell_v + (long long)ld * W);     // ... and column POSITIONS
    double* H = sc;                                     // [16][16] projected matrix / Gram matrix
    double* Q = sc + 256;                               // [16][16] rotations / Cholesky factor
    double* part = sc + 512;                            // [512] reduction scratch
    double* theta = sc + 1024;                          // [16] Ritz values (ascending after `ritz`)
    double* resn = theta + 16;                          // [16] residual norms
    double* misc = resn + 16;                           // [8]
    const int total = n * PB;
    (void)sl;
    // ---- rows of the Laplacian in ELL form, from the component blocks phase 1 has built (entries outside them are undefined)
    if (tid == 0) flag[1] = 0;
    __syncthreads();
    for (int s0 = wave; s0 < n; s0 += NW) {
        const int lo = cbeg[s0], hi = cend[s0];
        const double* row = A + (long long)perm[s0] * ld;
        int cnt = 0;
        for (int t0 = lo; t0 <= hi; t0 += 64) {
            const int t = t0 + lane;
            const double v = t <= hi ? row[perm[t]] : 0.0;
            const bool nz = v != 0.0;
            const unsigned long long mask = __ballot(nz);
            const int at = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (nz && at < W) {
                ell_i[(long long)s0 * W + at] = t;
                ell_v[(long long)s0 * W + at] = v;
            }
            cnt += __popcll(mask);
        }
        if (lane == 0) {
            ell_n[s0] = cnt < W ? cnt : W;
            if (cnt > W) flag[1] = 1;
        }
    }
    __syncthreads();
    if (flag[1]) return false;
    // ---- start block: deterministic pseudo-random entries tied to the ATOM
    for (int idx = tid; idx < total; idx += 1024) {
        const int r = idx >> 4, j = idx & 15;
        unsigned h = (unsigned)(perm[r] * 2654435761u) ^ (unsigned)((j + 11) * 40503u * 2246822519u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        Xl[idx] = (double)(h & 0xFFFFFF) / 8388608.0 - 1.0;
    }
    __syncthreads();
    auto lx = [&](int idx) -> double {                  // (L X)[r][j] from the block in LDS
        const int r = idx >> 4, j = idx & 15;
        const int cnt = ell_n[r];
        const double* ev = ell_v + (long long)r * W;
        const int* ei = ell_i + (long long)r * W;
        double acc = 0.0;
        for (int e = 0; e < cnt; ++e) acc += ev[e] * Xl[ei[e] * PB + j];
        return acc;
    };
    // G = X^T B (B = X in LDS: Gram matrix; B = Wg: projected matrix), symmetrised, -> H
    auto gram = [&](const double* B) {
        if (tid < 512) {
            const int p = tid & 255, ch = tid >> 8, a = p >> 4, b = p & 15;
            double acc = 0.0;
            for (int r = ch; r < n; r += 2) acc += Xl[r * PB + a] * B[r * PB + b];
            part[ch * 256 + p] = acc;
        }
        __syncthreads();
        if (tid < 256) Q[tid] = part[tid] + part[256 + tid];
        __syncthreads();
        if (tid < 256) H[tid] = 0.5 * (Q[tid] + Q[(tid & 15) * 16 + (tid >> 4)]);
        __syncthreads();
    };
    // X <- X R^-1 with G = X^T X = R^T R (twice: CholQR2); false if a pivot vanishes
    auto orthonormalise = [&]() -> bool {
        for (int pass = 0; pass < 2; ++pass) {
            gram(Xl);
            if (tid == 0) {
                misc[0] = 0.0;
                double tr = 0.0;
                for (int j = 0; j < PB; ++j) tr += H[j * 16 + j];
                for (int j = 0; j < PB; ++j) {
                    double d = H[j * 16 + j];
                    for (int k = 0; k < j; ++k) d -= Q[k * 16 + j] * Q[k * 16 + j];
                    if (!(d > 1e-26 * tr)) { misc[0] = 1.0; d = 1.0; }
                    const double rj = sqrt(d);
                    Q[j * 16 + j] = rj;
                    for (int i = j + 1; i < PB; ++i) {
                        double v = H[j * 16 + i];
                        for (int k = 0; k < j; ++k) v -= Q[k * 16 + j] * Q[k * 16 + i];
                        Q[j * 16 + i] = v / rj;
                    }
                }
            }
            __syncthreads();
            if (misc[0] != 0.0) return false;
            if (tid < n) {                  // the thread's own row, solved in place (y R = x): no register arrays
                double* xr = Xl + tid * PB;
#pragma unroll 1
                for (int j = 0; j < PB; ++j) {
                    double v = xr[j];
#pragma unroll 1
                    for (int i = 0; i < j; ++i) v -= xr[i] * Q[i * 16 + j];
                    xr[j] = v / Q[j * 16 + j];
                }
            }
            __syncthreads();
        }
        return true;
    };
    // Rayleigh-Ritz: Wg = L X, H = X^T L X, Jacobi, X <- X Q, Wg <- Wg Q (Ritz values ascending), residual norms
    auto ritz = [&]() {
        for (int idx = tid; idx < total; idx += 1024) Wg[idx] = lx(idx);
        __syncthreads();
        gram(Wg);
        if (tid < 256) Q[tid] = (tid >> 4) == (tid & 15) ? 1.0 : 0.0;
        __syncthreads();
        // cyclic Jacobi with the round-robin ordering: 8 disjoint pairs per step, 8 lanes per pair (two rows each)
        for (int sweep = 0; sweep < 10; ++sweep) {
            if (tid == 0) {
                double off = 0.0, dg = 0.0;
                for (int i = 0; i < PB; ++i)
                    for (int j = 0; j < PB; ++j) (i == j ? dg : off) += H[i * 16 + j] * H[i * 16 + j];
                misc[1] = off <= 1e-30 * (dg + 1e-300) ? 1.0 : 0.0;
            }
            __syncthreads();
            const bool done = misc[1] != 0.0;
            __syncthreads();
            if (done) break;
            for (int st = 0; st < 15; ++st) {
                const int pr = tid >> 3, sub = tid & 7;
                const int p = pr == 0 ? 15 : (st + pr) % 15, q = pr == 0 ? st : (st + 15 - pr) % 15;
                double c = 1.0, sn = 0.0;
                if (tid < 64) {
                    const double hpp = H[p * 16 + p], hqq = H[q * 16 + q], hpq = H[p * 16 + q];
                    if (fabs(hpq) > 1e-300) {
                        const double tau = (hqq - hpp) / (2.0 * hpq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        sn = t * c;
                    }
                }
                __syncthreads();
                if (tid < 64) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int i = sub + 8 * h2;
                        const double aip = H[i * 16 + p], aiq = H[i * 16 + q];
                        H[i * 16 + p] = c * aip - sn * aiq;
                        H[i * 16 + q] = sn * aip + c * aiq;
                        const double qip = Q[i * 16 + p], qiq = Q[i * 16 + q];
                        Q[i * 16 + p] = c * qip - sn * qiq;
                        Q[i * 16 + q] = sn * qip + c * qiq;
                    }
                }
                __syncthreads();
                if (tid < 64) {
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int i = sub + 8 * h2;
                        const double api = H[p * 16 + i], aqi = H[q * 16 + i];
                        H[p * 16 + i] = c * api - sn * aqi;
                        H[q * 16 + i] = sn * api + c * aqi;
                    }
                }
                __syncthreads();
            }
        }
        // ascending order of the Ritz values; part[0..255] = Q with its columns in that order
        if (tid == 0) {
            int ord[PB];
            for (int j = 0; j < PB; ++j) ord[j] = j;
            for (int a = 1; a < PB; ++a) {
                const int oa = ord[a];
                int b = a;
                while (b > 0 && H[ord[b - 1] * 16 + ord[b - 1]] > H[oa * 16 + oa]) { ord[b] = ord[b - 1]; --b; }
                ord[b] = oa;
            }
            for (int j = 0; j < PB; ++j) {
                theta[j] = H[ord[j] * 16 + ord[j]];
                for (int i = 0; i < PB; ++i) part[i * 16 + j] = Q[i * 16 + ord[j]];
            }
        }
        __syncthreads();
        if (tid < n) {                      // one row of X, then the same row of L X (one at a time: 16 doubles of registers each)
            for (int which = 0; which < 2; ++which) {
                double* rowp = which == 0 ? Xl + tid * PB : Wg + tid * PB;
                double x[PB];
#pragma unroll
                for (int i = 0; i < PB; ++i) x[i] = rowp[i];
#pragma unroll 1
                for (int j = 0; j < PB; ++j) {
                    double xs = 0.0;
#pragma unroll
                    for (int i = 0; i < PB; ++i) xs += x[i] * part[i * 16 + j];
                    rowp[j] = xs;
                }
            }
        }
        __syncthreads();
        {   // residual norms |L x_j - theta_j x_j|
            const int j = tid & 15, ch = tid >> 4;
            const double th = theta[j];
            double acc = 0.0;
            for (int r = ch; r < n; r += 64) {
                const double d = Wg[r * PB + j] - th * Xl[r * PB + j];
                acc += d * d;
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            __syncthreads();                 // (part is read above by every row thread)
            if (lane < 16) part[wave * 16 + lane] = acc;
            __syncthreads();
            if (tid < PB) {
                double t = 0.0;
                for (int w2 = 0; w2 < NW; ++w2) t += part[w2 * 16 + tid];
                resn[tid] = sqrt(t);
            }
            __syncthreads();
        }
    };
    if (!orthonormalise()) return false;
    bool ok = false;
    for (int it = 0; it < LAP_FSI_MAXIT; ++it) {
        ritz();
        double worst = 0.0;
        for (int j = 0; j < m; ++j) worst = fmax(worst, resn[j]);
        if (worst <= 2e-10) { ok = true; break; }
        // Chebyshev filter of degree LAP_FSI_DEG: damp [a, 2], scaled at a0 (Zhou & Saad's recurrence)
        const double a0 = theta[0], b = 2.0 + 1e-9;
        const double a = fmax(theta[PB - 1], a0 + 1e-4);
        if (!(a < b - 1e-3)) break;
        const double e = 0.5 * (b - a), c = 0.5 * (b + a);
        double sigma = e / (a0 - c);
        const double sigma1 = sigma;
        // (every step: next block -> global, barrier, then previous <- current, current <- next: a thread moves the elements it
        // computed, so the only cross-thread hazard is the gather from the current block in LDS)
        for (int idx = tid; idx < total; idx += 1024) Ng[idx] = (lx(idx) - c * Xl[idx]) * (sigma1 / e);
        __syncthreads();
        for (int idx = tid; idx < total; idx += 1024) { Yg[idx] = Xl[idx]; Xl[idx] = Ng[idx]; }
        __syncthreads();
        for (int i = 2; i <= LAP_FSI_DEG; ++i) {
            const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
            const double f1 = 2.0 * sigma2 / e, f2 = sigma * sigma2;
            for (int idx = tid; idx < total; idx += 1024) Ng[idx] = (lx(idx) - c * Xl[idx]) * f1 - f2 * Yg[idx];
            __syncthreads();
            for (int idx = tid; idx < total; idx += 1024) { Yg[idx] = Xl[idx]; Xl[idx] = Ng[idx]; }
            __syncthreads();
            sigma = sigma2;
        }
        if (!orthonormalise()) return false;
    }
    if (!ok) return false;
    // ---- the m lowest pairs -> Z[k][position] (the same LDS region: staged through global memory)
    for (int idx = tid; idx < total; idx += 1024) Yg[idx] = Xl[idx];
    __syncthreads();
    for (int r = tid; r < m * sl; r += 1024) {
        const int k = r / sl, i = r - k * sl;
        Xl[r] = i < n ? Yg[i * PB + k] : 0.0;
    }
    __syncthreads();
    return true;
}

template <int NQ>
__global__ void __launch_bounds__(1024) lap_pe_kernel(double* __restrict__ Aall, const int* __restrict__ esrc, const int* __restrict__ edst,
                                                      const int* __restrict__ eptr, const int* __restrict__ nnodes,
                                                      const int* __restrict__ first, double* __restrict__ work, float* __restrict__ out,
                                                      int ld, int kout, int fsi_min) {
    constexpr int NW = 16, KV = 9;                                 // wavefronts per workgroup; vectors computed (k + 1 <= 9)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nnodes[g];
    if (n <= 0) return;
    double* A = Aall + (long long)g * ld * ld;
    const int sl = ld < 32 ? 32 : ld;     // stride of the LDS arrays (the Sturm counts need 16 * sl >= 306 doubles)
    double* va = sm;                      // [sl] reflector of the current step
    double* wa = sm + sl;                 // [sl]
    double* vb = sm + 2 * sl;             // [sl] reflector of the next step
    double* wb = sm + 3 * sl;             // [sl]
    double* pw = sm + 4 * sl;             // [NW][sl] per-wavefront partial products; later the vectors Z [KV][sl]
    double* red = sm + (4 + NW) * sl;     // [NW + 8] reduction scratch / broadcast scalars
    int* label = reinterpret_cast<int*>(red + NW + 8);             // [sl] component label (smallest atom index) of an ATOM
    int* perm = label + sl;                                        // [sl] position -> atom
    int* cend = perm + sl;                                         // [sl] last position of the component of a POSITION
    int* cbeg = cend + sl;                                         // [sl] first position of that component
    int* posof = cbeg + sl;                                        // [sl] atom -> position
    int* flag = posof + sl;                                        // [2]
    double* dd = work + (long long)g * LAP_WORK_PER_LD * ld;       // diagonal
    double* work_fsi = dd + (long long)(3 + 4 * KV) * ld;          // [LAP_FSI_PER_LD * ld] scratch of the sparse route
    double* ee = dd + ld;                                          // off-diagonal
    double* bb = ee + ld;                                          // reflector factors
    double* lu = bb + ld;                                          // [KV][4][ld] scratch of the tridiagonal solves
    const int e0 = eptr[g], e1 = eptr[g + 1];

    auto block_sum = [&](double x) -> double {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[w];
        return s;
    };

    // ---- 0. connected components
    if (tid < n) label[tid] = tid;
    __syncthreads();
    for (int round = 0; round < n + 2; ++round) {
        if (tid == 0) flag[0] = 0;
        __syncthreads();
        for (int e = e0 + tid; e < e1; e += 1024) {
            const int a = esrc[e], b = edst[e];
            const int la = label[a], lb = label[b];
            if (la < lb) { atomicMin(&label[b], la); flag[0] = 1; }
            else if (lb < la) { atomicMin(&label[a], lb); flag[0] = 1; }
        }
        __syncthreads();
        if (tid < n) {                                             // pointer jumping: labels are atom indices
            int l = label[tid];
            for (int h = 0; h < 4; ++h) {
                const int l2 = label[l];
                if (l2 >= l) break;
                l = l2;
            }
            if (l < label[tid]) { atomicMin(&label[tid], l); flag[0] = 1; }
        }
        __syncthreads();
        const int changed = flag[0];
        __syncthreads();
        if (!changed) break;
    }
    // positions: atoms ordered by (label, index)
    if (tid < n) {
        const int li = label[tid];
        int before = 0, same_before = 0, same_after = 0;
        for (int j = 0; j < n; ++j) {
            const int lj = label[j];
            before += lj < li;
            same_before += (lj == li) & (j < tid);
            same_after += (lj == li) & (j > tid);
        }
        const int pos = before + same_before;
        perm[pos] = tid;
        posof[tid] = pos;
        cend[pos] = pos + same_after;
        cbeg[pos] = before;
        bb[pos] = 0.0;
        ee[pos] = 0.0;
    }
    __syncthreads();
    // ---- 1. the Laplacian's component blocks
    for (int s0 = wave; s0 < n; s0 += NW) {                        // zero: wavefront per row position, lanes over the block
        const int hi = cend[s0];
        double* row = A + (long long)perm[s0] * ld;
        for (int t = cbeg[s0] + lane; t <= hi; t += 64) row[perm[t]] = 0.0;
    }
    __syncthreads();
    for (int e = e0 + tid; e < e1; e += 1024) A[(long long)esrc[e] * ld + edst[e]] = 1.0;      // raw adjacency (repeats count once)
    __syncthreads();
    double* dinv = wa;                                             // [atom]
    if (tid < n) {                                                 // in-degree of atom tid = its raw column over the component's rows
        const int pos = posof[tid], hi = cend[pos];
        double deg = 0.0;
        for (int s0 = cbeg[pos]; s0 <= hi; ++s0) deg += A[(long long)perm[s0] * ld + tid];
        dinv[tid] = 1.0 / sqrt(deg < 1.0 ? 1.0 : deg);
    }
    __syncthreads();
    for (int s0 = wave; s0 < n; s0 += NW) {                        // scaling + symmetrisation: the pair (i, j), i < j, by one lane
        const int hi = cend[s0];
        const int i = perm[s0];
        const double di = dinv[i];
        for (int t = s0 + lane; t <= hi; t += 64) {
            const int j = perm[t];
            if (t == s0) {
                A[(long long)i * ld + i] = 1.0 - di * di * A[(long long)i * ld + i];
            } else {
                const double v = -0.5 * di * dinv[j] * (A[(long long)i * ld + j] + A[(long long)j * ld + i]);
                A[(long long)i * ld + j] = v;
                A[(long long)j * ld + i] = v;
            }
        }
    }
    __syncthreads();

    const int m = n < KV ? n : KV;          // eigenpairs computed (the smallest + up to kout more)
    double* Z = pw;                         // [KV][sl] the vectors, by POSITION (both routes leave them here)

    // ---- 1b. graphs with a LARGE component: Chebyshev-filtered subspace iteration on the SPARSE Laplacian instead of the dense
    // O(n^3) tridiagonalisation below (a bonded molecular graph has ~10 neighbours per atom: one product with the Laplacian is
    // ~10 n multiply-adds instead of n^2).  Whole graph at once (all components); a block of PB = 16 vectors, so eigenvalues of
    // multiplicity up to 16 - the zero eigenvalue of a many-component graph, symmetric fragments - come out as a basis of their
    // subspace, as from the dense route.  Per outer iteration: Rayleigh-Ritz on the block (16 x 16 Jacobi), residuals of the
    // lowest m pairs, then a degree-FSI_DEG Chebyshev polynomial of L that damps [theta_15, 2] and amplifies what lies below,
    // then CholQR2.  Anything that does not go by the book - a row with more than ELLW neighbours, a Cholesky pivot that
    // vanishes, no convergence in FSI_MAXIT iterations - returns false and the dense route runs: never a wrong answer, at worst
    // a slow one.
    bool fsi_ok = false;
    {
        int big = tid < n ? cend[tid] - cbeg[tid] + 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) big = max(big, __shfl_xor(big, o, 64));
        __syncthreads();
        if (tid == 0) flag[0] = 0;
        __syncthreads();
        if (lane == 0) atomicMax(&flag[0], big);
        __syncthreads();
        big = flag[0];
        __syncthreads();
        // (the iteration's small matrices live in the first 4 sl doubles of LDS: 1,064 are needed)
        if (big >= fsi_min && n > 2 * 16 && 4 * sl >= 1064) fsi_ok = lap_fsi<NW, KV>(A, ld, n, sl, perm, cbeg, cend, label, flag, pw, sm, work_fsi, m, tid, lane, wave);
        __syncthreads();
    }
    if (!fsi_ok) {
    // ---- 2. tridiagonalisation, component by component (positions cs .. ce; nend = ce + 1)
    // reflector from row `row` (positions row + 1 .. nend - 1) -> vec[] (other entries of the component zero)
    auto reflector = [&](int row, int nend, double* vec) {
        double x = 0.0, sq = 0.0;
        const long long prow = (long long)perm[row] * ld;
        if (tid > row && tid < nend) {
            x = A[prow + perm[tid]];
            if (tid > row + 1) sq = x * x;
        }
        const double sigma = block_sum(sq);
        if (tid == row + 1) red[NW] = x;
        __syncthreads();
        const double alpha = red[NW];
        double b = 0.0, mu = alpha, v0 = 1.0;
        if (sigma != 0.0) {
            mu = sqrt(alpha * alpha + sigma);
            v0 = alpha <= 0.0 ? alpha - mu : -sigma / (alpha + mu);
            b = 2.0 * v0 * v0 / (sigma + v0 * v0);
        }
        if (tid >= row && tid < nend) {
            double v = 0.0;
            if (tid == row + 1) v = 1.0;
            else if (tid > row + 1) v = sigma != 0.0 ? x / v0 : 0.0;
            vec[tid] = v;
            if (tid > row) A[prow + perm[tid]] = v;                // the reflector replaces the row it came from
        }
        if (tid == 0) {
            dd[row] = A[prow + perm[row]];
            ee[row] = mu;
            bb[row] = b;
            red[NW + 1] = b;
        }
        __syncthreads();
        return red[NW + 1];
    };
    // one pass over the trailing block positions lo .. nend - 1: a -= vu[c] wu[r] + wu[c] vu[r] (if upd), then
    // p[r] += a vn[c] (if acc); the wave partials of p end up in pw
    auto pass = [&](int lo, int nend, bool upd, const double* vu, const double* wu, bool acc, const double* vn) {
        constexpr int NR = NQ <= 8 ? NQ : 1;
        double pacc[NQ], vr[NR], wr[NR];
        int pr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = lo + lane + 64 * q;
            pacc[q] = 0.0;
            pr[q] = r < nend ? perm[r] : 0;
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = lo + lane + 64 * q;
            vr[q] = (upd && r < nend) ? vu[r] : 0.0;
            wr[q] = (upd && r < nend) ? wu[r] : 0.0;
        }
        for (int c = lo + wave; c < nend; c += NW) {
            const double vc = upd ? vu[c] : 0.0, wc = upd ? wu[c] : 0.0, nc = acc ? vn[c] : 0.0;
            double* row = A + (long long)perm[c] * ld;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = lo + lane + 64 * q;
                if (r < nend) {
                    double a = row[pr[q]];
                    if (upd) {
                        if constexpr (NQ <= 8) a -= vc * wr[q] + wc * vr[q];
                        else a -= vc * wu[r] + wc * vu[r];
                        row[pr[q]] = a;
                    }
                    pacc[q] += a * nc;
                }
            }
        }
        if (acc) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = lo + lane + 64 * q;
                if (r < nend) pw[wave * sl + r] = pacc[q];
            }
        }
        __syncthreads();
    };
    // w = b p - (b / 2) (b p . v) v with p = sum of the wave partials (positions lo .. nend - 1)
    auto finish_w = [&](int lo, int nend, double b, const double* vec, double* wout) {
        double p = 0.0, pv = 0.0;
        if (tid >= lo && tid < nend) {
#pragma unroll
            for (int w = 0; w < NW; ++w) p += pw[w * sl + tid];
            p *= b;
            pv = p * vec[tid];
        }
        const double gam = 0.5 * b * block_sum(pv);
        if (tid >= lo && tid < nend) wout[tid] = p - gam * vec[tid];
        __syncthreads();
    };
    for (int cs = 0; cs < n;) {
        const int ce = cend[cs], nend = ce + 1, m = nend - cs;
        if (m >= 3) {
            double b = reflector(cs, nend, vb);
            pass(cs + 1, nend, false, nullptr, nullptr, true, vb);
            finish_w(cs + 1, nend, b, vb, wb);
            for (int j = cs; j + 2 < nend; ++j) {
                double* t = va; va = vb; vb = t;
                t = wa; wa = wb; wb = t;
                const int row = j + 1;
                // row j + 1 gets its update first: it defines the next reflector
                if (tid >= row && tid < nend) A[(long long)perm[row] * ld + perm[tid]] -= va[row] * wa[tid] + wa[row] * va[tid];
                __syncthreads();
                const bool more = row + 2 < nend;
                double bn = 0.0;
                if (more) bn = reflector(row, nend, vb);
                pass(row + 1, nend, true, va, wa, more, vb);
                if (more) finish_w(row + 1, nend, bn, vb, wb);
            }
        }
        if (tid == 0) {
            if (m >= 2) {
                dd[nend - 2] = A[(long long)perm[nend - 2] * ld + perm[nend - 2]];
                ee[nend - 2] = A[(long long)perm[nend - 2] * ld + perm[nend - 1]];
                bb[nend - 2] = 0.0;
            }
            dd[nend - 1] = A[(long long)perm[nend - 1] * ld + perm[nend - 1]];
            ee[nend - 1] = 0.0;                                    // no coupling across the component boundary
            bb[nend - 1] = 0.0;
        }
        __syncthreads();
        cs = nend;
    }
    // ---- 3. eigenvalues 0 .. m - 1 of the tridiagonal matrix (d, e): multi-section on the Sturm count
    double* td = va;                       // diagonal and squared off-diagonal in LDS
    double* te2 = wa;
    double* lam = wb;                      // [KV]
    double gl = 0.0, gu = 0.0;
    if (tid < n) {
        const double d = dd[tid], el = tid > 0 ? fabs(ee[tid - 1]) : 0.0, er = tid + 1 < n ? fabs(ee[tid]) : 0.0;
        td[tid] = d;
        te2[tid] = tid + 1 < n ? ee[tid] * ee[tid] : 0.0;
        gl = d - el - er;
        gu = d + el + er;
    }
    {   // Gershgorin bounds
        double lo = tid < n ? gl : 1e300, hi = tid < n ? gu : -1e300;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo = fmin(lo, __shfl_xor(lo, o, 64));
            hi = fmax(hi, __shfl_xor(hi, o, 64));
        }
        __syncthreads();
        if (lane == 0) pw[wave] = lo, pw[NW + wave] = hi;
        __syncthreads();
        lo = pw[0], hi = pw[NW];
        for (int w = 1; w < NW; ++w) lo = fmin(lo, pw[w]), hi = fmax(hi, pw[NW + w]);
        __syncthreads();
        const double span = hi - lo + 1e-300;
        if (tid < KV) pw[tid] = lo - 1e-12 * span - 1e-300, pw[KV + tid] = hi + 1e-12 * span + 1e-300;   // [lo_k | hi_k]
        __syncthreads();
    }
    {
        constexpr int P = 64;
        const int ke = tid / P, pi = tid % P;              // eigenvalue index and shift index of this thread
        int* cnt = reinterpret_cast<int*>(pw + 2 * KV);    // [KV][P] Sturm counts
        for (int round = 0; round < 10; ++round) {
            double x = 0.0;
            if (ke < m) {
                const double lo = pw[ke], hi = pw[KV + ke];
                x = lo + (hi - lo) * (double)(pi + 1) / (double)(P + 1);
                int c = 0;
                double q = 1.0;
                for (int i = 0; i < n; ++i) {
                    q = td[i] - x - (i > 0 ? te2[i - 1] / q : 0.0);
                    if (fabs(q) < 1e-300) q = -1e-300;
                    c += q < 0.0;
                }
                cnt[ke * P + pi] = c;
            }
            __syncthreads();
            double nlo = 0.0, nhi = 0.0;
            bool setlo = false, sethi = false;
            if (ke < m) {
                // eigenvalue ke lies in (x_{i-1}, x_i] for the first shift i with count > ke
                const int c = cnt[ke * P + pi];
                const int cp = pi > 0 ? cnt[ke * P + pi - 1] : -1;
                const int cn = pi + 1 < P ? cnt[ke * P + pi + 1] : 1 << 30;
                if (c > ke && cp <= ke) sethi = true, nhi = x;
                if (c <= ke && cn > ke) setlo = true, nlo = x;
            }
            __syncthreads();
            if (sethi) pw[KV + ke] = nhi;
            if (setlo) pw[ke] = nlo;
            __syncthreads();
        }
        if (tid < m) lam[tid] = 0.5 * (pw[tid] + pw[KV + tid]);
        __syncthreads();
    }
    // ---- 4. inverse iteration, one lane per vector; Z[k] in LDS (pw region), indexed by position
    double tnorm = 0.0;
    {
        double t = tid < n ? fmax(fabs(td[tid]), sqrt(te2[tid])) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t = fmax(t, __shfl_xor(t, o, 64));
        __syncthreads();
        if (lane == 0) vb[wave] = t;
        __syncthreads();
        for (int w = 0; w < NW; ++w) tnorm = fmax(tnorm, vb[w]);
        __syncthreads();
    }
    const double eps = 2.220446049250313e-16, tiny = eps * (tnorm > 0.0 ? tnorm : 1.0);
    // (Z = pw: the eigenvalue intervals / counts kept there are dead now)
    if (tid < m) vb[tid] = lam[tid];         // shifts; separated inside clusters below (LAPACK dstein does the same)
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < m; ++k) {
            const double sep = 10.0 * eps * fmax(fabs(vb[k]), tnorm);
            if (vb[k] - vb[k - 1] < sep) vb[k] = vb[k - 1] + sep;
        }
    }
    __syncthreads();
    for (int r = tid; r < m * sl; r += 1024) {
        const int k = r / sl, i = r - k * sl;
        // deterministic start vectors in (-1, 1), tied to the ATOM (not to its position)
        const int at = i < n ? perm[i] : 0;
        unsigned h = (unsigned)(at * 2654435761u) ^ (unsigned)((k + 1) * 40503u * 2246822519u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        Z[r] = i < n ? ((double)(h & 0xFFFFFF) / 8388608.0 - 1.0) : 0.0;
    }
    __syncthreads();
    for (int it = 0; it < 5; ++it) {
        if (tid < m) {
            const int k = tid;
            const double shift = vb[k];
            double* dl = lu + (long long)(k * 4 + 0) * ld;      // sub-diagonal
            double* dg = lu + (long long)(k * 4 + 1) * ld;      // diagonal
            double* du = lu + (long long)(k * 4 + 2) * ld;      // first super-diagonal
            double* d2 = lu + (long long)(k * 4 + 3) * ld;      // second super-diagonal (pivoting fill-in)
            double* x = Z + k * sl;
            for (int i = 0; i < n; ++i) {
                dg[i] = td[i] - shift;
                const double e = i + 1 < n ? ee[i] : 0.0;
                dl[i] = e, du[i] = e, d2[i] = 0.0;
            }
            for (int i = 0; i + 1 < n; ++i) {
                if (fabs(dg[i]) >= fabs(dl[i])) {
                    if (dg[i] == 0.0) dg[i] = tiny;
                    const double f = dl[i] / dg[i];
                    dg[i + 1] -= f * du[i];
                    x[i + 1] -= f * x[i];
                    if (i + 2 < n) d2[i] = 0.0;
                } else {
                    const double f = dg[i] / dl[i];
                    dg[i] = dl[i];
                    const double t = dg[i + 1];
                    dg[i + 1] = du[i] - f * t;
                    if (i + 2 < n) {
                        d2[i] = du[i + 1];
                        du[i + 1] = -f * du[i + 1];
                    }
                    du[i] = t;
                    const double xi = x[i];
                    x[i] = x[i + 1];
                    x[i + 1] = xi - f * x[i + 1];
                }
            }
            if (dg[n - 1] == 0.0) dg[n - 1] = tiny;
            x[n - 1] /= dg[n - 1];
            if (n > 1) x[n - 2] = (x[n - 2] - du[n - 2] * x[n - 1]) / dg[n - 2];
            for (int i = n - 3; i >= 0; --i) x[i] = (x[i] - du[i] * x[i + 1] - d2[i] * x[i + 2]) / dg[i];
            // scale against overflow before the dot products
            double mx = 0.0;
            for (int i = 0; i < n; ++i) mx = fmax(mx, fabs(x[i]));
            const double sc = mx > 0.0 ? 1.0 / mx : 1.0;
            for (int i = 0; i < n; ++i) x[i] *= sc;
        }
        __syncthreads();
        // Gram-Schmidt inside clusters (in eigenvalue order) + normalisation, all threads
        for (int k = 0; k < m; ++k) {
            double* x = Z + k * sl;
            for (int j = 0; j < k; ++j) {
                if (fabs(lam[k] - lam[j]) < 1e-3 * (tnorm > 0.0 ? tnorm : 1.0)) {
                    const double* y = Z + j * sl;
                    const double dot = block_sum(tid < n ? x[tid] * y[tid] : 0.0);
                    if (tid < n) x[tid] -= dot * y[tid];
                    __syncthreads();
                }
            }
            const double nn = block_sum(tid < n ? x[tid] * x[tid] : 0.0);
            if (tid < n) x[tid] *= nn > 0.0 ? rsqrt(nn) : 0.0;
            __syncthreads();
        }
    }
    // ---- 5. back-transformation x = H_0 H_1 .. y: one wavefront per vector; reflector j lives in row perm[j], positions
    // j + 1 .. cend[j] of its component
    if (wave < m) {
        double* x = Z + wave * sl;
        for (int j = n - 3; j >= 0; --j) {
            const double b = bb[j];
            if (b == 0.0) continue;
            const double* v = A + (long long)perm[j] * ld;
            const int hi = cend[j];
            double s = 0.0;
            for (int r = j + 1 + lane; r <= hi; r += 64) s += v[perm[r]] * x[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            s *= b;
            for (int r = j + 1 + lane; r <= hi; r += 64) x[r] -= s * v[perm[r]];
        }
    }
    }   // (dense route)
    __syncthreads();
    // sign convention + output: columns 1 .. kout of the spectrum (the smallest eigenvalue's vector is dropped)
    float* o = out + (long long)first[g] * kout;
    for (int k = 1; k <= kout; ++k) {
        if (k >= m) {
            if (tid < n) o[(long long)tid * kout + k - 1] = 0.f;
            continue;
        }
        const double* x = Z + k * sl;
        // entry of largest magnitude (the lowest atom index on ties)
        double best = tid < n ? fabs(x[tid]) : -1.0;
        int bi = tid < n ? perm[tid] : (1 << 30), bp = tid;
#pragma unroll
        for (int oo = 32; oo > 0; oo >>= 1) {
            const double ob = __shfl_xor(best, oo, 64);
            const int oi = __shfl_xor(bi, oo, 64), op = __shfl_xor(bp, oo, 64);
            if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi, bp = op;
        }
        __syncthreads();
        int* redi = reinterpret_cast<int*>(red + NW);
        if (lane == 0) red[wave] = best, redi[wave] = bi, redi[NW + wave] = bp;
        __syncthreads();
        best = red[0], bi = redi[0], bp = redi[NW];
        for (int w = 1; w < NW; ++w) {
            const double ob = red[w];
            const int oi = redi[w];
            if (ob > best || (ob == best && oi < bi)) best = ob, bi = oi, bp = redi[NW + w];
        }
        const double sg = x[bp] < 0.0 ? -1.0 : 1.0;
        if (tid < n) o[(long long)perm[tid] * kout + k - 1] = (float)(sg * x[tid]);
        __syncthreads();
    }
}
#endif

// ------------------------------------------------------------------------------------------------ host helpers
bool attn_pitch(int token_major, int heads, long long lq, long long lk, long long lv, const float* q, const float* k, const float* v,
                AttnPitch* out) {
    out->q = (int)(lq > 0 ? lq : heads * 32);
    out->k = (int)(lk > 0 ? lk : heads * 32);
    out->v = (int)(lv > 0 ? lv : heads * 64);
    if (!token_major) return lq <= 0 && lk <= 0 && lv <= 0;
    if (lq > (1 << 30) || lk > (1 << 30) || lv > (1 << 30)) return false;
    if (out->q % 4 || out->k % 4 || out->v % 4 || out->q < heads * 32 || out->k < heads * 32 || out->v < heads * 64) return false;
    return !(((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15));
}

int grid_for(long long work, int cap = 256 * 32) {
    long long g = work < 1 ? 1 : work;
    return (int)(g > cap ? cap : g);
}
// one workgroup per node, rounded up to a multiple of 8 (SINGA_XCD_NODE_LOOP)
int kn_grid(long long N) {
    long long g = (N < 1 ? 1 : N);
    g = (g + 7) / 8 * 8;
    return (int)(g > (1 << 20) ? (1 << 20) : g);
}

bool pack(const singa_seg_t* s, int nseg, Segs* out) {
    if (!s || nseg < 1 || nseg > 3) return false;
    for (int i = 0; i < 3; ++i) {
        out->p[i] = i < nseg ? s[i].ptr : nullptr;
        out->ld[i] = i < nseg ? s[i].ld : 0;
        out->rows[i] = i < nseg ? s[i].rows : 0;
        if (i < nseg && !s[i].ptr) return false;
    }
    return true;
}

bool pack_mut(const singa_seg_mut_t* s, int nseg, SegsMut* out) {
    if (!s || nseg < 1 || nseg > 3) return false;
    for (int i = 0; i < 3; ++i) {
        out->p[i] = i < nseg ? s[i].ptr : nullptr;
        out->ld[i] = i < nseg ? s[i].ld : 0;
        out->rows[i] = i < nseg ? s[i].rows : 0;
        if (i < nseg && !s[i].ptr) return false;
    }
    return true;
}

#define SINGA_DISPATCH_L(lmax, mmax, ...)                                             \
    do {                                                                                \
        if ((mmax) != 2) return fail(SINGA_E_LMAX, "only mmax = 2 is built");           \
        switch (lmax) {                                                                 \
            case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;                              \
            case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;                              \
            case 6: { constexpr int L_ = 6; __VA_ARGS__; } break;                              \
            default: return fail(SINGA_E_LMAX, "lmax must be 2, 4 or 6");               \
        }                                                                               \
    } while (0)

// ------------------------------------------------------------------------------------------------ n1: kNN graph
// torch_cluster.knn_graph(pos, k, batch, flow='target_to_source') (reference model/CProMG.py:293,330): for every atom its k
// nearest other atoms of the same molecule.  One wavefront per centre atom; the molecule's atoms (a contiguous index range
// [ptr[g], ptr[g+1]) of the collated batch) are spread over the lanes, MAXC candidates per lane in registers.  Squared distances
// are formed from exact fp32 coordinate differences, (dx*dx + dy*dy) + dz*dz without contraction, as torch_cluster's kernels
// do - not through |a|^2 + |b|^2 - 2ab, which loses ~1e-4 A^2 at |a|^2 ~ 10^3 and flips near-ties of the k-th neighbour.
// Selection: k rounds of a wave-wide minimum over 64-bit keys (distance bits << 32 | atom index: a non-negative float's bit
// pattern orders like the float, ties go to the lower index); the winner's slot is retired.  Slots that do not exist (molecules
// with fewer than k + 1 atoms, atoms of no molecule: the inert padding of graph.pad_batch) hold -1.  Output: row = centre,
// col = neighbour, k consecutive entries per atom in order of increasing distance.
template <int MAXC>
__global__ __launch_bounds__(256) void knn_graph_kernel(const float* __restrict__ pos, const int* __restrict__ batch,
                                                        const long long* __restrict__ ptr, int B, int N, int k,
                                                        long long* __restrict__ row, long long* __restrict__ col) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= N) return;
    const int g = batch[i];
    long long* ro = row + (long long)i * k;
    long long* co = col + (long long)i * k;
    if (g < 0 || g >= B) {
        for (int r = lane; r < k; r += 64) ro[r] = co[r] = -1;
        return;
    }
    const int lo = (int)ptr[g], hi = (int)ptr[g + 1];
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    const unsigned long long NONE = ~0ull;
    unsigned long long key[MAXC];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
        const int c = lo + lane + 64 * t;
        key[t] = NONE;
        if (c < hi && c != i) {
            const float dx = __fsub_rn(px, pos[3 * c]), dy = __fsub_rn(py, pos[3 * c + 1]), dz = __fsub_rn(pz, pos[3 * c + 2]);
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            key[t] = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)c;
        }
    }
    for (int r = 0; r < k; ++r) {
        unsigned long long best = key[0];
#pragma unroll
        for (int t = 1; t < MAXC; ++t) best = key[t] < best ? key[t] : best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned blo = (unsigned)__shfl_xor((int)(unsigned)best, o, 64);
            const unsigned bhi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)bhi << 32) | blo;
            best = other < best ? other : best;
        }
        // the winner's slot is retired on the lane that owns it (keys are unique: the atom index is part of the key)
#pragma unroll
        for (int t = 0; t < MAXC; ++t)
            if (key[t] == best) key[t] = NONE;
        if (lane == 0) {
            const bool have = best != NONE;
            ro[r] = have ? (long long)i : -1;
            co[r] = have ? (long long)(unsigned)best : -1;
        }
    }
}

// Calibration kernel for the PMC byte counters (MI355X_MICROARCH.md §HBM: FETCH_SIZE is only calibrated for 16-B
// lanes): copies n floats with the access shape of the segment kernels (one dword per lane, 256 B per wave-instruction)
// so that a known byte count can be compared with FETCH_SIZE / WRITE_SIZE.
__global__ void calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// The same with 16 bytes per lane (the guide's float4 copy): the practical HBM ceiling next to the 8 TB/s spec peak.
template <int UNROLL>
__global__ void __launch_bounds__(256) calib_copy16_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    // a workgroup moves contiguous chunks of 256 * UNROLL float4s: UNROLL independent 16-byte loads in flight per lane, every
    // wave instruction a contiguous 1 KB
    const long long chunk = 256LL * UNROLL;
    for (long long base = (long long)blockIdx.x * chunk; base < n4; base += (long long)gridDim.x * chunk) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long i = base + threadIdx.x + 256LL * u;
            if (i < n4) v[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long long i = base + threadIdx.x + 256LL * u;
            if (i < n4) dst[i] = v[u];
        }
    }
}

}  // namespace

// ================================================================================================= C ABI
extern "C" {

int singa_prof_enable(int on) {
    g_prof_on = on != 0;
    return SINGA_OK;
}

// Graph mode on (buf != NULL: a device buffer of cap 64-bit words owned by the caller, 2 per tagged launch + 2) / off (NULL).
// While on, every tagged launch is bracketed by stamp kernels (see g_stamp_buf); the record table restarts.
int singa_prof_stamps(unsigned long long* buf, int cap) {
    g_stamp_buf = buf;
    g_stamp_cap = buf ? cap : 0;
    if (buf) {
        g_prof_on = false;
        g_prof_n = 0;
    }
    return SINGA_OK;
}

// The records of the stamped launches from a HOST copy of the stamp buffer (after a replay + synchronize): milliseconds between
// the two stamps minus the calibration pair's, tag, E, N.  Does not change the table; singa_prof_reset forgets it.
int singa_prof_read_stamps(const unsigned long long* host_stamps, float* ms, int* tags, int* edges, int* nodes, int cap) {
    if (!host_stamps) return 0;
    int n = 0;
    double calib = (double)(host_stamps[1] - host_stamps[0]);
    for (int i = 1; i < g_prof_n && n < cap; ++i) {
        double ticks = (double)(host_stamps[2 * i + 1] - host_stamps[2 * i]) - calib;      // 100 MHz: 10 ns per tick
        ms[n] = (float)(ticks * 1e-5);
        tags[n] = g_prof[i].tag;
        edges[n] = g_prof[i].E;
        nodes[n] = g_prof[i].N;
        ++n;
    }
    return n;
}

int singa_prof_reset(void) {
    g_prof_n = 0;
    return SINGA_OK;
}

int singa_prof_hint_edges(int E) {
    g_prof_edges_hint = E;
    return SINGA_OK;
}

int singa_prof_collect(float* ms, int* edges, int* nodes, int cap) {
    int n = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        float t = 0.f;
        hipError_t e = hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b);
        if (e == hipSuccess && n < cap) {
            ms[n] = t;
            edges[n] = g_prof[i].E;
            nodes[n] = g_prof[i].N;
            ++n;
        }
        (void)hipEventDestroy(g_prof[i].a);
        (void)hipEventDestroy(g_prof[i].b);
    }
    g_prof_n = 0;
    return n;
}

int singa_prof_collect_tagged(float* ms, int* tags, int* edges, int* nodes, int cap) {
    int n = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        float t = 0.f;
        hipError_t e = hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b);
        if (e == hipSuccess && n < cap) {
            ms[n] = t;
            tags[n] = g_prof[i].tag;
            edges[n] = g_prof[i].E;
            nodes[n] = g_prof[i].N;
            ++n;
        }
        (void)hipEventDestroy(g_prof[i].a);
        (void)hipEventDestroy(g_prof[i].b);
    }
    g_prof_n = 0;
    return n;
}

int singa_calib_copy(const float* src, float* dst, long long n, void* stream) {
    if (!src || !dst) return fail(SINGA_E_NULL, "calib_copy: null pointer");
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return check_launch("calib_copy");
}

int singa_knn_graph(const float* pos, const int32_t* batch, const long long* ptr, int B, int N, int k, int max_nodes, long long* row,
                    long long* col, void* stream) {
    if (!pos || !batch || !ptr || !row || !col) return fail(SINGA_E_NULL, "knn_graph: null pointer");
    if (B < 0 || N < 0 || k <= 0 || max_nodes < 0 || max_nodes > 2048) return fail(SINGA_E_SHAPE, "knn_graph: k > 0 and at most 2048 atoms per molecule");
    if (N == 0) return SINGA_OK;
    const dim3 grid((unsigned)((N + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (max_nodes <= 512) hipLaunchKernelGGL(knn_graph_kernel<8>, grid, block, 0, st, pos, batch, ptr, B, N, k, row, col);
    else if (max_nodes <= 1024) hipLaunchKernelGGL(knn_graph_kernel<16>, grid, block, 0, st, pos, batch, ptr, B, N, k, row, col);
    else hipLaunchKernelGGL(knn_graph_kernel<32>, grid, block, 0, st, pos, batch, ptr, B, N, k, row, col);
    return check_launch("knn_graph");
}

int singa_knn_edge_attr(const float* len, const int32_t* ptr, long long n_real, const float* offset, float coeff, float* out, int N,
                        int G, void* stream) {
    if (!ptr || !offset || !out || (!len && n_real > 0)) return fail(SINGA_E_NULL, "knn_edge_attr: null pointer");
    if (G != 64) return fail(SINGA_E_SHAPE, "knn_edge_attr: 64 Gaussians (the shipped edge_channels)");
    if (N < 0 || n_real < 0) return fail(SINGA_E_SHAPE, "knn_edge_attr: negative size");
    if (N == 0) return SINGA_OK;
    hipLaunchKernelGGL(knn_edge_attr_kernel, dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, len, ptr, n_real, offset, coeff, out, N);
    return check_launch("knn_edge_attr");
}

int singa_calib_copy16(const float* src, float* dst, long long n, int blocks, int unroll, void* stream) {
    if (!src || !dst) return fail(SINGA_E_NULL, "calib_copy16: null pointer");
    if (n % 4 || ((uintptr_t)src | (uintptr_t)dst) % 16) return fail(SINGA_E_SHAPE, "calib_copy16: n % 4 == 0 and 16-byte aligned pointers");
    if (blocks < 1 || blocks > (1 << 20) || (unroll != 1 && unroll != 2 && unroll != 4 && unroll != 8))
        return fail(SINGA_E_SHAPE, "calib_copy16: blocks >= 1, unroll in {1, 2, 4, 8}");
    const float4* s4 = (const float4*)src;
    float4* d4 = (float4*)dst;
    hipStream_t st = (hipStream_t)stream;
    if (unroll == 1) hipLaunchKernelGGL(calib_copy16_kernel<1>, dim3(blocks), dim3(256), 0, st, s4, d4, n / 4);
    else if (unroll == 2) hipLaunchKernelGGL(calib_copy16_kernel<2>, dim3(blocks), dim3(256), 0, st, s4, d4, n / 4);
    else if (unroll == 4) hipLaunchKernelGGL(calib_copy16_kernel<4>, dim3(blocks), dim3(256), 0, st, s4, d4, n / 4);
    else hipLaunchKernelGGL(calib_copy16_kernel<8>, dim3(blocks), dim3(256), 0, st, s4, d4, n / 4);
    return check_launch("calib_copy16");
}

int singa_version(void) { return 100; }

const char* singa_last_error_string(void) { return g_err; }

int singa_init(const double* jd_flat, int lmax_max) {
    if (!jd_flat) return fail(SINGA_E_NULL, "singa_init: jd_flat is null");
    if (lmax_max < 0 || lmax_max > 11) return fail(SINGA_E_LMAX, "singa_init: lmax_max must be in [0, 11]");
    static float host[MAX_J];
    int n = j_off(lmax_max + 1);
    for (int i = 0; i < n; ++i) host[i] = (float)jd_flat[i];
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_J), host, n * sizeof(float), 0, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "singa_init: %s", hipGetErrorString(e));
        return (int)e;
    }
    g_lmax_init = lmax_max;
    return SINGA_OK;
}

int singa_dims(int lmax, int mmax, int* kr, int* wsz, int* rad_rows) {
    SINGA_DISPATCH_L(lmax, mmax, {
        using I = SO3Idx<L_, 2>;
        if (kr) *kr = I::KR;
        if (wsz) *wsz = I::WSZ;
        if (rad_rows) *rad_rows = I::RAD_ROWS;
    });
    return SINGA_OK;
}

int singa_edge_frames(const float* vec, const float* rnd, float* rot, float* stats, int E, void* stream) {
    if (!vec || !rnd || !rot || !stats) return fail(SINGA_E_NULL, "edge_frames: null pointer");
    if (E <= 0) return SINGA_OK;
    hipLaunchKernelGGL(edge_frames_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, vec, rnd, rot,
                       (int*)stats, E);
    return check_launch("edge_frames");
}

int singa_wigner_rows(const float* rot, float* wr, int E, int lmax, int mmax, void* stream) {
    if (!rot || !wr) return fail(SINGA_E_NULL, "wigner_rows: null pointer");
    if (g_lmax_init < lmax) return fail(SINGA_E_NOINIT, "wigner_rows: singa_init not called for this lmax");
    if (E <= 0) return SINGA_OK;
    SINGA_DISPATCH_L(lmax, mmax, {
        using I = SO3Idx<L_, 2>;
        long long total = (long long)E * I::KR;
        int blocks = (int)((total + 255) / 256);
        hipLaunchKernelGGL((wigner_rows_kernel<L_, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, rot, wr, E);
    });
    return check_launch("wigner_rows");
}

int singa_gather_rotate_fwd(const float* x_src, const float* x_dst, const int32_t* src, const int32_t* dst,
                            const float* wr, const float* rad, float* out, int E, int C, int lmax, int mmax,
                            void* stream) {
    if (!x_src || !x_dst || !src || !dst || !wr || !out) return fail(SINGA_E_NULL, "gather_rotate_fwd: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "gather_rotate: built for C = 16 sphere channels");
    if (E <= 0) return SINGA_OK;
    SINGA_DISPATCH_L(lmax, mmax, {
        if (rad)
            SINGA_LAUNCH(SINGA_PROF_K4_FWD, E, 0, (gather_rotate_kernel<L_, 2, 16, 0, true>), dim3(grid_for(E)), dim3(64),
                         (hipStream_t)stream, x_src, x_dst, src, dst, wr, rad, (const float*)nullptr, out, E);
        else
            SINGA_LAUNCH(SINGA_PROF_K4_FWD, E, 0, (gather_rotate_kernel<L_, 2, 16, 0, false>), dim3(grid_for(E)), dim3(64),
                         (hipStream_t)stream, x_src, x_dst, src, dst, wr, rad, (const float*)nullptr, out, E);
    });
    return check_launch("gather_rotate_fwd");
}

int singa_gather_rotate_bwd(const float* g_out, const float* x_src, const float* x_dst, const int32_t* src,
                            const int32_t* dst, const float* wr, const float* rad, const int32_t* row_ptr,
                            const int32_t* col_ptr, const int32_t* eperm, float* g_rad, float* gx_src, float* gx_dst,
                            int E, int Ns, int Nd, int C, int lmax, int mmax, void* stream) {
    if (!g_out || !x_src || !x_dst || !src || !dst || !wr || !row_ptr || !col_ptr || !eperm || !gx_src || !gx_dst)
        return fail(SINGA_E_NULL, "gather_rotate_bwd: null pointer");
    if (g_rad && !rad) return fail(SINGA_E_NULL, "gather_rotate_bwd: g_rad requested without rad");
    if (C != 16) return fail(SINGA_E_SHAPE, "gather_rotate: built for C = 16 sphere channels");
    SINGA_DISPATCH_L(lmax, mmax, {
        hipStream_t st = (hipStream_t)stream;
        if (g_rad && E > 0)
            SINGA_LAUNCH(SINGA_PROF_K4_BWD_RAD, E, 0, (gather_rotate_kernel<L_, 2, 16, 1, true>), dim3(grid_for(E)), dim3(64), st,
                         x_src, x_dst, src, dst, wr, rad, g_out, g_rad, E);
        if (Nd > 0) {
            if (rad)
                SINGA_LAUNCH(SINGA_PROF_K4_BWD_DST, E, Nd, (gather_rotate_bwd_node_kernel<L_, 2, 16, 0, true>),
                             dim3(grid_for((Nd + 15) / 16)), dim3(256), st, g_out, wr, rad, row_ptr, (const int*)nullptr, gx_dst, Nd);
            else
                SINGA_LAUNCH(SINGA_PROF_K4_BWD_DST, E, Nd, (gather_rotate_bwd_node_kernel<L_, 2, 16, 0, false>),
                             dim3(grid_for((Nd + 15) / 16)), dim3(256), st, g_out, wr, rad, row_ptr, (const int*)nullptr, gx_dst, Nd);
        }
        if (Ns > 0) {
            if (rad)
                SINGA_LAUNCH(SINGA_PROF_K4_BWD_SRC, E, Ns, (gather_rotate_bwd_node_kernel<L_, 2, 16, 1, true>),
                             dim3(grid_for((Ns + 15) / 16)), dim3(256), st, g_out, wr, rad, col_ptr, eperm, gx_src, Ns);
            else
                SINGA_LAUNCH(SINGA_PROF_K4_BWD_SRC, E, Ns, (gather_rotate_bwd_node_kernel<L_, 2, 16, 1, false>),
                             dim3(grid_for((Ns + 15) / 16)), dim3(256), st, g_out, wr, rad, col_ptr, eperm, gx_src, Ns);
        }
    });
    return check_launch("gather_rotate_bwd");
}

int singa_rotate_back_scatter_fwd(const singa_seg_t* msg, int nseg, const float* alpha, const float* wr,
                                  const int32_t* row_ptr, float* out, int Nd, int CH, int heads, int lmax, int mmax,
                                  int m0_only, float out_scale, void* stream) {
    Segs s;
    if (!pack(msg, nseg, &s) || !wr || !row_ptr || !out) return fail(SINGA_E_NULL, "rotate_back_scatter_fwd: null pointer");
    if (CH < 1 || CH > 128 || heads < 1 || CH % heads) return fail(SINGA_E_SHAPE, "rotate_back_scatter: CH must be <= 128 and divisible by heads");
    if (!m0_only && nseg != 3) return fail(SINGA_E_SHAPE, "rotate_back_scatter: full mode takes the 3 per-m segments");
    if (m0_only ? (CH != 16 || heads != 1) : (CH != 112 || heads != 7))
        return fail(SINGA_E_SHAPE, "rotate_back_scatter: built for 7 heads x 16 value channels (full) and 16 channels (m0_only)");
    if (Nd <= 0) return SINGA_OK;
    int bs = CH <= 64 ? 64 : 128;
    const int n_edges_hint = g_prof_edges_hint;
    SINGA_DISPATCH_L(lmax, mmax, {
        if (!m0_only && (s.rows[0] != L_ + 1 || s.rows[1] != 2 * L_ || s.rows[2] != 2 * (L_ - 1)))
            return fail(SINGA_E_SHAPE, "rotate_back_scatter: segment row counts must be L+1, 2L, 2(L-1)");
        if (m0_only)
            hipLaunchKernelGGL((rotate_back_scatter_kernel<L_, 2, true, 4, 16, 16>), dim3(grid_for(Nd, 1 << 20)), dim3(bs),
                               0, (hipStream_t)stream, s, alpha, wr, row_ptr, out, Nd, out_scale);
        else
            SINGA_LAUNCH(SINGA_PROF_K10_FWD, n_edges_hint, Nd,
                         (rotate_back_scatter_kernel<L_, 2, false, (L_ == 2 ? 4 : 0), 112, 16>), dim3(grid_for(Nd, 1 << 20)),
                         dim3(bs), (hipStream_t)stream, s, alpha, wr, row_ptr, out, Nd, out_scale);
    });
    return check_launch("rotate_back_scatter_fwd");
}

int singa_rotate_back_scatter_bwd(const float* g_out, const singa_seg_t* msg, const singa_seg_mut_t* g_msg, int nseg,
                                  const float* alpha, const float* wr, const int32_t* row_ptr, float* g_alpha_part,
                                  int Nd, int CH, int heads, int lmax, int mmax, int m0_only, float out_scale,
                                  void* stream) {
    Segs s;
    SegsMut gm;
    memset(&s, 0, sizeof(s));
    if (!g_out || !pack_mut(g_msg, nseg, &gm) || !wr || !row_ptr) return fail(SINGA_E_NULL, "rotate_back_scatter_bwd: null pointer");
    if (alpha && (!pack(msg, nseg, &s) || !g_alpha_part)) return fail(SINGA_E_NULL, "rotate_back_scatter_bwd: alpha given without msg / g_alpha_part");
    if (CH < 1 || CH > 128 || heads < 1 || CH % heads) return fail(SINGA_E_SHAPE, "rotate_back_scatter: CH must be <= 128 and divisible by heads");
    if (!m0_only && nseg != 3) return fail(SINGA_E_SHAPE, "rotate_back_scatter: full mode takes the 3 per-m segments");
    if (m0_only ? (CH != 16 || heads != 1) : (CH != 112 || heads != 7))
        return fail(SINGA_E_SHAPE, "rotate_back_scatter: built for 7 heads x 16 value channels (full) and 16 channels (m0_only)");
    if (Nd <= 0) return SINGA_OK;
    int bs = CH <= 64 ? 64 : 128;
    SINGA_DISPATCH_L(lmax, mmax, {
        if (!m0_only && (gm.rows[0] != L_ + 1 || gm.rows[1] != 2 * L_ || gm.rows[2] != 2 * (L_ - 1)))
            return fail(SINGA_E_SHAPE, "rotate_back_scatter: segment row counts must be L+1, 2L, 2(L-1)");
        if (m0_only)
            hipLaunchKernelGGL((rotate_back_scatter_bwd_kernel<L_, 2, true, 16, 16>), dim3(grid_for(Nd, 1 << 20)), dim3(bs), 0,
                               (hipStream_t)stream, g_out, s, gm, alpha, wr, row_ptr, g_alpha_part, Nd, out_scale);
        else
            SINGA_LAUNCH(SINGA_PROF_K10_BWD, g_prof_edges_hint, Nd, (rotate_back_scatter_bwd_kernel<L_, 2, false, 112, 16>),
                         dim3(grid_for(Nd, 1 << 20)), dim3(bs), (hipStream_t)stream, g_out, s, gm, alpha, wr, row_ptr,
                         g_alpha_part, Nd, out_scale);
    });
    return check_launch("rotate_back_scatter_bwd");
}

int singa_segment_softmax_fwd(const float* x, const int32_t* row_ptr, float* out, int N, int H, float eps,
                              int dense_segments, void* stream) {
    if (!x || !row_ptr || !out) return fail(SINGA_E_NULL, "segment_softmax_fwd: null pointer");
    if (N <= 0 || H <= 0) return SINGA_OK;
    if (H == 4 && dense_segments) {
        hipLaunchKernelGGL(segment_softmax4_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, row_ptr, out, N,
                           eps);
        return check_launch("segment_softmax_fwd");
    }
    int blocks = (int)(((long long)N * H + 255) / 256);
    hipLaunchKernelGGL(segment_softmax_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, row_ptr, out, N,
                       H, eps);
    return check_launch("segment_softmax_fwd");
}

int singa_segment_softmax_bwd(const float* y, const float* gy, const int32_t* row_ptr, float* gx, int N, int H,
                              int dense_segments, void* stream) {
    if (!y || !gy || !row_ptr || !gx) return fail(SINGA_E_NULL, "segment_softmax_bwd: null pointer");
    if (N <= 0 || H <= 0) return SINGA_OK;
    if (H == 4 && dense_segments) {
        hipLaunchKernelGGL(segment_softmax4_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, gy, row_ptr, gx, N);
        return check_launch("segment_softmax_bwd");
    }
    int blocks = (int)(((long long)N * H + 255) / 256);
    hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, gy, row_ptr, gx,
                       N, H);
    return check_launch("segment_softmax_bwd");
}

int singa_segment_wsum_fwd(const float* w, const float* v, const int32_t* row_ptr, float* out, int N, int H, int F,
                           void* stream) {
    if (!w || !v || !row_ptr || !out) return fail(SINGA_E_NULL, "segment_wsum_fwd: null pointer");
    if (H * F > 1024 || H < 1 || F < 1) return fail(SINGA_E_SHAPE, "segment_wsum: H*F must be <= 1024");
    if (N <= 0) return SINGA_OK;
    int bs = ((H * F + 63) / 64) * 64;
    hipLaunchKernelGGL(segment_wsum_fwd_kernel, dim3(grid_for(N, 1 << 20)), dim3(bs), 0, (hipStream_t)stream, w, v,
                       row_ptr, out, N, H, F);
    return check_launch("segment_wsum_fwd");
}

int singa_segment_wsum_bwd(const float* g_out, const float* w, const float* v, const int32_t* row_ptr, float* gw,
                           float* gv, int N, int H, int F, void* stream) {
    if (!g_out || !w || !v || !row_ptr || !gw || !gv) return fail(SINGA_E_NULL, "segment_wsum_bwd: null pointer");
    if (H * F > 1024 || H < 1 || F < 1 || F > 64 || (F & (F - 1))) return fail(SINGA_E_SHAPE, "segment_wsum_bwd: F must be a power of two <= 64");
    if (N <= 0) return SINGA_OK;
    int bs = ((H * F + 63) / 64) * 64;
    hipLaunchKernelGGL(segment_wsum_bwd_kernel, dim3(grid_for(N, 1 << 20)), dim3(bs), 0, (hipStream_t)stream, g_out, w,
                       v, row_ptr, gw, gv, N, H, F);
    return check_launch("segment_wsum_bwd");
}

// (KIN, C, edge?) combinations that are built: attention grids [L][2] on 128 hidden channels over the three per-m
// segments (KIN = 9, 19, 29), FFN grids [L][L] on 512 hidden channels over one [KIN, C] record (KIN = 9, 25, 49).
#define SINGA_DISPATCH_S2(kin, ch, nseg, ...)                                                            \
    do {                                                                                                 \
        if ((nseg) == 3 && (ch) == 128) {                                                                \
            constexpr int C_ = 128; constexpr bool EDGE_ = true;                                         \
            switch (kin) {                                                                               \
                case 9: { constexpr int KIN_ = 9; __VA_ARGS__; } break;                                  \
                case 19: { constexpr int KIN_ = 19; __VA_ARGS__; } break;                                \
                case 29: { constexpr int KIN_ = 29; __VA_ARGS__; } break;                                \
                default: return fail(SINGA_E_SHAPE, "s2act(edge): KIN must be 9, 19 or 29");             \
            }                                                                                            \
        } else if ((nseg) == 1 && (ch) == 512) {                                                         \
            constexpr int C_ = 512; constexpr bool EDGE_ = false;                                        \
            switch (kin) {                                                                               \
                case 9: { constexpr int KIN_ = 9; __VA_ARGS__; } break;                                  \
                case 25: { constexpr int KIN_ = 25; __VA_ARGS__; } break;                                \
                case 49: { constexpr int KIN_ = 49; __VA_ARGS__; } break;                                \
                default: return fail(SINGA_E_SHAPE, "s2act(node): KIN must be 9, 25 or 49");             \
            }                                                                                            \
        } else {                                                                                         \
            return fail(SINGA_E_SHAPE, "s2act: built for (3 segments, C = 128) and (1 segment, C = 512)"); \
        }                                                                                                \
    } while (0)

int singa_s2act_fwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* to_grid,
                    const float* from_grid, float* out, int E, int C, int KIN, int G, void* stream) {
    Segs s;
    if (!pack(x, nseg, &s) || !gate || !to_grid || !from_grid || !out) return fail(SINGA_E_NULL, "s2act_fwd: null pointer");
    if (s.rows[0] + s.rows[1] + s.rows[2] != KIN) return fail(SINGA_E_SHAPE, "s2act: segment rows must sum to KIN");
    if (E <= 0) return SINGA_OK;
    long long EC = (long long)E * C;
    int blocks = (int)((EC + 255) / 256);
    if (nseg == 3 && (s.rows[0] != (KIN + 1) / 5 + 1 || s.rows[1] != 2 * ((KIN + 1) / 5)))
        return fail(SINGA_E_SHAPE, "s2act(edge): segment rows must be L+1, 2L, 2(L-1)");
    SINGA_DISPATCH_S2(KIN, C, nseg, hipLaunchKernelGGL((s2act_fwd_kernel<KIN_, C_, EDGE_>), dim3(blocks), dim3(256), 0,
                                                       (hipStream_t)stream, s, gate, (long long)ldg, to_grid, from_grid,
                                                       out, EC, G));
    return check_launch("s2act_fwd");
}

int singa_s2act_bwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* to_grid,
                    const float* from_grid, const float* g_out, float* gx, float* g_gate, int E, int C, int KIN,
                    int G, void* stream) {
    Segs s;
    if (!pack(x, nseg, &s) || !gate || !to_grid || !from_grid || !g_out || !gx || !g_gate)
        return fail(SINGA_E_NULL, "s2act_bwd: null pointer");
    if (s.rows[0] + s.rows[1] + s.rows[2] != KIN) return fail(SINGA_E_SHAPE, "s2act: segment rows must sum to KIN");
    if (E <= 0) return SINGA_OK;
    long long EC = (long long)E * C;
    int blocks = (int)((EC + 255) / 256);
    if (nseg == 3 && (s.rows[0] != (KIN + 1) / 5 + 1 || s.rows[1] != 2 * ((KIN + 1) / 5)))
        return fail(SINGA_E_SHAPE, "s2act(edge): segment rows must be L+1, 2L, 2(L-1)");
    SINGA_DISPATCH_S2(KIN, C, nseg, hipLaunchKernelGGL((s2act_bwd_kernel<KIN_, C_, EDGE_>), dim3(blocks), dim3(256), 0,
                                                       (hipStream_t)stream, s, gate, (long long)ldg, to_grid, from_grid,
                                                       g_out, gx, g_gate, EC, G));
    return check_launch("s2act_bwd");
}

int singa_edge_logits_fwd(const float* qp, const float* wk, const float* hk, const float* cterm, const int32_t* row_ptr,
                          const int32_t* col, float* qk, int N, int H, int D, float scale, void* stream) {
    if (!qp || !wk || !hk || !cterm || !row_ptr || !col || !qk) return fail(SINGA_E_NULL, "edge_logits_fwd: null pointer");
    if (H != 4 || D != 32) return fail(SINGA_E_SHAPE, "edge_logits: built for H = 4 heads, D = 32 key channels per head");
    if (N <= 0) return SINGA_OK;
    hipLaunchKernelGGL((edge_logits_fwd_kernel<32, 4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, qp, wk,
                       hk, cterm, row_ptr, col, qk, N, scale);
    return check_launch("edge_logits_fwd");
}

int singa_edge_logits_bwd(const float* g, const float* qp, const float* wk, const float* hk, const int32_t* row_ptr,
                          const int32_t* col, const int32_t* col_ptr, const int32_t* eperm, const int32_t* row,
                          float* g_qp, float* g_wk, float* g_hk, float* g_cterm, int N, int H, int D, float scale,
                          void* stream) {
    if (!g || !qp || !wk || !hk || !row_ptr || !col || !col_ptr || !eperm || !row || !g_qp || !g_wk || !g_hk || !g_cterm)
        return fail(SINGA_E_NULL, "edge_logits_bwd: null pointer");
    if (H != 4 || D != 32) return fail(SINGA_E_SHAPE, "edge_logits: built for H = 4 heads, D = 32 key channels per head");
    if (N <= 0) return SINGA_OK;
    hipLaunchKernelGGL((edge_logits_bwd_row_kernel<32, 4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, g,
                       qp, wk, hk, row_ptr, col, g_qp, g_wk, g_cterm, N, scale);
    hipLaunchKernelGGL((edge_logits_bwd_col_kernel<32, 4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, g,
                       qp, wk, col_ptr, eperm, row, g_hk, N, scale);
    return check_launch("edge_logits_bwd");
}

int singa_gather_wsum_fwd(const float* alpha, const float* wv, const float* hv, const int32_t* row_ptr, const int32_t* col,
                          float* out, int N, int H, int F, void* stream) {
    if (!alpha || !wv || !hv || !row_ptr || !col || !out) return fail(SINGA_E_NULL, "gather_wsum_fwd: null pointer");
    if (H != 4 || F != 64) return fail(SINGA_E_SHAPE, "gather_wsum: built for H = 4 heads, F = 64 value channels per head");
    if (N <= 0) return SINGA_OK;
    hipLaunchKernelGGL((gather_wsum_fwd_kernel<4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, alpha, wv,
                       hv, row_ptr, col, out, N);
    return check_launch("gather_wsum_fwd");
}

int singa_gather_wsum_bwd(const float* g, const float* alpha, const float* wv, const float* hv, const int32_t* row_ptr,
                          const int32_t* col, const int32_t* col_ptr, const int32_t* eperm, const int32_t* row,
                          float* g_alpha, float* g_wv, float* g_hv, int N, int H, int F, void* stream) {
    if (!g || !alpha || !wv || !hv || !row_ptr || !col || !col_ptr || !eperm || !row || !g_alpha || !g_wv || !g_hv)
        return fail(SINGA_E_NULL, "gather_wsum_bwd: null pointer");
    if (H != 4 || F != 64) return fail(SINGA_E_SHAPE, "gather_wsum: built for H = 4 heads, F = 64 value channels per head");
    if (N <= 0) return SINGA_OK;
    hipLaunchKernelGGL((gather_wsum_bwd_row_kernel<4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, g,
                       alpha, wv, hv, row_ptr, col, g_alpha, g_wv, N);
    hipLaunchKernelGGL((gather_wsum_bwd_col_kernel<4>), dim3(kn_grid(N)), dim3(64), 0, (hipStream_t)stream, g,
                       alpha, wv, col_ptr, eperm, row, g_hv, N);
    return check_launch("gather_wsum_bwd");
}

#define SINGA_DISPATCH_S2SEP(lmax, edge, ...)                                                            \
    do {                                                                                                 \
        if (edge) {                                                                                      \
            constexpr bool EDGE_ = true; constexpr int C_ = 128;                                         \
            switch (lmax) {                                                                              \
                case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;                                    \
                case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;                                    \
                case 6: { constexpr int L_ = 6; __VA_ARGS__; } break;                                    \
                default: return fail(SINGA_E_LMAX, "s2act_sep: lmax must be 2, 4 or 6");                 \
            }                                                                                            \
        } else {                                                                                         \
            constexpr bool EDGE_ = false; constexpr int C_ = 512;                                        \
            switch (lmax) {                                                                              \
                case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;                                    \
                case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;                                    \
                case 6: { constexpr int L_ = 6; __VA_ARGS__; } break;                                    \
                default: return fail(SINGA_E_LMAX, "s2act_sep: lmax must be 2, 4 or 6");                 \
            }                                                                                            \
        }                                                                                                \
    } while (0)

int singa_s2act_sep_fwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                        const float* A, float* out, int E, int C, int lmax, void* stream) {
    Segs s;
    if (!pack(x, nseg, &s) || !gate || !P || !Q || !A || !out) return fail(SINGA_E_NULL, "s2act_sep_fwd: null pointer");
    const bool edge = nseg == 3;
    if (!(edge && C == 128) && !(nseg == 1 && C == 512))
        return fail(SINGA_E_SHAPE, "s2act_sep: built for (3 segments, C = 128) and (1 segment, C = 512)");
    if (edge && (s.rows[0] != lmax + 1 || s.rows[1] != 2 * lmax || s.rows[2] != 2 * (lmax - 1)))
        return fail(SINGA_E_SHAPE, "s2act_sep(edge): segment rows must be L+1, 2L, 2(L-1)");
    if (!edge && s.rows[0] != (lmax + 1) * (lmax + 1)) return fail(SINGA_E_SHAPE, "s2act_sep(node): rows must be (L+1)^2");
    if (E <= 0) return SINGA_OK;
    long long EC = (long long)E * C;
    int blocks = (int)((EC + 255) / 256);
    const int tag = edge ? SINGA_PROF_S2_EDGE_FWD : SINGA_PROF_S2_NODE_FWD;
    // two channels per thread (packed f32) where every row start is 8-byte aligned and the registers allow it (not the
    // L = 6 node grid: 167 registers per channel)
    bool pair = (((uintptr_t)gate | (uintptr_t)out) & 7) == 0 && ldg % 2 == 0 && (edge || lmax <= 4);
    for (int i = 0; i < nseg; ++i) pair = pair && ((uintptr_t)s.p[i] & 7) == 0 && s.ld[i] % 2 == 0;
    if (pair) {
        const long long EC2 = EC / 2;
        blocks = (int)((EC2 + 255) / 256);
        SINGA_DISPATCH_S2SEP(lmax, edge, SINGA_LAUNCH(tag, E, 0, (s2act_sep_fwd2_kernel<L_, EDGE_, C_>), dim3(blocks), dim3(256),
                                                      (hipStream_t)stream, s, gate, (long long)ldg, P, Q, out, EC2));
        return check_launch("s2act_sep_fwd");
    }
    SINGA_DISPATCH_S2SEP(lmax, edge, SINGA_LAUNCH(tag, E, 0, (s2act_sep_fwd_kernel<L_, EDGE_, C_>), dim3(blocks), dim3(256),
                                                  (hipStream_t)stream, s, gate, (long long)ldg, P, Q, A, out, EC));
    return check_launch("s2act_sep_fwd");
}

int singa_s2act_sep_bwd_seg(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                            const float* A, const float* g_out, const singa_seg_mut_t* gx, float* g_gate, int64_t ld_gg, int E,
                            int C, int lmax, void* stream) {
    Segs s;
    SegsMut so;
    if (!pack(x, nseg, &s) || !pack_mut(gx, nseg, &so) || !gate || !P || !Q || !A || !g_out || !g_gate)
        return fail(SINGA_E_NULL, "s2act_sep_bwd: null pointer");
    const bool edge = nseg == 3;
    if (!(edge && C == 128) && !(nseg == 1 && C == 512))
        return fail(SINGA_E_SHAPE, "s2act_sep: built for (3 segments, C = 128) and (1 segment, C = 512)");
    if (edge && (s.rows[0] != lmax + 1 || s.rows[1] != 2 * lmax || s.rows[2] != 2 * (lmax - 1)))
        return fail(SINGA_E_SHAPE, "s2act_sep(edge): segment rows must be L+1, 2L, 2(L-1)");
    if (!edge && s.rows[0] != (lmax + 1) * (lmax + 1)) return fail(SINGA_E_SHAPE, "s2act_sep(node): rows must be (L+1)^2");
    for (int i = 0; i < nseg; ++i)
        if (so.rows[i] != s.rows[i] || so.ld[i] < (long long)s.rows[i] * C)
            return fail(SINGA_E_SHAPE, "s2act_sep_bwd: gradient segments must mirror the input segments");
    if (ld_gg < C) return fail(SINGA_E_SHAPE, "s2act_sep_bwd: row stride of g_gate below C");
    if (E <= 0) return SINGA_OK;
    long long EC = (long long)E * C;
    int blocks = (int)((EC + 255) / 256);
    const int tag = edge ? SINGA_PROF_S2_EDGE_BWD : SINGA_PROF_S2_NODE_BWD;
    SINGA_DISPATCH_S2SEP(lmax, edge, SINGA_LAUNCH(tag, E, 0, (s2act_sep_bwd_kernel<L_, EDGE_, C_>), dim3(blocks), dim3(256),
                                                  (hipStream_t)stream, s, gate, (long long)ldg, P, Q, A, g_out, so, g_gate,
                                                  (long long)ld_gg, EC, (const float*)nullptr));
    return check_launch("s2act_sep_bwd");
}

int singa_s2act_ffn_bwd(const float* x, const float* gate, int64_t ldg, const float* P, const float* Q, const float* g_small,
                        const float* W2, float* gx, float* g_gate, int N, int C, int lmax, void* stream) {
    if (!x || !gate || !P || !Q || !g_small || !W2 || !gx || !g_gate) return fail(SINGA_E_NULL, "s2act_ffn_bwd: null pointer");
    if (C != 512) return fail(SINGA_E_SHAPE, "s2act_ffn_bwd: built for the feed-forward block's 512 hidden channels");
    if (ldg < C) return fail(SINGA_E_SHAPE, "s2act_ffn_bwd: row stride of the gate below C");
    if (((uintptr_t)g_small & 15)) return fail(SINGA_E_SHAPE, "s2act_ffn_bwd: g_small must be 16-byte aligned");
    if (N <= 0) return SINGA_OK;
    const int K = (lmax + 1) * (lmax + 1);
    Segs s;
    SegsMut so;
    memset(&s, 0, sizeof(s));
    memset(&so, 0, sizeof(so));
    s.p[0] = x; s.ld[0] = (long long)K * C; s.rows[0] = K;
    so.p[0] = gx; so.ld[0] = (long long)K * C; so.rows[0] = K;
    const long long EC = (long long)N * C;                 // a multiple of 256: whole workgroups only (they share LDS + a barrier)
    const int blocks = (int)(EC / 256);
    const bool edge = false;
    SINGA_DISPATCH_S2SEP(lmax, edge, SINGA_LAUNCH(SINGA_PROF_S2_NODE_BWD, N, 0, (s2act_sep_bwd_kernel<L_, false, 512, true>), dim3(blocks),
                                                  dim3(256), (hipStream_t)stream, s, gate, (long long)ldg, P, Q, (const float*)nullptr,
                                                  g_small, so, g_gate, (long long)C, EC, W2));
    (void)edge;
    return check_launch("s2act_ffn_bwd");
}

int singa_s2act_sep_bwd(const singa_seg_t* x, int nseg, const float* gate, int64_t ldg, const float* P, const float* Q,
                        const float* A, const float* g_out, float* gx, float* g_gate, int E, int C, int lmax,
                        void* stream) {
    if (!x || nseg < 1 || nseg > 3 || !gx) return fail(SINGA_E_NULL, "s2act_sep_bwd: null pointer");
    singa_seg_mut_t o[3];
    long long kin = 0, at = 0;
    for (int i = 0; i < nseg; ++i) kin += x[i].rows;
    for (int i = 0; i < nseg; ++i) {                    // one contiguous [E, KIN, C] tensor
        o[i].ptr = gx + at * C;
        o[i].ld = kin * C;
        o[i].rows = x[i].rows;
        at += x[i].rows;
    }
    return singa_s2act_sep_bwd_seg(x, nseg, gate, ldg, P, Q, A, g_out, o, g_gate, C, E, C, lmax, stream);
}


int singa_so3_skinny_nparts(int N, int lmax, int C) {
    if (N <= 0 || !so3_skinny_channels_ok(C)) return 0;
    const int runs = C == 512 ? so3_skinny_reduce_runs<512>(lmax) : so3_skinny_reduce_runs<112>(lmax);
    const int npb = so3_skinny_npb(N, runs);
    return (N + npb - 1) / npb;
}

#define SINGA_SKINNY_LC(lmax, C, ...)                                                      \
    do {                                                                                   \
        if ((C) == 512) {                                                                  \
            constexpr int C_ = 512;                                                        \
            switch (lmax) {                                                                \
                case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;                      \
                case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;                      \
                case 6: { constexpr int L_ = 6; __VA_ARGS__; } break;                      \
                default: return fail(SINGA_E_LMAX, "so3_skinny: lmax must be 2, 4 or 6");  \
            }                                                                              \
        } else {                                                                           \
            constexpr int C_ = 112;                                                        \
            switch (lmax) {                                                                \
                case 2: { constexpr int L_ = 2; __VA_ARGS__; } break;                      \
                case 4: { constexpr int L_ = 4; __VA_ARGS__; } break;                      \
                case 6: { constexpr int L_ = 6; __VA_ARGS__; } break;                      \
                default: return fail(SINGA_E_LMAX, "so3_skinny: lmax must be 2, 4 or 6");  \
            }                                                                              \
        }                                                                                  \
    } while (0)

static int g_skinny_valu = 0;       // tests / lab: 1 = the VALU (lane-broadcast) kernels instead of the MFMA ones
int singa_so3_skinny_variant(int valu) {
    g_skinny_valu = valu ? 1 : 0;
    return SINGA_OK;
}

int singa_so3_skinny_expand(const float* small, const float* W, long long w_l, long long w_c, long long w_u, const float* bias,
                            float* big, int N, int C, int lmax, void* stream) {
    if (!small || !W || !big) return fail(SINGA_E_NULL, "so3_skinny_expand: null pointer");
    if (!so3_skinny_channels_ok(C)) return fail(SINGA_E_SHAPE, "so3_skinny: built for 512 and 112 wide channels");
    if (((uintptr_t)small & 15) || ((uintptr_t)big & 15) || (bias && ((uintptr_t)bias & 15)))
        return fail(SINGA_E_SHAPE, "so3_skinny_expand: 16-byte aligned tensors");
    if (N <= 0) return SINGA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (g_skinny_valu) {
        const int npb = so3_skinny_npb(N, 1536);
        SINGA_SKINNY_LC(lmax, C, hipLaunchKernelGGL((so3_skinny_expand_kernel<L_, C_>),
                                                    dim3((unsigned)(SkinnyCfg<C_>::HALVES * ((N + npb - 1) / npb))),
                                                    dim3(SkinnyCfg<C_>::BLOCK), 0, st, small, W, w_l, w_c, w_u, bias, big, N, npb));
    } else {
        const int npb = (so3_skinny_npb(N, 2048) + 15) / 16 * 16;      // runs of whole 16-node tiles
        SINGA_SKINNY_LC(lmax, C, hipLaunchKernelGGL((so3_skinny_expand_mfma_kernel<L_, C_>),
                                                    dim3((unsigned)(SkinnyCfg<C_>::HALVES * ((N + npb - 1) / npb))),
                                                    dim3(SkinnyCfg<C_>::BLOCK), 0, st, small, W, w_l, w_c, w_u, bias, big, N, npb));
    }
    return check_launch("so3_skinny_expand");
}

int singa_so3_skinny_reduce(const float* small, const float* big, float* part, int N, int C, int lmax, int out_cu, int bias_row,
                            void* stream) {
    if (!small || !big || !part) return fail(SINGA_E_NULL, "so3_skinny_reduce: null pointer");
    if (!so3_skinny_channels_ok(C)) return fail(SINGA_E_SHAPE, "so3_skinny: built for 512 and 112 wide channels");
    if (N <= 0) return SINGA_OK;
    const int runs = C == 512 ? so3_skinny_reduce_runs<512>(lmax) : so3_skinny_reduce_runs<112>(lmax);
    const int npb = so3_skinny_npb(N, runs);
    hipStream_t st = (hipStream_t)stream;
    if (((uintptr_t)small & 15) || ((uintptr_t)big & 15) || ((uintptr_t)part & 15))
        return fail(SINGA_E_SHAPE, "so3_skinny_reduce: 16-byte aligned tensors");
    if (g_skinny_valu)
        SINGA_SKINNY_LC(lmax, C, hipLaunchKernelGGL((so3_skinny_reduce_kernel<L_, C_>),
                                                    dim3((unsigned)(SkinnyCfg<C_>::HALVES * ((N + npb - 1) / npb))),
                                                    dim3(SkinnyCfg<C_>::BLOCK), 0, st, small, big, part, N, npb, out_cu, bias_row));
    else
        SINGA_SKINNY_LC(lmax, C, hipLaunchKernelGGL((so3_skinny_reduce_mfma_kernel<L_, C_>),
                                                    dim3((unsigned)(SkinnyCfg<C_>::HALVES * ((N + npb - 1) / npb))),
                                                    dim3(SkinnyCfg<C_>::BLOCK), 0, st, small, big, part, N, npb, out_cu, bias_row));
    return check_launch("so3_skinny_reduce");
}

int singa_block_weight_fwd(const float* w, float* out, int h, int k, void* stream) {
    if (!w || !out) return fail(SINGA_E_NULL, "block_weight_fwd: null pointer");
    if (h <= 0 || k <= 0) return SINGA_OK;
    const long long n = 4LL * h * k;
    hipLaunchKernelGGL(block_weight_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, out, h, k);
    return check_launch("block_weight_fwd");
}

int singa_block_weight_bwd(const float* g_block, float* g_w, int h, int k, int accumulate, void* stream) {
    if (!g_block || !g_w) return fail(SINGA_E_NULL, "block_weight_bwd: null pointer");
    if (h <= 0 || k <= 0) return SINGA_OK;
    const long long n = 2LL * h * k;
    hipLaunchKernelGGL(block_weight_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_block, g_w, h, k,
                       accumulate);
    return check_launch("block_weight_bwd");
}

int singa_rowdot_nparts(long long M) { return M <= 0 ? 0 : (int)((M + ROWDOT_ROWS - 1) / ROWDOT_ROWS); }

int singa_rowdot_fwd(const float* x, const float* b, float* out, long long M, int D, float scale, void* stream) {
    if (!x || !b || !out) return fail(SINGA_E_NULL, "rowdot_fwd: null pointer");
    if (D != 32) return fail(SINGA_E_SHAPE, "rowdot: built for 32 channels per row");
    if (((uintptr_t)x & 15) || ((uintptr_t)b & 15)) return fail(SINGA_E_SHAPE, "rowdot: 16-byte aligned operands");
    if (M <= 0) return SINGA_OK;
    hipLaunchKernelGGL(rowdot32_fwd_kernel, dim3((unsigned)((M * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, b, out, M, scale);
    return check_launch("rowdot_fwd");
}

int singa_rowdot_bwd(const float* g, const float* x, const float* b, float* gx, float* part, long long M, int D, float scale,
                     void* stream) {
    if (!g || !x || !b || !gx || !part) return fail(SINGA_E_NULL, "rowdot_bwd: null pointer");
    if (D != 32) return fail(SINGA_E_SHAPE, "rowdot: built for 32 channels per row");
    if (((uintptr_t)x & 15) || ((uintptr_t)b & 15) || ((uintptr_t)gx & 15)) return fail(SINGA_E_SHAPE, "rowdot: 16-byte aligned operands");
    if (M <= 0) return SINGA_OK;
    hipLaunchKernelGGL(rowdot32_bwd_kernel, dim3((unsigned)singa_rowdot_nparts(M)), dim3(256), 0, (hipStream_t)stream, g, x, b, gx, part, M,
                       scale);
    return check_launch("rowdot_bwd");
}

static int g_lap_fsi_min = 384;      // components of at least this many atoms take the sparse route (lab switch: singa_lap_pe_fsi_min)
int singa_lap_pe_fsi_min(int n) {
    if (n < 0) return fail(SINGA_E_SHAPE, "lap_pe_fsi_min: >= 0 (a value above 896 switches the sparse route off)");
    g_lap_fsi_min = n;
    return SINGA_OK;
}

int singa_lap_pe_work(int B, int ld) {
#ifdef SINGA_EMUL
    return (B < 0 || ld < 0) ? 0 : B * (3 + 4 * 9) * ld;
#else
    return (B < 0 || ld < 0) ? 0 : B * LAP_WORK_PER_LD * ld;
#endif
}

int singa_lap_pe(double* A, const int32_t* esrc, const int32_t* edst, const int32_t* eptr, const int32_t* nnodes, const int32_t* first,
                 double* work, float* out, int B, int ld, int kout, void* stream) {
    if (!A || !esrc || !edst || !eptr || !nnodes || !first || !work || !out) return fail(SINGA_E_NULL, "lap_pe: null pointer");
    if (kout < 1 || kout > 8) return fail(SINGA_E_SHAPE, "lap_pe: 1..8 eigenvectors");
    if (ld < 1 || ld > 896) return fail(SINGA_E_SHAPE, "lap_pe: graphs of up to 896 atoms (22.5 vectors of LDS per graph)");
    if (B <= 0) return SINGA_OK;
#ifdef SINGA_EMUL
    return fail(SINGA_E_SHAPE, "lap_pe: not part of the emulation build");
#else
    const int sl = ld < 32 ? 32 : ld;
    const size_t lds = (size_t)((4 + 16) * sl + 16 + 8) * sizeof(double) + (size_t)(5 * sl + 4) * sizeof(int);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (ld <= 512) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(lap_pe_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "lap_pe: LDS size refused");
        hipLaunchKernelGGL(lap_pe_kernel<8>, dim3(B), dim3(1024), lds, st, A, esrc, edst, eptr, nnodes, first, work, out, ld, kout, g_lap_fsi_min);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(lap_pe_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "lap_pe: LDS size refused");
        hipLaunchKernelGGL(lap_pe_kernel<16>, dim3(B), dim3(1024), lds, st, A, esrc, edst, eptr, nnodes, first, work, out, ld, kout, g_lap_fsi_min);
    }
    return check_launch("lap_pe");
#endif
}

int singa_adam_step(float* const* p, const float* const* g, float* const* m, float* const* v, const long long* sizes,
                    const int32_t* chunk_tensor, const long long* chunk_off, int nchunks, int chunk, float* step,
                    const float* lr, float beta1, float beta2, float eps, void* stream) {
    if (!p || !g || !m || !v || !sizes || !chunk_tensor || !chunk_off || !step || !lr)
        return fail(SINGA_E_NULL, "adam_step: null pointer");
    if (nchunks <= 0) return SINGA_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sizes, chunk_tensor,
                       chunk_off, chunk, step, lr, beta1, beta2, eps);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step);
    return check_launch("adam_step");
}

int singa_grad_norm(const float* const* g, const long long* sizes, const int32_t* chunk_tensor, const long long* chunk_off,
                    int nchunks, int chunk, float* partial, float* out, void* stream) {
    if (!g || !sizes || !chunk_tensor || !chunk_off || !partial || !out) return fail(SINGA_E_NULL, "grad_norm: null pointer");
    if (nchunks <= 0) return SINGA_OK;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, g, sizes, chunk_tensor,
                       chunk_off, chunk, partial);
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nchunks, out);
    return check_launch("grad_norm");
}

int singa_gemm_occupancy(int a_r_contig, int b_r_contig, int cfg) {
    int n = -1;
    hipError_t e = hipErrorInvalidValue;
#define SINGA_OCC(ARC, BRC, CFG) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_f32_kernel<ARC, BRC, CFG>, 256, 0)
#define SINGA_OCC4(ARC, BRC) do { if (cfg == 0) SINGA_OCC(ARC, BRC, 0); else if (cfg == 1) SINGA_OCC(ARC, BRC, 1); else if (cfg == 2) SINGA_OCC(ARC, BRC, 2); else SINGA_OCC(ARC, BRC, 3); } while (0)
    if (a_r_contig && b_r_contig) SINGA_OCC4(true, true);
    else if (a_r_contig) SINGA_OCC4(true, false);
    else SINGA_OCC4(false, false);
#undef SINGA_OCC4
#undef SINGA_OCC
    return e == hipSuccess ? n : -(int)e;
}

static int g_gemm_force_cfg = -1;
int singa_gemm_force_cfg(int cfg) {
    if (cfg < -1 || cfg > 3) return fail(SINGA_E_SHAPE, "gemm_force_cfg: -1 (automatic), 0 (128x128) or 3 (64x64)");
    g_gemm_force_cfg = cfg;
    return SINGA_OK;
}

int singa_gemm_f32(const singa_gemm_t* probs, int n, int a_r_contig, int b_r_contig, int splits, void* stream) {
    if (!probs || n < 1 || n > SINGA_GEMM_MAX) return fail(SINGA_E_SHAPE, "gemm_f32: 1..SINGA_GEMM_MAX problems per launch");
    if (splits < 1) return fail(SINGA_E_SHAPE, "gemm_f32: splits must be >= 1");
    if (!a_r_contig && b_r_contig) return fail(SINGA_E_SHAPE, "gemm_f32: A output-contiguous with B reduction-contiguous is not built");
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    gb.n = n;
    gb.splits = splits;
    long long rmax = 0;
    int jmax = 0;
    for (int k = 0; k < n; ++k) jmax = probs[k].J > jmax ? probs[k].J : jmax;
    int imax = 0;
    for (int k = 0; k < n; ++k) imax = probs[k].I > imax ? probs[k].I : imax;
    // 128 x 32 tiles for outputs with at most 32 columns, 32 x 128 for outputs with at most 32 rows; else 128 x 128 - unless
    // those would leave most of the 256 CUs (two workgroups each) without a tile: then 64 x 64
    int cfg = jmax <= 32 ? 1 : (imax <= 32 ? 2 : 0);
    if (cfg == 0) {
        long long t128 = 0;
        for (int k = 0; k < n; ++k) t128 += (long long)((probs[k].I + 127) / 128) * ((probs[k].J + 127) / 128);
        if (t128 * splits < 384 || (imax <= 64 && jmax <= 64)) cfg = 3;      // (an output of at most 64 x 64 is one small tile)
        if (g_gemm_force_cfg == 0 || g_gemm_force_cfg == 3) cfg = g_gemm_force_cfg;      // tests: both tile shapes on every case
    }
    const int BM = cfg == 2 ? 32 : (cfg == 3 ? 64 : 128), BN = cfg == 1 ? 32 : (cfg == 3 ? 64 : 128);
    int tj_total = 0, ti_max = 0, tiles_seq = 0;
    bool same_rows = true;
    for (int k = 1; k < n; ++k) same_rows = same_rows && probs[k].I == probs[0].I;
    for (int k = 0; k < n; ++k) {
        const singa_gemm_t& q = probs[k];
        GemmProb& P = gb.p[k];
        if (!q.a || !q.b || !q.c) return fail(SINGA_E_NULL, "gemm_f32: null operand");
        if (q.I < 0 || q.J < 0 || q.R < 0) return fail(SINGA_E_SHAPE, "gemm_f32: negative size");
        // float4 accesses: the contiguous axis of each operand must be a multiple of 4 floats and 16-byte aligned
        const bool a_ok = a_r_contig ? (q.R % 4 == 0) : (q.I % 4 == 0);
        const bool b_ok = b_r_contig ? (q.R % 4 == 0) : (q.J % 4 == 0);
        if (!a_ok || !b_ok || q.lda % 4 || q.ldb % 4 || q.a_group_ld % 4 || q.b_group_ld % 4 || ((uintptr_t)q.a & 15) ||
            ((uintptr_t)q.b & 15))
            return fail(SINGA_E_SHAPE, "gemm_f32: contiguous axes must be multiples of 4 floats and 16-byte aligned");
        if (b_r_contig && q.b_group > 0) return fail(SINGA_E_SHAPE, "gemm_f32: a reduction-contiguous B has plain rows");
        // float4 stores of the result: J, the row pitches and the base must be multiples of 4 floats / 16 bytes
        if (q.J % 4 || q.ldc % 4 || q.c_group_ld % 4 || q.c_split_stride % 4 || ((uintptr_t)q.c & 15) ||
            (q.bias && ((uintptr_t)q.bias & 15)))
            return fail(SINGA_E_SHAPE, "gemm_f32: the result's columns, pitches and base must be multiples of 4 floats / 16 bytes");
        if (splits > 1 && (q.c_group > 0 || q.ldc != q.J || q.bias || q.mask || q.addend || q.relu ||
                           q.c_split_stride < (long long)q.I * q.J))
            return fail(SINGA_E_SHAPE, "gemm_f32: split reductions write dense [I, J] partial slabs (c_split_stride apart), no epilogue options");
        if ((q.mask && q.c_group > 0) || ((uintptr_t)q.mask & 15) || ((uintptr_t)q.addend & 15))
            return fail(SINGA_E_SHAPE, "gemm_f32: mask / addend have the result's layout (a mask: plain rows only) and are 16-byte aligned");
        if (q.asum && (a_r_contig || q.a_group > 0 || q.asum_stride < q.I))
            return fail(SINGA_E_SHAPE, "gemm_f32: asum goes with the (0, 0) form, plain A rows, a slab of at least I floats per split");
        P.A = q.a; P.B = q.b; P.C = q.c; P.bias = q.bias; P.mask = q.mask; P.addend = q.addend; P.relu = q.relu;
        P.asum = q.asum; P.asum_stride = q.asum_stride;
        P.lda = q.lda; P.ldb = q.ldb; P.ldc = q.ldc;
        P.a_group = q.a_group > 0 ? q.a_group : (1 << 30);
        P.b_group = q.b_group > 0 ? q.b_group : (1 << 30);
        P.c_group = q.c_group > 0 ? q.c_group : (1 << 30);
        P.a_gld = q.a_group > 0 ? q.a_group_ld : 0;
        P.b_gld = q.b_group > 0 ? q.b_group_ld : 0;
        P.c_gld = q.c_group > 0 ? q.c_group_ld : 0;
        P.c_split = q.c_split_stride;
        P.I = q.I; P.J = q.J; P.R = q.R;
        P.tiles_j = (q.J + BN - 1) / BN;
        // interleaved order: first column tile of this problem inside a row tile's group; else: first tile of the problem
        P.tile_begin = same_rows ? tj_total : tiles_seq;
        tj_total += P.tiles_j;
        const int ti = (q.I + BM - 1) / BM;
        tiles_seq += ti * P.tiles_j;
        if (ti > ti_max) ti_max = ti;
        if (q.R > rmax) rmax = q.R;
    }
    const int tiles = same_rows ? ti_max * tj_total : tiles_seq;
    if (tiles == 0) return SINGA_OK;
    gb.tiles_total = tiles;
    gb.tj_total = same_rows ? tj_total : 0;
    gb.r_chunk = splits > 1 ? ((rmax + splits - 1) / splits + 31) / 32 * 32 : (rmax > 0 ? rmax : 1);
    const long long nblk = (long long)tiles * splits;
    if (nblk > (1 << 30)) return fail(SINGA_E_SHAPE, "gemm_f32: too many tiles");
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SINGA_GEMM_GO(tag, ARC, BRC)                                                                        \
    do {                                                                                                    \
        if (cfg == 1) SINGA_LAUNCH(tag, 0, tiles, (gemm_f32_kernel<ARC, BRC, 1>), grid, block, st, gb);     \
        else if (cfg == 2) SINGA_LAUNCH(tag, 0, tiles, (gemm_f32_kernel<ARC, BRC, 2>), grid, block, st, gb); \
        else if (cfg == 3) SINGA_LAUNCH(tag, 0, tiles, (gemm_f32_kernel<ARC, BRC, 3>), grid, block, st, gb); \
        else SINGA_LAUNCH(tag, 0, tiles, (gemm_f32_kernel<ARC, BRC, 0>), grid, block, st, gb);              \
    } while (0)
    if (a_r_contig && b_r_contig) SINGA_GEMM_GO(SINGA_PROF_GEMM_NT, true, true);
    else if (a_r_contig) SINGA_GEMM_GO(SINGA_PROF_GEMM_NN, true, false);
    else SINGA_GEMM_GO(SINGA_PROF_GEMM_TN, false, false);
#undef SINGA_GEMM_GO
    return check_launch("gemm_f32");
}

int singa_cgemm3m_f32(const singa_cgemm_t* probs, int n, int a_r_contig, int b_r_contig, int splits, void* stream) {
    if (!probs || n < 1 || n > SINGA_CGEMM_MAX) return fail(SINGA_E_SHAPE, "cgemm3m_f32: 1..SINGA_CGEMM_MAX problems per launch");
    if (splits < 1) return fail(SINGA_E_SHAPE, "cgemm3m_f32: splits must be >= 1");
    if (!a_r_contig && b_r_contig) return fail(SINGA_E_SHAPE, "cgemm3m_f32: A output-contiguous with B reduction-contiguous is not built");
    CGemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    gb.n = n;
    gb.splits = splits;
    constexpr int BM = 128, BN = 64;
    long long rmax = 0;
    int tj_total = 0, ti_max = 0, tiles_seq = 0;
    bool same_rows = true;
    for (int k = 1; k < n; ++k) same_rows = same_rows && probs[k].I == probs[0].I;
    for (int k = 0; k < n; ++k) {
        const singa_cgemm_t& q = probs[k];
        CGemmProb& P = gb.p[k];
        if (!q.a || !q.b || !q.c) return fail(SINGA_E_NULL, "cgemm3m_f32: null operand");
        if (q.I < 0 || q.J < 0 || q.R < 0) return fail(SINGA_E_SHAPE, "cgemm3m_f32: negative size");
        const bool a_ok = a_r_contig ? (q.R % 4 == 0) : (q.I % 4 == 0);
        const bool b_ok = b_r_contig ? (q.R % 4 == 0) : (q.J % 4 == 0);
        if (!a_ok || !b_ok || q.J % 4 || q.lda % 4 || q.ldb % 4 || q.ldc % 4 || q.a_im % 4 || q.b_im % 4 || q.c_im % 4 ||
            q.c_split_stride % 4 || ((uintptr_t)q.a & 15) || ((uintptr_t)q.b & 15) || ((uintptr_t)q.c & 15))
            return fail(SINGA_E_SHAPE, "cgemm3m_f32: contiguous axes, pitches and part offsets must be multiples of 4 floats, bases 16-byte aligned");
        if (q.sigma != 1.0f && q.sigma != -1.0f) return fail(SINGA_E_SHAPE, "cgemm3m_f32: sigma is +1 or -1");
        if (splits > 1 && q.c_split_stride <= 0) return fail(SINGA_E_SHAPE, "cgemm3m_f32: split reductions need c_split_stride");
        P.A = q.a; P.B = q.b; P.C = q.c;
        P.lda = q.lda; P.ldb = q.ldb; P.ldc = q.ldc;
        P.a_im = q.a_im; P.b_im = q.b_im; P.c_im = q.c_im;
        P.c_split = q.c_split_stride;
        P.I = q.I; P.J = q.J; P.R = q.R;
        P.sigma = q.sigma;
        P.tiles_j = (q.J + BN - 1) / BN;
        P.tile_begin = same_rows ? tj_total : tiles_seq;
        tj_total += P.tiles_j;
        const int ti = (q.I + BM - 1) / BM;
        tiles_seq += ti * P.tiles_j;
        if (ti > ti_max) ti_max = ti;
        if (q.R > rmax) rmax = q.R;
    }
    const int tiles = same_rows ? ti_max * tj_total : tiles_seq;
    if (tiles == 0) return SINGA_OK;
    gb.tiles_total = tiles;
    gb.tj_total = same_rows ? tj_total : 0;
    gb.r_chunk = splits > 1 ? ((rmax + splits - 1) / splits + 15) / 16 * 16 : (rmax > 0 ? rmax : 1);
    const long long nblk = (long long)tiles * splits;
    if (nblk > (1 << 30)) return fail(SINGA_E_SHAPE, "cgemm3m_f32: too many tiles");
    const dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (a_r_contig && b_r_contig) SINGA_LAUNCH(SINGA_PROF_CGEMM_NT, 0, tiles, (cgemm3m_f32_kernel<true, true>), grid, block, st, gb);
    else if (a_r_contig) SINGA_LAUNCH(SINGA_PROF_CGEMM_NN, 0, tiles, (cgemm3m_f32_kernel<true, false>), grid, block, st, gb);
    else SINGA_LAUNCH(SINGA_PROF_CGEMM_TN, 0, tiles, (cgemm3m_f32_kernel<false, false>), grid, block, st, gb);
    return check_launch("cgemm3m_f32");
}

int singa_alpha_logits_nslots(int E) {
    long long slots = ((long long)E + 0) < 1 ? 1 : E;
    long long cap = 256 * 8 * 8;   // 2048 blocks of 256 threads = 16384 slots
    return (int)(slots < cap ? ((slots + 7) / 8) * 8 : cap);
}

int singa_alpha_logits_fwd(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot, float* logits,
                           int E, int heads, int A, float eps, void* stream) {
    if (!h0 || !ln_w || !ln_b || !dot || !logits) return fail(SINGA_E_NULL, "alpha_logits_fwd: null pointer");
    if (heads != 7 || A != 32) return fail(SINGA_E_SHAPE, "alpha_logits: built for 7 heads x 32 alpha channels");
    if (E <= 0) return SINGA_OK;
    int blocks = singa_alpha_logits_nslots(E) / 8;
    hipLaunchKernelGGL((alpha_logits_fwd_kernel<7>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, h0, ld, ln_w, ln_b, dot,
                       logits, E, eps);
    return check_launch("alpha_logits_fwd");
}

int singa_alpha_logits_bwd_ld(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot,
                              const float* g_logits, float* g_x, long long ld_gx, float* part, int E, int heads, int A, float eps,
                              void* stream) {
    if (!h0 || !ln_w || !ln_b || !dot || !g_logits || !g_x || !part) return fail(SINGA_E_NULL, "alpha_logits_bwd: null pointer");
    if (heads != 7 || A != 32) return fail(SINGA_E_SHAPE, "alpha_logits: built for 7 heads x 32 alpha channels");
    if (ld_gx < (long long)heads * A) return fail(SINGA_E_SHAPE, "alpha_logits_bwd: row stride of g_x below heads * A");
    if (E <= 0) return SINGA_OK;
    int blocks = singa_alpha_logits_nslots(E) / 8;
    hipLaunchKernelGGL((alpha_logits_bwd_kernel<7>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, h0, ld, ln_w, ln_b, dot,
                       g_logits, g_x, ld_gx, part, E, eps);
    return check_launch("alpha_logits_bwd");
}

int singa_alpha_logits_bwd(const float* h0, long long ld, const float* ln_w, const float* ln_b, const float* dot,
                           const float* g_logits, float* g_x, float* part, int E, int heads, int A, float eps, void* stream) {
    return singa_alpha_logits_bwd_ld(h0, ld, ln_w, ln_b, dot, g_logits, g_x, (long long)heads * A, part, E, heads, A, eps, stream);
}

int singa_ln_silu_nparts(long long M) {
    long long blocks = (M + 63) / 64;
    return (int)(blocks < 256 ? (blocks < 1 ? 1 : blocks) : 256) * 64;      // threads = rows of the partial buffer
}

int singa_ln_silu_fwd(const float* x, const float* gamma, const float* beta, float* out, long long M, int C, float eps,
                      void* stream) {
    if (!x || !gamma || !beta || !out) return fail(SINGA_E_NULL, "ln_silu_fwd: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "ln_silu: built for rows of 16 channels");
    if (M <= 0) return SINGA_OK;
    long long blocks = (M + 63) / 64;
    hipLaunchKernelGGL((ln_silu_fwd_kernel<16>), dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(64), 0,
                       (hipStream_t)stream, x, gamma, beta, out, M, eps);
    return check_launch("ln_silu_fwd");
}

int singa_ln_silu_bwd(const float* x, const float* gamma, const float* beta, const float* g_out, float* g_x, float* part,
                      long long M, int C, float eps, void* stream) {
    if (!x || !gamma || !beta || !g_out || !g_x || !part) return fail(SINGA_E_NULL, "ln_silu_bwd: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "ln_silu: built for rows of 16 channels");
    if (M <= 0) return SINGA_OK;
    hipLaunchKernelGGL((ln_silu_bwd_kernel<16>), dim3(singa_ln_silu_nparts(M) / 64), dim3(64), 0, (hipStream_t)stream, x, gamma,
                       beta, g_out, g_x, part, M, eps);
    return check_launch("ln_silu_bwd");
}

int singa_bias_ssp_fwd(const float* u, const float* b, float* y, long long M, int n, void* stream) {
    if (!u || !b || !y) return fail(SINGA_E_NULL, "bias_ssp_fwd: null pointer");
    if (n <= 0 || n % 4) return fail(SINGA_E_SHAPE, "bias_ssp: row length must be a positive multiple of 4");
    if (M <= 0) return SINGA_OK;
    const long long total4 = M * n / 4;
    hipLaunchKernelGGL(bias_ssp_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, b, y,
                       total4, n);
    return check_launch("bias_ssp_fwd");
}

int singa_bias_ssp_bwd(const float* u, const float* b, const float* g, float* gu, long long M, int n, void* stream) {
    if (!u || !b || !g || !gu) return fail(SINGA_E_NULL, "bias_ssp_bwd: null pointer");
    if (n <= 0 || n % 4) return fail(SINGA_E_SHAPE, "bias_ssp: row length must be a positive multiple of 4");
    if (M <= 0) return SINGA_OK;
    const long long total4 = M * n / 4;
    hipLaunchKernelGGL(bias_ssp_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, b, g, gu,
                       total4, n);
    return check_launch("bias_ssp_bwd");
}

int singa_ln256_nparts(long long M) {
    long long blocks = (M + 3) / 4;
    return (int)(blocks < 1 ? 1 : (blocks < 1024 ? blocks : 1024)) * 4;     // wavefronts = rows of the partial buffer
}

int singa_ln256_fwd(const float* a, const float* r, const float* gamma, const float* beta, float* y, long long M, int C,
                    float eps, void* stream) {
    if (!a || !gamma || !beta || !y) return fail(SINGA_E_NULL, "ln256_fwd: null pointer");
    if (C != 256) return fail(SINGA_E_SHAPE, "ln256: built for rows of 256 channels");
    if (M <= 0) return SINGA_OK;
    long long blocks = (M + 3) / 4;
    hipLaunchKernelGGL(ln256_fwd_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, a, r,
                       gamma, beta, y, M, eps);
    return check_launch("ln256_fwd");
}

int singa_ln256_bwd(const float* a, const float* r, const float* gamma, const float* g, float* gs, float* part, long long M,
                    int C, float eps, void* stream) {
    if (!a || !gamma || !g || !gs || !part) return fail(SINGA_E_NULL, "ln256_bwd: null pointer");
    if (C != 256) return fail(SINGA_E_SHAPE, "ln256: built for rows of 256 channels");
    if (M <= 0) return SINGA_OK;
    hipLaunchKernelGGL(ln256_bwd_kernel, dim3(singa_ln256_nparts(M) / 4), dim3(256), 0, (hipStream_t)stream, a, r, gamma, g, gs,
                       part, M, eps);
    return check_launch("ln256_bwd");
}

int singa_dec_self_attn(const float* x, const float* wqkv_t, const float* bqkv, const float* wo_t, const float* bo,
                        const float* gamma, const float* beta, float* k_cache, float* v_cache, const long long* pos, int R, int P,
                        float* y, float eps, void* stream) {
    if (!x || !wqkv_t || !bqkv || !wo_t || !bo || !gamma || !beta || !k_cache || !v_cache || !pos || !y)
        return fail(SINGA_E_NULL, "dec_self_attn: null pointer");
    if (P <= 0 || P > 256) return fail(SINGA_E_SHAPE, "dec_self_attn: built for at most 256 cached positions");
    if (R <= 0) return SINGA_OK;
    hipLaunchKernelGGL(dec_self_attn_kernel, dim3(R), dim3(1024), 0, (hipStream_t)stream, x, wqkv_t, bqkv, wo_t, bo, gamma, beta,
                       k_cache, v_cache, pos, P, y, eps);
    return check_launch("dec_self_attn");
}

int singa_dec_cross_attn(const float* y, const float* wq_t, const float* bq, const float* ck, const float* cv,
                         const unsigned char* pad, const float* wo_t, const float* bo, const float* gamma, const float* beta,
                         int R, int beams, int S, float* z, float eps, void* stream) {
    if (!y || !wq_t || !bq || !ck || !cv || !pad || !wo_t || !bo || !gamma || !beta || !z)
        return fail(SINGA_E_NULL, "dec_cross_attn: null pointer");
    if (S <= 0 || S > DEC_MAX_S) return fail(SINGA_E_SHAPE, "dec_cross_attn: built for at most 1024 encoder positions");
    if (beams <= 0 || R % beams) return fail(SINGA_E_SHAPE, "dec_cross_attn: rows must be proteins x beams");
    if (R <= 0) return SINGA_OK;
    hipLaunchKernelGGL(dec_cross_attn_kernel, dim3(R), dim3(1024), 0, (hipStream_t)stream, y, wq_t, bq, ck, cv, pad, wo_t, bo, gamma,
                       beta, beams, S, z, eps);
    return check_launch("dec_cross_attn");
}

int singa_dec_ffn(const float* z, const float* w1_t, const float* b1, const float* w2_t, const float* b2, const float* gamma,
                  const float* beta, int R, float* out, float eps, void* stream) {
    if (!z || !w1_t || !b1 || !w2_t || !b2 || !gamma || !beta || !out) return fail(SINGA_E_NULL, "dec_ffn: null pointer");
    if (R <= 0) return SINGA_OK;
    hipLaunchKernelGGL(dec_ffn_kernel, dim3(R), dim3(1024), 0, (hipStream_t)stream, z, w1_t, b1, w2_t, b2, gamma, beta, out, eps);
    return check_launch("dec_ffn");
}

int singa_edge_mlp_fwd(const float* attr, const float* w1tk, const float* b1k, const float* w2tk, const float* b2k,
                       const float* w1tv, const float* b1v, const float* w2tv, const float* b2v, float* wk, float* wv, int E,
                       int CIN, int HK, int HV, void* stream) {
    if (!attr || !w1tk || !b1k || !w2tk || !b2k || !w1tv || !b1v || !w2tv || !b2v || !wk || !wv)
        return fail(SINGA_E_NULL, "edge_mlp_fwd: null pointer");
    if (CIN != 64 || HK != 32 || HV != 64)
        return fail(SINGA_E_SHAPE, "edge_mlp: built for 64 edge channels, 32 key / 64 value channels per head");
    if (E <= 0) return SINGA_OK;
    const long long tiles = ((long long)E + 31) / 32;
    const long long blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL(edge_mlp_mfma_fwd_kernel, dim3((unsigned)(blocks < 768 ? blocks : 768)), dim3(256), 0, (hipStream_t)stream,
                       attr, w1tk, b1k, w2tk, b2k, w1tv, b1v, w2tv, b2v, wk, wv, E);
    return check_launch("edge_mlp_fwd");
}

int singa_edge_mlp_bwd_nparts(int E, int H) {
    const long long blocks = (((long long)E + 31) / 32 + 3) / 4;
    const long long cap = H <= 32 ? 512 : 256;         // x (H / 32) slices = two workgroups per CU (256 CUs)
    return (int)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

int singa_edge_mlp_bwd(const float* attr, const float* g_out, const float* w1t, const float* b1, const float* w2, float* part,
                       int E, int CIN, int H, void* stream) {
    if (!attr || !g_out || !w1t || !b1 || !w2 || !part) return fail(SINGA_E_NULL, "edge_mlp_bwd: null pointer");
    if (CIN != 64 || (H != 32 && H != 64))
        return fail(SINGA_E_SHAPE, "edge_mlp: built for 64 edge channels, 32 key / 64 value channels per head");
    if (E <= 0) return SINGA_OK;
    const int blocks = singa_edge_mlp_bwd_nparts(E, H);
    if (H == 32)
        hipLaunchKernelGGL((edge_mlp_mfma_bwd_kernel<32>), dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, attr, g_out, w1t,
                           b1, w2, part, E);
    else
        hipLaunchKernelGGL((edge_mlp_mfma_bwd_kernel<64>), dim3(blocks, 2), dim3(256), 0, (hipStream_t)stream, attr, g_out, w1t,
                           b1, w2, part, E);
    return check_launch("edge_mlp_bwd");
}

int singa_masked_softmax_fwd(const float* s, const unsigned char* mask, long long mask_stride_b, long long mask_stride_t, float* p,
                             int BH, int T, int S, int heads, float scale, void* stream) {
    if (!s || !mask || !p) return fail(SINGA_E_NULL, "masked_softmax_fwd: null pointer");
    if (heads <= 0 || BH % heads) return fail(SINGA_E_SHAPE, "masked_softmax: BH must be batch x heads");
    if (BH <= 0 || T <= 0 || S <= 0) return SINGA_OK;
    const long long rows = (long long)BH * T;
    hipLaunchKernelGGL(masked_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, mask,
                       mask_stride_b, mask_stride_t, p, rows, T, S, heads, scale);
    return check_launch("masked_softmax_fwd");
}

int singa_masked_softmax_bwd(const float* p, const float* gp, const unsigned char* mask, long long mask_stride_b,
                             long long mask_stride_t, float* gs, int BH, int T, int S, int heads, float scale, void* stream) {
    if (!p || !gp || !mask || !gs) return fail(SINGA_E_NULL, "masked_softmax_bwd: null pointer");
    if (heads <= 0 || BH % heads) return fail(SINGA_E_SHAPE, "masked_softmax: BH must be batch x heads");
    if (BH <= 0 || T <= 0 || S <= 0) return SINGA_OK;
    const long long rows = (long long)BH * T;
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, gp, mask,
                       mask_stride_b, mask_stride_t, gs, rows, T, S, heads, scale);
    return check_launch("masked_softmax_bwd");
}

int singa_attn_fwd(const float* q, const float* k, const float* v, const unsigned char* mask, long long mask_stride_b,
                   long long mask_stride_t, float* ctx, float* lse, int BH, int T, int S, int heads, int DK, int DV,
                   int token_major, long long ld_q, long long ld_k, long long ld_v, float scale, void* stream) {
    if (!q || !k || !v || !mask || !ctx || !lse) return fail(SINGA_E_NULL, "attn_fwd: null pointer");
    if (DK != 32 || DV != 64) return fail(SINGA_E_SHAPE, "attn: built for 32 key / 64 value channels per head");
    if (heads <= 0 || BH % heads) return fail(SINGA_E_SHAPE, "attn: BH must be batch x heads");
    if (BH <= 0 || T <= 0 || S <= 0) return SINGA_OK;
    AttnPitch ld;
    if (!attn_pitch(token_major, heads, ld_q, ld_k, ld_v, q, k, v, &ld)) return fail(SINGA_E_SHAPE, "attn: token pitches must be multiples of 4 floats, at least heads * D, 16-byte aligned bases");
    const long long waves = (long long)BH * ((T + 31) / 32);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(((waves + 3) / 4 + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, q, k, v, mask,
                       mask_stride_b, mask_stride_t, ctx, lse, BH, T, S, heads, token_major ? 1 : 0, scale, ld);
    return check_launch("attn_fwd");
}

int singa_attn_bwd(const float* q, const float* k, const float* v, const unsigned char* mask, long long mask_stride_b,
                   long long mask_stride_t, const float* ctx, const float* lse, const float* g_ctx, float* g_q, float* g_k,
                   float* g_v, float* dsum, int BH, int T, int S, int heads, int DK, int DV, int token_major, long long ld_q,
                   long long ld_k, long long ld_v, float scale, void* stream) {
    if (!q || !k || !v || !mask || !ctx || !lse || !g_ctx || !g_q || !g_k || !g_v || !dsum)
        return fail(SINGA_E_NULL, "attn_bwd: null pointer");
    if (DK != 32 || DV != 64) return fail(SINGA_E_SHAPE, "attn: built for 32 key / 64 value channels per head");
    if (heads <= 0 || BH % heads) return fail(SINGA_E_SHAPE, "attn: BH must be batch x heads");
    if (BH <= 0 || T <= 0 || S <= 0) return SINGA_OK;
    AttnPitch ld;
    if (!attn_pitch(token_major, heads, ld_q, ld_k, ld_v, q, k, v, &ld) || ((uintptr_t)g_q & 15) || ((uintptr_t)g_k & 15) || ((uintptr_t)g_v & 15))
        return fail(SINGA_E_SHAPE, "attn: token pitches must be multiples of 4 floats, at least heads * D, 16-byte aligned bases");
    const long long wq = (long long)BH * ((T + 31) / 32), wk = (long long)BH * ((S + 31) / 32);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)(((wq + 3) / 4 + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, q, k, v, mask,
                       mask_stride_b, mask_stride_t, ctx, lse, g_ctx, g_q, dsum, BH, T, S, heads, token_major ? 1 : 0, scale, ld);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3((unsigned)(((wk + 3) / 4 + 7) / 8 * 8)), dim3(256), 0, (hipStream_t)stream, q, k, v, mask,
                       mask_stride_b, mask_stride_t, lse, dsum, g_ctx, g_k, g_v, BH, T, S, heads, token_major ? 1 : 0, scale, ld);
    return check_launch("attn_bwd");
}

long long singa_colsum_work(long long M, int n) {
    // floats of workspace: the first-level partials plus the second level (ping-pong)
    const int r = colsum_first_r(M);
    long long s1 = (M + r - 1) / r, s2 = (s1 + COLSUM_R - 1) / COLSUM_R;
    return (s1 + s2) * (long long)n + 2;
}

int singa_colsum(const float* x, long long ld, long long M, int n, float* work, float* out, void* stream) {
    if (!x || !work || !out) return fail(SINGA_E_NULL, "colsum: null pointer");
    if (n <= 0 || M <= 0) return SINGA_OK;
    const float* src = x;
    long long rows = M, sld = ld;
    int R = colsum_first_r(M);
    long long s1 = (M + R - 1) / R;
    float* bufs[2] = {work, work + s1 * n};
    int which = 0;
    while (true) {
        long long slabs = (rows + R - 1) / R;
        float* dst = slabs == 1 ? out : bufs[which];
        long long threads = slabs * n;
        hipLaunchKernelGGL(colsum_pass_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           src, sld, rows, n, dst, R);
        if (slabs == 1) break;
        src = dst;
        sld = n;
        rows = slabs;
        which ^= 1;
        R = COLSUM_R;
    }
    return check_launch("colsum");
}

long long singa_colsum_multi_work(long long M, int n) {
    if (M <= 0 || n <= 0) return 0;
    const int r = colsum_multi_r(M, n);
    const long long slabs = (M + r - 1) / r;
    return slabs > 1 ? slabs * (long long)n + 4 : 0;       // (+ 4: every job's partials start 16-byte aligned)
}

int singa_colsum_multi(int n_jobs, const float* const* x, const long long* ld, const long long* M, const int* n,
                       const int* job_seg0, int n_segs, const int* seg_col0, float* const* seg_dst, float* work,
                       long long work_floats, void* stream) {
    if (n_jobs < 0 || n_segs < 0) return fail(SINGA_E_SHAPE, "colsum_multi: negative count");
    if (n_jobs == 0) return SINGA_OK;
    if (!x || !ld || !M || !n || !job_seg0 || !seg_col0 || !seg_dst) return fail(SINGA_E_NULL, "colsum_multi: null table");
    long long woff = 0;
    int k = 0;
    while (k < n_jobs) {
        ColsumBatch b;
        memset(&b, 0, sizeof(b));
        int nj = 0, ns = 0, blk1 = 0, blk2 = 0;
        while (k < n_jobs && nj < COLSUM_MJ) {
            const int s0 = job_seg0[k], s1 = k + 1 < n_jobs ? job_seg0[k + 1] : n_segs;
            if (s0 < 0 || s1 <= s0 || s1 > n_segs || s1 - s0 > COLSUM_MS) return fail(SINGA_E_SHAPE, "colsum_multi: bad segment table");
            if (ns + (s1 - s0) > COLSUM_MS) break;
            if (M[k] <= 0 || n[k] <= 0) { ++k; continue; }          // nothing to add
            if (!x[k] || M[k] > 0x7fffffffLL) return fail(SINGA_E_SHAPE, "colsum_multi: bad job");
            ColsumJob& jb = b.job[nj];
            jb.x = x[k];
            jb.ld = ld[k];
            jb.M = (int)M[k];
            jb.n = n[k];
            jb.R = colsum_multi_r(M[k], n[k]);
            jb.slabs = (int)((M[k] + jb.R - 1) / jb.R);
            jb.blk1 = blk1;
            jb.blk2 = blk2;
            jb.seg0 = ns;
            jb.nseg = s1 - s0;
            bool vec = n[k] % 4 == 0 && ld[k] % 4 == 0 && !((uintptr_t)x[k] & 15) && n[k] >= 1024;
            for (int q = s0; q < s1; ++q) {
                if (!seg_dst[q] || seg_col0[q] < 0 || seg_col0[q] >= n[k] || (q > s0 && seg_col0[q] <= seg_col0[q - 1]) ||
                    (q == s0 && seg_col0[q] != 0))
                    return fail(SINGA_E_SHAPE, "colsum_multi: segments must start at column 0 and ascend");
                b.seg[ns].dst = seg_dst[q];
                b.seg[ns].col0 = seg_col0[q];
                vec = vec && seg_col0[q] % 4 == 0 && !((uintptr_t)seg_dst[q] & 15);
                ++ns;
            }
            if (jb.slabs > 1) {
                const long long need = (long long)jb.slabs * jb.n;
                woff = (woff + 3) & ~3LL;
                if (!work || woff + need > work_floats) return fail(SINGA_E_SHAPE, "colsum_multi: workspace too small");
                vec = vec && !((uintptr_t)work & 15);
                jb.work = work + woff;
                woff += need;
            }
            jb.vec = vec ? 1 : 0;
            const int per = vec ? jb.n / 4 : jb.n;                 // threads per slab
            if (jb.slabs > 1) blk2 += (per + 255) / 256;
            blk1 += (int)(((long long)jb.slabs * per + 255) / 256);
            ++nj;
            ++k;
        }
        if (nj == 0) continue;
        b.n_jobs = nj;
        hipLaunchKernelGGL(colsum_multi_pass1_kernel, dim3((unsigned)blk1), dim3(256), 0, (hipStream_t)stream, b);
        if (blk2 > 0)
            hipLaunchKernelGGL(colsum_multi_pass2_kernel, dim3((unsigned)blk2), dim3(256), 0, (hipStream_t)stream, b);
    }
    return check_launch("colsum_multi");
}

// (one wavefront per partial row: 2,048 wavefronts = 2 per SIMD left the backward kernel - three dependent wave reductions per
// node behind its loads - at 2.3 TB/s; the partial rows are only (L + 2) * 16 floats)
int singa_so3_rmsnorm_nparts(int N) { return grid_for((N + 3) / 4, 8192); }      // wavefronts of four nodes each

int singa_so3_rmsnorm_fwd(const float* x, const float* weight, const float* bias, float* y, int N, int C, int lmax,
                          float eps, void* stream) {
    if (!x || !weight || !bias || !y) return fail(SINGA_E_NULL, "so3_rmsnorm_fwd: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "so3_rmsnorm: built for C = 16 sphere channels");
    if (N <= 0) return SINGA_OK;
    SINGA_DISPATCH_L(lmax, 2, hipLaunchKernelGGL((rmsnorm_fwd4_kernel<L_>), dim3(singa_so3_rmsnorm_nparts(N)), dim3(64), 0,
                                                 (hipStream_t)stream, x, weight, bias, y, N, eps));
    return check_launch("so3_rmsnorm_fwd");
}

int singa_so3_rmsnorm_bwd(const float* x, const float* weight, const float* gy, float* gx, float* gw_part,
                          float* gb_part, int N, int C, int lmax, float eps, void* stream) {
    if (!x || !weight || !gy || !gx || !gw_part || !gb_part) return fail(SINGA_E_NULL, "so3_rmsnorm_bwd: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "so3_rmsnorm: built for C = 16 sphere channels");
    if (N <= 0) return SINGA_OK;
    SINGA_DISPATCH_L(lmax, 2, hipLaunchKernelGGL((rmsnorm_bwd4_kernel<L_, false>), dim3(singa_so3_rmsnorm_nparts(N)), dim3(64),
                                                 0, (hipStream_t)stream, x, weight, gy, (const float*)nullptr, gx, gw_part, gb_part, N, eps));
    return check_launch("so3_rmsnorm_bwd");
}

int singa_so3_rmsnorm_bwd_add(const float* x, const float* weight, const float* gy, const float* g_add, float* gx, float* gw_part,
                              float* gb_part, int N, int C, int lmax, float eps, void* stream) {
    if (!x || !weight || !gy || !g_add || !gx || !gw_part || !gb_part) return fail(SINGA_E_NULL, "so3_rmsnorm_bwd_add: null pointer");
    if (C != 16) return fail(SINGA_E_SHAPE, "so3_rmsnorm: built for C = 16 sphere channels");
    if (N <= 0) return SINGA_OK;
    SINGA_DISPATCH_L(lmax, 2, hipLaunchKernelGGL((rmsnorm_bwd4_kernel<L_, true>), dim3(singa_so3_rmsnorm_nparts(N)), dim3(64),
                                                 0, (hipStream_t)stream, x, weight, gy, g_add, gx, gw_part, gb_part, N, eps));
    return check_launch("so3_rmsnorm_bwd_add");
}

}  // extern "C"
