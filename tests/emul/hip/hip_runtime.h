// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets tests/test_kernels_emul.py compile singa_amd/csrc/singa_hip.hip with
// g++ (-I tests/emul) and run each kernel's exact source on the CPU, thread by thread, so that index algebra and
// bounds can be checked (and sanitised) in the GPU-less build container.  Never part of the product: the shipped
// library is built by hipcc for gfx950 only.  Kernels that use cross-lane intrinsics (__shfl_xor) cannot be emulated
// sequentially and abort here; they are covered by the -m gpu tests.
#pragma once
#define SINGA_EMUL 1
#define SINGA_GEMM_PIPE 0
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// sequential threads have no lanes: a lane broadcast of the Wigner record becomes a plain memory read
#define SINGA_LANE_BCAST(reg, lane, ptr, idx) ((ptr)[(idx)])

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define HIP_SYMBOL(x) x

struct float4 {
    float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct float2 {
    float x, y;
};
struct float3 {
    float x, y, z;
};
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
// sequential threads: an atomic is a plain update, and every thread is the leader of its own one-lane wave
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
#define SINGA_WAVE_MIN_MAX(lo, hi) ((void)0)
#define SINGA_WAVE_LEADER true

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emul"; }
template <class T>
static inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n, size_t off, int) {
    memcpy((char*)&sym + off, src, n);
    return hipSuccess;
}
#define __expf(x) expf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
#define SINGA_RCP(x) (1.0f / (x))
#define SINGA_KEEP_VGPR(x) ((void)0)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
enum { hipErrorInvalidValue = 1 };
template <class K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, int) {
    *n = 0;
    return hipSuccess;
}
static inline long long wall_clock64() { return 0; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __shfl_xor(int, int, int) {
    fprintf(stderr, "emul: cross-lane kernel cannot be emulated sequentially\n");
    abort();
}
static inline float __shfl_xor(float, int, int) {
    fprintf(stderr, "emul: cross-lane kernel cannot be emulated sequentially\n");
    abort();
}

#define __shared__ static
static inline void __syncthreads() {
    fprintf(stderr, "emul: workgroup-synchronised kernel cannot be emulated sequentially\n");
    abort();
}

#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_sched_barrier(a) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_waitcnt(a) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
struct floatx16_emul { float v[16]; float& operator[](int i) { return v[i]; } };
#define SINGA_FLOATX16 floatx16_emul
static inline floatx16_emul __builtin_amdgcn_mfma_f32_32x32x2f32(float, float, floatx16_emul c, int, int, int) {
    fprintf(stderr, "emul: matrix-core kernel cannot be emulated sequentially\n");
    abort();
    return c;
}

struct floatx4_emul { float v[4]; float& operator[](int i) { return v[i]; } };
#define SINGA_FLOATX4 floatx4_emul
// two packed floats (the S2 activation's two-channel kernel): GCC's vector extension has the element-wise arithmetic, the
// scalar broadcast and the subscripts the kernel uses
typedef float v2f_emul __attribute__((vector_size(8)));
#define SINGA_V2F v2f_emul
static inline floatx4_emul __builtin_amdgcn_mfma_f32_16x16x4f32(float, float, floatx4_emul c, int, int, int) {
    fprintf(stderr, "emul: matrix-core kernel cannot be emulated sequentially\n");
    abort();
    return c;
}

struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 4; return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* t, hipEvent_t, hipEvent_t) { *t = 0.f; return hipSuccess; }
#define hipExtLaunchKernelGGL(kern, g, b, shmem, stream, ea, eb, fl, ...) hipLaunchKernelGGL(kern, g, b, shmem, stream, __VA_ARGS__)

#define hipLaunchKernelGGL(kern, g, b, shmem, stream, ...)                  \
    do {                                                                    \
        dim3 _g = (g), _b = (b);                                            \
        gridDim = _g;                                                       \
        blockDim = _b;                                                      \
        for (unsigned _by = 0; _by < _g.y; ++_by)                           \
            for (unsigned _bx = 0; _bx < _g.x; ++_bx)                       \
                for (unsigned _tx = 0; _tx < _b.x; ++_tx) {                 \
                    blockIdx.x = _bx;                                       \
                    blockIdx.y = _by;                                       \
                    threadIdx.x = _tx;                                      \
                    kern(__VA_ARGS__);                                      \
                }                                                           \
    } while (0)
