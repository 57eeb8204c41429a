// Shared helpers for the c2m_amd HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/c2m_geom.h"      // names of the geom[] entries (one definition for the library and the ctypes host)

#define C2M_API extern "C" __attribute__((visibility("default")))

// Every entry point returns hipError_t as int; kernels never allocate or synchronise.
#define C2M_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e__ = hipGetLastError();     \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// hipGetLastError() is sticky across unrelated runtime calls of the process (e.g. a probe that returned
// hipErrorNoDevice during framework start-up): clear it on entry so that our launch check reports OUR launch only.
#define C2M_ENTER() (void)hipGetLastError()

static inline int c2m_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// memory-bound kernels: cap the grid and grid-stride (cdna guide, Guideline 11)
static inline int c2m_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    const long cap = 256L * 16;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). Result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* smem /* >= 4 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

__device__ __forceinline__ double block_sum_256_d(double v, double* smem /* >= 4 doubles */) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    return smem[0] + smem[1] + smem[2] + smem[3];
}

// XCD-aware work order.  Workgroups of a launch are handed to the 8 XCDs round-robin by linear id and every XCD has its own
// L2, so tiles that re-read the same operand panel should be neighbours on ONE XCD: a 1-D launch of gx*gy*gz workgroups,
// XCD x takes the contiguous range [x*q + min(x, r), ...) of the work list (q, r = total / 8, total % 8: bijective for any
// total), and the work list runs over dimension `fast` first (0: x, 1: y), z last.
struct C2mBlock { unsigned x, y, z, nz; };
__device__ __forceinline__ unsigned c2m_xcd_item() {
    const unsigned L = blockIdx.x, total = gridDim.x;
    const unsigned q = total >> 3, r = total & 7u, x = L & 7u, j = L >> 3;
    return x * q + (x < r ? x : r) + j;
}
__device__ __forceinline__ C2mBlock c2m_xcd_block(unsigned gx, unsigned gy, int fast) {
    const unsigned w = c2m_xcd_item();
    C2mBlock b;
    if (fast == 1) { b.y = w % gy; const unsigned t = w / gy; b.x = t % gx; b.z = t / gx; }
    else           { b.x = w % gx; const unsigned t = w / gx; b.y = t % gy; b.z = t / gy; }
    b.nz = gridDim.x / (gx * gy);
    return b;
}

enum { C2M_ACT_NONE = 0, C2M_ACT_RELU = 1, C2M_ACT_LRELU = 2, C2M_ACT_SIGMOID = 3 };

__device__ __forceinline__ float c2m_act(float v, int act, float slope) {
    switch (act) {
        case C2M_ACT_RELU: return v > 0.f ? v : 0.f;
        case C2M_ACT_LRELU: return v > 0.f ? v : v * slope;
        case C2M_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// Zero-fill of a workspace region as a KERNEL (grid-stride, 16-byte stores where aligned).  hipMemsetAsync becomes a memset
// NODE when the stream is being captured into a HIP graph; a plain kernel node keeps every captured step kernel-only.
__global__ static void c2m_zero_words_kernel(unsigned* __restrict__ p, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0u;
}

static inline hipError_t c2m_zero_async(void* p, long bytes, hipStream_t s) {
    const long n = bytes / 4;                       // workspaces are word arrays
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(c2m_zero_words_kernel, dim3(c2m_grid(n, 256)), dim3(256), 0, s, (unsigned*)p, n);
    return hipGetLastError();
}
