// Does a wave's v_pk_fma_f32 return wrong results while ANOTHER wave on the same SIMD runs v_mfma_f32_32x32x16_bf16?
// (round 3: conv_thin_wgrad_rows_kernel on a side stream next to the bf16 gather kernel produced wrong partial sums in exactly
// the accumulator pairs the compiler had packed into v_pk_fma_f32; a -fno-slp-vectorize build did not.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_fma_vs_bf16_mfma.hip -o /tmp/pk_test && /tmp/pk_test
// Victim: every lane runs ITER dependent v_pk_fma_f32 (or v_fma_f32 pairs) on values whose exact result is known.
// Aggressor (other stream): bf16 / fp32 MFMA loops, or nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool PK>
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[8];
    f32x2 a = {1.0f + (t & 7) * 0.125f, 1.0f + (t & 3) * 0.25f}, b = {0.5f, 0.25f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x2{(float)k, (float)(k + 1)};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].x) : "v"(a.x), "v"(b.x));
                   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].y) : "v"(a.y), "v"(b.y)); }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
    out[t] = s;
}

template <int KIND>   // 0: bf16 32x32x16, 1: fp32 32x32x2
__global__ __launch_bounds__(256) void aggressor(float* out, int iters) {
    f32x16 acc = {};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(1.0f + threadIdx.x * 0.001f); b[k] = (__bf16)0.5f; }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f + i, 0.5f, acc, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[7];
}

int main() {
    const int VB = 8192, AB = 4096, iters = 20000;
    float *vo, *ao;
    hipMalloc(&vo, VB * 256 * 4); hipMalloc(&ao, AB * 256 * 4);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> ref(VB * 256), got(VB * 256);
    auto run = [&](bool pk, int aggr) {     // aggr: -1 none, 0 bf16 MFMA, 1 fp32 MFMA
        hipDeviceSynchronize();
        // the victim starts FIRST and the neighbour arrives while it runs (the order of the failing case: the weight gradient
        // was launched on the side stream, then the data gradient on the main one); a few neighbour launches in a row give
        // several onsets
        if (pk) hipLaunchKernelGGL(victim<true>, dim3(VB), dim3(256), 0, s1, vo, iters);
        else hipLaunchKernelGGL(victim<false>, dim3(VB), dim3(256), 0, s1, vo, iters);
        for (int k = 0; k < 6; ++k) {
            if (aggr == 0) hipLaunchKernelGGL(aggressor<0>, dim3(AB / 4), dim3(256), 0, s2, ao, 8000);
            if (aggr == 1) hipLaunchKernelGGL(aggressor<1>, dim3(AB / 4), dim3(256), 0, s2, ao, 4000);
        }
        hipDeviceSynchronize();
        hipMemcpy(got.data(), vo, VB * 256 * 4, hipMemcpyDeviceToHost);
    };
    run(false, -1); ref = got;
    const char* names[] = {"alone", "next to bf16 MFMA 32x32x16", "next to fp32 MFMA 32x32x2"};
    for (int pk = 0; pk < 2; ++pk)
        for (int aggr = -1; aggr < 2; ++aggr)
            for (int rep = 0; rep < 2; ++rep) {
                run(pk, aggr);
                long bad = 0;
                for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
                printf("%-14s %-28s rep %d: %ld of %zu lanes differ from the scalar-alone result\n", pk ? "v_pk_fma_f32" : "v_fma_f32 x2",
                       names[aggr + 1], rep, bad, got.size());
            }
    return 0;
}
