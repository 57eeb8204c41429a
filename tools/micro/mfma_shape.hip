// Which fp32 MFMA shape delivers more FLOP/s on RANDOM operands (the chip lowers its clock under load, and by how much
// can depend on the shape -- MI355X_MICROARCH.md 'DVFS give-back' item 7)?  Bare MFMA loops, operands in registers,
// two waves per SIMD, same FLOPs per iteration: 8 x v_mfma_f32_32x32x2_f32 (64 cyc each) vs 16 x v_mfma_f32_16x16x4_f32
// (32 cyc each).  Operands are refreshed from a random buffer every iteration (one global load per operand).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int ZERO>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    const float* gp = g + (blockIdx.x * 256 + tid) % 4096;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = ZERO ? 0.f : gp[j * 4096]; b[j] = ZERO ? 0.f : gp[(j + 8) * 4096]; }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + u) & 7], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    } else {
        f32x4 acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7], b[(j + u) & 7], acc[j], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, int ZERO>
static void run(const char* name, const float* g, float* out) {
    const int iters = 200000;      // ~50 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(512);
    hipLaunchKernelGGL((k<SHAPE, ZERO>), grid, dim3(256), 0, 0, g, out, 2000);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, ZERO>), grid, dim3(256), 0, 0, g, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid.x * 4 * iters * 8 * 4096;
    printf("%-40s %8.2f ms  %6.1f TF/s (%.1f %% of 157.3)\n", name, best, flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 1e12 / 157.3 * 100);
    fflush(stdout);
}

int main() {
    const int n = 16 * 4096 + 4096;
    std::vector<float> h(n);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *g, *out;
    (void)hipMalloc(&g, n * 4); (void)hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 512 * 256 * 4);
    run<0, 1>("32x32x2  zero operands", g, out);
    run<1, 1>("16x16x4  zero operands", g, out);
    run<0, 0>("32x32x2  random operands in [-1,1)", g, out);
    run<1, 0>("16x16x4  random operands in [-1,1)", g, out);
    run<0, 0>("32x32x2  random (repeat)", g, out);
    run<1, 0>("16x16x4  random (repeat)", g, out);
    return 0;
}
