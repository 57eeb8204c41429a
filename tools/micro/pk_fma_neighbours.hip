// Round-4 follow-up of pk_fma_vs_bf16_mfma.hip: the wrong sums of conv_thin_wgrad_rows_kernel (SLP build: v_pk_fma_f32) appeared only
// next to conv_igemm_kernel<..., bf16>.  A bare MFMA loop as the neighbour did not reproduce it; this one tries the OTHER
// instruction classes that only the bf16 gather kernel issues, one at a time, next to three victims:
//   victim 0: dependent v_pk_fma_f32 chains (op_sel_hi [1,1,1] / [1,0,1] / [0,1,1], the three forms of the failing kernel)
//   victim 1: the same with ds_bpermute_b32 (the kernel's __shfl_up / __shfl_down) feeding one operand
//   victim 2: scalar v_fma_f32 with ds_bpermute (the no-SLP build's form)
//   victim 3: v_pk_fma_f32 whose operand pair is re-loaded from memory every iteration (global_load_dwordx2)
// neighbours: none | bf16 MFMA 32x32x16 | v_cvt_pk_bf16_f32 | v_perm_b32 | ds_write_b16 + ds_read_u16 | buffer_load_ushort gathers | all
//             | ds_write_b128 + ds_read_b128 | buffer_store_short | bf16 MFMA on AGPR accumulators + v_accvgpr moves | those three
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pk_fma_neighbours.hip -o tools/micro/pk_nb && tools/micro/pk_nb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim_mem(float* out, const f32x2* __restrict__ src, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[8];
    f32x2 a = {1.0f + (t & 7) * 0.125f, 1.0f + (t & 3) * 0.25f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x2{(float)k, (float)(k + 1)};
    for (int i = 0; i < iters; ++i) {
        const f32x2 bb = src[(t * 17 + i * 64) & 0xfffff];          // every element is {0.5, 0.25}
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k % 3 == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(bb));
            else if (k % 3 == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(a), "v"(bb));
            else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[k]) : "v"(a), "v"(bb));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
    out[t] = s;
}

template <int KIND>
__global__ __launch_bounds__(256) void victim(float* out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x2 acc[8];
    f32x2 a = {1.0f + (t & 7) * 0.125f, 1.0f + (t & 3) * 0.25f}, b = {0.5f, 0.25f};
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x2{(float)k, (float)(k + 1)};
    const int self = (threadIdx.x & 63) * 4;
    for (int i = 0; i < iters; ++i) {
        f32x2 bb = b;
        if (KIND >= 1) {      // a value that travels through the LDS crossbar and comes back unchanged (own lane)
            bb.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(self, __builtin_bit_cast(int, b.x)));
            bb.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(self, __builtin_bit_cast(int, b.y)));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (KIND <= 1) {
                if (k % 3 == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(bb));
                else if (k % 3 == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(a), "v"(bb));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[k]) : "v"(a), "v"(bb));
            } else {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].x) : "v"(a.x), "v"(bb.x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k].y) : "v"(a.y), "v"(bb.y));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
    out[t] = s;
}

template <int KIND>
__global__ __launch_bounds__(256) void neighbour(float* out, const unsigned short* src, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    f32x16 acc = {};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(1.0f + threadIdx.x * 0.001f); b[k] = (__bf16)0.5f; }
    float f0 = 1.0f + threadIdx.x, f1 = 2.0f;
    unsigned u = threadIdx.x * 0x01010101u, w = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, 1 << 20, 0x00020000);
    for (int i = 0; i < iters; ++i) {
        if (KIND == 1 || KIND == 6) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (KIND == 2 || KIND == 6) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(f0), "v"(f1)); f0 += __builtin_bit_cast(float, (w & 0xffffu) << 16) * 1e-9f; }
        if (KIND == 3 || KIND == 6) { asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(w) : "v"(u), "v"(u + i), "v"(0x07060302u)); u ^= w; }
        if (KIND == 4 || KIND == 6) { lds[(threadIdx.x * 7 + i) & 4095] = (unsigned short)u; u += lds[(threadIdx.x * 13 + i) & 4095]; }
        if (KIND == 5 || KIND == 6) u += __builtin_amdgcn_raw_buffer_load_b16(rs, ((threadIdx.x * 97 + i * 31) & 0x7ffff) * 2, 0, 0);
        if (KIND == 7 || KIND == 10) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4* l4 = reinterpret_cast<u4*>(lds);
            l4[(threadIdx.x + i) & 511] = u4{u, u + 1, u + 2, u + 3};
            const u4 r = l4[(threadIdx.x * 5 + i) & 511];
            u += r.x ^ r.y ^ r.z ^ r.w;
        }
        if (KIND == 8 || KIND == 10) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)u, rs, ((blockIdx.x * 256 + threadIdx.x) & 0x7ffff) * 2, 0, 0);
        if (KIND == 9 || KIND == 10) {
            asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\ts_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], %1, %2, a[0:15]\n\ts_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, a3"
                         : "+v"(f1) : "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[7] + f0 + (float)u;
}

int main() {
    const int VB = 8192, AB = 1024, iters = 20000;
    float *vo, *ao; unsigned short* src;
    hipMalloc(&vo, VB * 256 * 4); hipMalloc(&ao, AB * 256 * 4); hipMalloc(&src, 1 << 20); hipMemset(src, 1, 1 << 20);
    f32x2* vsrc; hipMalloc(&vsrc, (1 << 20) * 8);
    { std::vector<f32x2> h(1 << 20, f32x2{0.5f, 0.25f}); hipMemcpy(vsrc, h.data(), (1 << 20) * 8, hipMemcpyHostToDevice); }
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    std::vector<float> ref(VB * 256), got(VB * 256);
    auto launch_v = [&](int v) {
        if (v == 0) hipLaunchKernelGGL(victim<0>, dim3(VB), dim3(256), 0, s1, vo, iters);
        if (v == 1) hipLaunchKernelGGL(victim<1>, dim3(VB), dim3(256), 0, s1, vo, iters / 4);
        if (v == 2) hipLaunchKernelGGL(victim<2>, dim3(VB), dim3(256), 0, s1, vo, iters / 4);
        if (v == 3) hipLaunchKernelGGL(victim_mem, dim3(VB), dim3(256), 0, s1, vo, vsrc, iters / 4);
    };
    auto launch_n = [&](int n) {
        const int it = 3000;
        if (n == 1) hipLaunchKernelGGL(neighbour<1>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 2) hipLaunchKernelGGL(neighbour<2>, dim3(AB), dim3(256), 0, s2, ao, src, it * 4);
        if (n == 3) hipLaunchKernelGGL(neighbour<3>, dim3(AB), dim3(256), 0, s2, ao, src, it * 4);
        if (n == 4) hipLaunchKernelGGL(neighbour<4>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 5) hipLaunchKernelGGL(neighbour<5>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 6) hipLaunchKernelGGL(neighbour<6>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 7) hipLaunchKernelGGL(neighbour<7>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 8) hipLaunchKernelGGL(neighbour<8>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 9) hipLaunchKernelGGL(neighbour<9>, dim3(AB), dim3(256), 0, s2, ao, src, it);
        if (n == 10) hipLaunchKernelGGL(neighbour<10>, dim3(AB), dim3(256), 0, s2, ao, src, it);
    };
    const char* nn[] = {"alone", "bf16 MFMA 32x32x16", "v_cvt_pk_bf16_f32", "v_perm_b32", "ds_write_b16/ds_read_u16", "buffer_load_ushort", "all of them",
                        "ds_write_b128/ds_read_b128", "buffer_store_short", "bf16 MFMA on AGPRs", "b128 + short stores + AGPR"};
    const char* vn[] = {"v_pk_fma_f32 (3 op_sel forms)", "v_pk_fma_f32 + ds_bpermute", "v_fma_f32 + ds_bpermute", "v_pk_fma_f32 + operand loads"};
    long total = 0;
    for (int v = 0; v < 4; ++v) {
        hipDeviceSynchronize(); launch_v(v); hipDeviceSynchronize();
        hipMemcpy(ref.data(), vo, VB * 256 * 4, hipMemcpyDeviceToHost);
        for (int n = 0; n < 11; ++n) {
            hipDeviceSynchronize();
            launch_v(v);                                  // the victim starts first, neighbours arrive while it runs
            for (int k = 0; k < 8; ++k) launch_n(n);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), vo, VB * 256 * 4, hipMemcpyDeviceToHost);
            long bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
            total += bad;
            printf("%-32s next to %-26s: %ld of %zu lanes differ\n", vn[v], nn[n], bad, got.size());
        }
    }
    printf("total differing: %ld\n", total);
    return 0;
}
