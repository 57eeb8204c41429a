// Microbenchmark: what fraction of the fp32 MFMA peak does a SIMD sustain when VALU / LDS / VMEM instructions are issued
// between the MFMAs?  The loop body is inline asm (volatile statements keep their order), so the instruction pattern
// is exactly: 8 x { v_mfma_f32_32x32x2_f32 ; V x v_fma_f32 (independent registers) ; [1 ds_read_b128] ; [Gn global dword] }.
// Loads are consumed one iteration (512 MFMA cycles) later.  One workgroup = 4 waves (one per SIMD), grid = 256 CUs x WPS.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// V VALU per MFMA; L of the 8 MFMAs are followed by a ds_read_b128; G by a VMEM instruction of flavour KIND:
// 0 global_load_dword, 1 global_load_dwordx4, 2 buffer_load_dword offen, 3 buffer_load_dwordx4 offen, 4 buffer_load_dword ... lds,
// 10/11/12: per MFMA 2 ds_read_b32 + 2 SALU, VALU clumped after the 8th MFMA / after the LDS+SALU / before them
// 5 ds_write_b128 (not VMEM), 6 4 x s_add_u32 (SALU), 8 ds_read_b32 x2, 9 s_barrier after MFMA 0 and 4, 7 global_load_dword with saddr (SGPR base + 32-bit VGPR offset)
template <int V, int L, int G, int KIND>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ g, float* __restrict__ out, int iters) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 256) lds[i] = (float)i * 1e-9f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)j + lane;
    f32x4 ld[8];
    float gl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ld[j] = f32x4{0, 0, 0, 0}; gl[j] = 0.f; }
    const float* gp = g + (blockIdx.x * 256 + tid) % 4096;
    unsigned lds_addr = (unsigned)(size_t)lds + lane * 16;      // 16 B per lane, 1 KB per wave read: conflict-free
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned lds_addr4 = (unsigned)(size_t)lds + lane * 4;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rsrc;
    rsrc[0] = (int)(size_t)g; rsrc[1] = (int)((size_t)g >> 32); rsrc[2] = 1 << 20; rsrc[3] = 0x00020000;
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + (tid >> 6) * 2048);
    unsigned sacc = 0;
    const float* gbase = g;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (L | G) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (j < L) s += ld[j][j & 3]; if (j < G) s += gl[j]; }
            a += s * 1e-20f;
        }
        const float* gq = gp + ((it & 7) * 8) * 64;
        const float* gq4 = g + ((blockIdx.x * 256 + tid) % 1024) * 4 + (it & 3) * 4096;      // 16 B per lane, 1 KB per wave
        const unsigned boff = (unsigned)(((blockIdx.x * 256 + tid) % 4096) * 4 + (it & 7) * 2048);
        const unsigned boff4 = (unsigned)(((blockIdx.x * 256 + tid) % 1024) * 16 + (it & 3) * 16384);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
            if (KIND == 10 || KIND == 11) {
                asm volatile("ds_read_b32 %0, %3 offset:%4\n\tds_read_b32 %1, %3 offset:%5\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %2, %2, 1"
                             : "=v"(gl[j]), "=v"(ld[j][1]), "+s"(sacc) : "v"(lds_addr4), "n"(j * 512), "n"(j * 512 + 256) : "scc");
            }
            if (KIND == 10) {
                if (j == 7) {
#pragma unroll
                    for (int q = 0; q < 8 * V; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(c1), "v"(c2));
                }
                continue;
            }
            if (V >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c1), "v"(c2));
            if (V >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[1]) : "v"(c1), "v"(c2));
            if (V >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[2]) : "v"(c1), "v"(c2));
            if (V >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(c1), "v"(c2));
            if (V >= 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4]) : "v"(c1), "v"(c2));
            if (V >= 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[5]) : "v"(c1), "v"(c2));
            if (V >= 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[6]) : "v"(c1), "v"(c2));
            if (V >= 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[7]) : "v"(c1), "v"(c2));
            if (V >= 12) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c1), "v"(c2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[1]) : "v"(c1), "v"(c2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[2]) : "v"(c1), "v"(c2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(c1), "v"(c2));
            }
            if (j < L) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j]) : "v"(lds_addr), "n"(j * 1024));
            if (KIND == 12) {
                asm volatile("ds_read_b32 %0, %3 offset:%4\n\tds_read_b32 %1, %3 offset:%5\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %2, %2, 1"
                             : "=v"(gl[j]), "=v"(ld[j][1]), "+s"(sacc) : "v"(lds_addr4), "n"(j * 512), "n"(j * 512 + 256) : "scc");
            }
            if (j < G) {
                if (KIND == 0) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(gl[j]) : "v"(gq), "n"(j * 256));
                if (KIND == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[j]) : "v"(gq4), "n"(j * 1024 % 4096));
                if (KIND == 2) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(gl[j]) : "v"(boff), "s"(rsrc), "n"(j * 256));
                if (KIND == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(ld[j]) : "v"(boff4), "s"(rsrc), "n"(j * 1024 % 4096));
                if (KIND == 4) asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dword %0, %1, 0 offen offset:%3 lds" :: "v"(boff), "s"(rsrc), "s"(m0v + j * 256), "n"(j * 256) : "memory");
                if (KIND == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(lds_addr), "v"(ld[j]), "n"(j * 1024) : "memory");
                if (KIND == 6) asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(sacc) :: "scc");
                if (KIND == 8) asm volatile("ds_read_b32 %0, %2 offset:%3\n\tds_read_b32 %1, %2 offset:%4" : "=v"(gl[j]), "=v"(ld[j][1]) : "v"(lds_addr4), "n"(j * 512), "n"(j * 512 + 256));
                if (KIND == 9 && (j & 3) == 0) asm volatile("s_barrier" ::: "memory");
                if (KIND == 7) asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(gl[j]) : "v"(boff), "s"(gbase), "n"(j * 256));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += v[j] + gl[j] + ld[j][0]; for (int r = 0; r < 16; ++r) s += acc[j][r]; }
    out[blockIdx.x * 256 + tid] = s + a + b + (float)sacc;
}

template <int V, int L, int G, int KIND = 0>
static void run(const char* name, int wps, const float* g, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    dim3 grid(256 * wps);
    hipLaunchKernelGGL((k<V, L, G, KIND>), grid, dim3(256), 0, 0, g, out, 200);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<V, L, G, KIND>), grid, dim3(256), 0, 0, g, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid.x * 4 * iters * 8;
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %7.2f ms  %6.1f TF/s  (%5.1f %% of 157.3)\n", name, wps, ms, tf, tf / 157.3 * 100);
    fflush(stdout);
}

static int iters_override = 0;
int main() {
    float *g, *out;
    (void)hipMalloc(&g, 1 << 20);
    (void)hipMemset(g, 0, 1 << 20);
    (void)hipMalloc(&out, 256 * 2 * 256 * 4);
    for (int wps = 1; wps <= 2; ++wps) {
        if (iters_override) {}
        run<0, 0, 0>("pure MFMA", wps, g, out);
        run<1, 0, 0>("+1 VALU / MFMA", wps, g, out);
        run<2, 0, 0>("+2 VALU / MFMA", wps, g, out);
        run<4, 0, 0>("+4 VALU / MFMA", wps, g, out);
        run<8, 0, 0>("+8 VALU / MFMA", wps, g, out);
        run<12, 0, 0>("+12 VALU / MFMA", wps, g, out);
        run<0, 2, 0>("+2 ds_read_b128 / 8 MFMA", wps, g, out);
        run<0, 4, 0>("+4 ds_read_b128 / 8 MFMA", wps, g, out);
        run<0, 8, 0>("+8 ds_read_b128 / 8 MFMA", wps, g, out);
        run<0, 0, 2>("+2 global dword / 8 MFMA", wps, g, out);
        run<0, 0, 4>("+4 global dword / 8 MFMA", wps, g, out);
        run<0, 0, 8>("+8 global dword / 8 MFMA", wps, g, out);
        run<0, 0, 2, 1>("+2 global dwordx4 / 8", wps, g, out);
        run<0, 0, 4, 1>("+4 global dwordx4 / 8", wps, g, out);
        run<0, 0, 8, 1>("+8 global dwordx4 / 8", wps, g, out);
        run<0, 0, 4, 2>("+4 buffer dword / 8", wps, g, out);
        run<0, 0, 8, 2>("+8 buffer dword / 8", wps, g, out);
        run<0, 0, 2, 3>("+2 buffer dwordx4 / 8", wps, g, out);
        run<0, 0, 4, 3>("+4 buffer dwordx4 / 8", wps, g, out);
        run<0, 0, 8, 3>("+8 buffer dwordx4 / 8", wps, g, out);
        run<0, 0, 4, 4>("+4 buffer dword lds / 8", wps, g, out);
        run<0, 0, 8, 4>("+8 buffer dword lds / 8", wps, g, out);
        run<0, 0, 4, 5>("+4 ds_write_b128 / 8", wps, g, out);
        run<0, 0, 8, 5>("+8 ds_write_b128 / 8", wps, g, out);
        run<0, 0, 4, 7>("+4 global saddr dword / 8", wps, g, out);
        run<0, 0, 8, 7>("+8 global saddr dword / 8", wps, g, out);
        run<0, 0, 4, 6>("+2 SALU / MFMA", wps, g, out);
        run<0, 0, 8, 6>("+4 SALU / MFMA", wps, g, out);
        run<0, 0, 4, 8>("+1 ds_read_b32 / MFMA", wps, g, out);
        run<0, 0, 8, 8>("+2 ds_read_b32 / MFMA", wps, g, out);
        run<0, 0, 8, 9>("+2 s_barrier / 8 MFMA", wps, g, out);
        run<1, 0, 8, 8>("+1 VALU +2 ds_read_b32 / MFMA", wps, g, out);
        run<1, 0, 8, 6>("+1 VALU +4 SALU / MFMA", wps, g, out);
        run<1, 0, 0, 10>("1 VALU clumped, LDS+SALU", wps, g, out);
        run<1, 0, 0, 11>("1 VALU after LDS+SALU", wps, g, out);
        run<1, 0, 0, 12>("1 VALU before LDS+SALU", wps, g, out);
        run<2, 0, 0, 10>("2 VALU clumped, LDS+SALU", wps, g, out);
        run<2, 0, 0, 11>("2 VALU after LDS+SALU", wps, g, out);
        run<2, 0, 0, 12>("2 VALU before LDS+SALU", wps, g, out);
        run<4, 0, 0, 10>("4 VALU clumped, LDS+SALU", wps, g, out);
        run<4, 0, 0, 11>("4 VALU after LDS+SALU", wps, g, out);
        run<4, 0, 0, 12>("4 VALU before LDS+SALU", wps, g, out);
        run<8, 0, 0, 10>("8 VALU clumped, LDS+SALU", wps, g, out);
        run<8, 0, 0, 11>("8 VALU after LDS+SALU", wps, g, out);
        run<2, 4, 8>("+2 VALU/MFMA, 4 LDS, 8 glob", wps, g, out);
        run<4, 8, 8>("+4 VALU/MFMA, 8 LDS, 8 glob", wps, g, out);
    }
    return 0;
}
