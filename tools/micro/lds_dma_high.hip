// Do LDS-DMA destinations above 64 KB work on gfx950?  (conv_wino4.hip, round 3: "only destinations below 64 KB behaved".)
// One workgroup owns 160 KB of LDS; every wave DMAs 1 KB (64 x 16 B) from a known pattern to a destination at `base`, then the
// LDS is read back.  Forms: buffer_load_dwordx4 ... offen lds with M0 = the full byte address; the same with M0 = base & 0xffff and
// the rest in the instruction's offset field (max 4095: not usable for this); __builtin_amdgcn_global_load_lds (compiler-set M0).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_high.hip -o tools/micro/lds_dma_high && tools/micro/lds_dma_high
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDS_BYTES = 160 * 1024;

template <int FORM>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ src, uint4* __restrict__ out, unsigned base, int guard) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x; const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < LDS_BYTES / 16; i += 256) lds[i] = make_uint4(0xdeadbeefu, i, 0, 0);
    __syncthreads();
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) uint4*)&lds[0] + base + (unsigned)wave * 1024u);
    if (FORM == 0) {
        const unsigned long a = (unsigned long)src;
        const u32x4 rs = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 1u << 20, 0x00020000u};
        const unsigned vo = (unsigned)(tid * 16);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dst), "v"(vo), "s"(rs) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        __builtin_amdgcn_global_load_lds(src + tid, (__attribute__((address_space(3))) void*)(lds + (base + wave * 1024) / 16), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = tid; i < LDS_BYTES / 16; i += 256) out[i] = lds[i];
}

int main() {
    uint4 *src, *out;
    hipMalloc(&src, 1 << 20); hipMalloc(&out, LDS_BYTES);
    std::vector<uint4> h(65536), got(LDS_BYTES / 16);
    for (int i = 0; i < 65536; ++i) h[i] = make_uint4(0x5000000u + i, ~i, i * 3, 7);
    hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const unsigned bases[] = {0, 32 * 1024, 60 * 1024, 64 * 1024, 72 * 1024, 100 * 1024, 128 * 1024, 156 * 1024};
    for (int form = 0; form < 2; ++form)
        for (unsigned base : bases) {
            if (form == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), LDS_BYTES, 0, src, out, base, 0);
            else hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), LDS_BYTES, 0, src, out, base, 0);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(got.data(), out, LDS_BYTES, hipMemcpyDeviceToHost);
            int ok = 0, stray = 0;
            for (int i = 0; i < LDS_BYTES / 16; ++i) {
                const int rel = i - (int)(base / 16);
                const bool want_data = rel >= 0 && rel < 256;
                if (want_data) ok += got[i].x == h[rel].x && got[i].y == h[rel].y && got[i].z == h[rel].z && got[i].w == h[rel].w;
                else stray += !(got[i].x == 0xdeadbeefu && got[i].y == (unsigned)i);
            }
            printf("%s base %6u B: rc %d, %3d / 256 units landed at the destination, %d units elsewhere changed\n",
                   form == 0 ? "buffer_load_dwordx4 lds (M0 = address)" : "__builtin_amdgcn_global_load_lds      ", base, (int)e, ok, stray);
        }
    return 0;
}
