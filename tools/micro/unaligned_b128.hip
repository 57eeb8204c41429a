// Does a 16-byte buffer load from a 2-byte-aligned address return the right bytes on gfx950 (unaligned access mode)?
// And what does it cost against the aligned form?  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/micro/unaligned_b128.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* __restrict__ x, unsigned bytes, int shift, int iters, unsigned* __restrict__ out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x), 0, bytes, 0x00020000);
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const unsigned vo = ((tid * 8u + (unsigned)it * 8u * gridDim.x * blockDim.x) % ((bytes / 2 - 16) / 8 * 8) + (unsigned)shift) * 2u;
        const u32x4 v = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
        acc ^= v;
    }
    out[tid * 4 + 0] = acc.x; out[tid * 4 + 1] = acc.y; out[tid * 4 + 2] = acc.z; out[tid * 4 + 3] = acc.w;
}
int main() {
    const size_t n = 64 << 20;                                   // 128 MB of ushort
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(i * 2654435761u >> 7);
    unsigned short* d; unsigned* o;
    const int grid = 4096, block = 256;
    hipMalloc(&d, n * 2); hipMalloc(&o, (size_t)grid * block * 16);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<unsigned> ho((size_t)grid * block * 4);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, (unsigned)(n * 2), shift, 1, o);
        hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (size_t t = 0; t < (size_t)grid * block; ++t) {
            const size_t e = (t * 8 % ((n - 16) / 8 * 8)) + shift;
            for (int q = 0; q < 4; ++q) {
                const unsigned want = (unsigned)h[e + 2 * q] | ((unsigned)h[e + 2 * q + 1] << 16);
                bad += ho[t * 4 + q] != want;
            }
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, (unsigned)(n * 2), shift, 64, o);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("shift %d elements (%d bytes): %ld wrong dwords; 64 x 16 MB: %.3f ms = %.0f GB/s\n", shift, shift * 2, bad, ms,
               64.0 * grid * block * 16 / ms / 1e6);
    }
    return 0;
}
