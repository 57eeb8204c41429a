// optim.hip -- multi-tensor Adam for the four optimizers of the C2M step (reference: src/modules/model.py:54-99,
// torch.optim.Adam(betas=(0.5, 0.999), eps=1e-7), no weight decay / amsgrad; stepped in src/trainer/trainer.py:155-165).
//
// One launch updates EVERY tensor of a parameter group: a device table holds the four pointers (param, grad, exp_avg,
// exp_avg_sq) and the element count per tensor; a block map assigns each workgroup one 4096-element chunk of one
// tensor.  Pure HBM streaming: 16 B read + 12 B written per element, ~100 M elements per step.
// Arithmetic mirrors ATen's single-tensor Adam (_single_tensor_adam) operation by operation, in fp32:
//   m = lerp(m, g, 1-beta1);  v = fma((1-beta2)*g, g, v*beta2);  denom = sqrt(v)/sqrt(bc2) + eps;  p += (-step_size * m)/denom
// with bias corrections and step_size = lr/bc1 computed on the host in double, as torch does.
#include "common.h"

#define ADAM_CHUNK 4096

struct AdamP { float w1, beta2, w2, eps, step_size, bc2_sqrt; };   // w1 = 1-beta1, w2 = 1-beta2 (rounded from double)

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamP a) {
    const float w = a.w1;
    // at::lerp: weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    m = (w < 0.5f) ? m + w * (g - m) : g - (g - m) * (1.0f - w);
    v = fmaf(a.w2 * g, g, v * a.beta2);          // ATen addcmul: one fused multiply-add
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p + ((-a.step_size) * m) / denom;      // ATen addcdiv: self + alpha * t1 / t2, left to right
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const long* __restrict__ table, const long* __restrict__ sizes,
                                                         const int2* __restrict__ blockmap, int ntensors, const AdamP a) {
    const int2 bm = blockmap[blockIdx.x];
    const int t = bm.x;
    const long base = (long)bm.y * ADAM_CHUNK;
    const long n = sizes[t];
    float* __restrict__ p = reinterpret_cast<float*>(table[t]);
    const float* __restrict__ g = reinterpret_cast<const float*>(table[ntensors + t]);
    float* __restrict__ m = reinterpret_cast<float*>(table[2 * ntensors + t]);
    float* __restrict__ v = reinterpret_cast<float*>(table[3 * ntensors + t]);
    const long end = base + ADAM_CHUNK < n ? base + ADAM_CHUNK : n;
    const bool vec = (((table[t] | table[ntensors + t] | table[2 * ntensors + t] | table[3 * ntensors + t]) & 15) == 0) &&
                     end - base == ADAM_CHUNK;
    if (vec) {
#pragma unroll
        for (int it = 0; it < ADAM_CHUNK / (256 * 4); ++it) {
            const long i = base + (it * 256 + threadIdx.x) * 4;
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            adam_elem(pp.x, gg.x, mm.x, vv.x, a);
            adam_elem(pp.y, gg.y, mm.y, vv.y, a);
            adam_elem(pp.z, gg.z, mm.z, vv.z, a);
            adam_elem(pp.w, gg.w, mm.w, vv.w, a);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam_elem(pp, g[i], mm, vv, a);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

// table: device int64 [4][ntensors] = {param, grad, exp_avg, exp_avg_sq} pointers; sizes: device int64 [ntensors];
// blockmap: device int32 [nblocks][2] = (tensor index, chunk index) with chunks of c2m_adam_chunk() elements.
C2M_API int c2m_adam_chunk(void) { return ADAM_CHUNK; }

C2M_API int c2m_adam_step(const int64_t* table, const int64_t* sizes, const int32_t* blockmap, int ntensors, int nblocks,
                          double beta1, double beta2, double eps, double step_size, double bias_correction2_sqrt,
                          void* stream) {
    C2M_ENTER();
    if (ntensors <= 0 || nblocks <= 0) return 0;
    // 1 - beta is formed in double and then rounded, as the Python scalars ATen receives are
    AdamP a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)step_size, (float)bias_correction2_sqrt};
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long*>(table), reinterpret_cast<const long*>(sizes),
                       reinterpret_cast<const int2*>(blockmap), ntensors, a);
    return (int)hipGetLastError();
}
