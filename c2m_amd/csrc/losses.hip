// Per-pixel loss reductions (HBM-bound), gfx950.  Replaces the ATen chains of src/losses/losses.py:
//   c2m_l1_mean_fwd/_bwd   L1MaskedLoss (losses.py:180-189): mean |a*m - b*m| with the mask broadcast over channels;
//                          also the VGG feature L1 (losses.py:60-65) and feature matching (model.py:118-121)
//   c2m_ssim_fwd/_bwd      SSIMLoss (losses.py:152-177): 5x avg_pool2d(3,1) + ~15 elementwise ops fused in one pass
// Reductions are two-stage and deterministic: per-block partial sums (fp32 in-thread, fp64 across the block) and a
// fixed-order final sum in fp64 by a single wave; no float atomics.
#include "common.h"
#include "dtype.h"

#define RED_BLOCKS 1024

__global__ __launch_bounds__(256) void final_sum_kernel(const double* __restrict__ part, int n, double scale,
                                                         float* __restrict__ out) {
    __shared__ double sm[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_256_d(s, sm);
    if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

// a, b: [outer][C][inner]; mask (optional): [outer][inner] broadcast over C
template <class T>
__global__ __launch_bounds__(256) void l1_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          const float* __restrict__ mask, long total, int C, long inner,
                                                          double* __restrict__ part) {
    __shared__ double sm[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float va = c2m_ld(a, i), vb = c2m_ld(b, i);
        if (mask) {
            const long o = i / (inner * C), r = i % inner;
            const float m = mask[o * inner + r];
            va *= m; vb *= m;
        }
        s += fabsf(va - vb);
    }
    const double bs = block_sum_256_d((double)s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = bs;
}

// workspace: RED_BLOCKS doubles
C2M_API int c2m_l1_mean_fwd(const void* a, const void* b, const float* mask, float* out, long total, int C, long inner,
                            void* workspace, int dt, void* stream) {
    C2M_ENTER();
    if (total <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int blocks = c2m_grid(total, 256); if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(l1_partial_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)a, (const T*)b, mask,
                                           total, C, inner, (double*)workspace););
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, blocks, 1.0 / (double)total,
                       out);
    return (int)hipGetLastError();
}

// ga = gscale * sign(a*m - b*m) * m / total ; gb = -ga  (either may be null); gscale is a device scalar
template <class T>
__global__ void l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ mask,
                              const float* __restrict__ gscale, T* __restrict__ ga, T* __restrict__ gb,
                              long total, int C, long inner) {
    const float g = gscale[0] / (float)total;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float va = c2m_ld(a, i), vb = c2m_ld(b, i), m = 1.f;
        if (mask) {
            const long o = i / (inner * C), r = i % inner;
            m = mask[o * inner + r];
            va *= m; vb *= m;
        }
        const float d = va - vb;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float v = g * sgn * m;
        if (ga) c2m_st(ga, i, v);
        if (gb) c2m_st(gb, i, -v);
    }
}

C2M_API int c2m_l1_mean_bwd(const void* a, const void* b, const float* mask, const float* gscale, void* ga, void* gb,
                            long total, int C, long inner, int dt, void* stream) {
    C2M_ENTER();
    if (total <= 0) return 0;
    C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(l1_bwd_kernel<T>, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                                           (const T*)a, (const T*)b, mask, gscale, (T*)ga, (T*)gb, total, C, inner););
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- VGG tap backward
// A perceptual-loss tap (losses.py:60-65 on vgg.py:92-137's relu{1..5}_1 outputs): y = relu(conv(x)) feeds the next VGG conv
// AND mean|y - t|.  Autograd ran three element-wise passes over the tap tensor for that -- the L1 gradient, the sum with the
// next conv's data gradient, the ReLU mask -- 9 tensor passes; this is the same arithmetic in one (4 passes):
//   out = (gy + gl/total * sign(y - t)) * (y > 0)        gy: gradient from the next conv (or NULL), gl: device scalar
// bit-identical to the three-kernel chain in fp32 (same products, one commutative add, a 0/1 mask).
template <class T>
__global__ void relu_tap_bwd_kernel(const T* __restrict__ y, const T* __restrict__ t, const T* __restrict__ gy,
                                    const float* __restrict__ gl, T* __restrict__ out, long total4, long total) {
    const float g = gl[0] / (float)total;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const float4 a = c2m_ld4(y + 4 * i), b = c2m_ld4(t + 4 * i);
        const float4 u = gy ? c2m_ld4(gy + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 o;
#define C2M_TAP(f) { const float d = a.f - b.f; const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); \
                     const float v = g * sgn; o.f = a.f > 0.f ? u.f + v : 0.f; }
        C2M_TAP(x) C2M_TAP(y) C2M_TAP(z) C2M_TAP(w)
#undef C2M_TAP
        c2m_st4(out + 4 * i, o);
    }
}

template <class T>
__global__ void relu_tap_bwd_tail_kernel(const T* __restrict__ y, const T* __restrict__ t, const T* __restrict__ gy,
                                         const float* __restrict__ gl, T* __restrict__ out, long beg, long total) {
    const float g = gl[0] / (float)total;
    for (long i = beg + blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const float a = c2m_ld(y, i), d = a - c2m_ld(t, i);
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        c2m_st(out, i, a > 0.f ? (gy ? c2m_ld(gy, i) : 0.f) + g * sgn : 0.f);
    }
}

C2M_API int c2m_relu_tap_bwd(const void* y, const void* t, const void* gy, const float* gl, void* out, long total, int dt,
                             void* stream) {
    C2M_ENTER();
    if (total <= 0) return 0;
    C2M_DISPATCH_DT(dt,
        const bool vec = !(((uintptr_t)y | (uintptr_t)t | (uintptr_t)gy | (uintptr_t)out) & C2mVec4<T>::mask);
        const long n4 = vec ? total / 4 : 0;
        if (n4) hipLaunchKernelGGL(relu_tap_bwd_kernel<T>, dim3(c2m_grid(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)y,
                                   (const T*)t, (const T*)gy, gl, (T*)out, n4, total);
        if (4 * n4 < total) hipLaunchKernelGGL(relu_tap_bwd_tail_kernel<T>, dim3(c2m_grid(total - 4 * n4, 256)), dim3(256), 0,
                                               (hipStream_t)stream, (const T*)y, (const T*)t, (const T*)gy, gl, (T*)out, 4 * n4, total););
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- SSIM
struct SsimWin { float mx, my, ex2, ey2, exy; };

__device__ __forceinline__ SsimWin ssim_window(const float* __restrict__ x, const float* __restrict__ y, int W) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const float a = x[dy * W + dx], b = y[dy * W + dx];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    const float inv = 1.0f / 9.0f;
    return SsimWin{sx * inv, sy * inv, sxx * inv, syy * inv, sxy * inv};
}

#define SSIM_C1 (0.01f * 0.01f)
#define SSIM_C2 (0.03f * 0.03f)

// loss = mean over valid 3x3 windows of clamp((1 - ssim)/2, 0, 1);  x = generated, y = target, planes [NC][H][W]
__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            long NC, int H, int W, double* __restrict__ part) {
    __shared__ double sm[4];
    const int Ho = H - 2, Wo = W - 2;
    const long total = NC * Ho * Wo;
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % Wo); const long r = i / Wo;
        const int oy = (int)(r % Ho); const long nc = r / Ho;
        const long o = nc * (long)H * W + (long)oy * W + ox;
        const SsimWin w = ssim_window(x + o, y + o, W);
        const float sgx = w.ex2 - w.mx * w.mx, sgy = w.ey2 - w.my * w.my, sgxy = w.exy - w.mx * w.my;
        const float n = (2.f * w.mx * w.my + SSIM_C1) * (2.f * sgxy + SSIM_C2);
        const float d = (w.mx * w.mx + w.my * w.my + SSIM_C1) * (sgx + sgy + SSIM_C2);
        const float v = (1.f - n / d) * 0.5f;
        s += fminf(fmaxf(v, 0.f), 1.f);
    }
    const double bs = block_sum_256_d((double)s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = bs;
}

C2M_API int c2m_ssim_fwd(const float* x, const float* y, float* out, long NC, int H, int W, void* workspace,
                         void* stream) {
    C2M_ENTER();
    const long total = NC * (long)(H - 2) * (W - 2);
    if (total <= 0 || H < 3 || W < 3) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int blocks = c2m_grid(total, 256); if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(blocks), dim3(256), 0, s, x, y, NC, H, W, (double*)workspace);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, (const double*)workspace, blocks, 1.0 / (double)total,
                       out);
    return (int)hipGetLastError();
}

// per-window derivative coefficients wrt (mu_x, E[x^2], E[xy]) of the window's loss term (already divided by count*9)
__global__ void ssim_coef_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gscale,
                                 float* __restrict__ coef, long NC, int H, int W) {
    const int Ho = H - 2, Wo = W - 2;
    const long total = NC * Ho * Wo;
    const float g = gscale[0] / ((float)total * 9.0f);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % Wo); const long r = i / Wo;
        const int oy = (int)(r % Ho); const long nc = r / Ho;
        const long o = nc * (long)H * W + (long)oy * W + ox;
        const SsimWin w = ssim_window(x + o, y + o, W);
        const float sgx = w.ex2 - w.mx * w.mx, sgy = w.ey2 - w.my * w.my, sgxy = w.exy - w.mx * w.my;
        const float n1 = 2.f * w.mx * w.my + SSIM_C1, n2 = 2.f * sgxy + SSIM_C2;
        const float d1 = w.mx * w.mx + w.my * w.my + SSIM_C1, d2 = sgx + sgy + SSIM_C2;
        const float n = n1 * n2, d = d1 * d2;
        const float v = (1.f - n / d) * 0.5f;
        float cA = 0.f, cB = 0.f, cC = 0.f;
        if (v >= 0.f && v <= 1.f) {
            const float dn_dm = 2.f * w.my * (n2 - n1);
            const float dd_dm = 2.f * w.mx * (d2 - d1);
            const float ds_dm = (dn_dm * d - n * dd_dm) / (d * d);
            const float ds_dxx = -n * d1 / (d * d);
            const float ds_dxy = 2.f * n1 / d;
            cA = -0.5f * g * ds_dm; cB = -0.5f * g * ds_dxx; cC = -0.5f * g * ds_dxy;
        }
        coef[i * 3 + 0] = cA; coef[i * 3 + 1] = cB; coef[i * 3 + 2] = cC;
    }
}

// gx[p] = sum over the <=9 windows containing p of (A + 2*B*x[p] + C*y[p])
__global__ void ssim_gather_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef,
                                   float* __restrict__ gx, long NC, int H, int W) {
    const int Ho = H - 2, Wo = W - 2;
    const long total = NC * (long)H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int px = (int)(i % W); const long r = i / W;
        const int py = (int)(r % H); const long nc = r / H;
        float sa = 0.f, sb = 0.f, sc = 0.f;
        for (int oy = max(py - 2, 0); oy <= min(py, Ho - 1); ++oy)
            for (int ox = max(px - 2, 0); ox <= min(px, Wo - 1); ++ox) {
                const long q = ((nc * Ho + oy) * (long)Wo + ox) * 3;
                sa += coef[q]; sb += coef[q + 1]; sc += coef[q + 2];
            }
        gx[i] = sa + 2.f * sb * x[i] + sc * y[i];
    }
}

// coef workspace: NC*(H-2)*(W-2)*3 floats
C2M_API int c2m_ssim_bwd(const float* x, const float* y, const float* gscale, float* gx, float* coef, long NC, int H,
                         int W, void* stream) {
    C2M_ENTER();
    const long nwin = NC * (long)(H - 2) * (W - 2);
    if (nwin <= 0) return (int)hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_coef_kernel, dim3(c2m_grid(nwin, 256)), dim3(256), 0, s, x, y, gscale, coef, NC, H, W);
    hipLaunchKernelGGL(ssim_gather_kernel, dim3(c2m_grid(NC * (long)H * W, 256)), dim3(256), 0, s, x, y, coef, gx, NC, H,
                       W);
    return (int)hipGetLastError();
}
