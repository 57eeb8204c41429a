// Normalisation + activation kernels (HBM-bound), gfx950.
//
// Fuses what the reference runs as separate ATen ops around every convolution:
//   BatchNorm2d/3d (train mode, running-stat update)  -> LeakyReLU(0.2)/ReLU   layers/down_block.py:19-22,44-47
//                                                                              same_block.py:64-67 up_block.py:11-12
//                                                                              residual_block.py:20-28
//   InstanceNorm2d(affine)                             -> LeakyReLU            same_block.py:19-22,41-44
//   SPADE: InstanceNorm2d(affine=False)(x)*(1+gamma)+beta -> LeakyReLU         spade_block.py:68-77,
//                                                                              residual_block.py:56-70
// Layout: [N][C][S] with S = (T*)H*W contiguous ("planes").  mode: 0 = per-plane statistics (instance norm),
// 1 = per-channel statistics over N*S (batch norm).
//
// Statistics are a deterministic 3-stage reduction: (plane, chunk) partials (two passes over an L2-resident chunk:
// mean, then centred sum of squares) -> Chan combination in a fixed order -> mean / invstd (+ running stats).
// Backward: one reduction pass (sum g', sum g'*xhat, and the SPADE gamma/beta map gradients) + one apply pass.
#include "common.h"
#include "dtype.h"
#include <stdlib.h>

#define NORM_CHUNK 8192

struct NormShape { int N, C; long S; int mode; int chunks; };

static inline int norm_chunks(long S) { return (int)((S + NORM_CHUNK - 1) / NORM_CHUNK); }

// partial[(plane*chunks + chunk)*2 + {0: mean, 1: M2}]
template <class T>
__global__ __launch_bounds__(256) void norm_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                            long S, int chunks) {
    __shared__ float sm[4];
    const long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const long beg = (long)chunk * NORM_CHUNK;
    const long end = beg + NORM_CHUNK < S ? beg + NORM_CHUNK : S;
    const T* __restrict__ p = x + plane * S;
    const int cnt = (int)(end - beg);
    const bool vec = (S & 3) == 0 && ((uintptr_t)x & C2mVec4<T>::mask) == 0;      // chunk bounds are multiples of 4 then
    float s = 0.f;
    // a chunk is at most 8 float4 per thread: they stay in registers for the second (centred) pass -- the first form read the
    // chunk twice (the second time from L2); same arithmetic in the same order, so the statistics keep their bits
    constexpr int NV = NORM_CHUNK / 1024;
    float4 keep[NV];
    if (vec) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long i = beg + threadIdx.x * 4 + (long)k * 1024;
            if (i < end) {
                keep[k] = c2m_ld4(p + i);
                s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
            }
        }
    } else {
        for (long i = beg + threadIdx.x; i < end; i += 256) s += c2m_ld(p, i);
    }
    s = block_sum_256(s, sm);
    const float mean = s / (float)cnt;
    float m2 = 0.f;
    if (vec) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long i = beg + threadIdx.x * 4 + (long)k * 1024;
            if (i < end) {
                const float a = keep[k].x - mean, b = keep[k].y - mean, c = keep[k].z - mean, d = keep[k].w - mean;
                m2 += (a * a + b * b) + (c * c + d * d);
            }
        }
    } else {
        for (long i = beg + threadIdx.x; i < end; i += 256) { const float d = c2m_ld(p, i) - mean; m2 += d * d; }
    }
    m2 = block_sum_256(m2, sm);
    if (threadIdx.x == 0) {
        partial[(long)blockIdx.x * 2 + 0] = mean;
        partial[(long)blockIdx.x * 2 + 1] = m2;
    }
}

// one thread per statistic (plane for mode 0, channel for mode 1)
__global__ void norm_finalize_kernel(const float* __restrict__ partial, float* __restrict__ mean_out,
                                     float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, NormShape sh, float eps, float momentum) {
    const int nstat = sh.mode == 0 ? sh.N * sh.C : sh.C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nstat) return;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    const int nrep = sh.mode == 0 ? 1 : sh.N;
    for (int r = 0; r < nrep; ++r) {
        const long plane = sh.mode == 0 ? i : (long)r * sh.C + i;
        for (int c = 0; c < sh.chunks; ++c) {
            const long beg = (long)c * NORM_CHUNK;
            const double nb = (double)((beg + NORM_CHUNK < sh.S ? beg + NORM_CHUNK : sh.S) - beg);
            const double mb = partial[(plane * sh.chunks + c) * 2 + 0];
            const double vb = partial[(plane * sh.chunks + c) * 2 + 1];
            const double tot = n + nb;
            const double delta = mb - mean;
            mean += delta * nb / tot;
            m2 += vb + delta * delta * n * nb / tot;
            n = tot;
        }
    }
    const double var = m2 / n;
    mean_out[i] = (float)mean;
    invstd_out[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
        running_mean[i] = (float)((1.0 - momentum) * running_mean[i] + momentum * mean);
        running_var[i] = (float)((1.0 - momentum) * running_var[i] + momentum * unbiased);
    }
}

// Batch statistics (mode 1): one wave per channel.  A thread per channel walks N * chunks partials through a chain of
// dependent double divisions (8-10 us for N = 40); here each lane merges every 64th partial and the 64 lane results are
// merged pairwise (Chan et al.), lane 0 holding the channel's statistics.
__global__ __launch_bounds__(256) void norm_finalize_bn_kernel(const float* __restrict__ partial,
                                                               float* __restrict__ mean_out,
                                                               float* __restrict__ invstd_out,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, NormShape sh, float eps,
                                                               float momentum) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= sh.C) return;                                   // wave-uniform
    double n = 0.0, mean = 0.0, m2 = 0.0;
    const int total = sh.N * sh.chunks;
    for (int e = lane; e < total; e += 64) {
        const int r = e / sh.chunks, c = e - r * sh.chunks;
        const long q = (((long)r * sh.C + ch) * sh.chunks + c) * 2;
        const long beg = (long)c * NORM_CHUNK;
        const double nb = (double)((beg + NORM_CHUNK < sh.S ? beg + NORM_CHUNK : sh.S) - beg);
        const double mb = partial[q], vb = partial[q + 1];
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += vb + delta * delta * n * nb / tot;
        n = tot;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double nb = __shfl_down(n, o, 64), mb = __shfl_down(mean, o, 64), vb = __shfl_down(m2, o, 64);
        const double tot = n + nb;
        if (tot > 0.0) {
            const double delta = mb - mean;
            mean += delta * nb / tot;
            m2 += vb + delta * delta * n * nb / tot;
            n = tot;
        }
    }
    if (lane != 0) return;
    const double var = m2 / n;
    mean_out[ch] = (float)mean;
    invstd_out[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
        running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
        running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unbiased);
    }
}

C2M_API long c2m_norm_workspace_floats(int N, int C, long S) { return (long)N * C * norm_chunks(S) * 4; }

C2M_API int c2m_norm_stats(const void* x, float* mean, float* invstd, float* running_mean, float* running_var,
                           float* workspace, int N, int C, long S, int mode, float eps, float momentum, int dt, void* stream) {
    C2M_ENTER();
    if ((long)N * C * S <= 0) return 0;
    NormShape sh{N, C, S, mode, norm_chunks(S)};
    hipStream_t s = (hipStream_t)stream;
    C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(norm_partial_kernel<T>, dim3((unsigned)((long)N * C * sh.chunks)), dim3(256), 0, s,
                                           (const T*)x, workspace, S, sh.chunks););
    const int nstat = mode == 0 ? N * C : C;
    if (mode == 1 && N * sh.chunks >= 16)
        hipLaunchKernelGGL(norm_finalize_bn_kernel, dim3(c2m_cdiv(C, 4)), dim3(256), 0, s, workspace, mean, invstd,
                           running_mean, running_var, sh, eps, momentum);
    else
        hipLaunchKernelGGL(norm_finalize_kernel, dim3(c2m_cdiv(nstat, 128)), dim3(128), 0, s, workspace, mean, invstd,
                           running_mean, running_var, sh, eps, momentum);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- apply (forward)
// y = act( xhat * scale + shift ), xhat = (x - mean)*invstd
//   affine: scale = gamma[c], shift = beta[c];  plain: 1, 0;  SPADE: scale = 1 + gb[n, c, s], shift = gb[n, C + c, s]
template <class T>
struct ApplyP {
    const T* x; const float* mean; const float* invstd; const float* gamma; const float* beta; const T* gb;
    T* y;
    int N, C; long S; int mode; int act; float slope;
};

template <class T>
__global__ void norm_apply_kernel(const ApplyP<T> p) {
    const long total = (long)p.N * p.C * p.S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long plane = i / p.S;
        const long s = i - plane * p.S;
        const int c = (int)(plane % p.C);
        const int n = (int)(plane / p.C);
        const int st = p.mode == 0 ? (int)plane : c;
        const float xhat = (c2m_ld(p.x, i) - p.mean[st]) * p.invstd[st];
        float v;
        if (p.gb) {
            const long gbase = ((long)n * 2 * p.C + c) * p.S + s;
            v = xhat * (1.0f + c2m_ld(p.gb, gbase)) + c2m_ld(p.gb, gbase + (long)p.C * p.S);
        } else if (p.gamma) {
            v = xhat * p.gamma[c] + p.beta[c];
        } else {
            v = xhat;
        }
        c2m_st(p.y, i, c2m_act(v, p.act, p.slope));
    }
}

// Large planes (S >= 1024, S % 4 == 0): one workgroup per (plane, 8192-element chunk), 16-byte accesses, the plane's
// statistics and affine parameters in registers -- no per-element index arithmetic.
template <class T>
__global__ __launch_bounds__(256) void norm_apply_vec_kernel(const ApplyP<T> p, int chunks) {
    const long plane = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const long beg = (long)chunk * NORM_CHUNK;
    const long end = beg + NORM_CHUNK < p.S ? beg + NORM_CHUNK : p.S;
    const int c = (int)(plane % p.C);
    const int n = (int)(plane / p.C);
    const int st = p.mode == 0 ? (int)plane : c;
    const float mean = p.mean[st], invstd = p.invstd[st];
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    const T* __restrict__ x = p.x + plane * p.S;
    T* __restrict__ y = p.y + plane * p.S;
    const T* __restrict__ g0 = p.gb ? p.gb + ((long)n * 2 * p.C + c) * p.S : nullptr;
    const T* __restrict__ g1 = p.gb ? g0 + (long)p.C * p.S : nullptr;
    for (long i = beg + threadIdx.x * 4; i < end; i += 1024) {
        const float4 v = c2m_ld4(x + i);
        float4 sc = make_float4(ga, ga, ga, ga), sh = make_float4(be, be, be, be);
        if (p.gb) {
            const float4 a = c2m_ld4(g0 + i);
            sc = make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w);
            sh = c2m_ld4(g1 + i);
        }
        float4 o;
        o.x = c2m_act((v.x - mean) * invstd * sc.x + sh.x, p.act, p.slope);
        o.y = c2m_act((v.y - mean) * invstd * sc.y + sh.y, p.act, p.slope);
        o.z = c2m_act((v.z - mean) * invstd * sc.z + sh.z, p.act, p.slope);
        o.w = c2m_act((v.w - mean) * invstd * sc.w + sh.w, p.act, p.slope);
        c2m_st4(y + i, o);
    }
}

// ------------------------------------------------------------------------------------------- NC8 side output (round 4)
// bf16 data path: the normalised activation feeds a channel-blocked convolution (conv_nc8.hip), so the apply pass writes it in BOTH
// layouts at once -- NCHW for everybody else, NC8 ([N][C/8][S][8], the 8 channels of a pixel in one 16-byte unit) for that conv --
// instead of a separate c2m_nchw_to_nc8 pass over the tensor (one read of it less; the layout passes were 3.3 / 5.9 ms of a
// configs[3] / configs[2] step).  A thread owns 8 consecutive pixels of 8 consecutive channels: eight 16-byte loads, the per-channel
// arithmetic of norm_apply_vec_kernel, eight 16-byte NCHW stores, an 8x8 transpose of bf16 pairs in registers (v_perm_b32), eight
// 16-byte NC8 stores (128 contiguous bytes).  Same values bit for bit in both outputs.
__device__ __forceinline__ void nc8_transpose_store(const unsigned (&r)[8][4], uint4* __restrict__ dst) {
#pragma unroll
    for (int px = 0; px < 8; ++px) {
        const int q = px >> 1;
        const unsigned sel = (px & 1) ? 0x07060302u : 0x05040100u;
        uint4 o;
        o.x = __builtin_amdgcn_perm(r[1][q], r[0][q], sel);
        o.y = __builtin_amdgcn_perm(r[3][q], r[2][q], sel);
        o.z = __builtin_amdgcn_perm(r[5][q], r[4][q], sel);
        o.w = __builtin_amdgcn_perm(r[7][q], r[6][q], sel);
        dst[px] = o;
    }
}
__device__ __forceinline__ void nc8_unpack8(const uint4 v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned nc8_pack2(float a, float b) {
    typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t q = {(bf16_t)a, (bf16_t)b};
    return __builtin_bit_cast(unsigned, q);
}

__global__ __launch_bounds__(256) void norm_apply_nc8_kernel(const ApplyP<bf16_t> p, uint4* __restrict__ yn, int CB, long S8, long total) {
    const uint4* __restrict__ x4 = reinterpret_cast<const uint4*>(p.x);
    const uint4* __restrict__ g4 = reinterpret_cast<const uint4*>(p.gb);
    uint4* __restrict__ y4 = reinterpret_cast<uint4*>(p.y);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i % S8, ncb = i / S8;
        const int cb = (int)(ncb % CB), n = (int)(ncb / CB);
        unsigned r[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb * 8 + j;
            r[j][0] = r[j][1] = r[j][2] = r[j][3] = 0u;
            if (c < p.C) {
                const long plane = (long)n * p.C + c;
                const int st = p.mode == 0 ? (int)plane : c;
                const float mean = p.mean[st], invstd = p.invstd[st];
                float xv[8], sc[8], sh[8], o[8];
                nc8_unpack8(x4[plane * S8 + g], xv);
                if (p.gb) {
                    const long gpl = ((long)n * 2 * p.C + c) * S8 + g;
                    nc8_unpack8(g4[gpl], sc);
                    nc8_unpack8(g4[gpl + (long)p.C * S8], sh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sc[e] = 1.0f + sc[e];
                } else {
                    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sc[e] = ga; sh[e] = be; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = c2m_act((xv[e] - mean) * invstd * sc[e] + sh[e], p.act, p.slope);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[j][e] = nc8_pack2(o[2 * e], o[2 * e + 1]);
                if (y4) y4[plane * S8 + g] = make_uint4(r[j][0], r[j][1], r[j][2], r[j][3]);      // (NULL: the NC8 form is the only output)
            }
        }
        nc8_transpose_store(r, yn + (((long)n * CB + cb) * S8 + g) * 8);
    }
}

static inline bool norm_vec_ok(long S, const void* a, const void* b, const void* c, const void* d, int dt) {
    return S >= 1024 && (S & 3) == 0 &&
           ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & (dt == C2M_BF16 ? 7 : 15)) == 0;
}

// y_nc8 (optional, bf16 tensors with S % 8 == 0 only): the same result in the channel-blocked layout of conv_nc8.hip as well.
C2M_API int c2m_norm_apply(const void* x, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, const void* gb, void* y, void* y_nc8, int N, int C, long S, int mode, int act,
                           float slope, int dt, void* stream) {
    C2M_ENTER();
    const long total = (long)N * C * S;
    if (total <= 0) return 0;
    if (y_nc8) {
        if (dt != C2M_BF16 || (S & 7) || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gb) | ((uintptr_t)y_nc8)) & 15))
            return (int)hipErrorInvalidValue;
        ApplyP<bf16_t> p{(const bf16_t*)x, mean, invstd, gamma, beta, (const bf16_t*)gb, (bf16_t*)y, N, C, S, mode, act, slope};
        const int CB = (C + 7) / 8;
        const long n8 = (long)N * CB * (S / 8);
        hipLaunchKernelGGL(norm_apply_nc8_kernel, dim3(c2m_grid(n8, 256)), dim3(256), 0, (hipStream_t)stream, p, (uint4*)y_nc8, CB, S / 8, n8);
        return (int)hipGetLastError();
    }
    const bool vec = norm_vec_ok(S, x, y, gb, nullptr, dt);
    C2M_DISPATCH_DT(dt,
        ApplyP<T> p{(const T*)x, mean, invstd, gamma, beta, (const T*)gb, (T*)y, N, C, S, mode, act, slope};
        if (vec) {
            const int chunks = norm_chunks(S);
            hipLaunchKernelGGL(norm_apply_vec_kernel<T>, dim3((unsigned)((long)N * C * chunks)), dim3(256), 0,
                               (hipStream_t)stream, p, chunks);
        } else {
            hipLaunchKernelGGL(norm_apply_kernel<T>, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
        });
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- instance norm in ONE launch (round 5)
// mode 0 (per-plane statistics: InstanceNorm2d / SPADE, same_block.py:19-22, spade_block.py:68-77) with planes of <= 32768
// elements: one workgroup owns a whole plane, keeps it in registers (<= 32 float4 per thread), computes mean and centred sum
// of squares exactly like norm_partial_kernel (same two passes, same block sums -- bit-identical statistics for planes of one
// chunk) and applies scale / shift / activation straight from the registers.  Replaces partial + finalize + apply: one read of
// x instead of two, one launch instead of three (65 launches and ~0.5 ms of a BASELINE configs[1] step).
template <class T, int NV>
__global__ __launch_bounds__(256) void norm_inst_fused_kernel(const ApplyP<T> p, float* __restrict__ mean_out,
                                                              float* __restrict__ invstd_out, float eps) {
    __shared__ float sm[4];
    const long plane = blockIdx.x;
    const int c = (int)(plane % p.C);
    const int n = (int)(plane / p.C);
    const T* __restrict__ x = p.x + plane * p.S;
    const int S = (int)p.S;
    float4 keep[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x * 4 + k * 1024;
        if (i < S) {
            keep[k] = c2m_ld4(x + i);
            s += (keep[k].x + keep[k].y) + (keep[k].z + keep[k].w);
        }
    }
    s = block_sum_256(s, sm);
    const float mean = s / (float)S;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x * 4 + k * 1024;
        if (i < S) {
            const float a = keep[k].x - mean, b = keep[k].y - mean, cc = keep[k].z - mean, d = keep[k].w - mean;
            m2 += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    m2 = block_sum_256(m2, sm);
    const double var = (double)m2 / (double)S;                       // norm_finalize_kernel with one chunk
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) { mean_out[plane] = mean; invstd_out[plane] = invstd; }
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    T* __restrict__ y = p.y + plane * p.S;
    const T* __restrict__ g0 = p.gb ? p.gb + ((long)n * 2 * p.C + c) * p.S : nullptr;
    const T* __restrict__ g1 = p.gb ? g0 + (long)p.C * p.S : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = threadIdx.x * 4 + k * 1024;
        if (i < S) {
            const float4 v = keep[k];
            float4 sc = make_float4(ga, ga, ga, ga), sh = make_float4(be, be, be, be);
            if (p.gb) {
                const float4 a = c2m_ld4(g0 + i);
                sc = make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w);
                sh = c2m_ld4(g1 + i);
            }
            float4 o;
            o.x = c2m_act((v.x - mean) * invstd * sc.x + sh.x, p.act, p.slope);
            o.y = c2m_act((v.y - mean) * invstd * sc.y + sh.y, p.act, p.slope);
            o.z = c2m_act((v.z - mean) * invstd * sc.z + sh.z, p.act, p.slope);
            o.w = c2m_act((v.w - mean) * invstd * sc.w + sh.w, p.act, p.slope);
            c2m_st4(y + i, o);
        }
    }
}

// Batch norm (mode 1) of SMALL channel volumes in one launch (round 5): the deep encoder levels (8x16 maps and
// below, 40 folded frames: N * S <= 8192 elements per channel) ran partial + finalize + apply as three launches of a few
// microseconds each.  One workgroup owns a channel: its N planes are walked as ONE index range (coalesced inside a plane), kept in
// registers (NE elements per thread), two-pass statistics (mean, then centred squares: cancellation-safe like the chunked path),
// running-statistics update, apply.  Same for the backward up to 8192 elements per channel (x-hat and g' in registers).
template <class T, int NE>
__global__ __launch_bounds__(256) void norm_bn_small_fused_kernel(const ApplyP<T> p, float* __restrict__ mean_out,
                                                                  float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, float eps, float momentum) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    const int S = (int)p.S, total = p.N * S;
    const long cstride = (long)p.C * p.S;                   // from plane (n, c) to plane (n + 1, c)
    // e / S by multiply-high (exact for e < 2^32 / S: e, S <= 32768): a hardware integer division is ~35 VALU instructions, and
    // 3 x NE of them per thread made the first form of this kernel slower than the three launches it replaces
    const unsigned magic = (unsigned)((0x100000000ULL + (unsigned)S - 1) / (unsigned)S);
    const T* __restrict__ xc = p.x + (long)c * p.S;
    float keep[NE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * 256;
        keep[k] = 0.f;
        if (e < total) {
            const int n = S == 1 ? e : (int)__umulhi((unsigned)e, magic), q = e - n * S;
            keep[k] = c2m_ld(xc, (long)n * cstride + q);
            s += keep[k];
        }
    }
    s = block_sum_256(s, sm);
    const float mean = s / (float)total;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < total) { const float d = keep[k] - mean; m2 += d * d; }
    }
    m2 = block_sum_256(m2, sm);
    const double var = (double)m2 / (double)total;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mean_out[c] = mean; invstd_out[c] = invstd;
        if (running_mean) {
            const double unbiased = total > 1 ? (double)m2 / (double)(total - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * (double)mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    T* __restrict__ yc = p.y + (long)c * p.S;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < total) {
            const int n = S == 1 ? e : (int)__umulhi((unsigned)e, magic), q = e - n * S;
            c2m_st(yc, (long)n * cstride + q, c2m_act((keep[k] - mean) * invstd * ga + be, p.act, p.slope));
        }
    }
}

// A/B knob, bit mask: 1 = one-launch instance norm, 2 = one-launch small batch norm (C2M_NORM_FUSED=0 / 1 / 2 / 3; default 3)
static int norm_fused_on = [] { const char* e = getenv("C2M_NORM_FUSED"); return e ? atoi(e) & 3 : 3; }();
// tests / A/B runs: 0 = every norm on the three-launch path; returns the old mask
C2M_API int c2m_norm_set_fused(int mask) { const int old = norm_fused_on; norm_fused_on = mask & 3; return old; }

// Statistics + apply in one call (c2m_norm_stats followed by c2m_norm_apply, without the NC8 side output): instance-norm planes of
// 1024 ... 32768 elements take the one-launch kernel above, everything else the three launches.
C2M_API int c2m_norm_fwd(const void* x, float* mean, float* invstd, float* running_mean, float* running_var, float* workspace,
                         const float* gamma, const float* beta, const void* gb, void* y, int N, int C, long S, int mode,
                         float eps, float momentum, int act, float slope, int dt, void* stream) {
    C2M_ENTER();
    if ((long)N * C * S <= 0) return 0;
    if ((norm_fused_on & 1) && mode == 0 && S <= 32768 && norm_vec_ok(S, x, y, gb, nullptr, dt)) {
        C2M_DISPATCH_DT(dt,
            ApplyP<T> p{(const T*)x, mean, invstd, gamma, beta, (const T*)gb, (T*)y, N, C, S, mode, act, slope};
            if (S <= 8192)
                hipLaunchKernelGGL((norm_inst_fused_kernel<T, 8>), dim3((unsigned)((long)N * C)), dim3(256), 0, (hipStream_t)stream, p, mean, invstd, eps);
            else
                hipLaunchKernelGGL((norm_inst_fused_kernel<T, 32>), dim3((unsigned)((long)N * C)), dim3(256), 0, (hipStream_t)stream, p, mean, invstd, eps););
        return (int)hipGetLastError();
    }
    // (<= 8192 elements per channel: a 128-elements-per-thread variant for the 16 x 32 maps -- 20480 per channel, C workgroups --
    // measured 47 us against ~30 for the three launches it replaced, profiles/r05_norm_one_launch_kernel_times.txt)
    if ((norm_fused_on & 2) && mode == 1 && !gb && (long)N * S <= 8192) {
        const long tot = (long)N * S;
        C2M_DISPATCH_DT(dt,
            ApplyP<T> p{(const T*)x, mean, invstd, gamma, beta, nullptr, (T*)y, N, C, S, mode, act, slope};
            if (tot <= 2048)
                hipLaunchKernelGGL((norm_bn_small_fused_kernel<T, 8>), dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, p, mean, invstd, running_mean, running_var, eps, momentum);
            else
                hipLaunchKernelGGL((norm_bn_small_fused_kernel<T, 32>), dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, p, mean, invstd, running_mean, running_var, eps, momentum););
        return (int)hipGetLastError();
    }
    const int rc = c2m_norm_stats(x, mean, invstd, running_mean, running_var, workspace, N, C, S, mode, eps, momentum, dt, stream);
    if (rc) return rc;
    return c2m_norm_apply(x, mean, invstd, gamma, beta, gb, y, nullptr, N, C, S, mode, act, slope, dt, stream);
}

// ------------------------------------------------------------------------------------------- backward
template <class T>
struct BwdP {
    const T* x; const T* gy; const float* mean; const float* invstd; const float* gamma; const float* beta;
    const T* gb;
    T* ggb;          // SPADE map gradients [N,2C,S] (written in the reduce pass)
    float* partial;  // [(plane*chunks + chunk)*2]: sum g'*m, sum g'*m*xhat  (m = 1+gamma_map for SPADE else 1)
    float* coef;     // [nstat*2]: c1, c2 of dx = invstd*(g'*scale - c1 - xhat*c2)
    float* dgamma; float* dbeta;
    T* dx;
    int N, C; long S; int mode; int act; float slope; int chunks;
};

__device__ __forceinline__ float act_grad(float pre, int act, float slope) {
    switch (act) {
        case C2M_ACT_RELU: return pre > 0.f ? 1.f : 0.f;
        case C2M_ACT_LRELU: return pre > 0.f ? 1.f : slope;
        default: return 1.f;
    }
}

template <class T>
__global__ __launch_bounds__(256) void norm_bwd_reduce_kernel(const BwdP<T> p) {
    __shared__ float sm[4];
    const long plane = blockIdx.x / p.chunks;
    const int chunk = blockIdx.x % p.chunks;
    const long beg = (long)chunk * NORM_CHUNK;
    const long end = beg + NORM_CHUNK < p.S ? beg + NORM_CHUNK : p.S;
    const int c = (int)(plane % p.C);
    const int n = (int)(plane / p.C);
    const int st = p.mode == 0 ? (int)plane : c;
    const float mean = p.mean[st], invstd = p.invstd[st];
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const bool vec = (p.S & 3) == 0 && (((uintptr_t)p.x | (uintptr_t)p.gy | (uintptr_t)p.gb | (uintptr_t)p.ggb) & C2mVec4<T>::mask) == 0;
    if (vec) {                       // 16-byte accesses; element order inside a thread is fixed -> deterministic sums
        const T* __restrict__ xp = p.x + plane * p.S;
        const T* __restrict__ gp = p.gy + plane * p.S;
        const long gplane = ((long)n * 2 * p.C + c) * p.S, goff = (long)p.C * p.S;
        for (long s = beg + threadIdx.x * 4; s < end; s += 1024) {
            const float4 xv = c2m_ld4(xp + s);
            const float4 gv = c2m_ld4(gp + s);
            float4 sc = make_float4(ga, ga, ga, ga), sh = make_float4(be, be, be, be);
            if (p.gb) {
                const float4 a = c2m_ld4(p.gb + gplane + s);
                sc = make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w);
                sh = c2m_ld4(p.gb + gplane + goff + s);
            }
            const float xh[4] = {(xv.x - mean) * invstd, (xv.y - mean) * invstd, (xv.z - mean) * invstd, (xv.w - mean) * invstd};
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, gyv[4] = {gv.x, gv.y, gv.z, gv.w};
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[e] = gyv[e] * act_grad(xh[e] * scv[e] + shv[e], p.act, p.slope);
                if (p.gb) { s1 += g[e] * scv[e]; s2 += g[e] * scv[e] * xh[e]; }
                else { s1 += g[e]; s2 += g[e] * xh[e]; }
            }
            if (p.gb) {
                c2m_st4(p.ggb + gplane + s, make_float4(g[0] * xh[0], g[1] * xh[1], g[2] * xh[2], g[3] * xh[3]));
                c2m_st4(p.ggb + gplane + goff + s, make_float4(g[0], g[1], g[2], g[3]));
            }
        }
    } else
    for (long s = beg + threadIdx.x; s < end; s += 256) {
        const long i = plane * p.S + s;
        const float xhat = (c2m_ld(p.x, i) - mean) * invstd;
        float scale = ga, shift = be;
        long gbase = 0;
        if (p.gb) {
            gbase = ((long)n * 2 * p.C + c) * p.S + s;
            scale = 1.0f + c2m_ld(p.gb, gbase);
            shift = c2m_ld(p.gb, gbase + (long)p.C * p.S);
        }
        const float g = c2m_ld(p.gy, i) * act_grad(xhat * scale + shift, p.act, p.slope);
        if (p.gb) {
            c2m_st(p.ggb, gbase, g * xhat);
            c2m_st(p.ggb, gbase + (long)p.C * p.S, g);
            s1 += g * scale; s2 += g * scale * xhat;
        } else {
            s1 += g; s2 += g * xhat;
        }
    }
    s1 = block_sum_256(s1, sm);
    s2 = block_sum_256(s2, sm);
    if (threadIdx.x == 0) {
        p.partial[(long)blockIdx.x * 2 + 0] = s1;
        p.partial[(long)blockIdx.x * 2 + 1] = s2;
    }
}

template <class T>
__global__ void norm_bwd_finalize_kernel(const BwdP<T> p) {
    const int nstat = p.mode == 0 ? p.N * p.C : p.C;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (p.mode == 1) {
        if (i >= nstat) return;
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < p.N; ++r)
            for (int c = 0; c < p.chunks; ++c) {
                const long q = (((long)r * p.C + i) * p.chunks + c) * 2;
                s1 += p.partial[q]; s2 += p.partial[q + 1];
            }
        const double cnt = (double)p.N * (double)p.S;
        const double ga = p.gamma ? (double)p.gamma[i] : 1.0;
        p.coef[i * 2 + 0] = (float)(ga * s1 / cnt);
        p.coef[i * 2 + 1] = (float)(ga * s2 / cnt);
        if (p.dgamma) { p.dgamma[i] = (float)s2; p.dbeta[i] = (float)s1; }
    } else {
        // per-plane coefficients; per-channel dgamma/dbeta are finished by thread c over its N planes
        if (i < nstat) {
            double s1 = 0.0, s2 = 0.0;
            for (int c = 0; c < p.chunks; ++c) {
                const long q = ((long)i * p.chunks + c) * 2;
                s1 += p.partial[q]; s2 += p.partial[q + 1];
            }
            const double ga = p.gamma ? (double)p.gamma[i % p.C] : 1.0;
            p.coef[i * 2 + 0] = (float)(ga * s1 / (double)p.S);
            p.coef[i * 2 + 1] = (float)(ga * s2 / (double)p.S);
        }
        if (p.dgamma && i < p.C) {
            double s1 = 0.0, s2 = 0.0;
            for (int r = 0; r < p.N; ++r)
                for (int c = 0; c < p.chunks; ++c) {
                    const long q = (((long)r * p.C + i) * p.chunks + c) * 2;
                    s1 += p.partial[q]; s2 += p.partial[q + 1];
                }
            p.dgamma[i] = (float)s2; p.dbeta[i] = (float)s1;
        }
    }
}

// One wave per channel (see norm_finalize_bn_kernel): batch mode sums all N * chunks partials of the channel; instance
// mode gives each lane whole planes (their coefficients) and sums the lane totals for the affine gradients.
template <class T>
__global__ __launch_bounds__(256) void norm_bwd_finalize_wave_kernel(const BwdP<T> p) {
    const int lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ch >= p.C) return;
    const double ga = p.gamma ? (double)p.gamma[ch] : 1.0;
    double s1 = 0.0, s2 = 0.0;
    if (p.mode == 1) {
        const int total = p.N * p.chunks;
        for (int e = lane; e < total; e += 64) {
            const int r = e / p.chunks, c = e - r * p.chunks;
            const long q = (((long)r * p.C + ch) * p.chunks + c) * 2;
            s1 += p.partial[q]; s2 += p.partial[q + 1];
        }
    } else {
        for (int r = lane; r < p.N; r += 64) {
            const long plane = (long)r * p.C + ch;
            double a = 0.0, b = 0.0;
            for (int c = 0; c < p.chunks; ++c) {
                const long q = (plane * p.chunks + c) * 2;
                a += p.partial[q]; b += p.partial[q + 1];
            }
            p.coef[plane * 2 + 0] = (float)(ga * a / (double)p.S);
            p.coef[plane * 2 + 1] = (float)(ga * b / (double)p.S);
            s1 += a; s2 += b;
        }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if (lane != 0) return;
    if (p.mode == 1) {
        const double cnt = (double)p.N * (double)p.S;
        p.coef[ch * 2 + 0] = (float)(ga * s1 / cnt);
        p.coef[ch * 2 + 1] = (float)(ga * s2 / cnt);
    }
    if (p.dgamma) { p.dgamma[ch] = (float)s2; p.dbeta[ch] = (float)s1; }
}

template <class T>
__global__ void norm_bwd_apply_kernel(const BwdP<T> p) {
    const long total = (long)p.N * p.C * p.S;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long plane = i / p.S;
        const long s = i - plane * p.S;
        const int c = (int)(plane % p.C);
        const int n = (int)(plane / p.C);
        const int st = p.mode == 0 ? (int)plane : c;
        const float invstd = p.invstd[st];
        const float xhat = (c2m_ld(p.x, i) - p.mean[st]) * invstd;
        float scale = p.gamma ? p.gamma[c] : 1.f, shift = p.gamma ? p.beta[c] : 0.f;
        if (p.gb) {
            const long gbase = ((long)n * 2 * p.C + c) * p.S + s;
            scale = 1.0f + c2m_ld(p.gb, gbase);
            shift = c2m_ld(p.gb, gbase + (long)p.C * p.S);
        }
        const float g = c2m_ld(p.gy, i) * act_grad(xhat * scale + shift, p.act, p.slope);
        c2m_st(p.dx, i, invstd * (g * scale - p.coef[st * 2 + 0] - xhat * p.coef[st * 2 + 1]));
    }
}

template <class T>
__global__ __launch_bounds__(256) void norm_bwd_apply_vec_kernel(const BwdP<T> p) {
    const long plane = blockIdx.x / p.chunks;
    const int chunk = blockIdx.x % p.chunks;
    const long beg = (long)chunk * NORM_CHUNK;
    const long end = beg + NORM_CHUNK < p.S ? beg + NORM_CHUNK : p.S;
    const int c = (int)(plane % p.C);
    const int n = (int)(plane / p.C);
    const int st = p.mode == 0 ? (int)plane : c;
    const float mean = p.mean[st], invstd = p.invstd[st], c1 = p.coef[st * 2 + 0], c2 = p.coef[st * 2 + 1];
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    const T* __restrict__ xp = p.x + plane * p.S;
    const T* __restrict__ gp = p.gy + plane * p.S;
    T* __restrict__ dp = p.dx + plane * p.S;
    const long gplane = ((long)n * 2 * p.C + c) * p.S, goff = (long)p.C * p.S;
    for (long s = beg + threadIdx.x * 4; s < end; s += 1024) {
        const float4 xv = c2m_ld4(xp + s);
        const float4 gv = c2m_ld4(gp + s);
        float4 sc = make_float4(ga, ga, ga, ga), sh = make_float4(be, be, be, be);
        if (p.gb) {
            const float4 a = c2m_ld4(p.gb + gplane + s);
            sc = make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w);
            sh = c2m_ld4(p.gb + gplane + goff + s);
        }
        const float xh[4] = {(xv.x - mean) * invstd, (xv.y - mean) * invstd, (xv.z - mean) * invstd, (xv.w - mean) * invstd};
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, gyv[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float g = gyv[e] * act_grad(xh[e] * scv[e] + shv[e], p.act, p.slope);
            o[e] = invstd * (g * scv[e] - c1 - xh[e] * c2);
        }
        c2m_st4(dp + s, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// dx of the norm in both layouts (see norm_apply_nc8_kernel): the gradient it hands back is the dY of the convolution in front.
__global__ __launch_bounds__(256) void norm_bwd_apply_nc8_kernel(const BwdP<bf16_t> p, uint4* __restrict__ dxn, int CB, long S8, long total) {
    const uint4* __restrict__ x4 = reinterpret_cast<const uint4*>(p.x);
    const uint4* __restrict__ gy4 = reinterpret_cast<const uint4*>(p.gy);
    const uint4* __restrict__ g4 = reinterpret_cast<const uint4*>(p.gb);
    uint4* __restrict__ d4 = reinterpret_cast<uint4*>(p.dx);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i % S8, ncb = i / S8;
        const int cb = (int)(ncb % CB), n = (int)(ncb / CB);
        unsigned r[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb * 8 + j;
            r[j][0] = r[j][1] = r[j][2] = r[j][3] = 0u;
            if (c < p.C) {
                const long plane = (long)n * p.C + c;
                const int st = p.mode == 0 ? (int)plane : c;
                const float mean = p.mean[st], invstd = p.invstd[st], c1 = p.coef[st * 2 + 0], c2 = p.coef[st * 2 + 1];
                float xv[8], gv[8], sc[8], sh[8], o[8];
                nc8_unpack8(x4[plane * S8 + g], xv);
                nc8_unpack8(gy4[plane * S8 + g], gv);
                if (p.gb) {
                    const long gpl = ((long)n * 2 * p.C + c) * S8 + g;
                    nc8_unpack8(g4[gpl], sc);
                    nc8_unpack8(g4[gpl + (long)p.C * S8], sh);
#pragma unroll
                    for (int e = 0; e < 8; ++e) sc[e] = 1.0f + sc[e];
                } else {
                    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { sc[e] = ga; sh[e] = be; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (xv[e] - mean) * invstd;
                    const float gg = gv[e] * act_grad(xh * sc[e] + sh[e], p.act, p.slope);
                    o[e] = invstd * (gg * sc[e] - c1 - xh * c2);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[j][e] = nc8_pack2(o[2 * e], o[2 * e + 1]);
                if (d4) d4[plane * S8 + g] = make_uint4(r[j][0], r[j][1], r[j][2], r[j][3]);      // (NULL: dx exists in NC8 form only)
            }
        }
        nc8_transpose_store(r, dxn + (((long)n * CB + cb) * S8 + g) * 8);
    }
}

// Instance-norm backward in ONE launch (round 5): mode 0, no affine-parameter gradient (SPADE / plain instance norm), planes of
// <= 8192 elements.  The workgroup keeps x, gy and the SPADE scale / shift of its plane in registers between the reduction
// (norm_bwd_reduce_kernel's arithmetic and summation order, one chunk) and the apply pass (norm_bwd_apply_vec_kernel's):
// reduce + finalize + apply = three launches and two reads of three tensors become one launch and one read.
template <class T>
__global__ __launch_bounds__(256) void norm_inst_bwd_fused_kernel(const BwdP<T> p) {
    constexpr int NV = NORM_CHUNK / 1024;
    __shared__ float sm[4];
    const long plane = blockIdx.x;
    const int c = (int)(plane % p.C);
    const int n = (int)(plane / p.C);
    const int S = (int)p.S;
    const float mean = p.mean[plane], invstd = p.invstd[plane];
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    const T* __restrict__ xp = p.x + plane * p.S;
    const T* __restrict__ gp = p.gy + plane * p.S;
    T* __restrict__ dp = p.dx + plane * p.S;
    const long gplane = ((long)n * 2 * p.C + c) * p.S, goff = (long)p.C * p.S;
    float xh[NV][4], gsc[NV][4], scv[NV][4];          // xhat, g' * scale (the dx term), scale
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int s = threadIdx.x * 4 + k * 1024;
        if (s < S) {
            const float4 xv = c2m_ld4(xp + s);
            const float4 gv = c2m_ld4(gp + s);
            float4 sc = make_float4(ga, ga, ga, ga), sh = make_float4(be, be, be, be);
            if (p.gb) {
                const float4 a = c2m_ld4(p.gb + gplane + s);
                sc = make_float4(1.0f + a.x, 1.0f + a.y, 1.0f + a.z, 1.0f + a.w);
                sh = c2m_ld4(p.gb + gplane + goff + s);
            }
            const float xq[4] = {(xv.x - mean) * invstd, (xv.y - mean) * invstd, (xv.z - mean) * invstd, (xv.w - mean) * invstd};
            const float sq[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, gyv[4] = {gv.x, gv.y, gv.z, gv.w};
            float g[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[e] = gyv[e] * act_grad(xq[e] * sq[e] + shv[e], p.act, p.slope);
                if (p.gb) { s1 += g[e] * sq[e]; s2 += g[e] * sq[e] * xq[e]; }
                else { s1 += g[e]; s2 += g[e] * xq[e]; }
                xh[k][e] = xq[e]; gsc[k][e] = g[e] * sq[e]; scv[k][e] = sq[e];
            }
            if (p.gb) {
                c2m_st4(p.ggb + gplane + s, make_float4(g[0] * xq[0], g[1] * xq[1], g[2] * xq[2], g[3] * xq[3]));
                c2m_st4(p.ggb + gplane + goff + s, make_float4(g[0], g[1], g[2], g[3]));
            }
        }
    }
    (void)scv;
    s1 = block_sum_256(s1, sm);
    s2 = block_sum_256(s2, sm);
    const float c1 = (float)((double)ga * (double)s1 / (double)p.S), c2 = (float)((double)ga * (double)s2 / (double)p.S);
    if (threadIdx.x == 0) { p.coef[plane * 2 + 0] = c1; p.coef[plane * 2 + 1] = c2; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int s = threadIdx.x * 4 + k * 1024;
        if (s < S) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = invstd * (gsc[k][e] - c1 - xh[k][e] * c2);
            c2m_st4(dp + s, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// Batch-norm backward of a small channel volume (N * S <= 8192) in one launch: see norm_bn_small_fused_kernel.
template <class T, int NE>
__global__ __launch_bounds__(256) void norm_bn_small_bwd_fused_kernel(const BwdP<T> p) {
    __shared__ float sm[4];
    const int c = blockIdx.x;
    const int S = (int)p.S, total = p.N * S;
    const long cstride = (long)p.C * p.S;
    const unsigned magic = (unsigned)((0x100000000ULL + (unsigned)S - 1) / (unsigned)S);      // e / S by multiply-high (see the forward kernel)
    const float mean = p.mean[c], invstd = p.invstd[c];
    const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.gamma ? p.beta[c] : 0.f;
    const T* __restrict__ xc = p.x + (long)c * p.S;
    const T* __restrict__ gc = p.gy + (long)c * p.S;
    float xh[NE], g[NE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * 256;
        xh[k] = 0.f; g[k] = 0.f;
        if (e < total) {
            const int n = S == 1 ? e : (int)__umulhi((unsigned)e, magic), q = e - n * S;
            const long i = (long)n * cstride + q;
            xh[k] = (c2m_ld(xc, i) - mean) * invstd;
            g[k] = c2m_ld(gc, i) * act_grad(xh[k] * ga + be, p.act, p.slope);
            s1 += g[k]; s2 += g[k] * xh[k];
        }
    }
    s1 = block_sum_256(s1, sm);
    s2 = block_sum_256(s2, sm);
    const double cnt = (double)total;
    const float c1 = (float)((double)ga * (double)s1 / cnt), c2 = (float)((double)ga * (double)s2 / cnt);
    if (threadIdx.x == 0) {
        p.coef[c * 2 + 0] = c1; p.coef[c * 2 + 1] = c2;
        if (p.dgamma) { p.dgamma[c] = s2; p.dbeta[c] = s1; }
    }
    T* __restrict__ dc = p.dx + (long)c * p.S;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = threadIdx.x + k * 256;
        if (e < total) {
            const int n = S == 1 ? e : (int)__umulhi((unsigned)e, magic), q = e - n * S;
            c2m_st(dc, (long)n * cstride + q, invstd * (g[k] * ga - c1 - xh[k] * c2));
        }
    }
}

// workspace floats: N*C*chunks*2 (partials) + nstat*2 (coefficients)  <= c2m_norm_workspace_floats(N, C, S)
// dx_nc8 (optional, bf16 tensors with S % 8 == 0 only): dx in the channel-blocked layout as well (see c2m_norm_apply).
C2M_API int c2m_norm_bwd(const void* x, const void* gy, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const void* gb, void* ggb, float* dgamma, float* dbeta, void* dx, void* dx_nc8,
                         float* workspace, int N, int C, long S, int mode, int act, float slope, int dt, void* stream) {
    C2M_ENTER();
    const long total = (long)N * C * S;
    if (total <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = norm_vec_ok(S, x, gy, gb, dx, dt);
    C2M_DISPATCH_DT(dt,
        BwdP<T> p;
        p.x = (const T*)x; p.gy = (const T*)gy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
        p.gb = (const T*)gb; p.ggb = (T*)ggb;
        p.chunks = norm_chunks(S);
        p.partial = workspace;
        p.coef = workspace + (long)N * C * p.chunks * 2;
        p.dgamma = dgamma; p.dbeta = dbeta; p.dx = (T*)dx;
        p.N = N; p.C = C; p.S = S; p.mode = mode; p.act = act; p.slope = slope;
        if ((norm_fused_on & 1) && mode == 0 && !dgamma && !dx_nc8 && dx && vec && p.chunks == 1 &&
            (!gb || ((((uintptr_t)gb) | ((uintptr_t)ggb)) & C2mVec4<T>::mask) == 0)) {
            hipLaunchKernelGGL(norm_inst_bwd_fused_kernel<T>, dim3((unsigned)((long)N * C)), dim3(256), 0, s, p);
            return (int)hipGetLastError();
        }
        if ((norm_fused_on & 2) && mode == 1 && !gb && !dx_nc8 && dx && (long)N * S <= 8192) {
            if ((long)N * S <= 2048) hipLaunchKernelGGL((norm_bn_small_bwd_fused_kernel<T, 8>), dim3((unsigned)C), dim3(256), 0, s, p);
            else                     hipLaunchKernelGGL((norm_bn_small_bwd_fused_kernel<T, 32>), dim3((unsigned)C), dim3(256), 0, s, p);
            return (int)hipGetLastError();
        }
        hipLaunchKernelGGL(norm_bwd_reduce_kernel<T>, dim3((unsigned)((long)N * C * p.chunks)), dim3(256), 0, s, p);
        const int nthreads = mode == 0 ? N * C : C;
        if (N * p.chunks >= 16 && (mode == 1 || dgamma))
            hipLaunchKernelGGL(norm_bwd_finalize_wave_kernel<T>, dim3(c2m_cdiv(C, 4)), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(norm_bwd_finalize_kernel<T>, dim3(c2m_cdiv(nthreads, 128)), dim3(128), 0, s, p);
        if (dx_nc8) { }
        else if (vec)
            hipLaunchKernelGGL(norm_bwd_apply_vec_kernel<T>, dim3((unsigned)((long)N * C * p.chunks)), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(norm_bwd_apply_kernel<T>, dim3(c2m_grid(total, 256)), dim3(256), 0, s, p););
    if (dx_nc8) {
        if (dt != C2M_BF16 || (S & 7) || ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gb) | ((uintptr_t)dx) | ((uintptr_t)dx_nc8)) & 15))
            return (int)hipErrorInvalidValue;
        BwdP<bf16_t> p;
        p.x = (const bf16_t*)x; p.gy = (const bf16_t*)gy; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta;
        p.gb = (const bf16_t*)gb; p.ggb = (bf16_t*)ggb; p.chunks = norm_chunks(S);
        p.partial = workspace; p.coef = workspace + (long)N * C * p.chunks * 2;
        p.dgamma = dgamma; p.dbeta = dbeta; p.dx = (bf16_t*)dx;
        p.N = N; p.C = C; p.S = S; p.mode = mode; p.act = act; p.slope = slope;
        const int CB = (C + 7) / 8;
        const long n8 = (long)N * CB * (S / 8);
        hipLaunchKernelGGL(norm_bwd_apply_nc8_kernel, dim3(c2m_grid(n8, 256)), dim3(256), 0, s, p, (uint4*)dx_nc8, CB, S / 8, n8);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- activation backward
// For conv epilogue activations (no norm in between): gradient from the OUTPUT y.
__device__ __forceinline__ float act_bwd_elem(float v, float g, int act, float slope) {
    switch (act) {
        case C2M_ACT_RELU: return v > 0.f ? g : 0.f * g;
        case C2M_ACT_LRELU: return g * (v > 0.f ? 1.f : slope);
        case C2M_ACT_SIGMOID: return g * (v * (1.f - v));
        default: return g;
    }
}

template <class T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ gy, T* __restrict__ gx, long total, int act,
                               float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        c2m_st(gx, i, act_bwd_elem(c2m_ld(y, i), c2m_ld(gy, i), act, slope));
}

template <class T>
__global__ void act_bwd_vec_kernel(const T* __restrict__ y, const T* __restrict__ gy, T* __restrict__ gx, long total4,
                                   int act, float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = c2m_ld4(y + 4 * i), g = c2m_ld4(gy + 4 * i);
        c2m_st4(gx + 4 * i, make_float4(act_bwd_elem(v.x, g.x, act, slope), act_bwd_elem(v.y, g.y, act, slope),
                                        act_bwd_elem(v.z, g.z, act, slope), act_bwd_elem(v.w, g.w, act, slope)));
    }
}

C2M_API int c2m_act_bwd(const void* y, const void* gy, void* gx, long total, int act, float slope, int dt, void* stream) {
    C2M_ENTER();
    if (total <= 0) return 0;
    const uintptr_t mask = dt == C2M_BF16 ? 7 : 15;
    const bool vec = (total & 3) == 0 && ((((uintptr_t)y) | ((uintptr_t)gy) | ((uintptr_t)gx)) & mask) == 0;
    C2M_DISPATCH_DT(dt,
        if (vec)
            hipLaunchKernelGGL(act_bwd_vec_kernel<T>, dim3(c2m_grid(total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)y, (const T*)gy, (T*)gx, total / 4, act, slope);
        else
            hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)y,
                               (const T*)gy, (T*)gx, total, act, slope););
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- gradients straight into NC8 (round 4)
// The two element-wise kernels that produce the dY a bf16 convolution's backward consumes -- the convolution's own activation
// backward (act_bwd) and the perceptual-loss tap backward of the VGG chain (losses.hip: relu_tap_bwd) -- with their result written
// in the channel-blocked layout of conv_nc8.hip INSTEAD of NCHW: the caller uses them only where every consumer of that gradient
// reads the NC8 form (ops._ConvFn.backward, ops._ConvReluTapFn.backward), so the NCHW write and the layout pass over it disappear.
// Thread = 8 channels x 8 pixels (as nchw_to_nc8_kernel); same fp32 arithmetic per element, one RNE rounding to bf16.
// MODE 0: g = act_bwd_elem(y, gy);  MODE 1: g = y > 0 ? gy + (gl / total) * sign(y - t) : 0  (gy may be null: 0).
template <int MODE>
__global__ __launch_bounds__(256) void grad_to_nc8_kernel(const uint4* __restrict__ y, const uint4* __restrict__ t,
                                                          const uint4* __restrict__ gy, const float* __restrict__ gl,
                                                          uint4* __restrict__ gn, int C, int CB, long S8, long total, long count,
                                                          int act, float slope) {
    const float lg = MODE == 1 ? gl[0] / (float)count : 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i % S8, ncb = i / S8;
        const int cb = (int)(ncb % CB); const long n = ncb / CB;
        unsigned r[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb * 8 + j;
            r[j][0] = r[j][1] = r[j][2] = r[j][3] = 0u;
            if (c < C) {
                const long q = (n * C + c) * S8 + g;
                float yv[8], gv[8], tv[8], o[8];
                nc8_unpack8(y[q], yv);
                if (gy) nc8_unpack8(gy[q], gv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] = 0.f;
                }
                if (MODE == 1) nc8_unpack8(t[q], tv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (MODE == 0) o[e] = act_bwd_elem(yv[e], gv[e], act, slope);
                    else {
                        const float d = yv[e] - tv[e];
                        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                        o[e] = yv[e] > 0.f ? gv[e] + lg * sgn : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) r[j][e] = nc8_pack2(o[2 * e], o[2 * e + 1]);
            }
        }
        nc8_transpose_store(r, gn + ((n * CB + cb) * S8 + g) * 8);
    }
}

// y, gy (and t): bf16 [N][C][S] with S % 8 == 0, 16-byte aligned; g_nc8: [N][ceil(C/8)][S][8] bf16.  mode 0: act / slope as
// c2m_act_bwd; mode 1: t, gl, count (= the element count the tap's mean ran over) as c2m_relu_tap_bwd, gy may be NULL.
C2M_API int c2m_grad_to_nc8(int mode, const void* y, const void* t, const void* gy, const float* gl, void* g_nc8, long N, int C,
                            long S, long count, int act, float slope, void* stream) {
    C2M_ENTER();
    if (N <= 0 || C <= 0 || S <= 0) return 0;
    if ((S & 7) || ((((uintptr_t)y) | ((uintptr_t)t) | ((uintptr_t)gy) | ((uintptr_t)g_nc8)) & 15) || (mode != 0 && mode != 1) ||
        (mode == 0 && !gy) || (mode == 1 && (!t || !gl || count <= 0)))
        return (int)hipErrorInvalidValue;
    const int CB = (C + 7) / 8;
    const long total = N * CB * (S / 8);
    if (mode == 0)
        hipLaunchKernelGGL(grad_to_nc8_kernel<0>, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)y,
                           (const uint4*)t, (const uint4*)gy, gl, (uint4*)g_nc8, C, CB, S / 8, total, count, act, slope);
    else
        hipLaunchKernelGGL(grad_to_nc8_kernel<1>, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)y,
                           (const uint4*)t, (const uint4*)gy, gl, (uint4*)g_nc8, C, CB, S / 8, total, count, act, slope);
    return (int)hipGetLastError();
}

