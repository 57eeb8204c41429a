// The three custom operators of the online target-flow path: FlowNet2's Resample2d, ChannelNorm and
// Correlation, forward only -- the reference runs its flow network frozen under torch.no_grad()
// (src/modules/third_party/flow_net/flow_net.py:32,66-67), so no backward of these ops is ever taken on a C2M path.
// Reference kernels (CUDA, one thread per output element / one 32-lane warp per output pixel with padded NHWC copies):
//   resample2d/src/resample2d_kernel.cu:16-75, channelnorm/src/channelnorm_kernel.cu:19-62,
//   correlation/src/correlation_cuda_kernel.cu:47-147 (+ host launch :325-420, output size correlation_cuda.cc:25-38).
// Written for gfx950: plain NCHW in, NCHW out, x fastest over the lanes of a wave (coalesced 256-byte rows), zero padding
// resolved by bounds checks (no padded channel-last copies of the inputs), channel sums in a fixed sequential order --
// the order oracle/c2m_oracle_index.c::oracle_correlation / oracle_channelnorm use, so HIP and oracle agree bit for bit.
#include "common.h"

// ------------------------------------------------------------------------------------------------ Resample2d
// out[b,c,y,x] = bilinear(img[b,c], x + flow[b,0,y,x], y + flow[b,1,y,x]); pixel-unit flow, taps clamped to the border.
// The reference forms the first three weight products in double (its `1.` literals: (1. - alpha) * (1. - beta) * v), rounds
// each to float, forms the fourth ((alpha)*(beta) * v) in float, and adds the four terms in the order TL, TR, BL, BR
// (resample2d_kernel.cu:58-63, kernel_size = 1): reproduced literally, without contraction (-ffp-contract=off).
__global__ void resample2d_fwd_kernel(const float* __restrict__ img, const float* __restrict__ flow, float* __restrict__ out,
                                      int N, int C, int H, int W) {
    const long HW = (long)H * W, total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const long r = i / W;
        const int y = (int)(r % H); const long b = r / H;
        const float dx = flow[(b * 2 + 0) * HW + (long)y * W + x], dy = flow[(b * 2 + 1) * HW + (long)y * W + x];
        const float xf = (float)x + dx, yf = (float)y + dy;
        const float fxf = floorf(xf), fyf = floorf(yf);
        const float alpha = xf - fxf, beta = yf - fyf;
        const int xL = max(min((int)fxf, W - 1), 0), xR = max(min((int)(fxf + 1.0f), W - 1), 0);
        const int yT = max(min((int)fyf, H - 1), 0), yB = max(min((int)(fyf + 1.0f), H - 1), 0);
        const double wTL = (1. - alpha) * (1. - beta), wTR = (double)alpha * (1. - beta);
        const double wBL = (1. - alpha) * (double)beta;
        const float wBR = alpha * beta;         // `(alpha)*(beta) * v` has no double literal: evaluated in float, left to right
        for (int c = 0; c < C; ++c) {
            const float* __restrict__ p = img + (b * C + c) * HW;
            float val = 0.0f;
            val += (float)(wTL * (double)p[(long)yT * W + xL]);
            val += (float)(wTR * (double)p[(long)yT * W + xR]);
            val += (float)(wBL * (double)p[(long)yB * W + xL]);
            val += wBR * p[(long)yB * W + xR];
            out[(b * C + c) * HW + (long)y * W + x] = val;
        }
    }
}

C2M_API int c2m_resample2d_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream) {
    C2M_ENTER();
    if ((long)N * C * H * W <= 0) return 0;
    hipLaunchKernelGGL(resample2d_fwd_kernel, dim3(c2m_grid((long)N * H * W, 256)), dim3(256), 0, (hipStream_t)stream, img,
                       flow, out, N, C, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ChannelNorm
// out[b,0,y,x] = sqrt(sum_c x[b,c,y,x]^2), channels summed in ascending order (channelnorm_kernel.cu:54-61).
__global__ void channelnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int C, long HW) {
    const long total = (long)N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, sp = i - b * HW;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = x[(b * C + c) * HW + sp];
            acc += v * v;                                         // (-ffp-contract=off: the product is rounded first)
        }
        out[i] = sqrtf(acc);
    }
}

C2M_API int c2m_channelnorm_fwd(const float* x, float* out, int N, int C, int H, int W, void* stream) {
    C2M_ENTER();
    if ((long)N * H * W <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(channelnorm_fwd_kernel, dim3(c2m_grid((long)N * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       out, N, C, (long)H * W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Correlation
// FlowNetC cost volume (multiplicative): for displacement (tj, ti) in [-d, d]^2, d = max_displacement / stride2,
//   out[n, (tj+d)*D + (ti+d), oy, ox] = 1/(k*k*C) * sum_{j,i in patch} sum_c in1[n,c,y1+j,x1+i] * in2[n,c,y1+tj*s2+j,x1+ti*s2+i]
// with y1 = oy*stride1 + max_displacement - pad (zero outside the images), D = 2d+1.
// Output size (correlation_cuda.cc:25-38): oH = ceil((H + 2*pad - 2*(kernel_rad + max_displacement)) / stride1).
// One thread per output element, x fastest: in1 rows are read coalesced, in2 rows coalesced at a lane-uniform shift.
// Summation order: patch row j, patch column i, channel c ascending (the CUDA kernel splits c over 32 lanes and
// tree-reduces: another fp32 rounding of the same sum; parity for this third-party op is restatement-only).
struct CorrP {
    const float* in1; const float* in2; float* out;
    int N, C, H, W, oH, oW, pad, krad, maxd, s1, s2, D;
};

__global__ void correlation_fwd_kernel(const CorrP p) {
    const long HW = (long)p.H * p.W, oHW = (long)p.oH * p.oW;
    const long total = (long)p.N * p.D * p.D * oHW;
    const int drad = p.D / 2;
    const float nelems = (float)((2 * p.krad + 1) * (2 * p.krad + 1) * p.C);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(e % p.oW); long r = e / p.oW;
        const int oy = (int)(r % p.oH); r /= p.oH;
        const int tc = (int)(r % (p.D * p.D)); const long n = r / (p.D * p.D);
        const int tj = tc / p.D - drad, ti = tc % p.D - drad;
        const int y1 = oy * p.s1 + p.maxd - p.pad, x1 = ox * p.s1 + p.maxd - p.pad;
        const int y2 = y1 + tj * p.s2, x2 = x1 + ti * p.s2;
        const float* __restrict__ a = p.in1 + n * p.C * HW;
        const float* __restrict__ b = p.in2 + n * p.C * HW;
        float acc = 0.0f;
        for (int j = -p.krad; j <= p.krad; ++j)
            for (int i = -p.krad; i <= p.krad; ++i) {
                const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
                const bool ok = (unsigned)ya < (unsigned)p.H && (unsigned)xa < (unsigned)p.W &&
                                (unsigned)yb < (unsigned)p.H && (unsigned)xb < (unsigned)p.W;
                if (!ok) continue;                                 // a zero-padded operand: every product of this tap is 0
                const long oa = (long)ya * p.W + xa, ob = (long)yb * p.W + xb;
                for (int c = 0; c < p.C; ++c) acc += a[c * HW + oa] * b[c * HW + ob];
            }
        p.out[e] = acc / nelems;
    }
}

C2M_API int c2m_correlation_out_size(int H, int pad, int kernel_size, int max_displacement, int stride1) {
    const int border = (kernel_size - 1) / 2 + max_displacement;
    const int span = H + 2 * pad - 2 * border;
    return span <= 0 ? 0 : (span + stride1 - 1) / stride1;
}

C2M_API int c2m_correlation_fwd(const float* in1, const float* in2, float* out, int N, int C, int H, int W, int pad,
                                int kernel_size, int max_displacement, int stride1, int stride2, void* stream) {
    C2M_ENTER();
    if (kernel_size < 1 || !(kernel_size & 1) || stride1 < 1 || stride2 < 1 || max_displacement < 0 || pad < 0)
        return (int)hipErrorInvalidValue;
    CorrP p;
    p.in1 = in1; p.in2 = in2; p.out = out;
    p.N = N; p.C = C; p.H = H; p.W = W; p.pad = pad; p.krad = (kernel_size - 1) / 2; p.maxd = max_displacement;
    p.s1 = stride1; p.s2 = stride2; p.D = 2 * (max_displacement / stride2) + 1;
    p.oH = c2m_correlation_out_size(H, pad, kernel_size, max_displacement, stride1);
    p.oW = c2m_correlation_out_size(W, pad, kernel_size, max_displacement, stride1);
    const long total = (long)N * p.D * p.D * p.oH * p.oW;
    if (total <= 0 || C <= 0) return 0;
    hipLaunchKernelGGL(correlation_fwd_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ bias + LeakyReLU
// y = lrelu(x + bias[c]) in place: the epilogue of FlowNet2's transposed convolutions (submodules.py:75-80), whose matrix
// part runs on the data-gradient kernels of conv_igemm.hip.
__global__ void bias_act_kernel(float* __restrict__ x, const float* __restrict__ bias, long total, long HW, int C, int act,
                                float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        x[i] = c2m_act(x[i] + (bias ? bias[c] : 0.0f), act, slope);
    }
}

C2M_API int c2m_bias_act(float* x, const float* bias, long N, int C, long HW, int act, float slope, void* stream) {
    C2M_ENTER();
    const long total = N * C * HW;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(bias_act_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, bias, total, HW, C,
                       act, slope);
    return (int)hipGetLastError();
}
