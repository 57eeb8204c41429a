// gnn.hip -- the attention of a GATv2 layer on a small dense graph as ONE launch per direction (round 5).
// Reference: torch_geometric.nn.GATv2Conv(heads = 4, concat = False, add_self_loops = False) as used by
// src/modules/motion_estimator/sparse_motion_estimator.py:104-112 (one layer per predicted frame, 24 object nodes per batch of 8).
// c2m_amd/thirdparty.py restates it on all ordered pairs with the edge-multiplicity matrix A[i][j] (edges j -> i) as mask and
// weight of the softmax; in torch ops that is ~18 launches forward and ~35 backward per layer on [N, N, H, C] temporaries -- 5
// layers per step: most of the object branch's ~600 launches.  Here: one workgroup per target node i, heads in a loop,
//   logit[j] = sum_c att[h][c] * lrelu(xl[j][h][c] + xr[i][h][c])        (a wave per source node j, xor-shuffle sum: fixed order)
//   alpha[j] = A[i][j] * exp(logit[j] - max_j) / (sum_j .. + 1e-16)       (one wave; rows without edges give alpha = 0)
//   out[i][c] = 1/H * sum_h sum_j alpha[j] * xl[j][h][c]                  (a thread per channel, j in order)
// Backward: the same workgroup layout writes d(xr)[i], and per-target partials of d(xl) and d(att) that a second launch sums over
// i in index order -- no atomics, bit-repeatable.
#include "common.h"

#define GAT_MAX_N 64
#define GAT_SLOTS 4            // channels per thread: C <= 256 * GAT_SLOTS

static __device__ __forceinline__ float gat_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
static __device__ __forceinline__ float gat_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void gat_dense_fwd_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                            const float* __restrict__ att, const float* __restrict__ A,
                                                            float* __restrict__ out, float* __restrict__ alpha, int N, int H, int C,
                                                            float slope) {
    __shared__ float logit[GAT_MAX_N], al[GAT_MAX_N];
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float acc[GAT_SLOTS];
#pragma unroll
    for (int s = 0; s < GAT_SLOTS; ++s) acc[s] = 0.f;
    for (int h = 0; h < H; ++h) {
        const float* xri = xr + ((long)i * H + h) * C;
        const float* ah = att + (long)h * C;
        for (int j = wave; j < N; j += 4) {
            const float* xlj = xl + ((long)j * H + h) * C;
            float part = 0.f;
            for (int c = lane; c < C; c += 64) {
                float v = xlj[c] + xri[c];
                v = v > 0.f ? v : v * slope;
                part += ah[c] * v;
            }
            part = gat_wave_sum(part);
            if (lane == 0) logit[j] = part;
        }
        __syncthreads();
        if (wave == 0) {
            const float a = lane < N ? A[(long)i * N + lane] : 0.f;
            const float l = a != 0.f ? logit[lane] : -INFINITY;
            float mx = gat_wave_max(l);
            if (mx == -INFINITY) mx = 0.f;                    // a node without incoming edges
            const float ex = a != 0.f ? a * expf(l - mx) : 0.f;
            const float den = gat_wave_sum(ex) + 1e-16f;
            const float v = ex / den;
            if (lane < N) {
                al[lane] = v;
                alpha[((long)i * N + lane) * H + h] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GAT_SLOTS; ++s) {
            const int c = tid + s * 256;
            if (c < C) {
                float sum = 0.f;
                for (int j = 0; j < N; ++j) sum += al[j] * xl[((long)j * H + h) * C + c];
                acc[s] += sum;
            }
        }
        __syncthreads();
    }
    const float inv = 1.f / (float)H;
#pragma unroll
    for (int s = 0; s < GAT_SLOTS; ++s) {
        const int c = tid + s * 256;
        if (c < C) out[(long)i * C + c] = acc[s] * inv;
    }
}

// per target node i: d(xr)[i][h][c]; partials P_xl[i][j][h][c] (message + logit terms that node j receives from target i) and
// P_att[i][h][c]
__global__ __launch_bounds__(256) void gat_dense_bwd_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                            const float* __restrict__ att, const float* __restrict__ alpha,
                                                            const float* __restrict__ gout, float* __restrict__ dxr,
                                                            float* __restrict__ pxl, float* __restrict__ patt, int N, int H, int C,
                                                            float slope) {
    __shared__ float da[GAT_MAX_N], dl[GAT_MAX_N], al[GAT_MAX_N];
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float inv = 1.f / (float)H;
    const float* gi = gout + (long)i * C;
    for (int h = 0; h < H; ++h) {
        for (int j = wave; j < N; j += 4) {                   // d(alpha)[j] = sum_c g[i][c] / H * xl[j][h][c]
            const float* xlj = xl + ((long)j * H + h) * C;
            float part = 0.f;
            for (int c = lane; c < C; c += 64) part += gi[c] * inv * xlj[c];
            part = gat_wave_sum(part);
            if (lane == 0) da[j] = part;
        }
        __syncthreads();
        if (wave == 0) {                                      // softmax backward: d(logit) = alpha * (d(alpha) - sum_k alpha_k d(alpha)_k)
            const float a = lane < N ? alpha[((long)i * N + lane) * H + h] : 0.f;
            const float d = lane < N ? da[lane] : 0.f;
            const float s = gat_wave_sum(a * d);
            if (lane < N) {
                al[lane] = a;
                dl[lane] = a * (d - s);
            }
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < GAT_SLOTS; ++sl) {
            const int c = tid + sl * 256;
            if (c < C) {
                const float xrv = xr[((long)i * H + h) * C + c], at = att[(long)h * C + c], g = gi[c] * inv;
                float sxr = 0.f, satt = 0.f;
                for (int j = 0; j < N; ++j) {
                    const float pre = xl[((long)j * H + h) * C + c] + xrv;
                    const float e = pre > 0.f ? pre : pre * slope;
                    const float d = dl[j] * at * (pre > 0.f ? 1.f : slope);
                    sxr += d;
                    satt += dl[j] * e;
                    pxl[(((long)i * N + j) * H + h) * C + c] = al[j] * g + d;
                }
                dxr[((long)i * H + h) * C + c] = sxr;
                patt[((long)i * H + h) * C + c] = satt;
            }
        }
        __syncthreads();
    }
}

// d(xl)[j][h][c] = sum_i P_xl[i][j][h][c];  d(att)[h][c] = sum_i P_att[i][h][c]   (i in index order)
__global__ __launch_bounds__(256) void gat_dense_bwd_sum_kernel(const float* __restrict__ pxl, const float* __restrict__ patt,
                                                                float* __restrict__ dxl, float* __restrict__ datt, int N, long nhc,
                                                                long hc) {
    const long total = nhc + hc;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        if (e < nhc) {
            for (int i = 0; i < N; ++i) s += pxl[(long)i * nhc + e];
            dxl[e] = s;
        } else {
            const long k = e - nhc;
            for (int i = 0; i < N; ++i) s += patt[(long)i * hc + k];
            datt[k] = s;
        }
    }
}

static int gat_ok(int N, int H, int C) { return N >= 1 && N <= GAT_MAX_N && H >= 1 && C >= 1 && C <= 256 * GAT_SLOTS; }

// xl, xr [N][H][C] (lin_l(x), lin_r(x)), att [H][C], A [N][N] edge multiplicities (j -> i at A[i][j]);
// out [N][C] (heads averaged, WITHOUT the layer's bias), alpha [N][N][H] (kept for the backward).  N <= 64, C <= 1024.
C2M_API int c2m_gat_dense_fwd(const float* xl, const float* xr, const float* att, const float* A, float* out, float* alpha, int N,
                              int H, int C, float slope, void* stream) {
    if (!gat_ok(N, H, C) || !xl || !xr || !att || !A || !out || !alpha) return 1;
    hipLaunchKernelGGL(gat_dense_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, xl, xr, att, A, out, alpha, N, H, C, slope);
    return (int)hipGetLastError();
}

C2M_API long c2m_gat_dense_bwd_workspace_floats(int N, int H, int C) { return (long)N * N * H * C + (long)N * H * C; }

// gout [N][C] -> dxl, dxr [N][H][C], datt [H][C]; workspace: c2m_gat_dense_bwd_workspace_floats floats (no zero-initialisation)
C2M_API int c2m_gat_dense_bwd(const float* xl, const float* xr, const float* att, const float* alpha, const float* gout, float* dxl,
                              float* dxr, float* datt, float* workspace, int N, int H, int C, float slope, void* stream) {
    if (!gat_ok(N, H, C) || !xl || !xr || !att || !alpha || !gout || !dxl || !dxr || !datt || !workspace) return 1;
    float* pxl = workspace;
    float* patt = workspace + (long)N * N * H * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gat_dense_bwd_kernel, dim3(N), dim3(256), 0, st, xl, xr, att, alpha, gout, dxr, pxl, patt, N, H, C, slope);
    const long nhc = (long)N * H * C, hc = (long)H * C;
    hipLaunchKernelGGL(gat_dense_bwd_sum_kernel, dim3(c2m_grid(nhc + hc, 256)), dim3(256), 0, st, pxl, patt, dxl, datt, N, nhc, hc);
    return (int)hipGetLastError();
}
