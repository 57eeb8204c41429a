// gnn.hip -- the attention of a GATv2 layer on a small dense graph as ONE launch per direction (round 5).
// Reference: torch_geometric.nn.GATv2Conv(heads = 4, concat = False, add_self_loops = False) as used by
// src/modules/motion_estimator/sparse_motion_estimator.py:104-112 (one layer per predicted frame, 24 object nodes per batch of 8).
// c2m_amd/thirdparty.py restates it on all ordered pairs with the edge-multiplicity matrix A[i][j] (edges j -> i) as mask and
// weight of the softmax; in torch ops that is ~18 launches forward and ~35 backward per layer on [N, N, H, C] temporaries -- 5
// layers per step: most of the object branch's ~600 launches.  Here: one workgroup per target node i, a wave per head,
//   logit[j] = sum_c att[h][c] * lrelu(xl[j][h][c] + xr[i][h][c])        (xor-shuffle sum over the wave: fixed order; kept by lane j)
//   alpha[j] = A[i][j] * exp(logit[j] - max_j) / (sum_j .. + 1e-16)       (in the wave; rows without edges give alpha = 0)
//   out[i][c] = 1/H * sum_h sum_j alpha[j] * xl[j][h][c]                  (j in order, heads summed through LDS in index order)
// Backward: the same workgroup layout writes d(xr)[i], and per-target partials of d(xl) and d(att) that a second launch sums over
// i in index order -- no atomics, bit-repeatable.
#include "common.h"

#define GAT_MAX_N 64
#define GAT_SLOTS 4            // channels per lane: 4 * GAT_SLOTS slots of 64 -> C <= 1024
#define GAT_JB 4               // source nodes per load batch

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

// One workgroup per target node i; wave w takes heads w, w + 4, ...: everything of a head stays inside one wave (no barriers in
// the head loop; lane j holds logit / alpha / d(logit) of source node j, handed around by shuffles), the heads are averaged through
// LDS at the end.  NS = channel slots of a lane (channel = lane + 64 s): every load is UNCONDITIONAL on a clamped index and masked
// afterwards -- a load under a branch waits for itself (the first version took 80 us for 1.2 M multiply-adds: ~250 cycles per
// load) -- and source nodes go in batches of GAT_JB, so NS * GAT_JB loads are in flight per lane.
template <int NS>
__global__ __launch_bounds__(256) void gat_dense_fwd_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                            const float* __restrict__ att, const float* __restrict__ A,
                                                            float* __restrict__ out, float* __restrict__ alpha, int N, int H, int C,
                                                            float slope) {
    __shared__ float heads[4][64 * NS];                         // per-wave sums over its heads
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int cc[NS];
    float m[NS], acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = lane + s * 64;
        cc[s] = c < C ? c : C - 1;
        m[s] = c < C ? 1.f : 0.f;
        acc[s] = 0.f;
    }
    const float a = lane < N ? A[(long)i * N + lane] : 0.f;
    for (int h = wave; h < H; h += 4) {
        const float* xri = xr + ((long)i * H + h) * C;
        const float* ah = att + (long)h * C;
        float xrv[NS], av[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xrv[s] = xri[cc[s]];
            av[s] = ah[cc[s]] * m[s];                           // masked once: invalid channels contribute 0 to the logits
        }
        float mylogit = 0.f;
        for (int j0 = 0; j0 < N; j0 += GAT_JB) {
            float xv[GAT_JB][NS];
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const float* xlj = xl + ((long)min(j0 + b, N - 1) * H + h) * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) xv[b][s] = xlj[cc[s]];
            }
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float v = xv[b][s] + xrv[s];
                    v = v > 0.f ? v : v * slope;
                    part += av[s] * v;
                }
                part = gat_wave_sum(part);
                if (lane == j0 + b) mylogit = part;
            }
        }
        const float l = a != 0.f ? mylogit : -INFINITY;
        float mx = gat_wave_max(l);
        if (mx == -INFINITY) mx = 0.f;                          // a node without incoming edges
        const float ex = a != 0.f ? a * expf(l - mx) : 0.f;
        const float den = gat_wave_sum(ex) + 1e-16f;
        const float al = ex / den;
        if (lane < N) alpha[((long)i * N + lane) * H + h] = al;
        for (int j0 = 0; j0 < N; j0 += GAT_JB) {
            float xv[GAT_JB][NS];
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const float* xlj = xl + ((long)min(j0 + b, N - 1) * H + h) * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) xv[b][s] = xlj[cc[s]];
            }
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const float aj = j0 + b < N ? __shfl(al, min(j0 + b, N - 1), 64) : 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s] += aj * xv[b][s];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) heads[wave][lane + s * 64] = acc[s];
    __syncthreads();
    const float inv = 1.f / (float)H;
    for (int c = tid; c < C; c += 256) out[(long)i * C + c] = (heads[0][c] + heads[1][c] + heads[2][c] + heads[3][c]) * inv;
}

// per target node i: d(xr)[i][h][c]; partials P_xl[i][j][h][c] (message + logit terms that node j receives from target i) and
// P_att[i][h][c].  Same layout: a wave per head.
template <int NS>
__global__ __launch_bounds__(256) void gat_dense_bwd_kernel(const float* __restrict__ xl, const float* __restrict__ xr,
                                                            const float* __restrict__ att, const float* __restrict__ alpha,
                                                            const float* __restrict__ gout, float* __restrict__ dxr,
                                                            float* __restrict__ pxl, float* __restrict__ patt, int N, int H, int C,
                                                            float slope) {
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float inv = 1.f / (float)H;
    int cc[NS];
    bool ok[NS];
    float g[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = lane + s * 64;
        ok[s] = c < C;
        cc[s] = ok[s] ? c : C - 1;
        g[s] = ok[s] ? gout[(long)i * C + cc[s]] * inv : 0.f;
    }
    for (int h = wave; h < H; h += 4) {
        float myda = 0.f;                                       // d(alpha)[j] = sum_c g[i][c] / H * xl[j][h][c], kept by lane j
        for (int j0 = 0; j0 < N; j0 += GAT_JB) {
            float xv[GAT_JB][NS];
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const float* xlj = xl + ((long)min(j0 + b, N - 1) * H + h) * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) xv[b][s] = xlj[cc[s]];
            }
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                float part = 0.f;
#pragma unroll
                for (int s = 0; s < NS; ++s) part += g[s] * xv[b][s];
                part = gat_wave_sum(part);
                if (lane == j0 + b) myda = part;
            }
        }
        // softmax backward: d(logit) = alpha * (d(alpha) - sum_k alpha_k d(alpha)_k)
        const float al = lane < N ? alpha[((long)i * N + lane) * H + h] : 0.f;
        const float ssum = gat_wave_sum(al * myda);
        const float dl = al * (myda - ssum);
        float xrv[NS], av[NS], sxr[NS], satt[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xrv[s] = xr[((long)i * H + h) * C + cc[s]];
            av[s] = att[(long)h * C + cc[s]];
            sxr[s] = satt[s] = 0.f;
        }
        for (int j0 = 0; j0 < N; j0 += GAT_JB) {
            float xv[GAT_JB][NS];
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const float* xlj = xl + ((long)min(j0 + b, N - 1) * H + h) * C;
#pragma unroll
                for (int s = 0; s < NS; ++s) xv[b][s] = xlj[cc[s]];
            }
#pragma unroll
            for (int b = 0; b < GAT_JB; ++b) {
                const int j = j0 + b;
                if (j < N) {                                    // (uniform)
                    const float dlj = __shfl(dl, j, 64), alj = __shfl(al, j, 64);
                    float* pj = pxl + (((long)i * N + j) * H + h) * C;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const float pre = xv[b][s] + xrv[s];
                        const float e = pre > 0.f ? pre : pre * slope;
                        const float d = dlj * av[s] * (pre > 0.f ? 1.f : slope);
                        sxr[s] += d;
                        satt[s] += dlj * e;
                        if (ok[s]) pj[lane + s * 64] = alj * g[s] + d;
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (ok[s]) {
                dxr[((long)i * H + h) * C + lane + s * 64] = sxr[s];
                patt[((long)i * H + h) * C + lane + s * 64] = satt[s];
            }
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
    hipStream_t st = (hipStream_t)stream;
#define GAT_FWD(NS) hipLaunchKernelGGL(gat_dense_fwd_kernel<NS>, dim3(N), dim3(256), 0, st, xl, xr, att, A, out, alpha, N, H, C, slope)
    const int ns = (C + 63) / 64;
    if (ns <= 1) GAT_FWD(1); else if (ns <= 2) GAT_FWD(2); else if (ns <= 4) GAT_FWD(4); else if (ns <= 8) GAT_FWD(8); else GAT_FWD(16);
#undef GAT_FWD
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
#define GAT_BWD(NS) hipLaunchKernelGGL(gat_dense_bwd_kernel<NS>, dim3(N), dim3(256), 0, st, xl, xr, att, alpha, gout, dxr, pxl, patt, N, H, C, slope)
    const int ns = (C + 63) / 64;
    if (ns <= 1) GAT_BWD(1); else if (ns <= 2) GAT_BWD(2); else if (ns <= 4) GAT_BWD(4); else if (ns <= 8) GAT_BWD(8); else GAT_BWD(16);
#undef GAT_BWD
    const long nhc = (long)N * H * C, hc = (long)H * C;
    hipLaunchKernelGGL(gat_dense_bwd_sum_kernel, dim3(c2m_grid(nhc + hc, 256)), dim3(256), 0, st, pxl, patt, dxl, datt, N, nhc, hc);
    return (int)hipGetLastError();
}
