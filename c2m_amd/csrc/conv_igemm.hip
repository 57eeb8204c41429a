// Implicit-GEMM convolution for gfx950 (CDNA4) on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, exact f32).
//
// Replaces the ATen conv2d/conv3d (+ explicit ReflectionPad2d/3d) calls of the reference blocks:
//   src/modules/layers/down_block.py:14-23,35-47  same_block.py:14-23,36-46,55-67  up_block.py:9-13
//   residual_block.py:13-31,42-71  spade_block.py:47-49  vgg.py (torchvision features)  generator.py:76-78
//
// One kernel serves forward and data-gradient:  D[m][pix] = sum_k A[m][k] * G(k, pix)
//   * A  : row-major [M][lda] weights (native [Cout][Cin*taps] for forward, a packed transpose for dgrad)
//   * G  : gathered input: k -> (channel, tap) through a small host-built table (KEntry), pix -> (n, ot, oy, ox);
//          input coord = o*stride + tap_offset, zero or reflect boundary handled in the gather (no padded copy).
//   * D  : written with arbitrary output strides (NCHW / NCTHW, or the strided parity classes of a dgrad).
// Orientation is chosen so that the MFMA column index (lane&31) is the pixel: every accumulator register is
// stored as 2 x 128 contiguous bytes per wave -> coalesced NCHW stores; bias/activation fused in the epilogue.
//
// wgrad:  dW[co][j] = sum_pix dY[co][pix] * G(j, pix)  with split-K over pixels into deterministic slabs
// (no float atomics), an optional all-ones row giving the bias gradient, then a fixed-order slab reduction.
//
// Tiling: 256 threads = 4 waves, BK = 16, register-prefetched + double-buffered LDS, one barrier per K-step.
// The f32 MFMA takes 64 cycles per 32x32x2 issue, so LDS traffic (4 ds_read_b32 per 4 MFMAs) and the gather
// address arithmetic (mostly scalar: the k-row is wave-uniform) sit in its shadow.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvP {
    const float* A;
    const float* X;
    float* Y;
    const float* bias;   // per-row (may be null)
    const int4* ktab;    // [K] : {channel offset (<0: -1 zero row, -2 ones row), t, y, x tap offsets}
    int M, K, lda;       // K is a multiple of 16 (table padded with zero rows)
    int Npix, To, Ho, Wo;
    int Ti, Hi, Wi;
    int st, sh, sw;
    long in_sn, in_st, in_sh;
    long out_sn, out_sc, out_st, out_sh, out_sw, out_off;
    int act;
    float slope;
};

struct GatherDims { int Ti, Hi, Wi; long in_st, in_sh; };

template <bool REFLECT, bool IS3D>
__device__ __forceinline__ float gather_one(const float* __restrict__ Xn, const int4 e, int ots, int oys, int oxs,
                                            const GatherDims& p) {
    if (e.x < 0) return e.x == -2 ? 1.f : 0.f;  // wave-uniform branch (k-row is uniform)
    int it = 0, iy = oys + e.z, ix = oxs + e.w;
    if (IS3D) it = ots + e.y;
    bool ok = true;
    if (REFLECT) {
        if (IS3D) { it = it < 0 ? -it : it; it = it >= p.Ti ? 2 * p.Ti - 2 - it : it; }
        iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
        ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
    } else {
        ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        if (IS3D) ok = ok && (unsigned)it < (unsigned)p.Ti;
    }
    float v = 0.f;
    if (ok) {
        long off = (long)e.x + (long)iy * p.in_sh + ix;
        if (IS3D) off += (long)it * p.in_st;
        v = Xn[off];
    }
    return v;
}

__device__ __forceinline__ void decompose_pix(int pix, const ConvP& p, int& n, int& ot, int& oy, int& ox) {
    ox = pix % p.Wo; int r = pix / p.Wo;
    oy = r % p.Ho;   r = r / p.Ho;
    ot = r % p.To;   n = r / p.To;
}

template <int BM, int BN, int WGM, int WGN, bool REFLECT, bool IS3D>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr int BK = 16;
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int LDA_S = BM + 4, LDB_S = BN;
    constexpr int BROWS = 256 / BN;           // k rows gathered per pass
    constexpr int BPASS = BK / BROWS;         // gathers per thread per K-step
    constexpr int A_F4 = BM * BK / 4;         // float4 loads per K-step (whole block)
    constexpr int APASS = (A_F4 + 255) / 256;
    static_assert(WGM * WGN == 4 && BN >= 64 && BROWS >= 1 && BK % BROWS == 0, "tile");
    __shared__ float sA[2][BK][LDA_S];
    __shared__ float sB[2][BK][LDB_S];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // ---- gather side: this thread owns one pixel column of the tile
    const int bp = tid % BN;
    const int brow0 = __builtin_amdgcn_readfirstlane(tid / BN);
    int pn, pt, py, px;
    {
        int pix = n0 + bp; pix = pix < p.Npix ? pix : p.Npix - 1;
        decompose_pix(pix, p, pn, pt, py, px);
    }
    const float* __restrict__ Xn = p.X + (long)pn * p.in_sn;
    const int ots = pt * p.st, oys = py * p.sh, oxs = px * p.sw;
    const GatherDims gd{p.Ti, p.Hi, p.Wi, p.in_st, p.in_sh};

    // ---- weight side: float4 along k
    const int akq = (tid & 3) * 4;
    const int arow = tid >> 2;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[APASS];
    float rb[BPASS];
    const int nk = p.K / BK;

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int s = 0; s < APASS; ++s) {
            int r = arow + s * 64;
            if (A_F4 >= 256 || r < BM) {
                int row = m0 + r; row = row < p.M ? row : p.M - 1;
                ra[s] = *reinterpret_cast<const float4*>(p.A + (long)row * p.lda + k0 + akq);
            }
        }
#pragma unroll
        for (int s = 0; s < BPASS; ++s) {
            const int4 e = p.ktab[k0 + brow0 + s * BROWS];
            rb[s] = gather_one<REFLECT, IS3D>(Xn, e, ots, oys, oxs, gd);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int s = 0; s < APASS; ++s) {
            int r = arow + s * 64;
            if (A_F4 >= 256 || r < BM) {
                sA[buf][akq + 0][r] = ra[s].x; sA[buf][akq + 1][r] = ra[s].y;
                sA[buf][akq + 2][r] = ra[s].z; sA[buf][akq + 3][r] = ra[s].w;
            }
        }
#pragma unroll
        for (int s = 0; s < BPASS; ++s) sB[buf][brow0 + s * BROWS][bp] = rb[s];
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_tile((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = kk * 2 + (lane >> 5);
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = sA[cur][krow][wm * TM + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = sB[cur][krow][wn * TN + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: acc[i][j][r] -> row m = ..(r&3)+8*(r>>2)+4*(lane>>5), col pix = ..(lane&31)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int pix = n0 + wn * TN + j * 32 + (lane & 31);
        if (pix >= p.Npix) continue;
        int n, ot, oy, ox;
        decompose_pix(pix, p, n, ot, oy, ox);
        float* __restrict__ yb = p.Y + p.out_off + (long)n * p.out_sn + (long)ot * p.out_st + (long)oy * p.out_sh +
                                 (long)ox * p.out_sw;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) {
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[row];
                    v = c2m_act(v, p.act, p.slope);
                    yb[(long)row * p.out_sc] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_igemm(const ConvP& p, int reflect, int is3d, hipStream_t s) {
    dim3 grid(c2m_cdiv(p.Npix, BN), c2m_cdiv(p.M, BM));
    if (reflect) {
        if (is3d) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, true, true>), grid, dim3(256), 0, s, p);
        else      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, true, false>), grid, dim3(256), 0, s, p);
    } else {
        if (is3d) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, false, true>), grid, dim3(256), 0, s, p);
        else      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, false, false>), grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

// geom[] layout (int64): see include/c2m_hip.h  (C2M_CG_*)
C2M_API int c2m_conv_igemm(const float* A, const float* X, float* Y, const float* bias, const int* ktab,
                           const int64_t* g, int act, float slope, void* stream) {
    C2M_ENTER();
    ConvP p;
    p.A = A; p.X = X; p.Y = Y; p.bias = bias; p.ktab = reinterpret_cast<const int4*>(ktab);
    p.M = (int)g[0]; p.K = (int)g[1]; p.lda = (int)g[2];
    p.Npix = (int)g[3]; p.To = (int)g[4]; p.Ho = (int)g[5]; p.Wo = (int)g[6];
    p.Ti = (int)g[7]; p.Hi = (int)g[8]; p.Wi = (int)g[9];
    p.st = (int)g[10]; p.sh = (int)g[11]; p.sw = (int)g[12];
    p.in_sn = g[13]; p.in_st = g[14]; p.in_sh = g[15];
    p.out_sn = g[16]; p.out_sc = g[17]; p.out_st = g[18]; p.out_sh = g[19]; p.out_sw = g[20]; p.out_off = g[21];
    const int reflect = (int)g[22], is3d = (int)g[23];
    p.act = act; p.slope = slope;
    if (p.M <= 0 || p.Npix <= 0) return 0;
    if (p.K <= 0 || (p.K & 15) || (p.lda & 3) || (((uintptr_t)A) & 15)) return (int)hipErrorInvalidValue;
    if (reflect && ((is3d && p.Ti < 2) || p.Hi < 2 || p.Wi < 2)) {
        // reflect with an extent of 1 is only legal when no tap leaves the tensor; host guarantees that
    }
    hipStream_t s = (hipStream_t)stream;
    if (p.M <= 32)      return launch_igemm<32, 256, 1, 4>(p, reflect, is3d, s);
    else if (p.M <= 64) return launch_igemm<64, 128, 2, 2>(p, reflect, is3d, s);
    else                return launch_igemm<128, 128, 2, 2>(p, reflect, is3d, s);
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradP {
    const float* dY;     // [N][M][pix_per_image] contiguous
    const float* X;
    float* slab;         // [S][M][J]
    const int4* jtab;    // [Jpad]
    int M, J, Jpad;      // J = Cin*taps (+1 when the ones row is appended)
    int Npix, To, Ho, Wo, Ti, Hi, Wi, st, sh, sw;
    long in_sn, in_st, in_sh;
    long dy_sn, dy_sc;   // dY strides (pix stride 1)
    int pix_per_split;   // multiple of 64
};

template <int BM, int BN, int WGM, int WGN, bool REFLECT, bool IS3D>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradP p) {
    constexpr int BK = 64;                 // pixels per K-step (one wave-width: coalesced along pix)
    constexpr int LDS_S = BK + 1;          // odd stride: conflict-free fragment reads
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int AROWS = BM / 4, BROWSW = BN / 4;   // rows per wave
    __shared__ float sA[BM][LDS_S];
    __shared__ float sB[BN][LDS_S];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    const int split = blockIdx.z;
    const int pbeg = split * p.pix_per_split;
    int pend = pbeg + p.pix_per_split; pend = pend < p.Npix ? pend : p.Npix;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[AROWS], rb[BROWSW];
    const GatherDims g{p.Ti, p.Hi, p.Wi, p.in_st, p.in_sh};

    for (int pk = pbeg; pk < pend; pk += BK) {
        const int pix = pk + lane;
        const bool live = pix < pend;
        int n = 0, ot = 0, oy = 0, ox = 0;
        {
            const int q = live ? pix : pend - 1;
            ox = q % p.Wo; int r = q / p.Wo;
            oy = r % p.Ho; r = r / p.Ho;
            ot = r % p.To; n = r / p.To;
        }
        const int sp = (ot * p.Ho + oy) * p.Wo + ox;
        const float* __restrict__ dyn = p.dY + (long)n * p.dy_sn + sp;
        const float* __restrict__ Xn = p.X + (long)n * p.in_sn;
        const int ots = ot * p.st, oys = oy * p.sh, oxs = ox * p.sw;
#pragma unroll
        for (int s = 0; s < AROWS; ++s) {
            int row = m0 + wave * AROWS + s;
            float v = 0.f;
            if (live && row < p.M) v = dyn[(long)row * p.dy_sc];
            ra[s] = v;
        }
#pragma unroll
        for (int s = 0; s < BROWSW; ++s) {
            const int j = j0 + wave * BROWSW + s;   // < Jpad by construction of the grid
            const int4 e = p.jtab[j];
            float v = gather_one<REFLECT, IS3D>(Xn, e, ots, oys, oxs, g);
            rb[s] = live ? v : 0.f;
        }
        __syncthreads();   // previous K-step's fragment reads are done
#pragma unroll
        for (int s = 0; s < AROWS; ++s) sA[wave * AROWS + s][lane] = ra[s];
#pragma unroll
        for (int s = 0; s < BROWSW; ++s) sB[wave * BROWSW + s][lane] = rb[s];
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int kcol = kk * 2 + (lane >> 5);
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = sA[wm * TM + i * 32 + (lane & 31)][kcol];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = sB[wn * TN + j * 32 + (lane & 31)][kcol];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float* __restrict__ out = p.slab + (long)split * p.M * p.J;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int col = j0 + wn * TN + j * 32 + (lane & 31);
        if (col >= p.J) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) out[(long)row * p.J + col] = acc[i][j][r];
            }
    }
}

// dW[m][j] (+ db[m] from the trailing ones column) = sum over splits, fixed order
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, float* __restrict__ db,
                                    int M, int J, int Jw, int S) {
    const long total = (long)M * J;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int s = 0; s < S; ++s) acc += slab[(long)s * total + i];
        const int m = (int)(i / J), j = (int)(i % J);
        if (j < Jw) dW[(long)m * Jw + j] = acc;
        else if (db) db[m] = acc;
    }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_wgrad(const WgradP& p, int S, int reflect, int is3d, hipStream_t s) {
    dim3 grid(c2m_cdiv(p.J, BN), c2m_cdiv(p.M, BM), S);
    if (reflect) {
        if (is3d) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, true, true>), grid, dim3(256), 0, s, p);
        else      hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, true, false>), grid, dim3(256), 0, s, p);
    } else {
        if (is3d) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, false, true>), grid, dim3(256), 0, s, p);
        else      hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WGM, WGN, false, false>), grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

// Number of pixel splits the wgrad launch will use for (M, J, Npix): the caller sizes `slab` = S*M*J floats.
C2M_API int c2m_conv_wgrad_splits(int M, int J, int Npix) {
    const int BM = M <= 32 ? 32 : 64, BN = M <= 32 ? 128 : 64;
    const long tiles = (long)c2m_cdiv(M, BM) * c2m_cdiv(J, BN);
    long S = (1024 + tiles - 1) / tiles;
    const long maxS = (Npix + 2047) / 2048;   // at least 2048 pixels per split
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return (int)S;
}

C2M_API int c2m_conv_wgrad(const float* dY, const float* X, float* slab, float* dW, float* db, const int* jtab,
                           const int64_t* g, void* stream) {
    C2M_ENTER();
    WgradP p;
    p.dY = dY; p.X = X; p.slab = slab; p.jtab = reinterpret_cast<const int4*>(jtab);
    p.M = (int)g[0]; p.J = (int)g[1]; p.Jpad = (int)g[2];
    p.Npix = (int)g[3]; p.To = (int)g[4]; p.Ho = (int)g[5]; p.Wo = (int)g[6];
    p.Ti = (int)g[7]; p.Hi = (int)g[8]; p.Wi = (int)g[9];
    p.st = (int)g[10]; p.sh = (int)g[11]; p.sw = (int)g[12];
    p.in_sn = g[13]; p.in_st = g[14]; p.in_sh = g[15];
    p.dy_sn = g[16]; p.dy_sc = g[17];
    const int reflect = (int)g[22], is3d = (int)g[23];
    const int Jw = (int)g[24];   // columns that belong to dW (J - 1 when a ones row is present)
    if (p.M <= 0 || p.J <= 0 || p.Npix <= 0) return 0;
    const int BN = p.M <= 32 ? 128 : 64;
    if (p.Jpad < c2m_cdiv(p.J, BN) * BN) return (int)hipErrorInvalidValue;
    const int S = c2m_conv_wgrad_splits(p.M, p.J, p.Npix);
    int per = c2m_cdiv(p.Npix, S);
    per = ((per + 63) / 64) * 64;
    p.pix_per_split = per;
    const int Seff = c2m_cdiv(p.Npix, per);   // <= S; unused slabs are never read
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (p.M <= 32) rc = launch_wgrad<32, 128, 1, 4>(p, Seff, reflect, is3d, s);
    else           rc = launch_wgrad<64, 64, 2, 2>(p, Seff, reflect, is3d, s);
    if (rc) return rc;
    const long total = (long)p.M * p.J;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, s, slab, dW, db, p.M, p.J, Jw,
                       Seff);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ reflect fold
// dX[n,c,t,y,x] = sum of dXpad over every padded coordinate that reflects onto (t,y,x); fixed order.
struct FoldP { int T, H, W, pt, ph, pw; long total; };

__device__ __forceinline__ int fold_sources(int i, int I, int p, int* src) {
    int n = 0;
    src[n++] = i + p;
    if (p > 0) {
        if (i >= 1 && i <= p) src[n++] = p - i;
        if (i >= I - 1 - p && i <= I - 2) src[n++] = 2 * (I - 1) - i + p;
    }
    return n;
}

__global__ void reflect_fold_kernel(const float* __restrict__ dXp, float* __restrict__ dX, const FoldP f) {
    const int Tp = f.T + 2 * f.pt, Hp = f.H + 2 * f.ph, Wp = f.W + 2 * f.pw;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < f.total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % f.W); long r = idx / f.W;
        const int y = (int)(r % f.H); r /= f.H;
        const int t = (int)(r % f.T); const long nc = r / f.T;
        int st[3], sy[3], sx[3];
        const int nt = fold_sources(t, f.T, f.pt, st), ny = fold_sources(y, f.H, f.ph, sy),
                  nx = fold_sources(x, f.W, f.pw, sx);
        const float* __restrict__ base = dXp + nc * (long)Tp * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c) acc += base[((long)st[a] * Hp + sy[b]) * Wp + sx[c]];
        dX[idx] = acc;
    }
}

C2M_API int c2m_reflect_fold(const float* dXpad, float* dX, long NC, int T, int H, int W, int pt, int ph, int pw,
                             void* stream) {
    C2M_ENTER();
    FoldP f{T, H, W, pt, ph, pw, NC * (long)T * H * W};
    if (f.total <= 0) return 0;
    hipLaunchKernelGGL(reflect_fold_kernel, dim3(c2m_grid(f.total, 256)), dim3(256), 0, (hipStream_t)stream, dXpad, dX,
                       f);
    return (int)hipGetLastError();
}
