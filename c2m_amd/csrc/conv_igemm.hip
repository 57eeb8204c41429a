// Implicit-GEMM convolution for gfx950 (CDNA4) on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32, exact f32).
//
// Replaces the ATen conv2d/conv3d (+ explicit ReflectionPad2d/3d) calls of the reference blocks:
//   src/modules/layers/down_block.py:14-23,35-47  same_block.py:14-23,36-46,55-67  up_block.py:9-13
//   residual_block.py:13-31,42-71  spade_block.py:47-49  vgg.py (torchvision features)  generator.py:76-78
//
// One kernel serves forward and data-gradient:  D[m][pix] = sum_k A[m][k] * G(k, pix)
//   * K is ordered (channel chunk, tap group, tap, channel-in-chunk): one K-step (16 deep) = NS taps x CK channels
//     (NS*CK = 16).  The spatial part of the gather (o*stride + tap offset, zero / reflect boundary) is therefore
//     computed once per tap per K-step and the CK channel loads only add a channel stride: ~2 VALU per element.
//   * A  : row-major [M][K] weights packed by the host into that K order (zero rows for channel / tap padding).
//   * D  : written with arbitrary output strides (NCHW / NCTHW, or the strided parity classes of a dgrad).
//   * split-K over gridDim.z for layers with few output pixels (deep encoder layers: 2x4 maps, K up to 16384):
//     partial slabs + a fixed-order reduction that also applies bias / activation (no float atomics).
// Orientation: the MFMA column index (lane&31) is the pixel, so every accumulator register is stored as 2 x 128
// contiguous bytes per wave -> coalesced NCHW stores.
//
// wgrad:  dW[co][j] = sum_pix dY[co][pix] * G(j, pix), rows j in (tap, channel) order so that 16 consecutive rows
// share a tap; split-K over pixels into deterministic slabs, an all-ones row gives the bias gradient; the slab
// reduction permutes back to the [Cout][Cin][taps] weight layout.
//
// Tiling: 256 threads = 4 waves, BK = 16, register-prefetched + double-buffered LDS, one barrier per K-step.
#include "common.h"
#include "dtype.h"
#include <stdlib.h>
#include <string.h>
#include "conv_store.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvP {
    const float* A;
    const float* X;
    float* Y;            // output, or the slab base when splits > 1
    const float* bias;   // per-row (may be null); ignored when splits > 1
    const int4* ktab;    // per K-step: header {chan_off, nvalid_chan, 0, 0} + NS x {dt, dy, dx, valid}
    int M, nk, lda;      // nk K-steps of 16
    int Npix, To, Ho, Wo;
    int Ti, Hi, Wi;
    int st, sh, sw;
    int in_sc;           // channel stride of the gathered tensor (elements)
    long in_sn, in_st, in_sh;
    long out_sn, out_sc, out_st, out_sh, out_sw, out_off;
    long slab_stride;    // elements between split slabs
    unsigned x_bytes;    // size of X in bytes (buffer-load bounds: out-of-range lanes read 0)
    // optional second target (reflect-pad dgrad): outputs whose padded coordinate (o*ps + po) lies inside
    // [lo, lo+ext) go straight to the unpadded gradient Y2, only the pad ring is written to Y
    float* Y2;
    int ps_t, ps_y, ps_x, po_t, po_y, po_x, lo_t, lo_y, lo_x, ext_t, ext_y, ext_x;
    long y2_sn, y2_sc, y2_st, y2_sh;
    // LDS-patch variant (3x3, stride 1): input origin of a tile = o + (iy0, ix0); per-tap offsets inside the patch
    int iy0, ix0, pty[3], ptx[3], nchunks, cin;   // cin: input channels of the gathered tensor (LDS-patch kernels)
    int ksteps_per_split;
    int reflect, is3d;
    int act;
    float slope;
    // class batching (stride-s data gradient: the s^d parity classes share every dimension and differ only in
    // weights, tap table and output origin): blockIdx.z = cls * splits + split
    int ncls, splits, ktab_cls;       // ktab_cls: int4 entries per class table
    // bf16 data path (dtype.h): the bf16-operand kernels (BF) gather X as bf16 ALWAYS -- the host casts fp32 inputs once -- and
    // write Y / Y2 as bf16 when yh is set and the launch stores results directly (split-K slabs stay fp32).  The fp32 kernels
    // and the vector-ALU thin kernels read fp32 X, or bf16 X when xh is set (thin kernels only).
    int xh, yh;
    long a_cls;                       // floats per class weight matrix
    long out_off_c[8];
    int po_c[8][3];
    // NC8 gather kernel (conv_gather_nc8_kernel): X = [N][CB][Ti*Hi*Wi][8] bf16, A = c2m_pack_weights_bf16_gather image
    // [class][tap][16-channel chunk][half][Mpad rows] 16-byte units, ktab = [class][ntaps] {dt, dy, dx, 1}
    int g8_nch, g8_ntaps, g8_CB, g8_Mpad;
    unsigned g8_a_bytes, g8_plane_bytes;      // bytes of ONE class image; bytes of one channel-block plane
    int g8_dbg;                               // tuning only (geom[95] >> 8): 1 = no K loop, 2 = no stores
};

// spatial offset of one tap for this thread's pixel, or -1 when it falls in zero padding
__device__ __forceinline__ int spatial_off(const int4 tp, int ots, int oys, int oxs, int Ti, int Hi, int Wi, int in_st,
                                           int in_sh, int reflect, int is3d) {
    int it = is3d ? ots + tp.x : 0, iy = oys + tp.y, ix = oxs + tp.z;
    bool ok = tp.w != 0;
    if (reflect) {
        if (is3d) { it = it < 0 ? -it : it; it = it >= Ti ? 2 * Ti - 2 - it : it; }
        iy = iy < 0 ? -iy : iy; iy = iy >= Hi ? 2 * Hi - 2 - iy : iy;
        ix = ix < 0 ? -ix : ix; ix = ix >= Wi ? 2 * Wi - 2 - ix : ix;
    } else {
        ok = ok && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        if (is3d) ok = ok && (unsigned)it < (unsigned)Ti;
    }
    return ok ? it * in_st + iy * in_sh + ix : -1;
}

#define C2M_OOB 0x80000000u   // voffset beyond any tensor we accept (< 2 GiB): the buffer load returns 0

template <class P>
__device__ __forceinline__ void decompose_pix(int pix, const P& p, int& n, int& ot, int& oy, int& ox) {
    ox = pix % p.Wo; int r = pix / p.Wo;
    oy = r % p.Ho;   r = r / p.Ho;
    ot = r % p.To;   n = r / p.To;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// BF = true: bf16 operands (X is a bf16 tensor in HBM -- the bf16 data path --, weights are rounded to bf16, RNE, while they
// are staged into LDS; fp32 accumulation; fp32 or bf16 output, ConvP::yh) on v_mfma_f32_32x32x16_bf16 -- one instruction per 16-deep K-step and 32x32 tile.  LDS
// image [k half h][row][8 bf16] (16 B per lane, conflict-free ds_read_b128).  The K slot order inside a step is
// permuted identically for A and B so that each gathering thread's values are contiguous: slot(k) = (k % BROWS) *
// BPASS + k / BROWS (a sum over k does not care about the order).
template <int BM, int BN, int WGM, int WGN, int NS, int U, bool BF = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    // U = table K-steps (16 deep each) staged per barrier: U = 2 halves the barriers per MFMA at twice the LDS
    constexpr int BK = 16, CK = BK / NS, BKU = BK * U;
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int LDA_S = BM + 4, LDB_S = BN;
    constexpr int BROWS = 256 / BN;           // k rows gathered per pass
    constexpr int BPASS = BK / BROWS;         // gathers per thread per K-step
    constexpr int A_F4 = BM * BK / 4;         // float4 loads per K-step (whole block)
    constexpr int APASS = (A_F4 + 255) / 256;
    static_assert(WGM * WGN == 4 && BN >= 64 && CK % BROWS == 0, "tile");
    static_assert(!BF || BROWS <= 2, "bf16 slot packing is written for BN = 128 / 256");
    __shared__ float sA[BF ? 1 : 2][BF ? 1 : BKU][BF ? 1 : LDA_S];
    __shared__ float sB[BF ? 1 : 2][BF ? 1 : BKU][BF ? 1 : LDB_S];
    __shared__ uint4 hA[BF ? 2 * U * 2 * BM : 1];      // [buf][u][h][row] x 8 bf16
    __shared__ uint4 hB[BF ? 2 * U * 2 * BN : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // XCD-aware order (common.h): the row tiles of one pixel tile gather the same pixels, neighbouring pixel tiles share taps
    const C2mBlock blk = c2m_xcd_block((unsigned)(p.Npix + BN - 1) / BN, (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM, n0 = blk.x * BN;
    const int cls = p.ncls > 1 ? (int)blk.z / p.splits : 0;
    const int split = p.ncls > 1 ? (int)blk.z - cls * p.splits : (int)blk.z;
    const float* __restrict__ Acls = p.A + cls * p.a_cls;
    const int4* __restrict__ ktab = p.ktab + cls * p.ktab_cls;
    const int kt_beg = split * p.ksteps_per_split;
    int kt_end = kt_beg + p.ksteps_per_split; kt_end = kt_end < p.nk ? kt_end : p.nk;

    // ---- gather side: this thread owns one pixel column of the tile
    const int bp = tid % BN;
    const int brow0 = __builtin_amdgcn_readfirstlane(tid / BN);
    int pn, pt, py, px;
    {
        int pix = n0 + bp; pix = pix < p.Npix ? pix : p.Npix - 1;
        decompose_pix(pix, p, pn, pt, py, px);
    }
    const int ots = pt * p.st, oys = py * p.sh, oxs = px * p.sw;
    const int in_st = (int)p.in_st, in_sh = (int)p.in_sh;

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

    float4 ra[U][APASS];
    float rb[BF ? 1 : U][BF ? 1 : BPASS];
    unsigned short rh[BF ? U : 1][BF ? BPASS : 1];     // BF: X is bf16 in HBM, gathered as raw 16-bit values
    constexpr int XES = BF ? 2 : 4;                     // bytes per gathered element

    // per-thread weight row pointers (fixed for the whole K loop)
    const float* __restrict__ aptr[APASS];
#pragma unroll
    for (int s = 0; s < APASS; ++s) {
        int row = m0 + arow + s * 64; row = row < p.M ? row : p.M - 1;
        aptr[s] = Acls + (long)row * p.lda + akq;
    }
    // gather table of the NEXT K-step to load, fetched one step ahead (scalar loads: their latency must not sit in
    // front of the address arithmetic)
    int4 t_hdr, t_tap[NS];
    auto fetch_table = [&](int kt) {
        const int4* __restrict__ kd = ktab + (long)kt * (1 + NS);
        t_hdr = kd[0];
#pragma unroll
        for (int q = 0; q < NS; ++q) t_tap[q] = kd[1 + q];
    };

    // B gather through a raw buffer descriptor: per element the VECTOR offset is the per-tap spatial byte offset of
    // this lane (computed once per tap) and the SCALAR offset is the channel offset (SALU only); padding taps and
    // lanes outside the image carry an out-of-range voffset and read 0 -- no per-element VALU at all.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const unsigned img_byte = (unsigned)(pn * (int)p.in_sn) * (unsigned)XES;
    // one 16-deep table step into register set u; kt >= kt_end (odd tail of a U = 2 pair) loads zeros
    auto load_step = [&](int kt, float4 (&fa)[APASS], float (&fb)[BF ? 1 : BPASS], unsigned short (&fh)[BF ? BPASS : 1]) {
        const bool real = kt < kt_end;
#pragma unroll
        for (int s = 0; s < APASS; ++s) {
            if (A_F4 >= 256 || arow + s * 64 < BM)
                fa[s] = real ? *reinterpret_cast<const float4*>(aptr[s] + kt * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int4 hdr = t_hdr;
        unsigned vo[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int so = spatial_off(t_tap[q], ots, oys, oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
            vo[q] = (real && so >= 0) ? img_byte + (unsigned)so * (unsigned)XES : C2M_OOB;
        }
#pragma unroll
        for (int s = 0; s < BPASS; ++s) {
            const int slot = (s * BROWS) / CK;                       // compile-time
            const int cc = (s * BROWS) % CK + brow0;                  // wave-uniform
            // channels beyond nvalid only meet zero weights (A is zero-padded); they read in-range data or 0
            const int soff = (hdr.x + cc * p.in_sc) * XES;
            if constexpr (BF) fh[s] = __builtin_amdgcn_raw_buffer_load_b16(xrsrc, vo[slot], soff, 0);
            else fb[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, vo[slot], soff, 0));
        }
        fetch_table(kt + 1 < p.nk ? kt + 1 : p.nk - 1);
    };
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int u = 0; u < U; ++u) load_step(kt + u, ra[u], rb[BF ? 0 : u], rh[BF ? u : 0]);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (BF) {
                __bf16* __restrict__ a8 = reinterpret_cast<__bf16*>(hA) + (long)((buf * U + u) * 2) * BM * 8;
#pragma unroll
                for (int s = 0; s < APASS; ++s) {
                    const int r = arow + s * 64;
                    if (A_F4 >= 256 || r < BM) {
                        const float4 v = ra[u][s];
                        if constexpr (BROWS == 1) {        // slot = k: 4 consecutive slots in half akq / 8
                            bf16x4 q = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                            *reinterpret_cast<bf16x4*>(a8 + ((akq >> 3) * BM + r) * 8 + (akq & 7)) = q;
                        } else {                            // slot = (k & 1) * 8 + k / 2: even k -> half 0, odd k -> half 1
                            bf16x2 e = {(__bf16)v.x, (__bf16)v.z}, o = {(__bf16)v.y, (__bf16)v.w};
                            *reinterpret_cast<bf16x2*>(a8 + (0 * BM + r) * 8 + (akq >> 1)) = e;
                            *reinterpret_cast<bf16x2*>(a8 + (1 * BM + r) * 8 + (akq >> 1)) = o;
                        }
                    }
                }
                // this thread's BPASS values are slots brow0*BPASS .. +BPASS-1: whole 8-slot halves
                uint4* __restrict__ b8 = hB + (long)((buf * U + u) * 2) * BN;
#pragma unroll
                for (int hh = 0; hh < BPASS / 8; ++hh) {
                    bf16x8 q;
#pragma unroll
                    for (int e = 0; e < 8; ++e) q[e] = __builtin_bit_cast(__bf16, rh[u][hh * 8 + e]);
                    const int h = BROWS == 1 ? hh : brow0;
                    b8[h * BN + bp] = __builtin_bit_cast(uint4, q);
                }
            } else {
#pragma unroll
                for (int s = 0; s < APASS; ++s) {
                    int r = arow + s * 64;
                    if (A_F4 >= 256 || r < BM) {
                        sA[buf][u * BK + akq + 0][r] = ra[u][s].x; sA[buf][u * BK + akq + 1][r] = ra[u][s].y;
                        sA[buf][u * BK + akq + 2][r] = ra[u][s].z; sA[buf][u * BK + akq + 3][r] = ra[u][s].w;
                    }
                }
#pragma unroll
                for (int s = 0; s < BPASS; ++s) sB[buf][u * BK + brow0 + s * BROWS][bp] = rb[u][s];
            }
        }
    };

    if (kt_beg < kt_end) {
        fetch_table(kt_beg);
        load_tile(kt_beg);
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    for (int kt = kt_beg; kt < kt_end; kt += U) {
        const bool more = kt + U < kt_end;
        // interleaved gather issue (below): +9...14 % on the 64-row tile, a few % on the 32-row tile, nothing on the
        // 128-row tile (its 32-MFMA steps already cover the gathers with 3 blocks/CU, and it would pay 8 more VGPRs)
        constexpr bool IL = !BF && U == 1 && BM < 128;
        if (!IL && more) load_tile(kt + U);
        // all fragment reads of a 16-deep step are issued first (own registers each), so the LDS latency of k-pair
        // kk+1.. hides behind the MFMAs of kk (the compiler otherwise recycles 4 VGPRs and serialises read -> mfma)
        if constexpr (BF) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint4* __restrict__ a8 = hA + (long)((cur * U + u) * 2 + (lane >> 5)) * BM;
                const uint4* __restrict__ b8 = hB + (long)((cur * U + u) * 2 + (lane >> 5)) * BN;
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = __builtin_bit_cast(bf16x8, a8[wm * TM + i * 32 + (lane & 31)]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = __builtin_bit_cast(bf16x8, b8[wn * TN + j * 32 + (lane & 31)]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (IL) {
            // One scheduling region per K-step: the gathers (and their address arithmetic) of the NEXT step are issued in
            // the shadow of this step's MFMAs instead of in front of them, and the fragment reads run two k-pairs ahead
            // of the MFMAs that use them.  The last step re-gathers its own (in-range) tile and drops it; the table of
            // the step after next is fetched behind the region (scalar loads share the LDS counter).
            float a[BK / 2][MI], b[BK / 2][NI];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int krow = kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kk][i] = sA[cur][krow][wm * TM + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[kk][j] = sB[cur][krow][wn * TN + j * 32 + (lane & 31)];
            }
            {
                const int ktl = more ? kt + 1 : kt;
#pragma unroll
                for (int s2 = 0; s2 < APASS; ++s2)
                    if (A_F4 >= 256 || arow + s2 * 64 < BM) ra[0][s2] = *reinterpret_cast<const float4*>(aptr[s2] + ktl * BK);
                const int4 hdr = t_hdr;
                unsigned vo[NS];
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    const int so = spatial_off(t_tap[q], ots, oys, oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
                    vo[q] = so >= 0 ? img_byte + (unsigned)so * 4u : C2M_OOB;
                }
#pragma unroll
                for (int s2 = 0; s2 < BPASS; ++s2) {
                    const int slot = (s2 * BROWS) / CK;
                    const int cc = (s2 * BROWS) % CK + brow0;
                    rb[0][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        xrsrc, vo[slot], (hdr.x + cc * p.in_sc) * 4, 0));
                }
            }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            constexpr int NMF = (BK / 2) * MI * NI, RPM = (MI + NI + MI * NI - 1) / (MI * NI);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);        // two k-pairs of fragment reads
#pragma unroll
            for (int g = 0; g < NMF; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch_table(kt + 2 < p.nk ? kt + 2 : p.nk - 1);
        } else
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a[BK / 2][MI], b[BK / 2][NI];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int krow = u * BK + kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kk][i] = sA[cur][krow][wm * TM + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[kk][j] = sB[cur][krow][wn * TN + j * 32 + (lane & 31)];
            }
#ifdef C2M_IGEMM_SERIAL_FRAGS
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            }
#ifndef C2M_IGEMM_SERIAL_FRAGS
            // fragment reads two k-pairs ahead of the MFMAs that use them: only the first reads of a K-step are waited for
            {
                constexpr int NMF = (BK / 2) * MI * NI, RPM = (MI + NI + MI * NI - 1) / (MI * NI);
                __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
                for (int g = 0; g < NMF; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: acc[i][j][r] -> row m = ..(r&3)+8*(r>>2)+4*(lane>>5), col pix = ..(lane&31)
    const bool direct = p.splits == 1;
    const bool yh = BF && p.yh && direct;                 // bf16 output elements (slabs are fp32)
    const int yes = yh ? 2 : 4;
    float* __restrict__ Yb = yh ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(p.Y) + p.out_off_c[cls])
                                : p.Y + (long)split * p.slab_stride + p.out_off_c[cls];
    const int po_t = p.po_c[cls][0], po_y = p.po_c[cls][1], po_x = p.po_c[cls][2];
    {
        const int row0 = __builtin_amdgcn_readfirstlane(m0 + wm * TM);
        if (row0 + MI * 32 <= p.M) {
            // two targets (reflect data gradient): a pixel goes either to the interior tensor Y2 or, on the pad ring, to Y;
            // each target is one pass of stores with the other target's pixels out of range, and the ring pass is
            // skipped by the waves that hold no ring pixel (most of them)
            unsigned voff[NI], voff2[NI];
            bool ring = false;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int pix = n0 + wn * TN + j * 32 + (lane & 31);
                int n, ot, oy, ox;
                decompose_pix(pix < p.Npix ? pix : 0, p, n, ot, oy, ox);
                const long e = (long)n * p.out_sn + (long)ot * p.out_st + (long)oy * p.out_sh + (long)ox * p.out_sw +
                               4L * (lane >> 5) * p.out_sc;
                voff[j] = pix < p.Npix ? (unsigned)(e * yes) : 0x80000000u;
                voff2[j] = 0x80000000u;
                if (p.Y2) {
                    const int tp = ot * p.ps_t + po_t - p.lo_t, yp = oy * p.ps_y + po_y - p.lo_y,
                              xp = ox * p.ps_x + po_x - p.lo_x;
                    if (pix < p.Npix && (unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y &&
                        (unsigned)xp < (unsigned)p.ext_x) {
                        const long e2 = (long)n * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp +
                                        4L * (lane >> 5) * p.y2_sc;
                        voff2[j] = (unsigned)(e2 * yes);
                        voff[j] = 0x80000000u;
                    }
                }
                ring = ring || voff[j] != 0x80000000u;
            }
            if (p.Y2) c2m_store_tile_fast<MI, NI>(acc, p.Y2, voff2, row0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh);
            if (!p.Y2 || __any(ring))
                c2m_store_tile_fast<MI, NI>(acc, Yb, voff, row0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int pix = n0 + wn * TN + j * 32 + (lane & 31);
        if (pix >= p.Npix) continue;
        int n, ot, oy, ox;
        decompose_pix(pix, p, n, ot, oy, ox);
        float* ybase = Yb;                                     // element type: float, or bf16_t when yh
        long yidx = (long)n * p.out_sn + (long)ot * p.out_st + (long)oy * p.out_sh + (long)ox * p.out_sw;
        long row_stride = p.out_sc;
        if (p.Y2) {
            const int tp = ot * p.ps_t + po_t - p.lo_t, yp = oy * p.ps_y + po_y - p.lo_y,
                      xp = ox * p.ps_x + po_x - p.lo_x;
            if ((unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y && (unsigned)xp < (unsigned)p.ext_x) {
                ybase = p.Y2;
                yidx = (long)n * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp;
                row_stride = p.y2_sc;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) {
                    float v = acc[i][j][r];
                    if (direct) {
                        if (p.bias) v += p.bias[row];
                        v = c2m_act(v, p.act, p.slope);
                    }
                    if (yh) reinterpret_cast<bf16_t*>(ybase)[yidx + (long)row * row_stride] = (bf16_t)v;
                    else ybase[yidx + (long)row * row_stride] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ NC8 gather (round 4)
// The general bf16 forward / data-gradient kernel over CHANNEL-BLOCKED activations (conv_nc8.hip has the layout and the 3x3 /
// 4x4-stride-2 / 3x3x3 patch forms): any tap set, stride, 2-D or 3-D, stride-parity classes, two-target reflect epilogue, split-K --
// the geometry of conv_igemm_kernel -- for the layers the patch forms do not take ((3,4,4) / (4,4,4) stride-2 blocks, 7x7, 1x1,
// small or ragged maps, the reflect data gradients of the stride-2 layers whose (W/2 + 1)-wide class planes waste half of a
// 32-column tile).  A K-step is (tap, 16 channels); per K-step a wave fetches the two 16-byte units (channel blocks 2c, 2c + 1)
// of ITS 64 pixels with two LDS-DMA instructions whose per-lane address is the pixel's tap offset, computed once per TAP (the
// K loop runs tap-major: chunks inside) -- the NCHW gather kernel issued 16 two-byte loads, 8 packs and 2 ds_write_b128 per lane
// for the same K-step and spent 105 VALU per 8 MFMAs on them.  Weights come pre-packed in bf16
// ([tap][chunk][half][row]) by LDS-DMA as well.  Tile BM rows x 256 pixels, wave w owns pixels 64 w .. 64 w + 63 for all rows;
// U K-steps per stage, NBUF stages, one barrier per stage with counted vmcnt (every wave issues NDMA instructions per stage;
// steps past the split's end and channel blocks past the tensor's last go through zero-record descriptors).
template <int BM, int U, int NBUF, int WGS>
__global__ __launch_bounds__(256, WGS) void conv_gather_nc8_kernel(const ConvP p) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int MI = BM / 32, NI = 2;
    constexpr int A_ST = U * 2 * BM;                              // weight units per stage: [u][half][row]
    constexpr int NAI = (A_ST + 255) / 256;
    constexpr int A_PAD = NAI * 256;
    constexpr int BUF = A_PAD + U * 512;                          // + [u][half][256 pixels]
    constexpr int NDMA = NAI + 2 * U;
    static_assert(64 % (2 * BM) == 0 || (2 * BM) % 64 == 0, "a 64-unit DMA row belongs to one K-step");
    static_assert(NBUF * BUF * 16 * WGS <= 158 * 1024, "LDS");
    __shared__ uint4 smem[NBUF * BUF];
    __shared__ int4 s_taps[64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const C2mBlock blk = c2m_xcd_block((unsigned)(p.Npix + 255) / 256, (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM, n0 = blk.x * 256;
    const int cls = p.ncls > 1 ? (int)blk.z / p.splits : 0;
    const int split = p.ncls > 1 ? (int)blk.z - cls * p.splits : (int)blk.z;
    const int k_beg = split * p.ksteps_per_split;
    int k_end = k_beg + p.ksteps_per_split; k_end = k_end < p.nk ? k_end : p.nk;
    if (p.g8_dbg & 1) k_end = k_beg;
    const int nch = p.g8_nch, ntaps = p.g8_ntaps;

    if (tid < ntaps) s_taps[tid] = p.ktab[cls * ntaps + tid];

    // ---- this lane's pixel (DMA side and MFMA column side agree: wave w holds pixels 64 w ..)
    const int pix = n0 + wave * 64 + lane;
    const bool pvalid = pix < p.Npix;
    int pn, pt, py, px;
    decompose_pix(pvalid ? pix : p.Npix - 1, p, pn, pt, py, px);
    const int ots = pt * p.st, oys = py * p.sh, oxs = px * p.sw;
    const int in_st = (int)p.in_st, in_sh = (int)p.in_sh;
    const unsigned img_byte = (unsigned)pn * (unsigned)p.g8_CB * p.g8_plane_bytes;
    auto tap_voff = [&](int tap) -> unsigned {
        const int4 tp = s_taps[tap < ntaps ? tap : ntaps - 1];
        int it = p.is3d ? ots + tp.x : 0, iy = oys + tp.y, ix = oxs + tp.z;
        bool ok = pvalid;
        if (p.reflect) {
            if (p.is3d) { it = it < 0 ? -it : it; it = it >= p.Ti ? 2 * p.Ti - 2 - it : it; }
            iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
        } else {
            ok = ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && (unsigned)it < (unsigned)p.Ti;
        }
        return ok ? img_byte + (unsigned)(it * in_st + iy * in_sh + ix) * 16u : C2M_OOB;
    };

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) uint4*)&smem[0];
    const unsigned long aaddr = (unsigned long)p.A + (unsigned long)cls * p.g8_a_bytes;
    const u32x4 ars = {(unsigned)aaddr, (unsigned)(aaddr >> 32) & 0xffffu, p.g8_a_bytes, 0x00020000u};
    const unsigned long xaddr = (unsigned long)p.X;
    const u32x4 xrs = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const unsigned a_step_bytes = (unsigned)(2 * p.g8_Mpad * 16);
    unsigned avo[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int d = (i * 4 + wave) * 64 + lane, within = d % (2 * BM);
        avo[i] = d < A_ST ? (unsigned)(((within / BM) * p.g8_Mpad + m0 + within % BM) * 16) : C2M_OOB;
    }

    // issue-side running state: the K-steps are issued in ascending order, (tap, chunk) advance with them
    int is_tap = __builtin_amdgcn_readfirstlane(k_beg / nch);
    int is_chunk = __builtin_amdgcn_readfirstlane(k_beg - is_tap * nch);
    __syncthreads();                                       // tap table visible
    unsigned is_voff = tap_voff(is_tap);
    auto issue = [&](int k0, int buf) {
        const unsigned base = lds0 + (unsigned)(buf * BUF * 16);
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const int d0 = (i * 4 + wave) * 64;
            const int ka = k0 + d0 / (2 * BM);
            const bool live = d0 < A_ST && ka < k_end;
            u32x4 r = ars;
            r[2] = live ? p.g8_a_bytes : 0u;
            const int soff = live ? (int)((unsigned)ka * a_step_bytes) : 0;
            const unsigned dst = base + (unsigned)(d0 * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(avo[i]), "s"(r), "s"(soff) : "memory");
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = k0 + u < k_end;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cb = is_chunk * 2 + h;
                u32x4 r = xrs;
                const bool on = live && cb < p.g8_CB;
                r[2] = on ? p.x_bytes : 0u;
                const int soff = on ? (int)((unsigned)cb * p.g8_plane_bytes) : 0;
                const unsigned dst = base + (unsigned)((A_PAD + (u * 2 + h) * 256 + wave * 64) * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(is_voff), "s"(r), "s"(soff) : "memory");
            }
            ++is_chunk;
            if (is_chunk == nch) { is_chunk = 0; ++is_tap; is_voff = tap_voff(is_tap); }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) issue(k_beg + s * U, s);
    int cur = 0;
    for (int k = k_beg; k < k_end; k += U) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        issue(k + (NBUF - 1) * U, cur == 0 ? NBUF - 1 : cur - 1);
        const uint4* __restrict__ sb = smem + cur * BUF;
        bf16x8 a[U][MI], b[U][NI];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < MI; ++i) a[u][i] = __builtin_bit_cast(bf16x8, sb[(u * 2 + (lane >> 5)) * BM + i * 32 + (lane & 31)]);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[u][j] = __builtin_bit_cast(bf16x8, sb[A_PAD + (u * 2 + (lane >> 5)) * 256 + wave * 64 + j * 32 + (lane & 31)]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-record DMAs of the tail still write LDS

    // ---- epilogue: the mappings of conv_igemm_kernel with one row block of BM and the wave's 64 pixel columns; buffer stores with
    // per-pixel offsets for both targets (conv_store.h), rows past M masked per lane in the last row tile
    if ((p.g8_dbg & 2) && acc[0][0][0] != 12345.f) return;
    const bool direct = p.splits == 1;
    const bool yh = p.yh && direct;
    const int yes = yh ? 2 : 4;
    float* __restrict__ Yb = yh ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(p.Y) + p.out_off_c[cls])
                                : p.Y + (long)split * p.slab_stride + p.out_off_c[cls];
    const int po_t = p.po_c[cls][0], po_y = p.po_c[cls][1], po_x = p.po_c[cls][2];
    unsigned voff[NI], voff2[NI];
    bool ring = false;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int q = n0 + wave * 64 + j * 32 + (lane & 31);
        int n, ot, oy, ox;
        decompose_pix(q < p.Npix ? q : 0, p, n, ot, oy, ox);
        const long e = (long)n * p.out_sn + (long)ot * p.out_st + (long)oy * p.out_sh + (long)ox * p.out_sw +
                       4L * (lane >> 5) * p.out_sc;
        voff[j] = q < p.Npix ? (unsigned)(e * yes) : 0x80000000u;
        voff2[j] = 0x80000000u;
        if (p.Y2) {
            const int tp = ot * p.ps_t + po_t - p.lo_t, yp = oy * p.ps_y + po_y - p.lo_y, xp = ox * p.ps_x + po_x - p.lo_x;
            if (q < p.Npix && (unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y &&
                (unsigned)xp < (unsigned)p.ext_x) {
                const long e2 = (long)n * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp + 4L * (lane >> 5) * p.y2_sc;
                voff2[j] = (unsigned)(e2 * yes);
                voff[j] = 0x80000000u;
            }
        }
        ring = ring || voff[j] != 0x80000000u;
    }
    if (m0 + BM <= p.M) {
        if (p.Y2) c2m_store_tile_fast<MI, NI>(acc, p.Y2, voff2, m0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh);
        if (!p.Y2 || __any(ring))
            c2m_store_tile_fast<MI, NI>(acc, Yb, voff, m0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh);
    } else {
        if (p.Y2) c2m_store_tile_fast<MI, NI, true>(acc, p.Y2, voff2, m0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh, p.M);
        if (!p.Y2 || __any(ring))
            c2m_store_tile_fast<MI, NI, true>(acc, Yb, voff, m0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh, p.M);
    }
}

// ------------------------------------------------------------------------------------------------ LDS patch
// 3x3 stride-1 convolutions (forward and data gradient; >80 % of the step's conv FLOPs: VGG, SPADE MLPs, generator,
// decoder): the input patch of a 16-channel chunk ((rows+2) x 34 pixels per channel) is staged in LDS ONCE and serves
// all 9 taps -- the B fragments of every tap are read straight from the patch with a tap offset.  Compared with the
// gather kernel there is no per-tap global gather and no per-tap LDS write of B at all; per 16-deep K-step only the
// weight tile is loaded/stored.  Zero or reflect boundary is resolved while loading the patch.
// Output tile = (BN/32) rows x 32 columns of one image, so each MFMA column block is one contiguous row segment.
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_patch3x3_kernel(const ConvP p) {
    constexpr int BK = 16, TR = BN / 32, PH = TR + 2, PW = 34, PCH = PH * PW, PELEMS = BK * PCH;
    constexpr int PLOADS = (PELEMS + 255) / 256;
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int LDA_S = BM + 4;
    constexpr int A_F4 = BM * BK / 4, APASS = (A_F4 + 255) / 256;
    static_assert(WGM * WGN == 4, "tile");
    __shared__ float sA[2][BK][LDA_S];
    __shared__ float sP[2][PLOADS * 256];      // filled by LDS-DMA: 64 consecutive floats per wave-instruction

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    // tile -> (image, tile row, tile col); XCD-aware order (common.h): row tiles of a pixel tile fastest
    const int tiles_x = (p.Wo + 31) / 32, tiles_y = (p.Ho + TR - 1) / TR;
    const C2mBlock blk = c2m_xcd_block((unsigned)((p.Npix / (p.Ho * p.Wo)) * tiles_y * tiles_x), (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM;
    int tb = blk.x;
    const int tx = tb % tiles_x; tb /= tiles_x;
    const int ty = tb % tiles_y; const int img = tb / tiles_y;        // img = n*To + ot (2-D planes)
    const int oy0 = ty * TR, ox0 = tx * 32;
    const int n_img = img / p.To, t_img = img % p.To;

    // ---- patch loads.  The patch of a chunk is 16 channels x PCH positions, [c][row][col] dense in LDS.  A DMA row is 64
    // consecutive positions of ONE channel: wave w fetches the PROWS rows of channels w, w+4, w+8, w+12, the channel rides in
    // the scalar offset, so a lane owns the PROWS positions wr*64 + lane and the reflect / bounds arithmetic is done PROWS
    // (6 at the 8-row tile) times per thread instead of once per fetched element (22 times: ~550 VALU in the prologue, a third
    // of all VALU of a 288-deep layer; a VALU beside MFMAs costs matrix-pipe cycles, tools/micro/mfma_issue.hip).  The last
    // row of a channel is partial: its lanes beyond PCH are switched off (EXEC), not sent out of range -- an out-of-range
    // lane would write a zero into the next channel's first positions.
    constexpr int PROWS = (PCH + 63) / 64;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long xaddr = (unsigned long)p.X;
    const u32x4 xrs = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const unsigned sp_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)&sP[0][0];
    const unsigned img_byte = (unsigned)(n_img * (int)p.in_sn + t_img * (int)p.in_st) * 4u;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    unsigned pvo[PROWS];
#pragma unroll
    for (int wr = 0; wr < PROWS; ++wr) {
        const int pos = wr * 64 + lane;
        const int r = pos / PW, col = pos % PW;
        int iy = oy0 + p.iy0 + r, ix = ox0 + p.ix0 + col;
        if (p.reflect) {
            iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
        }
        const bool ok = pos < PCH && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;   // (tiles may hang off)
        pvo[wr] = ok ? img_byte + (unsigned)(iy * (int)p.in_sh + ix) * 4u : C2M_OOB;
    }
    auto load_patch = [&](int chunk, int buf) {
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
            const int c = wave_s + 4 * ci, ch = chunk * BK + c;
            u32x4 rsk = xrs;
            rsk[2] = ch < p.cin ? p.x_bytes : 0u;       // channels past K: zero records (the scalar offset is not range-checked)
            const int soff = ch * p.in_sc * 4;
#pragma unroll
            for (int wr = 0; wr < PROWS; ++wr) {
                const unsigned dst = sp_lds + (unsigned)((buf * PLOADS * 256 + c * PCH + wr * 64) * 4);
                if (wr * 64 + 64 <= PCH || wr * 64 + lane < PCH)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                                 :: "s"(dst), "v"(pvo[wr]), "s"(rsk), "s"(soff) : "memory");
            }
        }
    };

    // ---- weight side (same packed K order as the gather kernel with CK = 16: (chunk, tap, channel))
    const int akq = (tid & 3) * 4, arow = tid >> 2;
    const float* __restrict__ aptr[APASS];
#pragma unroll
    for (int s = 0; s < APASS; ++s) {
        int row = m0 + arow + s * 64; row = row < p.M ? row : p.M - 1;
        aptr[s] = p.A + (long)row * p.lda + akq;
    }
    float4 ra[APASS];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int s = 0; s < APASS; ++s)
            if (A_F4 >= 256 || arow + s * 64 < BM) ra[s] = *reinterpret_cast<const float4*>(aptr[s] + kt * BK);
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int s = 0; s < APASS; ++s) {
            const int r = arow + s * 64;
            if (A_F4 >= 256 || r < BM) {
                sA[buf][akq + 0][r] = ra[s].x; sA[buf][akq + 1][r] = ra[s].y;
                sA[buf][akq + 2][r] = ra[s].z; sA[buf][akq + 3][r] = ra[s].w;
            }
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // B fragment base inside the patch: channel (lane>>5), tile row (wn*NI + j), column (lane&31)
    int pbase[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) pbase[j] = (lane >> 5) * PCH + (wn * NI + j) * PW + (lane & 31);

    // split-K over whole channel chunks (gridDim.z); ksteps_per_split holds CHUNKS per split for this kernel
    const int chunk_beg = blk.z * p.ksteps_per_split;
    int chunk_end = chunk_beg + p.ksteps_per_split; chunk_end = chunk_end < p.nchunks ? chunk_end : p.nchunks;
    load_patch(chunk_beg, 0);
    load_a(chunk_beg * 9);
    store_a(0);
    // The patch DMA is inline asm the compiler's s_waitcnt insertion does not see.  It used to be covered only implicitly, by the
    // wait for the A-tile load issued after it (loads return in order) -- but with BM = 32 the threads >= 128 load no A element,
    // so waves 2 and 3 reached the barrier WITHOUT ever waiting for the channels they had fetched, and their part of the patch
    // could still be in flight when the other waves read it.  Alone on the chip the DMA always won the race; next to ANY other
    // kernel that keeps the memory system busy it did not (round 4, tools/concurrency_stress.py: wrong forward values and
    // gradients next to torch.add on another stream).  Every barrier that publishes a freshly fetched patch buffer now has an
    // explicit wait in front of it (free: the waves that load A were waiting there already).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0, pcur = 0;
    for (int chunk = chunk_beg; chunk < chunk_end; ++chunk) {
        const bool more_chunks = chunk + 1 < chunk_end;
        if (more_chunks) load_patch(chunk + 1, pcur ^ 1);  // the other buffer was last read before the previous barrier
#ifdef C2M_PATCH_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int tap = 0; tap < 9; ++tap) {
            const int kt = chunk * 9 + tap;
            const bool more = tap < 8 || more_chunks;
            if (more) load_a(kt + 1);
            const int toff = p.pty[tap / 3] * PW + p.ptx[tap % 3];
            float a[BK / 2][MI], b[BK / 2][NI];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const int krow = kk * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kk][i] = sA[cur][krow][wm * TM + i * 32 + (lane & 31)];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[kk][j] = sP[pcur][pbase[j] + kk * 2 * PCH + toff];
            }
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            {   // fragment reads two k-pairs ahead of their MFMAs (only the first reads of a tap are waited for)
                constexpr int NMF = (BK / 2) * MI * NI, RPM = (MI + NI + MI * NI - 1) / (MI * NI);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
                for (int g = 0; g < NMF; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (more) store_a(cur ^ 1);
            if (tap == 8 && more_chunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next chunk's patch DMA (see the prologue)
            __syncthreads();
            cur ^= 1;
        }
        pcur ^= 1;
    }

    // ---- epilogue
    const bool direct = blk.nz == 1;
    constexpr bool yh = false;                             // the fp32 kernels write fp32
    constexpr int yes = 4;
    float* __restrict__ Yb = p.Y + (long)blk.z * p.slab_stride;
    {
        const int row0 = __builtin_amdgcn_readfirstlane(m0 + wm * TM);
        if (row0 + MI * 32 <= p.M) {
            unsigned voff[NI], voff2[NI];
            bool ring = false;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int oy = oy0 + wn * NI + j, ox = ox0 + (lane & 31);
                const bool in = oy < p.Ho && ox < p.Wo;
                const long e = p.out_off + (long)n_img * p.out_sn + (long)t_img * p.out_st + (long)oy * p.out_sh +
                               (long)ox * p.out_sw + 4L * (lane >> 5) * p.out_sc;
                voff[j] = in ? (unsigned)(e * 4) : 0x80000000u;
                voff2[j] = 0x80000000u;
                if (p.Y2) {
                    const int tp = t_img * p.ps_t + p.po_t - p.lo_t, yp = oy * p.ps_y + p.po_y - p.lo_y,
                              xp = ox * p.ps_x + p.po_x - p.lo_x;
                    if (in && (unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y &&
                        (unsigned)xp < (unsigned)p.ext_x) {
                        const long e2 = (long)n_img * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp +
                                        4L * (lane >> 5) * p.y2_sc;
                        voff2[j] = (unsigned)(e2 * yes);
                        voff[j] = 0x80000000u;
                    }
                }
                ring = ring || voff[j] != 0x80000000u;
            }
            if (p.Y2) c2m_store_tile_fast<MI, NI>(acc, p.Y2, voff2, row0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh);
            if (!p.Y2 || __any(ring))
                c2m_store_tile_fast<MI, NI>(acc, Yb, voff, row0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh);
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int oy = oy0 + wn * NI + j, ox = ox0 + (lane & 31);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float* __restrict__ yb = Yb + p.out_off + (long)n_img * p.out_sn + (long)t_img * p.out_st + (long)oy * p.out_sh +
                                 (long)ox * p.out_sw;
        long row_stride = p.out_sc;
        if (p.Y2) {
            const int tp = t_img * p.ps_t + p.po_t - p.lo_t, yp = oy * p.ps_y + p.po_y - p.lo_y,
                      xp = ox * p.ps_x + p.po_x - p.lo_x;
            if ((unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y && (unsigned)xp < (unsigned)p.ext_x) {
                yb = p.Y2 + (long)n_img * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp;
                row_stride = p.y2_sc;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) {
                    float v = acc[i][j][r];
                    if (direct) {
                        if (p.bias) v += p.bias[row];
                        v = c2m_act(v, p.act, p.slope);
                    }
                    yb[(long)row * row_stride] = v;
                }
            }
    }
}

template <int BM, int BN, int WGM, int WGN>
static int launch_patch(const ConvP& p, int splits, hipStream_t s) {
    constexpr int TR = BN / 32;
    const long tiles = (long)(p.Npix / (p.Ho * p.Wo)) * ((p.Ho + TR - 1) / TR) * ((p.Wo + 31) / 32);
    dim3 grid((unsigned)(tiles * c2m_cdiv(p.M, BM) * splits));
    hipLaunchKernelGGL((conv_patch3x3_kernel<BM, BN, WGM, WGN>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ LDS patch, bf16
// The bf16 form of the kernel above (BASELINE configs[2-4]): v_mfma_f32_32x32x16_bf16 is 16x the fp32 MFMA rate, so the
// kernel lives or dies by operand delivery.  Per 16-channel chunk
//   * the (8+2) x 34 input patch is gathered ONCE from the bf16 NCHW activations (24 coalesced 2-byte loads per thread: on
//     NCHW the 8 channels a lane needs sit 2*H*W bytes apart; a 16-byte-load + 4x8 register-transpose form of this fetch was
//     measured 5-25 % SLOWER, profiles/r03_ab_bf16_patch_wide_vs_narrow.txt), packed in registers and stored as
//     [pixel][16 channels] (32 B per pixel): the B fragment of ANY of the 9 taps is then one conflict-free ds_read_b128 at a
//     pixel offset -- no per-tap gather, no conversion;
//   * the weights arrive pre-packed in bf16 as [chunk][tap][row][16 channels] (c2m_pack_weights_bf16_patch): one 16-byte
//     load + one ds_write_b128 per 512 MACs of MFMA work, and the A fragment is one ds_read_b128;
//   * one barrier per chunk (72 MFMAs per wave at BM = 128), next chunk's global loads in flight during the MFMAs.
// Tile: BM output channels x (8 rows x 32 columns); wave w owns rows 2w, 2w+1 for all BM channels (MI = BM/32, NI = 2).
template <int BM>
__global__ __launch_bounds__(256) void conv_patch3x3_bf16_kernel(const ConvP p) {
    constexpr int TR = 8, PH = TR + 2, PW = 34, NPIX = PH * PW;           // 340 patch pixels
    constexpr int MI = BM / 32, NI = 2;
    constexpr int A_UNITS = 9 * BM * 2;                                    // 16-byte units per chunk: [tap][half][row]
    constexpr int APT = (A_UNITS + 255) / 256;
    // LDS: double-buffered weight image + patch (73.7 + 21.8 KB at BM = 128: one workgroup per CU, one barrier per chunk;
    // a single-buffered weight image with two workgroups per CU and two barriers per chunk measured 20 % slower).
    // Both images keep the two 8-channel halves in separate planes ([half][pixel], [tap][half][row]): the 32 lanes of a
    // ds_read_b128 half-wave then read 512 contiguous bytes (interleaved halves were a 2-way bank conflict on every read).
    __shared__ uint4 sA[2][A_UNITS];
    __shared__ uint4 sP[2][NPIX * 2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.Wo + 31) / 32, tiles_y = (p.Ho + TR - 1) / TR;
    const C2mBlock blk = c2m_xcd_block((unsigned)((p.Npix / (p.Ho * p.Wo)) * tiles_y * tiles_x), (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM;
    int tb = blk.x;
    const int tx = tb % tiles_x; tb /= tiles_x;
    const int ty = tb % tiles_y; const int img = tb / tiles_y;            // img = n*To + ot (2-D planes)
    const int oy0 = ty * TR, ox0 = tx * 32;
    const int n_img = img / p.To, t_img = img % p.To;

    // ---- patch gather: three rounds of (pixel, 8-channel half); the half is wave-uniform in every round
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const unsigned img_byte = (unsigned)(n_img * (int)p.in_sn + t_img * (int)p.in_st) * 2u;     // X is bf16 (2-byte elements)
    const int ppix[3] = {tid, tid, 256 + (tid & 127)};
    const int phalf[3] = {0, 1, wave >> 1};
    unsigned pvo[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int e = ppix[r];
        const int row = e / PW, col = e % PW;
        int iy = oy0 + p.iy0 + row, ix = ox0 + p.ix0 + col;
        bool ok = e < NPIX;
        if (p.reflect) {
            iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
        }
        ok = ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        pvo[r] = ok ? img_byte + (unsigned)(iy * (int)p.in_sh + ix) * 2u : C2M_OOB;
    }
    struct Stage { unsigned short pv[3][8]; };
    auto fetch_patch = [&](int chunk, unsigned short (&pv)[3][8]) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // channels past Cin (the last chunk of a Cin % 16 != 0 layer): the scalar offset is NOT range-checked by the
                // hardware, so they get a zero-record descriptor (wave-uniform select, SALU only) and read 0 -- never the
                // memory behind X (a NaN there times the zero-padded weight would poison the pixel)
                const int ch = chunk * 16 + phalf[r] * 8 + j;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.X), 0, ch < p.cin ? p.x_bytes : 0, 0x00020000);
                const int soff = ch * p.in_sc * 2;
                pv[r][j] = __builtin_amdgcn_raw_buffer_load_b16(rs, pvo[r], soff, 0);
            }
    };
    auto stash_patch = [&](int buf, const unsigned short (&pv)[3][8]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            bf16x8 q;
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = __builtin_bit_cast(__bf16, pv[r][j]);
            if (r < 2 || ppix[r] < NPIX) sP[buf][phalf[r] * NPIX + ppix[r]] = __builtin_bit_cast(uint4, q);
        }
    };
    // ---- weights: global [chunk][tap][Mpad rows][2 halves] 16-byte units -> LDS [tap][half][row]
    const uint4* __restrict__ Ab = reinterpret_cast<const uint4*>(p.A);
    const long mpad2 = (long)p.lda * 2;                                    // lda = padded row count of the packed matrix
    long aoff[APT];
    int adst[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int u = tid + i * 256;
        const int tap = u / (BM * 2), rem = u % (BM * 2);
        aoff[i] = (long)tap * mpad2 + (long)m0 * 2 + rem;
        adst[i] = (tap * 2 + (rem & 1)) * BM + (rem >> 1);
    }
    auto fetch_a = [&](int chunk, uint4 (&av)[APT]) {
#pragma unroll
        for (int i = 0; i < APT; ++i)
            if (A_UNITS % 256 == 0 || tid + i * 256 < A_UNITS) av[i] = Ab[(long)chunk * 9 * mpad2 + aoff[i]];
    };
    auto stash_a = [&](int buf, const uint4 (&av)[APT]) {
#pragma unroll
        for (int i = 0; i < APT; ++i)
            if (A_UNITS % 256 == 0 || tid + i * 256 < A_UNITS) sA[buf][adst[i]] = av[i];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int pbase[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) pbase[j] = (lane >> 5) * NPIX + (wave * NI + j) * PW + (lane & 31);
    const int abase = (lane >> 5) * BM + (lane & 31);

    const int chunk_beg = blk.z * p.ksteps_per_split;
    int chunk_end = chunk_beg + p.ksteps_per_split; chunk_end = chunk_end < p.nchunks ? chunk_end : p.nchunks;
    // The activation gather (HBM / Infinity Cache) runs TWO chunks ahead of its LDS store (two register sets): one chunk of
    // MFMAs (~1.1 us) does not cover its loaded latency (~2.2 us per chunk with every CU fetching at once).  The weight
    // image comes from L2 and runs one chunk ahead (a second set of 36 registers would not fit the 256 VGPRs).
    Stage s0, s1;
    uint4 av[APT];
    const int nck = chunk_end - chunk_beg;
    fetch_a(chunk_beg, av);
    fetch_patch(chunk_beg, s0.pv);
    stash_a(0, av);
    stash_patch(0, s0.pv);
    __syncthreads();
    // vmcnt retires in order: the weight loads of chunk c+1 are issued BEFORE the patch loads of chunk c+2 / c+3, so
    // waiting for them at the end of chunk c leaves the deep patch loads in flight
    if (nck > 1) fetch_patch(chunk_beg + 1, s0.pv);
    if (nck > 1) fetch_a(chunk_beg + 1, av);
    if (nck > 2) fetch_patch(chunk_beg + 2, s1.pv);
    // fragments are read one tap ahead of the MFMAs that use them (two register sets): with one wave per SIMD nothing else
    // hides the LDS latency -- reading them right before use left the matrix pipe idle for most of every tap
    struct Frag { bf16x8 a[MI], b[NI]; };
    auto read_frag = [&](int buf, int tap, Frag& f) {
        const int toff = p.pty[tap / 3] * PW + p.ptx[tap % 3];
#pragma unroll
        for (int i = 0; i < MI; ++i) f.a[i] = __builtin_bit_cast(bf16x8, sA[buf][tap * 2 * BM + i * 32 + abase]);
#pragma unroll
        for (int j = 0; j < NI; ++j) f.b[j] = __builtin_bit_cast(bf16x8, sP[buf][pbase[j] + toff]);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };
    constexpr int NPAIR = MI * NI < MI + NI ? MI * NI : MI + NI;
    int cur = 0;
    Frag f0, f1;
    read_frag(0, 0, f0);
#define PB_TAP(T, FC, FN)                                                          \
        read_frag(cur, (T) + 1, FN);                                               \
        mma(FC);                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < NPAIR; ++g_) {                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     \
        }                                                                          \
        if (MI * NI > NPAIR) __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - NPAIR, 0);   \
        if (MI + NI > NPAIR) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI - NPAIR, 0);   \
        __builtin_amdgcn_sched_barrier(0);
    // one chunk: 9 taps from buffer cur; then the staged next chunk (set SC) goes to buffer cur ^ 1 and SC is refilled
    // with the chunk after next + 1
#define PB_CHUNK(SC)                                                               \
    do {                                                                           \
        PB_TAP(0, f0, f1) PB_TAP(1, f1, f0) PB_TAP(2, f0, f1) PB_TAP(3, f1, f0)    \
        PB_TAP(4, f0, f1) PB_TAP(5, f1, f0) PB_TAP(6, f0, f1) PB_TAP(7, f1, f0)    \
        mma(f0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                         \
        if (chunk + 1 < chunk_end) { stash_a(cur ^ 1, av); stash_patch(cur ^ 1, SC.pv); }   \
        __syncthreads();                                                           \
        if (chunk + 2 < chunk_end) fetch_a(chunk + 2, av);                         \
        if (chunk + 3 < chunk_end) fetch_patch(chunk + 3, SC.pv);                  \
        cur ^= 1;                                                                  \
        read_frag(cur, 0, f0);                                                     \
        ++chunk;                                                                   \
    } while (0)
    int chunk = chunk_beg;
    while (chunk < chunk_end) {
        PB_CHUNK(s0);
        if (chunk < chunk_end) PB_CHUNK(s1);
    }
#undef PB_CHUNK
#undef PB_TAP

    // ---- epilogue (same mapping as the fp32 patch kernel)
    const bool direct = blk.nz == 1;
    const bool yh = p.yh && direct;                        // bf16 output elements (split-K slabs are fp32)
    float* __restrict__ Yb = p.Y + (long)blk.z * p.slab_stride;
    // Vector path: a dword store per accumulator register (128 per wave) is store-ISSUE bound (~5 B/clk/CU: 27k cycles for
    // the 128 KB tile against 37k cycles of MFMAs at Cin = 256).  The tile goes through LDS instead (the weight buffers are
    // free now): [channel][64 pixels] per wave, 64 channels at a time, read back as float4 along the pixels and stored with
    // 16-byte stores (32 per wave instead of 128), still full 128-byte row segments per channel.
    if (yh && !p.Y2 && p.out_sw == 1 && (p.Wo & 7) == 0 && (p.out_off & 7) == 0 && (p.out_sc & 7) == 0 && (p.out_sn & 7) == 0 &&
        (p.out_st & 7) == 0 && (p.out_sh & 7) == 0 && ((uintptr_t)p.Y & 15) == 0 && BM >= 64) {
        // bf16 output: the same staging through LDS, read back as 8 pixels per lane -> one 16-byte store of 8 bf16
        float* __restrict__ T = reinterpret_cast<float*>(&sA[0][0]) + wave * (64 * 64);
        bf16_t* __restrict__ yimg = reinterpret_cast<bf16_t*>(p.Y) + p.out_off + (long)n_img * p.out_sn + (long)t_img * p.out_st;
#pragma unroll
        for (int h = 0; h < MI / 2; ++h) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jj = 0; jj < NI; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cl = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        float v = acc[2 * h + q][jj][r];
                        const int row = m0 + h * 64 + cl;
                        if (p.bias && row < p.M) v += p.bias[row];
                        T[cl * 64 + jj * 32 + (lane & 31)] = c2m_act(v, p.act, p.slope);
                    }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int cl = it * 8 + (lane >> 3), px = (lane & 7) * 8;
                const float4 v0 = *reinterpret_cast<const float4*>(&T[cl * 64 + px]);
                const float4 v1 = *reinterpret_cast<const float4*>(&T[cl * 64 + px + 4]);
                const int row = m0 + h * 64 + cl;
                const int oy = oy0 + wave * NI + (px >> 5), ox = ox0 + (px & 31);
                if (row < p.M && oy < p.Ho && ox < p.Wo) {
                    const bf16x8 o = {(bf16_t)v0.x, (bf16_t)v0.y, (bf16_t)v0.z, (bf16_t)v0.w,
                                      (bf16_t)v1.x, (bf16_t)v1.y, (bf16_t)v1.z, (bf16_t)v1.w};
                    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(yimg + (long)row * p.out_sc + (long)oy * p.out_sh + ox));
                }
            }
        }
        return;
    }
    if (!yh && !p.Y2 && p.out_sw == 1 && (p.Wo & 3) == 0 && (p.out_off & 3) == 0 && (p.out_sc & 3) == 0 && (p.out_sn & 3) == 0 &&
        (p.slab_stride & 3) == 0 && ((uintptr_t)p.Y & 15) == 0 && BM >= 64) {
        float* __restrict__ T = reinterpret_cast<float*>(&sA[0][0]) + wave * (64 * 64);     // 16 KB per wave
        float* __restrict__ yimg = Yb + p.out_off + (long)n_img * p.out_sn + (long)t_img * p.out_st;
#pragma unroll
        for (int h = 0; h < MI / 2; ++h) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int jj = 0; jj < NI; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cl = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                        float v = acc[2 * h + q][jj][r];
                        if (direct) {
                            const int row = m0 + h * 64 + cl;
                            if (p.bias && row < p.M) v += p.bias[row];
                            v = c2m_act(v, p.act, p.slope);
                        }
                        T[cl * 64 + jj * 32 + (lane & 31)] = v;
                    }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int cl = it * 4 + (lane >> 4), px = (lane & 15) * 4;
                const float4 v = *reinterpret_cast<const float4*>(&T[cl * 64 + px]);
                const int row = m0 + h * 64 + cl;
                const int oy = oy0 + wave * NI + (px >> 5), ox = ox0 + (px & 31);
                if (row < p.M && oy < p.Ho && ox < p.Wo) {
                    // non-temporal: the 100+ MB output would otherwise push the input patches out of L2 / Infinity Cache
                    // while they are still being gathered (+4...12 % on the configs[2] layers)
                    f32x4 vv = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(yimg + (long)row * p.out_sc + (long)oy * p.out_sh + ox));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int oy = oy0 + wave * NI + j, ox = ox0 + (lane & 31);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float* ybase = Yb;                                     // element type: float, or bf16_t when yh (then blk.z == 0)
        long yidx = p.out_off + (long)n_img * p.out_sn + (long)t_img * p.out_st + (long)oy * p.out_sh + (long)ox * p.out_sw;
        long row_stride = p.out_sc;
        if (p.Y2) {
            const int tp = t_img * p.ps_t + p.po_t - p.lo_t, yp = oy * p.ps_y + p.po_y - p.lo_y,
                      xp = ox * p.ps_x + p.po_x - p.lo_x;
            if ((unsigned)tp < (unsigned)p.ext_t && (unsigned)yp < (unsigned)p.ext_y && (unsigned)xp < (unsigned)p.ext_x) {
                ybase = p.Y2;
                yidx = (long)n_img * p.y2_sn + (long)tp * p.y2_st + (long)yp * p.y2_sh + xp;
                row_stride = p.y2_sc;
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) {
                    float v = acc[i][j][r];
                    if (direct) {
                        if (p.bias) v += p.bias[row];
                        v = c2m_act(v, p.act, p.slope);
                    }
                    if (yh) reinterpret_cast<bf16_t*>(ybase)[yidx + (long)row * row_stride] = (bf16_t)v;
                    else ybase[yidx + (long)row * row_stride] = v;
                }
            }
    }
}

template <int BM>
static int launch_patch_bf16(const ConvP& p, int splits, hipStream_t s) {
    const long tiles = (long)(p.Npix / (p.Ho * p.Wo)) * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32);
    dim3 grid((unsigned)(tiles * c2m_cdiv(p.M, BM) * splits));
    hipLaunchKernelGGL((conv_patch3x3_bf16_kernel<BM>), grid, dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

// out[i] = act( sum_z slab[z][i] + bias[(i / chan_stride) % M] ), fixed order
template <class T>
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, T* __restrict__ out,
                                     const float* __restrict__ bias, long total, int S, long chan_stride, int M, int act,
                                     float slope) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int z = 0; z < S; ++z) acc += slab[(long)z * total + i];
        if (bias) acc += bias[(int)((i / chan_stride) % M)];
        c2m_st(out, i, c2m_act(acc, act, slope));
    }
}

// 16-byte form: 4 consecutive outputs share a channel when chan_stride % 4 == 0.  I = 32-bit indices whenever the slab
// fits (the bias channel needs a division per element; in 64 bits it costs more than the S loads)
template <typename I, class T = float>
__global__ void splitk_reduce_vec_kernel(const float4* __restrict__ slab, T* __restrict__ out,
                                         const float* __restrict__ bias, long total4_, int S, long chan_stride4_, int M,
                                         int act, float slope) {
    const I total4 = (I)total4_, chan_stride4 = (I)chan_stride4_;
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total4; i += (I)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < S; ++z) {
            const float4 v = slab[(long)z * total4_ + i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (bias) {
            const float b = bias[(int)((i / chan_stride4) % (I)M)];
            acc.x += b; acc.y += b; acc.z += b; acc.w += b;
        }
        c2m_st4(out + 4 * (long)i, make_float4(c2m_act(acc.x, act, slope), c2m_act(acc.y, act, slope), c2m_act(acc.z, act, slope),
                                               c2m_act(acc.w, act, slope)));
    }
}

template <int NS> static int launch_thin_fwd(const ConvP& p, hipStream_t s);
template <int KW> static int launch_thin_rows(const ConvP& p, int ns, int ntg, int Cin, int xdesc, hipStream_t s);

#ifndef C2M_IGEMM_U
#define C2M_IGEMM_U 1
#endif
#ifndef C2M_BF16_U
#define C2M_BF16_U 2      // bf16: 4 MFMAs per 16-deep step and wave, so two steps per barrier
#endif

template <int BM, int BN, int WGM, int WGN>
static int launch_igemm(const ConvP& p, int ns, int splits, hipStream_t s, bool bf16) {
    dim3 grid((unsigned)c2m_cdiv(p.Npix, BN) * c2m_cdiv(p.M, BM) * splits * p.ncls);
    constexpr int U = (BM == 128) ? C2M_IGEMM_U : 1;
    if (bf16) {
        constexpr int UB = C2M_BF16_U;
        switch (ns) {
            case 1: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 1, UB, true>), grid, dim3(256), 0, s, p); break;
            case 2: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 2, UB, true>), grid, dim3(256), 0, s, p); break;
            case 4: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 4, UB, true>), grid, dim3(256), 0, s, p); break;
            default: return (int)hipErrorInvalidValue;
        }
        return (int)hipGetLastError();
    }
    switch (ns) {
        case 1: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 1, U>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 2, 1>), grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WGM, WGN, 4, 1>), grid, dim3(256), 0, s, p); break;
        default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

static void igemm_tile(int M, int& BM, int& BN) {
    if (M <= 32) { BM = 32; BN = 256; }
    else if (M <= 64) { BM = 64; BN = 128; }
    else { BM = 128; BN = 128; }
}

// Number of K splits the launch will use (the caller sizes the slab = splits * slab_stride floats when > 1).
// Blocks are dispatched dynamically, so a grid only loses time when it is too small to keep every CU busy to the end:
// below ~4 blocks per CU the K loop is split until there are ~5 per CU (e.g. 640 tiles = 2.5/CU would leave the CUs
// that got 2 blocks idle for a third of the launch).
C2M_API int c2m_conv_igemm_splits(int M, int nk, int Npix) {
    int BM, BN;
    igemm_tile(M, BM, BN);
    const long tiles = (long)c2m_cdiv(M, BM) * c2m_cdiv(Npix, BN);
#ifndef C2M_SPLIT_TARGET
#define C2M_SPLIT_TARGET 1280
#endif
    if (tiles >= 768 || nk < 8) return 1;
    long S = (C2M_SPLIT_TARGET + tiles - 1) / tiles;
    if (S > nk / 4) S = nk / 4;
    if (S > 128) S = 128;
    if (S < 2) return 1;
    const int per = c2m_cdiv(nk, (int)S);
    return c2m_cdiv(nk, per);                 // every split owns >= 1 K-step
}

// geom[] layout (int64): see include/c2m_hip.h
C2M_API int c2m_conv_igemm(const float* A, const void* X, void* Y, void* Y_interior, const float* bias,
                           const int* ktab, const int64_t* g, int act, float slope, void* stream) {
    C2M_ENTER();
    ConvP p;
    p.A = A; p.X = (const float*)X; p.Y = (float*)Y; p.bias = bias; p.ktab = reinterpret_cast<const int4*>(ktab);
    p.Y2 = (float*)Y_interior;
    p.ps_t = (int)g[C2M_G_PS_T]; p.ps_y = (int)g[C2M_G_PS_Y]; p.ps_x = (int)g[C2M_G_PS_X]; p.po_t = (int)g[C2M_G_PO_T]; p.po_y = (int)g[C2M_G_PO_Y];
    p.po_x = (int)g[C2M_G_PO_X]; p.lo_t = (int)g[C2M_G_LO_T]; p.lo_y = (int)g[C2M_G_LO_Y]; p.lo_x = (int)g[C2M_G_LO_X]; p.ext_t = (int)g[C2M_G_EXT_T];
    p.ext_y = (int)g[C2M_G_EXT_Y]; p.ext_x = (int)g[C2M_G_EXT_X]; p.y2_sn = g[C2M_G_Y2_SN]; p.y2_sc = g[C2M_G_Y2_SC]; p.y2_st = g[C2M_G_Y2_ST]; p.y2_sh = g[C2M_G_Y2_SH];
    p.M = (int)g[C2M_G_M]; p.nk = (int)g[C2M_G_NK]; p.lda = (int)g[C2M_G_LDA];
    p.Npix = (int)g[C2M_G_NPIX]; p.To = (int)g[C2M_G_TO]; p.Ho = (int)g[C2M_G_HO]; p.Wo = (int)g[C2M_G_WO];
    p.Ti = (int)g[C2M_G_TI]; p.Hi = (int)g[C2M_G_HI]; p.Wi = (int)g[C2M_G_WI];
    p.st = (int)g[C2M_G_ST]; p.sh = (int)g[C2M_G_SH]; p.sw = (int)g[C2M_G_SW];
    p.in_sn = g[C2M_G_IN_SN]; p.in_st = g[C2M_G_IN_ST]; p.in_sh = g[C2M_G_IN_SH];
    p.out_sn = g[C2M_G_OUT_SN]; p.out_sc = g[C2M_G_OUT_SC]; p.out_st = g[C2M_G_OUT_ST]; p.out_sh = g[C2M_G_OUT_SH]; p.out_sw = g[C2M_G_OUT_SW]; p.out_off = g[C2M_G_OUT_OFF];
    p.reflect = (int)g[C2M_G_REFLECT]; p.is3d = (int)g[C2M_G_IS3D];
    const int ns = (int)g[C2M_G_NS];
    p.in_sc = (int)g[C2M_G_IN_SC];
    const int splits = (int)g[C2M_G_SPLITS];           // 1, or the value returned by c2m_conv_igemm_splits
    p.slab_stride = g[C2M_G_SLAB_STRIDE];
    if (g[C2M_G_X_BYTES] <= 0 || g[C2M_G_X_BYTES] >= 0x80000000LL) return (int)hipErrorInvalidValue;   // X must be < 2 GiB
    p.x_bytes = (unsigned)g[C2M_G_X_BYTES];
    p.act = act; p.slope = slope;
    p.xh = (int)g[C2M_G_X_TYPE]; p.yh = (int)g[C2M_G_Y_TYPE];       // element types of X and of Y / Y_interior: 0 fp32, 1 bf16 (dtype.h)
    if ((p.xh | p.yh) & ~1) return (int)hipErrorInvalidValue;
    if (p.M <= 0 || p.Npix <= 0) return 0;
    if (p.nk <= 0 || (p.lda & 3) || (((uintptr_t)A) & 15) || splits < 1) return (int)hipErrorInvalidValue;
    if (p.Y2 && splits != 1) return (int)hipErrorInvalidValue;   // the two-target epilogue is a direct-store feature
    p.ksteps_per_split = c2m_cdiv(p.nk, splits);
    if (!g[C2M_G_PATCH] && c2m_cdiv(p.nk, p.ksteps_per_split) != splits) return (int)hipErrorInvalidValue;   // empty split
    p.splits = splits;
    p.ncls = g[C2M_G_NCLS] > 1 ? (int)g[C2M_G_NCLS] : 1;
    if (p.ncls > 8) return (int)hipErrorInvalidValue;
    p.a_cls = g[C2M_G_A_CLS]; p.ktab_cls = (int)g[C2M_G_KTAB_CLS];
    p.out_off_c[0] = p.out_off; p.po_c[0][0] = p.po_t; p.po_c[0][1] = p.po_y; p.po_c[0][2] = p.po_x;
    for (int c = 0; c < p.ncls && p.ncls > 1; ++c) {
        p.out_off_c[c] = g[C2M_G_CLS_OUT_OFF + c];
        for (int d = 0; d < 3; ++d) p.po_c[c][d] = (int)g[C2M_G_CLS_PO + 3 * c + d];     // 96 .. 119 (geom[90..92] are the type / form flags)
    }
    if (p.ncls > 1 && (g[C2M_G_PATCH] || (p.a_cls & 3))) return (int)hipErrorInvalidValue;   // gather kernel only
    hipStream_t s = (hipStream_t)stream;
    if (g[C2M_G_G8] == 1) {
        // NC8 gather form (conv_gather_nc8_kernel): X = NC8 of the gathered bf16 tensor ([N][ceil(C/8)][Ti*Hi*Wi][8], geom[32] its bytes),
        // A = c2m_pack_weights_bf16_gather image(s) of this launch's first class, ktab = [ncls][taps] {dt, dy, dx, 1}, nk = taps *
        // ceil(C/16) K-steps in (tap, chunk) order; C = geom[28], taps = geom[29]; geom[95] = tile variant (0 = rule)
        const int C = (int)g[C2M_G_CIN], taps = (int)g[C2M_G_TAPS];
        if (g[C2M_G_PRECISION] != 1 || !p.xh || g[C2M_G_PATCH] || C <= 0 || taps <= 0 || taps > 64 || (((uintptr_t)A | (uintptr_t)X) & 15))
            return (int)hipErrorInvalidValue;
        p.g8_nch = c2m_cdiv(C, 16); p.g8_ntaps = taps; p.g8_CB = c2m_cdiv(C, 8); p.g8_Mpad = c2m_cdiv(p.M, 128) * 128;
        if (p.nk != taps * p.g8_nch || p.in_st != (long)p.Hi * p.Wi || p.in_sh != p.Wi) return (int)hipErrorInvalidValue;
        const long plane = (long)p.Ti * p.Hi * p.Wi * 16, ab = (long)p.nk * 2 * p.g8_Mpad * 16;
        if (plane * p.g8_CB > (long)p.x_bytes || p.x_bytes % (plane * p.g8_CB) || ab * p.ncls >= 0x80000000LL)
            return (int)hipErrorInvalidValue;
        p.g8_plane_bytes = (unsigned)plane; p.g8_a_bytes = (unsigned)ab;
        const unsigned ptiles = (unsigned)c2m_cdiv(p.Npix, 256);
        int v = (int)g[C2M_G_G8_VARIANT] & 255;
        p.g8_dbg = (int)(g[C2M_G_G8_VARIANT] >> 8);
        // (kernel trace of tools/ab_g8.py on one box, all variants per shape: 64-row tiles 2 stages x 3 workgroups per CU >= 3 stages x 2
        // everywhere (45-row (4,4,4) data gradient 160 vs 209 us); 128-row tiles: variants 3 / 6 within 3 %; <= 32 rows: 4 K-steps per stage)
        if (v == 0) v = p.M <= 32 ? 1 : ((p.M <= 64 || (p.M % 128 >= 1 && p.M % 128 <= 64)) ? 5 : 3);
#define G8_LAUNCH(BM, UU, NB, WG) do {                                                                                     \
            dim3 grid(ptiles * (unsigned)c2m_cdiv(p.M, BM) * (unsigned)(splits * p.ncls));                                  \
            hipLaunchKernelGGL((conv_gather_nc8_kernel<BM, UU, NB, WG>), grid, dim3(256), 0, s, p); } while (0)
        switch (v) {
            case 1: G8_LAUNCH(32, 4, 2, 2); break;         // (rows, K-steps per stage, stages, workgroups per CU)
            case 2: G8_LAUNCH(64, 2, 3, 2); break;
            case 3: G8_LAUNCH(128, 2, 3, 2); break;
            case 4: G8_LAUNCH(64, 1, 4, 3); break;
            case 5: G8_LAUNCH(64, 2, 2, 3); break;
            case 6: G8_LAUNCH(128, 1, 4, 2); break;
            case 7: G8_LAUNCH(32, 2, 3, 2); break;
            default: return (int)hipErrorInvalidValue;
        }
#undef G8_LAUNCH
        return (int)hipGetLastError();
    }
    if (g[C2M_G_PATCH]) {                                           // LDS-patch path (3x3 stride 1, chosen by the host plan)
        if (ns != 1 || p.st != 1 || p.sh != 1 || p.sw != 1) return (int)hipErrorInvalidValue;
        p.iy0 = (int)g[C2M_G_PATCH_IY0]; p.ix0 = (int)g[C2M_G_PATCH_IX0];
        for (int i = 0; i < 3; ++i) { p.pty[i] = (int)g[C2M_G_PATCH_TY + i]; p.ptx[i] = (int)g[C2M_G_PATCH_TX + i]; }
        p.nchunks = p.nk / 9;
        p.cin = (int)g[C2M_G_CIN];
        if (p.nchunks * 9 != p.nk || p.cin <= 0 || p.cin > p.nchunks * 16) return (int)hipErrorInvalidValue;
        p.ksteps_per_split = c2m_cdiv(p.nchunks, splits);           // chunks per split
        if (c2m_cdiv(p.nchunks, p.ksteps_per_split) != splits) return (int)hipErrorInvalidValue;
        if (g[C2M_G_PRECISION] == 1) {
            // bf16: A = c2m_pack_weights_bf16_patch output, lda = its padded row count (a multiple of 128); X is bf16
            if (p.lda % 128 != 0 || p.lda < p.M || !p.xh) return (int)hipErrorInvalidValue;
            if (p.M <= 32)      return launch_patch_bf16<32>(p, splits, s);
            // 64-row tiles keep TWO workgroups on a CU (59 KB of LDS each; the 128-row tile's 95 KB leave one wave per SIMD
            // with nothing to hide its stalls): +5...14 % up to 256 input channels, -5 % at 512 (the weight image is
            // streamed twice as often) -- A/B on one box, bf16 tensors
            else if (p.M <= 64 || p.cin <= 256) return launch_patch_bf16<64>(p, splits, s);
            else                return launch_patch_bf16<128>(p, splits, s);
        }
        if (p.xh || p.yh) return (int)hipErrorInvalidValue;        // the fp32 kernels read and write fp32
        if (p.M <= 32)      return launch_patch<32, 256, 1, 4>(p, splits, s);
        else if (p.M <= 64) return launch_patch<64, 128, 2, 2>(p, splits, s);
        else                return launch_patch<128, 128, 2, 2>(p, splits, s);
    }
    if (p.M <= 4 && splits == 1 && p.Npix >= 16384 && !p.Y2 && p.ncls == 1) {      // thin output: vector-ALU kernel
        if (p.xh || p.yh) return (int)hipErrorInvalidValue;        // fp32 in, fp32 out (the host casts bf16 activations)
        // g[C2M_G_SQUARE_KW] = +-KW: 2-D stride-1 square KW x KW tap set in row-major order (dx ascending / descending)
        const int kw = (int)(g[C2M_G_SQUARE_KW] < 0 ? -g[C2M_G_SQUARE_KW] : g[C2M_G_SQUARE_KW]);
        if (p.M <= 3 && (kw == 3 || kw == 7) && !p.is3d && p.To == 1 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.Wo % 4 == 0 &&
            g[C2M_G_TAPS] == kw * kw && g[C2M_G_CIN] > 0 && g[C2M_G_CIN] * kw * kw * 16 <= 48 * 1024) {
            if (kw == 3) return launch_thin_rows<3>(p, ns, (int)g[C2M_G_NTG], (int)g[C2M_G_CIN], g[C2M_G_SQUARE_KW] < 0, s);
            return launch_thin_rows<7>(p, ns, (int)g[C2M_G_NTG], (int)g[C2M_G_CIN], g[C2M_G_SQUARE_KW] < 0, s);
        }
        if (ns == 1) return launch_thin_fwd<1>(p, s);
        if (ns == 2) return launch_thin_fwd<2>(p, s);
        return launch_thin_fwd<4>(p, s);
    }
    const bool bf16 = g[C2M_G_PRECISION] == 1;                          // operand precision: 0 fp32 (exact), 1 bf16 (fp32 accumulate)
    if (bf16 ? !p.xh : (p.xh || p.yh)) return (int)hipErrorInvalidValue;   // bf16 kernels gather bf16 X; fp32 kernels are fp32 only
    if (p.M <= 32)      return launch_igemm<32, 256, 1, 4>(p, ns, splits, s, bf16);
    else if (p.M <= 64) return launch_igemm<64, 128, 2, 2>(p, ns, splits, s, bf16);
    else                return launch_igemm<128, 128, 2, 2>(p, ns, splits, s, bf16);
}

C2M_API int c2m_splitk_reduce(const float* slab, void* out, const float* bias, long total, int splits,
                              long chan_stride, int M, int act, float slope, int dt, void* stream) {
    C2M_ENTER();
    if (total <= 0) return 0;
    const uintptr_t omask = dt == C2M_BF16 ? 7 : 15;
    const bool vec = (total & 3) == 0 && (!bias || (chan_stride & 3) == 0) && (((uintptr_t)slab) & 15) == 0 &&
                     (((uintptr_t)out) & omask) == 0;
    const dim3 grid(c2m_grid(vec ? total / 4 : total, 256));
    hipStream_t s = (hipStream_t)stream;
    C2M_DISPATCH_DT(dt,
        if (vec) {
            if (total / 4 < (1L << 31))
                hipLaunchKernelGGL((splitk_reduce_vec_kernel<unsigned, T>), grid, dim3(256), 0, s, reinterpret_cast<const float4*>(slab),
                                   (T*)out, bias, total / 4, splits, bias ? chan_stride / 4 : 1, M, act, slope);
            else
                hipLaunchKernelGGL((splitk_reduce_vec_kernel<long, T>), grid, dim3(256), 0, s, reinterpret_cast<const float4*>(slab),
                                   (T*)out, bias, total / 4, splits, bias ? chan_stride / 4 : 1, M, act, slope);
        } else {
            hipLaunchKernelGGL(splitk_reduce_kernel<T>, grid, dim3(256), 0, s, slab, (T*)out, bias, total, splits, chan_stride, M,
                               act, slope);
        });
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradP {
    const float* dY;     // [N][M][pix_per_image] contiguous
    const float* X;
    float* slab;         // [S][M][J]
    const int4* jtab;    // per 16-row group (= NS taps x CK channels, same format as the igemm K-step table):
                         // {chan_off, nvalid_chan (-2: ones row, 0: zero rows), 0, 0} + NS x {dt,dy,dx,valid}
    int M, J;            // J = padded (tap, channel) rows incl. the ones-row group
    int Npix, To, Ho, Wo, Ti, Hi, Wi, st, sh, sw;
    int in_sc;
    long in_sn, in_st, in_sh;
    long dy_sn, dy_sc;   // dY strides (pix stride 1)
    int pix_per_split;   // multiple of 64
    int reflect, is3d;
    unsigned x_bytes, dy_bytes;
};

template <int BM, int BN, int WGM, int WGN, int NS, bool BF = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradP p) {
    constexpr int CK = 16 / NS;
    constexpr int BK = 64;                 // pixels per K-step (one wave-width: coalesced along pix)
    constexpr int LDS_S = BK + 1;          // odd stride: conflict-free fragment reads
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int AROWS = BM / 4, BROWSW = BN / 4;   // rows per wave
    constexpr int BGROUPS = BROWSW / 16;              // 16-row (one tap) groups per wave
    static_assert(BROWSW % 16 == 0, "wgrad rows per wave must be whole tap groups");
    // BF: bf16 operands (dY and X rounded RNE while staged), v_mfma_f32_32x32x16_bf16 with 16 pixels per instruction;
    // LDS rows of 64 + 8 bf16 (144 B: conflict-free 16-B fragment reads)
    constexpr int LDH = BK + 8;
    __shared__ float sA[BF ? 1 : BM][BF ? 1 : LDS_S];
    __shared__ float sB[BF ? 1 : BN][BF ? 1 : LDS_S];
    __shared__ __attribute__((aligned(16))) __bf16 hA[BF ? BM : 1][BF ? LDH : 8];
    __shared__ __attribute__((aligned(16))) __bf16 hB[BF ? BN : 1][BF ? LDH : 8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // XCD-aware order (common.h): all tiles of a pixel split read the same pixels -> one XCD, back to back
    const C2mBlock blk = c2m_xcd_block((unsigned)p.J / BN, (unsigned)(p.M + BM - 1) / BM, 0);
    const int m0 = blk.y * BM, j0 = blk.x * BN;
    const int split = blk.z;
    const int pbeg = split * p.pix_per_split;
    int pend = pbeg + p.pix_per_split; pend = pend < p.Npix ? pend : p.Npix;
    const int in_st = (int)p.in_st, in_sh = (int)p.in_sh;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[AROWS], rb[BROWSW];
    // BF: dY and X are bf16 tensors in HBM (the bf16 data path; the host casts fp32 operands once): 2-byte gathers, widened
    // to fp32 registers by a shift (exact), rounded back (exactly) when they are staged into the bf16 LDS image
    constexpr int XES = BF ? 2 : 4;
    auto ldv = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned vo, int soff_elems) -> float {
        if constexpr (BF) return __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, soff_elems * 2, 0) << 16);
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, soff_elems * 4, 0));
    };
    // raw buffer descriptors: per-lane byte offset = this lane's pixel (out of range for dead lanes -> reads 0),
    // scalar offset = row / channel offset (SALU): no per-element VALU
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dY), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    auto issue_loads = [&](int pk) {
        const int pix = pk + lane;
        const bool live = pix < pend;
        int n, ot, oy, ox;
        decompose_pix(live ? pix : pend - 1, p, n, ot, oy, ox);
        const int sp = (ot * p.Ho + oy) * p.Wo + ox;
        const unsigned yvo = live ? (unsigned)(n * (int)p.dy_sn + sp) * (unsigned)XES : C2M_OOB;
        const unsigned ximg = (unsigned)(n * (int)p.in_sn) * (unsigned)XES;
        const int ots = ot * p.st, oys = oy * p.sh, oxs = ox * p.sw;
#pragma unroll
        for (int s = 0; s < AROWS; ++s) {
            const int row = m0 + wave * AROWS + s;            // rows >= M give columns that are never stored
            ra[s] = ldv(yrsrc, yvo, row * (int)p.dy_sc);
        }
#pragma unroll
        for (int gq = 0; gq < BGROUPS; ++gq) {
            const int grp = (j0 + wave * BROWSW) / 16 + gq;     // < J/16 by construction of the grid
            const int4* __restrict__ jd = p.jtab + (long)grp * (1 + NS);
            const int4 hdr = jd[0];
            unsigned vo[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int so = spatial_off(jd[1 + q], ots, oys, oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
                vo[q] = (live && so >= 0) ? ximg + (unsigned)so * (unsigned)XES : C2M_OOB;
            }
            if (hdr.y == -2) {                                   // ones row (bias gradient); wave-uniform branch
#pragma unroll
                for (int s = 0; s < 16; ++s) rb[gq * 16 + s] = (s == 0) ? 1.f : 0.f;
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int slot = s / CK, cc = s % CK;         // compile-time
                    // channels >= nvalid produce columns the reduction never reads
                    rb[gq * 16 + s] = ldv(xrsrc, vo[slot], hdr.x + cc * p.in_sc);
                }
            }
        }
    };
    // The same loads in four parts (128-row tile): each part is issued between the MFMAs of one k-group of the previous step
    // (no branch: the ones group loads through an out-of-range offset and is patched with a select).
    struct PixCtx { unsigned yvo, ximg; int ots, oys, oxs; bool live; };
    auto load_prep = [&](int pk) {
        const int pix = pk + lane;
        PixCtx c;
        c.live = pix < pend;
        int n, ot, oy, ox;
        decompose_pix(c.live ? pix : pend - 1, p, n, ot, oy, ox);
        const int sp = (ot * p.Ho + oy) * p.Wo + ox;
        c.yvo = c.live ? (unsigned)(n * (int)p.dy_sn + sp) * (unsigned)XES : C2M_OOB;
        c.ximg = (unsigned)(n * (int)p.in_sn) * (unsigned)XES;
        c.ots = ot * p.st; c.oys = oy * p.sh; c.oxs = ox * p.sw;
        return c;
    };
    auto load_a_part = [&](const PixCtx& c, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s) {
            const int row = m0 + wave * AROWS + s;
            ra[s] = ldv(yrsrc, c.yvo, row * (int)p.dy_sc);
        }
    };
    auto load_b_group = [&](const PixCtx& c, int gq) {
        const int grp = (j0 + wave * BROWSW) / 16 + gq;
        const int4* __restrict__ jd = p.jtab + (long)grp * (1 + NS);
        const int4 hdr = jd[0];
        const bool ones = hdr.y == -2;
        unsigned vo[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const int so = spatial_off(jd[1 + q], c.ots, c.oys, c.oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
            vo[q] = (c.live && so >= 0 && !ones) ? c.ximg + (unsigned)so * (unsigned)XES : C2M_OOB;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int slot = s / CK, cc = s % CK;
            const float v = ldv(xrsrc, vo[slot], hdr.x + cc * p.in_sc);       // ones group: every offset is out of range -> 0
            rb[gq * 16 + s] = (s == 0 && ones) ? 1.f : v;
        }
    };
    // software pipeline (64- and 32-row tiles; +8..15 % measured): the gathers of K-step i+1 are issued in front of the
    // MFMAs of step i and are in flight while they run.  For the 128x128 tile that form is 4 % slower than loading at the
    // top of the step (64 load issues in front of the MFMAs); it interleaves the loads with the MFMAs instead (IL).
    // fp32: the loads of step i+1 (and the fragment reads of k-group g+1) are interleaved with the MFMAs of step i
    // (k-group g) below: +3.5...6 % on the 128-row tile over loading at the top of the step, +6...12 % on the 64- and
    // 32-row tiles over the PIPE form
    constexpr bool IL = !BF && BGROUPS == 2;
    constexpr bool PIPE = BM < 128 && !IL;
    issue_loads(pbeg);
    for (int pk = pbeg; pk < pend; pk += BK) {
        if (!PIPE && !IL && pk > pbeg) issue_loads(pk);
        __syncthreads();   // previous K-step's fragment reads are done
        if constexpr (BF) {
#pragma unroll
            for (int s = 0; s < AROWS; ++s) hA[wave * AROWS + s][lane] = (__bf16)ra[s];
#pragma unroll
            for (int s = 0; s < BROWSW; ++s) hB[wave * BROWSW + s][lane] = (__bf16)rb[s];
        } else {
#pragma unroll
            for (int s = 0; s < AROWS; ++s) sA[wave * AROWS + s][lane] = ra[s];
#pragma unroll
            for (int s = 0; s < BROWSW; ++s) sB[wave * BROWSW + s][lane] = rb[s];
        }
        __syncthreads();
        if (PIPE && pk + BK < pend) issue_loads(pk + BK);
        if constexpr (BF) {
#pragma unroll
            for (int kq = 0; kq < BK / 16; ++kq) {
                const int kc = kq * 16 + 8 * (lane >> 5);
                bf16x8 a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&hA[wm * TM + i * 32 + (lane & 31)][kc]);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&hB[wn * TN + j * 32 + (lane & 31)][kc]);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else
        if constexpr (IL) {
            // the fragment reads of k-group g + 1 and a part of the NEXT step's loads ride in the shadow of group g's MFMAs
            // (past the split's end every lane is out of range: zeros that are never stored)
            float a[2][8][MI], b[2][8][NI];
            auto read_group = [&](int kg, float (&fa)[8][MI], float (&fb)[8][NI]) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int kcol = (kg * 8 + kk) * 2 + (lane >> 5);
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[kk][i] = sA[wm * TM + i * 32 + (lane & 31)][kcol];
#pragma unroll
                    for (int j = 0; j < NI; ++j) fb[kk][j] = sB[wn * TN + j * 32 + (lane & 31)][kcol];
                }
            };
            read_group(0, a[0], b[0]);
            PixCtx nctx;
#pragma unroll
            for (int kg = 0; kg < BK / 16; ++kg) {
                __builtin_amdgcn_sched_barrier(0);
                if (kg + 1 < BK / 16) read_group(kg + 1, a[(kg + 1) & 1], b[(kg + 1) & 1]);
                if constexpr (AROWS == 32) {
                    if (kg == 0) { nctx = load_prep(pk + BK); load_a_part(nctx, 0, 16); }
                    else if (kg == 1) load_a_part(nctx, 16, 32);
                    else load_b_group(nctx, kg - 2);
                } else {                                   // 64- / 32-row tiles: all of dY with the first group
                    if (kg == 0) { nctx = load_prep(pk + BK); load_a_part(nctx, 0, AROWS); }
                    else if (kg <= 2) load_b_group(nctx, kg - 1);
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg & 1][kk][i], b[kg & 1][kk][j], acc[i][j], 0, 0, 0);
                constexpr int MPG = 8 * MI * NI;           // MFMAs per k-group: 32 / 16 / 8
#pragma unroll
                for (int g = 0; g < 16; ++g) {             // per slot: MFMAs, a fragment read pair, a load (+ its addresses)
                    __builtin_amdgcn_sched_group_barrier(0x008, MPG >= 32 ? 2 : 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, MPG >= 16 ? 2 : 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, MPG >= 16 ? 1 : 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else
        // fragment reads in groups of 8 k-pairs issued ahead of their MFMAs (see the igemm kernel)
#pragma unroll
        for (int kg = 0; kg < BK / 16; ++kg) {
            float a[8][MI], b[8][NI];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int kcol = (kg * 8 + kk) * 2 + (lane >> 5);
#pragma unroll
                for (int i = 0; i < MI; ++i) a[kk][i] = sA[wm * TM + i * 32 + (lane & 31)][kcol];
#pragma unroll
                for (int j = 0; j < NI; ++j) b[kk][j] = sB[wn * TN + j * 32 + (lane & 31)][kcol];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
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

// ---- bf16 data path: weight gradient with 16-byte loads (round 3) -------------------------------------------------------
// On bf16 NCHW tensors the pixel axis -- the K dimension of this GEMM -- is the contiguous one, so a lane can fetch EIGHT
// consecutive pixels of a row with one 16-byte load and put them into the [row][pixel] LDS image with one ds_write_b128: no
// transposition, no conversion.  conv_wgrad_kernel<BF> (lane = pixel) issues 64 two-byte loads + 64 ds_write_b16 per lane for
// the 16 MFMAs of a 64-pixel step and is bound by exactly that; here the same step costs 8 loads + 8 LDS writes.
// A wave-instruction covers 8 rows x 8 pixel groups (lane = rsub * 8 + grp).  X rows are (tap, channel) pairs read at the
// tap's offset: with unit x stride the 8 input pixels of a group are contiguous, at a 2-byte-granular address (unaligned
// 16-byte buffer loads are exact and full speed on gfx950: tools/micro/unaligned_b128.hip).  Only groups at a row end
// reach into the padding (taps with dx = -1 / +1): those load the aligned neighbour group and shift by one element in
// registers (v_alignbit_b32), the vacated element being 0 (zeros) or the mirrored pixel (reflect).  Rows / frames outside
// the image are whole-group zeros or mirrored by address.  Eligibility (host): sw == 1, Wi == Wo, Wo % 8 == 0, |dx| <= 1.
// Same row order, slab layout and reduction as conv_wgrad_kernel.
// SW = 2 (round 3: the 4x4 stride-2 layers of the discriminators and down blocks, a third of the bf16 weight-gradient time on
// the lane-per-pixel kernel): the 8 input pixels of a group are every second element of a 16-element run starting at
// 2*ox + dx -- two 16-byte loads and four v_perm_b32 (even halves; the odd halves when the run would start at -1 and is loaded
// from 0 instead).  dx = -1 .. 2 (4 taps, pad 1): only the first element of a left-end group (pixel -1) and the last of a
// right-end group (pixel Wi) fall into the padding.  Eligibility: sw == 2, Wi == 2*Wo, Wo % 8 == 0, -1 <= dx <= 2.
template <int BM, int BN, int WGM, int WGN, int SW = 1>
__global__ __launch_bounds__(256) void conv_wgrad_wide_bf16_kernel(const WgradP p, const int ns) {
    constexpr int BK = 64, LDH = BK + 8;
    constexpr int TM = BM / WGM, TN = BN / WGN, MI = TM / 32, NI = TN / 32;
    constexpr int AROWS = BM / 4, BROWSW = BN / 4, AP = AROWS / 8, BP = BROWSW / 8;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) __bf16 hA[BM][LDH];
    __shared__ __attribute__((aligned(16))) __bf16 hB[BN][LDH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const C2mBlock blk = c2m_xcd_block((unsigned)p.J / BN, (unsigned)(p.M + BM - 1) / BM, 0);
    const int m0 = blk.y * BM, j0 = blk.x * BN;
    const int split = blk.z;
    const int pbeg = split * p.pix_per_split;
    int pend = pbeg + p.pix_per_split; pend = pend < p.Npix ? pend : p.Npix;
    const int grp = lane & 7, rsub = lane >> 3;
    const int ck = 16 / ns;

    // per-lane row constants (fixed over the K loop)
    unsigned arow_off[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) arow_off[i] = (unsigned)((m0 + wave * AROWS + 8 * i + rsub) * (int)p.dy_sc) * 2u;
    unsigned b_off[BP];                                      // channel offset of the row (bytes)
    int b_dt[BP], b_dy[BP], b_dx[BP];
    bool b_on[BP], b_ones[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int j = j0 + wave * BROWSW + 8 * i + rsub;     // < J by construction of the grid
        const int4* __restrict__ jd = p.jtab + (long)(j >> 4) * (1 + ns);
        const int4 hdr = jd[0];
        const int s16 = j & 15, slot = s16 / ck, cc = s16 - slot * ck;
        const int4 tp = jd[1 + slot];
        b_ones[i] = hdr.y == -2 && s16 == 0;
        b_on[i] = hdr.y > 0 && tp.w != 0;                    // a real (tap, channel) row; channels >= nvalid are never read back
        b_off[i] = (unsigned)((hdr.x + cc * p.in_sc)) * 2u;
        b_dt[i] = tp.x; b_dy[i] = tp.y; b_dx[i] = tp.z;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dY), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    u32x4 ra[AP], rb[BP], rb2[SW == 2 ? BP : 1];
    int redge[BP];                                           // -1: the group starts one pixel left of the row, +1: ends one past it
    bool rtail[SW == 2 ? BP : 1];                            // SW = 2: the 16th element of the run lies past the row end
    auto issue = [&](int pk) {
        const int pix = pk + 8 * grp;
        const bool live = pix < pend;
        int n, ot, oy, ox;
        decompose_pix(live ? pix : pend - 8, p, n, ot, oy, ox);
        const unsigned yvo = (unsigned)(n * (int)p.dy_sn + (ot * p.Ho + oy) * p.Wo + ox) * 2u;
#pragma unroll
        for (int i = 0; i < AP; ++i)
            ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, live ? yvo + arow_off[i] : C2M_OOB, 0, 0));
        const unsigned ximg = (unsigned)(n * (int)p.in_sn) * 2u;
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            int it = p.is3d ? ot * p.st + b_dt[i] : 0, iy = oy * p.sh + b_dy[i];
            const int ix0 = ox * SW + b_dx[i];
            bool ok = live && b_on[i];
            if (p.reflect) {
                if (p.is3d) { it = it < 0 ? -it : it; it = it >= p.Ti ? 2 * p.Ti - 2 - it : it; }
                iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            }
            ok = ok && (unsigned)iy < (unsigned)p.Hi && (!p.is3d || (unsigned)it < (unsigned)p.Ti);
            if constexpr (SW == 1) {
                const int e = ix0 < 0 ? -1 : (ix0 + 8 > p.Wi ? 1 : 0);
                redge[i] = ok ? e : 0;
                const unsigned vo = ximg + b_off[i] + (unsigned)(it * (int)p.in_st + iy * (int)p.in_sh + (ix0 - e)) * 2u;
                rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? vo : C2M_OOB, 0, 0));
            } else {
                // 16 elements from ix0 (from 0 when ix0 == -1: the wanted pixels are then the ODD elements 1, 3, .. 13 behind a
                // pad element).  A right-end run reaches up to two elements past its row: in-bounds memory (the next row) except
                // at the very end of the tensor, where the range check returns zeros -- that element is replaced in stash().
                const int e = ix0 < 0 ? -1 : (ix0 + 14 >= p.Wi ? 1 : 0);
                redge[i] = ok ? e : 0;
                // a run whose 16th element is past the row end (dx = 1, 2 in the last group of a row) loads its second half
                // one element EARLIER (elements 7 .. 14, wanted ones in the odd halves): at the end of the tensor the dword
                // (x[Wi-1], x[Wi]) straddles the buffer's range and would come back as zero, x[Wi-1] included
                const bool tail = e >= 0 && ix0 + 15 >= p.Wi;
                rtail[i] = tail;
                const unsigned vo = ximg + b_off[i] + (unsigned)(it * (int)p.in_st + iy * (int)p.in_sh + (e < 0 ? 0 : ix0)) * 2u;
                rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? vo : C2M_OOB, 0, 0));
                rb2[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? vo + (tail ? 14u : 16u) : C2M_OOB, 0, 0));
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<u32x4*>(&hA[wave * AROWS + 8 * i + rsub][8 * grp]) = ra[i];
        if constexpr (SW == 2) {
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const u32x4 lo = rb[i], hi = rb2[i];
                // even elements (0, 2, .. 14) of the 16-element run / odd elements (1, 3, .. 15) for a left-end group
                const unsigned hsel = rtail[i] ? 0x07060302u : 0x05040100u;      // shifted second half: the odd halves
                const u32x4 ev = {__builtin_amdgcn_perm(lo.y, lo.x, 0x05040100u), __builtin_amdgcn_perm(lo.w, lo.z, 0x05040100u),
                                  __builtin_amdgcn_perm(hi.y, hi.x, hsel), __builtin_amdgcn_perm(hi.w, hi.z, hsel)};
                const u32x4 od = {__builtin_amdgcn_perm(lo.y, lo.x, 0x07060302u), __builtin_amdgcn_perm(lo.w, lo.z, 0x07060302u),
                                  __builtin_amdgcn_perm(hi.y, hi.x, 0x07060302u), __builtin_amdgcn_perm(hi.w, hi.z, 0x07060302u)};
                // left end: (pad, x1, x3, .. x13): pad = x1 (reflect) or 0
                const unsigned padl = p.reflect ? od.x & 0xffffu : 0u;
                const u32x4 L = {(od.x << 16) | padl, __builtin_amdgcn_alignbit(od.y, od.x, 16), __builtin_amdgcn_alignbit(od.z, od.y, 16),
                                 __builtin_amdgcn_alignbit(od.w, od.z, 16)};
                // right end: the last element (pixel Wi) -> x[Wi-2] = the element before it (reflect) or 0
                const u32x4 R = {ev.x, ev.y, ev.z, (ev.w & 0xffffu) | (p.reflect ? ev.w << 16 : 0u)};
                rb[i] = redge[i] < 0 ? L : (redge[i] > 0 ? R : ev);
            }
        }
        bool any_edge = false;
#pragma unroll
        for (int i = 0; i < BP; ++i) any_edge = any_edge || (SW == 1 && redge[i] != 0);
        if (__any(any_edge)) {                               // wave-uniform: most K-steps of a wide map hold no row end
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const u32x4 v = rb[i];
                // left end: (pad, x0 .. x6) from the aligned load x0 .. x7; pad = x1 (reflect) or 0
                const unsigned padl = p.reflect ? v.x >> 16 : 0u;
                const u32x4 L = {(v.x << 16) | padl, __builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                 __builtin_amdgcn_alignbit(v.w, v.z, 16)};
                // right end: (x[W-7] .. x[W-1], pad) from x[W-8] .. x[W-1]; pad = x[W-2] (reflect) or 0
                const unsigned padr = p.reflect ? v.w << 16 : 0u;
                const u32x4 R = {__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16),
                                 __builtin_amdgcn_alignbit(v.w, v.z, 16), (v.w >> 16) | padr};
                rb[i] = redge[i] < 0 ? L : (redge[i] > 0 ? R : v);
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};       // bf16 1.0 x 8 (bias gradient row)
            *reinterpret_cast<u32x4*>(&hB[wave * BROWSW + 8 * i + rsub][8 * grp]) = b_ones[i] ? ones : rb[i];
        }
    };

    issue(pbeg);
    for (int pk = pbeg; pk < pend; pk += BK) {
        __syncthreads();                                     // the previous step's fragment reads are done
        stash();
        __syncthreads();
        if (pk + BK < pend) issue(pk + BK);                  // in flight during the MFMAs below
#pragma unroll
        for (int kq = 0; kq < BK / 16; ++kq) {
            const int kc = kq * 16 + 8 * (lane >> 5);
            bf16x8 a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(&hA[wm * TM + i * 32 + (lane & 31)][kc]);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8*>(&hB[wn * TN + j * 32 + (lane & 31)][kc]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
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

// dW[m][c*taps + tap] = sum_s slab[s][m][col(c, tap)] with the (chunk, tap group, slot, channel) row order;
// db[m] = sum_s slab[s][m][ones_col]
// One workgroup = 64 consecutive outputs in the slab's own (m, column) order x G groups of splits (G = 4, or 16 from 64
// splits on; wave g sums the splits s = g, g + G, ...; its 64 lanes read 256 contiguous bytes per split), the four partial sums are combined in a fixed order
// through LDS.  A thread per output with the whole S loop left a 32 x 304 layer with 256 splits on 38 workgroups of
// 256 dependent loads each (177 us for 12 MB).  The single write per output is the scattered one.
template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW,
                                                              float* __restrict__ db, int M, int J, int Cin, int taps, int NS,
                                                              int ntg, int ngroups, int S) {
    __shared__ float part[G][64];
    const int CK = 16 / NS;
    const int used = (ngroups + 1) * 16;                  // real groups + the ones group; pad columns are skipped
    const long total = (long)M * used;
    const long per = (long)M * J;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {
        const long i = base + lane;
        long dst = -1;                                     // -1: padding / out of range, -2: bias gradient
        long src = 0;
        if (i < total) {
            // (outputs < 2^31 in practice: a 32-bit division instead of a 64-bit one per output)
            const int m = total < (1L << 31) ? (int)((unsigned)i / (unsigned)used) : (int)(i / used);
            const int col = (int)(i - (long)m * used);
            const int g = col >> 4, within = col & 15;
            if (g == ngroups) {
                if (within == 0) dst = -2 - m;
            } else {
                const int c = (g / ntg) * CK + within % CK, tap = (g % ntg) * NS + within / CK;
                if (c < Cin && tap < taps) dst = ((long)m * Cin + c) * taps + tap;
            }
            src = (long)m * J + col;
        }
        float acc = 0.f;
        if (dst != -1) {
            int sp = grp;
            for (; sp + 3 * G < S; sp += 4 * G) {          // four independent loads in flight
                const float v0 = slab[(long)sp * per + src], v1 = slab[(long)(sp + G) * per + src];
                const float v2 = slab[(long)(sp + 2 * G) * per + src], v3 = slab[(long)(sp + 3 * G) * per + src];
                acc += v0; acc += v1; acc += v2; acc += v3;
            }
            for (; sp < S; sp += G) acc += slab[(long)sp * per + src];
        }
        __syncthreads();                                   // the previous round's reads of `part` are done
        part[grp][lane] = acc;
        __syncthreads();
        if (grp == 0 && dst != -1) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < G; q += 4) v += (part[q][lane] + part[q + 1][lane]) + (part[q + 2][lane] + part[q + 3][lane]);
            if (dst <= -2) { if (db) db[-2 - dst] = v; }
            else dW[dst] = v;
        }
    }
}

// Few splits, many outputs (the deep layers): one workgroup per (output row m, channel chunk).  The chunk's ntg * 16 slab
// columns are summed over the splits with coalesced reads into LDS, then written as ONE contiguous run of CK * taps floats of
// dW (the per-output form above scatters its 4-byte writes with a stride of `taps` floats: 16x write amplification, 85 us
// on the 512 x 8192 layers).  blockIdx.x == nch handles the bias column.
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ slab, float* __restrict__ dW,
                                                                float* __restrict__ db, int M, int J, int Cin, int taps,
                                                                int lns, int ntg, int ngroups, int S) {
    extern __shared__ float val[];                           // ntg * 16 column sums of this (m, chunk)
    const int NS = 1 << lns, lck = 4 - lns, CK = 16 >> lns;
    const int m = blockIdx.y, chunk = blockIdx.x;
    const int nch = ngroups / ntg;
    const long per = (long)M * J;
    const float* __restrict__ row = slab + (long)m * J;
    if (chunk == nch) {                                       // bias gradient: the ones group's first column
        if (db && threadIdx.x == 0) {
            float acc = 0.f;
            for (int sp = 0; sp < S; ++sp) acc += row[(long)sp * per + ngroups * 16];
            db[m] = acc;
        }
        return;
    }
    const int ncols = ntg * 16, col0 = chunk * ncols;
    for (int c = threadIdx.x; c < ncols; c += 256) {
        float acc = 0.f;
        int sp = 0;
        for (; sp + 3 < S; sp += 4) {
            const float v0 = row[(long)sp * per + col0 + c], v1 = row[(long)(sp + 1) * per + col0 + c];
            const float v2 = row[(long)(sp + 2) * per + col0 + c], v3 = row[(long)(sp + 3) * per + col0 + c];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; sp < S; ++sp) acc += row[(long)sp * per + col0 + c];
        val[c] = acc;
    }
    __syncthreads();
    const int cbase = chunk * CK;
    int nc = Cin - cbase; nc = nc < CK ? nc : CK;             // real channels of this chunk
    float* __restrict__ dst = dW + ((long)m * Cin + cbase) * taps;
    for (int o = threadIdx.x; o < nc * taps; o += 256) {
        const int cl = o / taps, tap = o - cl * taps;
        dst[o] = val[((tap >> lns) << 4) + ((tap & (NS - 1)) << lck) + cl];
    }
}

static int launch_wgrad_reduce(long total, hipStream_t s, const float* slab, float* dW, float* db, int M, int J, int Cin,
                               int taps, int NS, int ntg, int ngroups, int S) {
    const int lns = NS == 1 ? 0 : (NS == 2 ? 1 : 2);
    if (S < 64 && ntg > 0 && ngroups % ntg == 0 && (long)ntg * 16 * 4 <= 48 * 1024 && M <= 65535 && total >= (1L << 16)) {
        dim3 grid(ngroups / ntg + 1, M);
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, grid, dim3(256), (size_t)ntg * 16 * sizeof(float), s, slab, dW, db, M, J,
                           Cin, taps, lns, ntg, ngroups, S);
        return (int)hipGetLastError();
    }
    if (S >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(c2m_grid(total, 64)), dim3(1024), 0, s, slab, dW, db, M, J, Cin, taps,
                           NS, ntg, ngroups, S);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(c2m_grid(total, 64)), dim3(256), 0, s, slab, dW, db, M, J, Cin, taps,
                           NS, ntg, ngroups, S);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ thin layers
// Convolutions with <= 4 output channels (flow / occlusion heads, the RGB output conv) waste >= 87 % of a 32-row MFMA
// tile; they run on the vector ALU instead: one pixel per lane, the same K-step table and buffer-load gather, weights
// through scalar loads (the k index is wave-uniform).  Forward only needs this; the data gradient of these layers has
// M = Cin >= 32 and stays on the MFMA path.
template <int MT, int NS>
__global__ __launch_bounds__(256) void conv_thin_fwd_kernel(const ConvP p) {
    constexpr int CK = 16 / NS;
    const int in_st = (int)p.in_st, in_sh = (int)p.in_sh;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < p.Npix; pix += gridDim.x * 256) {
        int n, ot, oy, ox;
        decompose_pix(pix, p, n, ot, oy, ox);
        const unsigned img_byte = (unsigned)(n * (int)p.in_sn) * 4u;
        const int ots = ot * p.st, oys = oy * p.sh, oxs = ox * p.sw;
        float acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = 0.f;
        for (int kt = 0; kt < p.nk; ++kt) {
            const int4* __restrict__ kd = p.ktab + (long)kt * (1 + NS);
            const int4 hdr = kd[0];
            unsigned vo[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int so = spatial_off(kd[1 + q], ots, oys, oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
                vo[q] = so >= 0 ? img_byte + (unsigned)so * 4u : C2M_OOB;
            }
            float xv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e)
                xv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    xrsrc, vo[e / CK], (hdr.x + (e % CK) * p.in_sc) * 4, 0));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float* __restrict__ wr = p.A + (long)m * p.lda + kt * 16;      // uniform -> scalar loads
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m] = fmaf(xv[e], wr[e], acc[m]);
            }
        }
        float* __restrict__ yb = p.Y + p.out_off + (long)n * p.out_sn + (long)ot * p.out_st + (long)oy * p.out_sh +
                                 (long)ox * p.out_sw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < p.M) {
                float v = acc[m];
                if (p.bias) v += p.bias[m];
                yb[(long)m * p.out_sc] = c2m_act(v, p.act, p.slope);
            }
        }
    }
}

// Row-blocked form for 2-D stride-1 square kernels (the 7x7 RGB head, the 3x3 flow / occlusion heads, the data gradient
// of the first VGG conv): each lane computes PX = 4 horizontally adjacent outputs, so one gathered input row segment of
// PX + KW - 1 values feeds PX * KW taps -- 2.8x (7x7) / 2x (3x3) fewer loads per FMA than the per-pixel kernel above,
// which is bound by load issue.  Tap geometry comes from the same K-step table (row-major taps, dx ascending for the
// forward pass, descending for a data gradient), weights from the same packed matrix through scalar loads.
template <int MT, int KW>
__global__ __launch_bounds__(256) void conv_thin_rows_kernel(const ConvP p, int lns, int ntg, int Cin, int xdesc) {
    constexpr int PX = 4, NV = PX + KW - 1;
    extern __shared__ float4 sW[];                        // [c][i][j] -> weights of the (<= 4) output rows
    const int NS = 1 << lns, CK = 16 >> lns, lck = 4 - lns;
    auto tap_entry = [&](int t) { return p.ktab[(t >> lns) * (1 + NS) + 1 + (t & (NS - 1))]; };   // chunk 0's taps
    for (int e = threadIdx.x; e < Cin * KW * KW; e += 256) {
        const int c = e / (KW * KW), t = e - c * (KW * KW);
        const int k = (((c >> lck) * ntg + (t >> lns)) << 4) + ((t & (NS - 1)) << lck) + (c & (CK - 1));
        float w[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MT; ++m) w[m] = p.A[(long)m * p.lda + k];
        sW[e] = make_float4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const int groups_x = p.Wo / PX;
    const int total = p.Npix / PX;
    const int dx_first = tap_entry(0).z, dx_last = tap_entry(KW - 1).z;
    const int lo = dx_first < dx_last ? dx_first : dx_last;
    // Symmetric tap sets on inputs as wide as the output (every layer this kernel sees): each lane loads its own four input
    // pixels of a row with one 16-byte load and takes the HL halo pixels on each side from the neighbouring lanes, from its
    // own quad at a reflected image border, and from memory only as the first / last lane of a wave.  (Gathering the
    // PX + KW - 1 window per lane fetched every pixel up to KW times through requests that were all in flight together.)
    constexpr int HL = (KW - 1) / 2;
    const bool sym = lo == -HL && p.Wi == p.Wo && (p.in_sh % 4) == 0 && (p.in_sc % 4) == 0 && (p.in_sn % 4) == 0 &&
                     (((uintptr_t)p.X) & 15) == 0;
    const int lane = threadIdx.x & 63;
    const int rounds = (total - (int)blockIdx.x * 256 + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);   // uniform per workgroup
    int g = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < rounds; ++it, g += gridDim.x * 256) {
        const bool live = g < total;
        const int gg = live ? g : total - 1;
        const int xg = gg % groups_x; int r = gg / groups_x;
        const int oy = r % p.Ho; const int n = r / p.Ho;
        const int ox0 = xg * PX;
        const unsigned img_byte = (unsigned)(n * (int)p.in_sn) * 4u;
        float acc[MT][PX];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PX; ++q) acc[m][q] = 0.f;
        const bool first = xg == 0, last = xg == groups_x - 1;
        const bool memL = live && !first && lane == 0;
        const bool memR = live && !last && (lane == 63 || g + 1 >= total);
        unsigned cvo[NV];                                  // column byte offsets of this lane's NV inputs (gather form)
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            int ix = ox0 + lo + e;
            bool ok = live;
            if (p.reflect) { ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix; }
            else ok = ok && (unsigned)ix < (unsigned)p.Wi;
            cvo[e] = ok ? (unsigned)ix * 4u : C2M_OOB;
        }
        for (int i = 0; i < KW; ++i) {                    // tap rows (square kernel)
            int iy = oy + tap_entry(i * KW).y;
            bool rok = live;
            if (p.reflect) { iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy; }
            else rok = rok && (unsigned)iy < (unsigned)p.Hi;
            const unsigned rbase = img_byte + (unsigned)(iy * (int)p.in_sh) * 4u;
            unsigned vo[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) vo[e] = (rok && cvo[e] != C2M_OOB) ? rbase + cvo[e] : C2M_OOB;
            const unsigned own_vo = rok ? rbase + (unsigned)ox0 * 4u : C2M_OOB;
            const float4* __restrict__ wrow = sW + i * KW;
            for (int c = 0; c < Cin; ++c) {
                float v[NV];
                if (sym) {
                    const int soff = c * p.in_sc * 4;
                    const f32x4 own = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, own_vo, soff, 0));
                    v[HL + 0] = own.x; v[HL + 1] = own.y; v[HL + 2] = own.z; v[HL + 3] = own.w;
#pragma unroll
                    for (int e = 0; e < HL; ++e) {
                        const float ml = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            xrsrc, (rok && memL) ? own_vo - (unsigned)(HL - e) * 4u : C2M_OOB, soff, 0));
                        const float mr = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            xrsrc, (rok && memR) ? own_vo + (unsigned)(4 + e) * 4u : C2M_OOB, soff, 0));
                        const float up = __shfl_up(v[4 + e], 1, 64);                   // left neighbour's pixel 4 - HL + e
                        const float dn = __shfl_down(v[HL + e], 1, 64);                // right neighbour's pixel e
                        const float bl = p.reflect ? v[HL + HL - e] : 0.f;             // image border: own pixel HL - e
                        const float br = p.reflect ? v[HL + 2 - e] : 0.f;              //               own pixel 2 - e
                        v[e] = memL ? ml : (first ? bl : up);
                        v[HL + 4 + e] = memR ? mr : (last ? br : dn);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < NV; ++e)
                        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, vo[e], c * p.in_sc * 4, 0));
                }
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    const float4 w4 = wrow[c * (KW * KW) + j];          // same address in every lane: LDS broadcast
                    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
                    const int pos = xdesc ? KW - 1 - j : j;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < PX; ++q) acc[m][q] = fmaf(v[q + pos], wv[m], acc[m][q]);
                }
            }
        }
        if (!live) continue;
        float* __restrict__ yb = p.Y + p.out_off + (long)n * p.out_sn + (long)oy * p.out_sh + (long)ox0 * p.out_sw;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < p.M) {
                float o[PX];
#pragma unroll
                for (int q = 0; q < PX; ++q) {
                    float v = acc[m][q];
                    if (p.bias) v += p.bias[m];
                    o[q] = c2m_act(v, p.act, p.slope);
                }
                float* dst = yb + (long)m * p.out_sc;
                if (p.out_sw == 1 && (((uintptr_t)dst) & 15) == 0) {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < PX; ++q) dst[(long)q * p.out_sw] = o[q];
                }
            }
        }
    }
}

template <int KW>
static int launch_thin_rows(const ConvP& p, int ns, int ntg, int Cin, int xdesc, hipStream_t s) {
    // few, fat blocks: every block first copies the layer's weights into LDS (Cin * KW * KW float4)
    long blocks = (p.Npix / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks);
    const int lns = ns == 1 ? 0 : (ns == 2 ? 1 : 2);
    const size_t lds = (size_t)Cin * KW * KW * sizeof(float4);
    switch (p.M) {
        case 1: hipLaunchKernelGGL((conv_thin_rows_kernel<1, KW>), grid, dim3(256), lds, s, p, lns, ntg, Cin, xdesc); break;
        case 2: hipLaunchKernelGGL((conv_thin_rows_kernel<2, KW>), grid, dim3(256), lds, s, p, lns, ntg, Cin, xdesc); break;
        case 3: hipLaunchKernelGGL((conv_thin_rows_kernel<3, KW>), grid, dim3(256), lds, s, p, lns, ntg, Cin, xdesc); break;
        default: hipLaunchKernelGGL((conv_thin_rows_kernel<4, KW>), grid, dim3(256), lds, s, p, lns, ntg, Cin, xdesc); break;
    }
    return (int)hipGetLastError();
}

template <int NS>
static int launch_thin_fwd(const ConvP& p, hipStream_t s) {
    dim3 grid(c2m_grid(p.Npix, 256));
    switch (p.M) {
        case 1: hipLaunchKernelGGL((conv_thin_fwd_kernel<1, NS>), grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_thin_fwd_kernel<2, NS>), grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((conv_thin_fwd_kernel<3, NS>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((conv_thin_fwd_kernel<4, NS>), grid, dim3(256), 0, s, p); break;
    }
    return (int)hipGetLastError();
}

// wgrad for <= 4 output channels: lanes = pixels, each thread keeps GPB*16*MT partial sums in registers over its
// pixels of the split, then a block reduction; slab layout identical to the MFMA wgrad so the same reduction finishes.
template <int MT, int NS, int GPB>
__global__ __launch_bounds__(256) void conv_thin_wgrad_kernel(const WgradP p, int ngroups_total) {
    constexpr int CK = 16 / NS;
    __shared__ float red[4][GPB * 16 * MT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g0 = blockIdx.x * GPB;
    const int split = blockIdx.z;
    const int pbeg = split * p.pix_per_split;
    int pend = pbeg + p.pix_per_split; pend = pend < p.Npix ? pend : p.Npix;
    const int in_st = (int)p.in_st, in_sh = (int)p.in_sh;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    float acc[GPB][16][MT];
#pragma unroll
    for (int g = 0; g < GPB; ++g)
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[g][s][m] = 0.f;
    for (int pix = pbeg + threadIdx.x; pix < pend; pix += 256) {
        int n, ot, oy, ox;
        decompose_pix(pix, p, n, ot, oy, ox);
        const int sp = (ot * p.Ho + oy) * p.Wo + ox;
        float dy[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) dy[m] = m < p.M ? p.dY[(long)n * p.dy_sn + (long)m * p.dy_sc + sp] : 0.f;
        const unsigned ximg = (unsigned)(n * (int)p.in_sn) * 4u;
        const int ots = ot * p.st, oys = oy * p.sh, oxs = ox * p.sw;
#pragma unroll
        for (int g = 0; g < GPB; ++g) {
            const int grp = g0 + g;
            if (grp >= ngroups_total) break;
            const int4* __restrict__ jd = p.jtab + (long)grp * (1 + NS);
            const int4 hdr = jd[0];
            if (hdr.y == -2) {
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[g][0][m] += dy[m];
                continue;
            }
            unsigned vo[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int so = spatial_off(jd[1 + q], ots, oys, oxs, p.Ti, p.Hi, p.Wi, in_st, in_sh, p.reflect, p.is3d);
                vo[q] = so >= 0 ? ximg + (unsigned)so * 4u : C2M_OOB;
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    xrsrc, vo[s / CK], (hdr.x + (s % CK) * p.in_sc) * 4, 0));
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[g][s][m] = fmaf(dy[m], x, acc[g][s][m]);
            }
        }
    }
    // block reduction: wave shuffles, then 4 partials through LDS (fixed order)
#pragma unroll
    for (int g = 0; g < GPB; ++g)
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float v = wave_sum(acc[g][s][m]);
                if (lane == 0) red[wave][(g * 16 + s) * MT + m] = v;
            }
    __syncthreads();
    float* __restrict__ out = p.slab + (long)split * p.M * p.J;
    for (int i = threadIdx.x; i < GPB * 16 * MT; i += 256) {
        const int m = i % MT, gs = i / MT;
        const int col = g0 * 16 + gs;
        if (m < p.M && col < ngroups_total * 16)
            out[(long)m * p.J + col] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

// Row-blocked weight gradient for <= 4 output channels and a 2-D stride-1 square KW x KW kernel (the 7x7 RGB head, the 3x3
// flow / occlusion heads): a workgroup owns CPB input channels and ALL taps, every lane walks over groups of PX = 4
// horizontally adjacent pixels of the split: per tap row one gathered segment of PX + KW - 1 inputs feeds KW taps x PX pixels
// x MT outputs, i.e. 2.8x (7x7) / 2x (3x3) fewer loads per FMA than the per-pixel kernel above, which is bound by load
// issue (13 TF/s on the 7x7 head).  KW*KW*CPB*MT accumulators per lane (147 for the 7x7 head); block reduction and slab
// layout as above, so the same wgrad_reduce_kernel finishes.  Channel block 0 also sums dY for the bias gradient.
template <int MT, int KW, int CPB, int RI>
__global__ __launch_bounds__(256) void conv_thin_wgrad_rows_kernel(const WgradP p, int lns, int ntg, int Cin, int ngroups,
                                                                   int quads_per_split) {
    constexpr int PX = 4, NV = PX + KW - 1, NACC = CPB * RI * KW * MT;
    __shared__ float red[4][NACC + MT];
    const int NS = 1 << lns, lck = 4 - lns, CK = 16 >> lns;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * CPB;
    const int i0 = blockIdx.y * RI;              // first tap row of this workgroup (rows >= KW contribute nothing)
    const int split = blockIdx.z;
    const int groups_x = p.Wo / PX;
    const int total = p.Npix / PX;
    const int qbeg = split * quads_per_split;
    int qend = qbeg + quads_per_split; qend = qend < total ? qend : total;
    auto tap_entry = [&](int t) { return p.jtab[(t >> lns) * (1 + NS) + 1 + (t & (NS - 1))]; };   // chunk 0's taps
    const int dx0 = tap_entry(0).z, dy0 = tap_entry(0).y;        // row-major taps, dy / dx ascending by one
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    float acc[CPB][RI][KW][MT];
    float accb[MT];
#pragma unroll
    for (int c = 0; c < CPB; ++c)
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[c][i][j][m] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) accb[m] = 0.f;
    // Two register sets: the loads of pixel group g + 256 (all KW rows x CPB channels: one latency per group instead of one
    // per row) are in flight while group g is accumulated; the kernel runs one wave per SIMD (147 accumulators for the 7x7
    // head), so nothing else would hide them.  Groups past the split's end load through out-of-range offsets (zeros).
    // Input row segments: every lane loads its own four pixels once (one 16-byte load per row; the per-tap gathers of the
    // first version fetched each pixel up to KW times through separate requests that were all in flight together, and the
    // L2 served every one of them).  The HL halo pixels on each side come from the neighbouring lanes (shuffles in
    // accumulate()), from the pixel's own quad at a reflected image border, and from memory only for the first / last live
    // lane of a wave.
    constexpr int HL = (KW - 1) / 2;
    auto load_group = [&](int g, float (&dy)[MT][PX], float (&v)[RI][CPB][NV], int& fl) {
        const bool live = g < qend;
        const int gg = live ? g : qend - 1;
        const int xg = gg % groups_x; const int r = gg / groups_x;
        const int oy = r % p.Ho; const int n = r / p.Ho;
        const int ox0 = xg * PX;
        const bool first = xg == 0, last = xg == groups_x - 1;
        const bool memL = live && !first && lane == 0;
        const bool memR = live && !last && (lane == 63 || g + 1 >= qend);
        fl = (first ? 2 : 0) | (last ? 4 : 0) | (memL ? 8 : 0) | (memR ? 16 : 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m < p.M && live) {
                const float4 t = *reinterpret_cast<const float4*>(p.dY + (long)n * p.dy_sn + (long)m * p.dy_sc + (long)oy * p.Wo + ox0);
                dy[m][0] = t.x; dy[m][1] = t.y; dy[m][2] = t.z; dy[m][3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < PX; ++q) dy[m][q] = 0.f;
            }
        }
        const unsigned img_byte = (unsigned)(n * (int)p.in_sn) * 4u;
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            int iy = oy + dy0 + i0 + i;
            bool rok = live && i0 + i < KW;
            if (p.reflect) { iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy; }
            else rok = rok && (unsigned)iy < (unsigned)p.Hi;
            const unsigned rbase = img_byte + (unsigned)(iy * (int)p.in_sh + ox0) * 4u;
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
                const bool cok = rok && c0 + c < Cin;
                const int soff = (c0 + c) * p.in_sc * 4;
                const f32x4 own = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, cok ? rbase : C2M_OOB, soff, 0));
                v[i][c][HL + 0] = own.x; v[i][c][HL + 1] = own.y; v[i][c][HL + 2] = own.z; v[i][c][HL + 3] = own.w;
#pragma unroll
                for (int e = 0; e < HL; ++e) {
                    v[i][c][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        xrsrc, (cok && memL) ? rbase - (unsigned)(HL - e) * 4u : C2M_OOB, soff, 0));
                    v[i][c][HL + 4 + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        xrsrc, (cok && memR) ? rbase + (unsigned)(4 + e) * 4u : C2M_OOB, soff, 0));
                }
            }
        }
    };
    auto accumulate = [&](const float (&dy)[MT][PX], float (&v)[RI][CPB][NV], const int fl) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < PX; ++q) accb[m] += dy[m][q];
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int c = 0; c < CPB; ++c) {
#pragma unroll
                for (int e = 0; e < HL; ++e) {
#ifdef C2M_THIN_DPP      // bisect build (tools/concurrency_stress.py): whole-wave DPP shifts instead of ds_bpermute_b32 through the LDS crossbar
                    const int su = __builtin_bit_cast(int, v[i][c][4 + e]), sd = __builtin_bit_cast(int, v[i][c][HL + e]);
                    const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(su, su, 0x138, 0xf, 0xf, false));   // wave_shr:1
                    const float dn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(sd, sd, 0x130, 0xf, 0xf, false));   // wave_shl:1
#else
                    const float up = __shfl_up(v[i][c][4 + e], 1, 64);                 // left neighbour's pixel 4 - HL + e
                    const float dn = __shfl_down(v[i][c][HL + e], 1, 64);              // right neighbour's pixel e
#endif
                    const float bl = p.reflect ? v[i][c][HL + HL - e] : 0.f;           // image border: own pixel HL - e
                    const float br = p.reflect ? v[i][c][HL + 2 - e] : 0.f;            //               own pixel 2 - e
                    v[i][c][e] = (fl & 8) ? v[i][c][e] : ((fl & 2) ? bl : up);
                    v[i][c][HL + 4 + e] = (fl & 16) ? v[i][c][HL + 4 + e] : ((fl & 4) ? br : dn);
                }
#pragma unroll
                for (int j = 0; j < KW; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < PX; ++q) acc[c][i][j][m] = fmaf(dy[m][q], v[i][c][q + j], acc[c][i][j][m]);
            }
    };
    float dyA[MT][PX], dyB[MT][PX];
    float vA[RI][CPB][NV], vB[RI][CPB][NV];
    int flA = 0, flB = 0;
    int g = qbeg + threadIdx.x;
    if (qbeg < qend) {
        load_group(g, dyA, vA, flA);
        // uniform trip count (whole workgroup): the shuffles in accumulate() need every lane of the wave
        const int rounds = (qend - qbeg + 511) / 512;
        for (int it = 0; it < rounds; ++it, g += 512) {
            load_group(g + 256, dyB, vB, flB);
            accumulate(dyA, vA, flA);
            load_group(g + 512, dyA, vA, flA);
            accumulate(dyB, vB, flB);
        }
    }
    // block reduction: wave shuffles, then 4 partials through LDS (fixed order)
#pragma unroll
    for (int c = 0; c < CPB; ++c)
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float v = wave_sum(acc[c][i][j][m]);
                    if (lane == 0) red[wave][((c * RI + i) * KW + j) * MT + m] = v;
                }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const float v = wave_sum(accb[m]);
        if (lane == 0) red[wave][NACC + m] = v;
    }
    __syncthreads();
    float* __restrict__ out = p.slab + (long)split * p.M * p.J;
    for (int e = threadIdx.x; e < NACC + MT; e += 256) {
        const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (e >= NACC) {                                          // bias gradient column (ones group), channel block 0 only
            const int m = e - NACC;
            if (blockIdx.x == 0 && blockIdx.y == 0 && m < p.M) out[(long)m * p.J + ngroups * 16] = v;
            continue;
        }
        const int m = e % MT, j = (e / MT) % KW, i = i0 + (e / (MT * KW)) % RI, c = c0 + e / (MT * KW * RI);
        const int t = i * KW + j;
        if (m >= p.M || c >= Cin || i >= KW) continue;
        const int col = (((c >> lck) * ntg + (t >> lns)) << 4) + ((t & (NS - 1)) << lck) + (c & (CK - 1));
        out[(long)m * p.J + col] = v;
    }
}

#ifndef C2M_WG64_BN
#define C2M_WG64_BN 128
#endif
static void wgrad_tile(int M, int& BM, int& BN) {
    if (M <= 32) { BM = 32; BN = 128; }
    else if (M <= 64) { BM = 64; BN = C2M_WG64_BN; }
    else { BM = 128; BN = 128; }
}

// Number of pixel splits the wgrad launch will use for (M, J, Npix): the caller sizes `slab` = S*M*J floats.
C2M_API int c2m_conv_wgrad_splits(int M, int J, int Npix) {
    int BM, BN;
    wgrad_tile(M, BM, BN);
    const long tiles = (long)c2m_cdiv(M, BM) * c2m_cdiv(J, BN);
#ifdef C2M_WGRAD_OLD_SPLITS
    long S = (1024 + tiles - 1) / tiles;
    const long maxS = (Npix + 2047) / 2048;   // at least 2048 pixels per split
#else
    // One resident round: blocks <= CUs x blocks/CU (LDS-limited: 2 for the 128x128 tile, 3 otherwise), so no block
    // is left to run alone after the others (a 1026-block grid on 512 slots costs a third round for 2 blocks).
    const long slots = 256L * (BM == 128 ? 2 : 3);
    long S = slots / tiles;
    const long maxS = (Npix + 1023) / 1024;   // at least 1024 pixels (16 K-steps) per split
#endif
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return (int)S;
}

// Rows of the slab (a multiple of the tile width) for a conv with `ngroups` 16-row groups (incl. the ones group).
C2M_API int c2m_conv_wgrad_rows(int M, int ngroups) {
    int BM, BN;
    wgrad_tile(M, BM, BN);
    return c2m_cdiv(ngroups * 16, BN) * BN;
}

C2M_API int c2m_conv_wgrad(const void* dY, const void* X, float* slab, float* dW, float* db, const int* jtab,
                           const int64_t* g, void* stream) {
    C2M_ENTER();
    WgradP p;
    p.dY = (const float*)dY; p.X = (const float*)X; p.slab = slab; p.jtab = reinterpret_cast<const int4*>(jtab);
    p.M = (int)g[C2M_G_M]; p.J = (int)g[C2M_G_NK];
    p.Npix = (int)g[C2M_G_NPIX]; p.To = (int)g[C2M_G_TO]; p.Ho = (int)g[C2M_G_HO]; p.Wo = (int)g[C2M_G_WO];
    p.Ti = (int)g[C2M_G_TI]; p.Hi = (int)g[C2M_G_HI]; p.Wi = (int)g[C2M_G_WI];
    p.st = (int)g[C2M_G_ST]; p.sh = (int)g[C2M_G_SH]; p.sw = (int)g[C2M_G_SW];
    p.in_sn = g[C2M_G_IN_SN]; p.in_st = g[C2M_G_IN_ST]; p.in_sh = g[C2M_G_IN_SH];
    p.dy_sn = g[C2M_G_OUT_SN]; p.dy_sc = g[C2M_G_OUT_SC];
    p.reflect = (int)g[C2M_G_REFLECT]; p.is3d = (int)g[C2M_G_IS3D];
    p.in_sc = (int)g[C2M_G_IN_SC];
    const int NS = (int)g[C2M_G_NS], Cin = (int)g[C2M_G_CIN], taps = (int)g[C2M_G_TAPS], ntg = (int)g[C2M_G_NTG], ngroups = (int)g[C2M_G_NGROUPS];
    if (g[C2M_G_X_BYTES] <= 0 || g[C2M_G_X_BYTES] >= 0x80000000LL || g[C2M_G_DY_BYTES] <= 0 || g[C2M_G_DY_BYTES] >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)g[C2M_G_X_BYTES]; p.dy_bytes = (unsigned)g[C2M_G_DY_BYTES];
    if (p.M <= 0 || p.J <= 0 || p.Npix <= 0) return 0;
    int BM, BN;
    wgrad_tile(p.M, BM, BN);
    if (p.J % BN != 0 || p.J < (ngroups + 1) * 16 || (NS != 1 && NS != 2 && NS != 4)) return (int)hipErrorInvalidValue;
    const int S = c2m_conv_wgrad_splits(p.M, p.J, p.Npix);
    int per = c2m_cdiv(p.Npix, S);
    per = ((per + 63) / 64) * 64;
    p.pix_per_split = per;
    const int Seff = c2m_cdiv(p.Npix, per);   // <= S; unused slabs are never read
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.J / BN, c2m_cdiv(p.M, BM), Seff);
    const bool xh = g[C2M_G_X_TYPE] == 1;                           // dY and X are bf16 tensors (bf16 kernels only)
    if (p.M <= 4 && p.Npix >= 16384) {                    // thin output: vector-ALU kernels, same slab layout
        if (xh) return (int)hipErrorInvalidValue;         // fp32 operands (the host casts bf16 activations)
        // g[C2M_G_SQUARE_KW] = KW: 2-D stride-1 square KW x KW tap set in row-major order -> row-blocked kernel
        const int kw = (int)g[C2M_G_SQUARE_KW];
        if (p.M <= 3 && (kw == 3 || kw == 7) && !p.is3d && p.To == 1 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.Wo % 4 == 0 &&
            taps == kw * kw && Cin > 0 && (p.dy_sn % 4) == 0 && (p.dy_sc % 4) == 0 && ((uintptr_t)dY & 15) == 0) {
            const int lns = NS == 1 ? 0 : (NS == 2 ? 1 : 2);
            // tap rows per workgroup: everything must fit the 256 architectural VGPRs (see the kernel)
            const int cpb = kw == 7 ? 1 : 2, ri = kw == 7 ? (p.M == 1 ? 4 : (p.M == 2 ? 3 : 2)) : 3;
            const int cblocks = c2m_cdiv(Cin, cpb);
            const int quads = p.Npix / 4;
            const int units = cblocks * c2m_cdiv(kw, ri);
            int St = (1024 + units - 1) / units;          // ~1024 workgroups, >= 2048 pixel groups each
            if (St > S) St = S;                           // the caller's slab holds S splits
            if (St > c2m_cdiv(quads, 2048)) St = c2m_cdiv(quads, 2048);
            if (St < 1) St = 1;
            const int qps = c2m_cdiv(quads, St);
            const int Srows = c2m_cdiv(quads, qps);
            dim3 rg(cblocks, c2m_cdiv(kw, ri), Srows);
#define C2M_THIN_WR(MT, RI7) \
            if (kw == 7) hipLaunchKernelGGL((conv_thin_wgrad_rows_kernel<MT, 7, 1, RI7>), rg, dim3(256), 0, s, p, lns, ntg, Cin, ngroups, qps); \
            else         hipLaunchKernelGGL((conv_thin_wgrad_rows_kernel<MT, 3, 2, 3>), rg, dim3(256), 0, s, p, lns, ntg, Cin, ngroups, qps);
            if (p.M == 1) { C2M_THIN_WR(1, 4) } else if (p.M == 2) { C2M_THIN_WR(2, 3) } else { C2M_THIN_WR(3, 2) }   // 4 output rows would spill
#undef C2M_THIN_WR
            int rc3 = (int)hipGetLastError();
            if (rc3) return rc3;
            return launch_wgrad_reduce((long)p.M * (ngroups + 1) * 16, s, slab, dW, db, p.M, p.J, Cin, taps, NS, ntg, ngroups, Srows);
        }
        constexpr int GPB = 2;
        const int ng = ngroups + 1;                       // real groups + the ones group
        // pixel splits sized for ~1024 blocks but >= 16K pixels each (the block reduction is per block)
        int St = (1024 + c2m_cdiv(ng, GPB) - 1) / c2m_cdiv(ng, GPB);
        const int maxS = S;                               // the caller's slab holds S splits
        if (St > maxS) St = maxS;
        if (St < 1) St = 1;
        int per2 = ((c2m_cdiv(p.Npix, St) + 255) / 256) * 256;
        p.pix_per_split = per2;
        const int Sthin = c2m_cdiv(p.Npix, per2);
        dim3 tg(c2m_cdiv(ng, GPB), 1, Sthin);
#define C2M_THIN_W(MT) \
        if (NS == 1)      hipLaunchKernelGGL((conv_thin_wgrad_kernel<MT, 1, GPB>), tg, dim3(256), 0, s, p, ng); \
        else if (NS == 2) hipLaunchKernelGGL((conv_thin_wgrad_kernel<MT, 2, GPB>), tg, dim3(256), 0, s, p, ng); \
        else              hipLaunchKernelGGL((conv_thin_wgrad_kernel<MT, 4, GPB>), tg, dim3(256), 0, s, p, ng);
        if (p.M == 1) { C2M_THIN_W(1) } else if (p.M == 2) { C2M_THIN_W(2) } else if (p.M == 3) { C2M_THIN_W(3) } else { C2M_THIN_W(4) }
#undef C2M_THIN_W
        int rc2 = (int)hipGetLastError();
        if (rc2) return rc2;
        const long total2 = (long)p.M * (ngroups + 1) * 16;
        return launch_wgrad_reduce(total2, s, slab, dW, db, p.M, p.J, Cin, taps, NS, ntg, ngroups, Sthin);
    }
    const bool bf16 = g[C2M_G_PRECISION] == 1;
    if (bf16 != xh) return (int)hipErrorInvalidValue;  // the bf16 kernel gathers bf16 tensors, the fp32 kernel fp32 ones
    const dim3 grid1(grid.x * grid.y * grid.z);       // 1-D launch, decoded XCD-aware in the kernel (common.h)
    // g[C2M_G_WGRAD_WIDE] = 1: the layer qualifies for the 16-byte-load form (host: unit x stride, Wi == Wo, Wo % 8 == 0, |tap dx| <= 1)
    if (bf16 && g[C2M_G_WGRAD_WIDE] == 2 && !getenv("C2M_WGRAD_NARROW")) {      // the stride-2 form (host: sw == 2, Wi == 2 Wo, Wo % 8 == 0, dx in -1 .. 2)
        if (p.sw != 2 || p.Wi != 2 * p.Wo || (p.Wo & 7) || (p.dy_sc & 7) || (p.pix_per_split & 63)) return (int)hipErrorInvalidValue;
        if (p.M <= 32)      hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<32, 128, 1, 4, 2>), grid1, dim3(256), 0, s, p, NS);
        else if (p.M > 64)  hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<128, 128, 2, 2, 2>), grid1, dim3(256), 0, s, p, NS);
        else                hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<64, C2M_WG64_BN, 2, 2, 2>), grid1, dim3(256), 0, s, p, NS);
        int rcw = (int)hipGetLastError();
        if (rcw) return rcw;
        const long totalw = (long)p.M * (ngroups + 1) * 16;
        return launch_wgrad_reduce(totalw, s, slab, dW, db, p.M, p.J, Cin, taps, NS, ntg, ngroups, Seff);
    }
    if (bf16 && g[C2M_G_WGRAD_WIDE] == 1 && !getenv("C2M_WGRAD_NARROW")) {
        if (p.sw != 1 || p.Wi != p.Wo || (p.Wo & 7) || p.Wi < 8 || (p.dy_sc & 7) || (p.pix_per_split & 63)) return (int)hipErrorInvalidValue;
        if (p.M <= 32)      hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<32, 128, 1, 4>), grid1, dim3(256), 0, s, p, NS);
        else if (p.M > 64)  hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<128, 128, 2, 2>), grid1, dim3(256), 0, s, p, NS);
        else                hipLaunchKernelGGL((conv_wgrad_wide_bf16_kernel<64, C2M_WG64_BN, 2, 2>), grid1, dim3(256), 0, s, p, NS);
        int rcw = (int)hipGetLastError();
        if (rcw) return rcw;
        const long totalw = (long)p.M * (ngroups + 1) * 16;
        return launch_wgrad_reduce(totalw, s, slab, dW, db, p.M, p.J, Cin, taps, NS, ntg, ngroups, Seff);
    }
#define C2M_WG(BMv, BNv, WGMv, WGNv)                                                                                  \
    do {                                                                                                              \
        if (bf16) {                                                                                                   \
            if (NS == 1)      hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 1, true>), grid1, dim3(256), 0, s, p); \
            else if (NS == 2) hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 2, true>), grid1, dim3(256), 0, s, p); \
            else              hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 4, true>), grid1, dim3(256), 0, s, p); \
        } else {                                                                                                      \
            if (NS == 1)      hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 1>), grid1, dim3(256), 0, s, p); \
            else if (NS == 2) hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 2>), grid1, dim3(256), 0, s, p); \
            else              hipLaunchKernelGGL((conv_wgrad_kernel<BMv, BNv, WGMv, WGNv, 4>), grid1, dim3(256), 0, s, p); \
        }                                                                                                             \
    } while (0)
    if (p.M <= 32)      C2M_WG(32, 128, 1, 4);
    else if (p.M > 64)  C2M_WG(128, 128, 2, 2);
    else                C2M_WG(64, C2M_WG64_BN, 2, 2);
#undef C2M_WG
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    const long total = (long)p.M * (ngroups + 1) * 16;
    return launch_wgrad_reduce(total, s, slab, dW, db, p.M, p.J, Cin, taps, NS, ntg, ngroups, Seff);
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

template <class T>
__global__ void reflect_fold_kernel(const T* __restrict__ dXp, T* __restrict__ dX, const FoldP f) {
    const int Tp = f.T + 2 * f.pt, Hp = f.H + 2 * f.ph, Wp = f.W + 2 * f.pw;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < f.total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % f.W); long r = idx / f.W;
        const int y = (int)(r % f.H); r /= f.H;
        const int t = (int)(r % f.T); const long nc = r / f.T;
        int st[3], sy[3], sx[3];
        const int nt = fold_sources(t, f.T, f.pt, st), ny = fold_sources(y, f.H, f.ph, sy),
                  nx = fold_sources(x, f.W, f.pw, sx);
        const T* __restrict__ base = dXp + nc * (long)Tp * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c) acc += c2m_ld(base, ((long)st[a] * Hp + sy[b]) * Wp + sx[c]);
        c2m_st(dX, idx, acc);
    }
}

// W % 4 == 0: a thread owns four consecutive x.  Their direct sources are four consecutive floats of the padded row (one
// 4-byte aligned 16-byte load); mirrored x sources exist only for elements within `pad` of a border and are added per
// element.  Summation order: per (t, y) source pair the direct x term plus its mirrored x terms first, then the pairs in
// (t, y) order -- NOT the scalar kernel's strictly sequential order (the mirrored x terms are added to the direct term
// before the pair joins the running sum), so the two kernels agree to rounding, not bit for bit; either is deterministic.
__device__ __forceinline__ f32x4 fold_ld4u(const float* __restrict__ p) {      // 4-byte aligned 16-byte load
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    return *reinterpret_cast<const f32x4u*>(p);
}
__device__ __forceinline__ f32x4 fold_ld4u(const bf16_t* __restrict__ p) {     // 2-byte aligned 8-byte load (unaligned access mode)
    uint2 r;
    __builtin_memcpy(&r, p, 8);
    const f32x4 v = {__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                     __uint_as_float(r.y & 0xffff0000u)};
    return v;
}

template <class T>
__global__ void reflect_fold_vec_kernel(const T* __restrict__ dXp, T* __restrict__ dX, const FoldP f) {
    typedef float f32x4a __attribute__((ext_vector_type(4)));
    const int Tp = f.T + 2 * f.pt, Hp = f.H + 2 * f.ph, Wp = f.W + 2 * f.pw;
    const int W4 = f.W >> 2;
    const long total4 = f.total >> 2;
    for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < total4; i4 += (long)gridDim.x * blockDim.x) {
        const int x0 = (int)(i4 % W4) * 4; long r = i4 / W4;
        const int y = (int)(r % f.H); r /= f.H;
        const int t = (int)(r % f.T); const long nc = r / f.T;
        int st[3], sy[3];
        const int nt = fold_sources(t, f.T, f.pt, st), ny = fold_sources(y, f.H, f.ph, sy);
        const T* __restrict__ base = dXp + nc * (long)Tp * Hp * Wp;
        const bool xband = f.pw > 0 && (x0 <= f.pw || x0 + 3 >= f.W - 1 - f.pw);     // some element has mirrored x sources
        f32x4a acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < ny; ++b) {
                const T* __restrict__ row = base + ((long)st[a] * Hp + sy[b]) * Wp;
                f32x4a v = fold_ld4u(row + x0 + f.pw);
                if (xband) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int sx[3];
                        const int nx = fold_sources(x0 + e, f.W, f.pw, sx);
                        for (int c = 1; c < nx; ++c) v[e] += c2m_ld(row, sx[c]);
                    }
                }
                acc += v;
            }
        c2m_st4(dX + i4 * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// In-place variant for the two-target dgrad: dX already holds the direct term, add the mirrored pad-ring terms
// (only pixels within `pad` of a border have any).  Generic form: every element looks for extra sources.
template <class T>
__global__ void reflect_border_add_kernel(const T* __restrict__ dXp, T* __restrict__ dX, const FoldP f) {
    const int Tp = f.T + 2 * f.pt, Hp = f.H + 2 * f.ph, Wp = f.W + 2 * f.pw;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < f.total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % f.W); long r = idx / f.W;
        const int y = (int)(r % f.H); r /= f.H;
        const int t = (int)(r % f.T); const long nc = r / f.T;
        int st[3], sy[3], sx[3];
        const int nt = fold_sources(t, f.T, f.pt, st), ny = fold_sources(y, f.H, f.ph, sy),
                  nx = fold_sources(x, f.W, f.pw, sx);
        if (nt * ny * nx == 1) continue;
        const T* __restrict__ base = dXp + nc * (long)Tp * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c)
                    if (a + b + c > 0) acc += c2m_ld(base, ((long)st[a] * Hp + sy[b]) * Wp + sx[c]);
        c2m_st(dX, idx, c2m_ld(dX, idx) + acc);
    }
}

// Border-only form (every extent >= 2*pad + 2, so the low and high mirror bands are disjoint): enumerates just the
// elements that have a mirrored source, as three disjoint ranges per (n,c) volume:
//   R1  x in band,                      all y, all t        T * H * 2pw
//   R2  x outside band, y in band,      all t               T * 2ph * (W - 2pw)
//   R3  x, y outside their bands,       t in band           2pt * (H - 2ph) * (W - 2pw)
// band(i) = [1, p] U [L-1-p, L-2]; the complement is {0} U [p+1, L-2-p] U {L-1}.
__device__ __forceinline__ int band_index(int j, int L, int p) { return j < p ? 1 + j : L - 1 - 2 * p + j; }
__device__ __forceinline__ int offband_index(int k, int L, int p) {
    return k == 0 ? 0 : (k == L - 2 * p - 1 ? L - 1 : k + p);
}

template <class T>
__global__ void reflect_border_only_kernel(const T* __restrict__ dXp, T* __restrict__ dX, const FoldP f,
                                           const int r1, const int r2, const int r3, const long NC) {
    const int Tp = f.T + 2 * f.pt, Hp = f.H + 2 * f.ph, Wp = f.W + 2 * f.pw;
    const int per = r1 + r2 + r3;
    const long total = NC * per;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long nc = idx / per;
        int e = (int)(idx - nc * per);
        int t, y, x;
        if (e < r1) {
            const int bw = 2 * f.pw;
            x = band_index(e % bw, f.W, f.pw); e /= bw;
            y = e % f.H; t = e / f.H;
        } else if (e < r1 + r2) {
            e -= r1;
            const int ow = f.W - 2 * f.pw, bh = 2 * f.ph;
            x = offband_index(e % ow, f.W, f.pw); e /= ow;
            y = band_index(e % bh, f.H, f.ph); t = e / bh;
        } else {
            e -= r1 + r2;
            const int ow = f.W - 2 * f.pw, oh = f.H - 2 * f.ph;
            x = offband_index(e % ow, f.W, f.pw); e /= ow;
            y = offband_index(e % oh, f.H, f.ph); t = band_index(e / oh, f.T, f.pt);
        }
        int st[3], sy[3], sx[3];
        const int nt = fold_sources(t, f.T, f.pt, st), ny = fold_sources(y, f.H, f.ph, sy),
                  nx = fold_sources(x, f.W, f.pw, sx);
        const T* __restrict__ base = dXp + nc * (long)Tp * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < ny; ++b)
                for (int c = 0; c < nx; ++c)
                    if (a + b + c > 0) acc += c2m_ld(base, ((long)st[a] * Hp + sy[b]) * Wp + sx[c]);
        const long di = (nc * f.T + t) * (long)f.H * f.W + (long)y * f.W + x;
        c2m_st(dX, di, c2m_ld(dX, di) + acc);
    }
}

// The 2-D pad-1 case (every 3x3 / 4x4-stride-2 reflect layer; a 3-D layer folded in the plane only is T independent planes): the
// elements that receive ring contributions are rows 1, H-2 and columns 1, W-2 of each plane.  The general kernel above finds its
// element with ~8 integer divisions by run-time values (20 us per launch on a 40 x 128 x 64 x 128 gradient: ALU bound, 55 launches
// per step); here one division splits (plane, e) and comparisons do the rest.  Same sources in the same order: bit-identical sums.
template <class T>
__global__ void reflect_border_p1_kernel(const T* __restrict__ dXp, T* __restrict__ dX, const int H, const int W, const long planes) {
    const int Hp = H + 2, Wp = W + 2;
    const int per = 2 * W + 2 * (H - 2);
    const long total = planes * per;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long pl = idx / per;
        const int e = (int)(idx - pl * per);
        int y, x;
        if (e < 2 * W) {                               // the two rows: contiguous in x
            y = e < W ? 1 : H - 2;
            x = e < W ? e : e - W;
        } else {                                       // the two columns without the rows above
            const int k2 = e - 2 * W;
            x = k2 < H - 2 ? 1 : W - 2;
            const int k = k2 < H - 2 ? k2 : k2 - (H - 2);
            y = k == 0 ? 0 : (k == H - 3 ? H - 1 : k + 1);
        }
        int sy[3], sx[3];
        const int ny = fold_sources(y, H, 1, sy), nx = fold_sources(x, W, 1, sx);
        const T* __restrict__ base = dXp + pl * (long)Hp * Wp;
        float acc = 0.f;
        for (int b = 0; b < ny; ++b)
            for (int c = 0; c < nx; ++c)
                if (b + c > 0) acc += c2m_ld(base, (long)sy[b] * Wp + sx[c]);
        const long di = pl * (long)H * W + (long)y * W + x;
        c2m_st(dX, di, c2m_ld(dX, di) + acc);
    }
}

C2M_API int c2m_reflect_border_add(const void* dXpad, void* dX, long NC, int T_, int H, int W, int pt, int ph, int pw,
                                   int dt, void* stream) {
    C2M_ENTER();
    FoldP f{T_, H, W, pt, ph, pw, NC * (long)T_ * H * W};
    if (f.total <= 0) return 0;
    const bool roomy = (pt == 0 || T_ >= 2 * pt + 2) && (ph == 0 || H >= 2 * ph + 2) && (pw == 0 || W >= 2 * pw + 2);
    const long per_l = (long)T_ * H * 2 * pw + (long)T_ * 2 * ph * (W - 2 * pw) + 2L * pt * (H - 2 * ph) * (W - 2 * pw);
    if (pt == 0 && ph == 1 && pw == 1 && H >= 4 && W >= 4 && !getenv("C2M_FOLD_GENERAL")) {
        const long planes = NC * T_;
        C2M_DISPATCH_DT(dt,
            hipLaunchKernelGGL(reflect_border_p1_kernel<T>, dim3(c2m_grid(planes * (2L * W + 2L * (H - 2)), 256)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)dXpad, (T*)dX, H, W, planes););
        return (int)hipGetLastError();
    }
    C2M_DISPATCH_DT(dt,
        if (roomy && per_l > 0 && per_l < (1L << 30)) {
            const int r1 = T_ * H * 2 * pw; const int r2 = T_ * 2 * ph * (W - 2 * pw); const int r3 = 2 * pt * (H - 2 * ph) * (W - 2 * pw);
            hipLaunchKernelGGL(reflect_border_only_kernel<T>, dim3(c2m_grid(NC * per_l, 256)), dim3(256), 0,
                               (hipStream_t)stream, (const T*)dXpad, (T*)dX, f, r1, r2, r3, NC);
        } else {
            hipLaunchKernelGGL(reflect_border_add_kernel<T>, dim3(c2m_grid(f.total, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)dXpad, (T*)dX, f);
        });
    return (int)hipGetLastError();
}

C2M_API int c2m_reflect_fold(const void* dXpad, void* dX, long NC, int T_, int H, int W, int pt, int ph, int pw,
                             int dt, void* stream) {
    C2M_ENTER();
    FoldP f{T_, H, W, pt, ph, pw, NC * (long)T_ * H * W};
    if (f.total <= 0) return 0;
    if (dt == C2M_BF16 && (W & 3) == 0 && ((uintptr_t)dX & 7) == 0)
        hipLaunchKernelGGL(reflect_fold_vec_kernel<bf16_t>, dim3(c2m_grid(f.total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dXpad, (bf16_t*)dX, f);
    else if (dt == C2M_BF16)
        hipLaunchKernelGGL(reflect_fold_kernel<bf16_t>, dim3(c2m_grid(f.total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dXpad, (bf16_t*)dX, f);
    else if ((W & 3) == 0 && ((uintptr_t)dX & 15) == 0)
        hipLaunchKernelGGL(reflect_fold_vec_kernel<float>, dim3(c2m_grid(f.total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dXpad, (float*)dX, f);
    else
        hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(c2m_grid(f.total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dXpad, (float*)dX, f);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ weight packing
// Native conv weights -> the kernels' K order in one pass (coalesced writes, gathered reads; weights sit in L2).
//   out[r][((chunk * ntg + tg) * NS + slot) * CK + ch],  r = cls * M + m,  c = chunk*CK + ch,  tap = tg*NS + slot
//   source element: w[m*s_m + c*s_c + ((at*st + rt)*KH + (ay*sh + ry))*KW + (ax*sw + rx)]
//   with tap = (at, ay, ax) over (At, Ay, Ax) = (KT/st, KH/sh, KW/sw) and cls = (rt, ry, rx) over (st, sh, sw).
// Forward: st = sh = sw = 1, s_m = C*taps, s_c = taps.  Data gradient (rows = input channels, K = output channels,
// one class per stride parity): s_m = taps_full, s_c = M*taps_full.  Padding (c >= C or tap >= taps) is written as 0.
struct PackP { int M, C, CK, NS, nch, ntg, At, Ay, Ax, st, sh, sw, KH, KW; long s_m, s_c; };

// one workgroup row per packed row r = cls*M + m (blockIdx.y); the per-tap source offsets of the row's parity class are
// computed once into LDS, CK / NS are powers of two: one integer division (by ntg) per element remains
template <int CK>
__device__ __forceinline__ void pack_weights_row(const float* __restrict__ w, float* __restrict__ out, const PackP& q, int* toff,
                                                 const int r, const int bx, const int nbx) {
    constexpr int NS = 16 / CK, LCK = CK == 16 ? 4 : (CK == 8 ? 3 : 2), LNS = 4 - LCK;
    const int taps = q.At * q.Ay * q.Ax;
    const int lda = q.nch * q.ntg * 16;
    const int m = r % q.M; int cls = r / q.M;
    const int rx = cls % q.sw; cls /= q.sw;
    const int ry = cls % q.sh; const int rt = cls / q.sh;
    for (int tap = threadIdx.x; tap < taps; tap += 256) {
        const int ax = tap % q.Ax; const int ay = (tap / q.Ax) % q.Ay; const int at = tap / (q.Ax * q.Ay);
        toff[tap] = ((at * q.st + rt) * q.KH + (ay * q.sh + ry)) * q.KW + (ax * q.sw + rx);
    }
    __syncthreads();
    const float* __restrict__ wr = w + m * q.s_m;
    float* __restrict__ orow = out + (long)r * lda;
    for (int k = bx * 256 + threadIdx.x; k < lda; k += nbx * 256) {
        const int ch = k & (CK - 1); int t = k >> LCK;
        const int slot = t & (NS - 1); t >>= LNS;
        const int chunk = t / q.ntg, tg = t - chunk * q.ntg;
        const int c = chunk * CK + ch, tap = tg * NS + slot;
        orow[k] = (c < q.C && tap < taps) ? wr[c * q.s_c + toff[tap]] : 0.f;
    }
}

template <int CK>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                            const PackP q) {
    __shared__ int toff[128];
    pack_weights_row<CK>(w, out, q, toff, blockIdx.y, blockIdx.x, gridDim.x);
}

// bf16 weights of the bf16 LDS-patch kernel: out[chunk][tap][row (padded to Mpad)][16 channels] in bf16, RNE; rows >= M
// and channels >= C are zero.  3x3 taps only; forward (flip = 0): tap = ky*3+kx of w[m][c][ky][kx] addressed through
// s_m / s_c as in c2m_pack_weights; data gradient (flip = 1): the tap index is reversed (rotated filter).
__device__ __forceinline__ void pack_weights_bf16_patch_body(const float* __restrict__ w, uint4* __restrict__ out, int M, int C,
                                                             int Mpad, long s_m, long s_c, int flip, long units, long b0, long nb) {
    for (long u = b0 * (long)blockDim.x + threadIdx.x; u < units; u += nb * blockDim.x) {
        const int half = (int)(u & 1); long r = u >> 1;
        const int m = (int)(r % Mpad); r /= Mpad;
        int tap, chunk, st;
        if (flip == 2) {
            // 4x4 stride-2 layers on the parity-plane kernel (conv_nc8.hip, S2): out[16-channel chunk][parity py*2+px][tap a*2+b]
            // [row][16 channels] with original tap (ky, kx) = (2a + 1 - py, 2b + 1 - px)
            const int t4 = (int)(r % 4); r /= 4;
            const int par = (int)(r % 4); chunk = (int)(r / 4);
            st = (2 * (t4 >> 1) + 1 - (par >> 1)) * 4 + 2 * (t4 & 1) + 1 - (par & 1);
        } else if (flip == 3 || flip == 4) {
            // data gradient of those layers (conv_s2_dgrad_nc8_kernel): 16 virtual taps vt = class (ri, rj) * 4 + a * 2 + b with
            // ky = zeros: ri ? 2a : 2a + 1, reflect: ri ? 2a + 1 : 2a (same along x); rows = input channels, reduction = output channels
            const int vt = (int)(r % 16); chunk = (int)(r / 16);
            const int ri = vt >> 3, rj = (vt >> 2) & 1, a = (vt >> 1) & 1, b = vt & 1;
            const int ky = flip == 3 ? (ri ? 2 * a : 2 * a + 1) : (ri ? 2 * a + 1 : 2 * a);
            const int kx = flip == 3 ? (rj ? 2 * b : 2 * b + 1) : (rj ? 2 * b + 1 : 2 * b);
            st = ky * 4 + kx;
        } else {
            tap = (int)(r % 9); chunk = (int)(r / 9);
            st = flip ? 8 - tap : tap;
        }
        bf16x8 q;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 16 + half * 8 + j;
            q[j] = (__bf16)((m < M && c < C) ? w[m * s_m + c * s_c + st] : 0.f);
        }
        out[u] = __builtin_bit_cast(uint4, q);
    }
}

// bf16 weights of the NC8 gather kernel: out[class][tap][16-channel chunk][half][row (padded to Mpad)] 16-byte units, RNE; the
// source element of (class, tap, m, c) is the one pack_weights_row reads (PackP with CK = 16); rows >= M and channels >= C are zero.
__device__ __forceinline__ void pack_weights_bf16_gather_body(const float* __restrict__ w, uint4* __restrict__ out, const PackP& q,
                                                              int Mpad, long units, long b0, long nb) {
    const int taps = q.At * q.Ay * q.Ax;
    for (long u = b0 * (long)blockDim.x + threadIdx.x; u < units; u += nb * blockDim.x) {
        const int m = (int)(u % Mpad); long r = u / Mpad;
        const int half = (int)(r & 1); r >>= 1;
        const int chunk = (int)(r % q.nch); r /= q.nch;
        const int tap = (int)(r % taps); int cls = (int)(r / taps);
        const int rx = cls % q.sw; cls /= q.sw;
        const int ry = cls % q.sh; const int rt = cls / q.sh;
        const int ax = tap % q.Ax, ay = (tap / q.Ax) % q.Ay, at = tap / (q.Ax * q.Ay);
        const long toff = ((at * q.st + rt) * q.KH + (ay * q.sh + ry)) * q.KW + (ax * q.sw + rx);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 16 + half * 8 + j;
            v[j] = (__bf16)((m < q.M && c < q.C) ? w[m * q.s_m + c * q.s_c + toff] : 0.f);
        }
        out[u] = __builtin_bit_cast(uint4, v);
    }
}

__global__ void pack_weights_bf16_gather_kernel(const float* __restrict__ w, uint4* __restrict__ out, const PackP q, int Mpad,
                                                long units) {
    pack_weights_bf16_gather_body(w, out, q, Mpad, units, blockIdx.x, gridDim.x);
}

__global__ void pack_weights_bf16_patch_kernel(const float* __restrict__ w, uint4* __restrict__ out, int M, int C, int Mpad,
                                               long s_m, long s_c, int flip, long units) {
    pack_weights_bf16_patch_body(w, out, M, C, Mpad, s_m, s_c, flip, units, blockIdx.x, gridDim.x);
}

// out: ceil(C/16) * 9 * Mpad * 32 bytes with Mpad = ceil(M/128)*128.  g[]: 0 M, 1 C, 2 s_m, 3 s_c, 4 flip.
C2M_API long c2m_pack_weights_bf16_patch_bytes(int M, int C) {
    return (long)c2m_cdiv(C, 16) * 9 * (c2m_cdiv(M, 128) * 128) * 32;
}
// (flip = 2: the 4x4 stride-2 parity form, 16 (parity, tap) slots per chunk instead of 9 taps)
C2M_API long c2m_pack_weights_bf16_s2_bytes(int M, int C) {
    return (long)c2m_cdiv(C, 16) * 16 * (c2m_cdiv(M, 128) * 128) * 32;
}

C2M_API int c2m_pack_weights_bf16_patch(const float* w, void* out, const int64_t* g, void* stream) {
    C2M_ENTER();
    const int M = (int)g[0], C = (int)g[1];
    if (M <= 0 || C <= 0) return 0;
    if ((((uintptr_t)out) & 15) != 0) return (int)hipErrorInvalidValue;
    const int Mpad = c2m_cdiv(M, 128) * 128;
    const long units = (long)c2m_cdiv(C, 16) * (g[4] >= 2 ? 16 : 9) * Mpad * 2;
    hipLaunchKernelGGL(pack_weights_bf16_patch_kernel, dim3(c2m_grid(units, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (uint4*)out, M, C, Mpad, (long)g[2], (long)g[3], (int)g[4], units);
    return (int)hipGetLastError();
}

// g[]: 0 M, 1 C, 2 CK, 3 KT, 4 KH, 5 KW, 6 st, 7 sh, 8 sw, 9 s_m, 10 s_c;  out holds st*sh*sw * M rows of
// ceil(C/CK) * ceil(taps/NS) * 16 floats (NS = 16/CK, taps = (KT/st)*(KH/sh)*(KW/sw)).
static int pack_params(const int64_t* g, PackP& q, long& rows, int& lda) {
    q.M = (int)g[0]; q.C = (int)g[1]; q.CK = (int)g[2];
    const int KT = (int)g[3];
    q.KH = (int)g[4]; q.KW = (int)g[5]; q.st = (int)g[6]; q.sh = (int)g[7]; q.sw = (int)g[8];
    q.s_m = g[9]; q.s_c = g[10];
    rows = 0; lda = 0;
    if (q.M <= 0 || q.C <= 0) return 0;
    if ((q.CK != 4 && q.CK != 8 && q.CK != 16) || q.st < 1 || q.sh < 1 || q.sw < 1 || KT % q.st || q.KH % q.sh ||
        q.KW % q.sw) return (int)hipErrorInvalidValue;
    q.NS = 16 / q.CK;
    q.At = KT / q.st; q.Ay = q.KH / q.sh; q.Ax = q.KW / q.sw;
    q.nch = c2m_cdiv(q.C, q.CK); q.ntg = c2m_cdiv(q.At * q.Ay * q.Ax, q.NS);
    rows = (long)q.st * q.sh * q.sw * q.M;
    lda = q.nch * q.ntg * 16;
    if (q.At * q.Ay * q.Ax > 128 || rows > 65535) return (int)hipErrorInvalidValue;
    return 0;
}

// g[] as c2m_pack_weights (g[2] = CK must be 16).  out: prod(stride) class images of taps * ceil(C/16) * 2 * Mpad 16-byte units
// (taps = (KT/st)(KH/sh)(KW/sw), Mpad = ceil(M/128)*128), classes in (rt, ry, rx) row-major order.
C2M_API long c2m_pack_weights_bf16_gather_bytes(const int64_t* g) {
    PackP q; long rows; int lda;
    if (pack_params(g, q, rows, lda) || rows == 0 || q.CK != 16) return -1;
    return (long)q.st * q.sh * q.sw * q.At * q.Ay * q.Ax * q.nch * 2 * (c2m_cdiv(q.M, 128) * 128) * 16;
}

C2M_API int c2m_pack_weights_bf16_gather(const float* w, void* out, const int64_t* g, void* stream) {
    C2M_ENTER();
    PackP q; long rows; int lda;
    const int rc = pack_params(g, q, rows, lda);
    if (rc || rows == 0) return rc;
    if (q.CK != 16 || (((uintptr_t)out) & 15)) return (int)hipErrorInvalidValue;
    const int Mpad = c2m_cdiv(q.M, 128) * 128;
    const long units = (long)q.st * q.sh * q.sw * q.At * q.Ay * q.Ax * q.nch * 2 * Mpad;
    hipLaunchKernelGGL(pack_weights_bf16_gather_kernel, dim3(c2m_grid(units, 256)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)out,
                       q, Mpad, units);
    return (int)hipGetLastError();
}

C2M_API int c2m_pack_weights(const float* w, float* out, const int64_t* g, void* stream) {
    C2M_ENTER();
    PackP q; long rows; int lda;
    const int rc = pack_params(g, q, rows, lda);
    if (rc || rows == 0) return rc;
    dim3 grid(c2m_cdiv(lda, 1024) < 1 ? 1 : c2m_cdiv(lda, 1024), (unsigned)rows);
    if (q.CK == 16)     hipLaunchKernelGGL((pack_weights_kernel<16>), grid, dim3(256), 0, (hipStream_t)stream, w, out, q);
    else if (q.CK == 8) hipLaunchKernelGGL((pack_weights_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, w, out, q);
    else                hipLaunchKernelGGL((pack_weights_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, w, out, q);
    return (int)hipGetLastError();
}

// ---- all packs of a model in ONE launch (round 3).  With an optimizer in the step every trainable weight is re-packed once per
// step and layout -- ~195 launches of 5-7 us in a full G + D step (1.3 ms of GPU time, more host time than that in eager mode).
// The host keeps one PackJob per (weight, layout) it has packed before (c2m_pack_job_fill), the table lives in device memory,
// and after the optimizer step one launch refreshes every pack in place: workgroup b finds its job by bisection over the jobs'
// first workgroup index.  type 0: c2m_pack_weights (g[11]); type 1: c2m_pack_weights_bf16_patch (g[5]); type 2:
// c2m_pack_weights_bf16_gather (g[11]).
struct PackJob {
    const float* w; void* out;
    PackP q;                                                  // type 0
    int M, C, Mpad, flip; long s_m, s_c, units;               // type 1
    int type; unsigned first, xblocks, nblocks;
};

C2M_API int c2m_pack_job_bytes(void) { return (int)sizeof(PackJob); }

// fills *job (host memory, c2m_pack_job_bytes() bytes); returns the number of workgroups the job takes, < 0 on bad geometry
C2M_API long c2m_pack_job_fill(void* job, int type, const void* w, void* out, const int64_t* g, unsigned first_block) {
    PackJob j;
    memset(&j, 0, sizeof(j));
    j.w = (const float*)w; j.out = out; j.type = type; j.first = first_block;
    if (type == 0) {
        long rows; int lda;
        if (pack_params(g, j.q, rows, lda) || rows == 0) return -1;
        j.xblocks = (unsigned)(c2m_cdiv(lda, 1024) < 1 ? 1 : c2m_cdiv(lda, 1024));
        j.nblocks = j.xblocks * (unsigned)rows;
    } else if (type == 1) {
        j.M = (int)g[0]; j.C = (int)g[1]; j.s_m = g[2]; j.s_c = g[3]; j.flip = (int)g[4];
        if (j.M <= 0 || j.C <= 0 || (((uintptr_t)out) & 15) != 0) return -1;
        j.Mpad = c2m_cdiv(j.M, 128) * 128;
        j.units = (long)c2m_cdiv(j.C, 16) * (j.flip >= 2 ? 16 : 9) * j.Mpad * 2;
        j.xblocks = 1;
        j.nblocks = (unsigned)c2m_grid(j.units, 256);
    } else if (type == 2) {                                   // c2m_pack_weights_bf16_gather (g[11])
        long rows; int lda;
        if (pack_params(g, j.q, rows, lda) || rows == 0 || j.q.CK != 16 || (((uintptr_t)out) & 15) != 0) return -1;
        j.Mpad = c2m_cdiv(j.q.M, 128) * 128;
        j.units = (long)j.q.st * j.q.sh * j.q.sw * j.q.At * j.q.Ay * j.q.Ax * j.q.nch * 2 * j.Mpad;
        j.xblocks = 1;
        j.nblocks = (unsigned)c2m_grid(j.units, 256);
    } else {
        return -1;
    }
    memcpy(job, &j, sizeof(j));
    return (long)j.nblocks;
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, const int2* __restrict__ blocktab) {
    __shared__ int toff[128];
    // (job, workgroup within the job) of this workgroup from a host-built table: the first form bisected over the jobs' first
    // workgroup index -- eight dependent global loads (~5 us) in front of ~1 us of copying, 185 us per launch on the G + D set
    const int2 e = blocktab[blockIdx.x];
    const PackJob& j = jobs[e.x];
    const unsigned lb = (unsigned)e.y;
    if (j.type == 0) {
        const int r = (int)(lb / j.xblocks), bx = (int)(lb % j.xblocks);
        const PackP q = j.q;
        if (q.CK == 16)     pack_weights_row<16>(j.w, (float*)j.out, q, toff, r, bx, (int)j.xblocks);
        else if (q.CK == 8) pack_weights_row<8>(j.w, (float*)j.out, q, toff, r, bx, (int)j.xblocks);
        else                pack_weights_row<4>(j.w, (float*)j.out, q, toff, r, bx, (int)j.xblocks);
    } else if (j.type == 2) {
        const PackP q = j.q;
        pack_weights_bf16_gather_body(j.w, (uint4*)j.out, q, j.Mpad, j.units, lb, j.nblocks);
    } else {
        pack_weights_bf16_patch_body(j.w, (uint4*)j.out, j.M, j.C, j.Mpad, j.s_m, j.s_c, j.flip, j.units, lb, j.nblocks);
    }
}

// jobs: device copy of njobs PackJob records; blocktab: device int32 pairs (job index, workgroup index within the job), one per
// workgroup of the launch, in any order (total_blocks = the sum of the jobs' workgroup counts)
C2M_API int c2m_pack_multi(const void* jobs, const void* blocktab, int njobs, long total_blocks, void* stream) {
    C2M_ENTER();
    if (njobs <= 0 || total_blocks <= 0) return 0;
    if (total_blocks > 0x7fffffffL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                       (const int2*)blocktab);
    return (int)hipGetLastError();
}
