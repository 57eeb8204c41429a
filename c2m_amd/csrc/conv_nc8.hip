// bf16 convolutions over CHANNEL-BLOCKED activations (round 4; BASELINE configs[2-4]).
//
// On NCHW the 8 consecutive input channels one lane of v_mfma_f32_32x32x16_bf16 consumes sit 2*H*W bytes apart, so the bf16
// kernels of rounds 2-3 (conv_igemm.hip) gather 2 bytes per lane per load and run the matrix pipe 0.09-0.18 busy.  Here the
// conv INPUT is "NC8": [N][ceil(C/8)][H][W][8] bf16 -- the 8 channels of one pixel are one 16-byte unit -- produced by
// c2m_nchw_to_nc8 (one pass; fused into the producing kernels where they exist) and consumed by
//
//   conv_patch_nc8_kernel<BM>: 3x3 stride-1 layers, forward and data gradient (residual_block.py:13-31,42-71,
//   spade_block.py:47-49, vgg.py:92-137, up_block.py:9-13, same_block.py:14-23).  Per 16-channel chunk the (8+2) x 34 pixel
//   patch ([half][pixel] 16-byte units) and the 9 x BM x 16 weight image ([tap][half][row]) go global -> LDS by 16-byte
//   LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, 8 DMA instructions per wave and chunk instead of
//   24 two-byte gathers + 8 packs + 12 ds_write_b128.  Every B fragment of every tap is one conflict-free ds_read_b128 of the
//   patch at a pixel offset.  Two LDS buffers of 32 KB (DMA destinations must stay below 64 KB, conv_wino4.hip), two
//   workgroups per CU, ONE barrier per chunk: wait for this chunk's DMA, barrier, issue the next chunk's DMA into the buffer
//   everybody has just left, 36 MFMAs per wave (BM = 64).
//   Output: NCHW bf16 / fp32 exactly as conv_patch3x3_bf16_kernel writes it (two-target reflect data gradient, split-K slabs,
//   bias + activation), so nothing downstream changes.
#include "common.h"
#include "dtype.h"
#include "conv_store.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NC8_OOB 0x80000000u

// ------------------------------------------------------------------------------------------------ NCHW -> NC8
// out[n][cb][p][j] = in[n][cb*8 + j][p] (zeros for channels >= C).  A thread owns 8 consecutive pixels of one channel block:
// eight 16-byte loads (one per channel: contiguous over the lanes), an 8x8 transpose of 16-bit elements in registers, eight
// 16-byte stores = 128 contiguous bytes per thread.
__global__ __launch_bounds__(256) void nchw_to_nc8_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int C, int CB,
                                                          long HW8 /* HW / 8 */, long total /* N*CB*HW8 */) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long g = i % HW8;
        const long ncb = i / HW8;
        const int cb = (int)(ncb % CB);
        const long n = ncb / CB;
        unsigned r[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb * 8 + j;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (c < C) v = in[(n * C + c) * HW8 + g];
            r[j][0] = v.x; r[j][1] = v.y; r[j][2] = v.z; r[j][3] = v.w;
        }
        // r[j][q] = pixels (2q, 2q+1) of channel j  ->  o[px][q'] = channels (2q', 2q'+1) of pixel px
        uint4* __restrict__ dst = out + ((n * CB + cb) * HW8 + g) * 8;
#pragma unroll
        for (int px = 0; px < 8; ++px) {
            const int q = px >> 1;
            const unsigned sel = (px & 1) ? 0x07060302u : 0x05040100u;      // high / low halves of (src0 = odd channel, src1 = even channel)
            uint4 o;
            o.x = __builtin_amdgcn_perm(r[1][q], r[0][q], sel);
            o.y = __builtin_amdgcn_perm(r[3][q], r[2][q], sel);
            o.z = __builtin_amdgcn_perm(r[5][q], r[4][q], sel);
            o.w = __builtin_amdgcn_perm(r[7][q], r[6][q], sel);
            dst[px] = o;
        }
    }
}

// Elements per image plane must be a multiple of 8 (every map of the path is); x and y are 16-byte aligned.
C2M_API int c2m_nchw_to_nc8(const void* x, void* y, long N, int C, long HW, void* stream) {
    C2M_ENTER();
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if ((HW & 7) || (((uintptr_t)x | (uintptr_t)y) & 15)) return (int)hipErrorInvalidValue;
    const int CB = (C + 7) / 8;
    const long total = N * CB * (HW / 8);
    hipLaunchKernelGGL(nchw_to_nc8_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4*>(x), reinterpret_cast<uint4*>(y), C, CB, HW / 8, total);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 3x3 stride-1 patch kernel
struct Nc8P {
    const void* A;       // c2m_pack_weights_bf16_patch image: [chunk][tap 9][Mpad rows][2 halves] 16-byte units
    const void* X;       // NC8 activations
    float* Y;            // output (NCHW, fp32 or bf16), or the slab base when splits > 1
    float* Y2;           // optional interior target of the reflect data gradient (ConvP semantics, conv_igemm.hip)
    const float* bias;
    int M, Mpad, nchunks, CB;
    int Nimg, Ho, Wo, Hi, Wi;
    int iy0, ix0, pty[3], ptx[3];
    int reflect, act, yh, chunks_per_split;
    float slope;
    // 3x3x3 layers (same_block.py:50-68, the fuse_convs of the motion decoder) as 2-D launches over (sample, frame) images with the
    // chunk index running over (time tap, 16 channels): T frames per sample (1: a 2-D layer), nch2d chunks per time tap, input
    // frame = t + kt + t0 (reflected or zero outside [0, T)); out_st = output frame stride
    int T, nch2d, t0, treflect;
    const int* ptab;     // data gradient of a 3x3x3 reflect layer: per output frame {npairs, (dY frame, time tap) x 5}: the pad frames'
                         // contributions folded onto the frames they mirror (ops._time_pair_table_kt); chunk = (pair, 16 channels)
    long out_st;
    long out_sn, out_sc, out_sh, out_off, slab_stride;
    int ps_y, ps_x, po_y, po_x, lo_y, lo_x, ext_y, ext_x;
    long y2_sn, y2_sc, y2_sh;
    unsigned x_bytes, a_bytes;
};

// S2 = true: the 4x4 stride-2 pad-1 layers (down_block.py:14-23, the discriminators) as 2x2 stride-1 correlations over the four
// PARITY planes of the input: a "chunk" is (16 channels, parity (py, px)); its patch holds plane rows y' = oy0 + prow - py, i.e.
// input pixels (2 * (oy0 + prow) - py, 2 * (ox0 + pcol) - px) -- a stride-2 gather is free for an LDS-DMA, whose per-lane global
// address is arbitrary --, and output (y, x) takes taps (a, b) in {0,1}^2 at patch (y - oy0 + a, x - ox0 + b), which are the
// original taps ky = 2a + 1 - py, kx = 2b + 1 - px.  Four chunks per 16 channels, 4 taps each: no zero-stuffed MACs.
template <int BM, int NBUF, int WGS, int TR, bool S2 = false>
__global__ __launch_bounds__(256, WGS) void conv_patch_nc8_kernel(const Nc8P p) {
    constexpr int NTAPS = S2 ? 4 : 9;
    constexpr int PW = S2 ? 33 : 34, NPIX = (TR + (S2 ? 1 : 2)) * PW;   // 340 patch pixels (8-row tile), 612 (16-row tile); 297 (S2)
    constexpr int PROWS = ((NPIX + 63) / 64 + 1) / 2 * 2;         // DMA rows of 64 units per half plane: 6 / 10
    constexpr int PPL = PROWS * 64;
    constexpr int PPW = PROWS / 2;                                // ... per wave (wave w: half w >> 1, rows PPW * (w & 1) ...)
    static_assert(PROWS % 2 == 0, "patch DMA rows split evenly over the two waves of a half plane");
    constexpr int MI = BM / 32, NI = TR / 4;
    constexpr int A_UNITS = NTAPS * BM * 2;
    constexpr int NAI = (A_UNITS + 255) / 256;                    // weight DMA rows per wave and chunk
    constexpr int A_PAD = NAI * 256;
    constexpr int BUF = A_PAD + 2 * PPL;                          // units per buffer: 2048 (32 KB) at BM = 64
    constexpr int NDMA = NAI + PPW;
    constexpr int CG = BM >= 64 ? 64 : 32;                        // channels per staging pass of the epilogue
    constexpr int T_UNITS = 4 * CG * 64 / 4;                      // its tile: [wave][channel][64 pixels] fp32
    // (LDS-DMA destinations above 64 KB are fine: tools/micro/lds_dma_high.hip)
    static_assert(NBUF * BUF * 16 * WGS <= 160 * 1024, "LDS");
    __shared__ uint4 smem[NBUF * BUF >= T_UNITS ? NBUF * BUF : T_UNITS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.Wo + 31) / 32, tiles_y = (p.Ho + TR - 1) / TR;
    const C2mBlock blk = c2m_xcd_block((unsigned)(p.Nimg * tiles_y * tiles_x), (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM;
    int tb = blk.x;
    const int tx = tb % tiles_x; tb /= tiles_x;
    const int ty = tb % tiles_y; const int n_img = tb / tiles_y;
    const int oy0 = ty * TR, ox0 = tx * 32;
    const int n_smp = n_img / p.T, t_img = n_img - n_smp * p.T;     // (sample, output frame); T = 1 for 2-D layers
    int pr_n = 0, pr_to[5] = {0, 0, 0, 0, 0}, pr_kt[5] = {0, 0, 0, 0, 0};
    if (p.ptab) {
        const int* __restrict__ e = p.ptab + t_img * 11;
        pr_n = __builtin_amdgcn_readfirstlane(e[0]);
#pragma unroll
        for (int j = 0; j < 5; ++j) { pr_to[j] = __builtin_amdgcn_readfirstlane(e[1 + 2 * j]); pr_kt[j] = __builtin_amdgcn_readfirstlane(e[2 + 2 * j]); }
    }

    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) uint4*)&smem[0];
    // ---- weight DMA: destination unit d = (i * 4 + wave) * 64 + lane of [tap][half][row]; source [tap][m0 + row][half]
    const unsigned long aaddr = (unsigned long)p.A;
    const u32x4 ars = {(unsigned)aaddr, (unsigned)(aaddr >> 32) & 0xffffu, p.a_bytes, 0x00020000u};
    unsigned avo[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int d = (i * 4 + wave) * 64 + lane;
        const int tap = d / (2 * BM), half = (d / BM) & 1, row = d % BM;
        avo[i] = d < A_UNITS ? (unsigned)(((tap * p.Mpad + m0 + row) * 2 + half) * 16) : NC8_OOB;
    }
    const unsigned a_chunk_bytes = (unsigned)(NTAPS * p.Mpad * 32);
    // ---- patch DMA: wave w fetches rows j = 3 * (w & 1) .. + 2 of half plane w >> 1; unit u = j * 64 + lane -> patch pixel
    const unsigned long xaddr = (unsigned long)p.X;
    const u32x4 xrs = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const int phalf = wave >> 1, pj0 = PPW * (wave & 1);
    const unsigned plane_bytes = (unsigned)(p.Hi * p.Wi * 16);
    unsigned pvo[S2 ? 4 : 1][PPW];                              // [parity][row]
#pragma unroll
    for (int par = 0; par < (S2 ? 4 : 1); ++par)
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const int u = (pj0 + r) * 64 + lane;
            const int row = u / PW, col = u % PW;
            int iy = S2 ? 2 * (oy0 + row) - (par >> 1) : oy0 + p.iy0 + row;
            int ix = S2 ? 2 * (ox0 + col) - (par & 1) : ox0 + p.ix0 + col;
            if (p.reflect) {
                iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
                ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
            }
            const bool ok = u < NPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            pvo[par][r] = ok ? (unsigned)(iy * p.Wi + ix) * 16u : NC8_OOB;      // inside the plane; the plane rides in the scalar offset
        }
    // Every iteration issues exactly NDMA instructions per wave, so the counted waits below are constants: a chunk past the
    // split's end is "fetched" through zero-record descriptors (no memory traffic; zeros land in a buffer nobody reads again).
    auto issue_dma = [&](int chunk, int buf, bool live, auto PARC) {
        constexpr int parc = decltype(PARC)::value;        // S2: parity of `chunk` (compile time: the chunk loop is unrolled by 4)
        const unsigned base = lds0 + (unsigned)(buf * BUF * 16);
        const int c2d = S2 ? chunk >> 2 : chunk;
        int kt = S2 ? 0 : c2d / p.nch2d;                   // time tap (0 for 2-D layers: nch2d = all chunks); pair index with ptab
        const int cc = c2d - kt * p.nch2d;
        const int cb = cc * 2 + phalf;
        int ti = t_img + kt + p.t0;
        if (p.ptab) {                                      // (pair j) -> (dY frame, time tap): wave-uniform selects over SGPRs
            ti = kt == 0 ? pr_to[0] : (kt == 1 ? pr_to[1] : (kt == 2 ? pr_to[2] : (kt == 3 ? pr_to[3] : pr_to[4])));
            kt = kt == 0 ? pr_kt[0] : (kt == 1 ? pr_kt[1] : (kt == 2 ? pr_kt[2] : (kt == 3 ? pr_kt[3] : pr_kt[4])));
        }
        if (p.treflect) { ti = ti < 0 ? -ti : ti; ti = ti >= p.T ? 2 * p.T - 2 - ti : ti; }
        const bool tok = (unsigned)ti < (unsigned)p.T;     // zero padding in time: the whole chunk reads zeros
        const int achunk = S2 ? chunk : kt * p.nch2d + cc; // weight chunk: [time tap][16 channels]
        const int asoff = live ? (int)((unsigned)achunk * a_chunk_bytes) : 0;
        u32x4 ark = ars;
        ark[2] = live ? p.a_bytes : 0u;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const unsigned dst = base + (unsigned)((i * 4 + wave) * 64 * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(avo[i]), "s"(ark), "s"(asoff) : "memory");
        }
        // a channel block past the tensor's last one (odd block count, last chunk): zero records instead of the next image
        u32x4 rsk = xrs;
        rsk[2] = (live && cb < p.CB && tok) ? p.x_bytes : 0u;
        const int psoff = (live && tok) ? (int)((unsigned)((n_smp * p.CB + cb) * p.T + ti) * plane_bytes) : 0;
#pragma unroll
        for (int r = 0; r < PPW; ++r) {
            const unsigned dst = base + (unsigned)((A_PAD + phalf * PPL + (pj0 + r) * 64) * 16);
            const unsigned vo = pvo[S2 ? parc : 0][r];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(vo), "s"(rsk), "s"(psoff) : "memory");
        }
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
    for (int j = 0; j < NI; ++j) pbase[j] = A_PAD + (lane >> 5) * PPL + (wave * NI + j) * PW + (lane & 31);
    const int abase = (lane >> 5) * BM + (lane & 31);
    int toff[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) toff[t] = S2 ? (t >> 1) * PW + (t & 1) : p.pty[t / 3] * PW + p.ptx[t % 3];

    const int chunk_beg = blk.z * p.chunks_per_split;
    int chunk_end = chunk_beg + p.chunks_per_split; chunk_end = chunk_end < p.nchunks ? chunk_end : p.nchunks;
    if (p.ptab) chunk_end = pr_n * p.nch2d;                // (no split-K with a pair table: the chunk list depends on the frame)
    // (the table loads above are ordinary VMEM the compiler waits for before their first use -- before any DMA is in flight)

    struct Frag { bf16x8 a[MI], b[NI]; };
    auto read_frag = [&](int buf, int tap, Frag& f) {
        const uint4* __restrict__ s = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < MI; ++i) f.a[i] = __builtin_bit_cast(bf16x8, s[tap * 2 * BM + i * 32 + abase]);
#pragma unroll
        for (int j = 0; j < NI; ++j) f.b[j] = __builtin_bit_cast(bf16x8, s[pbase[j] + toff[tap]]);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
    };

    static_assert(!S2 || NBUF == 4, "S2: buffer = parity = chunk & 3");
    {
        typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;
        issue_dma(chunk_beg, 0, chunk_beg < chunk_end, I0{});
        if (NBUF > 2) issue_dma(chunk_beg + 1, 1, chunk_beg + 1 < chunk_end, I1{});
        if (NBUF > 3) issue_dma(chunk_beg + 2, 2, chunk_beg + 2 < chunk_end, I2{});
    }
    // one chunk from buffer `cur`; PAR = its parity (S2; then cur == PAR), fetching chunk + NBUF - 1 of parity (PAR + 3) & 3
    auto chunk_body = [&](int chunk, int cur, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        // The DMAs of the NBUF - 2 chunks after this one may stay in flight (vmcnt retires in order); after the barrier this
        // chunk's image is complete in EVERY wave's share, and everybody has left the buffer read during the previous chunk,
        // which the DMA of chunk + NBUF - 1 may now overwrite.
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int nb = cur == 0 ? NBUF - 1 : cur - 1;
            issue_dma(chunk + NBUF - 1, nb, chunk + NBUF - 1 < chunk_end, std::integral_constant<int, (par + 3) & 3>{});
        }
        // Fragments are read TWO taps ahead of their MFMAs (three register sets) and every read is pinned behind one MFMA of the
        // running tap (sched_group_barrier): left to itself the scheduler sank the reads to just in front of their use and every
        // tap's first MFMA waited out the full LDS latency (s_waitcnt lgkmcnt in front of it, visible in the ISA).
        Frag f[3];
        read_frag(cur, 0, f[0]);
        read_frag(cur, 1, f[1]);
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            if (tap + 2 < NTAPS) read_frag(cur, tap + 2, f[(tap + 2) % 3]);
            mma(f[tap % 3]);
            if (tap + 2 < NTAPS) {
#pragma unroll
                for (int g_ = 0; g_ < (MI * NI < MI + NI ? MI * NI : MI + NI); ++g_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if (MI * NI > MI + NI) __builtin_amdgcn_sched_group_barrier(0x008, MI * NI - (MI + NI), 0);
                if (MI + NI > MI * NI) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI - MI * NI, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (S2) {                                              // (chunk count and chunk_beg are multiples of 4: parity = unroll position)
        for (int chunk = chunk_beg; chunk < chunk_end; chunk += 4) {
            chunk_body(chunk, 0, std::integral_constant<int, 0>{});
            chunk_body(chunk + 1, 1, std::integral_constant<int, 1>{});
            chunk_body(chunk + 2, 2, std::integral_constant<int, 2>{});
            chunk_body(chunk + 3, 3, std::integral_constant<int, 3>{});
        }
    } else {
        int cur = 0;
        for (int chunk = chunk_beg; chunk < chunk_end; ++chunk) {
            chunk_body(chunk, cur, std::integral_constant<int, 0>{});
            cur = cur + 1 == NBUF ? 0 : cur + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-record DMAs of the tail still write LDS
    __syncthreads();                                       // the epilogue reuses the buffers as its staging tile

    // ---- epilogue (the mappings of conv_patch3x3_bf16_kernel)
    const bool direct = blk.nz == 1;
    const bool yh = p.yh && direct;
    float* __restrict__ Yb = p.Y + (long)blk.z * p.slab_stride;
    float* __restrict__ T = reinterpret_cast<float*>(&smem[0]) + wave * (CG * 64);
    const bool vec_ok = !p.Y2 && (p.Wo & 7) == 0 && (p.out_off & 7) == 0 && (p.out_sc & 7) == 0 && (p.out_sn & 7) == 0 && (p.out_st & 7) == 0 &&
                        (p.out_sh & 7) == 0 && (p.slab_stride & 7) == 0 && ((uintptr_t)p.Y & 15) == 0;
    if (vec_ok) {
#pragma unroll
        for (int h = 0; h < BM / CG; ++h) {
            float bv[CG / 32][16];
#pragma unroll
            for (int q = 0; q < CG / 32; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + h * CG + q * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    bv[q][r] = (direct && p.bias && row < p.M) ? p.bias[row] : 0.f;
                }
#pragma unroll
            for (int jp = 0; jp < NI / 2; ++jp) {          // two pixel rows (64 pixels) of this wave per staging pass
#pragma unroll
                for (int q = 0; q < CG / 32; ++q)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int cl = q * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            float v = acc[h * (CG / 32) + q][jp * 2 + jj][r] + bv[q][r];
                            if (direct) v = c2m_act(v, p.act, p.slope);
                            T[cl * 64 + jj * 32 + (lane & 31)] = v;
                        }
                // (each wave reads back only what it wrote, and a wave's LDS operations execute in order: no barrier)
#pragma unroll
                for (int it = 0; it < CG / 8; ++it) {
                    const int cl = it * 8 + (lane >> 3), px = (lane & 7) * 8;
                    const float4 v0 = *reinterpret_cast<const float4*>(&T[cl * 64 + px]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&T[cl * 64 + px + 4]);
                    const int row = m0 + h * CG + cl;
                    const int oy = oy0 + wave * NI + jp * 2 + (px >> 5), ox = ox0 + (px & 31);
                    if (row < p.M && oy < p.Ho && ox < p.Wo) {
                        const long e = p.out_off + (long)n_smp * p.out_sn + (long)t_img * p.out_st + (long)row * p.out_sc + (long)oy * p.out_sh + ox;
                        if (yh) {
                            const bf16x8 o = {(bf16_t)v0.x, (bf16_t)v0.y, (bf16_t)v0.z, (bf16_t)v0.w,
                                              (bf16_t)v1.x, (bf16_t)v1.y, (bf16_t)v1.z, (bf16_t)v1.w};
                            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.Y) + e));
                        } else {
                            const f32x4 a = {v0.x, v0.y, v0.z, v0.w}, b = {v1.x, v1.y, v1.z, v1.w};
                            __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(Yb + e));
                            __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(Yb + e + 4));
                        }
                    }
                }
            }
        }
        return;
    }
    if (!p.Y2 || p.T == 1) {
        // launches that cannot take the 16-byte stores above -- two-target launches (reflect data gradient over the padded
        // domain), output widths off the 8-pixel grid (the spatially padded target of a 3x3x3 reflect data gradient) -- use the
        // buffer-store epilogue of the gather kernels (conv_store.h): one pass per target with the other target's pixels out of
        // range, no per-element address arithmetic.  The generic form below cost a third of the launch on the full-resolution layers (32 -> 32 at 128x256: data
        // gradient 148 us against 81 us for the forward of the same shape).
        const int yes = yh ? 2 : 4;
        unsigned voff[NI], voff2[NI];
        bool ring = false;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int oy = oy0 + wave * NI + j, ox = ox0 + (lane & 31);
            const bool valid = oy < p.Ho && ox < p.Wo;
            const long e = p.out_off + (long)n_smp * p.out_sn + (long)t_img * p.out_st + (long)oy * p.out_sh + ox + 4L * (lane >> 5) * p.out_sc;
            voff[j] = valid ? (unsigned)(e * yes) : 0x80000000u;
            voff2[j] = 0x80000000u;
            const int yp = oy * p.ps_y + p.po_y - p.lo_y, xp = ox * p.ps_x + p.po_x - p.lo_x;
            if (p.Y2 && valid && (unsigned)yp < (unsigned)p.ext_y && (unsigned)xp < (unsigned)p.ext_x) {
                const long e2 = (long)n_img * p.y2_sn + (long)yp * p.y2_sh + xp + 4L * (lane >> 5) * p.y2_sc;
                voff2[j] = (unsigned)(e2 * yes);
                voff[j] = 0x80000000u;
            }
            ring = ring || voff[j] != 0x80000000u;
        }
        if (m0 + BM <= p.M) {
            if (p.Y2) c2m_store_tile_fast<MI, NI>(acc, p.Y2, voff2, m0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh);
            if (!p.Y2 || __any(ring)) c2m_store_tile_fast<MI, NI>(acc, Yb, voff, m0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh);
        } else {                                           // last row tile hangs over M: rows masked per lane
            if (p.Y2) c2m_store_tile_fast<MI, NI, true>(acc, p.Y2, voff2, m0, p.y2_sc, nullptr, false, 0, 0.f, lane, yh, p.M);
            if (!p.Y2 || __any(ring))
                c2m_store_tile_fast<MI, NI, true>(acc, Yb, voff, m0, p.out_sc, p.bias, direct, p.act, p.slope, lane, yh, p.M);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int oy = oy0 + wave * NI + j, ox = ox0 + (lane & 31);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        float* ybase = Yb;                                 // element type: float, or bf16_t when yh
        long yidx = p.out_off + (long)n_smp * p.out_sn + (long)t_img * p.out_st + (long)oy * p.out_sh + ox;
        long row_stride = p.out_sc;
        if (p.Y2) {
            const int yp = oy * p.ps_y + p.po_y - p.lo_y, xp = ox * p.ps_x + p.po_x - p.lo_x;
            if ((unsigned)yp < (unsigned)p.ext_y && (unsigned)xp < (unsigned)p.ext_x) {
                ybase = p.Y2;
                yidx = (long)n_img * p.y2_sn + (long)yp * p.y2_sh + xp;
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

static int nc8_launch_patch(Nc8P& p, int splits, int v, hipStream_t s);

// geom[]: the indices of c2m_conv_igemm's LDS-patch path (include/c2m_hip.h) -- M, nk = 9 * chunks, lda = padded rows of the
// weight image, Npix, Ho, Wo, Hi, Wi, output strides, reflect, splits, slab stride, cin, the two-target block [36..51], the patch
// origin / tap order [53..60], yh [91].  X is the NC8 form of the [N][cin][Hi][Wi] bf16 tensor.  2-D layers only.
C2M_API int c2m_conv_patch_nc8(const void* A, const void* X, void* Y, void* Y_interior, const float* bias, const int64_t* g,
                               int act, float slope, void* stream) {
    C2M_ENTER();
    Nc8P p;
    p.A = A; p.X = X; p.Y = (float*)Y; p.Y2 = (float*)Y_interior; p.bias = bias;
    p.M = (int)g[C2M_G_M]; p.Mpad = (int)g[C2M_G_LDA];
    p.Ho = (int)g[C2M_G_HO]; p.Wo = (int)g[C2M_G_WO]; p.Hi = (int)g[C2M_G_HI]; p.Wi = (int)g[C2M_G_WI];
    const int nk = (int)g[C2M_G_NK], cin = (int)g[C2M_G_CIN], splits = (int)g[C2M_G_SPLITS];
    if (p.M <= 0 || g[C2M_G_NPIX] <= 0) return 0;
    if (g[C2M_G_TO] != 1 || g[C2M_G_TI] != 1 || g[C2M_G_IS3D] != 0 || g[C2M_G_ST] != 1 || g[C2M_G_SH] != 1 || g[C2M_G_SW] != 1 || !g[C2M_G_PATCH] || g[C2M_G_OUT_SW] != 1)
        return (int)hipErrorInvalidValue;                  // 2-D, stride 1, unit pixel stride of the output
    p.nchunks = nk / 9;
    p.CB = (cin + 7) / 8;
    if (p.nchunks * 9 != nk || cin <= 0 || cin > p.nchunks * 16 || p.Mpad % 128 != 0 || p.Mpad < p.M || splits < 1 ||
        (((uintptr_t)A | (uintptr_t)X) & 15))
        return (int)hipErrorInvalidValue;
    p.Nimg = (int)(g[C2M_G_NPIX] / ((long)p.Ho * p.Wo));
    if ((long)p.Nimg * p.Ho * p.Wo != g[C2M_G_NPIX]) return (int)hipErrorInvalidValue;
    const long xb = (long)p.Nimg * p.CB * p.Hi * p.Wi * 16, ab = (long)p.nchunks * 9 * p.Mpad * 32;
    if (xb >= 0x80000000LL || ab >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)xb; p.a_bytes = (unsigned)ab;
    p.out_sn = g[C2M_G_OUT_SN]; p.out_sc = g[C2M_G_OUT_SC]; p.out_sh = g[C2M_G_OUT_SH]; p.out_off = g[C2M_G_OUT_OFF];
    p.reflect = (int)g[C2M_G_REFLECT]; p.slab_stride = g[C2M_G_SLAB_STRIDE];
    p.act = act; p.slope = slope; p.yh = (int)g[C2M_G_Y_TYPE];
    if (p.Y2 && splits != 1) return (int)hipErrorInvalidValue;
    p.ps_y = (int)g[C2M_G_PS_Y]; p.ps_x = (int)g[C2M_G_PS_X]; p.po_y = (int)g[C2M_G_PO_Y]; p.po_x = (int)g[C2M_G_PO_X]; p.lo_y = (int)g[C2M_G_LO_Y]; p.lo_x = (int)g[C2M_G_LO_X];
    p.ext_y = (int)g[C2M_G_EXT_Y]; p.ext_x = (int)g[C2M_G_EXT_X]; p.y2_sn = g[C2M_G_Y2_SN]; p.y2_sc = g[C2M_G_Y2_SC]; p.y2_sh = g[C2M_G_Y2_SH];
    p.iy0 = (int)g[C2M_G_PATCH_IY0]; p.ix0 = (int)g[C2M_G_PATCH_IX0];
    for (int i = 0; i < 3; ++i) { p.pty[i] = (int)g[C2M_G_PATCH_TY + i]; p.ptx[i] = (int)g[C2M_G_PATCH_TX + i]; }
    p.chunks_per_split = c2m_cdiv(p.nchunks, splits);
    if (c2m_cdiv(p.nchunks, p.chunks_per_split) != splits) return (int)hipErrorInvalidValue;
    p.T = 1; p.nch2d = p.nchunks; p.t0 = 0; p.treflect = 0; p.out_st = 0; p.ptab = nullptr;
    return nc8_launch_patch(p, splits, (int)g[C2M_G_NC8_VARIANT], (hipStream_t)stream);
}

static int nc8_launch_patch(Nc8P& p, int splits, int v, hipStream_t s) {
    // tile / buffering variant (tuning): 0 = rule below; (output rows, LDS buffers, workgroups per CU, tile rows):
    // 1 (64, 2, 2, 8)   2 (64, 2, 2, 16)   3 (128, 3, 1, 8)   4 (32, 3, 2, 8)   5 (32, 2, 2, 16)   6 (64, 3, 1, 16)
    if (v == 0) {
        // 16-row tiles (half the weight traffic per MFMA, twice the work between barriers) where they still give two full rounds
        // of workgroups (512 resident: 256 CUs x 2); measured per shape on one box (tools/ab_nc8.py): 938-986 vs 883-910 TF/s on
        // 256 -> 256 at 64x128, 629-662 vs 596-632 on 128 -> 128, but 811 vs 902 on the 320-workgroup 512 -> 512 layer at 16x32
        const int BMv = p.M <= 32 ? 32 : 64;
        const long wg16 = (long)p.Nimg * ((p.Ho + 15) / 16) * ((p.Wo + 31) / 32) * c2m_cdiv(p.M, BMv) * splits;
        const bool big = p.Ho >= 16 && wg16 >= 1024;
        v = p.M <= 32 ? (big ? 5 : 4) : (big ? 2 : 1);
    }
#define NC8_LAUNCH(BM, NB, WG, TRW) do {                                                                                 \
        const long tiles = (long)p.Nimg * ((p.Ho + TRW - 1) / TRW) * ((p.Wo + 31) / 32);                                 \
        dim3 grid((unsigned)(tiles * c2m_cdiv(p.M, BM) * splits));                                                       \
        hipLaunchKernelGGL((conv_patch_nc8_kernel<BM, NB, WG, TRW>), grid, dim3(256), 0, s, p); } while (0)
    switch (v) {
        case 1: NC8_LAUNCH(64, 2, 2, 8); break;
        case 2: NC8_LAUNCH(64, 2, 2, 16); break;
        case 3: NC8_LAUNCH(128, 3, 1, 8); break;
        case 4: NC8_LAUNCH(32, 3, 2, 8); break;
        case 5: NC8_LAUNCH(32, 2, 2, 16); break;
        case 6: NC8_LAUNCH(64, 3, 1, 16); break;
        default: return (int)hipErrorInvalidValue;
    }
#undef NC8_LAUNCH
    return (int)hipGetLastError();
}

// 4x4 stride-2 pad-1 2-D convolution (zeros / reflect) on the parity-plane form of the patch kernel: X NC8 of [N][C][Hi][Wi]
// (Hi, Wi even), A = c2m_pack_weights_bf16_patch with g[4] = 2 (c2m_pack_weights_bf16_s2_bytes), Y contiguous NCHW [N][M][Hi/2][Wi/2]
// bf16 (yh = 1) or fp32, bias + activation fused.
C2M_API int c2m_conv_s2_nc8(const void* A, const void* X, void* Y, const float* bias, int M, int C, long N, int Hi, int Wi,
                            int reflect, int yh, int act, float slope, void* stream) {
    C2M_ENTER();
    if (M <= 0 || C <= 0 || N <= 0) return 0;
    if ((Hi & 1) || (Wi & 1) || Hi < 4 || Wi < 4 || (((uintptr_t)A | (uintptr_t)X | (uintptr_t)Y) & 15)) return (int)hipErrorInvalidValue;
    Nc8P p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.X = X; p.Y = (float*)Y; p.Y2 = nullptr; p.bias = bias;
    p.M = M; p.Mpad = c2m_cdiv(M, 128) * 128;
    p.Hi = Hi; p.Wi = Wi; p.Ho = Hi / 2; p.Wo = Wi / 2; p.Nimg = (int)N;
    p.CB = (C + 7) / 8;
    p.nchunks = c2m_cdiv(C, 16) * 4;
    const long xb = N * p.CB * (long)Hi * Wi * 16, ab = (long)p.nchunks * 4 * p.Mpad * 32;
    if (xb >= 0x80000000LL || ab >= 0x80000000LL || N * (long)M * p.Ho * p.Wo * 4 >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)xb; p.a_bytes = (unsigned)ab;
    p.out_sn = (long)M * p.Ho * p.Wo; p.out_sc = (long)p.Ho * p.Wo; p.out_sh = p.Wo; p.out_off = 0;
    p.reflect = reflect; p.slab_stride = 0; p.act = act; p.slope = slope; p.yh = yh;
    p.chunks_per_split = p.nchunks;
    p.T = 1; p.nch2d = p.nchunks; p.t0 = 0; p.treflect = 0; p.out_st = 0; p.ptab = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const long tiles = N * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32);
    if (M <= 32) {
        dim3 grid((unsigned)(tiles * c2m_cdiv(M, 32)));
        hipLaunchKernelGGL((conv_patch_nc8_kernel<32, 4, 2, 8, true>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid((unsigned)(tiles * c2m_cdiv(M, 64)));
        hipLaunchKernelGGL((conv_patch_nc8_kernel<64, 4, 2, 8, true>), grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

// 3x3x3 stride-1 pad-1 convolution (zeros / reflect in all three dimensions) on [N][C][T][H][W]: X NC8 = [N][ceil(C/8)][T][H][W][8], A =
// three c2m_pack_weights_bf16_patch images back to back (time tap kt: w + 9 kt with strides s_m = 27 C, s_c = 27), Y contiguous NCTHW.
C2M_API int c2m_conv3d_nc8(const void* A, const void* X, void* Y, const float* bias, int M, int C, long N, int T, int H, int W,
                           int reflect, int yh, int act, float slope, void* stream) {
    C2M_ENTER();
    if (M <= 0 || C <= 0 || N <= 0 || T <= 0) return 0;
    if ((((uintptr_t)A | (uintptr_t)X | (uintptr_t)Y) & 15) || (reflect && (T < 2 || H < 2 || W < 2))) return (int)hipErrorInvalidValue;
    Nc8P p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.X = X; p.Y = (float*)Y; p.Y2 = nullptr; p.bias = bias;
    p.M = M; p.Mpad = c2m_cdiv(M, 128) * 128;
    p.Hi = H; p.Wi = W; p.Ho = H; p.Wo = W; p.Nimg = (int)(N * T);
    p.CB = (C + 7) / 8;
    p.nch2d = c2m_cdiv(C, 16); p.nchunks = 3 * p.nch2d; p.T = T; p.t0 = -1; p.treflect = reflect; p.ptab = nullptr;
    const long xb = N * p.CB * (long)T * H * W * 16, ab = (long)p.nchunks * 9 * p.Mpad * 32;
    if (xb >= 0x80000000LL || ab >= 0x80000000LL || N * (long)M * T * H * W * 4 >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)xb; p.a_bytes = (unsigned)ab;
    p.out_sn = (long)M * T * H * W; p.out_sc = (long)T * H * W; p.out_st = (long)H * W; p.out_sh = W; p.out_off = 0;
    p.reflect = reflect; p.act = act; p.slope = slope; p.yh = yh;
    p.iy0 = -1; p.ix0 = -1;
    for (int i = 0; i < 3; ++i) { p.pty[i] = i; p.ptx[i] = i; }
    p.chunks_per_split = p.nchunks;
    return nc8_launch_patch(p, 1, 0, (hipStream_t)stream);
}

// Data gradient of such a layer with REFLECT padding: dY NC8 [N][ceil(K/8)][T][H][W][8] -> the spatially padded gradient
// T_ [N][M][T][H+2][W+2] (bf16 / fp32), launched over the T real frames: frame t sums its (dY frame, time tap) pairs from `ptab`
// (device int32 [T][11]: {npairs, (frame, kt) x 5}; the pad frames folded onto the frames they mirror), taps rotated in the plane
// (patch origin -2, tap order 2,1,0).  A = three pack images (kt: w + 9 kt, rows = the first M input channels: s_m = 27, s_c = 27 Ctot;
// the target has Ctot >= M channels -- a concatenated input whose tail carries no gradient).
// c2m_reflect_fold(pt = 0, ph = pw = 1) finishes.  Zero padding: the table lists the in-range (t + 1 - kt, kt) pairs, target
// [N][M][T][H][W] (no fold).
C2M_API int c2m_conv3d_dgrad_nc8(const void* A, const void* dY, void* Tgt, const int* ptab, int M, int Ctot, int K, long N, int T, int H,
                                 int W, int reflect, int th, void* stream) {
    C2M_ENTER();
    if (M <= 0 || K <= 0 || N <= 0 || T <= 0) return 0;
    if ((((uintptr_t)A | (uintptr_t)dY | (uintptr_t)Tgt) & 15) || !ptab || Ctot < M) return (int)hipErrorInvalidValue;
    Nc8P p;
    memset(&p, 0, sizeof(p));
    const int pad = reflect ? 1 : 0;
    p.A = A; p.X = dY; p.Y = (float*)Tgt; p.Y2 = nullptr; p.bias = nullptr;
    p.M = M; p.Mpad = c2m_cdiv(M, 128) * 128;
    p.Hi = H; p.Wi = W; p.Ho = H + 2 * pad; p.Wo = W + 2 * pad; p.Nimg = (int)(N * T);
    p.CB = (K + 7) / 8;
    p.nch2d = c2m_cdiv(K, 16); p.nchunks = 3 * p.nch2d; p.T = T; p.treflect = 0;
    p.ptab = ptab; p.t0 = 0;                               // (zeros padding: the table holds the in-range (t + 1 - kt, kt) pairs)
    const long xb = N * p.CB * (long)T * H * W * 16, ab = (long)p.nchunks * 9 * p.Mpad * 32;
    if (xb >= 0x80000000LL || ab >= 0x80000000LL || N * (long)Ctot * T * p.Ho * p.Wo * 4 >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)xb; p.a_bytes = (unsigned)ab;
    p.out_sn = (long)Ctot * T * p.Ho * p.Wo; p.out_sc = (long)T * p.Ho * p.Wo; p.out_st = (long)p.Ho * p.Wo; p.out_sh = p.Wo; p.out_off = 0;
    p.reflect = 0; p.act = 0; p.slope = 0.f; p.yh = th;
    p.iy0 = -1 - pad; p.ix0 = -1 - pad;
    for (int i = 0; i < 3; ++i) { p.pty[i] = 2 - i; p.ptx[i] = 2 - i; }
    p.chunks_per_split = p.nchunks;
    return nc8_launch_patch(p, 1, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ weight gradient on NC8 operands
// dW[m][c][ky][kx] = sum_{n,y,x} dY[n][m][y][x] * X[n][c][y + ky - 1][x + kx - 1]   (3x3, stride 1, pad 1 zeros / reflect)
// as 9 GEMMs D_tap[m][c] over K = pixels on v_mfma_f32_32x32x16_bf16.  Both operands need 8 consecutive PIXELS of one channel
// per lane; NC8 stores 8 channels of one pixel per 16-byte unit -- the transposed form -- so the LDS images are plain copies of
// NC8 rows ([channel block][pixel], 16-byte LDS-DMA like the forward kernel) and the fragments come out of them with
// ds_read_b64_tr_b16 (per 16 lanes: 4 pixels x 16 channels delivered channel-per-lane).  A tap shift moves along the PIXEL axis
// (units), never inside a unit, so every transposed read stays 8-byte aligned for every tap -- on NCHW the +-1 pixel taps are what
// broke that alignment, and the NCHW kernel (conv_wgrad_wide_bf16_kernel) spends 15.8 VALU per MFMA on
// shifting pixel groups in registers.
// Workgroup = 64 dY channels x 32 X channels x 9 taps over a range of 4 x 32-pixel chunks; wave (mh, kh) owns 32 dY channels and
// the K-steps of two of the chunk's four rows for ALL taps (9 accumulator tiles).  The two kh waves write separate slabs (no
// cross-wave sum inside the kernel), deterministic reduction afterwards (wgrad_nc8_reduce_kernel).  Plane strides 132 / 204 units
// keep the four channel blocks of a transposed read on disjoint banks.
struct WgNc8P {
    const void* dY; const void* X;
    float* slab;          // [S][9][Mp][Cp]
    float* dbslab;        // [S][Mp]
    int M, C, Mp, Cp, CBy, CBx;
    int Nimg, H, W, reflect;        // H, W: the dY map
    int Hin, Win;                   // the X map (= H, W; 2H, 2W for the stride-2 layers)
    int T, kt, treflect;            // 3x3x3 layers: images are (sample, frame) pairs of T frames, X frame = t + kt - 1 (one launch per kt)
    int chunks_y, chunks_x, nchunks, chunks_per_split;
    unsigned dy_bytes, x_bytes;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 nc8_tr_frag(const unsigned lds_byte, const int off0) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unsigned long)(lds_byte + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(unsigned long)(lds_byte + off0 + 64));      // + 4 pixels
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// S2 = true: the 4x4 stride-2 pad-1 layers.  dW[m][c][ky][kx] = sum dY[n][m][y][x] * X[n][c][2y + ky - 1][2x + kx - 1]: the taps of one
// input-row parity py (ky = 2a + 1 - py) are 2 x 2 x 2 = 8 (column parity px, a, b) -- one workgroup takes one py (8 accumulator tiles),
// its X image holds the two column-parity planes of that row parity ([px][channel block][5 x 33 positions], plane stride 180 units,
// fetched with stride-2 addresses by the DMA), and tap (px, a, b) of pixel (r, c) is the unit (r + a, c + b) of plane px.
template <bool S2>
__global__ __launch_bounds__(256, 2) void conv_wgrad_nc8_kernel(const WgNc8P p) {
    constexpr int DPL = 132;                                      // dY plane stride (units)
    constexpr int XPL = S2 ? 180 : 204, PWX = S2 ? 33 : 34, XVAL = S2 ? 165 : 204, NPAR = S2 ? 2 : 1;
    constexpr int NXU = NPAR * 4 * XPL;                           // X image units: 816 / 1440
    constexpr int NXI = (NXU + 63) / 64, XPW = (NXI + 3) / 4;     // X DMA rows: 13 / 23; per wave 4 / 6
    constexpr int NT = S2 ? 8 : 9, NH = (NT + 1) / 2;
    constexpr int SD = 8 * DPL, SX = NXI * 64;                    // dY image, X image
    constexpr int BUF = SD + SX;
    constexpr int XCH = 2 * NH * 16 * 64 + 2 * 16 * 64;           // floats of the kh exchange
    static_assert(XCH * 4 <= 2 * BUF * 16, "exchange fits the images");
    __shared__ uint4 smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mh = wave & 1, kh = wave >> 1;
    const C2mBlock blk = c2m_xcd_block((unsigned)(p.Cp / 32) * (S2 ? 2u : 1u), (unsigned)(p.Mp / 64), 0);
    const int py = S2 ? (int)(blk.x & 1u) : 0;
    const int m0 = blk.y * 64, c0 = (S2 ? blk.x >> 1 : blk.x) * 32, split = blk.z;
    const int cby0 = m0 >> 3, cbx0 = c0 >> 3;
    const unsigned HW16 = (unsigned)(p.H * p.W) * 16u, XHW16 = (unsigned)(p.Hin * p.Win) * 16u;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) uint4*)&smem[0];

    const unsigned long ya = (unsigned long)p.dY, xa = (unsigned long)p.X;
    const u32x4 yrs = {(unsigned)ya, (unsigned)(ya >> 32) & 0xffffu, p.dy_bytes, 0x00020000u};
    const u32x4 xrs = {(unsigned)xa, (unsigned)(xa >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    // ---- dY DMA rows of this wave: I = wave * 4 + j -> channel block I >> 1, row pair I & 1; lane -> (row, column)
    const int d_r = lane >> 5, d_col = lane & 31;
    // ---- X DMA rows: I = wave + 4 * j < NXI (the waits of this kernel are vmcnt(0): the waves need not issue equal counts);
    // per lane and row one packed word: column parity | channel block << 1 | patch row << 3 | patch column << 6 | valid << 12
    unsigned x_code[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int u = (wave + 4 * j) * 64 + lane;
        const int par = u / (4 * XPL), rem = u % (4 * XPL);
        const int pl = rem / XPL, pos = rem % XPL;
        const bool ok = u < NXU && pos < XVAL && cbx0 + pl < p.CBx;
        x_code[j] = (unsigned)(par | (pl << 1) | ((pos / PWX) << 3) | ((pos % PWX) << 6) | ((ok ? 1 : 0) << 12));
    }
    auto issue_dma = [&](int chunk, int buf, bool live) {
        int t = chunk;
        const int cx = t % p.chunks_x; t /= p.chunks_x;
        const int cy = t % p.chunks_y; const int nimg = t / p.chunks_y;
        const int y0 = cy * 4, x0 = cx * 32;
        const int ns = nimg / p.T, tf = nimg - ns * p.T;            // (sample, frame); T = 1: 2-D layers
        int tx = tf + p.kt - (p.T > 1 ? 1 : 0);
        if (p.treflect) { tx = tx < 0 ? -tx : tx; tx = tx >= p.T ? 2 * p.T - 2 - tx : tx; }
        const bool tok = (unsigned)tx < (unsigned)p.T;
        const unsigned base = lds0 + (unsigned)(buf * BUF * 16);
        // dY: 4 rows of 64 units
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int I = wave * 4 + j, cb = I >> 1, r = (I & 1) * 2 + d_r;
            const bool ok = y0 + r < p.H && x0 + d_col < p.W;
            const unsigned vo = ok ? ((unsigned)((ns * p.CBy + cby0 + cb) * p.T + tf) * HW16 + (unsigned)((y0 + r) * p.W + x0 + d_col) * 16u) : NC8_OOB;
            u32x4 rs = yrs;
            rs[2] = (live && cby0 + cb < p.CBy) ? p.dy_bytes : 0u;
            const unsigned dst = base + (unsigned)((cb * DPL + (I & 1) * 64) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dst), "v"(vo), "s"(rs) : "memory");
        }
#pragma unroll
        for (int j = 0; j < XPW; ++j) {
            const int I = wave + 4 * j;
            if (I >= NXI) continue;                        // (wave-uniform)
            const int xpx = (int)(x_code[j] & 1u), xpl = (int)((x_code[j] >> 1) & 3u);
            const int xpr = (int)((x_code[j] >> 3) & 7u), xpc = (int)((x_code[j] >> 6) & 63u);
            int iy = S2 ? 2 * (y0 + xpr) - py : y0 - 1 + xpr;
            int ix = S2 ? 2 * (x0 + xpc) - xpx : x0 - 1 + xpc;
            if (p.reflect) {
                iy = iy < 0 ? -iy : iy; iy = iy >= p.Hin ? 2 * p.Hin - 2 - iy : iy;
                ix = ix < 0 ? -ix : ix; ix = ix >= p.Win ? 2 * p.Win - 2 - ix : ix;
            }
            const bool ok = (x_code[j] >> 12) != 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const unsigned vo = ok ? ((unsigned)((ns * p.CBx + cbx0 + xpl) * p.T + tx) * XHW16 + (unsigned)(iy * p.Win + ix) * 16u) : NC8_OOB;
            u32x4 rs = xrs;
            rs[2] = (live && tok) ? p.x_bytes : 0u;
            const unsigned dst = base + (unsigned)((SD + I * 64) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(dst), "v"(vo), "s"(rs) : "memory");
        }
    };

    f32x16 acc[NT], accb;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bool want_bias = blk.x == 0;
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    // transposed-read lane addresses: 16-lane group g = lane >> 4 reads channels 16 * (g & 1) ... of pixels 8 * (g >> 1) + q; lane
    // 4q + pp of the group supplies row (pixel) q, 8-byte half pp & 1 of channel block 2 * (g & 1) + (pp >> 1)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const unsigned a_lane = (unsigned)((((mh * 4 + 2 * (g & 1) + (pp >> 1)) * DPL + 8 * (g >> 1) + q + kh * 64) * 16) + 8 * (pp & 1));
    const unsigned b_lane = (unsigned)(((SD + (2 * (g & 1) + (pp >> 1)) * XPL + 8 * (g >> 1) + q + kh * 2 * PWX) * 16) + 8 * (pp & 1));

    const int chunk_beg = split * p.chunks_per_split;
    int chunk_end = chunk_beg + p.chunks_per_split; chunk_end = chunk_end < p.nchunks ? chunk_end : p.nchunks;
    issue_dma(chunk_beg < p.nchunks ? chunk_beg : 0, 0, chunk_beg < chunk_end);
    int cur = 0;
    for (int chunk = chunk_beg; chunk < chunk_end; ++chunk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_dma(chunk + 1 < chunk_end ? chunk + 1 : chunk, cur ^ 1, chunk + 1 < chunk_end);
        const unsigned ab = lds0 + (unsigned)(cur * BUF * 16) + a_lane, bb = lds0 + (unsigned)(cur * BUF * 16) + b_lane;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const int r = ss >> 1, cc = (ss & 1) * 16;
            const bf16x8 A = nc8_tr_frag(ab, (r * 32 + cc) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int off = S2 ? (t >> 2) * 4 * XPL + (r + ((t >> 1) & 1)) * PWX + cc + (t & 1) : (r + t / 3) * PWX + cc + t % 3;
                const bf16x8 B = nc8_tr_frag(bb, off * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[t], 0, 0, 0);
            }
            if (want_bias) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, ones, accb, 0, 0, 0);
        }
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the two pixel halves (kh) of a row tile are summed inside the workgroup (fixed order: kh 0 + kh 1) through the now free
    // LDS images, half of the taps at a time (a tap tile of one wave = 4 KB); halves the slab traffic of the launch
    __syncthreads();
    float* __restrict__ xch = reinterpret_cast<float*>(&smem[0]) + mh * (NH * 16 * 64);
    float* __restrict__ xcb = reinterpret_cast<float*>(&smem[0]) + 2 * NH * 16 * 64 + mh * 16 * 64;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        const int t0 = ph * NH, nt = ph ? NT - NH : NH;
        if (kh == 1) {
#pragma unroll
            for (int t = 0; t < NH; ++t)
                if (t < nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[(t * 16 + r) * 64 + lane] = acc[t0 + t][r];
            if (ph == 0 && want_bias)
#pragma unroll
                for (int r = 0; r < 16; ++r) xcb[r * 64 + lane] = accb[r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int t = 0; t < NH; ++t)
                if (t < nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t0 + t][r] += xch[(t * 16 + r) * 64 + lane];
            if (ph == 0 && want_bias)
#pragma unroll
                for (int r = 0; r < 16; ++r) accb[r] += xcb[r * 64 + lane];
        }
        __syncthreads();
    }
    if (kh != 0) return;
    // ---- slabs: [split][tap][m][c] (9 taps; S2: 16, this workgroup's 8 are ky = 2a + 1 - py, kx = 2b + 1 - px); an accumulator
    // register is 32 consecutive c of one row
    constexpr int NTS = S2 ? 16 : 9;
    float* __restrict__ sb = p.slab + (long)split * NTS * p.Mp * p.Cp;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ts = S2 ? (2 * ((t >> 1) & 1) + 1 - py) * 4 + 2 * (t & 1) + 1 - (t >> 2) : t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            sb[((long)ts * p.Mp + row) * p.Cp + c0 + (lane & 31)] = acc[t][r];
        }
    }
    if (want_bias && (lane & 31) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + mh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            p.dbslab[(long)split * p.Mp + row] = accb[r];
        }
    }
}

// dW[m][c][tap] = sum over slabs; db[m] likewise.  Thread = (tap, m, c) with c fastest (coalesced slab reads); four interleaved
// partial sums keep the loads in flight, combined in a fixed order -> bit-reproducible.
__global__ void wgrad_nc8_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dbslab, float* __restrict__ dW,
                                        float* __restrict__ db, int M, int C, int Mp, int Cp, int nslab, int NT, int out_nt, int out_t0) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const long MC = (long)M * C;
    if (i < NT * MC) {
        const int t = (int)(i / MC);
        const long mc = i % MC;
        const int c = (int)(mc % C), m = (int)(mc / C);
        const float* __restrict__ q = slab + ((long)t * Mp + m) * Cp + c;
        const long st = (long)NT * Mp * Cp;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 3 < nslab; k += 4) {
            s0 += q[(long)k * st]; s1 += q[(long)(k + 1) * st]; s2 += q[(long)(k + 2) * st]; s3 += q[(long)(k + 3) * st];
        }
        for (; k < nslab; ++k) s0 += q[(long)k * st];
        dW[mc * out_nt + out_t0 + t] = (s0 + s1) + (s2 + s3);      // (3x3x3 layers: 27 taps per (m, c), this launch's 9 at kt * 9)
    }
    if (db && i < M) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += dbslab[(long)k * Mp + i];
        db[i] = s;
    }
}

static void wgrad_nc8_shape(int M, int C, long N, int H, int W, int s2, int& Mp, int& Cp, long& nchunks, int& S) {
    Mp = c2m_cdiv(M, 64) * 64; Cp = c2m_cdiv(C, 32) * 32;
    nchunks = N * c2m_cdiv(H, 4) * c2m_cdiv(W, 32);
    const long tiles = (long)(Mp / 64) * (Cp / 32) * (s2 ? 2 : 1);
    static const long target = getenv("C2M_WGRAD_NC8_WGS") ? atol(getenv("C2M_WGRAD_NC8_WGS")) : 768;      // (tuning knob)
    long s = (target + tiles - 1) / tiles;                        // ~1.5 resident rounds of 512 workgroups ...
    // ... but >= 16 chunks (2048 pixels, ~10 us of MFMAs) per split: every workgroup writes a 73 KB slab that is read back
    const long maxs = nchunks / 16 > 0 ? nchunks / 16 : 1;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    const long per = (nchunks + s - 1) / s;
    S = (int)((nchunks + per - 1) / per);
}

// Floats the caller provides: slab = S * taps * Mp * Cp (taps = 9, or 16 for s2), then dbslab = S * Mp (S = c2m_conv_wgrad_nc8_splits).
C2M_API int c2m_conv_wgrad_nc8_splits(int M, int C, long N, int H, int W, int s2) {
    int Mp, Cp, S; long nch;
    wgrad_nc8_shape(M, C, N, H, W, s2, Mp, Cp, nch, S);
    return S;
}
C2M_API long c2m_conv_wgrad_nc8_slab_floats(int M, int C, long N, int H, int W, int s2) {
    int Mp, Cp, S; long nch;
    wgrad_nc8_shape(M, C, N, H, W, s2, Mp, Cp, nch, S);
    return (long)S * ((s2 ? 16L : 9L) * Mp * Cp + Mp);
}

static int wgrad_nc8_launch(const void* dY_nc8, const void* X_nc8, float* slab, float* dW, float* db, int M, int C, long N, int T,
                            int kt, int H, int W, int reflect, int s2, hipStream_t s) {
    WgNc8P p;
    long nch; int S;
    wgrad_nc8_shape(M, C, N * T, H, W, s2, p.Mp, p.Cp, nch, S);
    p.dY = dY_nc8; p.X = X_nc8; p.M = M; p.C = C; p.CBy = (M + 7) / 8; p.CBx = (C + 7) / 8;
    p.Nimg = (int)(N * T); p.H = H; p.W = W; p.reflect = reflect;
    p.T = T; p.kt = kt; p.treflect = reflect;
    p.Hin = s2 ? 2 * H : H; p.Win = s2 ? 2 * W : W;
    p.chunks_y = c2m_cdiv(H, 4); p.chunks_x = c2m_cdiv(W, 32); p.nchunks = (int)nch;
    p.chunks_per_split = c2m_cdiv(nch, S);
    const long yb = N * T * p.CBy * (long)H * W * 16, xb = N * T * p.CBx * (long)p.Hin * p.Win * 16;
    if (yb >= 0x80000000LL || xb >= 0x80000000LL || nch >= 0x7fffffffLL || (((uintptr_t)dY_nc8 | (uintptr_t)X_nc8) & 15) ||
        (reflect && (p.Hin < 2 || p.Win < 2 || (T > 1 && T < 2))))
        return (int)hipErrorInvalidValue;
    p.dy_bytes = (unsigned)yb; p.x_bytes = (unsigned)xb;
    const int NTS = s2 ? 16 : 9;
    p.slab = slab; p.dbslab = slab + (long)S * NTS * p.Mp * p.Cp;
    dim3 grid((unsigned)((p.Mp / 64) * (p.Cp / 32) * (s2 ? 2 : 1) * S));
    if (s2) hipLaunchKernelGGL(conv_wgrad_nc8_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_wgrad_nc8_kernel<false>, grid, dim3(256), 0, s, p);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    const long n = (long)NTS * M * C;
    hipLaunchKernelGGL(wgrad_nc8_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.slab, p.dbslab, dW, db, M, C,
                       p.Mp, p.Cp, S, NTS, T > 1 ? 27 : NTS, T > 1 ? kt * 9 : 0);
    return (int)hipGetLastError();
}

// dY_nc8: [N][ceil(M/8)][H][W][8]; s2 = 0: 3x3 stride-1 pad-1 layer, X_nc8 [N][ceil(C/8)][H][W][8], dW [M][C][3][3]; s2 = 1: 4x4
// stride-2 pad-1 layer, X_nc8 [N][ceil(C/8)][2H][2W][8], dW [M][C][4][4]; db: [M] fp32 or NULL.
C2M_API int c2m_conv_wgrad_nc8(const void* dY_nc8, const void* X_nc8, float* slab, float* dW, float* db, int M, int C, long N,
                               int H, int W, int reflect, int s2, void* stream) {
    C2M_ENTER();
    if (M <= 0 || C <= 0 || N <= 0) return 0;
    return wgrad_nc8_launch(dY_nc8, X_nc8, slab, dW, db, M, C, N, 1, 0, H, W, reflect, s2, (hipStream_t)stream);
}

// 3x3x3 stride-1 pad-1 layers: dY_nc8 [N][ceil(M/8)][T][H][W][8], X_nc8 [N][ceil(C/8)][T][H][W][8], dW [M][C][3][3][3]: one launch per
// time tap over the N * T (sample, frame) images, X frame t + kt - 1 (reflected / zero); slab as for c2m_conv_wgrad_nc8 with N * T images.
C2M_API int c2m_conv_wgrad3d_nc8(const void* dY_nc8, const void* X_nc8, float* slab, float* dW, float* db, int M, int C, long N, int T,
                                 int H, int W, int reflect, void* stream) {
    C2M_ENTER();
    if (M <= 0 || C <= 0 || N <= 0 || T <= 1) return (int)hipErrorInvalidValue;
    for (int kt = 0; kt < 3; ++kt) {
        const int rc = wgrad_nc8_launch(dY_nc8, X_nc8, slab, dW, kt == 1 ? db : nullptr, M, C, N, T, kt, H, W, reflect, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ 4x4 stride-2 data gradient on NC8
// dX of a 4x4 stride-2 pad-1 layer: output position i = 2q + ri takes only the taps ky of one parity -- two per axis -- so each of
// the four output parity classes (ri, rj) is a 2x2 stride-1 correlation of dY.  One workgroup computes ALL FOUR classes of an
// 8 x 32 tile of (q, p): they share one (8+2) x 34 patch of dY (rows q-1 .. q+1), every (class, tap) pair is one "virtual tap" with
// its own weight tile and patch offset, and each class has its own accumulators (32 input channels x 4 classes x 64 positions =
// 128 registers per wave).  Epilogue: the (rj = 0, 1) pair of a class row is two adjacent pixels -> one packed store per lane,
// 128 contiguous bytes per half wave.  zeros padding: the target is dX itself; reflect: the padded (Hi+2) x (Wi+2) gradient,
// folded afterwards by c2m_reflect_fold.
//   zeros   : ri = 0: (ky 1, dY row q), (ky 3, q-1);  ri = 1: (ky 0, q+1), (ky 2, q)
//   reflect : ri = 0: (ky 0, q), (ky 2, q-1);         ri = 1: (ky 1, q), (ky 3, q-1)        (padded coordinate i' = 2q + ri)
struct S2DgP {
    const void* A;       // pack mode 3 / 4: [chunk of 16 dY channels][virtual tap 16][Mpad rows = input channels][2 halves]
    const void* dY;      // NC8 [N][CBy][Ho][Wo][8]
    void* T;             // target [N][M][Ht][Wt] (bf16 or fp32)
    int M, Mpad, nchunks, CBy, Nimg, Ho, Wo, Ht, Wt, reflect, th;
    unsigned dy_bytes, a_bytes;
};

__global__ __launch_bounds__(256, 2) void conv_s2_dgrad_nc8_kernel(const S2DgP p) {
    constexpr int BM = 32, TR = 8, PW = 34, NPIX = (TR + 2) * PW, PPL = 384, NVT = 16;
    constexpr int A_UNITS = NVT * BM * 2, NAI = A_UNITS / 256;    // 1024 units: 4 DMA rows per wave
    constexpr int BUF = A_UNITS + 2 * PPL, NDMA = NAI + 3;
    __shared__ uint4 smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Qh = p.Ht / 2, Qw = p.Wt / 2;
    const int tiles_x = (Qw + 31) / 32, tiles_y = (Qh + TR - 1) / TR;
    const C2mBlock blk = c2m_xcd_block((unsigned)(p.Nimg * tiles_y * tiles_x), (unsigned)(p.M + BM - 1) / BM, 1);
    const int m0 = blk.y * BM;
    int tb = blk.x;
    const int tx = tb % tiles_x; tb /= tiles_x;
    const int ty = tb % tiles_y; const int n_img = tb / tiles_y;
    const int q0 = ty * TR, p0 = tx * 32;
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) uint4*)&smem[0];

    const unsigned long aaddr = (unsigned long)p.A, yaddr = (unsigned long)p.dY;
    const u32x4 ars = {(unsigned)aaddr, (unsigned)(aaddr >> 32) & 0xffffu, p.a_bytes, 0x00020000u};
    const u32x4 yrs = {(unsigned)yaddr, (unsigned)(yaddr >> 32) & 0xffffu, p.dy_bytes, 0x00020000u};
    unsigned avo[NAI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        const int d = (i * 4 + wave) * 64 + lane;                 // destination unit of [vt][half][row]
        const int vt = d / (2 * BM), half = (d / BM) & 1, row = d % BM;
        avo[i] = (unsigned)(((vt * p.Mpad + m0 + row) * 2 + half) * 16);
    }
    const unsigned a_chunk_bytes = (unsigned)(NVT * p.Mpad * 32);
    const int phalf = wave >> 1, pj0 = 3 * (wave & 1);
    const unsigned plane_bytes = (unsigned)(p.Ho * p.Wo * 16);
    unsigned pvo[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int u = (pj0 + r) * 64 + lane;
        const int iy = q0 - 1 + u / PW, ix = p0 - 1 + u % PW;
        const bool ok = u < NPIX && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo;      // dY outside its map: zeros
        pvo[r] = ok ? (unsigned)n_img * (unsigned)p.CBy * plane_bytes + (unsigned)(iy * p.Wo + ix) * 16u : NC8_OOB;
    }
    auto issue_dma = [&](int chunk, int buf, bool live) {
        const unsigned base = lds0 + (unsigned)(buf * BUF * 16);
        const int asoff = live ? (int)((unsigned)chunk * a_chunk_bytes) : 0;
        u32x4 ark = ars;
        ark[2] = live ? p.a_bytes : 0u;
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            const unsigned dst = base + (unsigned)((i * 4 + wave) * 64 * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(avo[i]), "s"(ark), "s"(asoff) : "memory");
        }
        const int cb = chunk * 2 + phalf;
        u32x4 rsk = yrs;
        rsk[2] = (live && cb < p.CBy) ? p.dy_bytes : 0u;
        const int psoff = live ? (int)((unsigned)cb * plane_bytes) : 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned dst = base + (unsigned)((A_UNITS + phalf * PPL + (pj0 + r) * 64) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(pvo[r]), "s"(rsk), "s"(psoff) : "memory");
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    int pbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) pbase[j] = A_UNITS + (lane >> 5) * PPL + (wave * 2 + j + 1) * PW + (lane & 31) + 1;   // dY (q, p)
    const int abase = (lane >> 5) * BM + (lane & 31);
    // patch offset of (class parity r, tap a) along one axis: zeros {0: (0, -1), 1: (+1, 0)}, reflect {(0, -1), (0, -1)}
    const int o10 = p.reflect ? 0 : 1, o11 = p.reflect ? -1 : 0;

    issue_dma(0, 0, p.nchunks > 0);
    int cur = 0;
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_dma(chunk + 1 < p.nchunks ? chunk + 1 : chunk, cur ^ 1, chunk + 1 < p.nchunks);
        const uint4* __restrict__ s = smem + cur * BUF;
#pragma unroll
        for (int vt = 0; vt < NVT; ++vt) {
            const int cls = vt >> 2, a = (vt >> 1) & 1, b = vt & 1, ri = cls >> 1, rj = cls & 1;
            const int oy = ri == 0 ? (a == 0 ? 0 : -1) : (a == 0 ? o10 : o11);
            const int ox = rj == 0 ? (b == 0 ? 0 : -1) : (b == 0 ? o10 : o11);
            const bf16x8 A = __builtin_bit_cast(bf16x8, s[vt * 2 * BM + abase]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 B = __builtin_bit_cast(bf16x8, s[pbase[j] + oy * PW + ox]);
                acc[cls][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[cls][j], 0, 0, 0);
            }
        }
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: T[n][m][2q + ri][2p + rj]; lane = position p, the rj pair is one packed store
    const long plane = (long)p.Ht * p.Wt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + wave * 2 + j, pp = p0 + (lane & 31);
        if (q >= Qh || pp >= Qw) continue;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                const long e = ((long)n_img * p.M + row) * plane + (long)(2 * q + ri) * p.Wt + 2 * pp;
                const float v0 = acc[ri * 2][j][r], v1 = acc[ri * 2 + 1][j][r];
                if (p.th) {
                    typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
                    const bf16x2_t o = {(bf16_t)v0, (bf16_t)v1};
                    *reinterpret_cast<bf16x2_t*>(reinterpret_cast<bf16_t*>(p.T) + e) = o;
                } else {
                    *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.T) + e) = make_float2(v0, v1);
                }
            }
    }
}

// dY_nc8: NC8 of [N][K][Ho][Wo]; A: c2m_pack_weights_bf16_patch with g = {M = input channels, C = K, s_m = 16, s_c = M * 16, mode 3
// (zeros) / 4 (reflect)}; T: [N][M][Ht][Wt] with (Ht, Wt) = (2 Ho, 2 Wo) (zeros: dX itself) or (2 Ho + 2, 2 Wo + 2) (reflect: the padded
// gradient, to be folded); th: 1 = bf16 target.
C2M_API int c2m_conv_s2_dgrad_nc8(const void* A, const void* dY_nc8, void* T, int M, int K, long N, int Ho, int Wo, int reflect,
                                  int th, void* stream) {
    C2M_ENTER();
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    S2DgP p;
    p.A = A; p.dY = dY_nc8; p.T = T; p.M = M; p.Mpad = c2m_cdiv(M, 128) * 128; p.nchunks = c2m_cdiv(K, 16); p.CBy = (K + 7) / 8;
    p.Nimg = (int)N; p.Ho = Ho; p.Wo = Wo; p.reflect = reflect; p.th = th;
    p.Ht = 2 * Ho + (reflect ? 2 : 0); p.Wt = 2 * Wo + (reflect ? 2 : 0);
    const long yb = N * p.CBy * (long)Ho * Wo * 16, ab = (long)p.nchunks * 16 * p.Mpad * 32;
    if (yb >= 0x80000000LL || ab >= 0x80000000LL || (((uintptr_t)A | (uintptr_t)dY_nc8 | (uintptr_t)T) & 15)) return (int)hipErrorInvalidValue;
    p.dy_bytes = (unsigned)yb; p.a_bytes = (unsigned)ab;
    const long tiles = N * c2m_cdiv(p.Ht / 2, 8) * c2m_cdiv(p.Wt / 2, 32);
    hipLaunchKernelGGL(conv_s2_dgrad_nc8_kernel, dim3((unsigned)(tiles * c2m_cdiv(M, 32))), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
