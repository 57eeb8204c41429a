// conv_wino.hip -- Winograd F(2x2, 3x3) convolution for the 3x3 stride-1 layers (VGG-19, SPADE MLPs, generator and
// decoder blocks: layers/vgg.py:92-137, spade_block.py:47-49, residual_block.py:13-31,42-71, up_block.py:9-13 and their
// data gradients), fp32 on v_mfma_f32_32x32x2_f32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input tile, g: 3x3 filter, Y: 2x2 outputs
// 16 multiplies per 4 outputs instead of 36: the contraction over input channels becomes 16 independent GEMMs
// M_xi[cout][tile] = sum_c U_xi[cout][c] * V_xi[c][tile] with 2.25x fewer MFMA FLOPs than the direct form.  The input
// transform uses only +-1 coefficients, the filter transform 1/2 and 1/4 (exact scalings), so the result differs from
// the direct fp32 convolution by a few ulp of the partial sums (tests: 2e-5 of the tensor scale, like the direct kernels).
//
// One workgroup = 64 output channels x an 8x16 output region (32 Winograd tiles) of one image; 4 waves, wave w owns the
// transformed row i = w (xi = 4w .. 4w+3) for all 64 x 32 outputs: 4 xi x 2 row-tiles x 16 = 128 accumulator registers.
// Per 8-channel chunk: the (8 x 10 x 18) input patch arrives by LDS-DMA (double buffered), 256 threads transform one
// (channel, tile) each into V[xi][k][tile] in LDS, the U fragments come straight from global memory in a pre-packed,
// per-lane-contiguous order (c2m_wino_filter_transform), 32 MFMAs per wave.  The inverse transform is split: the column
// combination (over j) happens in registers, the row combination (over i = waves) through LDS in the epilogue, which also
// applies bias + activation and writes coalesced NCHW rows.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define WINO_OOB 0x80000000u

struct WinoP {
    const float* U;      // packed filter transform (see wino_filter_kernel)
    const float* X;
    float* Y;
    const float* bias;
    int M, K, nimg;      // output channels (rows), input channels, images
    int Hi, Wi, Ho, Wo;
    int iy0, ix0;        // input row/col of output (0,0)'s first tap (= -pad forward; see host for the data gradient)
    int reflect;
    long in_sn, out_sn, out_sc, out_sh, out_off;
    int in_sc, in_sh;
    unsigned x_bytes;
    int nchunks, mtiles;
    int act;
    float slope;
    // optional second target (data gradient of a reflect-padded conv, computed over the padded domain): outputs inside
    // [lo, lo + ext) go straight to the unpadded gradient Y2, only the pad ring is written to Y (c2m_reflect_border_add
    // then folds the ring)
    float* Y2;
    long y2_sn, y2_sc, y2_sh, y2_st;         // y2_st: frame stride of Y2 (3x3x3 layers; 0 for 2-D)
    int lo_y, lo_x, ext_y, ext_x;
    // 3x3x3 convolutions as a 2-D Winograd over "virtual" input channels (time tap, channel): image = (sample n, frame t),
    // virtual channel v = kt * cin + ci reads frame t + kt + toff of channel ci (reflected in time or absent -> zero records).
    // nkt = 0: plain 2-D layer (To = 1, in_st = out_st = 0).
    int To, Ti, nkt, cin, toff, treflect;
    long in_st, out_st;
    // region shape in Winograd tiles (GEN kernels): th x tw <= 32 tiles, (2 th + 2)(2 tw + 2) <= 192 patch positions.
    // The fixed 4 x 8 shape (8 x 16 outputs) leaves e.g. the 18 x 34 padded domain of a 16 x 32 reflect data gradient at
    // 53 % fill (9 regions); 3 x 10 tiles cover it with 6.
    int th, tw;
    // Temporal reflect padding folded into the data gradient of a 3x3x3 layer (ptab != 0; needs cin % 8 == 0): output frame
    // t of the UNPADDED time axis sums over its (source frame, time tap) pairs -- 3 in the middle, 2 at the ends, + 1 for the
    // frames the pad frames mirror onto -- instead of launching over T + 2 frames and folding (T = 5: 7 frames, 29 % of the
    // launch for two frames that only feed a fold).  ptab[t][11] = {npairs, (source frame, U block) x 5}; cpk = cin / 8.
    const int* ptab;
    int cpk;
    // pad-ring terms of a reflect data gradient computed over the EXACT domain (conv_ring.hip, buffer mode): R [nimg][M][4][r_l],
    // added to rows 1 / Ho-2 (sides 0 / 1, indexed by column) and columns 1 / Wo-2 (sides 2 / 3, indexed by row); or NULL
    const float* R;
    int r_l;
};

constexpr int WR = 8, WC = 16;               // output region rows / cols
constexpr int PH = WR + 2, PW = WC + 2;      // input patch
constexpr int CKW = 8;                       // channels per chunk
constexpr int PPOS = PH * PW;                // 180 patch positions per channel: thread tid < 180 owns position tid
constexpr int PCS = 192;                     // channel stride of the patch in LDS = three 64-lane DMA rows
constexpr int PBUF = CKW * PCS;              // one patch buffer


typedef float f32x2 __attribute__((ext_vector_type(2)));
// v_pk_add_f32 with operand swizzles: p = (x0, x1), q = (x2, x3)
static __device__ __forceinline__ f32x2 pk_a(f32x2 p, f32x2 q) {       // (x0 - x2, x1 + x2)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(p), "v"(q));
    return d;
}
static __device__ __forceinline__ f32x2 pk_b(f32x2 p, f32x2 q) {       // (x2 - x1, x1 - x3)
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p), "v"(q));
    return d;
}

// MT = 32-row output-channel tiles per workgroup: 2 (64 output channels, 128 accumulator registers, two workgroups per CU) or
// 1 (<= 32 output channels -- the full-resolution heads and the data gradients of 32-channel inputs: 64 accumulators, three
// workgroups per CU; half the MFMAs per chunk at the same transform / DMA work, which still beats the direct kernels' 32-row
// tiles).  U is packed per 64 rows either way; MT = 1 reads the fragments of rows 0..31 (every second 1 KB record).
// GEN = region shape from p.th x p.tw instead of the fixed 4 x 8 tiles: same K loop (V is [xi][k][32 tile columns] either
// way; columns >= th*tw are idle), runtime patch width, and an epilogue that owns one tile row (2 pixels) per thread
// instead of 4 consecutive pixels (8-byte instead of 16-byte stores).
template <int MT, bool GEN>
__global__ __launch_bounds__(256, MT == 1 ? 3 : 2) void conv_wino_kernel(const WinoP p) {
    constexpr int NU = 4 * MT;                       // U fragment records (16 bytes per lane) per wave and chunk
    __shared__ float sP[3][PBUF];                   // input patch [k][PH][PW] (+pad), filled by LDS-DMA two chunks ahead
    __shared__ float sV[2 * 16 * CKW * 32];         // V[buf][xi][k][tile] (double buffered); reused by the epilogue
#ifdef WINO_OCC1
    __shared__ float sDummy[20000];
    if (threadIdx.x == 999) sDummy[p.M] = 1.f;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int TH = GEN ? p.th : WR / 2, TW = GEN ? p.tw : WC / 2;       // tiles per region
    const int pw_ = GEN ? 2 * TW + 2 : PW, ppos = GEN ? (2 * TH + 2) * pw_ : PPOS;
    const int regions_x = (p.Wo + 2 * TW - 1) / (2 * TW), regions_y = (p.Ho + 2 * TH - 1) / (2 * TH);
    // XCD-aware work order: workgroups are handed to the 8 XCDs round-robin by linear id, each XCD has its own L2.  Work
    // item w = region * mtiles + mt (the m-tiles of a region read the SAME input patch, neighbouring regions share its
    // halo); XCD x gets the contiguous range of items [x*q + min(x, r), ...) so those re-reads hit its L2 instead of
    // going out to the fabric once per m-tile (the launch used to read X mtiles times: 3-6x the tensor).
    int rb, mt;
    {
        const unsigned L = blockIdx.x, total = gridDim.x;
        const unsigned q = total >> 3, r = total & 7u, x = L & 7u, j = L >> 3;
        const unsigned w = x * q + (x < r ? x : r) + j;
        mt = (int)(w % (unsigned)p.mtiles); rb = (int)(w / (unsigned)p.mtiles);
    }
    const int rx = rb % regions_x; rb /= regions_x;
    const int ry = rb % regions_y; const int img = rb / regions_y;
    const int oy0 = ry * 2 * TH, ox0 = rx * 2 * TW;

    // ---- patch addresses of this thread (fixed over the K loop).  A chunk's patch is 8 channels x 3 DMA rows of 64 patch
    // positions (180 used); wave w fetches the three rows of channels 2w and 2w + 1, so a lane owns the three positions
    // lane, 64 + lane, 128 + lane and the channel rides in the scalar offset of the DMA: the reflect / bounds arithmetic
    // is done for 3 addresses per thread instead of 6 (and no div / mod by the channel size); a VALU next to the MFMAs of
    // the other resident workgroup costs 2.5 ... 5 matrix-pipe cycles (tools/micro/mfma_issue.hip).
    const int smp = p.nkt ? img / p.To : img, frm = p.nkt ? img - smp * p.To : 0;       // sample, frame (scalar)
    const unsigned img_byte = (unsigned)(smp * (int)p.in_sn) * 4u;
    int nchunks = p.nchunks;
    int pr_src[5] = {0, 0, 0, 0, 0}, pr_ub[5] = {0, 0, 0, 0, 0}, npairs = 0;
    if (p.ptab) {
        const int* __restrict__ pt = p.ptab + frm * 11;
        npairs = pt[0];
#pragma unroll
        for (int j = 0; j < 5; ++j) { pr_src[j] = pt[1 + 2 * j]; pr_ub[j] = pt[2 + 2 * j]; }
        nchunks = npairs * p.cpk;
    }
    auto sel5 = [&](const int (&a)[5], int j) { return j == 0 ? a[0] : (j == 1 ? a[1] : (j == 2 ? a[2] : (j == 3 ? a[3] : a[4]))); };
    unsigned pvo[3];
#pragma unroll
    for (int sg = 0; sg < 3; ++sg) {
        const int pos = sg * 64 + lane;
        const int r = pos / pw_, c = pos % pw_;
        int iy = oy0 + p.iy0 + r, ix = ox0 + p.ix0 + c;
        if (p.reflect) {
            iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
        }
        const bool ok = pos < ppos && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        pvo[sg] = ok ? img_byte + (unsigned)(iy * p.in_sh + ix) * 4u : WINO_OOB;
    }
    // The LDS-DMA is issued through inline asm and TWO chunks ahead (three patch buffers).  hipcc makes the first MFMA
    // after an LDS-DMA wait until every VMEM operation older than the U-fragment loads has completed; with the builtin and
    // a one-chunk-ahead prefetch that put the whole DMA latency in front of the MFMAs (40 % MFMA utilisation).  Issued
    // after the U loads and a full iteration before its data is needed, the DMA is never waited for while it is young;
    // the counted s_waitcnt at the top of each iteration (vmcnt(6): only the youngest DMA may still be in flight)
    // guarantees the patch of the current chunk has landed before the barrier.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long xaddr = (unsigned long)p.X;
    const u32x4 rs = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const unsigned sp_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)&sP[0][0];
    // DMA row j = 0..5 of this wave: channel 2 * wave + j / 3, positions 64 * (j % 3) + lane -> 64 consecutive floats of
    // sP[buf][k].  The scalar offset is outside the hardware's range check, so a channel beyond K (last chunk) gets zero
    // records instead (every lane reads 0)
    // The two channels of this wave in the chunk being fetched (chunks are fetched strictly in order 0, 1, 2, ...):
    // scalar offset and liveness, prepared ONCE per chunk by dma_prepare.  3x3x3 layers keep (time tap, channel) of channel
    // chunk*8 + 2*wave as running counters -- the first version divided by cin in every one of the six DMA rows, ~40 scalar
    // instructions each in front of the next MFMA pair (the 3-D layers ran 20-25 % below 2-D layers of the same depth).
    int d_kt = 0, d_ci = 2 * wave;                      // channel 2*wave of chunk 0 (pair mode: d_kt = pair index)
    if (p.nkt) { d_kt = d_ci / p.cin; d_ci -= d_kt * p.cin; }
    int u_j = 0, u_r = 0;                               // pair mode: (pair, chunk within the pair) of the next U fetch
    int d_soff[2]; bool d_on[2];
    auto dma_prepare = [&](int chunk, bool live) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = chunk * CKW + 2 * wave + h;
            bool on = live && ch < p.K;
            int soff = ch * p.in_sc * 4;
            if (p.ptab) {                               // (pair, channel); cin % 8 == 0: a chunk never straddles pairs
                const int tt = sel5(pr_src, d_kt);
                on = live && d_kt < npairs;
                soff = (int)(((long)(d_ci + h) * p.in_sc + (long)tt * p.in_st) * 4);
            } else if (p.nkt) {                         // (time tap, channel): all scalar arithmetic
                int kt = d_kt, ci = d_ci + h;
                if (ci >= p.cin) { ci -= p.cin; ++kt; }   // h = 1 wraps at most once (cin >= 1)
                int tt = frm + kt + p.toff;
                if (p.treflect) { tt = tt < 0 ? -tt : tt; tt = tt >= p.Ti ? 2 * p.Ti - 2 - tt : tt; }
                on = on && (unsigned)tt < (unsigned)p.Ti;
                soff = (int)(((long)ci * p.in_sc + (long)tt * p.in_st) * 4);
            }
            d_soff[h] = soff; d_on[h] = on;
        }
        if (p.nkt) {                                    // advance to the next chunk: + 8 channels
            d_ci += CKW;
            while (d_ci >= p.cin) { d_ci -= p.cin; ++d_kt; }
        }
    };
    auto dma_row = [&](int j, int buf) {
        const int h = j / 3, k = 2 * wave + h, sg = j % 3;
        u32x4 rsk = rs;
        rsk[2] = d_on[h] ? p.x_bytes : 0u;
        const unsigned dst = sp_lds + (unsigned)((buf * PBUF + k * PCS + sg * 64) * 4);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                     :: "s"(dst), "v"(pvo[sg]), "s"(rsk), "s"(d_soff[h]) : "memory");
    };
    auto load_patch = [&](int chunk, int buf) {
        dma_prepare(chunk, true);
#pragma unroll
        for (int j = 0; j < 6; ++j) dma_row(j, buf);
    };
    // ---- U fragments: 8 float4 per lane and chunk, [i = j*2 + mi][lane][kk 0..3]: every load instruction of a wave reads
    // 1 KB contiguous (per-lane-contiguous 128-byte records made the texture addresser the bottleneck: 64 lines per load)
    const float* __restrict__ ubase = p.U + (((long)mt * 4 + wave) * 8 * 64 + lane) * 4;
    const long ustride = (long)p.mtiles * 4 * 64 * 32;      // floats per chunk
    const unsigned long uaddr = (unsigned long)p.U;
    const u32x4 urs = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    const unsigned uvo = (unsigned)((((mt * 4 + wave) * 8 * 64) + lane) * 16);
    f32x4 ua[NU];
    // U chunk of the next K chunk in fetch order (chunks are fetched in order 0, 1, 2, ...): the chunk itself, or in pair mode
    // block pr_ub[pair] of cpk chunks
    auto next_uchunk = [&](int chunk) {
        if (!p.ptab) return chunk;
        const int uc = sel5(pr_ub, u_j) * p.cpk + u_r;
        if (++u_r == p.cpk) { u_r = 0; ++u_j; }
        return uc;
    };
    auto load_u = [&](int chunk) {
        chunk = next_uchunk(chunk);
        const f32x4* __restrict__ q = reinterpret_cast<const f32x4*>(ubase + chunk * ustride);
#pragma unroll
        for (int i = 0; i < NU; ++i) ua[i] = q[(i * 2 / MT) * 64];
    };

    f32x16 acc[4][MT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;

    // input-transform role of this thread: tile n (ty, tx), channel k of the chunk
    const int tn = tid & 31, tk = tid >> 5;
    int ty = tn >> 3, tx = tn & 7;
    if (GEN) {                                       // idle tile columns transform tile 0 again (result never stored)
        ty = tn < TH * TW ? tn / TW : 0; tx = tn < TH * TW ? tn % TW : 0;
    }
    const int pbase = tk * PCS + (2 * ty) * pw_ + 2 * tx;
    // V = B^T d B of this thread's (channel, tile) from patch buffer `pb` into V buffer `vb`
    auto read_d = [&](int pb, float (&d)[4][4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b] = sP[pb][pbase + a * pw_ + b];
    };
    auto transform_store = [&](const float (&d)[4][4], int vb) {
        float x[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[0][b] = d[0][b] - d[2][b];
            x[1][b] = d[1][b] + d[2][b];
            x[2][b] = d[2][b] - d[1][b];
            x[3][b] = d[1][b] - d[3][b];
        }
        float* __restrict__ v = sV + vb * (16 * CKW * 32) + tk * 32 + tn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(4 * i + 0) * CKW * 32] = x[i][0] - x[i][2];
            v[(4 * i + 1) * CKW * 32] = x[i][1] + x[i][2];
            v[(4 * i + 2) * CKW * 32] = x[i][2] - x[i][1];
            v[(4 * i + 3) * CKW * 32] = x[i][1] - x[i][3];
        }
    };

    // ---- prologue: patches 0 and 1 by DMA, U(0), V(0)
    load_patch(0, 0);
    if (nchunks > 1) load_patch(1, 1);
    load_u(0);
#pragma unroll
    for (int i = 0; i < NU; ++i) asm volatile("" :: "v"(ua[i]));   // retire these loads HERE: a load pending at loop entry
                                                       // costs a vmcnt(0) in front of the first MFMA of EVERY iteration
    if (nchunks > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        float d0[4][4];
        read_d(0, d0);
        transform_store(d0, 0);
    }
    int pnext = 1;                                     // patch buffer of chunk + 1 = (chunk + 1) % 3
    // One K-loop iteration; `ucur` holds U(chunk), U(chunk + 1) is loaded into `unext`.  The loop below is unrolled by two
    // with the two register sets swapping roles, so no register copies follow the wait.  The U requests are issued in the
    // shadow of the first eight MFMAs (two per MFMA pair) instead of in front of them, the DMA requests of patch(chunk + 2)
    // behind those (U loads stay older than the DMAs, so the counted wait still separates them), and the input transform of
    // the next chunk is mixed into the remaining MFMAs.
    auto iteration = [&](const int chunk, f32x4 (&ucur)[NU], f32x4 (&unext)[NU]) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < nchunks;
        // patch(chunk + 1) (this wave's part) has landed: nothing younger is in flight at this point
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();       // V(chunk) complete; patch(chunk+1) complete; everyone is done with V(chunk-1) = buffer cur^1
        float b[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                b[j][kk] = sV[cur * (16 * CKW * 32) + ((4 * wave + j) * CKW + 2 * kk + (lane >> 5)) * 32 + (lane & 31)];
        // U(chunk + 1) by buffer loads: one VGPR offset for all eight, the chunk in the scalar offset, i * 1 KB as the
        // immediate -- a global_load needs a 64-bit VALU add per address, and tools/micro/mfma_issue.hip prices a VALU
        // between MFMAs at 2.5 ... 5 MFMA-pipe cycles and a global load above a buffer load
        const int usoff0 = (int)((more ? next_uchunk(chunk + 1) : 0) * ustride * 4), usoff1 = usoff0 + 4096;
        auto mfma_pair = [&](int g) {                  // g = kk * 4 + j
            const int kk = g >> 2, j = g & 3;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const f32x4 v = ucur[j * MT + mi];
                const float av = kk == 0 ? v.x : (kk == 1 ? v.y : (kk == 2 ? v.z : v.w));
                acc[j][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][kk], acc[j][mi], 0, 0, 0);
            }
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mfma_pair(g);
#pragma unroll
            for (int i = MT * g; i < MT * g + MT; ++i) {
                constexpr int dummy = 0; (void)dummy;
                const int rec = i * 2 / MT;            // record j*2 + mi of the 64-row packing
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                             : "=&v"(unext[i]) : "v"(uvo), "s"(urs), "s"(rec < 4 ? usoff0 : usoff1), "n"((rec & 3) * 1024) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // patch(chunk + 2) -- always issued, so the iteration is one basic block and the requests sit between MFMAs; past
        // the last chunk they carry zero records (the loads return 0 into a patch buffer nobody reads)
        const bool pre = chunk + 2 < nchunks;
        const int dbuf = pnext == 2 ? 0 : pnext + 1;
        dma_prepare(chunk + 2, pre);
#pragma unroll
        for (int g = 4; g < 7; ++g) {
            mfma_pair(g);
            dma_row(2 * (g - 4), dbuf);
            dma_row(2 * (g - 4) + 1, dbuf);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Input transform of chunk + 1 (unconditional: one basic block; last chunk: stale but in-bounds data, result unused)
        // in packed fp32: V = B^T d B is 32 adds, i.e. 16 v_pk_add_f32 -- the row part needs the op_sel / neg forms the
        // compiler does not emit, hence pk_a / pk_b.  A VALU between MFMAs costs 2.5 ... 5 matrix-pipe cycles
        // (tools/micro/mfma_issue.hip), and in each gap the LDS writes of the previous row go first: a VALU behind an MFMA
        // waits for the pipe and holds back what is queued behind it, an LDS instruction does not.
        f32x2 dr[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int h = 0; h < 2; ++h) dr[a][h] = *reinterpret_cast<const f32x2*>(&sP[pnext][pbase + a * pw_ + 2 * h]);
        float* __restrict__ vdst = sV + (cur ^ 1) * (16 * CKW * 32) + tk * 32 + tn;
        f32x2 xr_[4][2], o01, o23;
        auto row = [&](int i) {
            o01 = pk_a(xr_[i][0], xr_[i][1]);          // (x0 - x2, x1 + x2)
            o23 = pk_b(xr_[i][0], xr_[i][1]);          // (x2 - x1, x1 - x3)
        };
        auto put = [&](int i) {
            vdst[(4 * i + 0) * CKW * 32] = o01.x; vdst[(4 * i + 1) * CKW * 32] = o01.y;
            vdst[(4 * i + 2) * CKW * 32] = o23.x; vdst[(4 * i + 3) * CKW * 32] = o23.y;
        };
        mfma_pair(7);
        xr_[0][0] = dr[0][0] - dr[2][0]; xr_[0][1] = dr[0][1] - dr[2][1];
        xr_[1][0] = dr[1][0] + dr[2][0]; xr_[1][1] = dr[1][1] + dr[2][1];
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(8);
        xr_[2][0] = dr[2][0] - dr[1][0]; xr_[2][1] = dr[2][1] - dr[1][1];
        xr_[3][0] = dr[1][0] - dr[3][0]; xr_[3][1] = dr[1][1] - dr[3][1];
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(9);
        row(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            mfma_pair(9 + i);
            put(i - 1);
            row(i);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma_pair(13);
        put(3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_pair(14);
        mfma_pair(15);
        // the 8 U loads are older than the 6 DMA loads issued after them
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NU; ++i) asm volatile("" : "+v"(unext[i]));   // uses of U(chunk + 1) stay behind the wait
        pnext = pnext == 2 ? 0 : pnext + 1;
    };
    f32x4 ub[NU];
    int chunk = 0;
    for (; chunk + 1 < nchunks; chunk += 2) {
        iteration(chunk, ua, ub);
        iteration(chunk + 1, ub, ua);
    }
    if (chunk < nchunks) iteration(chunk, ua, ub);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (empty) DMAs of the last iterations must not outlive the block's LDS
    __syncthreads();

    // ---- inverse transform.  Column part in registers: R_i[q] = sum_j M[i][j] A[j][q], A = [[1,0],[1,1],[1,-1],[0,-1]]
    // Row part through LDS: Y[0][q] = R_0 + R_1 + R_2, Y[1][q] = R_1 - R_2 - R_3 (i = wave)
    // Every thread owns one output pixel (x, y) of the region for the whole epilogue and walks over the output channels, so
    // the pixel's tile, row parity, bounds and target address are computed once (the per-element index arithmetic used to
    // cost as much as the stores: 27 % of the kernel on the 64-channel layers).
    constexpr int QS = 4 * 16 * 32 + 16;             // q stride: +16 floats keeps the two column parities on different banks
    float* __restrict__ sR = sV;                     // [q][w][16 cout][32 tiles] per pass (16 of the 64 couts)
    const int m0 = mt * 64;
    // 16-byte stores (round 2): a thread owns FOUR consecutive x of one output row and two of the 16 output channels of a
    // pass, so a pass is 2 store instructions per thread instead of 8 dword stores -- the store path of a CU takes ~5 B/clk
    // in dword stores, which made the epilogue 8 % (K = 256) to 30 % (K = 64) of the kernel.
    // The stores are 4-byte aligned vector stores, so any row length / stride works: a group that crosses the end of its
    // row -- padded domains are W + 2 wide -- is stored per pixel.  (The first version of this path required W % 4 == 0 and
    // left every reflect data gradient on a dword-store form.)
    if constexpr (GEN) {
        // one tile row (2 pixels) per thread: tile column n = tid & 31, row parity, 4 channel groups
        const int en = tid & 31, rp = (tid >> 5) & 1, cg = tid >> 6;
        const int ety = en < TH * TW ? en / TW : 0, etx = en < TH * TW ? en % TW : 0;
        const int oy = oy0 + 2 * ety + rp, ox = ox0 + 2 * etx;
        const bool inb = en < TH * TW && oy < p.Ho && ox < p.Wo;
        const bool top = rp == 0;
        int mode = 0;                                 // 0: Y (vector), 1: Y2 (vector), 2: mixed (per pixel)
        float* __restrict__ yb0 = p.Y + p.out_off + (long)smp * p.out_sn + (long)frm * p.out_st + (long)oy * p.out_sh + ox;
        long cs0 = p.out_sc;
        float* __restrict__ yb1 = yb0;
        long cs1 = cs0;
        const int yi = oy - p.lo_y, xi = ox - p.lo_x;
        if (p.Y2 && (unsigned)yi < (unsigned)p.ext_y) {
            const bool in0 = (unsigned)xi < (unsigned)p.ext_x, in1 = (unsigned)(xi + 1) < (unsigned)p.ext_x;
            yb1 = p.Y2 + (long)smp * p.y2_sn + (long)frm * p.y2_st + (long)yi * p.y2_sh + xi;
            cs1 = p.y2_sc;
            mode = (in0 && in1) ? 1 : ((in0 || in1) ? 2 : 0);
        }
        if (ox + 1 >= p.Wo && mode == 0) mode = 2;
        const int ro = top ? 0 : 3 * 16 * 32;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = hf * 8 + rr;
                    const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                    sR[0 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[0][mi][r] + acc[1][mi][r]) + acc[2][mi][r];
                    sR[1 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[1][mi][r] - acc[2][mi][r]) - acc[3][mi][r];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int co = cg + 4 * it;
                    const int cout = m0 + mi * 32 + hf * 16 + co;
                    float v[2];
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const float* __restrict__ rr_ = sR + dx * QS + co * 32 + en;
                        const float r1 = rr_[1 * 16 * 32], r2 = rr_[2 * 16 * 32], rx = rr_[ro];
                        v[dx] = top ? (rx + r1) + r2 : (r1 - r2) - rx;
                    }
                    if (inb && cout < p.M) {
                        if (p.R) {                             // uniform: ring terms of the reflect data gradient (2-D: img = smp)
                            const float* __restrict__ rb = p.R + ((long)img * p.M + cout) * 4 * p.r_l;
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                if (ox + dx >= p.Wo) continue;
                                if (oy == 1) v[dx] += rb[ox + dx];
                                if (oy == p.Ho - 2) v[dx] += rb[p.r_l + ox + dx];
                                if (ox + dx == 1) v[dx] += rb[2 * p.r_l + oy];
                                if (ox + dx == p.Wo - 2) v[dx] += rb[3 * p.r_l + oy];
                            }
                        }
                        const float bb = p.bias ? p.bias[cout] : 0.f;
                        v[0] = c2m_act(v[0] + bb, p.act, p.slope); v[1] = c2m_act(v[1] + bb, p.act, p.slope);
                        if (mode == 2) {
#pragma unroll
                            for (int dx = 0; dx < 2; ++dx) {
                                if (ox + dx >= p.Wo) continue;
                                if ((unsigned)(xi + dx) < (unsigned)p.ext_x) yb1[(long)cout * cs1 + dx] = v[dx];
                                else yb0[(long)cout * cs0 + dx] = v[dx];
                            }
                        } else {
                            typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
                            const f32x2u vv = {v[0], v[1]};
                            *reinterpret_cast<f32x2u*>((mode ? yb1 : yb0) + (long)cout * (mode ? cs1 : cs0)) = vv;
                        }
                    }
                }
            }
        }
    } else {
        const int e4 = tid & 3, ey = (tid >> 2) & 7, cg = tid >> 5;
        const int oy = oy0 + ey, ox = ox0 + 4 * e4;
        const bool inb = oy < p.Ho && ox < p.Wo;
        const bool top = (ey & 1) == 0;
        // target of the 4-pixel group: all interior -> Y2, all outside the interior -> Y, straddling -> per pixel
        int mode = 0;                                 // 0: Y (vector), 1: Y2 (vector), 2: mixed (scalar per pixel)
        float* __restrict__ yb0 = p.Y + p.out_off + (long)smp * p.out_sn + (long)frm * p.out_st + (long)oy * p.out_sh + ox;
        long cs0 = p.out_sc;
        float* __restrict__ yb1 = yb0;
        long cs1 = cs0;
        const int yi = oy - p.lo_y, xi = ox - p.lo_x;
        if (p.Y2 && (unsigned)yi < (unsigned)p.ext_y) {
            const bool in0 = (unsigned)xi < (unsigned)p.ext_x, in1 = (unsigned)(xi + 1) < (unsigned)p.ext_x,
                       in2 = (unsigned)(xi + 2) < (unsigned)p.ext_x, in3 = (unsigned)(xi + 3) < (unsigned)p.ext_x;
            yb1 = p.Y2 + (long)smp * p.y2_sn + (long)frm * p.y2_st + (long)yi * p.y2_sh + xi;
            cs1 = p.y2_sc;
            // the interior is shifted by the pad against the padded domain, so its 16-byte stores are only dword aligned
            // (fine for global_store_dwordx4); groups that straddle the interior's edge are stored per pixel
            mode = (in0 && in3) ? 1 : ((in0 || in1 || in2 || in3) ? 2 : 0);
        }
        if (ox + 3 >= p.Wo && mode == 0) mode = 2;    // the group crosses the end of the row
        const int tile0 = (ey >> 1) * 8 + 2 * e4;
        const int ro = top ? 0 : 3 * 16 * 32;        // row combined with r1, r2: r0 (top) or r3 (bottom)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = hf * 8 + rr;
                    const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);        // 0..15 within the half
                    sR[0 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[0][mi][r] + acc[1][mi][r]) + acc[2][mi][r];
                    sR[1 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[1][mi][r] - acc[2][mi][r]) - acc[3][mi][r];
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int co = cg + 8 * it;
                    const int cout = m0 + mi * 32 + hf * 16 + co;
                    float v[4];
#pragma unroll
                    for (int dx = 0; dx < 4; ++dx) {
                        const float* __restrict__ rr_ = sR + (dx & 1) * QS + co * 32 + tile0 + (dx >> 1);
                        const float r1 = rr_[1 * 16 * 32], r2 = rr_[2 * 16 * 32], rx = rr_[ro];
                        v[dx] = top ? (rx + r1) + r2 : (r1 - r2) - rx;
                    }
                    if (inb && cout < p.M) {
                        if (p.R) {                             // uniform: ring terms of the reflect data gradient (2-D: img = smp)
                            const float* __restrict__ rb = p.R + ((long)img * p.M + cout) * 4 * p.r_l;
                            if (oy == 1 || oy == p.Ho - 2) {
                                const float* __restrict__ rr = rb + (oy == 1 ? 0 : p.r_l) + ox;
#pragma unroll
                                for (int dx = 0; dx < 4; ++dx) if (ox + dx < p.Wo) v[dx] += rr[dx];
                            }
                            if (ox <= 1 && ox + 3 >= 1) {
                                const float q = rb[2 * p.r_l + oy];
#pragma unroll
                                for (int dx = 0; dx < 4; ++dx) if (ox + dx == 1) v[dx] += q;
                            }
                            if (ox <= p.Wo - 2 && ox + 3 >= p.Wo - 2) {
                                const float q = rb[3 * p.r_l + oy];
#pragma unroll
                                for (int dx = 0; dx < 4; ++dx) if (ox + dx == p.Wo - 2) v[dx] += q;
                            }
                        }
                        const float bb = p.bias ? p.bias[cout] : 0.f;
#pragma unroll
                        for (int dx = 0; dx < 4; ++dx) v[dx] = c2m_act(v[dx] + bb, p.act, p.slope);
                        if (mode == 2) {
#pragma unroll
                            for (int dx = 0; dx < 4; ++dx) {
                                if (ox + dx >= p.Wo) continue;
                                if ((unsigned)(xi + dx) < (unsigned)p.ext_x) yb1[(long)cout * cs1 + dx] = v[dx];
                                else yb0[(long)cout * cs0 + dx] = v[dx];
                            }
                        } else {
                            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                            const f32x4u vv = {v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4u*>((mode ? yb1 : yb0) + (long)cout * (mode ? cs1 : cs0)) = vv;
                        }
                    }
                }
            }
        }
    }
}

// Filter transform U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], written in the fragment order the conv
// kernel reads: Upack[chunk][mt][w][j][mi][lane][kk] = U[xi = 4w + j][m = mt*64 + mi*32 + (lane & 31)][c = chunk*8 + 2kk + (lane >> 5)].
// dgrad = 0: m = output channel, c = input channel, g = w[m][c];  dgrad = 1 (data gradient of a stride-1 conv): m = input
// channel, c = output channel, g = w[c][m] rotated by 180 degrees.
// One thread per (m, c): the nine filter taps are read once and all 16 transformed values written (each store instruction
// of a wave covers 256 consecutive floats of Upack).  A thread per output value re-read the taps 16 times.
__global__ void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ up, int M, int K, int Cin_native,
                                   int dgrad, int mtiles, long pairs) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 3); long r = i >> 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int mi = (int)(r & 1); r >>= 1;
        const int mt = (int)(r % mtiles); const int chunk = (int)(r / mtiles);
        const int m = mt * 64 + mi * 32 + (lane & 31), c = chunk * CKW + 2 * kk + (lane >> 5);
        float u[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) u[a][b2] = 0.f;
        if (m < M && c < K) {
            const float* __restrict__ g = dgrad ? w + ((long)c * Cin_native + m) * 9 : w + ((long)m * Cin_native + c) * 9;
            float gg[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) gg[a][b2] = dgrad ? g[(2 - a) * 3 + (2 - b2)] : g[a * 3 + b2];
            float row[4][3];                                  // G g
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) {
                const float g0 = gg[0][b2], g1 = gg[1][b2], g2 = gg[2][b2];
                row[0][b2] = g0; row[1][b2] = 0.5f * ((g0 + g1) + g2); row[2][b2] = 0.5f * ((g0 - g1) + g2); row[3][b2] = g2;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {                     // (G g) G^T
                u[a][0] = row[a][0];
                u[a][1] = 0.5f * ((row[a][0] + row[a][1]) + row[a][2]);
                u[a][2] = 0.5f * ((row[a][0] - row[a][1]) + row[a][2]);
                u[a][3] = row[a][2];
            }
        }
        const long base = ((long)chunk * mtiles + mt) * 4;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                up[((((base + wv) * 8 + (j * 2 + mi)) * 64 + lane) << 2) + kk] = u[wv][j];
    }
}

C2M_API long c2m_wino_upack_floats(int M, int K) {
    return 16L * (c2m_cdiv(M, 64) * 64L) * (c2m_cdiv(K, CKW) * (long)CKW);
}

// w: native [Cout][Cin][3][3].  dgrad = 0: M = Cout, K = Cin;  dgrad = 1: M = Cin, K = Cout.
C2M_API int c2m_wino_filter_transform(const float* w, float* upack, int Cout, int Cin, int dgrad, void* stream) {
    C2M_ENTER();
    const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    if (M <= 0 || K <= 0) return 0;
    const long total = c2m_wino_upack_floats(M, K);
    hipLaunchKernelGGL(wino_filter_kernel, dim3(c2m_grid(total / 16, 256)), dim3(256), 0, (hipStream_t)stream, w, upack, M,
                       K, Cin, dgrad, c2m_cdiv(M, 64), total / 16);
    return (int)hipGetLastError();
}

// Region shape of a launch over an Ho x Wo output domain: the fixed 4 x 8 tiles (8 x 16 outputs; 16-byte-store epilogue)
// unless another th x tw (th*tw <= 32 tile columns of the MFMA, (2th+2)(2tw+2) <= 192 = three DMA rows of patch positions)
// covers the domain with at most 0.9x the regions.  Ties: the wider shape (longer contiguous rows).
// C2M_WINO_SHAPE=fixed pins 4 x 8 (A/B).
static void wino_region_shape(int Ho, int Wo, int* th, int* tw) {
    static const bool fixed = [] { const char* e = getenv("C2M_WINO_SHAPE"); return e && e[0] == 'f'; }();
    *th = WR / 2; *tw = WC / 2;
    const long base = (long)c2m_cdiv(Ho, WR) * c2m_cdiv(Wo, WC);
    if (fixed) return;
    long best = base;
    for (int a = 1; a <= 16; ++a)
        for (int b = 1; b <= 32; ++b) {
            if (a * b > 32 || (2 * a + 2) * (2 * b + 2) > PCS) continue;
            const long r = (long)c2m_cdiv(Ho, 2 * a) * c2m_cdiv(Wo, 2 * b);
            if (10 * r > 9 * base) continue;
            if (r < best || (r == best && b > *tw)) { best = r; *th = a; *tw = b; }
        }
}

// regions per image the launch will use (callers: eligibility / fill estimates)
C2M_API int c2m_wino_regions(int Ho, int Wo) {
    int th, tw;
    wino_region_shape(Ho, Wo, &th, &tw);
    return c2m_cdiv(Ho, 2 * th) * c2m_cdiv(Wo, 2 * tw);
}

// geom[] (int64): 0 M, 1 K, 2 images, 3 Hi, 4 Wi, 5 Ho, 6 Wo, 7 iy0, 8 ix0, 9 reflect, 10 in_sn, 11 in_sc, 12 in_sh,
//                 13 out_sn, 14 out_sc, 15 out_sh, 16 out_off, 17 x_bytes; with Y_interior: 18 y2_sn, 19 y2_sc, 20 y2_sh,
//                 21 lo_y, 22 lo_x, 23 ext_y, 24 ext_x; 25..32 the 3x3x3 tail; 33 device pointer of the temporal pair table (or 0)
C2M_API int c2m_conv_wino(const float* upack, const float* X, float* Y, float* Y_interior, const float* bias,
                          const int64_t* g, int act, float slope, void* stream) {
    C2M_ENTER();
    WinoP p;
    p.U = upack; p.X = X; p.Y = Y; p.bias = bias;
    p.Y2 = Y_interior; p.y2_sn = p.y2_sc = p.y2_sh = p.y2_st = 0; p.lo_y = p.lo_x = p.ext_y = p.ext_x = 0;
    if (Y_interior) {
        p.y2_sn = g[C2M_WG_Y2_SN]; p.y2_sc = g[C2M_WG_Y2_SC]; p.y2_sh = g[C2M_WG_Y2_SH];
        p.lo_y = (int)g[C2M_WG_LO_Y]; p.lo_x = (int)g[C2M_WG_LO_X]; p.ext_y = (int)g[C2M_WG_EXT_Y]; p.ext_x = (int)g[C2M_WG_EXT_X];
    }
    p.M = (int)g[C2M_WG_M]; p.K = (int)g[C2M_WG_K]; p.nimg = (int)g[C2M_WG_NIMG];
    p.Hi = (int)g[C2M_WG_HI]; p.Wi = (int)g[C2M_WG_WI]; p.Ho = (int)g[C2M_WG_HO]; p.Wo = (int)g[C2M_WG_WO];
    p.iy0 = (int)g[C2M_WG_IY0]; p.ix0 = (int)g[C2M_WG_IX0]; p.reflect = (int)g[C2M_WG_REFLECT];
    p.in_sn = g[C2M_WG_IN_SN]; p.in_sc = (int)g[C2M_WG_IN_SC]; p.in_sh = (int)g[C2M_WG_IN_SH];
    p.out_sn = g[C2M_WG_OUT_SN]; p.out_sc = g[C2M_WG_OUT_SC]; p.out_sh = g[C2M_WG_OUT_SH]; p.out_off = g[C2M_WG_OUT_OFF];
    if (g[C2M_WG_X_BYTES] <= 0 || g[C2M_WG_X_BYTES] >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)g[C2M_WG_X_BYTES];
    p.To = (int)g[C2M_WG_TO]; p.in_st = g[C2M_WG_IN_ST]; p.out_st = g[C2M_WG_OUT_ST]; p.cin = (int)g[C2M_WG_CIN]; p.nkt = (int)g[C2M_WG_NKT]; p.toff = (int)g[C2M_WG_TOFF];
    p.Ti = (int)g[C2M_WG_TI]; p.treflect = (int)g[C2M_WG_TREFLECT];
    p.ptab = (const int*)(uintptr_t)g[C2M_WG_PTAB]; p.cpk = 0;
    p.R = (const float*)(uintptr_t)g[C2M_WG_RING]; p.r_l = (int)g[C2M_WG_RING_L];
    if (p.R && (Y_interior || p.reflect || g[C2M_WG_NKT] != 0 || p.Ho < 4 || p.Wo < 4 || p.r_l < p.Ho || p.r_l < p.Wo || p.out_off != 0))
        return (int)hipErrorInvalidValue;
    if (p.nkt) {
        if (p.To <= 0 || p.Ti <= 0 || p.cin <= 0 || p.nkt * p.cin != p.K || p.nimg % p.To) return (int)hipErrorInvalidValue;
        if (Y_interior) p.y2_st = (long)p.ext_y * p.y2_sh;      // frames of the interior target are dense [ext_y][y2_sh] planes
        if (p.ptab) {
            if (p.cin % CKW) return (int)hipErrorInvalidValue;
            p.cpk = p.cin / CKW;
        }
    } else {
        if (p.ptab) return (int)hipErrorInvalidValue;
        p.To = 1; p.in_st = p.out_st = 0;
    }
    p.act = act; p.slope = slope;
    if (p.M <= 0 || p.K <= 0 || p.nimg <= 0 || p.Ho <= 0 || p.Wo <= 0) return 0;
    if ((((uintptr_t)upack) & 15) != 0) return (int)hipErrorInvalidValue;
    p.nchunks = c2m_cdiv(p.K, CKW);
    p.mtiles = c2m_cdiv(p.M, 64);
    wino_region_shape(p.Ho, p.Wo, &p.th, &p.tw);
    const bool gen = !(p.th == WR / 2 && p.tw == WC / 2);
    const long regions = (long)p.nimg * c2m_cdiv(p.Ho, 2 * p.th) * c2m_cdiv(p.Wo, 2 * p.tw);
    if (regions * p.mtiles > 0x7fffffffL) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)(regions * p.mtiles));
    if (gen) {
        if (p.M <= 32) hipLaunchKernelGGL((conv_wino_kernel<1, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else           hipLaunchKernelGGL((conv_wino_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (p.M <= 32) hipLaunchKernelGGL((conv_wino_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else           hipLaunchKernelGGL((conv_wino_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}

// ================================================================================================ weight gradient
// Winograd form of the 3x3 stride-1 weight gradient (same layers as above):
//   dg[co][ci] = G^T [ sum_tiles (A dY A^T)[co][tile] (.) (B^T d B)[ci][tile] ] G
// i.e. 16 independent GEMMs dU_xi[co][ci] = dM_xi[co][:] . V_xi[ci][:] contracted over the 2x2-output tiles (N*H/2*W/2 of
// them) -- 16 multiplies where the direct form needs 36.
//
// One workgroup = 64 output channels x 64 input channels x 16 frequencies over a range of 2x16-output regions (8 tiles =
// one chunk = 4 MFMA k-steps).  Wave w owns the transformed ROW i = w (xi = 4w .. 4w+3): 4 xi x (2 x 2 tiles of 32x32) x
// 16 = 256 accumulator registers, one workgroup per CU.  Only the RAW patches are staged (double buffered: dY [co][2][16]
// and X [ci][4][24]); every lane builds its own MFMA fragments in registers: the A fragment of lane
// (co = lane & 31, tile = lane >> 5) is row i of A dY A^T of that lane's 2x2 dY block (4 floats from LDS, 7 VALU -> 4 xi),
// the B fragment is row i of B^T d B of its 4x4 input patch (2 raw rows x 4 floats, 8 VALU -> 4 xi).  No transformed
// operand ever goes through LDS (round 1's kernel wrote and re-read both and was bound by the ~170 non-MFMA instructions
// per 32 MFMAs); per 16 MFMAs a wave issues ~30 VALU and 16 ds_read_b64 (2-way bank conflicts at most: irrelevant at
// this intensity).
// Pixel ranges are split over workgroups into slabs [split][xi][Cout][Cin] (+ a bias-gradient slab from the plain sum of
// dY); wino_wgrad_reduce_kernel sums the splits in a fixed order, applies G^T (.) G and writes dW in its native layout.
struct WinoWgP {
    const float* dY; const float* X; float* slab; float* dbslab;
    int M, K, nimg, H, W;            // Cout, Cin, images, spatial size (dY and X have the same: stride 1, pad 1)
    int reflect;
    long dy_sn, x_sn;                // sample strides (elements)
    unsigned dy_bytes, x_bytes;
    // 3x3x3 layers: image = (sample, frame), input channels are virtual (time tap kt, channel): K = 3 * cin, the unit of
    // virtual channel v reads frame t + kt - 1 of channel v % cin (reflected in time at the first / last frame, or zeros).
    // T = 1, cin = K, cs = H * W for the 2-D layers.
    int T, cin;
    long cs;                         // channel stride (elements) of dY and X: T * H * W
    int regions, per_split;          // 2x16-output regions in total / per workgroup
};

constexpr int GR = 2, GC = 16, GT = 8;               // region rows, cols, tiles (one chunk: 4 k-steps of 2 tiles)
// LDS patches.  Lane (cl = lane & 31, kl = lane >> 5) reads float2 at [cl * stride + even offset]: with stride / 2 odd the
// 32 lanes of a ds_read_b64 pass hit 32 distinct bank pairs (strides 36 / 100 of the first version: lanes cl and cl + 16
// collided on every read).
constexpr int GY_STRIDE = GR * GC + 2;               // 34: dY patch [co][2][16] + 2 pad
constexpr int GX_COLS = GC + 4;                      // 20: image columns ox0-1 .. ox0+18 as five 16-byte units (4-B aligned)
constexpr int GX_STRIDE = 4 * GX_COLS + 6;           // 86: X patch [ci][4][20] + 6 pad

// v_pk_add_f32 form the compiler does not emit: (t.x + t.y, t.x - t.y)
static __device__ __forceinline__ f32x2 pk_sum_diff(f32x2 t) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(d) : "v"(t));
    return d;
}

// NI = 32-wide input-channel tiles per workgroup: NI = 2 -> 64 x 64 channels, 256 accumulator registers, one workgroup per
// CU; NI = 1 -> 64 x 32 channels, 128 accumulators, two workgroups per CU (their barriers / refills / epilogues overlap).
// MI = 32-row output-channel tiles per workgroup: 2 (64 x 32 channels) or 1 (Cout <= 32: 32 x 32 channels, 64 accumulators,
// three workgroups per CU -- half the MFMAs per k-step of the 64-row tile, none of them on empty rows).
// NB = LDS patch buffers: 2 (the refill of chunk c + 1 is loaded and stored during chunk c) or 3 (loaded during chunk c - 1,
// stored at the top of chunk c: a whole chunk in flight).  The 32-row variant's chunk is only 16 MFMAs per wave, shorter than
// the L2 / HBM latency of its refill -- with two buffers the 32 -> 32-channel layer ran 34 % MFMA-busy on ~2.9 TB/s of L2
// reads; the 64-row variant measured 3 % SLOWER with three buffers and keeps two.
template <int NI, int MI, int NB>
__global__ __launch_bounds__(256, NI == 1 ? (MI == 1 ? 3 : 2) : 1) void conv_wino_wgrad_kernel(const WinoWgP p) {
    constexpr int GY_UNITS = MI;                                   // dY float4 units per thread and chunk (32*MI co x 2 x 4 / 256)
    constexpr int NCI = 32 * NI;
    constexpr int GX_TOTAL = NCI * 4 * (GX_COLS / 4);              // 640 (NI = 1) / 1280 units per chunk
    constexpr int GX_UNITS = (GX_TOTAL + 255) / 256;               // 3 / 5 float4 per thread and chunk
    __shared__ float pY[NB][32 * MI * GY_STRIDE];
    __shared__ float pX[NB][NCI * GX_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware work order (see conv_wino_kernel): item w = split * tiles + tile.  All channel tiles of a split read the
    // same pixels -- dY once per input-channel tile, X once per output-channel tile -- so a split's tiles run on ONE XCD,
    // back to back, and the re-reads hit its L2 (a split's pixels x all channels is a few MB)
    int split, mt, nt;
    {
        const unsigned L = blockIdx.x, total = gridDim.x;
        const unsigned q = total >> 3, r = total & 7u, x = L & 7u, j = L >> 3;
        const unsigned w = x * q + (x < r ? x : r) + j;
        const unsigned mtiles = (unsigned)(p.M + 32 * MI - 1) / (32 * MI), ntiles = (unsigned)(p.K + NCI - 1) / NCI;
        const unsigned t = w % (mtiles * ntiles);
        split = (int)(w / (mtiles * ntiles)); mt = (int)(t % mtiles); nt = (int)(t / mtiles);
    }
    const int m0 = mt * 32 * MI, c0 = nt * NCI;
    const int HW = p.H * p.W;
    const int rbeg = split * p.per_split;
    int rend = rbeg + p.per_split; rend = rend < p.regions ? rend : p.regions;
    const int nchunks = rend - rbeg;
    const int regions_x = p.W / GC, regions_y = p.H / GR;

    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dY), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);

    // Raw patches are fetched as 16-byte units into registers one chunk ahead and stored with ds_write_b64 pairs (an
    // LDS-DMA piece costs the issuing wave 60-185 cycles and these waves are MFMA-issue bound).
    // dY unit u = tid + i*256: (co, row, quarter), 16-byte aligned.
    // X unit u: (ci, r, q): input row oy0-1+r, image columns ox0-1+4q .. ox0+2+4q (4-byte aligned loads), so the patch
    // column j of tile t is local column 2t + j: every lane reads its four values as two aligned float2.  The unit's
    // offset relative to the region origin is a per-thread constant (xbase), the origin is wave-uniform (roff, SALU): per
    // chunk a unit costs ONE VALU add.  Padding only matters in regions at the image border (wave-uniform flags):
    //   rows    r = 0 / r = 3 of a top / bottom region: reflect -> two rows inwards (rowfix), zeros -> out of range
    //   column  -1 (element 0 of unit q = 0, left region): the unit is loaded 4 bytes higher (columns 0..3; a 16-byte access
    //           that starts in front of the tensor would be out of range as a whole) and shifted in registers:
    //           (column 1 | 0, column 0, 1, 2)
    //   column  W (element 1 of unit q = 4, right region): the unit is loaded 12 bytes lower (columns W-4..W-1) and turned
    //           round in registers: (W-1, W-2 | 0)
    // The first version redid the reflect / bounds arithmetic per unit and chunk: ~45 VALU next to 32 MFMAs, and a VALU
    // costs 2.5 ... 5 matrix-pipe cycles (tools/micro/mfma_issue.hip).
    unsigned yvo[GY_UNITS]; int ydst[GY_UNITS];
#pragma unroll
    for (int i = 0; i < GY_UNITS; ++i) {
        const int u = tid + i * 256;
        const int co = u >> 3, row = (u >> 2) & 1, q = u & 3;
        yvo[i] = m0 + co < p.M ? (unsigned)((m0 + co) * (int)p.cs + row * p.W + q * 4) * 4u : WINO_OOB;
        ydst[i] = co * GY_STRIDE + row * GC + q * 4;
    }
    unsigned xbase[GX_UNITS]; int xdst[GX_UNITS], rowfix[GX_UNITS], cfix[GX_UNITS], tfix[GX_UNITS];
    const bool is3d = p.cin != p.K;              // virtual channels (time tap, channel); NOT `T > 1`: a one-frame 3x3x3 layer is still 3-D
#pragma unroll
    for (int i = 0; i < GX_UNITS; ++i) {
        const int u = tid + i * 256;
        const int ci = u / 20, rem = u % 20;
        const int r = rem / 5, q = rem % 5;
        const bool live = u < GX_TOTAL && c0 + ci < p.K;
        const int kt = is3d ? (c0 + ci) / p.cin : 1, cch = is3d ? (c0 + ci) - kt * p.cin : c0 + ci;
        xbase[i] = live ? (unsigned)((cch * (int)p.cs + (kt - 1) * HW + (r - 1) * p.W + 4 * q - 1) * 4) : WINO_OOB;
        tfix[i] = (!live || !is3d) ? 0 : (kt == 0 ? 2 * HW * 4 : (kt == 2 ? -2 * HW * 4 : 0));   // frame -1 -> 1, frame T -> T-2
        rowfix[i] = !live ? 0 : (r == 0 ? 2 * p.W * 4 : (r == 3 ? -2 * p.W * 4 : 0));
        cfix[i] = !live ? 0 : (q == 0 ? 4 : (q == 4 ? -12 : 0));       // > 0: holds column -1;  < 0: holds column W
        xdst[i] = u < GX_TOTAL ? ci * GX_STRIDE + r * GX_COLS + q * 4 : -1;
    }
    f32x4 gy[GY_UNITS], gx[GX_UNITS];
    float dbacc[GY_UNITS] = {};
    const bool want_db = nt == 0;                                   // workgroup-uniform
    // Regions are fetched strictly in order rbeg, rbeg + 1, ...: (rx, ry, frame, sample) of the next region are running
    // counters (the first version decomposed the region index with three scalar divisions + a modulo in every chunk -- ~100
    // scalar instructions in front of 16 MFMAs), and a region is always stashed before the next one is fetched, so the
    // border flags the stash needs are those of the last fetch.
    int f_rx, f_ry, f_frm, f_smp;
    {
        f_rx = rbeg % regions_x; const int t = rbeg / regions_x;
        f_ry = t % regions_y; const int img = t / regions_y;
        f_smp = img / p.T; f_frm = img - f_smp * p.T;                 // (sample, frame); T = 1 for the 2-D layers
    }
    bool s_left = false, s_right = false;
    auto fetch = [&](int) {
        const int rx = f_rx, ry = f_ry, smp = f_smp, frm = f_frm;
        if (++f_rx == regions_x) {
            f_rx = 0;
            if (++f_ry == regions_y) {
                f_ry = 0;
                if (++f_frm == p.T) { f_frm = 0; ++f_smp; }
            }
        }
        const int oy0 = ry * GR, ox0 = rx * GC;
        const int ysoff = (int)(((long)smp * p.dy_sn + (long)frm * HW + (long)oy0 * p.W + ox0) * 4);
#pragma unroll
        for (int i = 0; i < GY_UNITS; ++i)
            gy[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsy, yvo[i], ysoff, 0));
        const unsigned roff = (unsigned)(((long)smp * p.x_sn + (long)frm * HW + (long)oy0 * p.W + ox0) * 4);
        const bool top = ry == 0, bot = ry == regions_y - 1, left = rx == 0, right = rx == regions_x - 1; // wave-uniform
        const bool first = is3d && frm == 0, last = is3d && frm == p.T - 1;
        s_left = left; s_right = right;
#pragma unroll
        for (int i = 0; i < GX_UNITS; ++i) {
            unsigned vo = xbase[i] + roff;
            if (left) vo += (unsigned)(cfix[i] > 0 ? cfix[i] : 0);
            if (right) vo += (unsigned)(cfix[i] < 0 ? cfix[i] : 0);
            if (p.reflect) {
                if (top) vo += (unsigned)(rowfix[i] > 0 ? rowfix[i] : 0);
                if (bot) vo += (unsigned)(rowfix[i] < 0 ? rowfix[i] : 0);
                if (first) vo += (unsigned)(tfix[i] > 0 ? tfix[i] : 0);
                if (last) vo += (unsigned)(tfix[i] < 0 ? tfix[i] : 0);
            } else {
                if (top) vo = rowfix[i] > 0 ? WINO_OOB : vo;
                if (bot) vo = rowfix[i] < 0 ? WINO_OOB : vo;
                if (first) vo = tfix[i] > 0 ? WINO_OOB : vo;
                if (last) vo = tfix[i] < 0 ? WINO_OOB : vo;
            }
            gx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, vo, 0, 0));
        }
    };
    auto stash = [&](int, int buf) {
        const bool left = s_left, right = s_right;                                                          // wave-uniform
#pragma unroll
        for (int i = 0; i < GY_UNITS; ++i) {
            float* __restrict__ d = &pY[buf][ydst[i]];
            *reinterpret_cast<float2*>(d) = float2{gy[i][0], gy[i][1]};
            *reinterpret_cast<float2*>(d + 2) = float2{gy[i][2], gy[i][3]};
        }
        // bias gradient: the plain sum of dY.  Each thread stages the same two output channels in every chunk, so it sums
        // its own 16-byte units here (input-channel tile 0 only) and the eight threads of a channel are combined once at
        // the end -- inside the k-steps the sum cost every wave ~30 VALU per step (if-converted, never skipped)
        if (want_db) {
#pragma unroll
            for (int i = 0; i < GY_UNITS; ++i) {
                // one dependent chain per unit, fenced: the SLP vectoriser otherwise pairs the two units' adds into
                // v_pk_add_f32 through eight v_mov shuffles
                float t = gy[i][0] + gy[i][1];
                asm volatile("" : "+v"(t));
                t += gy[i][2];
                t += gy[i][3];
                dbacc[i] += t;
            }
        }
#pragma unroll
        for (int i = 0; i < GX_UNITS; ++i) {
            f32x4 v = gx[i];
            // selects only (the lane masks cfix > 0 / < 0 are loop invariants in SGPRs, combined with the region flags by
            // SALU): branches around these few moves cost more VALU copies than they saved
            const bool el = left & (cfix[i] > 0);       // loaded columns 0..3 -> (column -1, 0, 1, 2)
            const bool er = right & (cfix[i] < 0);      // loaded columns W-4..W-1 -> (column W-1, W, ..)
            const float l0 = p.reflect ? v[1] : 0.f, r0 = v[3], r1 = p.reflect ? v[2] : 0.f;
            const f32x4 w = {el ? l0 : (er ? r0 : v[0]), el ? v[0] : (er ? r1 : v[1]), el ? v[1] : v[2], el ? v[2] : v[3]};
            v = w;
            if (GX_TOTAL % 256 == 0 || xdst[i] >= 0) {
                float* __restrict__ d = &pX[buf][xdst[i]];
                *reinterpret_cast<float2*>(d) = float2{v[0], v[1]};
                *reinterpret_cast<float2*>(d + 2) = float2{v[2], v[3]};
            }
        }
    };

    // row i = wave of the two transforms as wave-uniform coefficients (exact: the factors are 0 / +-1)
    //   A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]:   T = ya * row0 + yb * row1
    //   B^T d,   B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]:   x = d[ra] + xs * d[rb]
    const float ya_c = wave == 3 ? 0.f : 1.f;
    const float yb_c = wave == 0 ? 0.f : (wave == 1 ? 1.f : -1.f);
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float xs_c = wave == 1 ? 1.f : -1.f;
    const int cl = lane & 31, kl = lane >> 5;
    const int yoff = cl * GY_STRIDE + 2 * kl;                       // + q * 32 * GY_STRIDE + 4 * ks (+ GC for row 1)
    const int xoff_a = cl * GX_STRIDE + ra * GX_COLS + 2 * kl;      // + q * 32 * GX_STRIDE + 4 * ks: local column 2t, t = 2ks + kl
    const int xoff_b = cl * GX_STRIDE + rb * GX_COLS + 2 * kl;

    f32x16 acc[4][MI][NI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mi][ni][r] = 0.f;

    struct Raw { f32x2 y0[MI], y1[MI]; f32x2 a0[NI], a1[NI], b0[NI], b1[NI]; };
    struct Frag { float a[MI][4], b[NI][4]; };
    auto load_raw = [&](int buf, int ks, Raw& r) {
#ifdef WG_DIAG_NOLDS        // timing diagnostics only: no LDS reads
        {
            const f32x2 c = {1.f + ks, 2.f + buf};
#pragma unroll
            for (int q = 0; q < MI; ++q) { r.y0[q] = c; r.y1[q] = c; }
#pragma unroll
            for (int q = 0; q < NI; ++q) { r.a0[q] = c; r.a1[q] = c; r.b0[q] = c; r.b1[q] = c; }
            return;
        }
#endif
        const float* __restrict__ py = &pY[buf][0];
        const float* __restrict__ px = &pX[buf][0];
#pragma unroll
        for (int q = 0; q < MI; ++q) {
            const int yo = yoff + q * 32 * GY_STRIDE + 4 * ks;
            r.y0[q] = *reinterpret_cast<const f32x2*>(py + yo);
            r.y1[q] = *reinterpret_cast<const f32x2*>(py + yo + GC);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int xa_ = xoff_a + q * 32 * GX_STRIDE + 4 * ks, xb_ = xoff_b + q * 32 * GX_STRIDE + 4 * ks;
            r.a0[q] = *reinterpret_cast<const f32x2*>(px + xa_);
            r.a1[q] = *reinterpret_cast<const f32x2*>(px + xa_ + 2);
            r.b0[q] = *reinterpret_cast<const f32x2*>(px + xb_);
            r.b1[q] = *reinterpret_cast<const f32x2*>(px + xb_ + 2);
        }
    };
    // Transforms in packed fp32 (the first version: 22 scalar VALU per k-step at NI = 1):
    //   dY: t = ya * row0 + yb * row1, then (t0, t0 + t1, t0 - t1, -t1)
    //   X:  (x0, x1) = a0 + xs * b0, (x2, x3) = a1 + xs * b1, then (x0 - x2, x1 + x2, x2 - x1, x1 - x3)
    auto transform_y = [&](const Raw& r, Frag& f, int q) {
#ifdef WG_DIAG_NOXFORM      // timing diagnostics only (wrong results)
        f.a[q][0] = r.y0[q].x; f.a[q][1] = r.y0[q].y; f.a[q][2] = r.y1[q].x; f.a[q][3] = r.y1[q].y;
        return;
#endif
        const f32x2 t = ya_c * r.y0[q] + yb_c * r.y1[q];
        const f32x2 sd = pk_sum_diff(t);
        f.a[q][0] = t.x; f.a[q][1] = sd.x; f.a[q][2] = sd.y; f.a[q][3] = -t.y;
    };
    auto transform_x = [&](const Raw& r, Frag& f, int q) {
#ifdef WG_DIAG_NOXFORM
        f.b[q][0] = r.a0[q].x; f.b[q][1] = r.a0[q].y; f.b[q][2] = r.b1[q].x; f.b[q][3] = r.b1[q].y;
        return;
#endif
        const f32x2 x01 = r.a0[q] + xs_c * r.b0[q], x23 = r.a1[q] + xs_c * r.b1[q];
        const f32x2 o01 = pk_a(x01, x23), o23 = pk_b(x01, x23);
        f.b[q][0] = o01.x; f.b[q][1] = o01.y; f.b[q][2] = o23.x; f.b[q][3] = o23.y;
    };
    auto transform = [&](const Raw& r, Frag& f) {
#pragma unroll
        for (int q = 0; q < MI; ++q) transform_y(r, f, q);
#pragma unroll
        for (int q = 0; q < NI; ++q) transform_x(r, f, q);
    };
    auto mma_j = [&](const Frag& f, int j) {           // the 2 * NI MFMAs of frequency column j
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[j][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[mi][j], f.b[ni][j], acc[j][mi][ni], 0, 0, 0);
    };
    // One k-step: the MFMAs of step g on the fragments prepared during step g-1, the transform of step g+1 from raw
    // values READ DURING STEP g-1, and the raw reads of step g+2.  The order inside the step is pinned (sched_barrier),
    // since the packed forms are inline asm the scheduler cannot classify: after each MFMA group the LDS reads first (they
    // do not hold up the in-order wave), VALU last -- a VALU behind an MFMA waits for the matrix pipe and delays
    // everything queued behind it (tools/micro/mfma_issue.hip).
#define WG_STEP(RBUF, RKS, FCUR, FNEXT)                                          \
    do {                                                                          \
        Raw rn;                                                                   \
        mma_j(FCUR, 0);                                                           \
        load_raw(RBUF, RKS, rn);                                                  \
        transform_y(rw, FNEXT, 0);                                                \
        __builtin_amdgcn_sched_barrier(0);                                        \
        mma_j(FCUR, 1);                                                           \
        if constexpr (MI == 2) transform_y(rw, FNEXT, 1);                         \
        __builtin_amdgcn_sched_barrier(0);                                        \
        mma_j(FCUR, 2);                                                           \
        _Pragma("unroll") for (int q_ = 0; q_ < NI; ++q_) transform_x(rw, FNEXT, q_); \
        __builtin_amdgcn_sched_barrier(0);                                        \
        mma_j(FCUR, 3);                                                           \
        rw = rn;                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)

    if (nchunks > 0) {
        fetch(rbeg);
        stash(rbeg, 0);
        if (NB == 3 && nchunks > 1) fetch(rbeg + 1);
        __syncthreads();
        Frag fa, fb;
        Raw rw;
        load_raw(0, 0, rw);
        transform(rw, fa);
        load_raw(0, 1, rw);
        // the chunk loop is written out per buffer: with a run-time buffer index every k-step spent four VALU on LDS
        // addresses right behind its first MFMA; as constants they are the offset fields of the reads
#define WG_CHUNK(CUR, NXT)                                                                             \
        {                                                                                              \
            const bool more = chunk + 1 < nchunks;                                                     \
            if (NB == 3) {                                   /* refill loaded during the previous chunk */ \
                if (more) { stash(rbeg + chunk + 1, NXT); if (chunk + 2 < nchunks) fetch(rbeg + chunk + 2); } \
            } else if (more) fetch(rbeg + chunk + 1);        /* lands during k-steps 0 and 1 */        \
            WG_STEP(CUR, 2, fa, fb);                         /* MFMAs of k-step 0, transform of 1, raw reads of 2 */ \
            WG_STEP(CUR, 3, fb, fa);                         /* k-step 1: the last raw reads of buffer CUR */ \
            if (NB == 2 && more) stash(rbeg + chunk + 1, NXT);   /* buffer NXT was last read before the previous barrier */ \
            __syncthreads();                                                                           \
            /* k-steps 2 and 3 + the first two raw sets of the next chunk (last chunk: stale LDS data, transformed and */ \
            /* dropped -- keeps the accumulators out of a conditional path) */                         \
            WG_STEP(NXT, 0, fa, fb);                                                                   \
            WG_STEP(NXT, 1, fb, fa);                                                                   \
        }
        int chunk = 0;
        if (NB == 2) {
            for (; chunk + 1 < nchunks; chunk += 2) {
                WG_CHUNK(0, 1)
                ++chunk;
                WG_CHUNK(1, 0)
                --chunk;
            }
            if (chunk < nchunks) WG_CHUNK(0, 1)
        } else {
            for (; chunk + 2 < nchunks; chunk += 3) {
                WG_CHUNK(0, 1)
                ++chunk;
                WG_CHUNK(1, NB - 1)
                ++chunk;
                WG_CHUNK(NB - 1, 0)
                chunk -= 2;
            }
            if (chunk < nchunks) { WG_CHUNK(0, 1) ++chunk; }
            if (chunk < nchunks) { WG_CHUNK(1, NB - 1) }
        }
#undef WG_CHUNK
    }
#undef WG_STEP
    // ---- slab [split][xi][ci][co]: co innermost, so the four consecutive rows a lane holds in acc[..][4g .. 4g+3] are
    // one 16-byte store (a dword store per accumulator register made the 256 KB epilogue store-issue bound)
    float* __restrict__ out = p.slab + (long)split * 16 * p.M * p.K;
    const bool m_vec = (p.M & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xi = 4 * wave + j;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int ci = c0 + ni * 32 + (lane & 31);
            if (ci >= p.K) continue;
            float* __restrict__ col = out + ((long)xi * p.K + ci) * p.M;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = m0 + mi * 32 + 8 * g + 4 * (lane >> 5);
                    if (m_vec && co + 3 < p.M) {
                        f32x4 v = {acc[j][mi][ni][4 * g], acc[j][mi][ni][4 * g + 1], acc[j][mi][ni][4 * g + 2],
                                   acc[j][mi][ni][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(col + co) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (co + e < p.M) col[co + e] = acc[j][mi][ni][4 * g + e];
                    }
                }
        }
    }
    // ---- bias gradient partial: sum of dY over this workgroup's regions (input-channel tile 0); unit u = tid + i*256
    // belongs to channel u >> 3, so eight consecutive lanes are combined (fixed order) and lane 0 of the group writes
    if (want_db) {
#pragma unroll
        for (int i = 0; i < GY_UNITS; ++i) {
            float v = dbacc[i];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            const int co = (tid + i * 256) >> 3;
            if ((tid & 7) == 0 && m0 + co < p.M) p.dbslab[(long)split * p.M + m0 + co] = v;
        }
    }
}

// Stage 1: u[e] = sum_s slab[s][e] over the 16*M*K elements, fixed order.  A workgroup = 16 float4 elements x 16 split lanes:
// lane sl sums the splits sl, sl + 16, ... (four loads in flight), the 16 partial sums are combined in order through LDS.
// The first form gave every element ONE thread that walked all S splits: with few channels (16*M*K/4 = 4096 float4 at
// 32 x 32 channels) that was 16 workgroups each chasing 768 dependent loads -- ~200 us of a 400 us launch.
__global__ __launch_bounds__(256) void wino_slab_sum_kernel(const f32x4* __restrict__ slab, f32x4* __restrict__ u, long n4, int S) {
    __shared__ f32x4 part[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (long base = (long)blockIdx.x * 16; base < n4; base += (long)gridDim.x * 16) {
        const long i = base + el;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            int s = sl;
            for (; s + 48 < S; s += 64) {
                const f32x4 a = slab[(long)s * n4 + i], b = slab[(long)(s + 16) * n4 + i], c = slab[(long)(s + 32) * n4 + i],
                            d = slab[(long)(s + 48) * n4 + i];
                acc = acc + a; acc = acc + b; acc = acc + c; acc = acc + d;
            }
            for (; s < S; s += 16) acc = acc + slab[(long)s * n4 + i];
        }
        part[sl][el] = acc;
        __syncthreads();
        if (sl == 0 && i < n4) {
            f32x4 t = part[0][el];
#pragma unroll
            for (int k = 1; k < 16; ++k) t = t + part[k][el];
            u[i] = t;
        }
        __syncthreads();
    }
}

// Stage 2: dW[co][ci][3][3] = G^T u G with u[xi][ci][co];  db[co] = sum_splits dbslab
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ usum, const float* __restrict__ dbslab,
                                         float* __restrict__ dW, float* __restrict__ db, int M, int K, int S) {
    const long total = (long)M * K;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i / M), co = (int)(i - (long)ci * M);          // reads coalesced along co
        float u[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) u[xi >> 2][xi & 3] = usum[(long)xi * total + i];
        // G^T u: rows a = 0..2 from i = 0..3 with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
            t[1][j] = 0.5f * (u[1][j] - u[2][j]);
            t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
        }
        float* __restrict__ o = dW + ((long)co * K + ci) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
            o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
            o[a * 3 + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        }
    }
    if (db) {
        // 16 lanes per output channel, lane sl sums the splits sl, sl + 16, ...; fixed-order butterfly over the 16 partial sums.
        // (One thread per channel walking all S splits was 180 us of dependent loads at S = 768 -- the 32-channel layers.)
        const long total16 = (long)M * 16;
        const long stride = (long)gridDim.x * blockDim.x;        // a multiple of 16
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < ((total16 + 15) & ~15L); i += stride) {
            const int m = (int)(i >> 4), sl = (int)(i & 15);
            float acc = 0.f;
            if (m < M)
                for (int s = sl; s < S; s += 16) acc += dbslab[(long)s * M + m];
            acc += __shfl_xor(acc, 8, 16); acc += __shfl_xor(acc, 4, 16); acc += __shfl_xor(acc, 2, 16); acc += __shfl_xor(acc, 1, 16);
            if (sl == 0 && m < M) db[m] = acc;
        }
    }
}

// Stages 1 + 2 in one launch for S <= 64 splits (the >= 128-channel layers; round 3): a workgroup owns 64 output channels of one
// input channel; thread (el, xi) sums ALL splits of frequency xi for the four output channels of float4 `el` in split order
// (four loads in flight), the 16 frequencies of an output meet through LDS, then one thread per output applies G^T (.) G and
// writes the nine taps.  The summed slab never goes to memory and the second launch (46 per step) disappears.  Layers with
// hundreds of splits (<= 64 channels) keep the two-stage form: a thread walking 768 splits is a chain of dependent loads.
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const f32x4* __restrict__ slab, const float* __restrict__ dbslab,
                                                                float* __restrict__ dW, float* __restrict__ db, int M, int K, int S) {
    __shared__ f32x4 us[16][17];
    const int el = threadIdx.x & 15, xi = threadIdx.x >> 4;
    const int m4 = M >> 2;                                   // float4 per (xi, ci) row; M % 4 == 0
    const int cblocks = (m4 + 15) / 16;
    const long n4 = 4L * M * K;                              // float4 per slab = 16 * M * K / 4
    const int item = blockIdx.x;
    const int ci = item / cblocks, c4 = (item - ci * cblocks) * 16 + el;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c4 < m4) {
        const f32x4* __restrict__ q = slab + ((long)xi * K + ci) * m4 + c4;
        int sp = 0;
        for (; sp + 3 < S; sp += 4) {
            const f32x4 a = q[(long)sp * n4], b = q[(long)(sp + 1) * n4], c = q[(long)(sp + 2) * n4], d = q[(long)(sp + 3) * n4];
            acc = acc + a; acc = acc + b; acc = acc + c; acc = acc + d;
        }
        for (; sp < S; ++sp) acc = acc + q[(long)sp * n4];
    }
    us[xi][el] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int e = threadIdx.x >> 2, sub = threadIdx.x & 3;
        const int co = ((item - ci * cblocks) * 16 + e) * 4 + sub;
        if (co < M) {
            float u[4][4];
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k >> 2][k & 3] = us[k][e][sub];
            float t[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
                t[1][j] = 0.5f * (u[1][j] - u[2][j]);
                t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
            }
            float* __restrict__ o = dW + ((long)co * K + ci) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
                o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
                o[a * 3 + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
            }
        }
    }
    // bias gradient: the workgroups of input channel 0 sum dbslab for their 64 output channels (16 lanes per channel, fixed order)
    if (db && ci == 0) {
        const int base = (item - ci * cblocks) * 64;
        for (int r = 0; r < 4; ++r) {
            const int m = base + r * 16 + (threadIdx.x >> 4), sl = threadIdx.x & 15;
            float a = 0.f;
            if (m < M)
                for (int sp = sl; sp < S; sp += 16) a += dbslab[(long)sp * M + m];
            a += __shfl_xor(a, 8, 16); a += __shfl_xor(a, 4, 16); a += __shfl_xor(a, 2, 16); a += __shfl_xor(a, 1, 16);
            if (sl == 0 && m < M) db[m] = a;
        }
    }
}

// Number of region splits the launch will use.  The caller provides slab = (S + 1)*16*M*K floats (S partial slabs + the
// summed one) and dbslab = S*M floats.
// Input-channel tile of the launch: 32 (NI = 1: 64 x 32 channels, 128 accumulators, two workgroups per CU).  The 64 x 64
// tile with one workgroup per CU (NI = 2) was 8...13 % slower on every eligible shape (tools/ab_wino_wgrad.py, first
// version of the kernel) and is no longer instantiated.
static int wino_wg_ni(int M, int K) { (void)M; (void)K; return 1; }

C2M_API int c2m_wino_wgrad_splits(int M, int K, int nimg, int H, int W) {
    const long regions = (long)nimg * (H / GR) * (W / GC);
    const int ni = wino_wg_ni(M, K);
    const int mrows = M <= 32 ? 32 : 64;        // 32-row variant for Cout <= 32 (three workgroups per CU)
    const long tiles = (long)c2m_cdiv(M, mrows) * c2m_cdiv(K, 32 * ni);
    long S = (mrows == 32 ? 768 : (ni == 1 ? 512 : 256)) / tiles;     // one resident round: 256 CUs x (3 | 2 | 1) workgroups
    if (S < 1) S = 1;
    const long maxS = (regions + 15) / 16;      // >= 16 regions (128 tiles) per split
    if (S > maxS) S = maxS;
    if (const char* f = getenv("C2M_WINO_WG_SPLITS")) { const long v = atol(f); if (v > 0) S = v < regions ? v : regions; }
    if (S < 1) S = 1;
    const long per = (regions + S - 1) / S;
    return (int)((regions + per - 1) / per);
}

// shared launcher of the 2-D and the 3x3x3 weight gradient (T = 1, cin = K for 2-D)
static int wino_wgrad_launch(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db,
                             int M, int K, int nimg, int H, int W, int reflect, int T, int cin, void* stream) {
    if (M <= 0 || K <= 0 || nimg <= 0) return 0;
    if (H % GR || W % GC || T <= 0 || nimg % T || cin <= 0 || K % cin) return (int)hipErrorInvalidValue;
    const long ybytes = 4L * nimg * M * H * W, xbytes = 4L * nimg * cin * H * W;
    if (ybytes >= 0x80000000LL || xbytes >= 0x80000000LL) return (int)hipErrorInvalidValue;
    WinoWgP p;
    p.dY = dY; p.X = X; p.slab = slab; p.dbslab = dbslab;
    p.M = M; p.K = K; p.nimg = nimg; p.H = H; p.W = W; p.reflect = reflect;
    p.T = T; p.cin = cin; p.cs = (long)T * H * W;
    p.dy_sn = (long)M * p.cs; p.x_sn = (long)cin * p.cs;
    p.dy_bytes = (unsigned)ybytes; p.x_bytes = (unsigned)xbytes;
    p.regions = nimg * (H / GR) * (W / GC);
    const int S = c2m_wino_wgrad_splits(M, K, nimg, H, W);
    p.per_split = c2m_cdiv(p.regions, S);
    const int ni = wino_wg_ni(M, K);
    const int mrows = M <= 32 ? 32 : 64;
    dim3 grid((unsigned)S * c2m_cdiv(M, mrows) * c2m_cdiv(K, 32 * ni));
    hipStream_t s = (hipStream_t)stream;
    if (mrows == 32) hipLaunchKernelGGL((conv_wino_wgrad_kernel<1, 1, 3>), grid, dim3(256), 0, s, p);
    else             hipLaunchKernelGGL((conv_wino_wgrad_kernel<1, 2, 2>), grid, dim3(256), 0, s, p);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    const long n = 16L * M * K;
    float* usum = slab + (long)S * n;
    static const bool two_stage = getenv("C2M_WINO_WGRAD_TWO_STAGE") != nullptr;      // A/B knob
    if (S <= 64 && (M & 3) == 0 && !two_stage) {
        const int cblocks = ((M >> 2) + 15) / 16;
        hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)(K * cblocks)), dim3(256), 0, s, (const f32x4*)slab, dbslab,
                           dW, db, M, K, S);
        return (int)hipGetLastError();
    }
    if ((n & 3) == 0) {
        hipLaunchKernelGGL(wino_slab_sum_kernel, dim3(c2m_grid(n / 4, 16)), dim3(256), 0, s, (const f32x4*)slab,
                           (f32x4*)usum, n / 4, S);
    } else {
        return (int)hipErrorInvalidValue;       // M*K is a multiple of 4 for every layer the host routes here (checked there)
    }
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(c2m_grid((long)M * K, 256)), dim3(256), 0, s, usum, dbslab, dW, db,
                       M, K, S);
    return (int)hipGetLastError();
}

// 3x3, stride 1, pad 1 (zeros or reflect), H % 2 == 0, W % 16 == 0.  dY [N][M][H][W], X [N][K][H][W] contiguous.
C2M_API int c2m_conv_wino_wgrad(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db,
                                int M, int K, int nimg, int H, int W, int reflect, void* stream) {
    C2M_ENTER();
    return wino_wgrad_launch(dY, X, slab, dbslab, dW, db, M, K, nimg, H, W, reflect, 1, K, stream);
}

// 3x3x3, stride 1, pad 1: dY [N][M][T][H][W], X [N][Cin][T][H][W]; dW comes out as [M][3][Cin][3][3] (time tap ahead of the
// channel: the virtual-channel order of the kernel), the caller permutes it to the native [M][Cin][3][3][3].  slab / dbslab as
// for c2m_conv_wino_wgrad with K = 3 * Cin and nimg = N * T.
C2M_API int c2m_conv_wino_wgrad3d(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db,
                                  int M, int Cin, int N, int T, int H, int W, int reflect, void* stream) {
    C2M_ENTER();
    return wino_wgrad_launch(dY, X, slab, dbslab, dW, db, M, 3 * Cin, N * T, H, W, reflect, T, Cin, stream);
}
