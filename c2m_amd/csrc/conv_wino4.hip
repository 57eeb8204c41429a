// conv_wino4.hip -- Winograd F(4x4, 3x3) convolution for the deep 3x3 stride-1 2-D layers (the same call sites as conv_wino.hip:
// layers/vgg.py:92-137, residual_block.py:13-31,42-71, spade_block.py:47-49 and their data gradients), fp32 on
// v_mfma_f32_32x32x2_f32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 6x6 input tile, g: 3x3 filter, Y: 4x4 outputs     (Toom-Cook points 0, +-3/4, +-3/2, inf)
// 36 multiplies per 16 outputs instead of 144: the contraction over input channels becomes 36 independent GEMMs with 4x fewer
// MFMA FLOPs than the direct form (F(2x2,3x3): 2.25x).  The transforms are no longer exact scalings: with the points 0, +-3/4,
// +-3/2, inf (tools/wino43_matrices.py; B^T and A^T dyadic, G in 1/81, 1/243) the fp32 result differs from a float64 convolution by
// 1.9e-6 ... 4.3e-6 of the tensor's scale on the bench model's layers (tools/bench_wino4.py on the GPU; the classic 0, +-1, +-2
// set measured 0.7e-5 ... 1.7e-5; F(2x2,3x3): 4e-7 ... 1e-6) -- inside the conv tests' 2e-5 / 5e-5 gates with 5x margin.
//
// One workgroup = 512 threads = 8 waves = ONE per CU (two waves per SIMD): 64 output channels x a 16x32 output region (4 x 8
// tiles = the 32 MFMA columns) of one image.  Wave w: output-channel half wm = w & 1, frequency group fg = w >> 1 (frequencies
// 9 fg .. 9 fg + 8): 9 x 16 = 144 accumulator registers.  Per 8-channel chunk:
//   * the 8 x 18 x 34 input patch (LDS row pitch 40: conflict-free transform reads) arrives by LDS-DMA two chunks ahead (three buffers; wave w fetches channel w, the per-lane
//     source offsets -- reflect / bounds resolved once -- sit in a 768-entry LDS table, read once into registers);
//   * thread (channel tk, tile tn, half h) transforms three rows of V = B^T d B of its 6x6 tile into V[xi][k][tile] in LDS
//     (double buffered);
//   * the U = G g G^T fragments come from L2 in a pre-packed order (c2m_wino4_filter_transform), 9 x 16 bytes per lane;
//   * 36 MFMAs per wave.
// Every wave interleaves its share of the next chunk's transform (column pass of one column pair per group of 12 MFMAs, row pass
// at the end) and of the patch DMA with its own MFMAs (sched_group_barrier: 1 MFMA : 2 VALU).  The inverse transform goes through LDS in four passes of 16 output channels: raw accumulators
// [xi][cout][tile] -> one thread per (cout, tile) computes A^T m A, adds the bias, applies the activation and stores four
// 16-byte row segments (or per pixel into Y / Y_interior for the two-target data gradient of reflect-padded layers).
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define W4_OOB 0x80000000u

struct Wino4P {
    const float* U;      // packed filter transform (wino4_filter_kernel)
    const float* X;
    float* Y;
    const float* bias;
    float* Y2;           // optional second target (see conv_wino.hip: interior of a padded-domain data gradient)
    long y2_sn, y2_sc, y2_sh;
    int lo_y, lo_x, ext_y, ext_x;
    int M, K, nimg;
    int Hi, Wi, Ho, Wo;
    int iy0, ix0;
    int reflect;
    long in_sn, out_sn, out_sc, out_sh, out_off;
    int in_sc, in_sh;
    unsigned x_bytes;
    int nchunks, mtiles;
    int act;
    float slope;
    // pad-ring terms of a reflect data gradient computed over the EXACT domain (conv_ring.hip, buffer mode): R [nimg][M][4][r_l],
    // added to rows 1 / Ho-2 (sides 0 / 1, indexed by column) and columns 1 / Wo-2 (sides 2 / 3, indexed by row); or NULL
    const float* R;
    int r_l;
};

// W4_FUSED (default): every wave interleaves the transform of chunk n + 1 and the patch DMA of chunk n + 2 with its OWN MFMAs of
// chunk n.  -DW4_SKEW builds the first form (two phases, opposite order in the two waves of a SIMD), kept for A/B runs: next to
// a partner wave that streams MFMAs a wave's VALU / LDS phase takes 3x as long (tools/trace_wino4.py), so the phases did not overlap.
#if !defined(W4_SKEW) && !defined(W4_FUSED)
#define W4_FUSED 1
#endif

constexpr int W4_CK = 8;                              // channels per chunk
constexpr int W4_TH = 4, W4_TW = 8;                   // tiles per region: 16 x 32 outputs
constexpr int W4_OR = 4 * W4_TH, W4_OC = 4 * W4_TW;   // output region rows / cols
constexpr int W4_PH = W4_OR + 2, W4_PW = W4_OC + 2;   // 18 x 34 input patch
// LDS image of a channel's patch: row pitch 40, channel stride 722 floats.  A transform read (ds_read_b64, lane = (channel tk of 2,
// tile (ty, tx))) hits bank pair (361 tk + 80 ty + 2 tx + const) mod 32: tx -> the 8 even pairs of a half, ty -> alternating halves,
// tk -> odd pairs: two lanes per bank pair, the minimum for 64 x 8 bytes.  The dense 34-pitch / 640-stride image put EIGHT lanes on
// a pair (32 cycles per read instead of 4-8): 18 reads x 8 waves kept the LDS busier than the matrix pipe (round 3: 280 -> see DESIGN).
constexpr int W4_PITCH = 40;
constexpr int W4_PPOS = W4_PH * W4_PITCH;             // 720 LDS positions per channel (34 of each 40 are patch columns)
constexpr int W4_ROWS = (W4_PPOS + 63) / 64;          // 12 DMA rows of 64 positions (the last one: 16 live lanes)
constexpr int W4_PCS = 722;                           // channel stride of the patch in LDS (half of it odd: see above)
constexpr int W4_PBUF = W4_CK * W4_PCS + 48;          // one patch buffer (floats); + the dead lanes of the last channel's last row
constexpr int W4_TAB = W4_ROWS * 64;                  // entries of the source-offset table
constexpr int W4_VBUF = 36 * W4_CK * 32;              // one V buffer (floats)

// Interpolation points 0, +-3/4, +-3/2, inf (tools/wino43_matrices.py): every coefficient of B^T and A^T is dyadic (exact in fp32), and
// the fp32 error is a quarter of the classic 0, +-1, +-2, inf set's (tools/wino43_error.py; measured on the GPU: see DESIGN 5.5).
// the 6-point transform B^T (same for rows and columns):
//   [81/64 0 -45/16 0 1 0; 0 -27/16 -9/4 3/4 1 0; 0 27/16 -9/4 -3/4 1 0; 0 -27/32 -9/16 3/2 1 0; 0 27/32 -9/16 -3/2 1 0; 0 81/64 0 -45/16 0 1]
#define W4_BT(o0, o1, o2, o3, o4, o5, x0, x1, x2, x3, x4, x5) \
    do {                                                      \
        const float a_ = (x4) - 2.25f * (x2), b_ = 0.75f * (x3) - 1.6875f * (x1);          \
        const float c_ = (x4) - 0.5625f * (x2), e_ = 1.5f * (x3) - 0.84375f * (x1);        \
        o0 = 1.265625f * (x0) - 2.8125f * (x2) + (x4);        \
        o1 = a_ + b_; o2 = a_ - b_; o3 = c_ + e_; o4 = c_ - e_; \
        o5 = 1.265625f * (x1) - 2.8125f * (x3) + (x5);        \
    } while (0)

// Phase timestamps of workgroup 0 (tuning builds only: python -m c2m_amd.build w4trace -DW4_TRACE; tools/trace_wino4.py): s_memtime at
// four points of the first 16 intervals of every wave, kept in LDS (a global store would count in vmcnt) and copied out at the end.
#ifdef W4_TRACE
__device__ unsigned long long w4_trace_buf[8 * 16 * 4 + 8 * 2];
#define W4_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && n < 16) sTr[(wave * 16 + n) * 4 + (i)] = clock64(); } while (0)
C2M_API int c2m_wino4_trace_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(w4_trace_buf), sizeof(w4_trace_buf));
}
#else
#define W4_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(const Wino4P p) {
    // ONE LDS block with the DMA targets first: the LDS address of a buffer_load ... lds travels in M0, and only destinations
    // below 64 KB behaved (with the patches behind the 72 KB of V -- addresses 0x12000 ... 0x21000 -- results were intermittently
    // wrong and a launch faulted)
    __shared__ __attribute__((aligned(16))) float smem[3 * W4_PBUF + W4_TAB + 2 * W4_VBUF];
#ifdef W4_TRACE
    __shared__ unsigned long long sTr[8 * 16 * 4 + 8 * 2];
    const unsigned long long t_begin = clock64();
#endif
#define sP (smem)                                              /* input patches [buf][k][18][34] (+pad): 60 KB, LDS-DMA two chunks ahead */
#define sVo (reinterpret_cast<unsigned*>(smem + 3 * W4_PBUF))  /* per patch position: byte offset inside X or W4_OOB */
#define sV (smem + 3 * W4_PBUF + W4_TAB)                       /* V[buf][xi][k][tile]; reused by the epilogue */
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, fg = wave >> 1;
    // opposite phase order for the two waves of a SIMD, whichever way the hardware pairs them: (w, w + 4) or (w, w + 1)
    const bool t_first = (wave ^ (wave >> 2)) & 1;
    const int regions_x = (p.Wo + W4_OC - 1) / W4_OC, regions_y = (p.Ho + W4_OR - 1) / W4_OR;
    // XCD-aware work order (conv_wino.hip): item w = region * mtiles + mt, XCD x owns a contiguous range of items
    int rb, mt;
    {
        const unsigned L = blockIdx.x, total = gridDim.x;
        const unsigned q = total >> 3, r = total & 7u, x = L & 7u, j = L >> 3;
        const unsigned w = x * q + (x < r ? x : r) + j;
        mt = (int)(w % (unsigned)p.mtiles); rb = (int)(w / (unsigned)p.mtiles);
    }
    const int rx = rb % regions_x; rb /= regions_x;
    const int ry = rb % regions_y; const int img = rb / regions_y;
    const int oy0 = ry * W4_OR, ox0 = rx * W4_OC;
    const int nchunks = p.nchunks;

    // ---- patch source offsets (fixed over the K loop)
    {
        const unsigned img_byte = (unsigned)(img * (int)p.in_sn) * 4u;
        for (int pos = tid; pos < W4_TAB; pos += 512) {
            const int r = pos / W4_PITCH, c = pos % W4_PITCH;
            int iy = oy0 + p.iy0 + r, ix = ox0 + p.ix0 + c;
            if (p.reflect) {
                iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
                ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
            }
            const bool ok = pos < W4_PPOS && c < W4_PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            sVo[pos] = ok ? img_byte + (unsigned)(iy * p.in_sh + ix) * 4u : W4_OOB;
        }
    }
    __syncthreads();

    const unsigned long xaddr = (unsigned long)p.X;
    const u32x4 rs = {(unsigned)xaddr, (unsigned)(xaddr >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const unsigned sp_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)&smem[0];
    // wave w fetches channel w of a chunk: twelve DMA rows of 64 positions.  A channel past K gets a zero-record descriptor (the
    // scalar offset is outside the hardware's range check); lanes past position 611 and positions outside the image carry an
    // out-of-range voffset and write zeros into the channel's own padding / halo.
    // this lane's ten source offsets, read ONCE: fetched from the table in front of every DMA instruction they cost an LDS round
    // trip each (ds_read -> wait -> DMA, ten times in a row at the start of every interval, in front of the matrix phase)
    unsigned pvo[W4_ROWS];
#pragma unroll
    for (int dr = 0; dr < W4_ROWS; ++dr) pvo[dr] = sVo[dr * 64 + lane];
    auto load_patch = [&](int chunk, int buf, const int dr0 = 0, const int dr1 = W4_ROWS) __attribute__((always_inline)) {
#ifdef W4_ZERO_RECORDS
        const int ch = chunk * W4_CK + wave;
        u32x4 rsk = rs;
        rsk[2] = (chunk < nchunks && ch < p.K) ? p.x_bytes : 0u;
#else
        // channels past K (last chunk, and the two chunks fetched past the end) re-read the LAST real channel: its values meet zero
        // rows of U or a patch nobody transforms into a used V -- no descriptor is ever modified, every address is inside X
        int ch = chunk * W4_CK + wave;
        ch = ch < p.K ? ch : p.K - 1;
        const u32x4 rsk = rs;
#endif
        const int soff = ch * p.in_sc * 4;
#pragma unroll
        for (int dr = dr0; dr < dr1; ++dr) {
            const unsigned vo = pvo[dr];
            const unsigned dst = sp_lds + (unsigned)((buf * W4_PBUF + wave * W4_PCS + dr * 64) * 4);
            // the last row's lanes past position 719 would write zeros into the NEXT channel's first positions: switched off
            if (dr * 64 + 64 <= W4_PPOS || dr * 64 + lane < W4_PPOS)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(vo), "s"(rsk), "s"(soff) : "memory");
        }
    };
    // ---- U fragments: Upack[chunk][mt][wave][f][lane][kk]: every load of a wave reads 1 KB contiguous
    const unsigned long uaddr = (unsigned long)p.U;
    const u32x4 urs = {(unsigned)uaddr, (unsigned)(uaddr >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    const unsigned uvo = (unsigned)((((mt * 8 + wave) * 9) * 64 + lane) * 16);
    const unsigned ustride_b = (unsigned)p.mtiles * 8u * 9u * 64u * 16u;       // bytes per chunk
    f32x4 ua[9];
    auto load_u = [&](int chunk) __attribute__((always_inline)) {
        const int s0 = (int)((unsigned)(chunk < nchunks ? chunk : 0) * ustride_b), s1 = s0 + 4096, s2 = s0 + 8192;
#pragma unroll
        for (int f = 0; f < 9; ++f)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                         : "=&v"(ua[f]) : "v"(uvo), "s"(urs), "s"(f < 4 ? s0 : (f < 8 ? s1 : s2)), "n"((f & 3) * 1024) : "memory");
    };

    f32x16 acc[9];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ---- input transform role: channel tk, tile tn = (ty, tx), rows 3h .. 3h + 2 of V
    const int item = tid & 255, th = __builtin_amdgcn_readfirstlane(tid >> 8);      // th is wave-uniform (waves 0-3 / 4-7)
    const int tk = item >> 5, tn = item & 31;
    const int pbase = tk * W4_PCS + (4 * (tn >> 3)) * W4_PITCH + 4 * (tn & 7);
    // (tools/trace_wino4.py: next to a partner wave that streams MFMAs this transform takes 3 300-3 600 cycles instead of the 1 200 it
    // takes next to an idle one -- s_setprio 3 around it changes nothing -- which is why W4_FUSED below interleaves each wave's
    // transform with its OWN MFMAs instead of running the two phases of a SIMD's waves in opposite order.)
    auto transform = [&](int pb, int vb) __attribute__((always_inline)) {
        const float* __restrict__ src = sP + pb * W4_PBUF + pbase;
        float t[3][6];
#pragma unroll
        for (int cp = 0; cp < 3; ++cp) {                      // column pairs
            f32x2 d[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = *reinterpret_cast<const f32x2*>(src + r * W4_PITCH + 2 * cp);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
                const int c = 2 * cp + e;
                if (th == 0) {
                    const float a_ = d4 - 2.25f * d2, b_ = 0.75f * d3 - 1.6875f * d1;
                    t[0][c] = 1.265625f * d0 - 2.8125f * d2 + d4; t[1][c] = a_ + b_; t[2][c] = a_ - b_;
                } else {
                    const float c_ = d4 - 0.5625f * d2, e_ = 1.5f * d3 - 0.84375f * d1;
                    t[0][c] = c_ + e_; t[1][c] = c_ - e_; t[2][c] = 1.265625f * d1 - 2.8125f * d3 + d5;
                }
            }
        }
        float* __restrict__ v = sV + vb * W4_VBUF + tk * 32 + tn;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float o0, o1, o2, o3, o4, o5;
            W4_BT(o0, o1, o2, o3, o4, o5, t[a][0], t[a][1], t[a][2], t[a][3], t[a][4], t[a][5]);
            const int xi0 = 6 * (3 * th + a);
            v[(xi0 + 0) * W4_CK * 32] = o0; v[(xi0 + 1) * W4_CK * 32] = o1; v[(xi0 + 2) * W4_CK * 32] = o2;
            v[(xi0 + 3) * W4_CK * 32] = o3; v[(xi0 + 4) * W4_CK * 32] = o4; v[(xi0 + 5) * W4_CK * 32] = o5;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Matrix phase of chunk n with the U fragments of chunk n + 1 re-loaded BEHIND their last use: three groups of three
    // frequencies (three independent accumulator chains per group keep dependent MFMAs 3 x 64 cycles apart); a group's U registers
    // are re-loaded as soon as its twelve MFMAs are issued, so the first loads have ~1 500 matrix cycles + the other phase to land
    // instead of all nine being issued behind the whole phase (transform-first waves had only their transform to hide them).
    auto mma = [&](int cur, int next_chunk) __attribute__((always_inline)) {
        const float* __restrict__ vb = sV + cur * W4_VBUF + (9 * fg * W4_CK + (lane >> 5)) * 32 + (lane & 31);
        const int s0 = (int)((unsigned)(next_chunk < nchunks ? next_chunk : 0) * ustride_b), s1 = s0 + 4096, s2 = s0 + 8192;
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3) {
            float b[3][4];
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) b[f][kk] = vb[((3 * g3 + f) * W4_CK + 2 * kk) * 32];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const f32x4 u = ua[3 * g3 + f];
                    const float av = kk == 0 ? u.x : (kk == 1 ? u.y : (kk == 2 ? u.z : u.w));
                    acc[3 * g3 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[f][kk], acc[3 * g3 + f], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 3 * g3; f < 3 * g3 + 3; ++f)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                             : "=&v"(ua[f]) : "v"(uvo), "s"(urs), "s"(f < 4 ? s0 : (f < 8 ? s1 : s2)), "n"((f & 3) * 1024) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    };

#ifdef W4_FUSED
    // Matrix phase of chunk n with the transform of chunk n + 1 INSIDE it: three groups of 12 MFMAs, each interleaved with the
    // column pass of one column pair of this thread's tile (5 raw rows: rows 1-4 feed the +- pair of output rows, row 0 / 5 the
    // single one; th selects coefficients and operands, no branch -- sched_group_barrier only orders within a basic block) and
    // followed by the re-load of its three U registers; the row pass + 18 V stores close the interval.
    const float cA = th ? 0.5625f : 2.25f, cB3 = th ? 1.5f : 0.75f, cB1 = th ? 0.84375f : 1.6875f;
    const int xrow = th ? 5 : 0, r_single = th ? 5 : 0, r_plus = th ? 3 : 1, r_minus = th ? 4 : 2;
    auto fused = [&](int cur, int next_chunk, int pb, int dbuf) __attribute__((always_inline)) {
        const float* __restrict__ vb = sV + cur * W4_VBUF + (9 * fg * W4_CK + (lane >> 5)) * 32 + (lane & 31);
        const float* __restrict__ src = sP + pb * W4_PBUF + pbase;
        const int s0 = (int)((unsigned)(next_chunk < nchunks ? next_chunk : 0) * ustride_b), s1 = s0 + 4096, s2 = s0 + 8192;
        float ts[6], tp[6], tm[6];                               // single / plus / minus output rows of the column pass
        float b[3][4];
        f32x2 d1, d2, d3, d4, dx;
        // the 17 LDS reads of group g: 12 B fragments + 5 raw rows of column pair g
        auto reads = [&](const int g3) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) b[f][kk] = vb[((3 * g3 + f) * W4_CK + 2 * kk) * 32];
            d1 = *reinterpret_cast<const f32x2*>(src + 1 * W4_PITCH + 2 * g3); d2 = *reinterpret_cast<const f32x2*>(src + 2 * W4_PITCH + 2 * g3);
            d3 = *reinterpret_cast<const f32x2*>(src + 3 * W4_PITCH + 2 * g3); d4 = *reinterpret_cast<const f32x2*>(src + 4 * W4_PITCH + 2 * g3);
            dx = *reinterpret_cast<const f32x2*>(src + xrow * W4_PITCH + 2 * g3);
        };
        // VMEM order of an interval: D0-5, U0-2, D6-11, U3-5, U6-8 (D = DMA rows of patch n + 2, U = fragments of chunk n + 1).
        // A group's U registers were re-loaded in the PREVIOUS interval: 12 / 12 / 18 younger requests may be in flight
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int f = 0; f < 3; ++f) asm volatile("" : "+v"(ua[f]));
        reads(0);
#pragma unroll
        for (int g3 = 0; g3 < 3; ++g3) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int f = 0; f < 3; ++f) {
                    const f32x4 u = ua[3 * g3 + f];
                    const float av = kk == 0 ? u.x : (kk == 1 ? u.y : (kk == 2 ? u.z : u.w));
                    acc[3 * g3 + f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[f][kk], acc[3 * g3 + f], 0, 0, 0);
                }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = 2 * g3 + e;
                const float q0 = th ? d1[e] : dx[e], q2 = th ? d3[e] : d2[e], q4 = th ? dx[e] : d4[e];
                ts[c] = 1.265625f * q0 - 2.8125f * q2 + q4;
                const float a_ = d4[e] - cA * d2[e], b_ = cB3 * d3[e] - cB1 * d1[e];
                tp[c] = a_ + b_; tm[c] = a_ - b_;
            }
            // 12 MFMAs, ~2 VALU behind each (group 0's LDS reads sit in front of them; the later groups' were issued before the
            // VMEM block below, whose issue time covers their latency)
            if (g3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 17, 0);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // next group's LDS reads first, then half of the patch DMA of chunk n + 2 and this group's U re-load (in front of the whole
            // phase the 96 DMA instructions of a workgroup kept the matrix pipes idle for 500-1 000 cycles per interval)
            if (g3 < 2) reads(g3 + 1);
            if (g3 < 2) load_patch(next_chunk + 1, dbuf, 6 * g3, 6 * g3 + 6);
#pragma unroll
            for (int f = 3 * g3; f < 3 * g3 + 3; ++f)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"
                             : "=&v"(ua[f]) : "v"(uvo), "s"(urs), "s"(f < 4 ? s0 : (f < 8 ? s1 : s2)), "n"((f & 3) * 1024) : "memory");
            if (g3 == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (g3 == 1) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            if (g3 < 2) {
#pragma unroll
                for (int f = 3 * g3 + 3; f < 3 * g3 + 6; ++f) asm volatile("" : "+v"(ua[f]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float* __restrict__ v = sV + (cur ^ 1) * W4_VBUF + tk * 32 + tn;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float* t = a == 0 ? ts : (a == 1 ? tp : tm);
            const int row = a == 0 ? r_single : (a == 1 ? r_plus : r_minus);
            float o0, o1, o2, o3, o4, o5;
            W4_BT(o0, o1, o2, o3, o4, o5, t[0], t[1], t[2], t[3], t[4], t[5]);
            float* __restrict__ vr = v + 6 * row * W4_CK * 32;
            vr[0] = o0; vr[1 * W4_CK * 32] = o1; vr[2 * W4_CK * 32] = o2;
            vr[3 * W4_CK * 32] = o3; vr[4 * W4_CK * 32] = o4; vr[5 * W4_CK * 32] = o5;
        }
    };
#endif

    // ---- prologue: patches 0 and 1, U(0), V(0)
    load_patch(0, 0);
    load_patch(1, 1);
    load_u(0);
    asm volatile("s_waitcnt vmcnt(21)" ::: "memory");         // patch 0 has landed (patch 1 + U(0) may still be in flight)
    __syncthreads();
    transform(0, 0);
#ifdef W4_FUSED
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // patch 1 and U(0) complete: the loop's counted waits assume a steady state
#else
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");          // patch 1 (this wave's part) complete before the first barrier
#endif
    // Interval n: MFMAs of chunk n on V(n) / U(n); transform of chunk n + 1 (patch (n+1) % 3 -> V buffer (n+1) & 1); DMA of patch
    // n + 2 issued first so that it has the whole interval to land.  VMEM order per wave: DMA(n+2) x 12, later U(n+1) x 9.
    //   before the MFMAs:   vmcnt(12) -- U(n), issued in the previous interval, complete; the 12 DMA rows may be in flight
    //   before the barrier: vmcnt(9)  -- DMA(n+2) complete; the 9 U loads may be in flight
    int pn = 1;                                               // patch buffer of chunk n + 1
#ifdef W4_PRIO
    // MI355X_MICROARCH.md, "Two waves per SIMD", item 4: the second-dispatched half of an 8-wave workgroup loses every VALU
    // arbitration to its older SIMD partner; one static s_setprio 1 for that half (no per-phase flips) -- tuning build, A/B below
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    for (int n = 0; n < nchunks; ++n) {
        __syncthreads();                                      // V(n) and patch(n+1) complete; V(n-1) and patch(n) are free
        W4_STAMP(0);
        const int cur = n & 1;
        const int dbuf = pn == 2 ? 0 : pn + 1;
#ifndef W4_FUSED
        load_patch(n + 2, dbuf);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // one definition point for the accumulators and the U registers (two copies of the interval under an if / else made the
        // register allocator spill the U tuples at the join); only the transform is placed before or after the matrix phase
#ifdef W4_FUSED
        W4_STAMP(1);
        fused(cur, n + 1, pn, dbuf);
        W4_STAMP(2);
        W4_STAMP(3);
#else
        if (t_first) transform(pn, cur ^ 1);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll
        for (int f = 0; f < 9; ++f) asm volatile("" : "+v"(ua[f]));
        W4_STAMP(1);
        mma(cur, n + 1);
        W4_STAMP(2);
        if (!t_first) transform(pn, cur ^ 1);
        W4_STAMP(3);
#endif
#ifdef W4_FUSED
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // DMA(n+2) complete; U3-8 of chunk n + 1 may be in flight
#else
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
#endif
        pn = pn == 2 ? 0 : pn + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing of this workgroup may still write its LDS
    // The U loads of the last interval are still in flight when the loop exits and nothing reads them: without a use BEHIND the wait
    // the allocator hands their registers to the epilogue's values, which the late-returning loads then overwrite (store addresses
    // among them -- the intermittent far faults of the first version)
#pragma unroll
    for (int f = 0; f < 9; ++f) asm volatile("" :: "v"(ua[f]));
    __syncthreads();

    // ---- inverse transform: four passes of 16 output channels through LDS
    float* __restrict__ sR = sV;                              // [xi][16 cout][32 tiles]
    const int etile = tid & 31, ecl = tid >> 5;               // epilogue role: one (cout, tile) per thread
    const int ety = etile >> 3, etx = etile & 7;
    const int oyb = oy0 + 4 * ety, oxb = ox0 + 4 * etx;
    // the four passes' bias values up front: loaded inside a pass, the ~2 000-cycle round trip sat in front of its stores four times
    float bias4[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int co = mt * 64 + (pass >> 1) * 32 + (pass & 1) * 16 + ecl;
        bias4[pass] = (p.bias && co < p.M) ? p.bias[co] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int pwm = pass >> 1, hf = pass & 1;
        __syncthreads();
        if (wm == pwm) {
#pragma unroll
            for (int f = 0; f < 9; ++f)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);
                    sR[((9 * fg + f) * 16 + rowl) * 32 + (lane & 31)] = hf ? acc[f][8 + rr] : acc[f][rr];
                }
        }
        __syncthreads();
        const int cout = mt * 64 + pwm * 32 + hf * 16 + ecl;
        float y[4][4];
        {
            float s[6][4];                                     // m A  (columns 6 -> 4)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float* __restrict__ q = sR + ((6 * i) * 16 + ecl) * 32 + etile;
                const float m0 = q[0], m1 = q[16 * 32], m2 = q[2 * 16 * 32], m3 = q[3 * 16 * 32], m4 = q[4 * 16 * 32], m5 = q[5 * 16 * 32];
                const float pp = m1 + m2, nn = m1 - m2, PP = m3 + m4, NN = m3 - m4;
                // A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1]
                s[i][0] = m0 + pp + PP; s[i][1] = 0.75f * nn + 1.5f * NN; s[i][2] = 0.5625f * pp + 2.25f * PP;
                s[i][3] = 0.421875f * nn + 3.375f * NN + m5;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {                      // A^T (rows 6 -> 4)
                const float pp = s[1][c] + s[2][c], nn = s[1][c] - s[2][c], PP = s[3][c] + s[4][c], NN = s[3][c] - s[4][c];
                y[0][c] = s[0][c] + pp + PP; y[1][c] = 0.75f * nn + 1.5f * NN; y[2][c] = 0.5625f * pp + 2.25f * PP;
                y[3][c] = 0.421875f * nn + 3.375f * NN + s[5][c];
            }
        }
        if (cout < p.M) {
            const float bb = bias4[pass];
            const float* __restrict__ rb = p.R ? p.R + ((long)img * p.M + cout) * 4 * p.r_l : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oyb + r;
                if (oy >= p.Ho) continue;
                if (rb) {                                      // uniform: the ring terms of the reflect data gradient
                    if (oy == 1 || oy == p.Ho - 2) {
                        const float* __restrict__ rr = rb + (oy == 1 ? 0 : p.r_l) + oxb;
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (oxb + c < p.Wo) y[r][c] += rr[c];
                    }
                    if (oxb <= 1 && oxb + 3 >= 1) {
                        const float v = rb[2 * p.r_l + oy];
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (oxb + c == 1) y[r][c] += v;
                    }
                    if (oxb <= p.Wo - 2 && oxb + 3 >= p.Wo - 2) {
                        const float v = rb[3 * p.r_l + oy];
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (oxb + c == p.Wo - 2) y[r][c] += v;
                    }
                }
                if (p.act == C2M_ACT_NONE) {                   // one uniform branch per row instead of a switch per element
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[r][c] += bb;
                } else if (p.act == C2M_ACT_LRELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float v = y[r][c] + bb; y[r][c] = v > 0.f ? v : p.slope * v; }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) y[r][c] = c2m_act(y[r][c] + bb, p.act, p.slope);
                }
                float* __restrict__ yb0 = p.Y + p.out_off + (long)img * p.out_sn + (long)cout * p.out_sc + (long)oy * p.out_sh + oxb;
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                const f32x4u vv = {y[r][0], y[r][1], y[r][2], y[r][3]};
                const int yi = oy - p.lo_y, xi0 = oxb - p.lo_x;
                const bool row_in = p.Y2 && (unsigned)yi < (unsigned)p.ext_y;
                // two-target launches: a 4-pixel group that lies entirely inside the interior goes to Y_interior with one 16-byte
                // store, one entirely outside it to Y (the pad ring); only groups that straddle the interior's edge go per pixel
                const bool all_in = row_in && xi0 >= 0 && xi0 + 3 < p.ext_x;
                const bool all_out = !row_in || xi0 + 3 < 0 || xi0 >= p.ext_x;
                if (oxb + 3 < p.Wo && (!p.Y2 || all_out)) {
                    *reinterpret_cast<f32x4u*>(yb0) = vv;
                } else if (oxb + 3 < p.Wo && all_in) {
                    *reinterpret_cast<f32x4u*>(p.Y2 + (long)img * p.y2_sn + (long)cout * p.y2_sc + (long)yi * p.y2_sh + xi0) = vv;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (oxb + c >= p.Wo) continue;
                        const int xi = oxb + c - p.lo_x;
                        if (row_in && (unsigned)xi < (unsigned)p.ext_x)
                            p.Y2[(long)img * p.y2_sn + (long)cout * p.y2_sc + (long)yi * p.y2_sh + xi] = y[r][c];
                        else
                            yb0[c] = y[r][c];
                    }
                }
            }
        }
    }
#ifdef W4_TRACE
    if (blockIdx.x == 0 && lane == 0) { sTr[8 * 16 * 4 + wave * 2] = t_begin; sTr[8 * 16 * 4 + wave * 2 + 1] = clock64(); }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = tid; i < 8 * 16 * 4 + 8 * 2; i += 512) w4_trace_buf[i] = sTr[i];
#endif
}

#undef sP
#undef sVo
#undef sV

// Filter transform U = G g G^T (6x6), G = [[64/81,0,0],[-128/243,-32/81,-8/27],[-128/243,32/81,-8/27],[32/243,16/81,8/27],[32/243,-16/81,8/27],[0,0,1]],
// written in the fragment order the kernel reads:
//   Upack[chunk][mt][wave = (fg << 1) | wm][f][lane][kk] = U[xi = 9 fg + f][m = mt*64 + wm*32 + (lane & 31)][c = chunk*8 + 2kk + (lane >> 5)]
// dgrad as in wino_filter_kernel (conv_wino.hip): m = input channel, c = output channel, g = w[c][m] rotated by 180 degrees.
__global__ void wino4_filter_kernel(const float* __restrict__ w, float* __restrict__ up, int M, int K, int Cin_native,
                                    int dgrad, int mtiles, long pairs) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i & 3); long r = i >> 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int wm = (int)(r & 1); r >>= 1;
        const int mt = (int)(r % mtiles); const int chunk = (int)(r / mtiles);
        const int m = mt * 64 + wm * 32 + (lane & 31), c = chunk * W4_CK + 2 * kk + (lane >> 5);
        float u[6][6];
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) u[a][b2] = 0.f;
        if (m < M && c < K) {
            const float* __restrict__ g = dgrad ? w + ((long)c * Cin_native + m) * 9 : w + ((long)m * Cin_native + c) * 9;
            float gg[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 3; ++b2) gg[a][b2] = dgrad ? g[(2 - a) * 3 + (2 - b2)] : g[a * 3 + b2];
            float row[6][3];                                   // G g
#pragma unroll
            for (int b2 = 0; b2 < 3; ++b2) {
                const float g0 = gg[0][b2], g1 = gg[1][b2], g2 = gg[2][b2];
                row[0][b2] = (64.f / 81.f) * g0;
                row[1][b2] = -(128.f / 243.f) * g0 - (32.f / 81.f) * g1 - (8.f / 27.f) * g2;
                row[2][b2] = -(128.f / 243.f) * g0 + (32.f / 81.f) * g1 - (8.f / 27.f) * g2;
                row[3][b2] = (32.f / 243.f) * g0 + (16.f / 81.f) * g1 + (8.f / 27.f) * g2;
                row[4][b2] = (32.f / 243.f) * g0 - (16.f / 81.f) * g1 + (8.f / 27.f) * g2;
                row[5][b2] = g2;
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {                      // (G g) G^T
                const float r0 = row[a][0], r1 = row[a][1], r2 = row[a][2];
                u[a][0] = (64.f / 81.f) * r0;
                u[a][1] = -(128.f / 243.f) * r0 - (32.f / 81.f) * r1 - (8.f / 27.f) * r2;
                u[a][2] = -(128.f / 243.f) * r0 + (32.f / 81.f) * r1 - (8.f / 27.f) * r2;
                u[a][3] = (32.f / 243.f) * r0 + (16.f / 81.f) * r1 + (8.f / 27.f) * r2;
                u[a][4] = (32.f / 243.f) * r0 - (16.f / 81.f) * r1 + (8.f / 27.f) * r2;
                u[a][5] = r2;
            }
        }
        const long base = ((long)chunk * mtiles + mt) * 8;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 6; ++b2) {
                const int xi = 6 * a + b2, fgi = xi / 9, f = xi % 9;
                up[((((base + ((fgi << 1) | wm)) * 9 + f) * 64 + lane) << 2) + kk] = u[a][b2];
            }
    }
}

C2M_API long c2m_wino4_upack_floats(int M, int K) {
    return 36L * (c2m_cdiv(M, 64) * 64L) * (c2m_cdiv(K, W4_CK) * (long)W4_CK);
}

// w: native [Cout][Cin][3][3].  dgrad = 0: M = Cout, K = Cin;  dgrad = 1: M = Cin, K = Cout.
C2M_API int c2m_wino4_filter_transform(const float* w, float* upack, int Cout, int Cin, int dgrad, void* stream) {
    C2M_ENTER();
    const int M = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
    if (M <= 0 || K <= 0) return 0;
    const long total = c2m_wino4_upack_floats(M, K);
    if (total * 4 >= 0xffffffffL) return (int)hipErrorInvalidValue;         // the kernel addresses U through a 4 GB buffer record
    hipLaunchKernelGGL(wino4_filter_kernel, dim3(c2m_grid(total / 36, 256)), dim3(256), 0, (hipStream_t)stream, w, upack, M,
                       K, Cin, dgrad, c2m_cdiv(M, 64), total / 36);
    return (int)hipGetLastError();
}

// regions (16 x 32 outputs) per image
C2M_API int c2m_wino4_regions(int Ho, int Wo) { return c2m_cdiv(Ho, W4_OR) * c2m_cdiv(Wo, W4_OC); }

// geom[]: the 2-D entries of c2m_conv_wino (0 .. 24); 3x3x3 layers (geom[29] != 0) and the temporal pair table are refused.
C2M_API int c2m_conv_wino4(const float* upack, const float* X, float* Y, float* Y_interior, const float* bias,
                           const int64_t* g, int act, float slope, void* stream) {
    C2M_ENTER();
    Wino4P p;
    p.U = upack; p.X = X; p.Y = Y; p.bias = bias;
    p.Y2 = Y_interior; p.y2_sn = p.y2_sc = p.y2_sh = 0; p.lo_y = p.lo_x = p.ext_y = p.ext_x = 0;
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
    if (g[C2M_WG_NKT] != 0 || g[C2M_WG_PTAB] != 0) return (int)hipErrorInvalidValue;
    p.R = (const float*)(uintptr_t)g[C2M_WG_RING]; p.r_l = (int)g[C2M_WG_RING_L];
    if (p.R && (Y_interior || p.reflect || p.Ho < 4 || p.Wo < 4 || p.r_l < p.Ho || p.r_l < p.Wo || p.out_off != 0)) return (int)hipErrorInvalidValue;
    p.act = act; p.slope = slope;
    if (p.M <= 0 || p.K <= 0 || p.nimg <= 0 || p.Ho <= 0 || p.Wo <= 0) return 0;
    if ((((uintptr_t)upack) & 15) != 0) return (int)hipErrorInvalidValue;
    p.nchunks = c2m_cdiv(p.K, W4_CK);
    p.mtiles = c2m_cdiv(p.M, 64);
    if (c2m_wino4_upack_floats(p.M, p.K) * 4 >= 0xffffffffL) return (int)hipErrorInvalidValue;
    const long regions = (long)p.nimg * c2m_wino4_regions(p.Ho, p.Wo);
    if (regions * p.mtiles > 0x7fffffffL) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_wino4_kernel, dim3((unsigned)(regions * p.mtiles)), dim3(512), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
