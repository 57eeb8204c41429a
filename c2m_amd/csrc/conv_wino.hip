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
    long y2_sn, y2_sc, y2_sh;
    int lo_y, lo_x, ext_y, ext_x;
};

constexpr int WR = 8, WC = 16;               // output region rows / cols
constexpr int PH = WR + 2, PW = WC + 2;      // input patch
constexpr int CKW = 8;                       // channels per chunk
constexpr int PELEMS = CKW * PH * PW;        // 1440
constexpr int PLOADS = (PELEMS + 255) / 256; // 6

__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const WinoP p) {
    __shared__ float sP[3][PLOADS * 256];           // input patch [k][PH][PW], filled by LDS-DMA two chunks ahead
    __shared__ float sV[2 * 16 * CKW * 32];         // V[buf][xi][k][tile] (double buffered); reused by the epilogue
#ifdef WINO_OCC1
    __shared__ float sDummy[20000];
    if (threadIdx.x == 999) sDummy[p.M] = 1.f;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int regions_x = (p.Wo + WC - 1) / WC, regions_y = (p.Ho + WR - 1) / WR;
    int rb = blockIdx.x;
    const int rx = rb % regions_x; rb /= regions_x;
    const int ry = rb % regions_y; const int img = rb / regions_y;
    const int oy0 = ry * WR, ox0 = rx * WC;
    const int mt = blockIdx.y;

    // ---- patch addresses of this thread (fixed over the K loop)
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, p.x_bytes, 0x00020000);
    const unsigned img_byte = (unsigned)(img * (int)p.in_sn) * 4u;
    unsigned pvo[PLOADS];
#pragma unroll
    for (int i = 0; i < PLOADS; ++i) {
        const int e = tid + i * 256;
        const int k = e / (PH * PW), r = (e % (PH * PW)) / PW, c = e % PW;
        int iy = oy0 + p.iy0 + r, ix = ox0 + p.ix0 + c;
        if (p.reflect) {
            iy = iy < 0 ? -iy : iy; iy = iy >= p.Hi ? 2 * p.Hi - 2 - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= p.Wi ? 2 * p.Wi - 2 - ix : ix;
        }
        const bool ok = e < PELEMS && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        pvo[i] = ok ? img_byte + (unsigned)(k * p.in_sc + iy * p.in_sh + ix) * 4u : WINO_OOB;
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
    auto load_patch = [&](int chunk, int buf) {
        // channels beyond K only meet zero filter coefficients; beyond the tensor they read 0
        const int soff = chunk * CKW * p.in_sc * 4;
#pragma unroll
        for (int i = 0; i < PLOADS; ++i) {
            const unsigned dst = sp_lds + (unsigned)((buf * PLOADS * 256 + wave * 64 + i * 256) * 4);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(pvo[i]), "s"(rs), "s"(soff) : "memory");
        }
    };
    // ---- U fragments: 8 float4 per lane and chunk, [i = j*2 + mi][lane][kk 0..3]: every load instruction of a wave reads
    // 1 KB contiguous (per-lane-contiguous 128-byte records made the texture addresser the bottleneck: 64 lines per load)
    const float* __restrict__ ubase = p.U + (((long)mt * 4 + wave) * 8 * 64 + lane) * 4;
    const long ustride = (long)p.mtiles * 4 * 64 * 32;      // floats per chunk
    f32x4 ua[8];
    auto load_u = [&](int chunk) {
        const f32x4* __restrict__ q = reinterpret_cast<const f32x4*>(ubase + chunk * ustride);
#pragma unroll
        for (int i = 0; i < 8; ++i) ua[i] = q[i * 64];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][mi][r] = 0.f;

    // input-transform role of this thread: tile n (ty, tx), channel k of the chunk
    const int tn = tid & 31, tk = tid >> 5;
    const int ty = tn >> 3, tx = tn & 7;
    const int pbase = tk * (PH * PW) + (2 * ty) * PW + 2 * tx;
    // V = B^T d B of this thread's (channel, tile) from patch buffer `pb` into V buffer `vb`
    auto read_d = [&](int pb, float (&d)[4][4]) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b] = sP[pb][pbase + a * PW + b];
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
    if (p.nchunks > 1) load_patch(1, 1);
    load_u(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(ua[i]));   // retire these loads HERE: a load pending at loop entry
                                                       // costs a vmcnt(0) in front of the first MFMA of EVERY iteration
    if (p.nchunks > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
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
    auto iteration = [&](const int chunk, f32x4 (&ucur)[8], f32x4 (&unext)[8]) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
        // patch(chunk + 1) (this wave's part) has landed: nothing younger is in flight at this point
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();       // V(chunk) complete; patch(chunk+1) complete; everyone is done with V(chunk-1) = buffer cur^1
        float b[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                b[j][kk] = sV[cur * (16 * CKW * 32) + ((4 * wave + j) * CKW + 2 * kk + (lane >> 5)) * 32 + (lane & 31)];
        const float* q = ubase + (more ? chunk + 1 : chunk) * ustride;
        auto mfma_pair = [&](int g) {                  // g = kk * 4 + j
            const int kk = g >> 2, j = g & 3;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const f32x4 v = ucur[j * 2 + mi];
                const float av = kk == 0 ? v.x : (kk == 1 ? v.y : (kk == 2 ? v.z : v.w));
                acc[j][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][kk], acc[j][mi], 0, 0, 0);
            }
        };
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mfma_pair(g);
#pragma unroll
            for (int i = 2 * g; i < 2 * g + 2; ++i)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(unext[i]) : "v"(q + i * 256) : "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // patch(chunk + 2) -- always issued, so the iteration is one basic block and the requests sit between MFMAs; past
        // the last chunk every lane's offset is out of range (the loads return 0 into a patch buffer nobody reads)
        const bool pre = chunk + 2 < p.nchunks;
        const int soff = pre ? (chunk + 2) * CKW * p.in_sc * 4 : 0;
        const int dbuf = pnext == 2 ? 0 : pnext + 1;
#pragma unroll
        for (int g = 4; g < 4 + PLOADS / 2; ++g) {
            mfma_pair(g);
#pragma unroll
            for (int i = 2 * (g - 4); i < 2 * (g - 4) + 2; ++i) {
                const unsigned dst = sp_lds + (unsigned)((dbuf * PLOADS * 256 + wave * 64 + i * 256) * 4);
                const unsigned vo = pre ? pvo[i] : WINO_OOB;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(vo), "s"(rs), "s"(soff) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float d[4][4];
        read_d(pnext, d);                               // (last chunk: stale but in-bounds data, result unused)
#pragma unroll
        for (int g = 4 + PLOADS / 2; g < 16; ++g) mfma_pair(g);
        transform_store(d, cur ^ 1);                    // unconditional: one basic block, so it can interleave
#pragma unroll
        for (int g = 0; g < 9; ++g) {                  // 2 MFMA : 4 VALU : 2 LDS writes (18 MFMAs)
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
        // the 8 U loads are older than the 6 DMA loads issued after them
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(unext[i]));   // uses of U(chunk + 1) stay behind the wait
        pnext = pnext == 2 ? 0 : pnext + 1;
    };
    f32x4 ub[8];
    int chunk = 0;
    for (; chunk + 1 < p.nchunks; chunk += 2) {
        iteration(chunk, ua, ub);
        iteration(chunk + 1, ub, ua);
    }
    if (chunk < p.nchunks) iteration(chunk, ua, ub);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the (empty) DMAs of the last iterations must not outlive the block's LDS
    static_assert(PLOADS == 6, "the DMA issue slots assume six patch loads per thread");
    __syncthreads();

    // ---- inverse transform.  Column part in registers: R_i[q] = sum_j M[i][j] A[j][q], A = [[1,0],[1,1],[1,-1],[0,-1]]
    // Row part through LDS: Y[0][q] = R_0 + R_1 + R_2, Y[1][q] = R_1 - R_2 - R_3 (i = wave)
    // Every thread owns one output pixel (x, y) of the region for the whole epilogue and walks over the output channels, so
    // the pixel's tile, row parity, bounds and target address are computed once (the per-element index arithmetic used to
    // cost as much as the stores: 27 % of the kernel on the 64-channel layers).
    constexpr int QS = 4 * 16 * 32 + 16;             // q stride: +16 floats keeps the two column parities on different banks
    float* __restrict__ sR = sV;                     // [q][w][16 cout][32 tiles] per pass (16 of the 64 couts)
    const int m0 = mt * 64;
    const int ex = tid & 15, ey = (tid >> 4) & 7, c0 = tid >> 7;
    const int oy = oy0 + ey, ox = ox0 + ex;
    const bool inb = oy < p.Ho && ox < p.Wo;
    const bool top = (ey & 1) == 0;
    float* __restrict__ ybase;
    long cstride;
    {
        const int yi = oy - p.lo_y, xi = ox - p.lo_x;
        if (p.Y2 && (unsigned)yi < (unsigned)p.ext_y && (unsigned)xi < (unsigned)p.ext_x) {
            ybase = p.Y2 + (long)img * p.y2_sn + (long)yi * p.y2_sh + xi;
            cstride = p.y2_sc;
        } else {
            ybase = p.Y + p.out_off + (long)img * p.out_sn + (long)oy * p.out_sh + ox;
            cstride = p.out_sc;
        }
    }
    // R rows this pixel combines: top row r0 + r1 + r2, bottom row r1 - r2 - r3; `rq` points at r1, `ro` at r0 or r3
    const float* __restrict__ rq = sR + (ex & 1) * QS + (1 * 16 + c0) * 32 + (ey >> 1) * 8 + (ex >> 1);
    const int ro = top ? -16 * 32 : 2 * 16 * 32;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {               // accumulator registers r in [8hf, 8hf + 8) = tile rows 16hf .. 16hf+15
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = hf * 8 + rr;
                const int rowl = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5);        // 0..15 within the half
                sR[0 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[0][mi][r] + acc[1][mi][r]) + acc[2][mi][r];
                sR[1 * QS + (wave * 16 + rowl) * 32 + (lane & 31)] = (acc[1][mi][r] - acc[2][mi][r]) - acc[3][mi][r];
            }
            __syncthreads();
            // 16 couts x 8 rows x 16 columns; consecutive threads -> consecutive x (64-byte row segments)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int co = c0 + 2 * it;
                const float r1 = rq[co * 32 - c0 * 32], r2 = rq[co * 32 - c0 * 32 + 16 * 32], rx = rq[co * 32 - c0 * 32 + ro];
                float v = top ? (rx + r1) + r2 : (r1 - r2) - rx;
                const int cout = m0 + mi * 32 + hf * 16 + co;
                if (inb && cout < p.M) {
                    if (p.bias) v += p.bias[cout];
                    v = c2m_act(v, p.act, p.slope);
                    ybase[(long)cout * cstride] = v;
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

// geom[] (int64): 0 M, 1 K, 2 images, 3 Hi, 4 Wi, 5 Ho, 6 Wo, 7 iy0, 8 ix0, 9 reflect, 10 in_sn, 11 in_sc, 12 in_sh,
//                 13 out_sn, 14 out_sc, 15 out_sh, 16 out_off, 17 x_bytes; with Y_interior: 18 y2_sn, 19 y2_sc, 20 y2_sh,
//                 21 lo_y, 22 lo_x, 23 ext_y, 24 ext_x
C2M_API int c2m_conv_wino(const float* upack, const float* X, float* Y, float* Y_interior, const float* bias,
                          const int64_t* g, int act, float slope, void* stream) {
    C2M_ENTER();
    WinoP p;
    p.U = upack; p.X = X; p.Y = Y; p.bias = bias;
    p.Y2 = Y_interior; p.y2_sn = p.y2_sc = p.y2_sh = 0; p.lo_y = p.lo_x = p.ext_y = p.ext_x = 0;
    if (Y_interior) {
        p.y2_sn = g[18]; p.y2_sc = g[19]; p.y2_sh = g[20];
        p.lo_y = (int)g[21]; p.lo_x = (int)g[22]; p.ext_y = (int)g[23]; p.ext_x = (int)g[24];
    }
    p.M = (int)g[0]; p.K = (int)g[1]; p.nimg = (int)g[2];
    p.Hi = (int)g[3]; p.Wi = (int)g[4]; p.Ho = (int)g[5]; p.Wo = (int)g[6];
    p.iy0 = (int)g[7]; p.ix0 = (int)g[8]; p.reflect = (int)g[9];
    p.in_sn = g[10]; p.in_sc = (int)g[11]; p.in_sh = (int)g[12];
    p.out_sn = g[13]; p.out_sc = g[14]; p.out_sh = g[15]; p.out_off = g[16];
    if (g[17] <= 0 || g[17] >= 0x80000000LL) return (int)hipErrorInvalidValue;
    p.x_bytes = (unsigned)g[17];
    p.act = act; p.slope = slope;
    if (p.M <= 0 || p.K <= 0 || p.nimg <= 0 || p.Ho <= 0 || p.Wo <= 0) return 0;
    if ((((uintptr_t)upack) & 15) != 0) return (int)hipErrorInvalidValue;
    p.nchunks = c2m_cdiv(p.K, CKW);
    p.mtiles = c2m_cdiv(p.M, 64);
    const long regions = (long)p.nimg * c2m_cdiv(p.Ho, WR) * c2m_cdiv(p.Wo, WC);
    if (regions > 0x7fffffffL) return (int)hipErrorInvalidValue;
    dim3 grid((unsigned)regions, (unsigned)p.mtiles);
    hipLaunchKernelGGL(conv_wino_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

// ================================================================================================ weight gradient
// Winograd form of the 3x3 stride-1 weight gradient (same layers as above):
//   dg[co][ci] = G^T [ sum_tiles (A dY A^T)[co][tile] (.) (B^T d B)[ci][tile] ] G
// i.e. 16 independent GEMMs dU_xi[co][ci] = dM_xi[co][:] . V_xi[ci][:] contracted over the 2x2-output tiles (N*H/2*W/2 of
// them) -- again 16 multiplies where the direct form needs 36.  One workgroup = 64 output channels x 32 input channels
// over a range of 2x16-output regions (8 tiles each = one chunk, K = 8): both patches arrive by LDS-DMA one chunk ahead
// of their transform, the two transforms of chunk c+1 are interleaved with the MFMAs of chunk c (double-buffered
// operand images in LDS: dM [xi][tile][co], V [xi][tile][ci] -- lanes along the channel, conflict-free on both sides).
// Pixel ranges are split over workgroups into slabs [split][xi][Cout][Cin] (+ a bias-gradient slab from the plain sum of
// dY); wino_wgrad_reduce_kernel sums the splits in a fixed order, applies G^T (.) G and writes dW in its native layout.
struct WinoWgP {
    const float* dY; const float* X; float* slab; float* dbslab;
    int M, K, nimg, H, W;            // Cout, Cin, images, spatial size (dY and X have the same: stride 1, pad 1)
    int reflect;
    long dy_sn, x_sn;                // image strides (elements); channel stride = H*W for both
    unsigned dy_bytes, x_bytes;
    int regions, per_split;          // 2x16-output regions in total / per workgroup
};

constexpr int GR = 2, GC = 8, GT = 4;                // region rows, cols, tiles (one chunk: K = 4 tiles)
constexpr int GY_STRIDE = GR * GC + 1;               // 17: dY patch [co][2][8] + 1 pad (odd stride: conflict-free)
constexpr int GX_STRIDE = (GR + 2) * (GC + 2) + 1;   // 41: X patch [ci][4][10] + 1 pad
constexpr int GY_LOADS = (64 * GY_STRIDE + 255) / 256;   // 5
constexpr int GX_LOADS = (64 * GX_STRIDE + 255) / 256;   // 11

// Workgroup = 64 output channels x 64 input channels x 16 frequencies = 256 accumulator registers per lane (AGPRs; one
// workgroup per CU).  Issue-slot budget: the 32 MFMAs of a chunk occupy 32 x 16 issue quads; everything else of the
// chunk (patch DMA, both transforms, fragment reads, address arithmetic) must stay well below that and be spread
// between the MFMAs (one MFMA : ~6 other instructions), because instructions behind a stalled MFMA cannot overtake it.
__global__ __launch_bounds__(256) void conv_wino_wgrad_kernel(const WinoWgP p) {
    __shared__ float pY[2][GY_LOADS * 256];
    __shared__ float pX[2][GX_LOADS * 256];
    __shared__ float sDM[2][16 * GT * 64];             // [xi][tile][co]
    __shared__ float sVV[2][16 * GT * 64];             // [xi][tile][ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x, mt = blockIdx.y, nt = blockIdx.z;
    const int m0 = mt * 64, c0 = nt * 64;
    const int HW = p.H * p.W;
    const int rbeg = split * p.per_split;
    int rend = rbeg + p.per_split; rend = rend < p.regions ? rend : p.regions;
    const int nchunks = rend - rbeg;
    const int regions_x = p.W / GC, regions_y = p.H / GR;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long ya = (unsigned long)p.dY, xa = (unsigned long)p.X;
    const u32x4 rsy = {(unsigned)ya, (unsigned)(ya >> 32) & 0xffffu, p.dy_bytes, 0x00020000u};
    const u32x4 rsx = {(unsigned)xa, (unsigned)(xa >> 32) & 0xffffu, p.x_bytes, 0x00020000u};
    const unsigned py_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)&pY[0][0];
    const unsigned px_lds = (unsigned)(unsigned long)(__attribute__((address_space(3))) float*)&pX[0][0];

    // patch slots of this thread; byte offsets relative to the region origin (the origin is the scalar offset per chunk)
    unsigned yvo[GY_LOADS];
#pragma unroll
    for (int i = 0; i < GY_LOADS; ++i) {
        const int s = tid + i * 256;
        const int co = s / GY_STRIDE, rem = s % GY_STRIDE;
        const bool ok = co < 64 && rem < GR * GC && m0 + co < p.M;
        yvo[i] = ok ? (unsigned)((m0 + co) * HW + (rem / GC) * p.W + rem % GC) * 4u : WINO_OOB;
    }
    // X patch: (ci, r, c), r in 0..3 (input rows oy0-1 ..), c in 0..9.  Interior regions use the precomputed offsets;
    // regions touching the image border recompute theirs (zero / reflect padding)
    unsigned xvo[GX_LOADS];
    int xci[GX_LOADS], xrc[GX_LOADS];
#pragma unroll
    for (int i = 0; i < GX_LOADS; ++i) {
        const int s = tid + i * 256;
        const int ci = s / GX_STRIDE, rem = s % GX_STRIDE;
        const bool ok = ci < 64 && rem < (GR + 2) * (GC + 2) && c0 + ci < p.K;
        const int r = rem / (GC + 2), c = rem % (GC + 2);
        xci[i] = ok ? (c0 + ci) * HW : -1;
        xrc[i] = r * 16 + c;
        xvo[i] = ok ? (unsigned)((c0 + ci) * HW + (r - 1) * p.W + (c - 1)) * 4u : WINO_OOB;     // may wrap: added to soff
    }
    auto issue_dma = [&](int region, int buf) {
        const int rx = region % regions_x; int t = region / regions_x;
        const int ry = t % regions_y; const int img = t / regions_y;
        const int oy0 = ry * GR, ox0 = rx * GC;
        const int ysoff = (int)(((long)img * p.dy_sn + (long)oy0 * p.W + ox0) * 4);
#pragma unroll
        for (int i = 0; i < GY_LOADS; ++i) {
            const unsigned dst = py_lds + (unsigned)((buf * GY_LOADS * 256 + wave * 64 + i * 256) * 4);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(yvo[i]), "s"(rsy), "s"(ysoff) : "memory");
        }
        const bool interior = ry > 0 && ry + 1 < regions_y && rx > 0 && rx + 1 < regions_x;    // wave-uniform
        const unsigned ximg = (unsigned)((long)img * p.x_sn * 4);
        const unsigned xorg = ximg + (unsigned)(oy0 * p.W + ox0) * 4u;
#pragma unroll
        for (int i = 0; i < GX_LOADS; ++i) {
            unsigned vo;
            if (interior) {
                vo = xvo[i] == WINO_OOB ? WINO_OOB : xvo[i] + xorg;
            } else {
                int iy = oy0 - 1 + (xrc[i] >> 4), ix = ox0 - 1 + (xrc[i] & 15);
                bool ok = xci[i] >= 0;
                if (p.reflect) {
                    iy = iy < 0 ? -iy : iy; iy = iy >= p.H ? 2 * p.H - 2 - iy : iy;
                    ix = ix < 0 ? -ix : ix; ix = ix >= p.W ? 2 * p.W - 2 - ix : ix;
                } else {
                    ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                }
                vo = ok ? ximg + (unsigned)(xci[i] + iy * p.W + ix) * 4u : WINO_OOB;
            }
            const unsigned dst = px_lds + (unsigned)((buf * GX_LOADS * 256 + wave * 64 + i * 256) * 4);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(vo), "s"(rsx), "s"(0) : "memory");
        }
    };

    // transform role of a thread: channel ch = tid & 63 (as output channel AND as input channel), tile tq = tid >> 6
    const int ch = tid & 63, tq = tid >> 6;
    float dbacc = 0.f;
    float yv[4], xv[4][4];
    auto read_patches = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[e] = pY[buf][ch * GY_STRIDE + (e >> 1) * GC + 2 * tq + (e & 1)];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) xv[a][b] = pX[buf][ch * GX_STRIDE + a * (GC + 2) + 2 * tq + b];
    };
    auto transform_store = [&](int ob) {
        // dM = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]
        const float y00 = yv[0], y01 = yv[1], y10 = yv[2], y11 = yv[3];
        dbacc += (y00 + y01) + (y10 + y11);
        const float T[4][2] = {{y00, y01}, {y00 + y10, y01 + y11}, {y00 - y10, y01 - y11}, {-y10, -y11}};
        float* __restrict__ d = &sDM[ob][tq * 64 + ch];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            d[(4 * i + 0) * GT * 64] = T[i][0];
            d[(4 * i + 1) * GT * 64] = T[i][0] + T[i][1];
            d[(4 * i + 2) * GT * 64] = T[i][0] - T[i][1];
            d[(4 * i + 3) * GT * 64] = -T[i][1];
        }
        // V = B^T d B
        float x[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[0][b] = xv[0][b] - xv[2][b];
            x[1][b] = xv[1][b] + xv[2][b];
            x[2][b] = xv[2][b] - xv[1][b];
            x[3][b] = xv[1][b] - xv[3][b];
        }
        float* __restrict__ v = &sVV[ob][tq * 64 + ch];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[(4 * i + 0) * GT * 64] = x[i][0] - x[i][2];
            v[(4 * i + 1) * GT * 64] = x[i][1] + x[i][2];
            v[(4 * i + 2) * GT * 64] = x[i][2] - x[i][1];
            v[(4 * i + 3) * GT * 64] = x[i][1] - x[i][3];
        }
    };

    f32x16 acc[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mi][ni][r] = 0.f;

    if (nchunks > 0) {
        issue_dma(rbeg, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_patches(0);
        if (nchunks > 1) issue_dma(rbeg + 1, 1);
        transform_store(0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int cur = chunk & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // patches(chunk + 1) of this wave have landed
            __syncthreads();   // operands(chunk) + patches(chunk+1) complete; everyone is done with operand buffer cur ^ 1
            float a[4][2][2], b[4][2][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int row = ((4 * wave + j) * GT + 2 * kk + (lane >> 5)) * 64 + (lane & 31);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        a[j][q][kk] = sDM[cur][row + q * 32];
                        b[j][q][kk] = sVV[cur][row + q * 32];
                    }
                }
            const bool more = chunk + 1 < nchunks;
            read_patches(cur ^ 1);                       // chunk + 1 (last chunk: stale data, result unused)
            __builtin_amdgcn_sched_barrier(0);
            if (chunk + 2 < nchunks) issue_dma(rbeg + chunk + 2, cur);   // patch buffer `cur` was consumed one iteration ago
            const float dbkeep = dbacc;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[j][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][mi][kk], b[j][ni][kk],
                                                                                  acc[j][mi][ni], 0, 0, 0);
            transform_store(cur ^ 1);
            if (!more) dbacc = dbkeep;                   // the stale transform of the last iteration must not count
#pragma unroll
            for (int g = 0; g < 32; ++g) {               // one MFMA : ~3 other instructions
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- slab: dU_xi[co][ci] partials of this workgroup's region range
    float* __restrict__ out = p.slab + (long)split * 16 * p.M * p.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xi = 4 * wave + j;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int ci = c0 + ni * 32 + (lane & 31);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (co < p.M && ci < p.K) out[((long)xi * p.M + co) * p.K + ci] = acc[j][mi][ni][r];
                }
        }
    }
    // ---- bias gradient partial: sum of dY over this workgroup's regions (input-channel tile 0 only)
    if (nt == 0) {
        float* __restrict__ red = &sDM[0][0];
        red[tq * 64 + ch] = dbacc;
        __syncthreads();
        if (tid < 64 && m0 + tid < p.M)
            p.dbslab[(long)split * p.M + m0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
}

// dW[co][ci][3][3] = G^T (sum_splits dU) G;  db[co] = sum_splits dbslab
__global__ void wino_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ dbslab,
                                         float* __restrict__ dW, float* __restrict__ db, int M, int K, int S) {
    const long total = (long)M * K;
    const long per = 16 * total;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float u[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float acc = 0.f;
            for (int s = 0; s < S; ++s) acc += slab[(long)s * per + (long)xi * total + i];
            u[xi >> 2][xi & 3] = acc;
        }
        // G^T u: rows a = 0..2 from i = 0..3 with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
            t[1][j] = 0.5f * (u[1][j] - u[2][j]);
            t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
        }
        float* __restrict__ o = dW + i * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            o[a * 3 + 0] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
            o[a * 3 + 1] = 0.5f * (t[a][1] - t[a][2]);
            o[a * 3 + 2] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
        }
    }
    if (db) {
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
            float acc = 0.f;
            for (int s = 0; s < S; ++s) acc += dbslab[(long)s * M + m];
            db[m] = acc;
        }
    }
}

// Number of region splits (= slabs) the launch will use: slab = S*16*M*K floats, dbslab = S*M floats.
C2M_API int c2m_wino_wgrad_splits(int M, int K, int nimg, int H, int W) {
    const long regions = (long)nimg * (H / GR) * (W / GC);
    const long tiles = (long)c2m_cdiv(M, 64) * c2m_cdiv(K, 64);
    long S = 256 / tiles;                       // one workgroup per CU (256 accumulator registers): one resident round
    if (S < 1) S = 1;
    const long maxS = (regions + 31) / 32;      // >= 32 regions (128 tiles) per split
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    const long per = (regions + S - 1) / S;
    return (int)((regions + per - 1) / per);
}

// 3x3, stride 1, pad 1 (zeros or reflect), H % 2 == 0, W % 8 == 0.  dY [N][M][H][W], X [N][K][H][W] contiguous.
C2M_API int c2m_conv_wino_wgrad(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db,
                                int M, int K, int nimg, int H, int W, int reflect, void* stream) {
    C2M_ENTER();
    if (M <= 0 || K <= 0 || nimg <= 0) return 0;
    if (H % GR || W % GC) return (int)hipErrorInvalidValue;
    const long ybytes = 4L * nimg * M * H * W, xbytes = 4L * nimg * K * H * W;
    if (ybytes >= 0x80000000LL || xbytes >= 0x80000000LL) return (int)hipErrorInvalidValue;
    WinoWgP p;
    p.dY = dY; p.X = X; p.slab = slab; p.dbslab = dbslab;
    p.M = M; p.K = K; p.nimg = nimg; p.H = H; p.W = W; p.reflect = reflect;
    p.dy_sn = (long)M * H * W; p.x_sn = (long)K * H * W;
    p.dy_bytes = (unsigned)ybytes; p.x_bytes = (unsigned)xbytes;
    p.regions = nimg * (H / GR) * (W / GC);
    const int S = c2m_wino_wgrad_splits(M, K, nimg, H, W);
    p.per_split = c2m_cdiv(p.regions, S);
    dim3 grid(S, c2m_cdiv(M, 64), c2m_cdiv(K, 64));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wino_wgrad_kernel, grid, dim3(256), 0, s, p);
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(c2m_grid((long)M * K, 256)), dim3(256), 0, s, slab, dbslab, dW, db,
                       M, K, S);
    return (int)hipGetLastError();
}
