// conv_ring.hip -- the pad-ring part of the data gradient of a reflect-padded 3x3 stride-1 convolution, folded straight into dX
// (fp32; the same call sites as conv_wino.hip / conv_wino4.hip: layers/residual_block.py:13-31,42-71, same_block.py:50-68,
// spade_block.py:47-49 -- every `nn.Conv2d(..., padding=1, padding_mode='reflect')` of the generator, reference ATen chain:
// reflection_pad2d_backward(conv2d_backward_input(...))).
//
//   forward   Y = conv_valid(Xp, w),  Xp[py][px] = X[refl(py-1)][refl(px-1)]            (H+2) x (W+2) padded input
//   backward  dXp[m][py][px] = sum_{c,ky,kx} w[c][m][ky][kx] dY[c][py-ky][px-kx]        (full correlation)
//             dX[m][y][x]    = sum over the padded positions that mirror onto (y, x) of dXp
// The INTERIOR of dXp (py = y+1, px = x+1) is the zero-padded "same" data gradient over the exact H x W domain: that part runs on
// the Winograd kernels with full regions (the padded (H+2) x (W+2) domain they ran over before filled 28-67 % of a 16x32 F(4x4)
// region set and 44-89 % of the F(2x2) one).  What is left is the ring of dXp -- rows 0, H+1 and columns 0, W+1 -- which mirrors
// onto rows 1, H-2 and columns 1, W-2 of dX.  Each ring row / column is a 1-D three-tap correlation along a line of dY:
//   top     dX[m][1][i]   += sum_c sum_kx w[c][m][0][kx] dY[c][0][i+1-kx]        i in [0, W)
//   bottom  dX[m][H-2][i] += sum_c sum_kx w[c][m][2][kx] dY[c][H-1][i+1-kx]
//   left    dX[m][i][1]   += sum_c sum_ky w[c][m][ky][0] dY[c][i+1-ky][0]        i in [0, H)
//   right   dX[m][i][W-2] += sum_c sum_ky w[c][m][ky][2] dY[c][i+1-ky][W-1]
// i.e. four small GEMMs  M x 3C x (N * line length)  on v_mfma_f32_32x32x2_f32 (exact fp32), 3/9 * (2W + 2H) / (H W) of the
// layer's direct FLOPs (1.6 % at 64x128).  The four targets (1 | H-2, 1 | W-2) of a plane receive THREE ring terms each (row
// side, column side and the true corner of dXp): the GEMM part skips them and one thread per (image, channel, corner) sums its
// seven (tap, dY element) products over the channels -- so every element of dX has exactly one writer in this launch, the
// read-modify-write needs no atomics and the result is bit-repeatable.  No padded scratch tensor, no separate fold pass.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RING_OOB 0x80000000u

constexpr int RG_CK = 16;            // channels per chunk
constexpr int RG_PX = 128;           // pixels (line positions x images) per workgroup
constexpr int RG_MR = 64;            // rows (input channels of the layer = output rows of the GEMM) per workgroup

struct RingP {
    const float* A;                  // c2m_ring_pack: [side 4][chunk][tap 3][m-tile of 32][lane 64][8]
    const float* w;                  // native [C][M][3][3] (corner part)
    const float* dY;                 // [N][C][H][W]
    float* dX;                       // [N][M][H][W], already holds the interior term
    int N, C, M, H, W;
    int nchunks, mt32, mt64;
    int pt[4];                       // pixel tiles of 128 per side
    int gemm_blocks, corner_blocks;
    unsigned dy_bytes;
};

// Weights -> the A fragments the kernel loads: for side s, chunk q, tap j, 32-row tile t the lane (ml = lane & 31, kk = lane >> 5)
// reads 8 consecutive floats = channels 16 q + 2 e + kk (e = 0..7) of row 32 t + ml: a wave reads 2 KB contiguous per (s, q, j, t).
__global__ void ring_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int M, int nchunks, int mt32, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7); long r = i >> 3;
        const int lane = (int)(r & 63); r >>= 6;
        const int t = (int)(r % mt32); r /= mt32;
        const int j = (int)(r % 3); r /= 3;
        const int q = (int)(r % nchunks); const int s = (int)(r / nchunks);
        const int m = 32 * t + (lane & 31), c = RG_CK * q + 2 * e + (lane >> 5);
        const int ky = s == 0 ? 0 : (s == 1 ? 2 : j), kx = s == 2 ? 0 : (s == 3 ? 2 : j);
        out[i] = (m < M && c < C) ? w[((long)c * M + m) * 9 + ky * 3 + kx] : 0.f;
    }
}

C2M_API long c2m_ring_pack_floats(int C, int M) {
    return 4L * c2m_cdiv(C, RG_CK) * 3L * c2m_cdiv(M, 32) * 64L * 8L;
}

// w: native [Cout = C][Cin = M][3][3] of the forward layer
C2M_API int c2m_ring_pack(const float* w, float* apack, int C, int M, void* stream) {
    C2M_ENTER();
    if (C <= 0 || M <= 0) return 0;
    const long total = c2m_ring_pack_floats(C, M);
    hipLaunchKernelGGL(ring_pack_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, apack, C, M,
                       c2m_cdiv(C, RG_CK), c2m_cdiv(M, 32), total);
    return (int)hipGetLastError();
}

__global__ __launch_bounds__(256, 2) void reflect_ring_dgrad_kernel(const RingP p) {
    __shared__ float sB[3][RG_CK][RG_PX];            // the three shifted copies of a 16-channel slice of the line: 24 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W;
    if ((int)blockIdx.x < p.corner_blocks) {
        // ---- corner targets: item = (image, corner, row m), m fastest
        const long item = (long)blockIdx.x * 256 + tid;
        const long total = (long)p.N * 4 * p.M;
        if (item >= total) return;
        const int m = (int)(item % p.M); const int r = (int)(item / p.M);
        const int corner = r & 3, n = r >> 2;
        const int ty = (corner & 1) ? H - 2 : 1, ey = (corner & 1) ? H + 1 : 0, kyr = (corner & 1) ? 2 : 0;
        const int tx = (corner & 2) ? W - 2 : 1, ex = (corner & 2) ? W + 1 : 0, kxr = (corner & 2) ? 2 : 0;
        int tap[7], off[7];
#pragma unroll
        for (int k = 0; k < 3; ++k) {                // (ey, tx + 1): the ring ROW above / below the target, taps (kyr, k)
            tap[k] = kyr * 3 + k; off[k] = (ey - kyr) * W + (tx + 1 - k);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {                // (ty + 1, ex): the ring COLUMN beside it, taps (k, kxr)
            tap[3 + k] = k * 3 + kxr; off[3 + k] = (ty + 1 - k) * W + (ex - kxr);
        }
        tap[6] = kyr * 3 + kxr; off[6] = (ey - kyr) * W + (ex - kxr);        // (ey, ex): the corner of dXp
        const float* __restrict__ dy = p.dY + (long)n * p.C * HW;
        const float* __restrict__ wm = p.w + (long)m * 9;
        float acc = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const float* __restrict__ wc = wm + (long)c * p.M * 9;
            const float* __restrict__ dc = dy + (long)c * HW;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 7; ++k) s += wc[tap[k]] * dc[off[k]];
            acc += s;
        }
        float* __restrict__ d = p.dX + ((long)n * p.M + m) * HW + (long)ty * W + tx;
        *d += acc;
        return;
    }
    // ---- GEMM part: blocks ordered side-major, then pixel tile, then 64-row tile (neighbours share the B lines in L2)
    int b = (int)blockIdx.x - p.corner_blocks;
    int side = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int nb = p.pt[s] * p.mt64;
        if (side == s && b >= nb) { b -= nb; side = s + 1; }
    }
    const int mt = b % p.mt64, ptile = b / p.mt64;
    const bool rowside = side < 2;
    const int L = rowside ? W : H;                            // line length = targets per image on this side
    const long npix = (long)p.N * L;
    const int lbase = side == 1 ? (H - 1) * W : (side == 3 ? W - 1 : 0);      // first element of the dY line inside a plane
    const int es = rowside ? 1 : W;                           // element stride along the line
    // loader role: pixel px = tid & 127, channel half hf = tid >> 7 (wave-uniform): 8 channels x 3 taps per chunk
    const int lpx = tid & 127, hf = __builtin_amdgcn_readfirstlane(tid >> 7);
    unsigned voff[3];
    {
        const long pp = (long)ptile * RG_PX + lpx;
        const int n = (int)(pp / L), i = (int)(pp - (long)n * L);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int u = i + 1 - j;
            const bool ok = pp < npix && u >= 0 && u < L;
            voff[j] = ok ? (unsigned)(((long)n * p.C * HW + lbase + (long)u * es) * 4) : RING_OOB;
        }
    }
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dY), 0, p.dy_bytes, 0x00020000);
    // MFMA role: wave (wm = wave & 1: 32-row half, wn = wave >> 1: 64-pixel half), two 32x32 tiles side by side
    const int wm = wave & 1, wn = wave >> 1;
    const int t32 = mt * 2 + wm;
    const bool rows_live = t32 < p.mt32;
    const float* __restrict__ abase = p.A + ((((long)side * p.nchunks) * 3) * p.mt32 + t32) * 512 + lane * 8;
    const long a_tap = (long)p.mt32 * 512, a_chunk = 3 * a_tap;

    float braw[3][8];
    f32x4 araw[3][2];
    auto fetch = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = q * RG_CK + hf * 8 + e;             // wave-uniform
            const int soff = c * HW * 4;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                braw[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsy, c < p.C ? voff[j] : RING_OOB, soff, 0));
        }
        if (rows_live) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4* __restrict__ a = reinterpret_cast<const f32x4*>(abase + q * a_chunk + j * a_tap);
                araw[j][0] = a[0]; araw[j][1] = a[1];
            }
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    fetch(0);
    for (int q = 0; q < p.nchunks; ++q) {
        __syncthreads();                                       // the previous chunk's fragment reads are done
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) sB[j][hf * 8 + e][lpx] = braw[j][e];
        f32x4 a[3][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) { a[j][0] = araw[j][0]; a[j][1] = araw[j][1]; }
        __syncthreads();
        if (q + 1 < p.nchunks) fetch(q + 1);                   // in flight during this chunk's MFMAs
        if (rows_live) {
            const float* __restrict__ bb = &sB[0][lane >> 5][wn * 64 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float av = e < 4 ? a[j][0][e & 3] : a[j][1][e & 3];
                    const float b0 = bb[(j * RG_CK + 2 * e) * RG_PX], b1 = bb[(j * RG_CK + 2 * e) * RG_PX + 32];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
                }
        }
    }
    if (!rows_live) return;
    // ---- dX[target] += acc: lane owns pixel column (lane & 31) of each tile, rows 4 (lane >> 5) + (r & 3) + 8 (r >> 2)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long pp = (long)ptile * RG_PX + wn * 64 + t * 32 + (lane & 31);
        if (pp >= npix) continue;
        const int n = (int)(pp / L), i = (int)(pp - (long)n * L);
        if (i == 1 || i == L - 2) continue;                    // the corner targets of the plane: one thread of the corner part each
        const int ty = side == 0 ? 1 : (side == 1 ? H - 2 : i), tx = side == 2 ? 1 : (side == 3 ? W - 2 : i);
        float* __restrict__ d = p.dX + (long)n * p.M * HW + (long)ty * W + tx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = t32 * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < p.M) d[(long)m * HW] += acc[t][r];
        }
    }
}

// dX [N][M][H][W] += the ring terms of the reflect-pad-1 data gradient (see the head of this file); dX must already hold the
// zero-padded "same" data gradient.  apack = c2m_ring_pack(w); dY [N][C][H][W]; w native [C][M][3][3].  H, W >= 4.
C2M_API int c2m_reflect_ring_dgrad(const float* apack, const float* w, const float* dY, float* dX, int N, int C, int M, int H,
                                   int W, void* stream) {
    C2M_ENTER();
    if (N <= 0 || C <= 0 || M <= 0) return 0;
    if (H < 4 || W < 4) return (int)hipErrorInvalidValue;
    const long dy_bytes = 4L * N * C * H * W, dx_bytes = 4L * N * M * H * W;
    if (dy_bytes >= 0x80000000L || dx_bytes >= 0x80000000L) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)apack) & 15) != 0) return (int)hipErrorInvalidValue;
    RingP p;
    p.A = apack; p.w = w; p.dY = dY; p.dX = dX;
    p.N = N; p.C = C; p.M = M; p.H = H; p.W = W;
    p.nchunks = c2m_cdiv(C, RG_CK); p.mt32 = c2m_cdiv(M, 32); p.mt64 = c2m_cdiv(M, RG_MR);
    p.pt[0] = p.pt[1] = c2m_cdiv((long)N * W, RG_PX);
    p.pt[2] = p.pt[3] = c2m_cdiv((long)N * H, RG_PX);
    p.gemm_blocks = (p.pt[0] + p.pt[1] + p.pt[2] + p.pt[3]) * p.mt64;
    p.corner_blocks = c2m_cdiv((long)N * 4 * M, 256);
    p.dy_bytes = (unsigned)dy_bytes;
    hipLaunchKernelGGL(reflect_ring_dgrad_kernel, dim3((unsigned)(p.corner_blocks + p.gemm_blocks)), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
