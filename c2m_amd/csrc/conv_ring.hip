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
// side, column side and the true corner of dXp): the GEMM part skips them and the corner part -- one workgroup per (image,
// corner), one thread per row -- sums their seven (tap, dY element) products over the channels.  So every element of dX has
// exactly one writer in this launch: the read-modify-write needs no atomics and the result is bit-repeatable.  No padded scratch
// tensor, no separate fold pass.
//
// GEMM part: one workgroup = one side x 64 line positions (pixels: images x positions, tiles may span images) x up to 256 rows.
// The column sides read ONE useful float per 128-byte line of dY (elements W floats apart), and a fully divergent load costs the
// texture path ~4 cycles per lane: the first version (three shifted copies of the line loaded per tap, re-gathered by every
// 64-row tile) spent ~12 us PER 16-channel chunk in those gathers.  Now the 66 elements a tile needs (64 + one neighbour either
// side) are loaded once per chunk into LDS -- every tap is the same slice read at another offset, the line ends are per-lane masks
// -- and serve all the rows of the workgroup: 6-12x fewer line requests per CU.
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RING_OOB 0x80000000u

constexpr int RG_CK = 16;            // channels per chunk
constexpr int RG_PX = 64;            // pixels (line positions x images) per workgroup of a row side; column sides: 32

struct RingP {
    const float* A;                  // c2m_ring_pack: [side 4][chunk][tap 3][m-tile of 32][lane 64][8]
    const float* dY;                 // [N][C][H][W]
    float* dX;                       // [N][M][H][W], already holds the interior term (in-place mode)
    float* R;                        // buffer mode (c2m_reflect_ring_buffer): [N][M][4 sides][r_l] ring terms for the Winograd epilogue
    int r_l;
    int N, C, M, H, W;
    int nchunks, mt32, mgroups;      // 16-channel chunks, 32-row tiles, row groups of 128 * MTW rows
    int pt[4];                       // pixel tiles of 64 per side
    int gemm_blocks, corner_blocks;
    int diag;                        // timing diagnostics (C2M_RING_DIAG): 1 no epilogue, 2 no MFMAs, 4 no dY gathers, 8 no A loads
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

constexpr int RG_SP = 72;            // LDS pitch of a channel's 66-element slice

// MTW = 32-row tiles per wave: the workgroup's four waves cover 128 * MTW rows of ALL 64 pixels
template <int MTW>
__global__ __launch_bounds__(256, MTW == 1 ? 2 : 1) void reflect_ring_dgrad_kernel(const RingP p) {
    __shared__ float sS[2][RG_CK][RG_SP];            // 2 x 16 channels x (pixel p0 - 1 .. p0 + 64) of the side's line(s)
    __shared__ float sD[7][RG_CK];                   // corner part: the seven dY elements of 16 channels
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, HW = H * W;
    if ((int)blockIdx.x >= p.gemm_blocks) {
        // ---- corner targets: block = (image, corner), thread = row m (looped over M in steps of 256).  Per 16-channel chunk the
        // seven dY elements of every channel go through LDS (112 loads per block, not per row) and the weights come from the
        // packed A fragments: 8 channels per 32-byte load, rows contiguous.
        const int r = (int)blockIdx.x - p.gemm_blocks;
        const int corner = r & 3, n = r >> 2;
        const int ty = (corner & 1) ? H - 2 : 1, ey = (corner & 1) ? H + 1 : 0, kyr = (corner & 1) ? 2 : 0;
        const int tx = (corner & 2) ? W - 2 : 1, ex = (corner & 2) ? W + 1 : 0, kxr = (corner & 2) ? 2 : 0;
        const int side_row = (corner & 1) ? 1 : 0, side_col = (corner & 2) ? 3 : 2;
        // term k = tid >> 4 of the loader threads (tid < 112): its dY element inside a channel plane
        int my_off = 0;
        {
            const int k = tid >> 4;
            if (k < 3) my_off = (ey - kyr) * W + (tx + 1 - k);                 // ring ROW above / below the target, taps (kyr, k)
            else if (k < 6) my_off = (ty + 1 - (k - 3)) * W + (ex - kxr);      // ring COLUMN beside it, taps (k - 3, kxr)
            else my_off = (ey - kyr) * W + (ex - kxr);                         // the corner of dXp, tap (kyr, kxr)
        }
        const float* __restrict__ dy = p.dY + (long)n * p.C * HW;
        const long a_tap = (long)p.mt32 * 512, a_chunk = 3 * a_tap, a_side = (long)p.nchunks * a_chunk;
        for (int mbase = 0; mbase < p.M; mbase += 256) {
            const int m = mbase + tid;
            const int mc = m < p.M ? m : p.M - 1;
            const float* __restrict__ am = p.A + (long)(mc >> 5) * 512 + (mc & 31) * 8;
            float acc = 0.f;
            for (int q = 0; q < p.nchunks; ++q) {
                __syncthreads();
                if (tid < 112) {
                    const int c = q * RG_CK + (tid & 15);
                    sD[tid >> 4][tid & 15] = c < p.C ? dy[(long)c * HW + my_off] : 0.f;
                }
                __syncthreads();
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const int sd = k < 3 ? side_row : (k < 6 ? side_col : side_row);
                    const int j = k < 3 ? k : (k < 6 ? k - 3 : kxr);
                    const f32x4* __restrict__ a = reinterpret_cast<const f32x4*>(am + sd * a_side + q * a_chunk + j * a_tap);
                    const f32x4 e0 = a[0], e1 = a[1], o0 = a[64], o1 = a[65];      // even channels (kk = 0) / odd channels (lane + 32)
                    const float* __restrict__ d = sD[k];
                    s += e0[0] * d[0] + o0[0] * d[1] + e0[1] * d[2] + o0[1] * d[3] + e0[2] * d[4] + o0[2] * d[5] + e0[3] * d[6] + o0[3] * d[7]
                       + e1[0] * d[8] + o1[0] * d[9] + e1[1] * d[10] + o1[1] * d[11] + e1[2] * d[12] + o1[2] * d[13] + e1[3] * d[14] + o1[3] * d[15];
                }
                acc += s;
            }
            if (m < p.M) {
                float* __restrict__ d = p.dX + ((long)n * p.M + m) * HW + (long)ty * W + tx;
                *d += acc;
            }
        }
        return;
    }
    // ---- GEMM part (the first blocks of the grid: they are the long ones): side-major, then pixel tile, then row group
    int b = (int)blockIdx.x;
    int side = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int nb = p.pt[s] * p.mgroups;
        if (side == s && b >= nb) { b -= nb; side = s + 1; }
    }
    const int mg = b % p.mgroups, ptile = b / p.mgroups;
    const bool rowside = side < 2;
    const int L = rowside ? W : H;                            // line length = targets per image on this side
    const long npix = (long)p.N * L;
    const int lbase = side == 1 ? (H - 1) * W : (side == 3 ? W - 1 : 0);      // first element of the dY line inside a plane
    const int es = rowside ? 1 : W;                           // element stride along the line
    const int nhalf = rowside ? 2 : 1;                        // 32-pixel halves per tile: rows 64 pixels, columns 32 (see RG_PX)
    const int tpx = 32 * nhalf;
    const long p0 = (long)ptile * tpx;
    // loader roles: slice index x = 1 + (tid & 63) (pixel p0 + (tid & 63)) for the four channels 4 (tid >> 6) + e (wave-uniform);
    // the two neighbour elements x = 0 / 65 of channel tid & 15 by the first 32 threads
    auto slice_off = [&](long g) -> unsigned {
        if (g < 0 || g >= npix) return RING_OOB;
        const int n = (int)(g / L), u = (int)(g - (long)n * L);
        return (unsigned)(((long)n * p.C * HW + lbase + (long)u * es) * 4);
    };
    const unsigned voff_main = (tid & 63) < tpx ? slice_off(p0 + (tid & 63)) : RING_OOB;
    const unsigned voff_halo = tid < 32 ? slice_off(tid < 16 ? p0 - 1 : p0 + tpx) : RING_OOB;
    const int cg = wave;                                      // = tid >> 6
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dY), 0, p.dy_bytes, 0x00020000);
    // MFMA roles: wave w owns the 32-row tiles t32 = (mg * 4 + w) * MTW + k, k < MTW, over both 32-pixel halves of the tile
    const int t32_0 = (mg * 4 + wave) * MTW;
    const long a_tap = (long)p.mt32 * 512, a_chunk = 3 * a_tap;
    const float* __restrict__ abase = p.A + ((long)side * p.nchunks) * a_chunk + lane * 8;
    // line-end masks of this lane's two pixels: tap 0 reads the NEXT line position, tap 2 the previous one
    // Buffer mode, row sides: the TRUE corners of the padded gradient (row 0 | H+1, column 0 | W+1: one tap each) mirror onto the
    // line positions 1 and L-2 -- tap 0 at i = 1 also takes the element at i - 1, tap 2 at i = L-2 the one at i + 1 -- so the row
    // terms of those positions are complete and the Winograd epilogue adds row and column terms independently (no corner part).
    bool m_next[2], m_prev[2], fix_lo[2], fix_hi[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long g = p0 + t * 32 + (lane & 31);
        const int i = (int)(g % L);
        m_next[t] = i != L - 1; m_prev[t] = i != 0;
        fix_lo[t] = p.R != nullptr && rowside && i == 1;
        fix_hi[t] = p.R != nullptr && rowside && i == L - 2;
    }

    // Two 16-channel chunks per loop iteration (one barrier pair and one prefetch round trip per 32 channels: with one chunk per
    // iteration the 24-48 MFMAs of a chunk were shorter than the latency of the next chunk's divergent gathers)
    constexpr int NSUB = 2;
    float sraw[NSUB][4], hraw[NSUB] = {0.f, 0.f};
    f32x4 araw[NSUB][MTW][3][2];
    auto fetch = [&](int q2) __attribute__((always_inline)) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int q = q2 * NSUB + sub;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (p.diag & 4) break;
                const int c = q * RG_CK + cg * 4 + e;         // wave-uniform; chunks past the end: c >= C -> zeros
                sraw[sub][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsy, c < p.C ? voff_main : RING_OOB, c * HW * 4, 0));
            }
            if (wave == 0 && !(p.diag & 4)) {
                const int c = q * RG_CK + (tid & 15);
                hraw[sub] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rsy, (c < p.C && voff_halo != RING_OOB) ? voff_halo + (unsigned)(c * HW * 4) : RING_OOB, 0, 0));
            }
#pragma unroll
            for (int k = 0; k < MTW; ++k) {
                const bool live = t32_0 + k < p.mt32 && q < p.nchunks && !(p.diag & 8);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (live) {
                        const f32x4* __restrict__ a = reinterpret_cast<const f32x4*>(abase + q * a_chunk + j * a_tap + (long)(t32_0 + k) * 512);
                        araw[sub][k][j][0] = a[0]; araw[sub][k][j][1] = a[1];
                    } else {
                        araw[sub][k][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; araw[sub][k][j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    };
    f32x16 acc[MTW][2];
#pragma unroll
    for (int k = 0; k < MTW; ++k)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][t][r] = 0.f;

    const int niter = (p.nchunks + NSUB - 1) / NSUB;
    fetch(0);
    for (int q2 = 0; q2 < niter; ++q2) {
        __syncthreads();                                       // the previous iteration's fragment reads are done
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((tid & 63) < tpx) sS[sub][cg * 4 + e][1 + (tid & 63)] = sraw[sub][e];   // (column sides: 32 live lanes; index tpx + 1 is the neighbour element's)
            if (tid < 32) sS[sub][tid & 15][tid < 16 ? 0 : tpx + 1] = hraw[sub];
        }
        f32x4 a[NSUB][MTW][3][2];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int k = 0; k < MTW; ++k)
#pragma unroll
                for (int j = 0; j < 3; ++j) { a[sub][k][j][0] = araw[sub][k][j][0]; a[sub][k][j][1] = araw[sub][k][j][1]; }
        __syncthreads();
        if (q2 + 1 < niter) fetch(q2 + 1);                     // in flight during this iteration's MFMAs
        if (p.diag & 2) continue;
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const float* __restrict__ bb = &sS[sub][lane >> 5][(lane & 31) + 2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* __restrict__ be = bb + 2 * e * RG_SP;
                const float n0 = be[0], o0 = be[-1], q0 = be[-2];             // next / own / previous line position, first half
                const float n1 = rowside ? be[32] : 0.f, o1 = rowside ? be[31] : 0.f, q1 = rowside ? be[30] : 0.f;
                float b[3][2];
                b[0][0] = (m_next[0] ? n0 : 0.f) + (fix_lo[0] ? q0 : 0.f); b[0][1] = (m_next[1] ? n1 : 0.f) + (fix_lo[1] ? q1 : 0.f);
                b[1][0] = o0; b[1][1] = o1;
                b[2][0] = (m_prev[0] ? q0 : 0.f) + (fix_hi[0] ? n0 : 0.f); b[2][1] = (m_prev[1] ? q1 : 0.f) + (fix_hi[1] ? n1 : 0.f);
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int k = 0; k < MTW; ++k) {
                        const float av = e < 4 ? a[sub][k][j][0][e & 3] : a[sub][k][j][1][e & 3];
                        acc[k][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][0], acc[k][0], 0, 0, 0);
                        if (rowside) acc[k][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][1], acc[k][1], 0, 0, 0);
                    }
            }
        }
    }
    if (p.diag & 1) return;
    // ---- dX[target] += acc: lane owns pixel column (lane & 31) of each half, rows 4 (lane >> 5) + (r & 3) + 8 (r >> 2)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long pp = p0 + t * 32 + (lane & 31);
        if (t >= nhalf || pp >= npix) continue;
        const int n = (int)(pp / L), i = (int)(pp - (long)n * L);
        if (p.R) {                                             // buffer mode: plain coalesced stores, every line position
#pragma unroll
            for (int k = 0; k < MTW; ++k) {
                if (t32_0 + k >= p.mt32) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (t32_0 + k) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M) p.R[(((long)n * p.M + m) * 4 + side) * p.r_l + i] = acc[k][t][r];
                }
            }
            continue;
        }
        if (i == 1 || i == L - 2) continue;                    // the corner targets of the plane: the corner part's
        const int ty = side == 0 ? 1 : (side == 1 ? H - 2 : i), tx = side == 2 ? 1 : (side == 3 ? W - 2 : i);
        float* __restrict__ d = p.dX + (long)n * p.M * HW + (long)ty * W + tx;
#pragma unroll
        for (int k = 0; k < MTW; ++k) {
            if (t32_0 + k >= p.mt32) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (t32_0 + k) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M) d[(long)m * HW] += acc[k][t][r];
            }
        }
    }
}

// dX [N][M][H][W] += the ring terms of the reflect-pad-1 data gradient (see the head of this file); dX must already hold the
// zero-padded "same" data gradient.  apack = c2m_ring_pack(w); dY [N][C][H][W]; w native [C][M][3][3] (unused since the corner
// part reads the packed fragments too; kept in the signature for the reference-side binding).  H, W >= 4.
static int ring_launch(const float* apack, const float* dY, float* dX, float* R, int r_l, int N, int C, int M, int H, int W,
                       void* stream) {
    if (N <= 0 || C <= 0 || M <= 0) return 0;
    if (H < 4 || W < 4) return (int)hipErrorInvalidValue;
    const long dy_bytes = 4L * N * C * H * W, dx_bytes = 4L * N * M * H * W;
    if (dy_bytes >= 0x80000000L || dx_bytes >= 0x80000000L) return (int)hipErrorInvalidValue;
    if ((((uintptr_t)apack) & 15) != 0) return (int)hipErrorInvalidValue;
    if (R && (r_l < H || r_l < W || (r_l & 3) || (((uintptr_t)R) & 15))) return (int)hipErrorInvalidValue;
    RingP p;
    p.A = apack; p.dY = dY; p.dX = dX; p.R = R; p.r_l = r_l;
    p.N = N; p.C = C; p.M = M; p.H = H; p.W = W;
    p.nchunks = c2m_cdiv(C, RG_CK); p.mt32 = c2m_cdiv(M, 32);
    // 128 rows per workgroup (MTW = 1): the column sides are bound by their divergent gathers and read-modify-writes (one 128-byte
    // line per element), which only more workgroups spread -- 256-row groups (MTW = 2) halved them and measured slower
    const int mtw = 1;
    p.mgroups = c2m_cdiv(p.mt32, 4 * mtw);
    p.pt[0] = p.pt[1] = c2m_cdiv((long)N * W, 64);
    p.pt[2] = p.pt[3] = c2m_cdiv((long)N * H, 32);
    p.gemm_blocks = (p.pt[0] + p.pt[1] + p.pt[2] + p.pt[3]) * p.mgroups;
    p.corner_blocks = R ? 0 : N * 4;                            // buffer mode: the row terms carry the corners (see the kernel)
    p.dy_bytes = (unsigned)dy_bytes;
    static const int diag = [] { const char* e = getenv("C2M_RING_DIAG"); return e ? atoi(e) : 0; }();
    p.diag = diag;
    static const int part = [] { const char* e = getenv("C2M_RING_PART"); return e ? atoi(e) : 0; }();      // timing diagnostics only
    if (part == 1) p.corner_blocks = 0;                         // GEMM part only (wrong corner targets)
    if (part == 2) { p.gemm_blocks = 0; p.pt[0] = p.pt[1] = p.pt[2] = p.pt[3] = 0; }      // corner part only
    if (p.corner_blocks + p.gemm_blocks == 0) return 0;
    const dim3 grid((unsigned)(p.corner_blocks + p.gemm_blocks));
    if (mtw == 1) hipLaunchKernelGGL(reflect_ring_dgrad_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else          hipLaunchKernelGGL(reflect_ring_dgrad_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

C2M_API int c2m_reflect_ring_dgrad(const float* apack, const float* w, const float* dY, float* dX, int N, int C, int M, int H,
                                   int W, void* stream) {
    C2M_ENTER();
    (void)w;
    return ring_launch(apack, dY, dX, nullptr, 0, N, C, M, H, W, stream);
}

// Buffer mode (round 5, second form): the ring terms are WRITTEN (coalesced, no read-modify-write, no corner part) into
// R [N][M][4][r_l] -- side 0 / 1: what to add to row 1 / H-2 at column x; side 2 / 3: to column 1 / W-2 at row y; r_l >= max(H, W),
// r_l % 4 == 0 -- and the Winograd launch that FOLLOWS on the same stream adds them in its epilogue (c2m_conv_wino / c2m_conv_wino4:
// geom[C2M_WG_RING], geom[C2M_WG_RING_L]).  The in-place form spent 19 of its 50 us reading and re-writing one float per 128-byte
// line of dX along the two columns.
C2M_API int c2m_reflect_ring_buffer(const float* apack, const float* dY, float* R, int N, int C, int M, int H, int W, int r_l,
                                    void* stream) {
    C2M_ENTER();
    if (!R) return (int)hipErrorInvalidValue;
    return ring_launch(apack, dY, nullptr, R, r_l, N, C, M, H, W, stream);
}
