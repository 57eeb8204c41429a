// Optical-flow feature warping and resampling kernels (HBM-bound gathers), gfx950.
//
//   c2m_flow_warp_fwd / _bwd   utils/ops.py:183-202  resample(): get_grid (align_corners=True convention) + pixel
//                              flow, sampled by grid_sample(bilinear, border, align_corners=False); optional fused
//                              "* occlusion" (motion_autoencoder.py:125, model.py:208).
//   c2m_upsample2x_fwd / _bwd  nn.Upsample(scale_factor=2, bilinear) (up_block.py:10, generator.py:77)
//   c2m_resize_bilinear        F.interpolate(size, bilinear) both align modes (utils.py:349, motion_autoencoder.py:123)
//   c2m_maxpool2x2_fwd / _bwd  VGG-19 pools (vgg.py via torchvision features 4,9,18,27)
//
// This file is compiled with -ffp-contract=off: the coordinate arithmetic follows the exact fp32 operation order
// of ATen's CPU kernels (oracle/c2m_oracle_index.c documents and pins it), every fused multiply-add is explicit.
// Layout NCHW; a wave covers 64 consecutive x of one row, so the 4 bilinear taps of a smooth flow are coalesced
// row segments; coordinates/weights are computed once per pixel and reused across a chunk of channels.
#include "common.h"
#include "dtype.h"

__device__ __forceinline__ float lin_m1_1(int i, int steps) {
    if (steps <= 1) return -1.0f;
    const float step = 2.0f / (float)(steps - 1);
    const int half = steps / 2;
    return i < half ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

struct WarpCoord {
    int x0, y0, x1, y1;
    float nw, ne, sw, se;   // weights of (y0,x0) (y0,x1) (y1,x0) (y1,x1)
    float w_, e_, n_, s_;
    bool okx1, oky1;
    float gmx, gmy;         // d(ix)/d(flow_x), d(iy)/d(flow_y) incl. the border-clip gate
};

__device__ __forceinline__ WarpCoord warp_coord(float fx, float fy, int x, int y, int H, int W) {
    WarpCoord c;
    const float cx = (float)(((double)W - 1.0) / 2.0), cy = (float)(((double)H - 1.0) / 2.0);
    const float gx = lin_m1_1(x, W) + fx / cx;
    const float gy = lin_m1_1(y, H) + fy / cy;
    float ix = fmaf(gx + 1.0f, (float)W / 2.0f, -0.5f);
    float iy = fmaf(gy + 1.0f, (float)H / 2.0f, -0.5f);
    // clip_coordinates_set_grad (ATen GridSampler.h): gradient gate is 0 on/outside the border
    c.gmx = (ix > 0.0f && ix < (float)(W - 1)) ? ((float)W / 2.0f) / cx : 0.0f;
    c.gmy = (iy > 0.0f && iy < (float)(H - 1)) ? ((float)H / 2.0f) / cy : 0.0f;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float xw = floorf(ix), yn = floorf(iy);
    c.w_ = ix - xw; c.e_ = 1.0f - c.w_; c.n_ = iy - yn; c.s_ = 1.0f - c.n_;
    c.nw = c.s_ * c.e_; c.ne = c.s_ * c.w_; c.sw = c.n_ * c.e_; c.se = c.n_ * c.w_;
    c.x0 = (int)xw; c.y0 = (int)yn; c.x1 = c.x0 + 1; c.y1 = c.y0 + 1;
    c.okx1 = c.x1 < W; c.oky1 = c.y1 < H;
    return c;
}

// grid: x = pixel blocks over N*H*W, y = channel chunk
template <class T>
__global__ void flow_warp_fwd_kernel(const T* __restrict__ img, const float* __restrict__ flow,
                                     const float* __restrict__ occ, T* __restrict__ out, int N, int C, int H, int W,
                                     int cchunk) {
    const long HW = (long)H * W;
    const long total = (long)N * HW;
    const int c0 = blockIdx.y * cchunk;
    const int c1 = min(c0 + cchunk, C);
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % W); const long r = p / W;
        const int y = (int)(r % H); const int n = (int)(r / H);
        const long sp = (long)y * W + x;
        const float fx = flow[((long)n * 2 + 0) * HW + sp], fy = flow[((long)n * 2 + 1) * HW + sp];
        const WarpCoord k = warp_coord(fx, fy, x, y, H, W);
        const float o = occ ? occ[(long)n * HW + sp] : 1.0f;
        const long i00 = (long)k.y0 * W + k.x0, i01 = (long)k.y0 * W + k.x1, i10 = (long)k.y1 * W + k.x0,
                   i11 = (long)k.y1 * W + k.x1;
        for (int c = c0; c < c1; ++c) {
            const T* __restrict__ pl = img + ((long)n * C + c) * HW;
            const float vnw = c2m_ld(pl, i00);
            const float vne = k.okx1 ? c2m_ld(pl, i01) : 0.0f;
            const float vsw = k.oky1 ? c2m_ld(pl, i10) : 0.0f;
            const float vse = (k.okx1 && k.oky1) ? c2m_ld(pl, i11) : 0.0f;
            float v = vnw * k.nw;
            v = fmaf(vne, k.ne, v);
            v = fmaf(vsw, k.sw, v);
            v = fmaf(vse, k.se, v);
            if (occ) v = v * o;
            c2m_st(out, ((long)n * C + c) * HW + sp, v);
        }
    }
}

// ---- backward, deterministic (no float atomics) -------------------------------------------------------------------
// d(image): a scatter in the reference (grid_sampler_2d_backward adds g*w into the 4 taps of every output pixel).  Here
// it is a GATHER over an inverted tap list that is built once per image n and shared by all C channels (the flow does not
// depend on the channel): count taps per source pixel -> exclusive scan -> fill (dest, corner) keys + weights -> sort each
// bucket by key -> every source pixel sums its contributions in ascending (dest, corner) order.  Integer atomics only
// place the entries; the fp32 summation order is fixed, so gradients are bit-reproducible run to run.
// d(flow): each output pixel owns its two values; when channels are chunked over gridDim.y the chunks write partial
// sums that a second kernel adds in chunk order.
struct WarpInvP {
    const float* flow;
    const float* occ;
    int N, H, W;
};

__global__ void warp_inv_count_kernel(const WarpInvP p, int* __restrict__ count) {
    const long HW = (long)p.H * p.W, total = (long)p.N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.W); const long r = i / p.W;
        const int y = (int)(r % p.H); const long n = r / p.H;
        const long sp = (long)y * p.W + x;
        const WarpCoord k = warp_coord(p.flow[(n * 2 + 0) * HW + sp], p.flow[(n * 2 + 1) * HW + sp], x, y, p.H, p.W);
        int* __restrict__ c = count + n * HW;
        atomicAdd(c + (long)k.y0 * p.W + k.x0, 1);
        if (k.okx1) atomicAdd(c + (long)k.y0 * p.W + k.x1, 1);
        if (k.oky1) atomicAdd(c + (long)k.y1 * p.W + k.x0, 1);
        if (k.okx1 && k.oky1) atomicAdd(c + (long)k.y1 * p.W + k.x1, 1);
    }
}

// exclusive scan of count within each image (one 1024-thread block per image); also copies offsets into cursor
__global__ __launch_bounds__(1024) void warp_inv_scan_kernel(const int* __restrict__ count, int* __restrict__ offset,
                                                             int* __restrict__ cursor, int HW) {
    __shared__ int part[1024];
    const long base = (long)blockIdx.x * HW;
    const int per = (HW + 1023) / 1024;
    const int beg = min((int)threadIdx.x * per, HW), end = min(beg + per, HW);
    int s = 0;
    for (int i = beg; i < end; ++i) s += count[base + i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = beg; i < end; ++i) {
        offset[base + i] = run;
        cursor[base + i] = run;
        run += count[base + i];
    }
}

__global__ void warp_inv_fill_kernel(const WarpInvP p, int* __restrict__ cursor, int* __restrict__ keys,
                                     float* __restrict__ vals) {
    const long HW = (long)p.H * p.W, total = (long)p.N * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.W); const long r = i / p.W;
        const int y = (int)(r % p.H); const long n = r / p.H;
        const int sp = y * p.W + x;
        const WarpCoord k = warp_coord(p.flow[(n * 2 + 0) * HW + sp], p.flow[(n * 2 + 1) * HW + sp], x, y, p.H, p.W);
        int* __restrict__ cur = cursor + n * HW;
        int* __restrict__ kk = keys + n * HW * 4;
        float* __restrict__ vv = vals + n * HW * 4;
        int slot = atomicAdd(cur + (long)k.y0 * p.W + k.x0, 1);
        kk[slot] = sp * 4 + 0; vv[slot] = k.nw;
        if (k.okx1) { slot = atomicAdd(cur + (long)k.y0 * p.W + k.x1, 1); kk[slot] = sp * 4 + 1; vv[slot] = k.ne; }
        if (k.oky1) { slot = atomicAdd(cur + (long)k.y1 * p.W + k.x0, 1); kk[slot] = sp * 4 + 2; vv[slot] = k.sw; }
        if (k.okx1 && k.oky1) { slot = atomicAdd(cur + (long)k.y1 * p.W + k.x1, 1); kk[slot] = sp * 4 + 3; vv[slot] = k.se; }
    }
}

// per source pixel: sort its bucket by key in place.  Buckets hold ~4 entries for a smooth flow: insertion sort by the
// owning thread up to WARP_SMALL_BUCKET entries.  Larger buckets (a diverged flow collapses whole rows or regions onto a
// border pixel: 1e4 - 1e5 entries) are queued for warp_inv_sort_big_kernel -- a single-thread insertion sort would be
// O(n^2) dependent global accesses, a multi-second kernel that looks like a hang.
constexpr int WARP_SMALL_BUCKET = 32;

__global__ void warp_inv_sort_kernel(const int* __restrict__ count, const int* __restrict__ offset,
                                     int* __restrict__ keys, float* __restrict__ vals, long nimg, int HW,
                                     int* __restrict__ nbig, int* __restrict__ biglist) {
    const long total = nimg * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long o = (i / HW) * (long)HW * 4 + offset[i];
        const int n = count[i];
        if (n > WARP_SMALL_BUCKET) { biglist[atomicAdd(nbig, 1)] = (int)i; continue; }   // at most 4*total/33 of them
        for (int a = 1; a < n; ++a) {
            const int ka = keys[o + a]; const float va = vals[o + a];
            int j = a;
            while (j > 0 && keys[o + j - 1] > ka) { keys[o + j] = keys[o + j - 1]; vals[o + j] = vals[o + j - 1]; --j; }
            keys[o + j] = ka; vals[o + j] = va;
        }
    }
}

// One 1024-thread workgroup per queued bucket: bitonic sorting network in place (global memory, O(n log^2 n) compare-
// exchanges spread over the workgroup).  The all-ascending form of the network (first step of every merge compares i with
// its mirror in the block, the rest are half-cleaners) lets the bucket be padded VIRTUALLY to a power of two with +inf
// keys behind its end: a pair whose upper index is >= n never swaps.  Keys are unique, so the result is the same
// permutation the insertion sort produces (the summation order stays fixed).
__global__ __launch_bounds__(1024) void warp_inv_sort_big_kernel(const int* __restrict__ count, const int* __restrict__ offset,
                                                                 int* __restrict__ keys, float* __restrict__ vals, int HW,
                                                                 const int* __restrict__ nbig, const int* __restrict__ biglist) {
    const int nb = *nbig;
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const long i = biglist[b];
        const long o = (i / HW) * (long)HW * 4 + offset[i];
        int* __restrict__ kk = keys + o;
        float* __restrict__ vv = vals + o;
        const int n = count[i];
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        auto cmpswap = [&](int lo, int hi) {
            if (hi < n) {
                const int a = kk[lo], c = kk[hi];
                if (a > c) { kk[lo] = c; kk[hi] = a; const float t = vv[lo]; vv[lo] = vv[hi]; vv[hi] = t; }
            }
        };
        for (int k = 2; k <= np2; k <<= 1) {
            const int h = k >> 1;
            for (int t = threadIdx.x; t < (np2 >> 1); t += blockDim.x) {
                const int blk = t / h, r = t - blk * h;
                cmpswap(blk * k + r, blk * k + k - 1 - r);
            }
            __syncthreads();
            for (int j = k >> 2; j >= 1; j >>= 1) {
                for (int t = threadIdx.x; t < (np2 >> 1); t += blockDim.x) {
                    const int lo = (t / j) * 2 * j + (t % j);
                    cmpswap(lo, lo + j);
                }
                __syncthreads();
            }
        }
    }
}

// grid: x = source-pixel blocks over N*H*W, y = channel chunk.  Every element of gimg is written (no zero-init needed).
template <class T>
__global__ void flow_warp_bwd_img_kernel(const T* __restrict__ gout, const float* __restrict__ occ,
                                         const int* __restrict__ count, const int* __restrict__ offset,
                                         const int* __restrict__ keys, const float* __restrict__ vals,
                                         T* __restrict__ gimg, int N, int C, int HW, int cchunk) {
    constexpr int FAST = 8;
    const long total = (long)N * HW;
    const int c0 = blockIdx.y * cchunk;
    const int c1 = min(c0 + cchunk, C);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / HW; const int sp = (int)(i - n * HW);
        const int cnt = count[i];
        const long o = n * (long)HW * 4 + offset[i];
        int d[FAST]; float w[FAST];
#pragma unroll
        for (int e = 0; e < FAST; ++e) {
            const bool ok = e < cnt;
            const int key = ok ? keys[o + e] : 0;
            d[e] = key >> 2;
            // the reference multiplies the incoming gradient by the occlusion value first: (g * occ) * weight
            w[e] = ok ? vals[o + e] : 0.0f;
        }
        float oc[FAST];
#pragma unroll
        for (int e = 0; e < FAST; ++e) oc[e] = (occ && e < cnt) ? occ[n * HW + d[e]] : 1.0f;
        for (int c = c0; c < c1; ++c) {
            const T* __restrict__ g = gout + (n * C + c) * (long)HW;
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < FAST; ++e)
                if (e < cnt) acc += (c2m_ld(g, d[e]) * oc[e]) * w[e];
            for (int e = FAST; e < cnt; ++e) {          // strongly converging flow: the tail of a long list
                const int de = keys[o + e] >> 2;
                acc += (c2m_ld(g, de) * (occ ? occ[n * HW + de] : 1.0f)) * vals[o + e];
            }
            c2m_st(gimg, (n * C + c) * (long)HW + sp, acc);
        }
    }
}

// d(flow): partial sums per channel chunk -> gfpart[chunk][n][2][HW] (or straight into gflow when there is one chunk)
template <class T>
__global__ void flow_warp_bwd_flow_kernel(const T* __restrict__ img, const float* __restrict__ flow,
                                          const float* __restrict__ occ, const T* __restrict__ gout,
                                          float* __restrict__ dst, int N, int C, int H, int W, int cchunk) {
    const long HW = (long)H * W;
    const long total = (long)N * HW;
    const int c0 = blockIdx.y * cchunk;
    const int c1 = min(c0 + cchunk, C);
    float* __restrict__ out = dst + (long)blockIdx.y * N * 2 * HW;
    for (long p = blockIdx.x * (long)blockDim.x + threadIdx.x; p < total; p += (long)gridDim.x * blockDim.x) {
        const int x = (int)(p % W); const long r = p / W;
        const int y = (int)(r % H); const int n = (int)(r / H);
        const long sp = (long)y * W + x;
        const float fx = flow[((long)n * 2 + 0) * HW + sp], fy = flow[((long)n * 2 + 1) * HW + sp];
        const WarpCoord k = warp_coord(fx, fy, x, y, H, W);
        const float o = occ ? occ[(long)n * HW + sp] : 1.0f;
        const long i00 = (long)k.y0 * W + k.x0, i01 = (long)k.y0 * W + k.x1, i10 = (long)k.y1 * W + k.x0,
                   i11 = (long)k.y1 * W + k.x1;
        float gix = 0.0f, giy = 0.0f;
        for (int c = c0; c < c1; ++c) {
            const long pb = ((long)n * C + c) * HW;
            const float g = c2m_ld(gout, pb + sp) * o;
            const T* __restrict__ pl = img + pb;
            const float vnw = c2m_ld(pl, i00);
            const float vne = k.okx1 ? c2m_ld(pl, i01) : 0.0f;
            const float vsw = k.oky1 ? c2m_ld(pl, i10) : 0.0f;
            const float vse = (k.okx1 && k.oky1) ? c2m_ld(pl, i11) : 0.0f;
            gix += g * (k.s_ * (vne - vnw) + k.n_ * (vse - vsw));
            giy += g * (k.e_ * (vsw - vnw) + k.w_ * (vse - vne));
        }
        out[((long)n * 2 + 0) * HW + sp] = gix * k.gmx;
        out[((long)n * 2 + 1) * HW + sp] = giy * k.gmy;
    }
}

__global__ void warp_flow_chunk_sum_kernel(const float* __restrict__ part, float* __restrict__ gflow, long numel, int chunks) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < numel; i += (long)gridDim.x * blockDim.x) {
        float acc = part[i];
        for (int c = 1; c < chunks; ++c) acc += part[(long)c * numel + i];
        gflow[i] = acc;
    }
}

static void warp_grid(int N, int C, int H, int W, dim3& grid, int& cchunk) {
    const long pix = (long)N * H * W;
    int gx = c2m_grid(pix, 256);
    // enough threads to fill the chip: split channels while pixel-threads alone are fewer than ~256k
    int chunks = 1;
    while (chunks < C && (long)gx * 256 * chunks < 262144) chunks *= 2;
    if (chunks > C) chunks = C;
    cchunk = (C + chunks - 1) / chunks;
    grid = dim3(gx, (C + cchunk - 1) / cchunk);
}

C2M_API int c2m_flow_warp_fwd(const void* img, const float* flow, const float* occ, void* out, int N, int C, int H,
                              int W, int dt, void* stream) {
    C2M_ENTER();
    if ((long)N * C * H * W <= 0) return 0;
    dim3 grid; int cchunk;
    warp_grid(N, C, H, W, grid, cchunk);
    C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(flow_warp_fwd_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)img, flow,
                                           occ, (T*)out, N, C, H, W, cchunk););
    return (int)hipGetLastError();
}

// Workspace of c2m_flow_warp_bwd (bytes): the inverted tap list of d(image) (count, offset, cursor: N*HW ints each;
// keys, weights: 4*N*HW each; a queue of the buckets too large for the per-thread sort) and the per-chunk partial sums of d(flow).  Neither output needs to be zeroed.
C2M_API long c2m_flow_warp_bwd_workspace_bytes(int N, int C, int H, int W, int want_gimg, int want_gflow) {
    dim3 grid; int cchunk;
    warp_grid(N, C, H, W, grid, cchunk);
    const long px = (long)N * H * W;
    long b = 0;
    if (want_gimg) b += px * 4 * 3 + px * 4 * 4 * 2 + 16 + (px / 8 + 1) * 4;     // + queue of the large buckets
    if (want_gflow && grid.y > 1) b += (long)grid.y * px * 2 * 4;
    return b > 0 ? b : 4;
}

C2M_API int c2m_flow_warp_bwd(const void* img, const float* flow, const float* occ, const void* gout, void* gimg,
                              float* gflow, int N, int C, int H, int W, void* workspace, int dt, void* stream) {
    C2M_ENTER();
    if ((long)N * C * H * W <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid; int cchunk;
    warp_grid(N, C, H, W, grid, cchunk);
    const long HW = (long)H * W, px = (long)N * HW;
    char* ws = (char*)workspace;
    if (gimg) {
        int* nbig = (int*)ws;                 // [0]: number of queued large buckets (zeroed together with count)
        int* count = nbig + 4;
        int* offset = count + px;
        int* cursor = offset + px;
        int* keys = cursor + px;
        float* vals = (float*)(keys + px * 4);
        int* biglist = (int*)(vals + px * 4);
        ws += px * 4 * 3 + px * 4 * 4 * 2 + 16 + (px / 8 + 1) * 4;
        hipError_t e = c2m_zero_async(nbig, sizeof(int) * (px + 4), s);
        if (e != hipSuccess) return (int)e;
        WarpInvP p{flow, occ, N, H, W};
        const int g1 = c2m_grid(px, 256);
        hipLaunchKernelGGL(warp_inv_count_kernel, dim3(g1), dim3(256), 0, s, p, count);
        hipLaunchKernelGGL(warp_inv_scan_kernel, dim3((unsigned)N), dim3(1024), 0, s, count, offset, cursor, (int)HW);
        hipLaunchKernelGGL(warp_inv_fill_kernel, dim3(g1), dim3(256), 0, s, p, cursor, keys, vals);
        hipLaunchKernelGGL(warp_inv_sort_kernel, dim3(g1), dim3(256), 0, s, count, offset, keys, vals, (long)N, (int)HW,
                           nbig, biglist);
        hipLaunchKernelGGL(warp_inv_sort_big_kernel, dim3(64), dim3(1024), 0, s, count, offset, keys, vals, (int)HW, nbig,
                           biglist);
        C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(flow_warp_bwd_img_kernel<T>, grid, dim3(256), 0, s, (const T*)gout, occ, count,
                                               offset, keys, vals, (T*)gimg, N, C, (int)HW, cchunk););
    }
    if (gflow) {
        float* dst = grid.y > 1 ? (float*)ws : gflow;
        C2M_DISPATCH_DT(dt, hipLaunchKernelGGL(flow_warp_bwd_flow_kernel<T>, grid, dim3(256), 0, s, (const T*)img, flow, occ,
                                               (const T*)gout, dst, N, C, H, W, cchunk););
        if (grid.y > 1)
            hipLaunchKernelGGL(warp_flow_chunk_sum_kernel, dim3(c2m_grid(px * 2, 256)), dim3(256), 0, s, dst, gflow,
                               px * 2, (int)grid.y);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- bilinear resize
struct Lerp { int i0, i1; float l0, l1; };

// ATen upsample_bilinear2d source index: align_corners ? o*scale : max((o+0.5)*scale-0.5, 0)
__device__ __forceinline__ Lerp lerp_src(int o, int in, float scale, bool align) {
    float src = align ? (float)o * scale : fmaxf(((float)o + 0.5f) * scale - 0.5f, 0.0f);
    Lerp l;
    l.i0 = (int)src;                       // src >= 0
    if (l.i0 > in - 1) l.i0 = in - 1;
    l.i1 = l.i0 + (l.i0 < in - 1 ? 1 : 0);
    l.l1 = src - (float)l.i0;
    l.l0 = 1.0f - l.l1;
    return l;
}

// Index type of the element-parallel kernels below: 32-bit when the tensor allows (a 64-bit division per element costs
// more than the memory access it addresses).
#define C2M_IDX_DISPATCH(total, KERNEL, grid, stream, ...)                                                   \
    do {                                                                                                     \
        if ((total) < (1L << 31)) hipLaunchKernelGGL((KERNEL<unsigned>), grid, dim3(256), 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<long>), grid, dim3(256), 0, stream, __VA_ARGS__);                  \
    } while (0)

template <typename I, class T = float>
__global__ void resize_bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, long NC, int Hi, int Wi,
                                       int Ho, int Wo, float sh, float sw, int align, float mul) {
    const I total = (I)(NC * Ho * Wo);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int ox = (int)(i % (I)Wo); const I r = i / (I)Wo;
        const int oy = (int)(r % (I)Ho); const I nc = r / (I)Ho;
        const Lerp ly = lerp_src(oy, Hi, sh, align), lx = lerp_src(ox, Wi, sw, align);
        const T* __restrict__ p = in + (long)nc * Hi * Wi;
        const float v = ly.l0 * (lx.l0 * c2m_ld(p, ly.i0 * Wi + lx.i0) + lx.l1 * c2m_ld(p, ly.i0 * Wi + lx.i1)) +
                        ly.l1 * (lx.l0 * c2m_ld(p, ly.i1 * Wi + lx.i0) + lx.l1 * c2m_ld(p, ly.i1 * Wi + lx.i1));
        c2m_st(out, (long)i, v * mul);
    }
}

static float area_scale(int in, int out, int align, double scale_factor) {
    if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.0f;
    if (scale_factor > 0.0) return (float)(1.0 / scale_factor);
    return (float)in / (float)out;
}

static int resize_bilinear_launch(const void* in, void* out, long NC, int Hi, int Wi, int Ho, int Wo, int align,
                                  double scale_factor, int dt, void* stream) {
    const long total = NC * Ho * Wo;
    if (total <= 0) return 0;
    const float sh = area_scale(Hi, Ho, align, scale_factor), sw = area_scale(Wi, Wo, align, scale_factor);
    const dim3 grid(c2m_grid(total, 256));
    hipStream_t s = (hipStream_t)stream;
    if (dt == C2M_BF16) {
        if (total < (1L << 31)) hipLaunchKernelGGL((resize_bilinear_kernel<unsigned, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, NC, Hi, Wi, Ho, Wo, sh, sw, align, 1.0f);
        else hipLaunchKernelGGL((resize_bilinear_kernel<long, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, NC, Hi, Wi, Ho, Wo, sh, sw, align, 1.0f);
    } else {
        if (total < (1L << 31)) hipLaunchKernelGGL((resize_bilinear_kernel<unsigned, float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, NC, Hi, Wi, Ho, Wo, sh, sw, align, 1.0f);
        else hipLaunchKernelGGL((resize_bilinear_kernel<long, float>), grid, dim3(256), 0, s, (const float*)in, (float*)out, NC, Hi, Wi, Ho, Wo, sh, sw, align, 1.0f);
    }
    return (int)hipGetLastError();
}

C2M_API int c2m_resize_bilinear(const void* in, void* out, long NC, int Hi, int Wi, int Ho, int Wo, int align,
                                double scale_factor, int dt, void* stream) {
    C2M_ENTER();
    return resize_bilinear_launch(in, out, NC, Hi, Wi, Ho, Wo, align, scale_factor, dt, stream);
}

// x2 upsample (align_corners=False; up_block.py:10): one thread per INPUT pixel writes its 2x2 outputs as two 8-byte
// stores; the arithmetic is the generic kernel's expression with the same lerp_src weights -> bit-identical results.
template <typename I, class T = float>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, long NC, int Hi, int Wi) {
    const int Wo = 2 * Wi;
    const I total = (I)(NC * Hi * Wi);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int x = (int)(i % (I)Wi); const I r = i / (I)Wi;
        const int y = (int)(r % (I)Hi); const I nc = r / (I)Hi;
        const T* __restrict__ p = in + (long)nc * Hi * Wi;
        T* __restrict__ o = out + (long)nc * 4 * Hi * Wi + (long)(2 * y) * Wo + 2 * x;
        const Lerp lx0 = lerp_src(2 * x, Wi, 0.5f, false), lx1 = lerp_src(2 * x + 1, Wi, 0.5f, false);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const Lerp ly = lerp_src(2 * y + a, Hi, 0.5f, false);
            const T* __restrict__ r0 = p + ly.i0 * Wi;
            const T* __restrict__ r1 = p + ly.i1 * Wi;
            const float a00 = c2m_ld(r0, lx0.i0), a01 = c2m_ld(r0, lx0.i1), a10 = c2m_ld(r1, lx0.i0), a11 = c2m_ld(r1, lx0.i1);
            const float b00 = c2m_ld(r0, lx1.i0), b01 = c2m_ld(r0, lx1.i1), b10 = c2m_ld(r1, lx1.i0), b11 = c2m_ld(r1, lx1.i1);
            float2 v;
            v.x = ly.l0 * (lx0.l0 * a00 + lx0.l1 * a01) + ly.l1 * (lx0.l0 * a10 + lx0.l1 * a11);
            v.y = ly.l0 * (lx1.l0 * b00 + lx1.l1 * b01) + ly.l1 * (lx1.l0 * b10 + lx1.l1 * b11);
            c2m_st2(o + (long)a * Wo, v);
        }
    }
}

// The same x2 upsample on channel-blocked tensors (round 4): in [NCB][Hi][Wi][8] bf16 -> out [NCB][2Hi][2Wi][8] bf16 (NC8, conv_nc8.hip),
// for the up block whose convolution reads NC8 only (ops.upsample2x feeds=): one thread per input pixel and channel block loads the
// 2 x 2 source units of each of its four outputs (16 bytes = 8 channels each) and stores four 16-byte units.  Per channel the
// expression and the lerp_src weights of upsample2x_fwd_kernel -> the same bits as that kernel followed by the layout pass.
__device__ __forceinline__ void up_nc8_unpack(const uint4 v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__global__ __launch_bounds__(256) void upsample2x_nc8_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long NCB, int Hi,
                                                             int Wi) {
    const int Wo = 2 * Wi;
    const long total = NCB * Hi * Wi;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wi); const long r = i / Wi;
        const int y = (int)(r % Hi); const long ncb = r / Hi;
        const uint4* __restrict__ p = in + ncb * Hi * Wi;
        uint4* __restrict__ o = out + ncb * 4 * Hi * Wi + (long)(2 * y) * Wo + 2 * x;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const Lerp ly = lerp_src(2 * y + a, Hi, 0.5f, false);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const Lerp lx = lerp_src(2 * x + b, Wi, 0.5f, false);
                float a00[8], a01[8], a10[8], a11[8];
                up_nc8_unpack(p[ly.i0 * Wi + lx.i0], a00); up_nc8_unpack(p[ly.i0 * Wi + lx.i1], a01);
                up_nc8_unpack(p[ly.i1 * Wi + lx.i0], a10); up_nc8_unpack(p[ly.i1 * Wi + lx.i1], a11);
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
                    const float v0 = ly.l0 * (lx.l0 * a00[2 * e] + lx.l1 * a01[2 * e]) + ly.l1 * (lx.l0 * a10[2 * e] + lx.l1 * a11[2 * e]);
                    const float v1 = ly.l0 * (lx.l0 * a00[2 * e + 1] + lx.l1 * a01[2 * e + 1]) +
                                     ly.l1 * (lx.l0 * a10[2 * e + 1] + lx.l1 * a11[2 * e + 1]);
                    const bf16x2_t q = {(bf16_t)v0, (bf16_t)v1};
                    w[e] = __builtin_bit_cast(unsigned, q);
                }
                o[(long)a * Wo + b] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
}

// in / out: NC8 tensors (16-byte aligned), NCB = N * ceil(C/8)
C2M_API int c2m_upsample2x_nc8(const void* in, void* out, long NCB, int Hi, int Wi, void* stream) {
    C2M_ENTER();
    const long total = NCB * Hi * Wi;
    if (total <= 0) return 0;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(upsample2x_nc8_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)in,
                       (uint4*)out, NCB, Hi, Wi);
    return (int)hipGetLastError();
}

// adjoint of the x2 (align_corners=False, scale 0.5) upsample, gather form (deterministic, no atomics)
template <class T = float>
__device__ __forceinline__ float upsample2x_bwd_pixel(const T* __restrict__ g, int y, int x, int Hi, int Wi) {
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    float wy[4], wx[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int oy = 2 * y - 1 + d, ox = 2 * x - 1 + d;
        wy[d] = 0.0f; wx[d] = 0.0f;
        if (oy >= 0 && oy < Ho) { const Lerp l = lerp_src(oy, Hi, 0.5f, false); wy[d] = (l.i0 == y ? l.l0 : 0.0f) + (l.i1 == y ? l.l1 : 0.0f); }
        if (ox >= 0 && ox < Wo) { const Lerp l = lerp_src(ox, Wi, 0.5f, false); wx[d] = (l.i0 == x ? l.l0 : 0.0f) + (l.i1 == x ? l.l1 : 0.0f); }
    }
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int oy = 2 * y - 1 + a;
        if (wy[a] == 0.0f) continue;
        float row = 0.0f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ox = 2 * x - 1 + b;
            if (wx[b] != 0.0f) row += wx[b] * c2m_ld(g, oy * Wo + ox);
        }
        acc += wy[a] * row;
    }
    return acc;
}

template <typename I, class T = float>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ gout, T* __restrict__ gin, long NC, int Hi, int Wi) {
    const I total = (I)(NC * Hi * Wi);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int x = (int)(i % (I)Wi); const I r = i / (I)Wi;
        const int y = (int)(r % (I)Hi); const I nc = r / (I)Hi;
        c2m_st(gin, (long)i, upsample2x_bwd_pixel<T>(gout + (long)nc * (4L * Hi * Wi), y, x, Hi, Wi));
    }
}

// Even widths, 8-byte aligned tensors: each lane produces two adjacent gradients from four 8-byte loads per output row
// (the interior weights are the constants 1/4, 3/4, 3/4, 1/4 in both directions); image borders take the generic path.
template <typename I>
__global__ void upsample2x_bwd_pair_kernel(const float* __restrict__ gout, float* __restrict__ gin, long NC, int Hi, int Wi) {
    const int Wo = 2 * Wi, Wp = Wi / 2;
    const I total = (I)(NC * Hi * Wp);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int j = (int)(i % (I)Wp); const I r = i / (I)Wp;
        const int y = (int)(r % (I)Hi); const I nc = r / (I)Hi;
        const int x = 2 * j;
        const float* __restrict__ g = gout + (long)nc * (4L * Hi * Wi);
        float2 out;
        if (y >= 1 && y < Hi - 1 && j >= 1 && x + 2 < Wi) {
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float wy = (a == 0 || a == 3) ? 0.25f : 0.75f;
                const float2* __restrict__ rp = reinterpret_cast<const float2*>(g + (2 * y - 1 + a) * Wo + 2 * x - 2);
                const float2 v0 = rp[0], v1 = rp[1], v2 = rp[2], v3 = rp[3];
                float row0 = 0.0f, row1 = 0.0f;
                row0 += 0.25f * v0.y; row0 += 0.75f * v1.x; row0 += 0.75f * v1.y; row0 += 0.25f * v2.x;
                row1 += 0.25f * v1.y; row1 += 0.75f * v2.x; row1 += 0.75f * v2.y; row1 += 0.25f * v3.x;
                acc0 += wy * row0; acc1 += wy * row1;
            }
            out = make_float2(acc0, acc1);
        } else {
            out = make_float2(upsample2x_bwd_pixel(g, y, x, Hi, Wi), upsample2x_bwd_pixel(g, y, x + 1, Hi, Wi));
        }
        *reinterpret_cast<float2*>(gin + ((long)nc * Hi + y) * Wi + x) = out;
    }
}

// Large maps: one workgroup = an 8 x 64 tile of the input-resolution gradient; the 18 x 130 window of `gout` it needs is
// staged in LDS once with coalesced loads (the gather forms above fetch every gout element four times through the L1),
// then each lane combines two outputs from 8-byte LDS reads.  Same weights as upsample2x_bwd_pixel (out-of-range window
// elements are stored as 0 and carry weight 0).
// UB_TX = 64 (two outputs per lane) for wide maps, 32 (one per lane) for the 32- and 48-wide ones.
constexpr int UB_TY = 8, UB_LR = 2 * UB_TY + 2;
template <int UB_TX, class T = float>
__global__ __launch_bounds__(256) void upsample2x_bwd_tile_kernel(const T* __restrict__ gout, T* __restrict__ gin,
                                                                  int Hi, int Wi) {
    constexpr int UB_LC = 2 * UB_TX + 2, UB_LS = UB_LC + 2;
    __shared__ __attribute__((aligned(8))) float tile[UB_LR][UB_LS];
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int tiles_x = (Wi + UB_TX - 1) / UB_TX, tiles_y = (Hi + UB_TY - 1) / UB_TY;
    unsigned tb = blockIdx.x;
    const int tx = tb % tiles_x; tb /= tiles_x;
    const int ty = tb % tiles_y; const unsigned nc = tb / tiles_y;
    const int y0 = ty * UB_TY, x0 = tx * UB_TX;
    const int gy0 = 2 * y0 - 1, gx0 = 2 * x0 - 1;
    const T* __restrict__ g = gout + (long)nc * Ho * Wo;
    // tile column c is output column gx0 + c with gx0 = 2 x0 - 1 odd: columns 1 .. UB_LC - 2 are (even, odd) pairs -> one
    // aligned 2-element load each (8 B of fp32 / 4 B of bf16; Wo is even, so a pair is inside the row or outside as a whole),
    // columns 0 and UB_LC - 1 single elements.  (The first form fetched every element with its own 4-byte load.)
    constexpr int UB_UNITS = UB_LC / 2 + 1;                   // per row: [col 0] [pairs (1,2) .. (UB_LC-3, UB_LC-2)] [col UB_LC-1]
    for (int e = threadIdx.x; e < UB_LR * UB_UNITS; e += 256) {
        const int r = e / UB_UNITS, u = e - r * UB_UNITS;
        const int gy = gy0 + r;
        const bool rok = (unsigned)gy < (unsigned)Ho;
        if (u == 0 || u == UB_UNITS - 1) {
            const int c = u == 0 ? 0 : UB_LC - 1, gx = gx0 + c;
            tile[r][c] = (rok && (unsigned)gx < (unsigned)Wo) ? c2m_ld(g, gy * Wo + gx) : 0.0f;
        } else {
            const int c = 2 * u - 1, gx = gx0 + c;            // gx even
            float2 v = make_float2(0.0f, 0.0f);
            if (rok && (unsigned)gx < (unsigned)Wo) v = c2m_ld2(g + (long)gy * Wo + gx);
            tile[r][c] = v.x; tile[r][c + 1] = v.y;
        }
    }
    __syncthreads();
    const int ly = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int y = y0 + ly;
    if (y >= Hi) return;
    float wy[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int oy = 2 * y - 1 + d;
        wy[d] = 0.0f;
        if (oy >= 0 && oy < Ho) { const Lerp q = lerp_src(oy, Hi, 0.5f, false); wy[d] = (q.i0 == y ? q.l0 : 0.0f) + (q.i1 == y ? q.l1 : 0.0f); }
    }
#pragma unroll
    for (int h = 0; h < UB_TX / 32; ++h) {
        const int lx = l + 32 * h, x = x0 + lx;
        if (x >= Wi) continue;
        float wx[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int ox = 2 * x - 1 + d;
            wx[d] = 0.0f;
            if (ox >= 0 && ox < Wo) { const Lerp q = lerp_src(ox, Wi, 0.5f, false); wx[d] = (q.i0 == x ? q.l0 : 0.0f) + (q.i1 == x ? q.l1 : 0.0f); }
        }
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float2 v0 = *reinterpret_cast<const float2*>(&tile[2 * ly + a][2 * lx]);
            const float2 v1 = *reinterpret_cast<const float2*>(&tile[2 * ly + a][2 * lx + 2]);
            float row = 0.0f;
            row += wx[0] * v0.x; row += wx[1] * v0.y; row += wx[2] * v1.x; row += wx[3] * v1.y;
            acc += wy[a] * row;
        }
        c2m_st(gin, ((long)nc * Hi + y) * Wi + x, acc);
    }
}

C2M_API int c2m_upsample2x_fwd(const void* in_, void* out_, long NC, int Hi, int Wi, int dt, void* stream) {
    C2M_ENTER();
    const long total = NC * Hi * Wi;
    if (total <= 0) return 0;
    if ((((uintptr_t)out_) & 7) != 0) return resize_bilinear_launch(in_, out_, NC, Hi, Wi, 2 * Hi, 2 * Wi, 0, 2.0, dt, stream);
    if (dt == C2M_BF16) {                  // same kernel on 2-byte elements: 4-byte stores of output pixel pairs
        const dim3 grid(c2m_grid(total, 256));
        if (total * 4 < (1L << 31)) hipLaunchKernelGGL((upsample2x_fwd_kernel<unsigned, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (bf16_t*)out_, NC, Hi, Wi);
        else hipLaunchKernelGGL((upsample2x_fwd_kernel<long, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (bf16_t*)out_, NC, Hi, Wi);
        return (int)hipGetLastError();
    }
    const float* in = (const float*)in_; float* out = (float*)out_;
    C2M_IDX_DISPATCH(total * 4, upsample2x_fwd_kernel, dim3(c2m_grid(total, 256)), (hipStream_t)stream, in, out, NC, Hi, Wi);
    return (int)hipGetLastError();
}

C2M_API int c2m_upsample2x_bwd(const void* gout_, void* gin_, long NC, int Hi, int Wi, int dt, void* stream) {
    C2M_ENTER();
    const long total = NC * Hi * Wi;
    if (total <= 0) return 0;
    if (dt == C2M_BF16) {
        if (Wi >= 32 && Hi >= 8 && total * 4 < (1L << 31) && (((uintptr_t)gout_) & 3) == 0) {   // LDS-tiled form (fp32 window in LDS; 2-element loads)
            const int tx = Wi >= 64 ? 64 : 32;
            const long tiles = (long)c2m_cdiv(Wi, tx) * c2m_cdiv(Hi, UB_TY) * NC;
            if (tiles < (1L << 31)) {
                if (tx == 64) hipLaunchKernelGGL((upsample2x_bwd_tile_kernel<64, bf16_t>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gout_, (bf16_t*)gin_, Hi, Wi);
                else hipLaunchKernelGGL((upsample2x_bwd_tile_kernel<32, bf16_t>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gout_, (bf16_t*)gin_, Hi, Wi);
                return (int)hipGetLastError();
            }
        }
        const dim3 grid(c2m_grid(total, 256));
        if (total * 4 < (1L << 31)) hipLaunchKernelGGL((upsample2x_bwd_kernel<unsigned, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gout_, (bf16_t*)gin_, NC, Hi, Wi);
        else hipLaunchKernelGGL((upsample2x_bwd_kernel<long, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gout_, (bf16_t*)gin_, NC, Hi, Wi);
        return (int)hipGetLastError();
    }
    const float* gout = (const float*)gout_; float* gin = (float*)gin_;
    if (Wi >= 32 && Hi >= 8 && total * 4 < (1L << 31) && (((uintptr_t)gout) & 7) == 0) {     // (2-element loads of the window)
        const int tx = Wi >= 64 ? 64 : 32;
        const long tiles = (long)c2m_cdiv(Wi, tx) * c2m_cdiv(Hi, UB_TY) * NC;
        if (tiles < (1L << 31)) {
            if (tx == 64)
                hipLaunchKernelGGL(upsample2x_bwd_tile_kernel<64>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                                   gout, gin, Hi, Wi);
            else
                hipLaunchKernelGGL(upsample2x_bwd_tile_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream,
                                   gout, gin, Hi, Wi);
            return (int)hipGetLastError();
        }
    }
    if ((Wi & 1) == 0 && Wi >= 4 && ((((uintptr_t)gout) | ((uintptr_t)gin)) & 7) == 0) {
        C2M_IDX_DISPATCH(total * 4, upsample2x_bwd_pair_kernel, dim3(c2m_grid(total / 2, 256)), (hipStream_t)stream, gout, gin, NC, Hi, Wi);
    } else {
        C2M_IDX_DISPATCH(total * 4, upsample2x_bwd_kernel, dim3(c2m_grid(total, 256)), (hipStream_t)stream, gout, gin, NC, Hi, Wi);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- maxpool 2x2 / 2
template <typename I, class T = float>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, long NC, int Hi, int Wi) {
    const int Ho = Hi / 2, Wo = Wi / 2;
    const I total = (I)(NC * Ho * Wo);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int ox = (int)(i % (I)Wo); const I r = i / (I)Wo;
        const int oy = (int)(r % (I)Ho); const I nc = r / (I)Ho;
        const T* __restrict__ p = in + (long)nc * Hi * Wi + (2 * oy) * Wi + 2 * ox;
        c2m_st(out, (long)i, fmaxf(fmaxf(c2m_ld(p, 0), c2m_ld(p, 1)), fmaxf(c2m_ld(p, Wi), c2m_ld(p, Wi + 1))));
    }
}

// gradient goes to the first maximum in (row, col) scan order, like ATen's max_pool2d_with_indices
template <typename I, class T = float>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ in, const T* __restrict__ gout,
                                    T* __restrict__ gin, long NC, int Hi, int Wi, int relu) {
    const int Ho = Hi / 2, Wo = Wi / 2;
    const I total = (I)(NC * Hi * Wi);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int x = (int)(i % (I)Wi); const I r = i / (I)Wi;
        const int y = (int)(r % (I)Hi); const I nc = r / (I)Hi;
        const int oy = y / 2, ox = x / 2;
        float g = 0.0f;
        if (oy < Ho && ox < Wo) {
            const T* __restrict__ p = in + (long)nc * Hi * Wi + (2 * oy) * Wi + 2 * ox;
            const float v[4] = {c2m_ld(p, 0), c2m_ld(p, 1), c2m_ld(p, Wi), c2m_ld(p, Wi + 1)};
            int arg = 0; float m = v[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; arg = k; }
            if (arg == (y & 1) * 2 + (x & 1) && !(relu && !(m > 0.f))) g = c2m_ld(gout, (long)nc * Ho * Wo + oy * Wo + ox);
        }
        c2m_st(gin, (long)i, g);
    }
}

C2M_API int c2m_maxpool2x2_fwd(const void* in_, void* out_, long NC, int Hi, int Wi, int dt, void* stream) {
    C2M_ENTER();
    const long total = NC * (Hi / 2) * (Wi / 2);
    if (total <= 0) return 0;
    if (dt == C2M_BF16) {
        const dim3 grid(c2m_grid(total, 256));
        if (NC * Hi * Wi < (1L << 31)) hipLaunchKernelGGL((maxpool2_fwd_kernel<unsigned, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (bf16_t*)out_, NC, Hi, Wi);
        else hipLaunchKernelGGL((maxpool2_fwd_kernel<long, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (bf16_t*)out_, NC, Hi, Wi);
        return (int)hipGetLastError();
    }
    const float* in = (const float*)in_; float* out = (float*)out_;
    C2M_IDX_DISPATCH(NC * Hi * Wi, maxpool2_fwd_kernel, dim3(c2m_grid(total, 256)), (hipStream_t)stream, in, out, NC, Hi, Wi);
    return (int)hipGetLastError();
}

// even sizes, 8-byte aligned tensors: one thread per 2x2 window (every input is read once: a thread per input pixel reads
// each window four times), two 8-byte loads + one gradient, two 8-byte stores
template <typename I, class T = float>
__global__ void maxpool2_bwd_win_kernel(const T* __restrict__ in, const T* __restrict__ gout,
                                        T* __restrict__ gin, long NC, int Hi, int Wi, int relu) {
    const int Ho = Hi / 2, Wo = Wi / 2;
    const I total = (I)(NC * Ho * Wo);
    for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
        const int ox = (int)(i % (I)Wo); const I r = i / (I)Wo;
        const int oy = (int)(r % (I)Ho); const I nc = r / (I)Ho;
        const long base = (long)nc * Hi * Wi + (long)(2 * oy) * Wi + 2 * ox;
        const float2 a = c2m_ld2(in + base), b = c2m_ld2(in + base + Wi);
        const float v[4] = {a.x, a.y, b.x, b.y};
        int arg = 0; float m = v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; arg = k; }
        const float g = (relu && !(m > 0.f)) ? 0.f : c2m_ld(gout, (long)i);     // relu: `in` is a ReLU output, gin its pre-activation's
        c2m_st2(gin + base, make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f));
        c2m_st2(gin + base + Wi, make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f));
    }
}

static int maxpool2x2_bwd_impl(const void* in_, const void* gout_, void* gin_, long NC, int Hi, int Wi, int dt, void* stream,
                               int relu) {
    const long total = NC * Hi * Wi;
    if (total <= 0) return 0;
    if (dt == C2M_BF16) {
        if ((Hi & 1) == 0 && (Wi & 1) == 0 && ((((uintptr_t)in_) | ((uintptr_t)gin_)) & 3) == 0 && total < (1L << 31)) {
            hipLaunchKernelGGL((maxpool2_bwd_win_kernel<unsigned, bf16_t>), dim3(c2m_grid(total / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)in_, (const bf16_t*)gout_, (bf16_t*)gin_, NC, Hi, Wi, relu);
            return (int)hipGetLastError();
        }
        const dim3 grid(c2m_grid(total, 256));
        if (total < (1L << 31)) hipLaunchKernelGGL((maxpool2_bwd_kernel<unsigned, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (const bf16_t*)gout_, (bf16_t*)gin_, NC, Hi, Wi, relu);
        else hipLaunchKernelGGL((maxpool2_bwd_kernel<long, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in_, (const bf16_t*)gout_, (bf16_t*)gin_, NC, Hi, Wi, relu);
        return (int)hipGetLastError();
    }
    const float* in = (const float*)in_; const float* gout = (const float*)gout_; float* gin = (float*)gin_;
    if ((Hi & 1) == 0 && (Wi & 1) == 0 && ((((uintptr_t)in) | ((uintptr_t)gin)) & 7) == 0) {
        C2M_IDX_DISPATCH(total, maxpool2_bwd_win_kernel, dim3(c2m_grid(total / 4, 256)), (hipStream_t)stream, in, gout, gin, NC, Hi, Wi, relu);
        return (int)hipGetLastError();
    }
    C2M_IDX_DISPATCH(total, maxpool2_bwd_kernel, dim3(c2m_grid(total, 256)), (hipStream_t)stream, in, gout, gin, NC, Hi, Wi, relu);
    return (int)hipGetLastError();
}

C2M_API int c2m_maxpool2x2_bwd(const void* in_, const void* gout_, void* gin_, long NC, int Hi, int Wi, int dt, void* stream) {
    C2M_ENTER();
    return maxpool2x2_bwd_impl(in_, gout_, gin_, NC, Hi, Wi, dt, stream, 0);
}

// `in` is the OUTPUT of a ReLU (layers/vgg.py: conv -> ReLU -> MaxPool2d): gin is the gradient of the ReLU's INPUT, i.e. the pool
// backward times (in > 0) -- the window maximum of a ReLU output is 0 only where the whole window is, so the mask is one
// compare on the value the kernel already holds, and the separate activation-backward pass over the full-resolution tensor
// (read, read, write) goes away.
C2M_API int c2m_maxpool2x2_relu_bwd(const void* in_, const void* gout_, void* gin_, long NC, int Hi, int Wi, int dt, void* stream) {
    C2M_ENTER();
    return maxpool2x2_bwd_impl(in_, gout_, gin_, NC, Hi, Wi, dt, stream, 1);
}

// ------------------------------------------------------------------------------------------- RoIAlign
// torchvision.ops.roi_align(aligned=False, sampling_ratio=-1) as called at appearance_encoder.py:67-69.
// feat [N,C,H,W]; boxes [K,5] = (batch index, x1, y1, x2, y2) on the DEVICE (no host round trip for the adaptive
// sampling grid); out [K,C,PH,PW].  torchvision's backward scatters with float atomics; ours GATHERS per feature pixel in a
// fixed (box, sample row, sample column) order (roi_align_bwd_gather_kernel below): no atomics, bit-repeatable.
struct RoiSample { int yl, yh, xl, xh; float w1, w2, w3, w4; bool ok; };

__device__ __forceinline__ RoiSample roi_sample(float y, float x, int H, int W) {
    RoiSample s;
    s.ok = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    s.yl = (int)y; s.xl = (int)x;
    if (s.yl >= H - 1) { s.yh = s.yl = H - 1; y = (float)s.yl; } else s.yh = s.yl + 1;
    if (s.xl >= W - 1) { s.xh = s.xl = W - 1; x = (float)s.xl; } else s.xh = s.xl + 1;
    const float ly = y - (float)s.yl, lx = x - (float)s.xl, hy = 1.f - ly, hx = 1.f - lx;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    return s;
}

__global__ void roi_align_kernel(const float* __restrict__ feat, const float* __restrict__ boxes,
                                 float* __restrict__ out_or_gfeat, int K, int C, int H, int W, int PH, int PW, float scale) {
    const long total = (long)K * C * PH * PW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pw = (int)(i % PW); long r = i / PW;
        const int ph = (int)(r % PH); r /= PH;
        const int c = (int)(r % C); const int k = (int)(r / C);
        const float* __restrict__ bx = boxes + (long)k * 5;
        const int b = (int)bx[0];
        const float x1 = bx[1] * scale, y1 = bx[2] * scale, x2 = bx[3] * scale, y2 = bx[4] * scale;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bh = rh / (float)PH, bw = rw / (float)PW;
        const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
        const float cnt = fmaxf((float)(gh * gw), 1.f);
        const long plane = ((long)b * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                const RoiSample s = roi_sample(y, x, H, W);
                if (!s.ok) continue;
                const float* __restrict__ f = feat + plane;
                acc += s.w1 * f[(long)s.yl * W + s.xl] + s.w2 * f[(long)s.yl * W + s.xh] +
                       s.w3 * f[(long)s.yh * W + s.xl] + s.w4 * f[(long)s.yh * W + s.xh];
            }
        }
        out_or_gfeat[i] = acc / cnt;
    }
}

C2M_API int c2m_roi_align_fwd(const float* feat, const float* boxes, float* out, int K, int C, int H, int W, int PH,
                              int PW, float spatial_scale, void* stream) {
    C2M_ENTER();
    const long total = (long)K * C * PH * PW;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(roi_align_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, feat, boxes, out, K, C,
                       H, W, PH, PW, spatial_scale);
    return (int)hipGetLastError();
}

// 1-D part of roi_sample for one coordinate: which of the (at most two) taps of sample coordinate v is pixel `p`, and with
// which weight.  Returns the number of terms (0..2; 2 when the clamped taps coincide on the last pixel).
__device__ __forceinline__ int roi_taps_1d(float v, int S, int p, float (&w)[2]) {
    if (v < -1.0f || v > (float)S) return 0;
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= S - 1) { hi = lo = S - 1; v = (float)lo; } else hi = lo + 1;
    const float l = v - (float)lo, h = 1.f - l;
    int n = 0;
    if (lo == p) w[n++] = h;
    if (hi == p) w[n++] = l;
    return n;
}

// d(feat)[b,c,y,x] = sum over the boxes of image b, the sample rows that touch y and the sample columns that touch x of
// (gout[k,c,ph,pw] / count) * (wy * wx), in ascending (k, sample row, sample column) order.  One thread per feature pixel and
// channel chunk (blockIdx.y); every element of gfeat is written (no zero-init).
template <int CH>
__global__ void roi_align_bwd_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ gout,
                                            float* __restrict__ gfeat, int N, int K, int C, int H, int W, int PH, int PW,
                                            float scale) {
    const long total = (long)N * H * W;
    const int c0 = blockIdx.y * CH;
    // the box list in LDS (every thread walks all K boxes: as dependent global loads that walk was the whole 0.3 ms of this launch)
    constexpr int KMAX = 512;
    __shared__ float sbox[KMAX * 5];
    const bool staged = K <= KMAX;
    if (staged) {
        for (int j = threadIdx.x; j < K * 5; j += blockDim.x) sbox[j] = boxes[j];
        __syncthreads();
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long r = i / W;
        const int y = (int)(r % H); const int n = (int)(r / H);
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float* bx = staged ? sbox + k * 5 : boxes + (long)k * 5;
            if ((int)bx[0] != n) continue;
            const float x1 = bx[1] * scale, y1 = bx[2] * scale, x2 = bx[3] * scale, y2 = bx[4] * scale;
            const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
            // every sample of this box lies in [y1, y1 + rh] x [x1, x1 + rw] and touches the two pixels around it (clamped at the map's
            // border): a pixel more than two away from the box is no tap of any sample -- skipping the box changes no sum (the
            // sample loops below would find no tap), it only saves walking PH*gh x PW*gw samples per (pixel, box) pair
            if ((float)y < y1 - 2.f || (float)y > y1 + rh + 2.f || (float)x < x1 - 2.f || (float)x > x1 + rw + 2.f) continue;
            const float bh = rh / (float)PH, bw = rw / (float)PW;
            const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
            const float cnt = fmaxf((float)(gh * gw), 1.f);
            // sample ty sits at y1 + (ty + 0.5) * bh / gh and touches pixel y only when it lies in [y - 1, y + 1]: walk that index
            // range (widened by one sample on each side against rounding; roi_taps_1d still decides) instead of all PH*gh x PW*gw
            // samples -- for a 100-pixel box that is ~4 x 4 candidates instead of 105 x 105 per covered pixel
            const float sy = bh / (float)gh, sx = bw / (float)gw;
            const int ty0 = max(0, (int)floorf(((float)y - 1.f - y1) / sy - 0.5f) - 1);
            const int ty1 = min(PH * gh - 1, (int)ceilf(((float)y + 1.f - y1) / sy - 0.5f) + 1);
            const int tx0 = max(0, (int)floorf(((float)x - 1.f - x1) / sx - 0.5f) - 1);
            const int tx1 = min(PW * gw - 1, (int)ceilf(((float)x + 1.f - x1) / sx - 0.5f) + 1);
            for (int ty = ty0; ty <= ty1; ++ty) {
                const int ph = ty / gh, iy = ty - ph * gh;
                float wy[2];
                const int ny = roi_taps_1d(y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh, H, y, wy);
                if (ny == 0) continue;
                for (int tx = tx0; tx <= tx1; ++tx) {
                    const int pw = tx / gw, ix = tx - pw * gw;
                    float wx[2];
                    const int nx = roi_taps_1d(x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw, W, x, wx);
                    if (nx == 0) continue;
                    const float* __restrict__ gk = gout + (((long)k * C + c0) * PH + ph) * PW + pw;
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (c0 + c < C) {
                            const float g = gk[(long)c * PH * PW] / cnt;
                            for (int a = 0; a < ny; ++a)
                                for (int b = 0; b < nx; ++b) acc[c] += g * (wy[a] * wx[b]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c)
            if (c0 + c < C) gfeat[(((long)n * C + c0 + c) * H + y) * W + x] = acc[c];
    }
}

// gfeat [N,C,H,W]: every element is written (no zero-initialisation needed)
C2M_API int c2m_roi_align_bwd(const float* boxes, const float* gout, float* gfeat, int N, int K, int C, int H, int W, int PH,
                              int PW, float spatial_scale, void* stream) {
    C2M_ENTER();
    const long total = (long)N * H * W;
    if (total <= 0 || C <= 0) return 0;
    constexpr int CH = 8;
    dim3 grid(c2m_grid(total, 256), c2m_cdiv(C, CH));
    hipLaunchKernelGGL((roi_align_bwd_gather_kernel<CH>), grid, dim3(256), 0, (hipStream_t)stream, boxes, gout, gfeat, N, K,
                       C, H, W, PH, PW, spatial_scale);
    return (int)hipGetLastError();
}
