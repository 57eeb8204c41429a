// Sparse-motion rasteriser and forward-splat occlusion map: the index/mask path (bit-exact vs the reference).
//
//   c2m_sparse_raster     motion_estimator/dense_motion.py:94-168 generate_sparse_motion + warp: the reference loops
//                         in Python over objects x frames (affine_grid + grid_sample + 3 torch.where per pair, ~8
//                         launches each); here ONE launch covers every (sample, frame, pixel) and walks the objects
//                         in order (later objects overwrite earlier ones).
//   c2m_occlusion_splat   utils/ops.py:205-275 get_occlusion_map/get_corresponding_map (+ clip_mask,
//                         dense_motion.py:155-160).  The reference sums with a sequential scatter_add_; float sums
//                         are order-dependent and the result is thresholded at 0.5, so we reproduce the exact order:
//                         count -> exclusive scan -> bucket fill -> per-target sort by (class, source) -> fp32 sum.
//                         No float atomics anywhere.
//
// Compiled with -ffp-contract=off; the fp32 operation order is the one pinned by oracle/c2m_oracle_index.c
// (FMA mode 7: fused linspace, fused unnormalize, fused bilinear chain, fused 3-term affine dot).
#include "common.h"

__device__ __forceinline__ float lin_m1_1(int i, int steps) {
    if (steps <= 1) return -1.0f;
    const float step = 2.0f / (float)(steps - 1);
    const int half = steps / 2;
    return i < half ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(steps - 1 - i), 1.0f);
}

// instance [B,H,W] float ids; obj_id/obj_batch [K] int32; thetas [K,T,6]
// bw, fw [B,2,T,H,W]; bin [B,1,T,H,W]  (all fully written)
__global__ void sparse_raster_kernel(const float* __restrict__ instance, const int* __restrict__ obj_id,
                                     const int* __restrict__ obj_batch, const float* __restrict__ thetas,
                                     float* __restrict__ bw, float* __restrict__ fw, float* __restrict__ bin, int B,
                                     int K, int T, int H, int W) {
    const long HW = (long)H * W;
    const long total = (long)B * T * HW;
    const float cx = (float)(((double)W - 1.0) / 2.0), cy = (float)(((double)H - 1.0) / 2.0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int t = (int)(r % T); const int b = (int)(r / T);
        const float* __restrict__ inst = instance + (long)b * HW;
        const float here = inst[(long)y * W + x];
        const float lx = lin_m1_1(x, W), ly = lin_m1_1(y, H);
        float bx = lx * (float)(W - 1) / (float)W;
        float by = ly * (float)(H - 1) / (float)H;
        if (W <= 1) bx = 0.0f;
        if (H <= 1) by = 0.0f;
        float bwx = 0.0f, bwy = 0.0f, fwx = 0.0f, fwy = 0.0f, bn = 0.0f;
        for (int k = 0; k < K; ++k) {
            const int id = obj_id[k];
            if (id == 0 || obj_batch[k] != b) continue;
            const float fid = (float)id;
            const float* __restrict__ th = thetas + ((long)k * T + t) * 6;
            const float gx = fmaf(1.0f, th[2], fmaf(by, th[1], bx * th[0]));
            const float gy = fmaf(1.0f, th[5], fmaf(by, th[4], bx * th[3]));
            const float flx = (gx - lx) * cx, fly = (gy - ly) * cy;
            const float ix = fmaf(gx + 1.0f, (float)W / 2.0f, -0.5f);
            const float iy = fmaf(gy + 1.0f, (float)H / 2.0f, -0.5f);
            const float xw = floorf(ix), yn = floorf(iy);
            const float w_ = ix - xw, e_ = 1.0f - w_, n_ = iy - yn, s_ = 1.0f - n_;
            const float nw = s_ * e_, ne = s_ * w_, sw = n_ * e_, se = n_ * w_;
            // float -> int conversion saturates on the GPU; coordinates far outside simply fail the range test
            const int x0 = (int)xw, y0 = (int)yn, x1 = x0 + 1, y1 = y0 + 1;
            const bool ox0 = x0 >= 0 && x0 < W, ox1 = x1 >= 0 && x1 < W, oy0 = y0 >= 0 && y0 < H, oy1 = y1 >= 0 && y1 < H;
            const float vnw = (ox0 && oy0 && inst[(long)y0 * W + x0] == fid) ? 1.0f : 0.0f;
            const float vne = (ox1 && oy0 && inst[(long)y0 * W + x1] == fid) ? 1.0f : 0.0f;
            const float vsw = (ox0 && oy1 && inst[(long)y1 * W + x0] == fid) ? 1.0f : 0.0f;
            const float vse = (ox1 && oy1 && inst[(long)y1 * W + x1] == fid) ? 1.0f : 0.0f;
            float warped = vnw * nw;
            warped = fmaf(vne, ne, warped);
            warped = fmaf(vsw, sw, warped);
            warped = fmaf(vse, se, warped);
            if (warped == 1.0f) { bwx = flx; bwy = fly; bn = 1.0f; }
            if (here == fid) { fwx = flx * -1.0f; fwy = fly * -1.0f; }
        }
        const long sp = (long)y * W + x;
        bw[(((long)b * 2 + 0) * T + t) * HW + sp] = bwx;
        bw[(((long)b * 2 + 1) * T + t) * HW + sp] = bwy;
        fw[(((long)b * 2 + 0) * T + t) * HW + sp] = fwx;
        fw[(((long)b * 2 + 1) * T + t) * HW + sp] = fwy;
        bin[((long)b * T + t) * HW + sp] = bn;
    }
}

C2M_API int c2m_sparse_raster(const float* instance, const int* obj_id, const int* obj_batch, const float* thetas,
                              float* bw, float* fw, float* bin, int B, int K, int T, int H, int W, void* stream) {
    C2M_ENTER();
    const long total = (long)B * T * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(sparse_raster_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, instance,
                       obj_id, obj_batch, thetas, bw, fw, bin, B, K, T, H, W);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------- forward splat
// One "image" = one [2,H,W] flow field with explicit strides so that [B,2,T,H,W] tensors are consumed in place:
// image index q = b*T + t;  flow_x at flow + b*sb + t*st, flow_y at + sc.
struct SplatP {
    const float* flow;
    long sb, sc, st;
    int T, H, W;
    long nimg;
};

struct Contrib { int target; float v; };

// class order = the reference's torch.cat order: 0 (x_ceil,y_ceil) 1 (x_ceil,y_floor) 2 (x_floor,y_ceil) 3 (x_floor,y_floor)
__device__ __forceinline__ Contrib splat_contrib(const SplatP& p, long img, int src, int cls) {
    const int W = p.W, H = p.H;
    const int x = src % W, y = src / W;
    const long b = img / p.T, t = img % p.T;
    const float* __restrict__ f = p.flow + b * p.sb + t * p.st;
    const float px = (float)x + f[(long)y * W + x];
    const float py = (float)y + f[p.sc + (long)y * W + x];
    const float xf0 = floorf(px), yf0 = floorf(py), xc0 = xf0 + 1.0f, yc0 = yf0 + 1.0f;
    const float xf = fminf(fmaxf(xf0, 0.0f), (float)(W - 1)), yf = fminf(fmaxf(yf0, 0.0f), (float)(H - 1));
    const float xc = fminf(fmaxf(xc0, 0.0f), (float)(W - 1)), yc = fminf(fmaxf(yc0, 0.0f), (float)(H - 1));
    const bool use_xc = cls < 2, use_yc = (cls & 1) == 0;
    const float tx = use_xc ? xc : xf, ty = use_yc ? yc : yf;
    const bool bad = (use_xc ? xc0 != xc : xf0 != xf) || (use_yc ? yc0 != yc : yf0 != yf);
    float v = (1.0f - fabsf(px - tx)) * (1.0f - fabsf(py - ty));
    if (bad || !(v == v)) v = 0.0f;   // NaN flows contribute nothing (reference would poison the map)
    Contrib c;
    c.target = (int)(tx + ty * (float)W);
    c.v = v;
    return c;
}

// adding +0.0 never changes an fp32 sum that starts at +0.0 with non-negative terms, so zero contributions are dropped
__global__ void splat_count_kernel(const SplatP p, int* __restrict__ count) {
    const long HW = (long)p.H * p.W;
    const long total = p.nimg * HW * 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int src = (int)(i % HW); const long r = i / HW;
        const int cls = (int)(r % 4); const long img = r / 4;
        const Contrib c = splat_contrib(p, img, src, cls);
        if (c.v != 0.0f) atomicAdd(count + img * HW + c.target, 1);
    }
}

// exclusive scan of count within each image (one 1024-thread block per image); also copies offsets into cursor.
// Chunks of 4096 counts: a thread loads FOUR consecutive counts with one 16-byte load (coalesced over the block), scans them,
// the block scans its 1024 partial sums (wave scan with DPP-free shuffles, 16 wave totals through LDS) and a running carry links
// the chunks.  The first form gave every thread one contiguous HW/1024 segment: two strided 4-byte passes per thread, 290 us per
// launch at 256x512 on 20 of 256 CUs.  Integer arithmetic: the result is the same scan.
__global__ __launch_bounds__(1024) void splat_scan_kernel(const int* __restrict__ count, int* __restrict__ offset,
                                                          int* __restrict__ cursor, int HW) {
    __shared__ int wsum[16];
    const long base = (long)blockIdx.x * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)(count + base)) | ((uintptr_t)(offset + base)) | ((uintptr_t)(cursor + base))) & 15) == 0;
    int carry = 0;
    for (int c0 = 0; c0 < HW; c0 += 4096) {
        const int i0 = c0 + (int)threadIdx.x * 4;
        int v[4] = {0, 0, 0, 0};
        if (vec && i0 + 3 < HW) {
            const int4 q = *reinterpret_cast<const int4*>(count + base + i0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (i0 + e < HW) v[e] = count[base + i0 + e];
        }
        const int s = v[0] + v[1] + v[2] + v[3];
        int inc = s;                                   // inclusive scan of s over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int wpre = 0;                                  // sum of the waves in front
#pragma unroll
        for (int w = 0; w < 16; ++w) wpre += w < wave ? wsum[w] : 0;
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += wsum[w];
        int run = carry + wpre + inc - s;              // exclusive prefix of this thread's four counts
        int o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o4[e] = run; run += v[e]; }
        if (vec && i0 + 3 < HW) {
            const int4 q = make_int4(o4[0], o4[1], o4[2], o4[3]);
            *reinterpret_cast<int4*>(offset + base + i0) = q;
            *reinterpret_cast<int4*>(cursor + base + i0) = q;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (i0 + e < HW) { offset[base + i0 + e] = o4[e]; cursor[base + i0 + e] = o4[e]; }
        }
        carry += tot;
        __syncthreads();                               // wsum is rewritten by the next chunk
    }
}

__global__ void splat_fill_kernel(const SplatP p, int* __restrict__ cursor, int* __restrict__ keys,
                                  float* __restrict__ vals) {
    const long HW = (long)p.H * p.W;
    const long total = p.nimg * HW * 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int src = (int)(i % HW); const long r = i / HW;
        const int cls = (int)(r % 4); const long img = r / 4;
        const Contrib c = splat_contrib(p, img, src, cls);
        if (c.v != 0.0f) {
            const int slot = atomicAdd(cursor + img * HW + c.target, 1);
            const long o = img * HW * 4 + slot;
            keys[o] = cls * (int)HW + src;     // reference summation order = ascending key
            vals[o] = c.v;
        }
    }
}

// per target pixel: order its contributions by key, sum sequentially in fp32, clamp to [0,1] (and threshold)
__global__ void splat_reduce_kernel(const int* __restrict__ count, const int* __restrict__ offset,
                                    const int* __restrict__ keys, const float* __restrict__ vals,
                                    float* __restrict__ occ, float* __restrict__ clip, long nimg, int HW) {
    const long total = nimg * HW;
    constexpr int FAST = 24;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long img = i / HW;
        const int n = count[i];
        const long o = img * (long)HW * 4 + offset[i];
        float acc = 0.0f;
        if (n <= 4) {
            // the common case (a bilinear splat puts ~4 contributions on a pixel): four predicated loads, a 5-comparator sorting
            // network on registers (empty slots carry the largest key and the value 0: adding 0.0f last changes nothing), the sum
            // in ascending key order -- the run-time-indexed arrays of the general path live in scratch memory (196 us per launch
            // at 256x512)
            int k0 = 0x7fffffff, k1 = 0x7fffffff, k2 = 0x7fffffff, k3 = 0x7fffffff;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (n > 0) { k0 = keys[o]; v0 = vals[o]; }
            if (n > 1) { k1 = keys[o + 1]; v1 = vals[o + 1]; }
            if (n > 2) { k2 = keys[o + 2]; v2 = vals[o + 2]; }
            if (n > 3) { k3 = keys[o + 3]; v3 = vals[o + 3]; }
#define C2M_CSWAP(ka, va, kb, vb) { const bool sw = ka > kb; const int tk = sw ? kb : ka; const float tv = sw ? vb : va; \
                                    kb = sw ? ka : kb; vb = sw ? va : vb; ka = tk; va = tv; }
            C2M_CSWAP(k0, v0, k1, v1) C2M_CSWAP(k2, v2, k3, v3) C2M_CSWAP(k0, v0, k2, v2) C2M_CSWAP(k1, v1, k3, v3) C2M_CSWAP(k1, v1, k2, v2)
#undef C2M_CSWAP
            acc = ((v0 + v1) + v2) + v3;
            if (n < 4) acc = n == 0 ? 0.0f : (n == 1 ? v0 : (n == 2 ? v0 + v1 : (v0 + v1) + v2));
        } else if (n <= FAST) {
            int k[FAST]; float v[FAST];
            for (int a = 0; a < n; ++a) {       // insertion sort while loading
                const int ka = keys[o + a]; const float va = vals[o + a];
                int j = a;
                while (j > 0 && k[j - 1] > ka) { k[j] = k[j - 1]; v[j] = v[j - 1]; --j; }
                k[j] = ka; v[j] = va;
            }
            for (int a = 0; a < n; ++a) acc += v[a];
        } else {
            // long list (strongly converging flow): selection by increasing key, O(n^2) but bounded and exact
            int last = -1;
            for (int a = 0; a < n; ++a) {
                int best = 0x7fffffff; float bv = 0.0f;
                for (int j = 0; j < n; ++j) {
                    const int kj = keys[o + j];
                    if (kj > last && kj < best) { best = kj; bv = vals[o + j]; }
                }
                acc += bv;
                last = best;
            }
        }
        acc = fminf(fmaxf(acc, 0.0f), 1.0f);
        if (occ) occ[i] = acc;
        if (clip) clip[i] = acc > 0.5f ? 1.0f : 0.0f;
    }
}

// workspace: count, offset, cursor: nimg*HW ints each; keys: nimg*HW*4 ints; vals: nimg*HW*4 floats
C2M_API long c2m_occlusion_splat_workspace_bytes(long nimg, int H, int W) {
    const long HW = (long)H * W;
    return nimg * HW * 4 * 3 + nimg * HW * 4 * 4 * 2;
}

C2M_API int c2m_occlusion_splat(const float* flow, long sb, long sc, long st, int B, int T, int H, int W, float* occ,
                                float* clip, void* workspace, void* stream) {
    C2M_ENTER();
    const long nimg = (long)B * T, HW = (long)H * W;
    if (nimg * HW <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int* count = (int*)workspace;
    int* offset = count + nimg * HW;
    int* cursor = offset + nimg * HW;
    int* keys = cursor + nimg * HW;
    float* vals = (float*)(keys + nimg * HW * 4);
    hipError_t e = c2m_zero_async(count, sizeof(int) * nimg * HW, s);
    if (e != hipSuccess) return (int)e;
    SplatP p{flow, sb, sc, st, T, H, W, nimg};
    const long work = nimg * HW * 4;
    hipLaunchKernelGGL(splat_count_kernel, dim3(c2m_grid(work, 256)), dim3(256), 0, s, p, count);
    hipLaunchKernelGGL(splat_scan_kernel, dim3((unsigned)nimg), dim3(1024), 0, s, count, offset, cursor, (int)HW);
    hipLaunchKernelGGL(splat_fill_kernel, dim3(c2m_grid(work, 256)), dim3(256), 0, s, p, cursor, keys, vals);
    hipLaunchKernelGGL(splat_reduce_kernel, dim3(c2m_grid(nimg * HW, 256)), dim3(256), 0, s, count, offset, keys, vals,
                       occ, clip, nimg, (int)HW);
    return (int)hipGetLastError();
}
