// data_prep.hip -- the input-pipeline stage just upstream of the hot path, on the device.
// Reference (CPU, per sample, PIL arrays -> tensors): src/datasets/cityscapes.py
//   :30-33,59-61  image frames     ToTensor(): uint8 HWC -> float CHW / 255, frames stacked along dim 1  -> video [B,3,T,H,W]
//   :35-41,62-70  label-id maps    ToTensor()*255 == i for i in 0..10 (bg) / 11..19 (fg)                -> bg/fg one-hot masks
//   :212-216,255,262-265 occlusion PNGs  ToTensor()/255 -> clip_mask(> 0.5)                             -> target_bw_occ
//   :219-231,254,261     .flo flows      HWC -> CHW (no rescale at the native size), stacked over time  -> target_bw_of
// The decoded, resized arrays (PIL's job) are uploaded once as uint8 / float; everything after that runs here, so the host
// never builds the 20-channel one-hot volumes (20x the bytes of the label map).
// (v/255)*255 == v holds exactly in fp32 for every uint8 v, so the one-hot test is an integer compare.
#include "common.h"

// frames [B][T][H][W][3] uint8 -> video [B][3][T][H][W] float, x / 255 (true division, as Tensor.div(255))
__global__ void prep_video_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long BT, int T, long HW) {
    const long total = BT * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bt = i / HW, p = i - bt * HW;
        const long b = bt / T, t = bt - b * T;
        const uint8_t* __restrict__ s = src + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[((b * 3 + c) * T + t) * HW + p] = (float)s[c] / 255.0f;
    }
}

// labels [B][T][H][W] uint8 -> bg [B][11][T][H][W], fg [B][9][T][H][W] one-hot floats
__global__ void prep_seg_onehot_kernel(const uint8_t* __restrict__ lab, float* __restrict__ bg, float* __restrict__ fg,
                                       long BT, int T, long HW) {
    const long total = BT * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bt = i / HW, p = i - bt * HW;
        const long b = bt / T, t = bt - b * T;
        const int v = lab[i];
#pragma unroll
        for (int c = 0; c < 11; ++c) bg[((b * 11 + c) * T + t) * HW + p] = v == c ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 9; ++c) fg[((b * 9 + c) * T + t) * HW + p] = v == 11 + c ? 1.0f : 0.0f;
    }
}

// occ [B][T][H][W] uint8 -> [B][1][T][H][W] float: (v / 255 > 0.5) ? 1 : 0;  flow [B][T][H][W][2] -> [B][2][T][H][W]
__global__ void prep_flow_occ_kernel(const uint8_t* __restrict__ occ, const float* __restrict__ flo,
                                     float* __restrict__ occ_out, float* __restrict__ flow_out, long BT, int T, long HW) {
    const long total = BT * HW;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long bt = i / HW, p = i - bt * HW;
        const long b = bt / T, t = bt - b * T;
        if (occ) occ_out[i] = ((float)occ[i] / 255.0f > 0.5f) ? 1.0f : 0.0f;
        if (flo) {
            flow_out[((b * 2 + 0) * T + t) * HW + p] = flo[i * 2 + 0];
            flow_out[((b * 2 + 1) * T + t) * HW + p] = flo[i * 2 + 1];
        }
    }
}

C2M_API int c2m_prep_video(const uint8_t* frames, float* video, int B, int T, int H, int W, void* stream) {
    C2M_ENTER();
    const long total = (long)B * T * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(prep_video_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, frames, video,
                       (long)B * T, T, (long)H * W);
    return (int)hipGetLastError();
}

C2M_API int c2m_prep_seg_onehot(const uint8_t* labels, float* bg, float* fg, int B, int T, int H, int W, void* stream) {
    C2M_ENTER();
    const long total = (long)B * T * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(prep_seg_onehot_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, labels, bg,
                       fg, (long)B * T, T, (long)H * W);
    return (int)hipGetLastError();
}

C2M_API int c2m_prep_flow_occ(const uint8_t* occ, const float* flow_hwc, float* occ_out, float* flow_out, int B, int T,
                              int H, int W, void* stream) {
    C2M_ENTER();
    const long total = (long)B * T * H * W;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(prep_flow_occ_kernel, dim3(c2m_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, occ, flow_hwc,
                       occ_out, flow_out, (long)B * T, T, (long)H * W);
    return (int)hipGetLastError();
}
