// Shared store epilogue of the MFMA convolution kernels (conv_igemm.hip, conv_nc8.hip).
#pragma once
#include "common.h"
#include "dtype.h"

typedef float c2m_f32x16 __attribute__((ext_vector_type(16)));

// Fast epilogue of the MFMA conv kernels (single target, whole 32-row tiles inside M): every store is
// buffer_store_dword(value, voff[j], rsrc(Y), soff(i, r)) -- the pixel part of the address is one VGPR per pixel column block
// (0x80000000 = pixel outside the tensor: the hardware drops the store), the row part a scalar -- so an element costs the
// bias add + activation and NO address arithmetic.  The generic form (64-bit multiply-add, a bias load and an activation
// switch per element) was ~15 VALU per element: on the shallow layers (K = 288: 288 MFMAs per wave) the prologue and the
// epilogue together executed 5 of the 6.2 VALU per MFMA that PMC counts for the 32-row kernels.
// RAGGED = true (round 4, the bf16 NC8 kernels): the tile may hang over the last output row `mlim` -- rows past it get the
// out-of-range offset per lane (one compare + NI selects per (i, r)), their bias reads are predicated.
template <int MI, int NI, bool RAGGED = false>
__device__ __forceinline__ void c2m_store_tile_fast(const c2m_f32x16 (&acc)[MI][NI], float* __restrict__ ybase,
                                                    const unsigned (&voff)[NI], const int row0, const long row_stride,
                                                    const float* __restrict__ bias, const bool direct, const int act,
                                                    const float slope, const int lane, const bool yh = false,
                                                    const int mlim = 0x7fffffff) {
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(ybase, 0, 0x80000000u, 0x00020000);
    const int rs4 = (int)row_stride * (yh ? 2 : 4);      // voff[] is in bytes of the output element type as well
    float bv[MI][16];
    if (RAGGED && direct && bias) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                bv[i][r] = row < mlim ? bias[row] : 0.f;
            }
    } else if (direct && bias) {
        const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, 0x80000000u, 0x00020000);
        const unsigned hb = (unsigned)(16 * (lane >> 5));
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bv[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    br, hb, (row0 + i * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[i][r] = 0.f;
    }
    const int a = direct ? act : C2M_ACT_NONE;
#define C2M_STORE_LOOP(EXPR)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                                     \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
            const int rowb = row0 + i * 32 + (r & 3) + 8 * (r >> 2);                                                   \
            if (RAGGED && rowb >= mlim) continue;                                                                      \
            const int soff = rowb * rs4;                                                                               \
            const bool rok = !RAGGED || rowb + 4 * (lane >> 5) < mlim;                                                 \
            _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                           \
                float v = acc[i][j][r] + bv[i][r];                                                                     \
                v = (EXPR);                                                                                            \
                const unsigned vo = rok ? voff[j] : 0x80000000u;                                                       \
                if (yh) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)v), yr, vo, soff, 0); \
                else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, vo, soff, 0);          \
            }                                                                                                          \
        }
    if (a == C2M_ACT_NONE)       { C2M_STORE_LOOP(v) }
    else if (a == C2M_ACT_RELU)  { C2M_STORE_LOOP(v > 0.f ? v : 0.f) }
    else if (a == C2M_ACT_LRELU) { C2M_STORE_LOOP(v > 0.f ? v : v * slope) }
    else                         { C2M_STORE_LOOP(c2m_act(v, a, slope)) }
#undef C2M_STORE_LOOP
}

