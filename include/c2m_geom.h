/* c2m_geom.h -- the int64 geometry blocks of the convolution entry points, by NAME.
 *
 * Rounds 1-4 addressed geom[] by bare numbers on both sides of the C ABI; one collision (a per-class offset written to
 * geom[72 + 3c] ran into the element-type flags at geom[90..92] for class 6) produced silently wrong gradients.  This header is
 * the ONE definition: the kernels' launchers (c2m_amd/csrc, through common.h) and the ctypes host (c2m_amd/_lib.py parses the two
 * enums below and checks C2M_ABI_VERSION / the block lengths against the loaded library at import) both use these names, and the
 * ranges are checked against each other at compile time.  Meaning of every entry: include/c2m_hip.h, at the entry point. */
#pragma once

#define C2M_ABI_VERSION 6          /* bump when an entry is added, moved or changes meaning */

/* geom[] of c2m_conv_igemm / c2m_conv_wgrad / c2m_conv_patch_nc8 */
enum c2m_geom_index {
    C2M_G_M = 0, C2M_G_NK = 1, C2M_G_LDA = 2, C2M_G_NPIX = 3,
    C2M_G_TO = 4, C2M_G_HO = 5, C2M_G_WO = 6, C2M_G_TI = 7, C2M_G_HI = 8, C2M_G_WI = 9,
    C2M_G_ST = 10, C2M_G_SH = 11, C2M_G_SW = 12,
    C2M_G_IN_SN = 13, C2M_G_IN_ST = 14, C2M_G_IN_SH = 15,
    C2M_G_OUT_SN = 16, C2M_G_OUT_SC = 17, C2M_G_OUT_ST = 18, C2M_G_OUT_SH = 19, C2M_G_OUT_SW = 20, C2M_G_OUT_OFF = 21,
    C2M_G_REFLECT = 22, C2M_G_IS3D = 23, C2M_G_NS = 24, C2M_G_IN_SC = 25, C2M_G_SPLITS = 26, C2M_G_SLAB_STRIDE = 27,
    C2M_G_CIN = 28, C2M_G_TAPS = 29, C2M_G_NTG = 30, C2M_G_NGROUPS = 31,
    C2M_G_X_BYTES = 32, C2M_G_DY_BYTES = 33,
    C2M_G_PRECISION = 34,            /* 0 fp32 operands, 1 bf16 operands */
    C2M_G_SQUARE_KW = 35,            /* +-KW: row-major square tap set of a 2-D stride-1 layer (thin row-blocked kernels) */
    C2M_G_PS_T = 36, C2M_G_PS_Y = 37, C2M_G_PS_X = 38,        /* two-target epilogue: padded coordinate = o * ps + po */
    C2M_G_PO_T = 39, C2M_G_PO_Y = 40, C2M_G_PO_X = 41,
    C2M_G_LO_T = 42, C2M_G_LO_Y = 43, C2M_G_LO_X = 44,
    C2M_G_EXT_T = 45, C2M_G_EXT_Y = 46, C2M_G_EXT_X = 47,
    C2M_G_Y2_SN = 48, C2M_G_Y2_SC = 49, C2M_G_Y2_ST = 50, C2M_G_Y2_SH = 51,
    C2M_G_PATCH = 52, C2M_G_PATCH_IY0 = 53, C2M_G_PATCH_IX0 = 54,
    C2M_G_PATCH_TY = 55,             /* 55..57: patch row of tap row 0, 1, 2 */
    C2M_G_PATCH_TX = 58,             /* 58..60: patch column of tap column 0, 1, 2 */
    C2M_G_NCLS = 61, C2M_G_A_CLS = 62, C2M_G_KTAB_CLS = 63,
    C2M_G_CLS_OUT_OFF = 64,          /* 64..71: out_off of class c (<= 8 stride parity classes) */
    C2M_G_X_TYPE = 90, C2M_G_Y_TYPE = 91,                     /* element types in memory: 0 fp32, 1 bf16 */
    C2M_G_WGRAD_WIDE = 92,           /* bf16 weight gradient: 0 gather form, 1 16-byte-load form, 2 its stride-2 form */
    C2M_G_NC8_VARIANT = 93, C2M_G_G8 = 94, C2M_G_G8_VARIANT = 95,
    C2M_G_CLS_PO = 96,               /* 96..119: (po_t, po_y, po_x) of class c */
    C2M_G_LEN = 120
};
#define C2M_G_MAX_CLS 8

/* geom[] of c2m_conv_wino / c2m_conv_wino4 */
enum c2m_wino_geom_index {
    C2M_WG_M = 0, C2M_WG_K = 1, C2M_WG_NIMG = 2, C2M_WG_HI = 3, C2M_WG_WI = 4, C2M_WG_HO = 5, C2M_WG_WO = 6,
    C2M_WG_IY0 = 7, C2M_WG_IX0 = 8, C2M_WG_REFLECT = 9,
    C2M_WG_IN_SN = 10, C2M_WG_IN_SC = 11, C2M_WG_IN_SH = 12,
    C2M_WG_OUT_SN = 13, C2M_WG_OUT_SC = 14, C2M_WG_OUT_SH = 15, C2M_WG_OUT_OFF = 16, C2M_WG_X_BYTES = 17,
    C2M_WG_Y2_SN = 18, C2M_WG_Y2_SC = 19, C2M_WG_Y2_SH = 20, C2M_WG_LO_Y = 21, C2M_WG_LO_X = 22, C2M_WG_EXT_Y = 23, C2M_WG_EXT_X = 24,
    C2M_WG_TO = 25, C2M_WG_IN_ST = 26, C2M_WG_OUT_ST = 27, C2M_WG_CIN = 28, C2M_WG_NKT = 29, C2M_WG_TOFF = 30, C2M_WG_TI = 31,
    C2M_WG_TREFLECT = 32,
    C2M_WG_PTAB = 33,                /* device pointer of the temporal pair table, or 0 */
    C2M_WG_RING = 34,                /* device pointer of the pad-ring terms R [N][M][4][RING_L] (c2m_reflect_ring_buffer), or 0: added */
    C2M_WG_RING_L = 35,              /* to rows 1 / Ho-2 and columns 1 / Wo-2 of the output by the epilogue (2-D, single target only)  */
    C2M_WG_LEN = 36
};

#ifdef __cplusplus
static_assert(C2M_G_PATCH_TY + 3 == C2M_G_PATCH_TX && C2M_G_PATCH_TX + 3 == C2M_G_NCLS, "patch tap rows / columns");
static_assert(C2M_G_CLS_OUT_OFF + C2M_G_MAX_CLS <= C2M_G_X_TYPE, "per-class output offsets run into the element-type flags");
static_assert(C2M_G_G8_VARIANT < C2M_G_CLS_PO && C2M_G_CLS_PO + 3 * C2M_G_MAX_CLS == C2M_G_LEN, "per-class pad offsets");
static_assert(C2M_WG_RING_L + 1 == C2M_WG_LEN, "wino geom length");
#endif
