/*
 * tinygemm_hip.h -- C ABI of the MI355X (gfx950) tinygemm W4A16 small-batch GEMM library.
 *
 * This is the drop-in boundary for the reference's native extension
 * (facebookresearch/any4, tinygemm_lib/ *.cu sources): every entry point below replaces one host
 * function the reference registers with torch (tinygemm_lib/TinyGemm.cpp:17-200,
 * prototypes tinygemm_lib/TinyGemm.h:23-216).  The signatures carry plain device
 * pointers and sizes only: no torch types, no allocation inside the library (the caller
 * owns every buffer, outputs included), no mutable global state (the only statics are
 * write-once per-kernel launch attributes), no environment variables, no host synchronisation.
 * All functions launch asynchronously on `stream` and are safe to call from several host
 * threads on distinct streams, and under hipGraph stream capture.
 *
 * Return value: 0 on success; a negative TG_E_* code when a precondition the reference
 * checks with TORCH_CHECK fails (nothing is launched); a positive value is the hipError_t
 * reported by the launch.  Never throws.
 *
 * Layout vocabulary (identical to the reference, TinyGemm.h:19-103):
 *   RM       row-major 16-bit matrix [rows][k]
 *   A16      m16n8k16 "A" fragment order  [ceil(m/16)][ceil(k/16)][32][8]        16-bit
 *   B16      m16n8k16 "B" fragment order  [ceil(n/8)][ceil(k/(16 I))][32][4 I]   16-bit, I in {1,2}
 *   Aint4    packed 4-bit, weights on the left   [ceil(m/16)][ceil(k/(16 I))][32][I]   int32, I in {1,2,4}
 *   Bint4    packed 4-bit, weights on the right  [ceil(n/8)][k/(16 I)][32][I/2]       int32, I in {2,4,8}
 * tg_convert_to_{A,B}int4 produce words bit-identical to the reference's (TinyGemmConvertA.cu:226-285,
 * TinyGemmConvertB.cu:252-308) and tg_gemm_w4 consumes them: a weight packed by the CUDA implementation runs here as it
 * is, on either operand side.  For weights on the LEFT this library additionally accepts -- and its Python convert op
 * produces by default -- a second packed format, TG_WFMT_ROWS (tg_w4_gemm.w_format): the Bint4 tensor of the weight rows
 * padded to 16.  The CUDA implementation cannot read that one; at this boundary the caller names the format in the
 * argument struct, and the Python layer reads it off the tensor's shape (the two formats never share a shape for one k:
 * any4_amd/ops.py aside_format), so a checkpoint of either format is multiplied correctly or rejected, never misread.
 * tg_unpack_int4 + a packer converts between the two losslessly.
 */
#ifndef TINYGEMM_HIP_H_
#define TINYGEMM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 8

#if defined(__GNUC__)
#define TG_API __attribute__((visibility("default")))
#else
#define TG_API
#endif

typedef void* tg_stream_t; /* a hipStream_t; NULL = the null stream */

/* 16-bit float type of activations / scales / LUT / outputs */
enum { TG_BF16 = 0, TG_F16 = 1 };

/* 4-bit quantisation variant; mirrors Int4_QType (tinygemm_lib/TinyGemmUtils.cuh:21-32) */
enum {
  TG_Q_INT4 = 0,         /* uniform int4: value = code - 8                                  */
  TG_Q_ANY4_GLOBAL = 1,  /* one 16-entry LUT for the whole matrix (the reference's NF4 path) */
  TG_Q_ANY4_ROWWISE = 2, /* one 16-entry LUT per weight row (any4)                         */
  TG_Q_MX4 = 3,          /* fp4-e2m1 codes with an e8m0 exponent per group                 */
  TG_Q_INT8 = 4          /* tg_gemm_w8 only: uniform int8, value = byte - 128              */
};

/* GEMM numerics (tg_w4_gemm.numerics).  Both contract bf16/fp16 products in f32 and round the result once.
 *   TG_NUM_FAST       where a kernel for it exists (Bint4 weights, activation block small enough to stage on chip) the
 *                     per-group affine map is applied to the f32 accumulator of the group instead of to every weight:
 *                       y = sum_g ( scale_g * sum_{k in g} x_k * lut[code_k]  +  zero_g * sum_{k in g} x_k ).
 *                     No per-weight rounding to 16 bits, so y differs from the reference's by at most the reference's own
 *                     rounding of its dequantised weights (<= 2^-9 * sum_k |x_k w_k|, about one output ulp at k = 4096);
 *                     mx4 weights are exact either way (and its kernels the same in both settings).  More than 16 activation
 *                     rows, up to 64 (row-major operands, no fused norm / SwiGLU), are issued as ceil(m / 16) launches of up to 16 rows
 *                     on the caller's stream (the reference's grid walks m's 16-row tiles the same way, TinyGemmImpl.cuh:379-392).
 *                     Other shapes run the TG_NUM_REFERENCE kernels.
 *   TG_NUM_REFERENCE  w = RNE16(fma(lut[code], scale, zero)) per element exactly as the reference kernels compute it
 *                     (MatrixLayoutB.cuh:1042-1046, MatrixLayoutA.cuh:747-754): bit-identical weights, e.g. the identity
 *                     known-answer tests of the reference come out bit-equal.                                           */
enum { TG_NUM_FAST = 0, TG_NUM_REFERENCE = 1, TG_NUM_FAST_MFMA = 2, TG_NUM_FAST_DOT2 = 3 };
/*   TG_NUM_FAST_MFMA  TG_NUM_FAST with the m = 1 contraction on the matrix cores: since round 3 the default contracts ONE activation
 *                     row with per-lane v_dot2_f32_bf16 (a 32x32x16 MFMA spends 16384 multiplier slots on 512 useful products and,
 *                     under the power cap, clock: 76 -> 81 % of the HBM roofline).  This value keeps the matrix-core contraction
 *                     reachable for stacked m = 1 launches so that both can be timed and compared (bench.py `m1_mfma`,
 *                     tests/test_gpu_fast.py): on w4_gemm_xr_kernel's 16x16x32 MFMAs where that kernel applies (k = 4096, rows a
 *                     multiple of 64, >= 512 work items: 79 %), else on the 32x32x16 ones of w4_gemm_pair_kernel (76 %);
 *                     everything else runs as TG_NUM_FAST.
 *   TG_NUM_FAST_DOT2  TG_NUM_FAST with the m = 1 contraction of stacked launches pinned to the per-lane v_dot2_f32_bf16 (the other arm of
 *                     the same A/B).  Which of the two TG_NUM_FAST itself takes is a build-time default, reported by
 *                     tg_m1_default_contraction() and timed both ways, alternating, by bench.py (`m1_contraction_ab`). */

/* precondition failures (wording of the matching TORCH_CHECK is in tg_error_string) */
enum {
  TG_E_NULL = -1,      /* a required pointer is NULL                                       */
  TG_E_INNER_K = -2,   /* innerKTiles not valid for this layout                            */
  TG_E_K_DIV = -3,     /* k not divisible as the layout / kernel requires                  */
  TG_E_GROUP = -4,     /* qGroupSize not in {32,64,128,256} or does not divide k           */
  TG_E_DTYPE = -5,     /* dtype not bf16/fp16, or mx4 with fp16                            */
  TG_E_QTYPE = -6,     /* unknown quantisation variant                                     */
  TG_E_SHAPE = -7,     /* negative / zero / inconsistent sizes                             */
  TG_E_ALIGN = -8,     /* a buffer is not 16-byte aligned                                  */
  TG_E_DEVICE = -9,    /* hipSetDevice failed / no such device                             */
  TG_E_SIZE = -10,     /* one problem's operand exceeds the kernels' 32-bit byte offsets   */
  TG_E_INTERNAL = -11, /* build inconsistency (should not happen)                          */
  TG_E_LAYOUT = -12,   /* x_layout / y_layout not available for this problem               */
  TG_E_FUSION = -13,   /* norm_weight / epilogue: no kernel with that fused stage for this problem (issue it as its own launch) */
  TG_E_STRUCT = -14    /* tg_w4_gemm.struct_bytes: smaller than the ABI-1 prefix of the struct, or larger than this library's struct */
};

TG_API int tg_abi_version(void);
/* ABI 7: which contraction a STACKED one-row launch takes under TG_NUM_FAST: 0 = per-lane v_dot2_f32_bf16, 1 = the matrix core
 * (v_mfma_f32_16x16x32 / 32x32x16_bf16).  TG_NUM_FAST_DOT2 / TG_NUM_FAST_MFMA pin either one. */
TG_API int tg_m1_default_contraction(void);
TG_API const char* tg_error_string(int code);

/* ---- layout conversion ------------------------------------------------------------------ */

/* replaces convert_matrix_to_m16n8k16_Bint4_layout (TinyGemmConvertB.cu:312-364).
 * in  : int32 [n][k], codes 0..15            out : int32 [ceil(n/8)][k/(16 I)][32][I/2]
 * requires I in {2,4,8}, k % (16 I) == 0 */
TG_API int tg_convert_to_Bint4(const int32_t* in, int64_t n, int64_t k, int inner_k_tiles, int32_t* out,
                        int device, tg_stream_t stream);

/* replaces convert_matrix_to_m16n8k16_Aint4_layout (TinyGemmConvertA.cu:289-333).
 * in  : int32 [m][k]                         out : int32 [ceil(m/16)][ceil(k/(16 I))][32][I]
 * requires I in {1,2,4}; ragged m, k are zero-padded */
TG_API int tg_convert_to_Aint4(const int32_t* in, int64_t m, int64_t k, int inner_k_tiles, int32_t* out,
                        int device, tg_stream_t stream);

/* replace convert_matrix_{to,from}_m16n8k16_A_layout (TinyGemmConvertA.cu:150-223, 554-626).
 * 16-bit payload (bf16 or fp16: pure data movement).  rm: [m][k]; tc: [ceil(m/16)][ceil(k/16)][32][8] */
TG_API int tg_convert_to_A16(const void* rm, int64_t m, int64_t k, void* tc, int device, tg_stream_t stream);
TG_API int tg_convert_from_A16(const void* tc, int64_t m, int64_t k, void* rm, int device, tg_stream_t stream);

/* replace convert_matrix_{to,from}_m16n8k16_B_layout (TinyGemmConvertB.cu:76-133, 186-249).
 * rm: [n][k]; tc: [ceil(n/8)][ceil(k/(16 I))][32][4 I], I in {1,2} */
TG_API int tg_convert_to_B16(const void* rm, int64_t n, int64_t k, int inner_k_tiles, void* tc, int device,
                      tg_stream_t stream);
TG_API int tg_convert_from_B16(const void* tc, int64_t n, int64_t k, int inner_k_tiles, void* rm, int device,
                        tg_stream_t stream);

/* The inverse of the two int4 packers (no counterpart in the reference: it never unpacks): packed words -> int32 codes [rows][k].
 * layout_a = 0: Bint4 words [ceil(rows/8)][k/(16 I)][32][I/2], I in {2,4,8}; layout_a = 1: Aint4 words [ceil(rows/16)][ceil(k/(16 I))][32][I],
 * I in {1,2,4}.  With a packer this is the lossless repack between the reference's Aint4 words and TG_WFMT_ROWS (tg_w4_gemm.w_format). */
TG_API int tg_unpack_int4(const int32_t* packed, int layout_a, int64_t rows, int64_t k, int inner_k_tiles, int32_t* codes,
                          int device, tg_stream_t stream);

/* replaces the debug op tinygemm_dequant_int4 (TinyGemmDequantize.cu:36-58):
 * each int32 -> 8 bf16 (nibble - 8) in the order [n0,n4,n1,n5,n2,n6,n3,n7]. */
TG_API int tg_dequant_int4(const int32_t* in, int64_t count, void* out_bf16, int device, tg_stream_t stream);

/* ---- the hot path: Y[act][wrow] = X[act][k] . dequant(W)[wrow][k]^T ----------------------- */

/*
 * replaces tinygemm_y_f16RM_x_f16RM_w_{int4,any4,mx4}TC
 * (TinyGemm_int4.cu:294-548 -> launch_tinygemm_kernel, TinyGemmImpl.cuh:347-431 ->
 *  tinygemm_m16n8k16_chunk_kernel :23-345 with {A,B}Layout_TC_int4 + {A,B}Layout_RM).
 *
 * Both sides of the reference API produce the same row-major result indexed
 * [activation row][weight row] (weightOnRight: [m][n]; otherwise [n_act][m_w],
 * TinyGemm_int4.cu:450-456), so one entry point serves both; `w_on_right` only says
 * which packed layout `w` is in.
 */
typedef struct tg_w4_gemm {
  /* ---- ABI version 6: the struct says how long it is.  sizeof(struct tg_w4_gemm) as the CALLER was compiled: the library reads
   *      exactly that many bytes and takes every later field as zero (= the feature is off), so a binding written against an
   *      older header keeps working when fields are appended; a value below the ABI-1 prefix (everything up to stride_y) or above
   *      the library's own struct is rejected with TG_E_STRUCT, never read past.  C: `struct tg_w4_gemm a = {sizeof a};` ---- */
  uint32_t struct_bytes;
  uint32_t struct_reserved; /* must be 0 */
  const void* x;     /* activations, RM 16-bit [m][k]                                          */
  const void* w;     /* packed weights: Bint4 if w_on_right else Aint4                         */
  const void* qinfo; /* int4/any4: 16-bit scales_and_zeros [k/group][wrows][2] (scale, zero)   */
                     /* mx4: uint8 e8m0 exponents [wrows][k/group]                              */
  const void* lut;   /* any4: 16-bit LUT [16] (global) or [wrows][16] (row-wise); else NULL    */
  void* y;           /* out, RM 16-bit [m][wrows]                                              */
  int64_t m;         /* activation rows (any m >= 1; tuned for m <= 16)                        */
  int64_t wrows;     /* weight rows INCLUDING tile padding: 8*size(0) (Bint4) / 16*size(0) (Aint4) */
  int64_t k;         /* reduction length; k % 32 == 0 and k % (16 I) == 0                      */
  int32_t group;     /* quantisation group size along k: 32, 64, 128 or 256                    */
  int32_t qtype;     /* TG_Q_*                                                                 */
  int32_t dtype;     /* TG_BF16 / TG_F16 (mx4: bf16 only)                                      */
  int32_t w_on_right;    /* 1: w is Bint4, 0: w is Aint4                                      */
  int32_t inner_k_tiles; /* I of the packed layout                                            */
  /* optional stacked launch over `batch` independent problems of identical shape:
   * operand b lives at base + b*stride (bytes).  batch <= 1 ignores the strides.             */
  int32_t batch;
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y;
  /* ---- ABI version 2 ---- */
  int32_t numerics;   /* TG_NUM_FAST (0) or TG_NUM_REFERENCE                                      */
  int32_t reserved;   /* must be 0                                                                */
  const void* bias;   /* optional 16-bit [wrows]: y[a][row] = RNE16(RNE16(acc) + bias[row]), i.e. bit-identical to the
                         reference module's separate `y + bias` (modules.py:221-222) without its extra launch; NULL = none */
  int64_t stride_bias;
  /* ---- ABI version 3 ---- */
  void* workspace;         /* optional device scratch (16-byte aligned), written by the call on `stream`; NULL = none.       */
  int64_t workspace_bytes; /* tg_gemm_w4_workspace_bytes() says how much lets the fastest kernel run; with less (or none) the  */
                           /* call still succeeds on a kernel that needs no scratch.  Do not share one workspace between       */
                           /* calls that may overlap (different streams).                                                      */
  int32_t x_layout;        /* TG_LAYOUT_RM (0) or TG_LAYOUT_TC_A: activations in the m16n8k16 A-fragment order                 */
                           /*   [m/16][k/16][32][8] of tinygemm_y_f16TC_x_f16TC_w_*TC (weightOnRight; TinyGemm_int4.cu:28-292,  */
                           /*   MatrixLayoutA.cuh:211-373): m % 16 == 0, w_on_right = 1, no bias.                              */
  int32_t y_layout;        /* same for the output: [m/16][ceil(wrows/16)][32][8] (allocate zero-filled when wrows % 16 != 0)   */
                           /* Only the TG_NUM_FAST kernels read / write fragment order themselves; otherwise TG_E_LAYOUT       */
                           /* (tg_gemm_w4_plan reports it without launching): convert around a row-major call instead.         */
  /* ---- ABI version 5: stages of a decoder layer fused into the GEMM launch (all-zero = off).  A hipGraph of dependent
   *      kernels advances at >= 4.5 us per node on MI355X whatever the node does, so the element-wise stages around the four
   *      GEMMs of a layer (the reference times them through HuggingFace's LlamaDecoderLayer, benchmark.py:113-215) cost as
   *      much as the GEMMs at batch 1.  Rounding points are those of the separate kernels of include/decode_glue_hip.h. ---- */
  int64_t bias_row_stride; /* elements between the `bias` rows of consecutive ACTIVATION rows: 0 = one [wrows] row for all (a       */
                           /* Linear's bias); wrows = a full [m][wrows] addend, i.e. the residual stream:                          */
                           /* y[a][row] = RNE16(RNE16(acc) + bias[a * bias_row_stride + row]).  Every 4-/8-bit kernel.             */
  const void* norm_weight; /* non-NULL: LlamaRMSNorm of the activations inside the launch's activation staging, 16-bit [k]:         */
                           /* x'[a][j] = RNE16(RNE16(x[a][j] * rsqrt(mean_j(x[a][j]^2) + norm_eps)) * norm_weight[j])               */
                           /* (dg_add_rmsnorm's formula).  TG_NUM_FAST pair-table kernels, row-major x, k % 2048 == 0, the          */
                           /* activation block staged whole on chip; otherwise TG_E_FUSION.  Launches that w4_gemv_kernel takes    */
                           /* (m <= 4, one problem; tg_gemm_w4_plan = TG_PLAN_GEMV) apply the row's factor to the f32 sum instead:  */
                           /* y[a][row] = RNE16(rsqrt(mean_j(x[a][j]^2) + norm_eps) * sum_j RNE16(x[a][j] * norm_weight[j]) w[row][j]) */
                           /* -- one rounding of the activation where the separate kernel has two; within 2^-8 sum|x' w| of it.     */
  float norm_eps;
  int32_t epilogue;        /* TG_EPI_NONE or TG_EPI_SWIGLU: the weight rows come in blocks of 16 = 8 "gate" rows followed by the 8  */
                           /* matching "up" rows, and y is [m][wrows / 2]:                                                         */
                           /* y[a][8 B + c] = RNE16(RNE16(silu(g)) * u), g / u = RNE16(acc) of rows 16 B + c / 16 B + 8 + c         */
                           /* (dg_swiglu's formula).  TG_NUM_FAST pair-table kernels, weights on the right, row-major y, no bias;   */
                           /* otherwise TG_E_FUSION.                                                                               */
  /* ---- ABI version 6 ---- */
  int32_t w_format;        /* w_on_right = 0 only.  TG_WFMT_M16N8K16 (0): `w` holds the reference's Aint4 words.  TG_WFMT_ROWS (1): `w`  */
                           /* is the Bint4 tensor [wrows/8][k/(16 J)][32][J/2] of the weight rows (padded to 16; J = 4 when k % 64 == 0,  */
                           /* else 2; inner_k_tiles is then only range-checked) -- what SURVEY 8(b) calls a native                        */
                           /* packed layout behind convert_matrix_to_m16n8k16_Aint4_layout: a packed word then holds 8 codes of ONE weight */
                           /* row instead of 4 + 4 of rows r and r + 8, and the A-side ops run the B-side kernels (the result is          */
                           /* [activation row][weight row] either way).  tg_convert_to_Bint4 on the [wrows][k] codes produces it,         */
                           /* tg_unpack_int4 + a packer converts between the two losslessly.                                              */
  int32_t reserved6;       /* must be 0 */
} tg_w4_gemm;

enum { TG_EPI_NONE = 0, TG_EPI_SWIGLU = 1 };
enum { TG_WFMT_M16N8K16 = 0, TG_WFMT_ROWS = 1 };

TG_API int tg_gemm_w4(const tg_w4_gemm* args, int device, tg_stream_t stream);

/* Which kernel family tg_gemm_w4 would launch for these arguments (same validation, nothing is launched; the data
 * pointers are only checked for NULL / alignment).  Negative: the TG_E_* code tg_gemm_w4 would return.
 *   TG_PLAN_SPLITK  w4_gemm_kernel         one 16-wave split-K workgroup per 16-row tile (small launches), reference numerics
 *   TG_PLAN_STREAM  w4_gemm_stream_kernel  per-(row, group) tables of final 16-bit weights, reference numerics
 *   TG_PLAN_PAIR    w4_gemm_pair_kernel    per-row tables of LUT pairs, group-scaled numerics (TG_NUM_FAST; mx4 -- converted in
 *                                          registers by v_cvt_scalef32_pk_bf16_fp4, exact -- in both numerics)
 *   TG_PLAN_PAIR_XR w4_gemm_xr_kernel      the same tables and numerics, activations resident in registers (Bint4 weights,
 *                                          2 ... 16 activation rows, k = 4096, stacked launches)
 *   TG_PLAN_GEMV    w4_gemv_kernel         the same tables and numerics for ONE layer per launch with 1 ... 4 activation rows (a
 *                                          decode step's GEMMs): one workgroup per CU over a contiguous range of weight rows,
 *                                          v_dot2 contraction, fused norm / residual / SwiGLU stages
 *   TG_PLAN_TILE    w4_gemm_tile_kernel    more than 64 activation rows (Bint4 innerKTiles 4, int4 / any4): an LDS-tiled MFMA GEMM
 *                                          (128 x 64 / 128 tiles) whose weight tile is dequantised on the way in -- the reference's
 *                                          weights bit for bit (both numerics settings), no workspace  */
enum { TG_PLAN_SPLITK = 1, TG_PLAN_STREAM = 2, TG_PLAN_PAIR = 3, TG_PLAN_PAIR_XR = 4, TG_PLAN_GEMV = 5, TG_PLAN_TILE = 6 };
enum { TG_LAYOUT_RM = 0, TG_LAYOUT_TC_A = 1 };
TG_API int tg_gemm_w4_plan(const tg_w4_gemm* args, int device);

/* Bytes of `workspace` with which tg_gemm_w4 takes its fastest kernel for these arguments (0: none needed; negative: the
 * TG_E_* code tg_gemm_w4 would return).  Today: TG_NUM_FAST, Bint4 weights, stacked launches whose activation block of one
 * pass does not fit next to the pair table in LDS (m > 4 at k = 4096, any m at k >= 8192): the activations are re-arranged
 * once per call (w4_xprep_kernel) so that the waves can stream them like the weights.  Needs no GPU. */
TG_API int64_t tg_gemm_w4_workspace_bytes(const tg_w4_gemm* args);

/* out[wrows][k] (16-bit, row-major) = the dequantised weights of a Bint4-packed tensor (also the native weights-on-the-left format,
 * which holds the same words): w = RNE16(fma(lut[row][code], scale[g][row], zero[g][row])), the reference's per-element formula
 * (MatrixLayoutB.cuh:1042-1046; int4: code - 8) -- what quantize.py:612-637 (anyq_dequantize_tensor) computes op by op on unpacked codes.  For MANY
 * activation rows (beyond 64) the Python layer multiplies by this matrix with the GEMM library instead of walking the 4-bit
 * weights once per 16-row tile: qinfo [k/group][wrows][2], lut [wrows][16] / [16] / NULL (int4), I = innerKTiles of the packed tensor.
 * k % 512 == 0; mx4 is not covered (TG_E_QTYPE). */
TG_API int tg_dequant_w4(const void* packed, const void* qinfo, const void* lut, int64_t wrows, int64_t k, int group, int qtype, int dtype,
                         int inner_k_tiles, void* out, int device, tg_stream_t stream);
/* ABI 7: a PANEL of `wrows` weight rows of a matrix whose quantisation info has `wrows_q` rows per group (qinfo [k / group][wrows_q][2]):
 * packed / qinfo / lut point at the panel's first row (a multiple of 8); out [wrows][k].  tg_dequant_w4 is the panel wrows_q = wrows. */
TG_API int tg_dequant_w4_panel(const void* packed, const void* qinfo, const void* lut, int64_t wrows, int64_t wrows_q, int64_t k, int group,
                               int qtype, int dtype, int inner_k_tiles, void* out, int device, tg_stream_t stream);

/*
 * int8 weights.
 * tg_convert_to_Bint8 replaces convert_matrix_to_m16n8k16_Bint8_layout (TinyGemmConvertB.cu:415-465, kernel 366-411):
 *   in int32 [n][k] (byte codes) -> out int32 [ceil(n/8)][k/(16 I)][32][I], I in {1,2,4}, k % (16 I) == 0.
 * tg_convert_to_Aint8 replaces convert_matrix_to_m16n8k16_Aint8_layout (TinyGemmConvertA.cu:400-440, kernel 337-397):
 *   in int32 [m][k] -> out int32 [ceil(m/16)][ceil(ceil(k/16)/I)][32][2 I], I in {1,2}.
 * Both are bit-identical to the reference's words.
 * tg_gemm_w8 replaces tinygemm_y_f16RM_x_f16RM_w_int8TC (TinyGemm_int8.cu:216-399, 430-458): same argument struct as
 *   tg_gemm_w4 with qtype = TG_Q_INT8, inner_k_tiles = I of the packed layout (B: 1,2,4; A: 1,2 = size(3)/2), lut unused;
 *   w = RNE16(fma(byte - 128, scale, zero)) (Dequantization.cuh:262-330, MatrixLayoutB.cuh:1296-1316).
 *   x (and stride_x of a batch) must be 16-byte aligned (TG_E_ALIGN otherwise): the kernels load activation fragments in 16-byte pieces.
 */
TG_API int tg_convert_to_Bint8(const int32_t* in, int64_t n, int64_t k, int inner_k_tiles, int32_t* out, int device,
                               tg_stream_t stream);
TG_API int tg_convert_to_Aint8(const int32_t* in, int64_t m, int64_t k, int inner_k_tiles, int32_t* out, int device,
                               tg_stream_t stream);
TG_API int tg_gemm_w8(const tg_w4_gemm* args, int device, tg_stream_t stream);
/* ABI version 8: bytes of `workspace` with which tg_gemm_w8 takes its fastest kernel (needs no GPU; 0: none; negative: the TG_E_* code
 * tg_gemm_w8 would return): many activation rows of innerKTiles-2 words run the tile GEMM as a split-K launch (f32 partial tiles).  Without
 * a workspace the call still succeeds (unsplit, or on the 16-row kernel). */
TG_API int64_t tg_gemm_w8_workspace_bytes(const tg_w4_gemm* args);

/*
 * replaces tinygemm_y_f16RM_x_f16RM_w_f16TC (TinyGemm_bf16.cu:163-327): un-quantised 16-bit
 * weights in A16 (w_on_right = 0) or B16 (w_on_right = 1, I in {1,2}) fragment order.
 * x RM [m][k]; y RM [m][wrows]; wrows includes tile padding.
 */
TG_API int tg_gemm_f16(const void* x, const void* w, void* y, int64_t m, int64_t wrows, int64_t k, int dtype,
                int w_on_right, int inner_k_tiles, int device, tg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TINYGEMM_HIP_H_ */
