/* decode_glue_hip.h -- C ABI of the non-GEMM kernels of one batch-1 decode step.
 *
 * Not part of the tinygemm drop-in boundary (include/tinygemm_hip.h).  The reference times model-level
 * decode through HuggingFace `transformers` (benchmark.py:113-215 -> LlamaDecoderLayer, an external
 * dependency that is not vendored in the reference): per layer that is ~45 small elementwise / reduction
 * launches around the 4 (fused) quantized GEMMs, which at batch 1 cost 3x the GEMMs themselves.  These
 * five kernels replace them so that the model-level number measures the GEMM path, not launch gaps:
 *
 *   dg_add_rmsnorm   h' = h + delta;  y = rmsnorm(h') * w          (residual add + LlamaRMSNorm)
 *   dg_rope_kv       rotary embedding of q and k, k/v written into the static KV cache at *pos
 *   dg_decode_attn   grouped-query attention of one new token against cache[0 .. *pos]
 *   dg_rope_attn     the two above fused (what the decode harness launches)
 *   dg_swiglu        silu(gate) * up
 *
 * Conventions are those of include/tinygemm_hip.h: raw device pointers, caller-owned outputs, explicit
 * stream, no allocation, no host sync, graph-capturable (`pos` is read on the device, so a captured graph
 * can be replayed for the next position), return 0 / TG_E_* (negative) / hipError_t (positive).
 * All 16-bit tensors are bf16 (TG_BF16) or fp16 (TG_F16); cos/sin tables are float32.
 * Rounding points follow the plain-torch formulation in any4_amd/decode.py (tests compare the two).
 */
#ifndef DECODE_GLUE_HIP_H
#define DECODE_GLUE_HIP_H

#include "tinygemm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* h_out[r][:] = h[r][:] + delta[r][:] (delta may be NULL: h_out = h; h_out may alias h);
 * y[r][:] = to16(float(h_out) * rsqrt(mean(float(h_out)^2) + eps)) * w   (y may be NULL: residual add only).
 * dim % 8 == 0, dim <= 16384. */
TG_API int dg_add_rmsnorm(const void* h, const void* delta, const void* w, void* h_out, void* y,
                          int64_t rows, int64_t dim, float eps, int dtype, int device, tg_stream_t stream);

/* qkv [bs][(hl + 2 kvl) * d] (q heads, then k heads, then v heads); cos/sin float32 [max_seq][d]
 * (both halves filled, HF convention); *pos = sequence position (int64 on the device).
 * q_out [bs][hl][d] = rope(q); k_cache[b][kv][*pos][:] = rope(k); v_cache[b][kv][*pos][:] = v.
 * caches are [bs][kvl][max_seq][d]; d even, d <= 256. */
TG_API int dg_rope_kv(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* q_out,
                      void* k_cache, void* v_cache, int64_t bs, int hl, int kvl, int d, int64_t max_seq,
                      int dtype, int device, tg_stream_t stream);

/* out[b][h][:] = softmax_s( to16(q[b][h] . k_cache[b][h / (hl/kvl)][s]) * scale ) over s <= *pos, applied to
 * v_cache; probabilities rounded to 16 bit before the value contraction, fp32 accumulation, one rounding of
 * the output.  d % 8 == 0, d <= 256, max_seq <= 8192. */
TG_API int dg_decode_attn(const void* q, const void* k_cache, const void* v_cache, const int64_t* pos, void* out,
                          int64_t bs, int hl, int kvl, int d, int64_t max_seq, float scale, int dtype, int device,
                          tg_stream_t stream);

/* dg_rope_kv followed by dg_decode_attn in ONE launch (a hipGraph of dependent kernels advances at ~5 us per node on
 * MI355X): same arguments, same arithmetic in the same order, bit-identical results; q is not
 * materialised.  d % 8 == 0, d <= 256, max_seq <= 8192. */
TG_API int dg_rope_attn(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache,
                        void* v_cache, void* out, int64_t bs, int hl, int kvl, int d, int64_t max_seq, float scale,
                        int dtype, int device, tg_stream_t stream);

/* dg_rope_attn for latency (what the decode harness launches at d = 64 / 128): every load of the launch is issued behind the read
 * of `pos`, scores are reduced with DPP, the softmax statistics are kept per row group and combined once (flash-decoding style,
 * ONE barrier), so probabilities are normalised after the value contraction: the cache rows written are bit-identical to
 * dg_rope_kv's, the output agrees with dg_rope_attn within 16-bit rounding.  qkv and the rotary tables 16-byte aligned. */
TG_API int dg_rope_attn_online(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache,
                               void* v_cache, void* out, int64_t bs, int hl, int kvl, int d, int64_t max_seq, float scale,
                               int dtype, int device, tg_stream_t stream);

/* dg_rope_attn with the sequence split over `nsplit` blocks per head (flash-decoding style combine by the last block to
 * arrive): fills the GPU at batch 1 and long contexts.  With d = 64 / 128 (and 16-byte aligned qkv / tables) it is
 * dg_rope_attn_online's one-barrier kernel with gridDim.y = nsplit: block c of a head takes the 32-row iterations c, c + nsplit,
 * ... of the context; what crosses blocks goes through agent-scope atomics, not through a device-scope fence.  `scratch`: dg_rope_attn_split_scratch_bytes(...) bytes, 16-byte
 * aligned, ZEROED ONCE by the caller before the first launch (the kernel leaves its counters at zero again); launches
 * sharing a scratch buffer must be stream-ordered.  Probabilities are normalised after the value contraction, so results
 * agree with dg_rope_attn within 16-bit rounding, not bit for bit.  max_seq / nsplit <= ~15000. */
TG_API int64_t dg_rope_attn_split_scratch_bytes(int64_t bs, int hl, int d, int nsplit);
TG_API int dg_rope_attn_split(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache,
                              void* v_cache, void* out, void* scratch, int64_t scratch_bytes, int64_t bs, int hl, int kvl,
                              int d, int64_t max_seq, float scale, int nsplit, int dtype, int device, tg_stream_t stream);

/* out[b][j] = to16(silu(gu[b][j])) * gu[b][il + j], j < il; gu [bs][2 il]; il % 8 == 0. */
TG_API int dg_swiglu(const void* gu, void* out, int64_t bs, int64_t il, int dtype, int device, tg_stream_t stream);

/* y[m][n] = RNE16(x[m][k] . w[n][k]^T), ROW-MAJOR 16-bit weights (an nn.Linear's), f32 accumulation: the LM head of the decode step,
 * which the reference leaves un-quantised (quantize.py:34-36).  m = 1 ... 4 at k = 2048 / 4096, 1 ... 2 at k = 8192 (TG_E_SHAPE otherwise: the caller
 * keeps its GEMM); x and w 16-byte aligned. */
TG_API int dg_linear16(const void* x, const void* w, void* y, int64_t m, int64_t n, int64_t k, int dtype, int device, tg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
