// tg_tile.hip -- launch path of w4_gemm_tile_kernel (MANY activation rows: the LDS-tiled MFMA GEMM that dequantises on the way into its
// weight tile; w4_gemm_tile.cuh); see tg_common.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm_tile.cuh"

// KS = 2 (two super-tiles per step, 128 x 64 tiles, k % 128 == 0): 5-7 % faster than KS = 1 there (4096^2 at m = 512: 37.4 -> 34.7 us); the
// 128 x 128 tile does not fit two super-tiles per stage in 160 KiB
template <typename DT, int BM, int BN, int KS, bool QMX = false>
int go(const TileParams& tp, hipStream_t st) {
  constexpr int DX = KS == 2 ? 2 : 3;
  constexpr auto kern = w4_gemm_tile_kernel<DT, BM, BN, DX, 8, KS, 4, QMX>;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA;  // (a part with less LDS: the older kernels take over)
  constexpr unsigned lds = TileLds<BM, BN, DX, KS>::BYTES;
  const int ns = tp.splits > 1 ? tp.splits : 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tp.tiles_m * tp.tiles_n * ns)), dim3(1024), lds, st, tp);
  if (ns > 1) {
    const int64_t quads = (int64_t)tp.m * tp.wrows / 4;
    hipLaunchKernelGGL(tile_split_sum_kernel<DT>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, tp.part, ns, (int64_t)tp.m * tp.wrows, tp.y,
                       tp.bias, tp.wrows, quads);
  }
  return launch_status();
}
template <typename DT, bool QMX = false>
int go_dt(const TileParams& tp, bool wide, bool small, bool two, hipStream_t st) {
  if constexpr (!QMX) {   // (mx4: the 128 x 128 tile's table build spills at 128 registers; 128 x 64 tiles throughout)
    if (wide) return go<DT, 128, 128, 1>(tp, st);
  }
  if (small) return two ? go<DT, 64, 64, 2, QMX>(tp, st) : go<DT, 64, 64, 1, QMX>(tp, st);
  return two ? go<DT, 128, 64, 2, QMX>(tp, st) : go<DT, 128, 64, 1, QMX>(tp, st);
}
template <typename DT, int KS, int W8>
int go_w8(const TileParams& tp, hipStream_t st) {
  constexpr int DX = KS == 2 ? 2 : 3;
  constexpr auto kern = w4_gemm_tile_kernel<DT, 128, 64, DX, 8, KS, 4, false, W8>;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA;
  constexpr unsigned lds = TileLds<128, 64, DX, KS>::BYTES;
  const int ns = tp.splits > 1 ? tp.splits : 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tp.tiles_m * tp.tiles_n * ns)), dim3(1024), lds, st, tp);
  if (ns > 1) {
    const int64_t quads = (int64_t)tp.m * tp.wrows / 4;
    hipLaunchKernelGGL(tile_split_sum_kernel<DT>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, tp.part, ns, (int64_t)tp.m * tp.wrows, tp.y,
                       tp.bias, tp.wrows, quads);
  }
  return launch_status();
}
}  // namespace

namespace tgx {
// int8 weights (tg_gemm_w8) at more than TG_TILE_W8_MIN_M - 1 activation rows on the same tile GEMM: Bint8 / Aint8 words of innerKTiles 2 (what
// Int8Linear packs by default, modules.py:85-152), groups of 64 or more, k % 64 == 0; 128 x 64 tiles, split-K like the 4-bit launches.  The
// 16-row kernel this replaces walks m in 16-row tiles that each re-read (and re-convert) the weights: 4096^2 at 128 / 512 / 2048 rows 130 / 509 /
// 2023 us.  TG_PAIR_NA: not this kernel's call.
int tile_w8(int dt, bool on_right, int I, GemmParams& p, int64_t batch, hipStream_t st) {
  if (I != 2 || p.gshift < 6 || p.k % 64 != 0 || p.bias_row_stride || p.m < TG_TILE_W8_MIN_M || p.wrows % (on_right ? 8 : 16) != 0) return TG_PAIR_NA;
  const int ksuper = p.k / 64;
  const int tiles_m = (p.m + 127) / 128, tiles_n = (p.wrows + 63) / 64;
  const int g64 = (1 << p.gshift) / 64;        // 64-k super-tiles per quantisation group (>= 1)
  int splits = 1;
  while (splits < 8 && (int64_t)tiles_m * tiles_n * splits * 2 <= cu_count() && ksuper % (splits * 2) == 0 && (ksuper / (splits * 2)) % g64 == 0 &&
         ksuper / (splits * 2) >= 8)
    splits *= 2;
  const int64_t need = splits > 1 ? (int64_t)splits * p.m * p.wrows * 4 : 0;
  if (splits > 1 && !p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) splits = 1;
  p.ws_need = splits > 1 ? need : 0;
  if (p.dry) return TG_PLAN_TILE;
  const bool two = p.gshift >= 7 && (ksuper / splits) % 2 == 0;   // (two super-tiles per step: one group per step needs g >= 128)
  for (int64_t b = 0; b < batch; ++b) {
    TileParams tp;
    tp.x = p.x + b * p.stride_x; tp.w = p.w + b * p.stride_w; tp.qinfo = p.qinfo + b * p.stride_qinfo; tp.lut = nullptr;
    tp.y = p.y + b * p.stride_y; tp.bias = p.bias ? p.bias + b * p.stride_bias : nullptr;
    tp.m = p.m; tp.wrows = p.wrows; tp.k = p.k; tp.ksuper = ksuper; tp.gshift = p.gshift; tp.qtype = p.qtype;
    tp.tiles_m = tiles_m; tp.tiles_n = tiles_n;
    tp.splits = splits; tp.part = splits > 1 ? reinterpret_cast<float*>(p.ws) : nullptr;
    tp.x_pitch = p.k;
    int rc;
    if (dt == TG_BF16) rc = on_right ? (two ? go_w8<BF16, 2, 1>(tp, st) : go_w8<BF16, 1, 1>(tp, st)) : (two ? go_w8<BF16, 2, 2>(tp, st) : go_w8<BF16, 1, 2>(tp, st));
    else rc = on_right ? (two ? go_w8<F16, 2, 1>(tp, st) : go_w8<F16, 1, 1>(tp, st)) : (two ? go_w8<F16, 2, 2>(tp, st) : go_w8<F16, 1, 2>(tp, st));
    if (rc != 0) return rc;
  }
  return 0;
}

// Bint4 words of innerKTiles 4 (k % 64 == 0), int4 / any4 (global or per-row LUT) / mx4, row-major operands, no fused norm / SwiGLU; any
// numerics setting: the kernel computes the reference's own weights, RNE16(fma(lut, scale, zero)).  TG_PAIR_NA: not this kernel's call.
//
// Split-K: a tile's k-steps are a chain of dependent LDS round trips (0.6-0.9 us per 128 k whatever the tile holds), so a launch with fewer
// tiles than CUs leaves the chip idle AND takes as long as a full one.  With the caller's workspace the k range is cut into 2 / 4 / 8 splits
// (as many as keep tiles x splits <= CUs), f32 partial tiles, one small kernel adds them in split order: 4096^2 at m = 64 / 128 / 256:
// 15.4 / 17.9 / 25.5 us against 28.6 / 29.0 / 30.1 unsplit (and 28 us as four 16-row passes at m = 64); 128 x 4096 x 14336: 97 -> 36 us.
// One layer per call from 17 rows on (64 x 64 tiles up to 64 rows) when the split is available (through the modules, graph nodes of one
// 4096^2 layer at 33 / 48 / 64 / 128 rows: 13.6 / 14.2 / 14.8 / 19.5 us; 16-row passes: 19.6 / 22.2 / 29.2 / -); without a workspace: from 65 rows, unsplit.
int tile(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  if (I != 4 || p.x_tc || p.y_tc || p.norm_w || p.epilogue || p.bias_row_stride || p.m < TG_TILE_MIN_M_SPLIT) return TG_PAIR_NA;
  if (p.k % 64 != 0 || p.wrows % 8 != 0 || p.wrows < 8) return TG_PAIR_NA;
  if (!(p.qtype == TG_Q_INT4 || p.qtype == TG_Q_ANY4_GLOBAL || p.qtype == TG_Q_ANY4_ROWWISE || p.qtype == TG_Q_MX4)) return TG_PAIR_NA;
  if ((p.qtype == TG_Q_MX4) != qmx || (qmx && (dt != TG_BF16 || p.gshift != 5))) return TG_PAIR_NA;   // (mx4: bf16, groups of 32: TinyGemm_int4.cu:758)
  const bool small = p.m <= 64;
  const int tiles_m = small ? 1 : (p.m + 127) / 128;
  // 128 x 128 tiles once they fill the chip (half the activation traffic per weight row), else 128 x 64 (twice the workgroups)
  const bool wide = !small && !qmx && (int64_t)tiles_m * ((p.wrows + 127) / 128) >= cu_count();
  const int tiles_n = (p.wrows + (wide ? 127 : 63)) / (wide ? 128 : 64);
  int splits = 1;
  if (!wide) {
    const int64_t tiles = (int64_t)tiles_m * tiles_n;
    const int g64 = (1 << p.gshift) > 64 ? (1 << p.gshift) / 64 : 1;   // super-tiles per quantisation group
    // (at least eight super-tiles = 512 k per split: below that the prologue and the sum kernel cost more than the split saves)
    while (splits < 8 && tiles * splits * 2 <= cu_count() && p.ksuper % (splits * 2) == 0 && (p.ksuper / (splits * 2)) % g64 == 0 &&
           p.ksuper / (splits * 2) >= 8)
      splits *= 2;
  }
  const int64_t need = splits > 1 ? (int64_t)splits * p.m * p.wrows * 4 : 0;
  if (splits > 1 && !p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) splits = 1;   // (the caller did not bring the workspace)
  // (up to 64 rows 16-row passes are faster than an unsplit tile, and STACKED layers are faster in 16-row passes: one launch per pass over all of them)
  if ((splits == 1 || batch != 1) && p.m < TG_TILE_MIN_M) return TG_PAIR_NA;
  p.ws_need = splits > 1 ? need : 0;
  if (p.dry) return TG_PLAN_TILE;
  for (int64_t b = 0; b < batch; ++b) {
    TileParams tp;
    tp.x = p.x + b * p.stride_x; tp.w = p.w + b * p.stride_w; tp.qinfo = p.qinfo + b * p.stride_qinfo;
    tp.lut = p.lut ? p.lut + b * p.stride_lut : nullptr; tp.y = p.y + b * p.stride_y;
    tp.bias = p.bias ? p.bias + b * p.stride_bias : nullptr;
    tp.m = p.m; tp.wrows = p.wrows; tp.k = p.k; tp.ksuper = p.ksuper; tp.gshift = p.gshift; tp.qtype = p.qtype;
    tp.tiles_m = tiles_m; tp.tiles_n = tiles_n;
    tp.splits = splits; tp.part = splits > 1 ? reinterpret_cast<float*>(p.ws) : nullptr;   // (the launches of a batch are ordered on the stream: one scratch)
    tp.x_pitch = p.k;
    const bool two = !wide && (p.ksuper / splits) % 2 == 0;
    const int rc = qmx ? go_dt<BF16, true>(tp, wide, small, two, st) : dt == TG_BF16 ? go_dt<BF16>(tp, wide, small, two, st) : go_dt<F16>(tp, wide, small, two, st);
    if (rc != 0) return rc;
  }
  return 0;
}
}  // namespace tgx
