// tg_tile.hip -- launch path of w4_gemm_tile_kernel (MANY activation rows: the LDS-tiled MFMA GEMM that dequantises on the way into its
// weight tile; w4_gemm_tile.cuh); see tg_common.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm_tile.cuh"

// KS = 2 (two super-tiles per step, 128 x 64 tiles, k % 128 == 0): 5-7 % faster than KS = 1 there (4096^2 at m = 512: 37.4 -> 34.7 us); the
// 128 x 128 tile does not fit two super-tiles per stage in 160 KiB
template <typename DT, int BN, int KS>
int go(const TileParams& tp, hipStream_t st) {
  constexpr int DX = KS == 2 ? 2 : 3;
  constexpr auto kern = w4_gemm_tile_kernel<DT, 128, BN, DX, 8, KS>;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA;  // (a part with less LDS: the older kernels take over)
  constexpr unsigned lds = TileLds<128, BN, DX, KS>::BYTES;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tp.tiles_m * tp.tiles_n)), dim3(1024), lds, st, tp);
  return launch_status();
}
}  // namespace

namespace tgx {
// Bint4 words of innerKTiles 4 (k % 64 == 0), int4 / any4 (global or per-row LUT), row-major operands, no fused norm / SwiGLU; any
// numerics setting: the kernel computes the reference's own weights, RNE16(fma(lut, scale, zero)).  TG_PAIR_NA: not this kernel's call.
int tile(int dt, int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st) {
  if (I != 4 || qmx || p.x_tc || p.y_tc || p.norm_w || p.epilogue || p.bias_row_stride || p.m < TG_TILE_MIN_M) return TG_PAIR_NA;
  if (p.k % 64 != 0 || p.wrows % 8 != 0 || p.wrows < 8) return TG_PAIR_NA;
  if (!(p.qtype == TG_Q_INT4 || p.qtype == TG_Q_ANY4_GLOBAL || p.qtype == TG_Q_ANY4_ROWWISE)) return TG_PAIR_NA;
  if (p.dry) return TG_PLAN_TILE;
  const int tiles_m = (p.m + 127) / 128;
  // 128 x 128 tiles once they fill the chip (half the activation traffic per weight row), else 128 x 64 (twice the workgroups)
  const bool wide = (int64_t)tiles_m * ((p.wrows + 127) / 128) >= cu_count();
  for (int64_t b = 0; b < batch; ++b) {
    TileParams tp;
    tp.x = p.x + b * p.stride_x; tp.w = p.w + b * p.stride_w; tp.qinfo = p.qinfo + b * p.stride_qinfo;
    tp.lut = p.lut ? p.lut + b * p.stride_lut : nullptr; tp.y = p.y + b * p.stride_y;
    tp.bias = p.bias ? p.bias + b * p.stride_bias : nullptr;
    tp.m = p.m; tp.wrows = p.wrows; tp.k = p.k; tp.ksuper = p.ksuper; tp.gshift = p.gshift; tp.qtype = p.qtype;
    tp.tiles_m = tiles_m; tp.tiles_n = (p.wrows + (wide ? 127 : 63)) / (wide ? 128 : 64);
    int rc;
    const bool two = !wide && p.ksuper % 2 == 0;
    if (dt == TG_BF16) rc = wide ? go<BF16, 128, 1>(tp, st) : two ? go<BF16, 64, 2>(tp, st) : go<BF16, 64, 1>(tp, st);
    else rc = wide ? go<F16, 128, 1>(tp, st) : two ? go<F16, 64, 2>(tp, st) : go<F16, 64, 1>(tp, st);
    if (rc != 0) return rc;
  }
  return 0;
}
}  // namespace tgx
