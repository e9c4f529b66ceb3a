// tg_pair16.hip -- launch path of w4_gemm_pair16_kernel (one layer per launch); see tg_common.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm_pair.cuh"   // shared device helpers (tc_a_index, dot2, chunk_rmsnorm, swiglu16, PairParams); its kernel is not instantiated here
#include "w4_gemm_pair16.cuh"
#include "w4_gemm_pair16_loop.cuh"
#ifndef TG_P16_XREG_MIN_M
#define TG_P16_XREG_MIN_M 5  // activation rows from which the A operands are arranged in registers instead of staged through LDS
#endif
// Small launches of Bint4 weights (one layer per call): w4_gemm_pair16_kernel, 16 weight rows per workgroup, the whole k-slice
// of a wave requested up front.  Taken when the launch is too small for the persistent kernel (or its LDS plan does not fit)
// and the activations (m <= 16 rows) fit in LDS next to the table.
#if GEMV_TRACE
unsigned long long* g_p16_trace = nullptr;  // developer builds only (-DGEMV_TRACE=1): [slots][512 workgroups][8 stamps]
int g_p16_slots = 0, g_p16_launch = 0;
#endif
template <typename DT, int I, bool QMX>
int launch_pair16(const GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (QMX && !std::is_same<DT, BF16>::value) return TG_E_DTYPE;  // mx4 is bf16-only (TinyGemm_int4.cu:758)
  else {
#ifdef TG_DEV_MIN
  if constexpr (!(std::is_same<DT, BF16>::value && I == 4 && !QMX)) return TG_PAIR_NA;
#endif
  if (p.m > 16 || batch > 65535) return TG_PAIR_NA;
  // m = 1 and more than one round of workgroups (one per CU): the streaming kernel's split-K launches are faster there
  // (per hipGraph node, 6144 x 4096: 8.6 us against 10.2 us; 14336 x 4096: 13.8 against 18.3)
  // (not when a fused stage is asked for: only the pair-table kernels have them)
  if (p.m == 1 && !p.x_tc && !p.y_tc && !p.norm_w && !p.epilogue && (int64_t)((p.wrows + 15) / 16) * batch > 256 && (1 << p.gshift) >= 128) return TG_PAIR_NA;
  if (p.norm_w && (QMX || (int64_t)p.m * p.k > 32768)) return TG_PAIR_NA;  // the norm pass: one 32-k chunk per thread
  const int g = 1 << p.gshift;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  Pair16Params pp;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  pp.gch_mask = g / 32 - 1;
  pp.lds_x = 65536;
  const int64_t wgs = (int64_t)((p.wrows + 15) / 16) * batch;
  // one workgroup per CU may take the whole LDS; a launch of more than two rounds of workgroups should fit two per CU
  const unsigned lds_limit = (wgs <= 512 ? 160u : 80u) * 1024u;
  // activation rows that do not fit next to the table are staged one part of k at a time (whole groups per part)
  unsigned lds = 0;
  int phases = 1;
  // XREG (w4_gemm_pair16.cuh): no LDS for activations, one pass whatever m x k is; one workgroup per CU (two rounds at most)
  bool xreg = p.m >= TG_P16_XREG_MIN_M && !p.norm_w && wgs <= 512 && p.ksuper % nsg == 0;
  if (xreg && g == 32 && I == 4 && ((p.ksuper / nsg + 15) / 16) * nsg > 4) xreg = false;  // (see TG_P16: that instantiation spills)
  if (xreg) {
    pp.x_pitch = 0;
    pp.lds_xs = pp.lds_x;
    lds = 65536u;
  } else
  for (; phases <= (wgs <= 512 && !p.norm_w ? 8 : 1); phases *= 2) {  // (the fused norm needs a row's whole k in one part)
    if (p.ksuper % (phases * nsg) != 0 || p.ngroups % phases != 0) return TG_PAIR_NA;
    const int kp = p.k / phases;
    pp.x_pitch = kp * 2 + 16;
    pp.lds_xs = (pp.lds_x + p.m * pp.x_pitch + 16 + 15) & ~15;
    lds = (unsigned)pp.lds_xs + (QMX ? 0u : (unsigned)(p.ngroups / phases) * 64u);
    if (lds <= lds_limit) break;
  }
  if (lds > lds_limit) return TG_PAIR_NA;
  if (xreg) phases = 1;
  pp.phases = phases;
  pp.ksuper_p = p.ksuper / phases;
  pp.spw = ((pp.ksuper_p / nsg + 15) / 16) * nsg;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = p.norm_w; pp.norm_eps = p.norm_eps; pp.epilogue = p.epilogue;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  if (p.dry) return TG_PLAN_PAIR;
  const dim3 grid((unsigned)((p.wrows + 15) / 16), (unsigned)batch);
#if GEMV_TRACE
  pp.trace = (g_p16_trace && g_p16_slots > 0 && grid.x <= 512) ? g_p16_trace + (size_t)(g_p16_launch++ % g_p16_slots) * 512 * 8 : nullptr;
#endif
#define TG_P16K(CPG_, NORM_, XREG_, CH_) do { if (XREG_ && p.x_tc) TG_P16KX(CPG_, NORM_, XREG_, CH_, XREG_); else TG_P16KX(CPG_, NORM_, XREG_, CH_, false); } while (0)
#define TG_P16KX(CPG_, NORM_, XREG_, CH_, XTC_)                                         \
  do {                                                                                  \
    constexpr auto kern = w4_gemm_pair16_kernel<DT, I, QMX, CPG_, 1, NORM_, XREG_, CH_, XTC_>; \
    const int prc = prepare_lds_kernel<kern>();                                         \
    if (prc != 0) return prc;                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, pp);                            \
  } while (0)
#define TG_P16(CPG_)                                              \
  do {                                                            \
    if constexpr (!QMX) {                                         \
      if (p.norm_w) { TG_P16K(CPG_, true, false, 4); break; }     \
    }                                                             \
    if (xreg && pp.spw <= 4) { TG_P16K(CPG_, false, true, 4); break; }  /* the whole slice in one block */ \
    /* (groups of 32 at innerKTiles 4 with slices longer than a block: 84 ... 128 bytes of scratch -- not instantiated, the LDS path) */ \
    if constexpr (!(CPG_ == 1 && I == 4)) {                       \
      if (xreg) { TG_P16K(CPG_, false, true, 2); break; }         \
    }                                                             \
    TG_P16K(CPG_, false, false, 4);                               \
  } while (0)
  if constexpr (QMX) TG_P16(1);  // mx4: group = 32
  else if (g == 32) TG_P16(1);
  else if (g == 64) TG_P16(2);
  else if (g == 128) TG_P16(4);
  else TG_P16(8);
#undef TG_P16K
#undef TG_P16KX
#undef TG_P16
  return launch_status();
  }
}

// ONE layer per launch with more 16-row tiles than compute units (w4_gemm_pair16_loop.cuh): 5 ... 16 activation rows, k = 4096, innerKTiles 4,
// row-major operands, no fused stage; a workgroup owns up to 32 tiles.
#ifndef TG_P16_LOOP
#define TG_P16_LOOP 1
#endif
#ifndef TG_P16_LOOP_MAX_TILES
#define TG_P16_LOOP_MAX_TILES 8   // tiles per workgroup up to which this path takes the launch.  Per graph node at m = 16, rows 5120 / 8192 / 11008 / 12288 /
                                  // 14336 / 16384 / 20480 / 28672: 8.5 / 9.0 / 10.8 / 10.9 / 13.0 / 13.1 / 15.2 / 19.5 us here; w4_gemm_xr_kernel with one workgroup per
                                  // 64-row item: 12.3 / 12.4 / 12.6 / - / 13.2 / 13.7 / - / 19.4 (profiles/r05_ab_p16_loop.txt, r05_p16_loop_sweep.txt)
#endif
template <typename DT>
int launch_pair16_loop(const GemmParams& p, int64_t batch, hipStream_t st) {
  if (!TG_P16_LOOP || batch != 1 || p.m < TG_P16_XREG_MIN_M || p.m > 16 || p.k != 4096 || p.ksuper != 64 || (p.epilogue && (p.epilogue != TG_EPI_SWIGLU || p.bias || p.wrows % 16 != 0)) || p.x_tc || p.y_tc ||
      p.qtype == TG_Q_MX4)
    return TG_PAIR_NA;
  if (p.norm_w && p.gshift == 5) return TG_PAIR_NA;  // (groups of 32 with the fused norm: not instantiated, see the kernel)
  const int tiles = (p.wrows + 15) / 16;
  const int cus = p.dry ? 256 : cu_count();
  if (tiles <= cus) return TG_PAIR_NA;  // (one tile per workgroup: w4_gemm_pair16_kernel)
  const int per = (tiles + cus - 1) / cus;
  if (per > TG_P16_LOOP_MAX_TILES || per > 32) return TG_PAIR_NA;
  Pair16LoopParams pp;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y; pp.bias = p.bias; pp.bias_row_stride = p.bias_row_stride;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper; pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  pp.tbase = tiles / cus; pp.trem = tiles % cus;
  pp.epilogue = p.epilogue;
  pp.norm_w = p.norm_w; pp.norm_eps = p.norm_eps;
  pp.lds_red = 65536;
  pp.lds_lut = 65536 + 2 * 16384;
  pp.lds_nrm = pp.lds_lut + (p.qtype == TG_Q_ANY4_ROWWISE ? per * 544 : 0);
  const unsigned lds = (unsigned)pp.lds_nrm + 16u * 16u * 4u;
  if (p.dry) return TG_PLAN_PAIR;
  const int g = 1 << p.gshift;
#define TG_P16L(CPG_, NORM_)                                                  \
  do {                                                                        \
    constexpr auto kern = w4_gemm_pair16_loop_kernel<DT, CPG_, NORM_>;        \
    const int prc = prepare_lds_kernel<kern>();                               \
    if (prc != 0) return prc;                                                 \
    hipLaunchKernelGGL(kern, dim3((unsigned)cus), dim3(1024), lds, st, pp);   \
  } while (0)
  if (p.norm_w) {
    if (g == 64) TG_P16L(2, true);
    else if (g == 128) TG_P16L(4, true);
    else TG_P16L(8, true);
  } else {
    if (g == 32) TG_P16L(1, false);
    else if (g == 64) TG_P16L(2, false);
    else if (g == 128) TG_P16L(4, false);
    else TG_P16L(8, false);
  }
#undef TG_P16L
  return launch_status();
}

template <typename DT, int I>
int p16_q(bool qmx, const GemmParams& p, int64_t batch, hipStream_t st) {
  return qmx ? launch_pair16<DT, I, true>(p, batch, st) : launch_pair16<DT, I, false>(p, batch, st);
}
template <typename DT>
int p16_i(int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st) {
  // (innerKTiles 8: every instantiation compiled with 76 ... 308 bytes of scratch per lane at the 128-VGPR budget of a 1024-thread
  //  workgroup -- not instantiated; those layers take the streaming kernels)
  return I == 2 ? p16_q<DT, 2>(qmx, p, batch, st) : I == 4 ? p16_q<DT, 4>(qmx, p, batch, st) : (int)TG_PAIR_NA;
}
}  // namespace
#if GEMV_TRACE
extern "C" TG_API void tg_dev_p16_trace(unsigned long long* buf, int slots) {
  g_p16_trace = buf;
  g_p16_slots = slots;
  g_p16_launch = 0;
}
#endif
int tgx::pair16_loop(int dt, int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st) {
  if (I != 4 || qmx) return TG_PAIR_NA;
  return dt == TG_BF16 ? launch_pair16_loop<BF16>(p, batch, st) : launch_pair16_loop<F16>(p, batch, st);
}
int tgx::pair16(int dt, int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? p16_i<BF16>(I, qmx, p, batch, st) : p16_i<F16>(I, qmx, p, batch, st);
}
