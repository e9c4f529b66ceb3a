// tg_xr.hip -- launch path of w4_gemm_xr_kernel (stacked launches, activations resident in registers); see tg_common.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm_pair.cuh"   // shared device helpers; its kernel is not instantiated here
#include "w4_gemm_xr.cuh"
// Bint4 weights, stacked launches, TG_XR_MIN_M ... 16 activation rows, k = 4096: w4_gemm_xr_kernel (one 8-wave workgroup per CU, the
// activations of a wave's k-slice resident in its registers, 64-row work items, two tables).  No workspace, no pre-pass.
#ifndef TG_XR_WV16
#define TG_XR_WV16 0  // 1 (developer builds): k = 4096, not mx4, on sixteen k-slices per workgroup (w4_gemm_xr.cuh, WV) -- four waves per
                      // SIMD instead of two, measured EQUAL at m = 8 (71.5 vs 71.3 %) and slower at m = 16 (67.1 vs 69.0 %): not shipped
#endif
#ifndef TG_XR_R16
#define TG_XR_R16 2   // ring depth of the sixteen-slice variant (a slice is four super-tiles)
#endif
#ifndef TG_XR_WV4
#define TG_XR_WV4 0   // 1 (developer builds): k = 4096 with at most 8 rows on two 4-slice workgroups per CU, packed rows, one table each (w4_gemm_xr.cuh,
                      // WV = 4) -- the tail of one workgroup under the main loop of the other.  Measured SLOWER, same box: m = 8 72.3 vs 73.3 %, m = 4 72.7 vs
                      // 75.4 %, m = 2 75.2 vs 77.1 % (profiles/r05_ab_xr_wv4.txt): it is not the stall of the tail that costs, not shipped
#endif
#ifndef TG_XR_WINDOWS_14336
#define TG_XR_WINDOWS_14336 16, 16, 24   // k = 14336 at 9 ... 16 rows as k-windows of 256 x these chunk counts (launch_pair_xr_windows)
#endif
#ifndef TG_XR_PK_K4096
#define TG_XR_PK_K4096 0  // 1 (developer builds): k = 4096 with at most 8 rows on the packed-rows variant too (32 instead of 64 activation registers)
#endif
// a k-window [k0, k0 + 256 NCH) of a longer contraction: the kernel runs on the window's pointers (the packed layout keeps the whole
// matrix's tile stride and the activations the whole row pitch) and leaves its UNROUNDED f32 sums in y32 (launch_pair_xr_windows)
struct XrWindow {
  int64_t w_off, x_off, q_off;  // byte offsets of the window in w / x / qinfo
  int32_t ngroups;              // quantisation groups inside the window
  char* y32;
  int64_t stride_y32;
};
template <typename DT, int I, bool QMX, int NCH, int WV = 8, bool PK = false>
int launch_pair_xr_n(GemmParams& p, int64_t batch, hipStream_t st, const XrWindow* win = nullptr) {
  if constexpr (I != 4 || (QMX && (NCH != 16 || !std::is_same<DT, BF16>::value))) return TG_PAIR_NA;  // (mx4: bf16, k = 4096)
  else {
#ifdef TG_DEV_MIN
  if constexpr (!std::is_same<DT, BF16>::value) return TG_PAIR_NA;
  else {
#endif
  // (one row: only when the caller asks for the MATRIX-CORE contraction, TG_NUM_FAST_MFMA -- this kernel's 16x16x32 MFMAs do it at
  //  79.4 % where the 32x32x16 ones of w4_gemm_pair_kernel reach 76.3 %; the default m = 1 contraction is v_dot2 there: 81.9 %, same box)
  if (p.m > (PK ? 8 : 16) || p.m < (p.numerics == TG_NUM_FAST_MFMA ? 1 : TG_XR_MIN_M) || p.norm_w || p.epilogue) return TG_PAIR_NA;
  if (p.ksuper * 16 * I != p.k || p.wrows % 64 != 0 || p.ntiles * 8 != p.wrows) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int cpg = g / 32 < NCH ? g / 32 : NCH;  // 32-k chunks per group inside a wave's slice
#ifdef TG_DEV_MIN
  if (cpg != (QMX ? 1 : 4)) return TG_PAIR_NA;
#endif
  if (QMX ? cpg != 1 : (cpg != 1 && cpg != 2 && cpg != 4 && cpg != 8)) return TG_PAIR_NA;  // g = 32, 64, 128, 256; mx4: g = 32
  if (QMX && p.ngroups % 16 != 0) return TG_PAIR_NA;  // 16-byte exponent blocks
  if (NCH > 24 && cpg == 2 && !PK) return TG_PAIR_NA;   // (k = 8192, g = 64, unpacked: that instantiation spills four registers)
  XrParams xp;
  xp.w = p.w; xp.qinfo = p.qinfo; xp.lut = p.lut; xp.y = p.y;
  xp.m = p.m; xp.wrows = p.wrows; xp.k = p.k; xp.ntiles = p.ntiles; xp.ksuper = p.ksuper;
  xp.gshift = p.gshift; xp.ngroups = p.ngroups; xp.qtype = p.qtype;
  xp.rblocks = (p.wrows + 63) / 64;
  const int64_t items = (int64_t)xp.rblocks * batch;
#ifndef TG_XR_MIN_ITEMS_PER_WG
#define TG_XR_MIN_ITEMS_PER_WG 2  // work items per workgroup from which this kernel takes a launch (developer builds: 1 for one large layer per launch)
#endif
  // Two items per workgroup at least -- except ONE layer per launch at k = 4096 (what a module's forward issues at that batch; up to 8
  // rows only what w4_gemv_kernel declined: groups of 32 / 64): one workgroup per 64-row item.  Per hipGraph node, same box
  // (profiles/r05_ab_xr_single_crossover.txt; the first problem's activations staged through LDS, r05_ab_xr_xlds.txt):
  //     rows x 4096, m = 16    4096   5120   8192  11008  14336  16384  28672      m = 8, g = 64:  8192  16384   m = 6: 28672
  //     pair16 / stream         8.4   14.9   15.5   38.6   39.2   39.2   41.1                      13.6   17.6          29.7
  //     here                   12.8   13.0   13.7   14.0   14.7   14.8   22.0                      14.0   15.4          22.2
  // (an item is ~6.5 us of one CU's time whatever the launch: below 80 items the 16-row workgroups of w4_gemm_pair16_kernel win)
#ifndef TG_XR_SINGLE_MIN_ITEMS
#define TG_XR_SINGLE_MIN_ITEMS 80
#endif
#ifndef TG_XR_SINGLE_MIN_ITEMS_M8
#define TG_XR_SINGLE_MIN_ITEMS_M8 256
#endif
  // (... and up to eight 16-row tiles per CU -- 32768 rows on 256 CUs -- w4_gemm_pair16_loop_kernel is faster or equal where it applies:
  //  row-major operands, not mx4: 5120 / 8192 / 12288 / 16384 / 28672 rows at m = 16 8.5 / 9.0 / 10.9 / 13.1 / 19.5 us against 12.3 / 12.4 / 12.9 /
  //  13.7 / 19.4 here; this route keeps fragment-order operands, mx4 and longer layers)
  const bool p16_loop = !p.x_tc && !p.y_tc && !QMX && I == 4 && p.m >= 5 && (int64_t)((p.wrows + 15) / 16) <= 8 * (int64_t)(p.dry ? 256 : cu_count());
  const bool single = batch == 1 && NCH == 16 && !PK && WV == 8 && !win && !p16_loop &&
                      ((p.m >= 9 && items >= TG_XR_SINGLE_MIN_ITEMS) || (p.m >= 5 && items >= TG_XR_SINGLE_MIN_ITEMS_M8));
  if (items > INT32_MAX) return TG_PAIR_NA;
  if (!single && items < TG_XR_MIN_ITEMS_PER_WG * 256 * (WV == 4 ? 2 : 1)) return TG_PAIR_NA;
  xp.items = (int32_t)items;
  xp.lds_xs = WV == 4 ? 65536 : 2 * 65536;
  // two tables (WV = 4: one, and the 8 KiB hand-over region behind the sums), the activation sums (mx4: the partial sums only)
  const unsigned lds = QMX ? 32768u : (unsigned)xp.lds_xs + (unsigned)(win ? win->ngroups : p.ngroups) * 64u + (WV == 4 ? 8192u : 0u);
  if (lds > 160u * 1024u) return TG_PAIR_NA;
  xp.x = p.x; xp.stride_x = p.stride_x; xp.x_tc = p.x_tc;  // (no pre-pass, no workspace: the kernel arranges the activations itself)
  p.ws_need = 0;
  xp.stride_w = p.stride_w; xp.stride_qinfo = p.stride_qinfo; xp.stride_lut = p.stride_lut; xp.stride_y = p.stride_y;
  xp.bias = p.bias; xp.stride_bias = p.stride_bias; xp.bias_row_stride = p.bias_row_stride;
  xp.y_tc = p.y_tc; xp.y_tiles = (p.wrows + 15) / 16; xp.dry = p.dry;
  xp.y_f32 = 0;
  if (win) {
    if (p.x_tc || p.y_tc) return TG_PAIR_NA;
    xp.w += win->w_off; xp.x += win->x_off; xp.qinfo += win->q_off; xp.ngroups = win->ngroups;
    xp.y = win->y32; xp.stride_y = win->stride_y32; xp.y_f32 = 1; xp.bias = nullptr;
  }
  if (p.dry) return TG_PLAN_PAIR_XR;
  unsigned wgs = (unsigned)cu_count() * (WV == 4 ? 2u : 1u);  // one 8-wave (two 4-wave) workgroup(s) per compute unit, whatever the part has
  if (single) wgs = items < (int64_t)wgs ? (unsigned)items : wgs;
  else if (items < TG_XR_MIN_ITEMS_PER_WG * (int64_t)wgs) return TG_PAIR_NA;
#define TG_XR_LAUNCH(CPG_)                                                  \
  do {                                                                      \
    constexpr auto kern = w4_gemm_xr_kernel<DT, I, NCH, CPG_, (WV == 16 ? TG_XR_R16 : ((NCH > 24 && !PK) || NCH > 32) ? TG_XR_R8K : TG_XR_R), false, WV, PK>; \
    const int prc = prepare_lds_kernel<kern>();                             \
    if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA; /* (a part with less LDS: the older kernels take over) */ \
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(WV * 64), lds, st, xp);        \
  } while (0)
  if constexpr (QMX) {
    constexpr auto kern = w4_gemm_xr_kernel<DT, I, NCH, 1, TG_XR_RMX, true>;
    const int prc = prepare_lds_kernel<kern>();
    if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, xp);
  } else {
#ifdef TG_DEV_MIN
  TG_XR_LAUNCH(4);
#else
  if (cpg == 1) TG_XR_LAUNCH(1);
  else if (cpg == 2) TG_XR_LAUNCH(2);
  else if (cpg == 4) TG_XR_LAUNCH(4);
  else if constexpr (NCH % 8 == 0) TG_XR_LAUNCH(8);
  else return TG_PAIR_NA;  // (a wave's slice must hold whole groups)
#endif
  }
#undef TG_XR_LAUNCH
  return launch_status();
#ifdef TG_DEV_MIN
  }
#endif
  }
}

// y[b][a][row] = RNE16(sum over the k-windows, in window order) (+ bias, rounded again as everywhere): two adjacent rows per thread
template <typename DT, int NP>
__global__ void __launch_bounds__(256) xr_window_sum_kernel(const float* __restrict__ parts, int64_t part_stride, char* __restrict__ y, int64_t stride_y,
                                                            const char* __restrict__ bias, int64_t stride_bias, int64_t bias_row_stride, int m, int wrows,
                                                            int64_t pairs_per_problem) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= pairs_per_problem) return;
  const int64_t b = blockIdx.y;
  const int a = (int)(i / (wrows / 2)), row = (int)(i % (wrows / 2)) * 2;
  const float* src = parts + (b * m + a) * (int64_t)wrows + row;
  f32x2 acc = *reinterpret_cast<const f32x2*>(src);
#pragma unroll
  for (int p = 1; p < NP; ++p) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(src + p * part_stride);
    acc[0] += v[0];
    acc[1] += v[1];
  }
  uint16_t oa = DT::from_f32(acc[0]), ob = DT::from_f32(acc[1]);
  if (bias) {
    const uint32_t bv = *reinterpret_cast<const uint32_t*>(bias + b * stride_bias + ((int64_t)a * bias_row_stride + row) * 2);
    oa = DT::from_f32(DT::lo_f32(oa) + DT::lo_f32(bv));
    ob = DT::from_f32(DT::lo_f32(ob) + DT::hi_f32(bv));
  }
  *reinterpret_cast<uint32_t*>(y + b * stride_y + ((int64_t)a * wrows + row) * 2) = (uint32_t)oa | ((uint32_t)ob << 16);
}

// k = a sum of windows of 256 NCH (TinyGemmImpl.cuh:132-217 takes any k % 32 == 0; Llama-3-8B's down-projection: k = 14336) with
// 9 ... 16 activation rows: the register file holds a wave's activations for a window, not for the whole k.  One launch of the kernel
// per window over ALL the work items (the activations stay resident across the items of a problem, which is the point of this kernel),
// f32 partial sums in the caller's workspace (NP x batch x m x wrows x 4 bytes: 3.4 % of the weight bytes at m = 16), one small
// kernel adds the windows in order.  The 16x16x32 workspace kernel this replaces re-reads the activations per work item: 45 %.
template <typename DT, int I, int... NCHS>
int launch_pair_xr_windows(GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (I != 4) return TG_PAIR_NA;
  else {
  constexpr int NP = sizeof...(NCHS);
  constexpr int nch[NP] = {NCHS...};
  int ktot = 0;
  const int g = 1 << p.gshift;
  for (int i = 0; i < NP; ++i) {
    if ((256 * nch[i]) % g != 0) return TG_PAIR_NA;
    ktot += 256 * nch[i];
  }
  if (p.k != ktot || p.x_tc || p.y_tc || p.m < 9 || p.m > 16 || p.wrows % 64 != 0 || p.norm_w || p.epilogue) return TG_PAIR_NA;
  const int64_t part_elems = batch * p.m * (int64_t)p.wrows;
  const int64_t need = NP * part_elems * 4;
  if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;  // (the caller did not bring the workspace: the older kernels)
  XrWindow win;
  win.stride_y32 = (int64_t)p.m * p.wrows * 4;
  int k0 = 0, part = 0, rc_all = 0;
  auto one = [&](auto NCH_) {
    constexpr int NCH = decltype(NCH_)::value;
    if (rc_all != 0) return;
    const int kw = 256 * NCH;
    win.ngroups = kw / g;
    win.w_off = (int64_t)(k0 / (16 * I)) * (64 * I);
    win.x_off = (int64_t)k0 * 2;
    win.q_off = (int64_t)(k0 / g) * p.wrows * 4;
    win.y32 = p.ws + part * part_elems * 4;
    const int rc = launch_pair_xr_n<DT, I, false, NCH>(p, batch, st, &win);
    if (rc != 0) rc_all = rc;  // (TG_PAIR_NA, a TG_E_* code, a hipError_t, or TG_PLAN_PAIR_XR from a dry run)
    k0 += kw;
    ++part;
  };
  // every window must have a kernel BEFORE the first one is launched (a dry pass: validation only)
  const auto dry0 = p.dry;
  p.dry = 1;
  int planned = 0;
  auto check = [&](auto NCH_) {
    XrWindow w0 = win;
    w0.ngroups = 256 * decltype(NCH_)::value / g;
    w0.w_off = w0.x_off = w0.q_off = 0;
    w0.y32 = p.ws;
    planned += launch_pair_xr_n<DT, I, false, decltype(NCH_)::value>(p, batch, st, &w0) == TG_PLAN_PAIR_XR;
  };
  (check(std::integral_constant<int, NCHS>{}), ...);
  p.dry = dry0;
  if (planned != NP) { p.ws_need = 0; return TG_PAIR_NA; }
  if (p.dry) { p.ws_need = need; return TG_PLAN_PAIR_XR; }
  (one(std::integral_constant<int, NCHS>{}), ...);
  if (rc_all != 0) {
    p.ws_need = 0;
    return rc_all;
  }
  p.ws_need = need;
  const int64_t pairs = (int64_t)p.m * (p.wrows / 2);
  hipLaunchKernelGGL((xr_window_sum_kernel<DT, NP>), dim3((unsigned)((pairs + 255) / 256), (unsigned)batch), dim3(256), 0, st,
                     reinterpret_cast<const float*>(p.ws), part_elems, p.y, p.stride_y, p.bias, p.stride_bias, p.bias_row_stride, p.m, p.wrows, pairs);
  return launch_status();
  }
}

template <typename DT, int I, bool QMX>
int launch_pair_xr(GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (!QMX && TG_XR_WV16) {
    if (p.k == 4096 && (1 << p.gshift) <= 256) return launch_pair_xr_n<DT, I, QMX, 8, 16>(p, batch, st);
  }
  if constexpr (!QMX && TG_XR_WV4) {
    if (p.k == 4096 && p.m <= 8) {
      const int rc = launch_pair_xr_n<DT, I, QMX, 32, 4, true>(p, batch, st);
      if (rc != TG_PAIR_NA) return rc;  // (fewer than four items per CU: the 8-wave kernel below)
    }
  }
  if constexpr (!QMX && TG_XR_PK_K4096) {
    if (p.k == 4096 && p.m <= 8) return launch_pair_xr_n<DT, I, QMX, 16, 8, true>(p, batch, st);
  }
  if (p.k == 4096) return launch_pair_xr_n<DT, I, QMX, 16>(p, batch, st);
  // k = 8192, 9 ... 16 rows: 128 registers of activations per lane leave room for two super-tiles in flight only -- faster than
  // the 16x16x32 workspace kernel it replaces (8192^2, m = 16: 62 vs 47-51 %).  Up to 8 rows: two chunks per register set (PK),
  // 64 registers, four super-tiles in flight like k = 4096
  if constexpr (!QMX) {
    if (p.k == 8192 && p.m <= 8) return launch_pair_xr_n<DT, I, QMX, 32, 8, true>(p, batch, st);
  }
#ifndef TG_XR_WINDOWS_8192
#define TG_XR_WINDOWS_8192 1   // 1: k = 8192 at 9 ... 16 rows as two k-windows of 4096 (the ring stays four deep) when the caller brings the workspace
#endif
  if constexpr (!QMX && TG_XR_WINDOWS_8192) {
    if (p.k == 8192 && p.m >= 9) {
      const int rc = launch_pair_xr_windows<DT, I, 16, 16>(p, batch, st);
      if (rc != TG_PAIR_NA) return rc;
    }
  }
  if (p.k == 8192 && p.m >= 9) return launch_pair_xr_n<DT, I, QMX, 32>(p, batch, st);
  // k = 14336 (Llama-3-8B's down-projection) with at most 8 rows: 56 chunks per slice, packed: 112 activation registers, ring of two
  if constexpr (!QMX) {
    // (same box, 4096 x 14336 against the workspace variant of w4_gemm_pair_kernel: m = 8 67.9 -> 70.0 %, but m = 4 72.9 -> 69.1 and
    //  m = 2 76.5 -> 72.3, m = 6 71.1 -> 68.8 -- profiles/r05_ab_xr_k14336.txt: eight rows only)
    if (p.k == 14336 && p.m == 8) return launch_pair_xr_n<DT, I, QMX, 56, 8, true>(p, batch, st);
    if (p.k == 14336 && p.m >= 9) return launch_pair_xr_windows<DT, I, TG_XR_WINDOWS_14336>(p, batch, st);
  }
  return TG_PAIR_NA;
}
template <typename DT>
int xr_i(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  if (I != 4) return TG_PAIR_NA;
  return qmx ? launch_pair_xr<DT, 4, true>(p, batch, st) : launch_pair_xr<DT, 4, false>(p, batch, st);
}
}  // namespace
int tgx::pair_xr(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? xr_i<BF16>(I, qmx, p, batch, st) : xr_i<F16>(I, qmx, p, batch, st);
}
