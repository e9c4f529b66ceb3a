// tg_gemv.hip -- launch path of w4_gemv_kernel (one layer per launch, 1 ... 4 activation rows: the GEMMs of a batch-1 decode step
// and of Any4Linear.forward / Int4Linear.forward at batch 1); see tg_common.cuh and w4_gemv.cuh
#include "tg_common.cuh"
namespace {
#include "w4_gemm_pair.cuh"   // shared device helpers (dot2, chunk_rmsnorm, swiglu16); its kernel is not instantiated here
#include "w4_gemv.cuh"

#ifndef TG_GEMV_MF_MIN_M
#define TG_GEMV_MF_MIN_M 1  // activation rows from which the contraction runs on the matrix core where its conditions hold (k <= 4096, groups
                            // of 128 / 256).  Same box, per graph node, v_dot2 -> matrix core: m = 1: 4096^2 5.10 -> 4.92 us, 28672 x 4096
                            // 15.4 -> 14.85, the decode step 622 -> 630 tokens/s; m = 2: 5.8 -> 5.1, 19.3 -> 14.9, decode at batch 2
                            // 1054 -> 1173 tokens/s (the kernel's bound was vector-ALU issue; profiles/r04_ab_gemv_mf_min_m.txt)
#endif
#ifndef TG_GEMV_MAX_TILES
#define TG_GEMV_MAX_TILES 16384  // 8-row tiles per launch up to which this kernel takes single-problem launches (131072 rows)
#endif

template <typename DT, int M, int GPS, int D, bool NORM, bool MF = false>
int go(const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  constexpr auto kern = w4_gemv_kernel<DT, M, GPS, D, NORM, MF>;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc == TG_E_INTERNAL ? prc : TG_PAIR_NA;  // (a part with less LDS: the older kernels take over)
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, gp);
  return launch_status();
}
template <typename DT, int M, int GPS, int D>
int go_n(bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  return norm ? go<DT, M, GPS, D, true>(gp, grid, lds, st) : go<DT, M, GPS, D, false>(gp, grid, lds, st);
}
template <typename DT, int M, int GPS>
int go_d(int d, bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  return d == 4 ? go_n<DT, M, GPS, 4>(norm, gp, grid, lds, st) : go_n<DT, M, GPS, 8>(norm, gp, grid, lds, st);
}
template <typename DT, int M>
int go_g(int gps, int d, bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  return gps == 1 ? go_d<DT, M, 1>(d, norm, gp, grid, lds, st) : go_d<DT, M, 2>(d, norm, gp, grid, lds, st);
}
// the matrix-core contraction (w4_gemv.cuh, MF): 3 ... 8 rows, one group per step, ring of four (k <= 4096: four steps per pass)
template <typename DT, int M>
int go_mf(int d, bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  if (d == 8) {  // (k > 4096: slices of more than four steps; up to four rows -- the activation block has to fit next to the table)
    if constexpr (M <= 4) return norm ? go<DT, M, 1, 8, true, true>(gp, grid, lds, st) : go<DT, M, 1, 8, false, true>(gp, grid, lds, st);
    else return TG_PAIR_NA;
  }
  return norm ? go<DT, M, 1, 4, true, true>(gp, grid, lds, st) : go<DT, M, 1, 4, false, true>(gp, grid, lds, st);
}
template <typename DT>
int go_mf_m(int m, int d, bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  switch (m) {
#if TG_GEMV_MF_MIN_M <= 1
    case 1: return go_mf<DT, 1>(d, norm, gp, grid, lds, st);
#endif
#if TG_GEMV_MF_MIN_M <= 2
    case 2: return go_mf<DT, 2>(d, norm, gp, grid, lds, st);
#endif
    case 3: return go_mf<DT, 3>(d, norm, gp, grid, lds, st);
    case 4: return go_mf<DT, 4>(d, norm, gp, grid, lds, st);
    case 5: return go_mf<DT, 5>(d, norm, gp, grid, lds, st);
    case 6: return go_mf<DT, 6>(d, norm, gp, grid, lds, st);
    case 7: return go_mf<DT, 7>(d, norm, gp, grid, lds, st);
    default: return go_mf<DT, 8>(d, norm, gp, grid, lds, st);
  }
}
template <typename DT>
int go_m(int m, int gps, int d, bool norm, const GemvParams& gp, dim3 grid, unsigned lds, hipStream_t st) {
  switch (m) {
    case 1: return go_g<DT, 1>(gps, d, norm, gp, grid, lds, st);
    case 2: return go_g<DT, 2>(gps, d, norm, gp, grid, lds, st);
    case 3: return go_g<DT, 3>(gps, d, norm, gp, grid, lds, st);
    default: return go_g<DT, 4>(gps, d, norm, gp, grid, lds, st);
  }
}
#if GEMV_TRACE
unsigned long long* g_trace = nullptr;  // developer builds only (-DGEMV_TRACE=1): [slots][512 workgroups][8 stamps]
int g_trace_slots = 0, g_trace_launch = 0;
#endif
}  // namespace

#if GEMV_TRACE
extern "C" TG_API void tg_dev_gemv_trace(unsigned long long* buf, int slots) {
  g_trace = buf;
  g_trace_slots = slots;
  g_trace_launch = 0;
}
#endif

int tgx::gemv(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  const int g0 = 1 << p.gshift;
  // matrix-core contraction: 16-row passes of two super-tiles per step inside ONE group, a piece per thread in the staging
#ifndef TG_GEMV_MF_MAX_M
#define TG_GEMV_MF_MAX_M 8  // (developer A/B against w4_gemm_pair16_kernel / the loop kernel at 5 ... 8 rows)
#endif
  const bool mf0 = p.m >= TG_GEMV_MF_MIN_M && p.m <= TG_GEMV_MF_MAX_M && (g0 == 128 || g0 == 256) && (p.k <= 4096 || p.m <= 4) && p.ksuper % 2 == 0;
  if (I != 4 || qmx || (p.m > 4 && !mf0) || p.x_tc || p.y_tc || batch != 1) return TG_PAIR_NA;
  if (p.ksuper * 64 != p.k || p.ntiles * 8 != p.wrows || p.ntiles > TG_GEMV_MAX_TILES) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int gps = g == 32 ? 2 : 1;
  GemvParams gp;
  gp.x = p.x; gp.w = p.w; gp.qinfo = p.qinfo; gp.lut = p.lut; gp.y = p.y; gp.bias = p.bias; gp.norm_w = p.norm_w;
  gp.stride_x = p.stride_x; gp.stride_w = p.stride_w; gp.stride_qinfo = p.stride_qinfo; gp.stride_lut = p.stride_lut;
  gp.stride_y = p.stride_y; gp.stride_bias = p.stride_bias; gp.bias_row_stride = p.bias_row_stride;
  gp.m = p.m; gp.wrows = p.wrows; gp.k = p.k; gp.ntiles = p.ntiles; gp.ksuper = p.ksuper; gp.qtype = p.qtype;
  gp.sg_shift = g <= 64 ? 0 : g == 128 ? 1 : 2;
  gp.norm_eps = p.norm_eps; gp.epilogue = p.epilogue; gp.trace = nullptr;
#ifndef TG_GEMV_CM
#define TG_GEMV_CM 1
#endif
  // (per graph node, same box: 4096^2 m = 8 6.74 -> 6.50 us, 6144 rows 9.2 -> 8.7, m = 6 + 1.6 %, m = 5 - 2 %: from six rows on;
  //  profiles/r05_ab_gemv_chunk_staging.txt)
  gp.cm = (TG_GEMV_CM && mf0 && p.m >= 6 && (p.k == 4096 || p.k == 2048)) ? 1 : 0;
  gp.unit = p.epilogue == TG_EPI_SWIGLU ? 2 : 1;
  if (p.ntiles % gp.unit != 0) return TG_PAIR_NA;
  const int units = p.ntiles / gp.unit;
  const int cus = p.dry ? 256 : cu_count();
  // one workgroup per CU (two per CU were measured on gate_up of Llama-3-8B: the second workgroup of a CU trails the first by 3 us
  // through its prologue and the launch ends no earlier -- the CU's instruction issue, not latency, bounds the main loop)
  // ... except for the longest layers: a dependent graph node that streams 64 MiB takes 14.3 us from one workgroup per CU and 13.1 us
  // from two (tools/ubench/graph_chain.hip, round 5: 32 MiB equal, 8 MiB slower) -- more requests in flight per CU, and with the
  // contraction on the matrix core (MF) the vector ALU no longer bounds the main loop
#ifndef TG_GEMV_WG2_MIN_BYTES
#define TG_GEMV_WG2_MIN_BYTES (48ll << 20)
#endif
  // (both workgroups of a CU must fit its 160 KiB of LDS together: the formula of `lds` below for the halved ranges)
  auto lds_two_per_cu = [&]() -> int64_t {
    const int tpw2 = ((units + 2 * cus - 1) / (2 * cus)) * gp.unit;
    const int64_t xs = (int64_t)p.m * (p.k / 4), xsmf = mf0 ? (int64_t)(p.k / 128) * 64 : 0;
    return 65536 + (p.qtype == TG_Q_ANY4_ROWWISE ? tpw2 * 8 * 32 : 0) + (int64_t)p.m * (p.k * 2 + (p.m > 1 ? 16 : 0)) + (xsmf > xs ? xsmf : xs) +
           2 * 8 * p.m * 32 * 4 + 8 * p.m * 4;
  };
  // (... and the halved ranges must not pad more 16-row passes than the whole ones: 28672 rows as 7-tile ranges run 4 passes where
  //  14-tile ranges run 7 -- 15.3 -> 16.4 us; the same layer with SwiGLU tile pairs, 8- and 6-tile ranges: the decode step -2.2 %)
  auto padded_tiles = [&](int n_wgs) -> int64_t {
    const int ub = units / n_wgs, ur = units % n_wgs;
    auto pad2 = [&](int u) -> int64_t { return (int64_t)((u * gp.unit + 1) / 2) * 2; };
    return ur * pad2(ub + 1) + (int64_t)(n_wgs - ur) * pad2(ub);
  };
#ifndef TG_GEMV_WG2_MIN_UNITS_PER_CU
#define TG_GEMV_WG2_MIN_UNITS_PER_CU 4
#endif
  const bool two_per_cu = (int64_t)p.wrows * p.k / 2 >= TG_GEMV_WG2_MIN_BYTES && units >= TG_GEMV_WG2_MIN_UNITS_PER_CU * cus && lds_two_per_cu() <= 80 * 1024 &&
                          padded_tiles(2 * cus) <= padded_tiles(cus);
  const int wgs = units < cus ? units : (two_per_cu ? 2 * cus : cus);
  gp.ubase = units / wgs;
  gp.urem = units % wgs;
  gp.uextra = 1;
  // Two workgroups per CU: the workgroups 0 ... cus - 1 are the first on their CU and through their prologue ~2 us before the second
  // ones (dev/gemv_trace.py).  TG_GEMV_WG2_EXTRA = E (odd) deals them E units more than the second ones instead of one (units a
  // multiple of the CU count: Llama-3-8B's gate_up, 7 SwiGLU units per CU as 4 + 3)
  // Same box, the decode step: 4 + 3 (E = 1) 1.523-1.545 ms, 5 + 2 (E = 3) **1.490-1.498**, 6 + 1 (E = 5) 1.542-1.550
  // (profiles/r05_ab_gemv_wg2_extra.txt)
#ifndef TG_GEMV_WG2_EXTRA
#define TG_GEMV_WG2_EXTRA 3
#endif
  if (TG_GEMV_WG2_EXTRA > 1 && two_per_cu && wgs == 2 * cus && units % cus == 0 && (units / cus) % 2 == 1 && units / cus > TG_GEMV_WG2_EXTRA) {
    gp.ubase = (units / cus - TG_GEMV_WG2_EXTRA) / 2;
    gp.urem = cus;
    gp.uextra = TG_GEMV_WG2_EXTRA;
  }
  gp.uh = 0;
#ifndef TG_GEMV_WG2_HALF
#define TG_GEMV_WG2_HALF 0  // (developer A/B) with E = 1: this many of the CUs run 5 + 2 instead of 4 + 3
#endif
  if (TG_GEMV_WG2_HALF > 0 && gp.uextra == 1 && two_per_cu && wgs == 2 * cus && gp.urem == cus && gp.ubase >= 2) gp.uh = TG_GEMV_WG2_HALF < cus ? TG_GEMV_WG2_HALF : cus;
  const int tpw = (gp.ubase + (gp.urem ? gp.uextra : 0) + (gp.uh ? 1 : 0)) * gp.unit;  // tiles of the largest range
  // Ranges of THREE tiles at one activation row (q/k/v of Llama-3-8B: 6144 rows over 256 CUs): two 16-row passes carry a padding
  // tile (4 tile slots for 3 tiles); three 8-row passes of the v_dot2 contraction would stream exactly the range -- measured SLOWER
  // (6144 x 4096 per graph node 6.8 -> 7.2 us, the decode step unchanged; profiles/r05_ab_gemv_odd_p8.txt): developer knob only
#ifndef TG_GEMV_ODD_P8
#define TG_GEMV_ODD_P8 0
#endif
  const bool odd3 = TG_GEMV_ODD_P8 && p.m == 1 && tpw == 3 && gp.unit == 1 && p.ksuper % 4 == 0;
  const bool mf = mf0 && !odd3;
  // rows per pass of ranges longer than two tiles: 16 (two super-tiles of k per ring step), not 32 -- a range is rarely a multiple of
  // four tiles (Llama-3-8B: gate_up 14, q/k/v 3) and the padding tiles of its last pass cost what real ones do.  Same box, per graph
  // node, 32 -> 16: 28672 x 4096 17.6 -> 16.2-16.6 us, 8192 rows 7.6 -> 6.4, 10240 11.2 -> 8.9, 12288 10.8 -> 8.4, 32768 (a multiple
  // of four tiles) 16.5 -> 16.7; 8-row passes 22.5 us at 28672 rows; the decode step 1.667 -> 1.600 ms (profiles/r04_ab_gemv_pass_rows.txt)
#ifndef TG_GEMV_P_BIG
#define TG_GEMV_P_BIG 16
#endif
  gp.P = mf ? 16 : (tpw <= 1 || odd3) ? 8 : tpw <= 2 ? 16 : TG_GEMV_P_BIG;
  // a step covers SS = 32 / P consecutive super-tiles: they must all lie inside the matrix (the kernel's addressing has no per-lane
  // clamp), so k = 64 x odd runs 32-row passes whatever the range, k = 128 x odd at least 16-row passes
  if (p.ksuper % 2 != 0) gp.P = 32;
  else if (p.ksuper % 4 != 0 && gp.P < 16) gp.P = 16;
  gp.p_shift = gp.P == 8 ? 3 : gp.P == 16 ? 4 : 5;
  const int tpp = gp.P / 8;
  const int passes = (tpw + tpp - 1) / tpp;
  const int SS = 32 / gp.P;
  gp.spw = ((p.ksuper + 7) / 8 + SS - 1) / SS * SS;  // a wave's slice: whole steps
  gp.spp = gp.spw / SS;
  const int d = gp.spp <= 4 ? 4 : 8;  // ring depth: a pass occupies whole rounds of D slots
  gp.rounds = (gp.spp + d - 1) / d;
  gp.xcd4 = (gp.urem == 0 && wgs % 32 == 0) ? 1 : 0;
  if (p.k > 16384) return TG_PAIR_NA;  // a thread keeps its pieces of the activation block in registers: four rounds of 512 per row
  gp.lds_lut = 65536;
  const bool stage_lut = p.qtype == TG_Q_ANY4_ROWWISE && passes > 1;
  gp.x_pitch = p.k * 2 + (p.m > 1 ? 16 : 0);
  gp.lds_x = gp.lds_lut + (stage_lut ? tpw * 8 * 32 : 0);
  gp.xs_pitch = p.k / 4;
  gp.lds_xs = gp.lds_x + p.m * gp.x_pitch;
  // (matrix-core contraction: the sums are [step of 128 k][16 rows] f32 instead -- more than one row's k / 4 bytes)
  gp.lds_red = gp.lds_xs + (mf && (p.k / 128) * 64 > p.m * gp.xs_pitch ? (p.k / 128) * 64 : p.m * gp.xs_pitch);
  gp.lds_nrm = gp.lds_red + 2 * 8 * p.m * 32 * 4;
  const unsigned lds = (unsigned)gp.lds_nrm + (unsigned)(8 * p.m * 4);
  if (lds > 160u * 1024u) return TG_PAIR_NA;
  if (p.dry) return TG_PLAN_GEMV;
  const dim3 grid((unsigned)wgs, (unsigned)batch);
#if GEMV_TRACE
  if (g_trace && g_trace_slots > 0) gp.trace = g_trace + (size_t)(g_trace_launch++ % g_trace_slots) * 512 * 8;
#endif
  if (mf) return dt == TG_BF16 ? go_mf_m<BF16>(p.m, d, p.norm_w != nullptr, gp, grid, lds, st)
                               : go_mf_m<F16>(p.m, d, p.norm_w != nullptr, gp, grid, lds, st);
  return dt == TG_BF16 ? go_m<BF16>(p.m, gps, d, p.norm_w != nullptr, gp, grid, lds, st)
                       : go_m<F16>(p.m, gps, d, p.norm_w != nullptr, gp, grid, lds, st);
}
