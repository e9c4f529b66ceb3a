// tg_stream.hip -- launch path of w4_gemm_stream_kernel (reference numerics); one object per 16-bit type (-DTG_TU_F16); see tg_common.cuh
#include "tg_common.cuh"
#ifdef TG_TU_F16
#define TG_TU_DT F16
#define TG_TU_FN stream_f16
#else
#define TG_TU_DT BF16
#define TG_TU_FN stream_bf16
#endif
namespace {
#include "w4_gemm_stream.cuh"
#ifndef STREAM_MINW
#define STREAM_MINW 4
#endif
// Bint4, TG_STREAM_LK_MIN <= m <= 15: the lookup block with the X fragment read under an EXEC mask (w4_gemm_stream.cuh, LK).  Same-box
// A/B at 4096^2, reference numerics: m = 8 46.9 -> 50.1 % of the roofline; m = 4 50.9 -> 49.8 and m = 1 55.6 -> 54.9 (the block's
// lgkmcnt(0) costs more than the few lanes' reads save); m = 16 reads with every lane either way.
#ifndef TG_STREAM_LK_MIN
#define TG_STREAM_LK_MIN 6
#endif
// ---- streaming kernel launch ---------------------------------------------------------------------
// LDS per workgroup: lookup tables (4 KiB per wave and row set) + two X slabs (+ split-K tiles).
template <bool LAYOUT_A>
inline unsigned stream_lds_bytes(int sw, int mrows, int sk, bool privx = false) {
  const unsigned nr = 1u;  // lookup tables per wave
  const unsigned unit = LAYOUT_A ? 64u : 128u;
  // shared slab: rows of all k-slices; private slabs: one per wave with the rows of its own slice; + the all-zero row
  const unsigned slab = (unsigned)(mrows * 4 * (privx ? 1 : sk) + 1) * (unit * 2u + 16u);
  return (unsigned)sw * nr * 4096u + 2u * slab * (privx ? (unsigned)sw : 1u) + (sk > 1 ? (unsigned)sw * nr * 1024u : 0u);
}

template <typename DT, bool LAYOUT_A, int WPL, bool QMX, int SW, bool privx = (SW == 1)>
int launch_stream_sw(StreamParams& sp, int sk, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int UNIT = LAYOUT_A ? 64 : 128;
  constexpr unsigned NR = 1u;
  const int nunits = (sp.k + UNIT - 1) / UNIT;
  const int upg = (1 << sp.gshift) / UNIT;  // units per quantisation group (>= 1)
  const int mrows = sp.m < 16 ? sp.m : 16;
  int nu = (nunits + 4 * sk - 1) / (4 * sk);
  nu = (nu + upg - 1) / upg * upg;
  sp.splitk = sk;
  sp.sk_shift = 0;
  while ((1 << sp.sk_shift) < sk) ++sp.sk_shift;
  sp.units_per_lane = nu;
  sp.upg_mask = upg - 1;
  const int xrows = mrows * 4 * (privx ? 1 : sk);
  sp.xslab_bytes = (xrows + 1) * (UNIT * 2 + 16);
  sp.red_off = (int32_t)(SW * NR * 4096u + 2u * (unsigned)sp.xslab_bytes * (privx ? SW : 1));
  const int pieces = xrows * (UNIT * 2 / 16);
  // privx: every wave stages its own X slab (no barrier in the main loop)
  const int nstage = privx ? 64 : SW * 64;
  const int xl = pieces <= nstage ? 1 : (pieces <= 2 * nstage ? 2 : 4);
  const unsigned lds = stream_lds_bytes<LAYOUT_A>(SW, mrows, sk, privx);
  const int tpb = SW / sk;
  dim3 grid((unsigned)((sp.rowtiles + tpb - 1) / tpb), (unsigned)coltiles, (unsigned)batch);
#define TG_LAUNCH_STREAM_LK(XL, LK_)                                                                      \
  do {                                                                                                    \
    constexpr auto kern = w4_gemm_stream_kernel<DT, LAYOUT_A, WPL, QMX, SW, STREAM_MINW, XL, privx, 0, false, LK_>; \
    if (sp.dry) return TG_PLAN_STREAM;                                                                    \
    const int prc = prepare_lds_kernel<kern>();                                                           \
    if (prc != 0) return prc;                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(SW * 64), lds, st, sp);                                           \
  } while (0)
#define TG_LAUNCH_STREAM(XL)                                                                              \
  do {                                                                                                    \
    if constexpr (!LAYOUT_A && !privx) {                                                                  \
      if (sp.m >= TG_STREAM_LK_MIN && sp.m <= 15) { TG_LAUNCH_STREAM_LK(XL, 1); break; }                  \
    }                                                                                                     \
    TG_LAUNCH_STREAM_LK(XL, 0);                                                                           \
  } while (0)
  if constexpr (privx && SW > 1) {
    if (xl != 1) return TG_E_SHAPE;  // private slabs with split-K are only instantiated for one piece per lane (m = 1)
    TG_LAUNCH_STREAM(1);
  } else if constexpr (SW == 1) {
    if (xl == 1) TG_LAUNCH_STREAM(1);
    else TG_LAUNCH_STREAM(2);
  } else {
    if (xl == 1) TG_LAUNCH_STREAM(1);
    else if (xl == 2) TG_LAUNCH_STREAM(2);
    else TG_LAUNCH_STREAM(4);
  }
#undef TG_LAUNCH_STREAM
#undef TG_LAUNCH_STREAM_LK
  return launch_status();
}

// Resident-X launch: 16-wave workgroups, the whole [mrows][k] activation block staged once per workgroup.
template <typename DT, bool LAYOUT_A, int WPL, bool QMX>
int launch_stream_xres(StreamParams& sp, int64_t coltiles, int64_t batch, unsigned lds, hipStream_t st) {
  // one workgroup walks up to 4 consecutive groups of 16 tiles of its layer (X staged once, tile-granularity tail
  // amortised) as long as that leaves at least two workgroups per CU
  int tpw = 4;
  while (tpw > 1 && ((sp.rowtiles + 16 * tpw - 1) / (16 * tpw)) * coltiles * batch < 512) tpw >>= 1;
  sp.tiles_per_wave = tpw;
  if (sp.dry) return TG_PLAN_STREAM;
  dim3 grid((unsigned)((sp.rowtiles + 16 * tpw - 1) / (16 * tpw)), (unsigned)coltiles, (unsigned)batch);
  if constexpr (!LAYOUT_A) {
    if (sp.m >= TG_STREAM_LK_MIN && sp.m <= 15) {
      constexpr auto kern = w4_gemm_stream_kernel<DT, LAYOUT_A, WPL, QMX, 16, STREAM_MINW, 1, false, 0, true, 1>;
      const int prc = prepare_lds_kernel<kern>();
      if (prc != 0) return prc;
      hipLaunchKernelGGL(kern, grid, dim3(16 * 64), lds, st, sp);
      return launch_status();
    }
  }
  constexpr auto kern = w4_gemm_stream_kernel<DT, LAYOUT_A, WPL, QMX, 16, STREAM_MINW, 1, false, 0, true, 0>;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc;
  hipLaunchKernelGGL(kern, grid, dim3(16 * 64), lds, st, sp);
  return launch_status();
}

template <typename DT, bool LAYOUT_A, int WPL, bool QMX>
int launch_stream(const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int UNIT = LAYOUT_A ? 64 : 128;
  constexpr int RPW = 16;  // weight rows per wave
  StreamParams sp;
  sp.x = p.x; sp.w = p.w; sp.qinfo = p.qinfo; sp.lut = p.lut; sp.y = p.y;
  sp.m = p.m; sp.wrows = p.wrows; sp.k = p.k; sp.ntiles = p.ntiles; sp.ksuper = p.ksuper;
  sp.gshift = p.gshift; sp.ngroups = p.ngroups; sp.qtype = p.qtype;
  sp.rowtiles = (p.wrows + RPW - 1) / RPW;
  sp.tiles_per_wave = 1;
  sp.stride_x = p.stride_x; sp.stride_w = p.stride_w; sp.stride_qinfo = p.stride_qinfo;
  sp.stride_lut = p.stride_lut; sp.stride_y = p.stride_y;
  sp.bias = p.bias; sp.stride_bias = p.stride_bias; sp.bias_row_stride = p.bias_row_stride; sp.dry = p.dry;
  const int mrows = p.m < 16 ? p.m : 16;
  const int nunits = (p.k + UNIT - 1) / UNIT;
  const int upg = (1 << p.gshift) / UNIT;
  // split-K: aim for at least two rounds of 16 waves on every CU; the X slab limits act rows * splitk to 16
  const int64_t wave_tiles = (int64_t)sp.rowtiles * coltiles * batch;
  int sk = 1;
  // (m = 1, private slabs: one round of 16 waves per CU is enough -- measured on the Llama-3-8B shapes)
  const int64_t want = mrows == 1 ? 256 * 16 : 2 * 256 * 16;
  while (sk < 8 && wave_tiles * sk < want && nunits >= 8 * sk * upg && mrows * sk * 2 <= 16) sk *= 2;
#ifdef TG_DEV
  static const int sk_env = getenv("TG_SK") ? atoi(getenv("TG_SK")) : 0;  // developer override
  if (sk_env > 0) sk = sk_env;
#endif
  // m == 1: every wave stages its own X slab (no barrier in the main loop); a workgroup is the sk waves of one tile
#ifdef TG_DEV
  static const int xres_env = getenv("TG_XRES") ? atoi(getenv("TG_XRES")) : 1;  // developer knob: 0 off, 2 also for m = 1
#else
  constexpr int xres_env = 1;
#endif
  if (mrows == 1 && xres_env != 2) {
    switch (sk) {
      case 1: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 1>(sp, 1, coltiles, batch, st);
      case 2: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 2, true>(sp, 2, coltiles, batch, st);
      case 4: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 4, true>(sp, 4, coltiles, batch, st);
      default: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 8, true>(sp, 8, coltiles, batch, st);
    }
  }
  // m >= 2, one tile per wave: keep the whole activation block resident in LDS when it fits next to 16 lookup
  // tables (m = 8 at k = 4096 does: 66 KiB + 64 KiB) -- one barrier per workgroup instead of one per unit
  // (measured: wins for m >= 8 at k = 4096 and for m >= 2 at k = 8192; the 16-wave workgroup costs ~15 % in tile-granularity
  //  tail against 4-wave workgroups, which the small slabs of m <= 4 at k = 4096 do not pay back)
  if (sk == 1 && xres_env && (mrows * UNIT >= 1024 || sp.k >= 8192 || xres_env == 2)) {
    const int nu = (int)(((nunits + 3) / 4 + upg - 1) / upg * upg);
    const unsigned xrow = (unsigned)(nu * UNIT * 2 + 16);
    const unsigned lds = 16u * 4096u + (unsigned)(mrows * 4) * xrow + (unsigned)(UNIT * 2 + 16);
    if (lds <= 160u * 1024u) {
      sp.splitk = 1; sp.sk_shift = 0; sp.units_per_lane = nu; sp.upg_mask = upg - 1;
      sp.xslab_bytes = (int32_t)xrow; sp.red_off = 0;
      return launch_stream_xres<DT, LAYOUT_A, WPL, QMX>(sp, coltiles, batch, lds, st);
    }
  }
  // otherwise 4-wave workgroups while their LDS footprint lets 16 waves live on a CU and the X slab is small;
  // X slabs of 8 KiB or more per unit (Bint4: m >= 8, Aint4: m = 16): 8-wave workgroups halve the staging work per wave
  const int sk4 = sk < 4 ? sk : 4;
  if (mrows * UNIT < 1024 && 160u * 1024u / stream_lds_bytes<LAYOUT_A>(4, mrows, sk4) >= 4)
    return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 4>(sp, sk4, coltiles, batch, st);
  return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 8>(sp, sk, coltiles, batch, st);
}

template <typename DT, bool LAYOUT_A, int WPL>
int stream_q(bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  if constexpr (!std::is_same<DT, BF16>::value) {
    if (qmx) return TG_E_DTYPE;
    return launch_stream<DT, LAYOUT_A, WPL, false>(p, coltiles, batch, st);
  } else {
    return qmx ? launch_stream<DT, LAYOUT_A, WPL, true>(p, coltiles, batch, st) : launch_stream<DT, LAYOUT_A, WPL, false>(p, coltiles, batch, st);
  }
}
template <typename DT, bool LAYOUT_A>
int stream_w(int wpl, bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  switch (wpl) {
    case 1: return stream_q<DT, LAYOUT_A, 1>(qmx, p, coltiles, batch, st);
    case 2: return stream_q<DT, LAYOUT_A, 2>(qmx, p, coltiles, batch, st);
    default: return stream_q<DT, LAYOUT_A, 4>(qmx, p, coltiles, batch, st);
  }
}
}  // namespace
namespace tgx {
int TG_TU_FN(bool layout_a, int wpl, bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  return layout_a ? stream_w<TG_TU_DT, true>(wpl, qmx, p, coltiles, batch, st) : stream_w<TG_TU_DT, false>(wpl, qmx, p, coltiles, batch, st);
}
}  // namespace tgx
