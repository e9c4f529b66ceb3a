// tg_pair.hip -- launch paths of w4_gemm_pair_kernel (Bint4 32x32x16 / Bint4 16x16x32 / Aint4) and its activation pre-pass;
// one object per 16-bit type (-DTG_TU_F16); see tg_common.cuh
#include "tg_common.cuh"
#ifdef TG_TU_F16
#define TG_TU_DT F16
#define TG_TU_SUF(n) n##_f16
#else
#define TG_TU_DT BF16
#define TG_TU_SUF(n) n##_bf16
#endif
namespace {
#include "w4_gemm_pair.cuh"
template <typename DT, int I, int GPS, int MR, bool QMX, int NSG, bool XG = false, int LA = 0, bool NORM = false, int ABLV = TG_PAIR_ABL>
int launch_pair_k(PairParams& pp, unsigned lds, hipStream_t st) {
#ifdef TG_DEV_MIN  // developer builds: only the headline instantiation (fast A/B builds)
#ifndef TG_DEV_GPS
#define TG_DEV_GPS 1
#endif
#ifndef TG_DEV_QMX
#define TG_DEV_QMX false
#endif
#ifndef TG_DEV_MR
#define TG_DEV_MR TG_PAIR_MR1
#endif
#ifndef TG_DEV_LA
#define TG_DEV_LA 0
#endif
  if constexpr (!(std::is_same<DT, BF16>::value && I == 4 && GPS == TG_DEV_GPS && MR == TG_DEV_MR && QMX == TG_DEV_QMX && NSG == TG_DEV_MIN && LA == TG_DEV_LA && !NORM)) return TG_PAIR_NA;
  else {
#endif
  if constexpr (QMX && !std::is_same<DT, BF16>::value) return TG_E_DTYPE;  // mx4 is bf16-only (TinyGemm_int4.cu:758)
  else {
  // several groups per super-tile (group 32 / 64 with wide super-tiles): more per-slot state, one slot in flight fits the
  // 128-VGPR budget without spills (ring depth measured irrelevant between 2 and 4)
  // (mx4 on the 32x32x16 tiles converts its weights in registers and has no per-group state in the slots: the usual depth)
  constexpr int RING = (QMX && LA == 0) ? TG_PAIR_R : (MR == 1 && NSG == 4 && LA == 0) ? 4 : GPS > 1 ? (LA ? TG_PAIR_RA1 : 1) : LA == 1 ? TG_PAIR_RA : LA == 2 ? TG_PAIR_RB16 : TG_PAIR_R;
  constexpr auto kern = w4_gemm_pair_kernel<DT, I, GPS, MR, QMX, RING, NSG, ABLV, XG, LA, NORM>;
  if (pp.dry) return TG_PLAN_PAIR;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc;
  const unsigned wgs = (unsigned)(pp.items < TG_PAIR_WGS ? pp.items : TG_PAIR_WGS);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, pp);
  return launch_status();
  }
#ifdef TG_DEV_MIN
  }
#endif
}

// The activation block of one pass does not fit next to the table (m = 8 at k = 4096, m = 1 at k >= 8192): the XG variant
// takes the activations pre-arranged from a caller-provided workspace (w4_xprep_kernel, one small launch in front).
// Workspace = [batch][m k 2 bytes] arranged activations, then [batch][passes][groups][xs_rows] f32 sums.
template <typename DT>
int launch_xprep(const PairParams& pp, int I, int ma, int64_t batch, hipStream_t st, int la = 0) {
  XPrepParams xq;
  xq.la = la;
  xq.x = pp.x; xq.xp = const_cast<char*>(pp.xp); xq.xsum = const_cast<char*>(pp.xsum);
  xq.x_tc = pp.x_tc;
  xq.m = pp.m; xq.k = pp.k; xq.ma = ma; xq.cps = I / 2; xq.gshift = pp.gshift; xq.gch_mask = pp.gch_mask;
  xq.ngroups = pp.ngroups; xq.xs_rows = pp.xs_rows;
  xq.stride_x = pp.stride_x; xq.stride_xp = pp.stride_xp; xq.stride_xsum = pp.stride_xsum;
  const int64_t chunks = (int64_t)pp.m * (pp.k / 32);
  hipLaunchKernelGGL(w4_xprep_kernel<DT>, dim3((unsigned)cdiv(chunks, 256), (unsigned)batch), dim3(256), 0, st, xq);
  return launch_status();
}

// m = 1 has its own specialisation (one accumulator register finalised per group, taken as a running difference) -- except
// with several groups per super-tile, where the general kernel's zero-C group starts compile without spills; `norm`: the
// instantiations with LlamaRMSNorm fused into the activation staging (staged activations, m <= 8, not mx4)
template <typename DT, int I, int GPS, bool QMX, int NSG>
int launch_pair_m(PairParams& pp, unsigned lds, hipStream_t st, bool xg, int m, int mregs, bool norm) {
  const bool m1 = m == 1 && TG_PAIR_MR1 == 1 && (QMX || GPS <= TG_PAIR_MR1_GPS);  // (mx4: no per-group state, the specialisation fits at any GPS)
  // Not instantiated (round 4: every one of them compiled with 70 ... 1100 bytes of scratch per lane, and a scratch reload drains
  // the weight ring behind vmcnt(0)): the 32-activation-row accumulator sets (m > 8 on staged activations), innerKTiles 8 beyond the
  // m = 1 kernel of int4 / any4, the fused norm with several groups per super-tile.  Those calls take the next kernel family
  // (16x16x32 tiles with workspace activations, w4_gemm_pair16_kernel, or the reference-numerics kernels).
  if (mregs != 4) return TG_PAIR_NA;
  if constexpr (I == 8) {
    if constexpr (QMX) return TG_PAIR_NA;
    else {
      if (!m1 || norm) return TG_PAIR_NA;
      return xg ? launch_pair_k<DT, I, GPS, 1, false, NSG, true>(pp, lds, st) : launch_pair_k<DT, I, GPS, 1, false, NSG>(pp, lds, st);
    }
  } else {
  if (xg) return m1 ? launch_pair_k<DT, I, GPS, 1, QMX, NSG, true>(pp, lds, st) : launch_pair_k<DT, I, GPS, 4, QMX, NSG, true>(pp, lds, st);
  if (norm) {
    if constexpr (QMX || GPS > 1) return TG_PAIR_NA;
    else {
      return m1 ? launch_pair_k<DT, I, GPS, 1, false, NSG, false, false, true>(pp, lds, st)
                : launch_pair_k<DT, I, GPS, 4, false, NSG, false, false, true>(pp, lds, st);
    }
  }
  if (m1) return launch_pair_k<DT, I, GPS, 1, QMX, NSG>(pp, lds, st);
  return launch_pair_k<DT, I, GPS, 4, QMX, NSG>(pp, lds, st);
  }
}

template <typename DT, int I, bool QMX>
int launch_pair(GemmParams& p, int64_t batch, hipStream_t st) {
  constexpr int RW = 64;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  const int mregs = p.m <= 8 ? 4 : 16;  // accumulator registers of a row set (8 or 32 activation rows per pass)
  const int ma = 2 * mregs;
  PairParams pp;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;  // super-tiles per group
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  const int mrows = p.m < ma ? p.m : ma;
  pp.rused = mrows < 4 ? mrows : mregs;
  pp.xs_rows = mrows <= 4 ? 4 : ma;
  pp.red_lanes = mrows <= 4 ? 32 : 64;
  pp.x_pitch = p.k * 2 + 16;
  pp.lds_x = QMX ? 0 : 65536;  // mx4 converts its weights in registers (v_cvt_scalef32_pk_bf16_fp4): no table, the LDS starts with the activations
  pp.lds_xs = (pp.lds_x + mrows * pp.x_pitch + 32 * I + 15) & ~15;  // staged rows + a zero piece of one super-tile
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
  pp.red_alias = !QMX && mrows > 4;  // 16 KiB and more of partial sums: reuse the table's LDS instead
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  // mx4: exponent blocks of 16 bytes per row, read at 4-byte alignment (w4_gemm_pair.cuh, e_request)
  // (a slice that starts off a 4-byte boundary loses up to 3 bytes of its one block)
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  // fused RMSNorm: done in the workgroup's own staging of the whole activation block (its partial sums borrow the activation-sum
  // area, which mx4 does not have); the workspace variant would need it in the pre-pass
  if (p.norm_w && (QMX || lds > 80u * 1024u || p.m > ma)) return TG_PAIR_NA;
  bool xg = false;
#ifdef TG_PAIR_FORCE_XG  // developer A/B: the workspace variant also where the staged plan fits
  if (mregs == 4 && p.m <= ma) lds = 1u << 30;
#endif
  if (lds > 80u * 1024u) {  // two workgroups per CU
    // XG: every wave keeps one super-tile of the pass's activations (<= 8 rows) in a private buffer
    if (mregs != 4 || p.m > ma) return TG_PAIR_NA;
    pp.xw_pitch = 32 * I + 16;
    pp.xw_bytes = (I == 2 ? 16 : 8) * pp.xw_pitch;  // a row for every 2 I lanes of the wave's (unmasked) store
    pp.lds_xs = (pp.lds_x + 8 * pp.xw_bytes + 32 * I + 15) & ~15;  // 8 buffers + the zero piece
    pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
    lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
    if (lds > 80u * 1024u) pp.red_alias = 1;
    if (QMX && pp.red_alias) return TG_PAIR_NA;  // (no table to put the partial sums over; cannot happen: 8 one-KiB buffers + 16 KiB)
    if (pp.red_alias) {
      lds = (unsigned)pp.lds_red;
      pp.lds_red = 0;
    }
    if (lds > 80u * 1024u) return TG_PAIR_NA;
    pp.stride_xp = (int64_t)p.m * p.k * 2;
    pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
    const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
    p.ws_need = need;
    if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
    pp.xp = p.ws;
    pp.xsum = p.ws + batch * pp.stride_xp;
    xg = true;
  }
  pp.rblocks = (p.wrows + RW - 1) / RW;
  pp.cblocks = (p.m + ma - 1) / ma;
  const int64_t items = (int64_t)pp.rblocks * pp.cblocks * batch;
  if (items > INT32_MAX) return TG_PAIR_NA;
  // The kernel's unit of work is a 64-row block over the whole k (8 waves): a launch needs about one item per workgroup slot
  // (2 per CU) to fill the chip.  Smaller launches (one 4096-row layer = 64 items) are latency-bound and stay on the
  // split-K kernels, which spread one 16-row tile over up to 16 waves.
  if (items < TG_PAIR_MIN_ITEMS) { p.ws_need = 0; return TG_PAIR_NA; }
  pp.items = (int32_t)items;
  // XG item dealing: chunks of consecutive items once every workgroup still gets several chunks
  pp.chunk = items >= (int64_t)TG_PAIR_WGS * TG_XG_CHUNK * 4 ? TG_XG_CHUNK : 1;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = p.norm_w; pp.norm_eps = p.norm_eps; pp.epilogue = p.epilogue;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  if (xg && !p.dry) {
    const int rc = launch_xprep<DT>(pp, I, ma, batch, st);
    if (rc != 0) return rc;
  }
#define TG_PAIR_M(GPS_, NSG_) launch_pair_m<DT, I, GPS_, QMX, NSG_>(pp, lds, st, xg, p.m, mregs, p.norm_w != nullptr)
  if (gps == 1) {
    // group boundaries at fixed places of the unrolled round when a group is one super-tile or one whole round
    // (the m = 1 specialisation too since its group update is spelled out instruction by instruction: before that, fixed
    //  boundaries made the compiler scatter its accumulator chain over several register tuples and spill)
    // (a group of ONE super-tile, g = 64 at I = 4, also keeps the run-time test: its fixed-boundary build spills 27 registers,
    //  m = 8 50 % against 59 %)
    const bool fixed = TG_PAIR_NSG2 && (TG_PAIR_NSG2_M1 || !(p.m == 1 && TG_PAIR_MR1 == 1));
    // TG_NUM_FAST_MFMA: the headline shape's m = 1 kernel with the 32x32x16 MFMA as its contraction (north_star: "dequantized
    // in-register and fed to bf16 MFMA") instead of the per-lane v_dot2 the default takes -- one instantiation, g = 128 at innerKTiles 4
    if constexpr (I == 4 && !QMX) {
      if (p.numerics == TG_NUM_FAST_MFMA && p.m == 1 && !xg && !p.norm_w && fixed && nsg == TG_PAIR_R)
        return launch_pair_k<DT, I, 1, 1, false, TG_PAIR_R, false, 0, false, 100>(pp, lds, st);
    }
    if (fixed && nsg == TG_PAIR_R) return TG_PAIR_M(1, TG_PAIR_R);
    // m = 1, a group of ONE super-tile (g = 64 at innerKTiles 4): fixed boundaries too since the dot2 contraction freed the registers
    // (with the MFMA this build spilled 27; 77 -> 81 %), and a group of FOUR super-tiles (g = 256) as one round of a ring of four
    // (76.7 -> 84.0 %; a ring of four at g = 128 / 64 measured 1-1.5 points below the ring of two).  Only the m = 1 kernels are
    // instantiated for these (launch_pair_k directly: launch_pair_m would drag the general kernels in as well).
    if constexpr (!QMX) {
      if (fixed && p.m == 1 && TG_PAIR_MR1 == 1 && !p.norm_w && (nsg == 1 || nsg == 4)) {
        if (nsg == 1) return xg ? launch_pair_k<DT, I, 1, 1, false, 1, true>(pp, lds, st) : launch_pair_k<DT, I, 1, 1, false, 1>(pp, lds, st);
        if constexpr (I <= 4)  // (innerKTiles 8: four super-tiles would be g = 512)
          return xg ? launch_pair_k<DT, I, 1, 1, false, 4, true>(pp, lds, st) : launch_pair_k<DT, I, 1, 1, false, 4>(pp, lds, st);
      }
    }
    return TG_PAIR_M(1, 0);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return TG_PAIR_M(2, 0);
  }
  if constexpr (I >= 8) {
    if (gps == 4) return TG_PAIR_M(4, 0);
  }
#undef TG_PAIR_M
  return TG_PAIR_NA;
}

// Aint4 weights (weightOnRight = false) on the pair-table kernel: 32 weight rows per work item, v_mfma_f32_16x16x32,
// activations always through the workspace (one pass of at most 8 rows).
template <typename DT, int I, bool QMX>
int launch_pair_a(GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (I < 2) return TG_PAIR_NA;  // one 16-k tile per word set: no word pair for a 32-k MFMA step
  else {
  if (p.m > 16 || p.x_tc || p.y_tc || p.norm_w || p.epilogue) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  PairParams pp;
  pp.x_tc = pp.y_tc = 0; pp.y_tiles = 0;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = nullptr; pp.norm_eps = 0.f; pp.epilogue = 0;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  const int mrows = p.m;
  pp.rused = mrows < 4 ? mrows : 4;
  pp.xs_rows = mrows <= 4 ? 4 : mrows <= 8 ? 8 : 16;
  pp.red_lanes = mrows <= 8 ? 32 : 64;  // lanes 0..31 hold activation rows 0..7
  pp.x_pitch = 0;
  pp.lds_x = QMX ? 0 : 65536;
  pp.xw_pitch = 0;  // the lanes' MFMA operands come straight from the workspace: LDS only holds a zero piece here
  pp.xw_bytes = 0;
  pp.lds_xs = (pp.lds_x + 32 * I + 15) & ~15;
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  pp.red_alias = lds > 80u * 1024u;
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  if (lds > 80u * 1024u) return TG_PAIR_NA;
  pp.stride_xp = (int64_t)(p.m + 1) * p.k * 2;  // + a zero row
  pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
  const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
  pp.rblocks = (p.wrows + 31) / 32;
  pp.cblocks = 1;
  const int64_t items = (int64_t)pp.rblocks * batch;
  if (items > INT32_MAX || items < TG_PAIR_MIN_ITEMS) return TG_PAIR_NA;
  p.ws_need = need;
  if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
  pp.xp = p.ws;
  pp.xsum = p.ws + batch * pp.stride_xp;
  pp.items = (int32_t)items;
  pp.chunk = 1;  // plain round-robin dealing (chunks of consecutive items measured slower for the 32-row items of this layout)
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  if (!p.dry) {
    const int rc = launch_xprep<DT>(pp, I, 16, batch, st, 1);
    if (rc != 0) return rc;
  }
  if (gps == 1) {
    if (TG_PAIR_NSG2 && nsg == 1) return launch_pair_k<DT, I, 1, 4, QMX, 1, true, true>(pp, lds, st);
    if (TG_PAIR_NSG2 && nsg == TG_PAIR_R) return launch_pair_k<DT, I, 1, 4, QMX, TG_PAIR_R, true, true>(pp, lds, st);
    return launch_pair_k<DT, I, 1, 4, QMX, 0, true, true>(pp, lds, st);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return launch_pair_k<DT, I, 2, 4, QMX, 0, true, true>(pp, lds, st);
  }
  return TG_PAIR_NA;
  }
}

// Bint4 weights with 9 ... 16 activation rows: the 16x16x32 structure of the A-side kernel (32-row work items, duplicated
// table, activations of one pass -- all m <= 16 rows -- straight from the workspace into the MFMA operand) on B-layout words:
// one packed word is one B operand, 4 vector ops per word.  (The 32x32x16 kernel holds 8 rows per pass; a second pass would
// stream the weights twice.)
template <typename DT, int I, bool QMX>
int launch_pair_b16(GemmParams& p, int64_t batch, hipStream_t st) {
  if (p.m > 16 || p.norm_w || p.epilogue) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  PairParams pp;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = nullptr; pp.norm_eps = 0.f; pp.epilogue = 0;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  const int mrows = p.m;
  pp.rused = mrows < 4 ? mrows : 4;
  pp.xs_rows = mrows <= 4 ? 4 : mrows <= 8 ? 8 : 16;
  pp.red_lanes = mrows <= 8 ? 32 : 64;  // lanes 0..31 hold activation rows 0..7
  pp.x_pitch = 0;
  pp.lds_x = QMX ? 0 : 65536;
  pp.xw_pitch = 0;
  pp.xw_bytes = 0;
  pp.lds_xs = (pp.lds_x + 32 * I + 15) & ~15;
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  pp.red_alias = lds > 80u * 1024u;
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  if (lds > 80u * 1024u) return TG_PAIR_NA;
  pp.stride_xp = (int64_t)(p.m + 1) * p.k * 2;  // + a zero row
  pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
  const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
  pp.rblocks = (p.wrows + 31) / 32;
  pp.cblocks = 1;
  const int64_t items = (int64_t)pp.rblocks * batch;
  if (items > INT32_MAX || items < 2 * TG_PAIR_MIN_ITEMS) return TG_PAIR_NA;  // (32-row items: two per 64-row item of the other kernel)
  p.ws_need = need;
  if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
  pp.xp = p.ws;
  pp.xsum = p.ws + batch * pp.stride_xp;
  pp.items = (int32_t)items;
  pp.chunk = items >= (int64_t)TG_PAIR_WGS * TG_B16_CHUNK * 4 ? TG_B16_CHUNK : 1;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  if (!p.dry) {
    const int rc = launch_xprep<DT>(pp, I, 16, batch, st, 1);
    if (rc != 0) return rc;
  }
  if (gps == 1) {
    if (TG_PAIR_NSG2 && nsg == 1) return launch_pair_k<DT, I, 1, 4, QMX, 1, true, 2>(pp, lds, st);
    if (TG_PAIR_NSG2 && nsg == TG_PAIR_RB16) return launch_pair_k<DT, I, 1, 4, QMX, TG_PAIR_RB16, true, 2>(pp, lds, st);
    return launch_pair_k<DT, I, 1, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return launch_pair_k<DT, I, 2, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  if constexpr (I >= 8) {
    if (gps == 4) return launch_pair_k<DT, I, 4, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  return TG_PAIR_NA;
}

template <typename DT, int I>
int pair_q(bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return qmx ? launch_pair<DT, I, true>(p, batch, st) : launch_pair<DT, I, false>(p, batch, st);
}
template <typename DT, int I>
int pair_a_q(bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return qmx ? launch_pair_a<DT, I, true>(p, batch, st) : launch_pair_a<DT, I, false>(p, batch, st);
}
template <typename DT, int I>
int pair_b16_q(bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return qmx ? launch_pair_b16<DT, I, true>(p, batch, st) : launch_pair_b16<DT, I, false>(p, batch, st);
}
}  // namespace
namespace tgx {
int TG_TU_SUF(pair)(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return I == 2 ? pair_q<TG_TU_DT, 2>(qmx, p, batch, st) : I == 4 ? pair_q<TG_TU_DT, 4>(qmx, p, batch, st) : pair_q<TG_TU_DT, 8>(qmx, p, batch, st);
}
int TG_TU_SUF(pair_a)(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return I == 1 ? pair_a_q<TG_TU_DT, 1>(qmx, p, batch, st) : I == 2 ? pair_a_q<TG_TU_DT, 2>(qmx, p, batch, st) : pair_a_q<TG_TU_DT, 4>(qmx, p, batch, st);
}
int TG_TU_SUF(pair_b16)(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  // (innerKTiles 8 on the 16x16x32 tiles compiled with > 100 bytes of scratch per lane: not instantiated)
  return I == 2 ? pair_b16_q<TG_TU_DT, 2>(qmx, p, batch, st) : I == 4 ? pair_b16_q<TG_TU_DT, 4>(qmx, p, batch, st) : (int)TG_PAIR_NA;
}
}  // namespace tgx
