// w4_gemm_xr.cuh -- the pair-table W4A16 kernel with REGISTER-RESIDENT activations ("XR"): Bint4 weights, up to 16 activation rows,
// k = 32 * 8 * NCH (k = 4096: NCH = 16; k = 8192: NCH = 32).
//
// Same contract and numerics as w4_gemm_pair.cuh (TG_NUM_FAST, group-scaled; reference TinyGemmImpl.cuh:23-345 with
// BLayout_TC_int4, MatrixLayoutB.cuh:686-1101, Dequantization.cuh:55-178).  Why another decomposition: with 9 ... 16 activation
// rows the 16x16x32 variant of w4_gemm_pair.cuh re-reads the whole activation block for every 32-row work item -- two bytes of
// activations through the vector-memory path per byte of weights, measured as 26 % of its time (same-box ablation, DESIGN.md section 9;
// the workspace variant for m <= 8 pays 0.5 bytes per byte) -- and the 64 KiB pair table leaves no LDS to keep the block in.  The register file does: a wave only ever needs the
// activations of ITS k-slice, 16 rows x 512 k = 16 KiB = 64 VGPRs, IF the wave may use 256 registers.  So:
//
//   workgroup  = 8 waves, ONE per CU (256 VGPRs per lane), persistent over a contiguous range of work items; item = 64 weight
//                rows (four 16-row MFMA tiles) x the whole k; wave w walks the k-slice w (split-K 8, partial sums meet in LDS
//                and are added in wave order: deterministic).
//   MFMA       = v_mfma_f32_16x16x32: A operand = activations (lane (i = lane & 15, kb = lane >> 4): row i, the 16-byte
//                piece (chunk, kb) of the "byte order" of w4_gemm_pair.cuh), B operand = weights, D[i][n]: lane (n, kb)
//                holds activation rows 4 kb + r of ITS weight row, so scale / zero are per-lane scalars.
//   activations= xr[NCH]: the wave's NCH pieces, arranged by the kernel itself from the caller's x (x_prepare: no pre-pass launch,
//                no workspace) once per PROBLEM of the batch and workgroup, together with the per-group sums of the slice.
//   weights    = "load layout": lane (n = lane & 15, b = (lane >> 4) & 1, a = lane >> 5) reads the 4 I bytes of row
//                32 u + 16 b + n (tile 2 u + b of the item) at lane-quads 2 a, 2 a + 1 of the reference layout: one wave-load =
//                four fully used 256-byte segments (I = 4).
//   table      = [256 byte values][64 columns] x 4 bytes, column = row of the item, every row ONCE: in the load layout a 32-lane
//                LDS access group (n, b) touches 32 distinct rows = 32 distinct banks.  The looked-up registers V (quad 2 a) and
//                W (quad 2 a + 1) then go through ONE v_permlane16_swap: V' = tile 2 u at quad 2 a + b = kb, W' = tile 2 u + 1 at
//                quad kb -- the two B operands.  6 vector ops per packed word instead of 4; the LDS work per weight is that of
//                the m = 1 kernel and the table is built once per 64 rows.
//   two tables = the NEXT item's table is built into the other 64 KiB buffer by steps interleaved with the second half of the
//                main loop (its LUT rows are requested at the item's start), so no wave ever waits for a table.
//   tail       = partial sums over the (finished) current table, three barriers per item; the weight ring (R super-tiles per
//                wave, refilled with the next item's positions) keeps streaming through it.
//   mx4        = template flag QMX below: no table at all.
#pragma once
#ifndef XR_ABL
#define XR_ABL 0  // developer ablations (dev/README.md): 1 no table lookups, 3 no weight loads, 5 no table build, 6 no split-K tail, 7 no output stores, 8 no third barrier (a race: timing only); 0 in the product
#endif

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): the stage loop of the kernel as a fold expression
// (`#pragma unroll` gives up on the 64 stages of k = 8192 -- and every register array becomes scratch)
template <int... I, class F>
__device__ __forceinline__ void xr_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void xr_static_for(F&& f) {
  xr_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

struct XrParams {
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  const char* x;     // activations [problem][m][k] 16-bit, row-major or (x_tc) in A-fragment order
  int64_t stride_x;
  int32_t x_tc;
  int32_t m, wrows, k;
  int32_t ntiles, ksuper, gshift, ngroups, qtype;
  int32_t rblocks;   // 64-row blocks per problem
  int32_t items;     // rblocks * batch
  int32_t lds_xs;    // LDS byte offset of the staged activation sums, f32 [ngroups][16] (behind the two tables)
  int64_t stride_w, stride_qinfo, stride_lut, stride_y;
  const char* bias;
  int64_t stride_bias, bias_row_stride;
  int32_t y_tc, y_tiles;
  int32_t dry;
  int32_t y_f32;     // 1: y is f32 [problem][m][wrows] holding the UNROUNDED sums of a k-window (no bias): the caller adds the windows
};

// I   = innerKTiles of the Bint4 layout (2, 4, 8)
// NCH = 32-k chunks of a wave's k-slice (k = 256 NCH)
// CPG = 32-k chunks per quantisation group, at most NCH (a group that spans several waves' slices: CPG = NCH)
// R   = super-tiles a wave keeps in flight
// QMX = mx4: the weights are converted in registers (v_cvt_scalef32_pk_bf16_fp4, w4_gemm_pair.cuh: mx4_cvt_word) with the group's
//       scale 2^(e - 127) inside the conversion: no table, no lookups, no per-group accumulator updates; the word pair of a lane is
//       swapped BEFORE the conversion (one v_permlane16_swap per stage); the e8m0 exponents of a lane's four rows over its k-slice
//       are one 16-byte load per row and item, requested one item ahead.  CPG is 1 (a group is one 32-k chunk).
// WV  = waves per workgroup = k-slices (8: two waves per SIMD with 256 registers each; 16: four per SIMD with 128 -- a slice is half
//       as long, so the activation registers are 32 instead of 64 and the ring two super-tiles deep: the same bytes in flight per CU,
//       twice the waves to hide LDS and MFMA latency behind)
// PK  = at most 8 activation rows: the A operand's rows 8 ... 15 would be zeros, so a register set holds TWO chunks -- lanes of
//       rows 0 ... 7 the piece of chunk 2 c, lanes of rows 8 ... 15 the piece of chunk 2 c + 1 (of row i - 8).  The even chunk's MFMAs
//       take the registers as they are, the odd chunk's a copy rotated by 8 lanes within every 16-lane row (four v_mov_b32_dpp
//       row_ror:8 per chunk, shared by the stage's two tile pairs); what the other eight rows of the operand then hold only reaches
//       accumulator rows 8 ... 15, which are never stored.  Half the activation registers: k = 8192 in the 64 registers k = 4096
//       takes unpacked, so its ring is four super-tiles deep like there (unpacked, k = 8192 leaves room for two: 66 % at m = 8).
// WV = 4 (with PK: at most 8 rows): TWO workgroups of four k-slices per CU instead of one of eight.  The same registers per wave
//       (a slice is twice as long, packed rows halve it again) and the same two waves per SIMD -- but the split-K tail and the
//       table build of one workgroup, which stall the whole CU when it is alone on it (same-box ablation: the tail costs 8-10
//       points), run under the other workgroup's main loop.  Each workgroup has ONE table (64 KiB, rebuilt between two barriers
//       after the item's sums are handed over) and its own 8 KiB hand-over region: 2 x 74 KiB of LDS.
template <typename DT, int I, int NCH, int CPG, int R, bool QMX = false, int WV = 8, bool PK = false>
__global__ void __launch_bounds__(WV * 64, WV == 16 ? 1 : 2) w4_gemm_xr_kernel(const XrParams p) {
  constexpr int WAVES = WV;
  constexpr bool ONE_TABLE = WV == 4;
  static_assert(WV == 8 || (WV == 16 && !QMX) || (WV == 4 && PK), "8 or 16 k-slices (mx4: 8); 4 with packed rows");
  static_assert(!PK || (!QMX && WV <= 8 && NCH % 2 == 0), "packed rows: the lookup kernel, chunk pairs");
  constexpr int NXR = PK ? NCH / 2 : NCH;          // activation register sets of a wave
  // ZM: the zero-point term sum_g zero[g, row] X[g][a] (X = the group's sum of activations) on the MATRIX CORE, once per item and tile,
  // instead of one FMA per accumulator register and group (64 of the ~680 vector instructions of a wave and item, plus the LDS reads
  // of the staged sums).  zero is a 16-bit value already; X (f32) is split into three 16-bit parts hi + mid + lo (exact to 2^-27),
  // so the K = 32 slots of one 16x16x32 MFMA hold up to 8 groups x 3 parts: A operand lane (row i, k-quad kb) = part p = kb R + rep
  // (R = 8 / NG repeats of the NG groups per k-quad) of X[gidx][i], B operand = zero[gidx][row n] in every k-quad (lane-uniform
  // in kb: no selects), C = the item's sums: y += A B directly.
  constexpr int NG = NCH / CPG;                    // quantisation groups of a wave's slice
#ifndef TG_XR_ZM
#define TG_XR_ZM 1
#endif
  constexpr bool ZM = TG_XR_ZM && !QMX && WV == 8 && NG <= 8 && (PK ? (CPG > 1 && NCH <= 32) : NCH <= 24);  // (32 unpacked chunks: its extra registers spill)
  constexpr int ZR = NG <= 8 ? 8 / NG : 1;         // repeats of the group pattern inside a k-quad's 8 slots
  constexpr int CPS = I / 2;                       // 32-k chunks per super-tile
  constexpr int NST = NCH / CPS;                   // super-tiles of a wave's slice
  constexpr int NWL = 2 * CPS;                     // packed words per lane, tile pair and super-tile
  constexpr int GPS = CPG < CPS ? CPS / CPG : 1;   // groups per super-tile
  constexpr int SPG = CPG > CPS ? CPG / CPS : 1;   // super-tiles per group
  static_assert(NCH % CPS == 0 && NST % R == 0 && NCH % CPG == 0, "slice = whole super-tiles, whole rounds of the ring, whole groups");
  static_assert(NCH >= 16 && NCH % 2 == 0, "the table build (16 steps) is spread over the chunks of the slice's second half");
  constexpr uint32_t TABLE = 65536u;
  static_assert(!QMX || (CPG == 1 && NCH == 16), "mx4: one group per chunk, one 16-byte exponent block per row and slice");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, lb = (lane >> 4) & 1, la = lane >> 5;
  const uint32_t lds_xs = (uint32_t)p.lds_xs;

  // ---- work items: a contiguous range per workgroup (the activation registers change only when the problem does) ----
  const int it_begin = (int)(((int64_t)blockIdx.x * p.items) / gridDim.x);
  const int it_end = (int)(((int64_t)(blockIdx.x + 1) * p.items) / gridDim.x);
  if (it_begin >= it_end) return;
  struct Item {
    int b, rb;
  };
  Item cur{it_begin / p.rblocks, it_begin % p.rblocks};

  // ---- per-lane addressing of an item ----
  // (the host guarantees wrows % 64 == 0: no row of an item is padding, so tile pair u / tile t are at wave-uniform distances
  //  from the lane's first row and need no registers of their own)
  struct Rows {
    uint32_t wbase;  // byte offset of this lane's words in super-tile 0 of tile pair 0
    uint32_t qrow4;  // byte offset of tile 0's row (lane & 15) in a group's scale | zero words
    const char* wb;
    const char* qb;
  };
  const uint32_t pair_stride = (uint32_t)(4 * p.ksuper * 32 * 2 * I);  // bytes between the tile pairs' words (four 8-row tiles)
  auto rows_of = [&](const Item& e) -> Rows {
    Rows r;
    const int row = e.rb * 64 + 16 * lb + ln;
    r.wbase = ((uint32_t)(row >> 3) * (uint32_t)p.ksuper * 32u + (uint32_t)(4 * (row & 7) + 2 * la)) * (uint32_t)(2 * I);
    r.qrow4 = (uint32_t)(e.rb * 64 + ln) * 4u;
    r.wb = p.w + (int64_t)e.b * p.stride_w;
    r.qb = p.qinfo + (int64_t)e.b * p.stride_qinfo;
    return r;
  };

  // ---- ring of R super-tiles ----
  struct Slot {
    uint32_t w[2][NWL];
    uint32_t q[4][GPS];
  };
  Slot ring[R];
  auto pin = [](uint32_t& v) -> uint32_t { asm volatile("" : "+v"(v)); return v; };
  auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  const int s_begin = wave * NST;
  // position l of the slice (compile-time after unrolling); an invalid request (past the last item) re-reads super-tile 0:
  // issued all the same so that the number of loads in flight is the same on every path (w4_gemm_pair.cuh, issue)
  auto issue = [&](Rows& rw, int l, Slot& sl, bool valid) {
    const uint32_t sv = valid ? (uint32_t)(s_begin + l) : 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const char* src = rw.wb + uni(sv * (uint32_t)(64 * I) + (uint32_t)u * pair_stride) + pin(rw.wbase);
      if constexpr (XR_ABL == 3) {  // ablation: no weight loads
#pragma unroll
        for (int j = 0; j < NWL; ++j) sl.w[u][j] = (uint32_t)(l * 0x01030507 + j * 0x11 + u) + rw.wbase;
      } else if constexpr (NWL == 2) {
        const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src));
        sl.w[u][0] = v[0]; sl.w[u][1] = v[1];
      } else {
#pragma unroll
        for (int v4 = 0; v4 < NWL / 4; ++v4) {
          const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + v4);
#pragma unroll
          for (int j = 0; j < 4; ++j) sl.w[u][4 * v4 + j] = v[j];
        }
      }
    }
    if (!QMX && l % SPG == 0) {  // (compile-time) the first super-tile of its group(s): scale | zero words
#pragma unroll
      for (int gg = 0; gg < GPS; ++gg) {
        const uint32_t g = (uint32_t)(((sv * CPS + gg * CPG) * 32) >> p.gshift);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          sl.q[t][gg] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(rw.qb + uni(g * (uint32_t)p.wrows * 4u + (uint32_t)(t * 64)) + pin(rw.qrow4)));
      }
    }
  };

  // ---- LUT rows of this thread's table column (= row of the item), requested one item ahead ----
  u32x4 lpa, lpb;  // (two register vectors, every element index a constant: a private ARRAY here was promoted to static LDS)
  uint32_t lhw2 = 0u;  // WV = 4: the LUT values 4 wave + 2, 4 wave + 3 (lhw: 4 wave, 4 wave + 1)
  uint32_t lhw;    // the LUT values 2 wave, 2 wave + 1 of the column: its own load (a select chain over the eight dwords by the wave
                   // index became a dynamically indexed stack object: scratch loads behind vmcnt(0) in the main loop)
  auto lpe = [&](int i) -> uint32_t { return i < 4 ? lpa[i & 3] : lpb[i & 3]; };
  const int tcol = tid & 63;
  const bool lut_loaded = p.qtype == TG_Q_ANY4_GLOBAL || p.qtype == TG_Q_ANY4_ROWWISE;
  auto lut_request = [&](const Item& e) {
    const int lrow = min(e.rb * 64 + tcol, p.wrows - 1);
    const char* lsrc = p.lut + (int64_t)e.b * p.stride_lut + (p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)lrow * 32 : 0);
    lpa = reinterpret_cast<const u32x4*>(lsrc)[0];
    lpb = reinterpret_cast<const u32x4*>(lsrc)[1];
    if constexpr (WAVES == 4) {
      lhw = reinterpret_cast<const uint32_t*>(lsrc)[2 * wave];
      lhw2 = reinterpret_cast<const uint32_t*>(lsrc)[2 * wave + 1];
    } else {
      lhw = reinterpret_cast<const uint32_t*>(lsrc)[WAVES == 8 ? wave : wave >> 1];
    }
  };
  // mx4: the 16 exponent bytes of row 16 t + (lane & 15) over this wave's slice, tile t = 0 ... 3, current and next item
  u32x4 ecur[4], enext[4];
  auto e_request = [&](const Item& e, u32x4 (&dst)[4]) {
    const char* eb = p.qinfo + (int64_t)e.b * p.stride_qinfo + (uint32_t)(wave * NCH);
#pragma unroll
    for (int t = 0; t < 4; ++t) dst[t] = *reinterpret_cast<const u32x4*>(eb + (int64_t)(e.rb * 64 + 16 * t + ln) * p.ngroups);
  };
  // 2^(e - 127) as f32 bits-wise: e << 23, e = 0 -> 2^-127 (a denormal), e = 255 -> NaN (Dequantization.cuh:331-339)
  auto e_scale = [](const u32x4& ev, int ci) -> float {
    const uint32_t e23 = __builtin_amdgcn_ubfe(ev[(ci >> 2) & 3], (uint32_t)((ci & 3) * 8), 8u) << 23;
    const float sc = u2f(e23 > 0x00400000u ? e23 : 0x00400000u);
    return __builtin_fmaf(sc, 0.f, sc);  // inf (e = 255) -> NaN, everything else unchanged
  };
  if constexpr (QMX) {
    e_request(cur, ecur);
  } else if (lut_loaded) {
    lut_request(cur);
  } else {  // int4: code - 8, exact
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const uint32_t v = DT::pack2((float)(e - 8), (float)(e - 7));
      if (e < 8) lpa[(e >> 1) & 3] = v;
      else lpb[(e >> 1) & 3] = v;
    }
    if constexpr (WAVES == 4) {
      lhw = DT::pack2((float)(4 * wave - 8), (float)(4 * wave - 7));
      lhw2 = DT::pack2((float)(4 * wave - 6), (float)(4 * wave - 5));
    } else {
      lhw = DT::pack2((float)(2 * (WAVES == 8 ? wave : wave >> 1) - 8), (float)(2 * (WAVES == 8 ? wave : wave >> 1) - 7));
    }
  }
  // table build: thread = (column, high nibbles 2 wave and 2 wave + 1); step a = low nibble a: entries (lut[a], lut[2 wave (+1)])
  auto build_step = [&](uint32_t buf, int a, uint32_t hw) {
    if constexpr (WAVES == 4) {  // thread = (column, high nibbles 4 wave ... 4 wave + 3): four entries per step
      const uint32_t lo = lpe(a >> 1);
      const uint32_t ls = (a & 1) ? 0x0302u : 0x0100u;
      const lds_u32ptr tb = (lds_u32ptr)((uint32_t)(wave * 4 * 16 * 256 + tcol * 4));
      tb[a * 64] = __builtin_amdgcn_perm(hw, lo, 0x05040000u | ls);
      tb[(16 + a) * 64] = __builtin_amdgcn_perm(hw, lo, 0x07060000u | ls);
      tb[(32 + a) * 64] = __builtin_amdgcn_perm(lhw2, lo, 0x05040000u | ls);
      tb[(48 + a) * 64] = __builtin_amdgcn_perm(lhw2, lo, 0x07060000u | ls);
      return;
    }
    if constexpr (WAVES == 16) {  // thread = (column, high nibble `wave`): one entry per step
      const uint32_t hsel = (wave & 1) ? 0x07060000u : 0x05040000u;
      const uint32_t e = __builtin_amdgcn_perm(hw, lpe(a >> 1), hsel | ((a & 1) ? 0x0302u : 0x0100u));
      const lds_u32ptr tb = (lds_u32ptr)(buf * TABLE + (uint32_t)(wave * 16 * 256 + tcol * 4));
      tb[a * 64] = e;
      return;
    }
    const uint32_t e0 = __builtin_amdgcn_perm(hw, lpe(a >> 1), (a & 1) ? 0x05040302u : 0x05040100u);
    const uint32_t e1 = __builtin_amdgcn_perm(hw, lpe(a >> 1), (a & 1) ? 0x07060302u : 0x07060100u);
    const lds_u32ptr tb = (lds_u32ptr)(buf * TABLE + (uint32_t)(wave * 2 * 16 * 256 + tcol * 4));
    tb[a * 64] = e0;
    tb[(16 + a) * 64] = e1;
  };

  // ---- activations of a problem: this lane's NCH pieces and the staged sums ----
  u32x4 xr[NXR];
  u32x4 xza = {0u, 0u, 0u, 0u};  // ZM: the A operand of the zero-point MFMA (this lane's slots of the split group sums)
  // x_prepare: straight from the caller's activations, no pre-pass and no workspace.  Lane (row i, k-quad kb) reads the four dwords
  // (k, k + 1), k = 32 c + 2 kb + {0, 8, 16, 24}, of row i and rearranges them into the "byte order" of the packed words
  // (w4_gemm_pair.cuh: x[2q], x[2q+8], x[2q+16], x[2q+24], x[2q+1], x[2q+9], x[2q+17], x[2q+25]); rows >= m are zero.  The
  // per-group sums of the wave's slice (its NCH / CPG groups) are formed from the same registers -- two-element dot products, the
  // group's chunks, then the four k-quads across lanes -- and written to LDS by the lanes of k-quad 0.  Runs once per PROBLEM and
  // workgroup, behind the barrier that ends the previous problem's last main loop.
  // The launch's FIRST problem (XLDS instantiations, row-major x) goes through LDS instead: the gather above is 64 four-byte loads per
  // lane whose wave-instruction touches 16 different 64-byte lines -- ~8 us of vector-memory issue per CU, nothing next to the 128
  // items of a stacked launch, a third of a launch with ONE item per workgroup (16384 x 4096 at 16 rows: 25 us per graph node).  There
  // the wave reads its slice of 8 rows as eight whole 1 KiB runs (16 bytes per lane, requested in front of the weight ring: `xg`),
  // writes them into its 8 KiB of table buffer 1 (free until the first item's second half) with the 16-byte piece index XORed by the
  // row -- lanes (row i, k-quad kq) of a 4-byte read then hit 8 x 4 distinct banks, rows 8 ... 15 of the OTHER round read their
  // partner's address (a broadcast) --, and picks its dwords up from there: rows 0 ... 7, then rows 8 ... 15.  No barrier: the
  // region is the wave's own and LDS operations of a wave execute in order.
#ifndef TG_XR_XLDS
#define TG_XR_XLDS 1
#endif
#ifndef TG_XR_XT
#define TG_XR_XT 1   // the later problems' activations by 16-byte loads + a lane transpose (0: four 4-byte loads per chunk; developer A/B)
#endif
  constexpr bool XLDS = TG_XR_XLDS && !PK && !QMX && WV == 8 && NCH == 16;  // (mx4: its eight-deep ring plus the 64 staging registers spill)
  u32x4 xg[XLDS ? 16 : 1];
  auto x_prepare = [&](int b, auto STAGED) {
    constexpr bool staged = decltype(STAGED)::value;
    const char* xb = p.x + (int64_t)b * p.stride_x;
    // (the lane id read HERE, opaque: derived from the kernel's `tid` the 64 per-lane offsets below are loop-invariant, get hoisted
    //  in front of the item loop and spilled -- 500 bytes of scratch)
    uint32_t lane_p;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_p));
    const int li = (int)(lane_p & 15u);
    const int lrow = PK ? (li & 7) : li;             // activation row of this lane
    const int par = PK ? (li >> 3) : 0;              // PK: which chunk of a pair this lane holds
    const int xi = min(lrow, p.m - 1);
    const bool on = lrow < p.m;
    const int kq = (int)(lane_p >> 4);
    float gsum = 0.f;
    if constexpr (ZM) xza = u32x4{0u, 0u, 0u, 0u};
    if constexpr (staged) {
      const uint32_t stg = TABLE + (uint32_t)wave * 8192u;
      const uint32_t il = (uint32_t)li & 7u;
      const uint32_t rb = stg + il * 1024u + (uint32_t)kq * 4u, sw = il << 4;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *(lds_u32x4ptr)(stg + (uint32_t)(j * 1024) + ((lane_p ^ (uint32_t)j) << 4)) = xg[8 * r + j];
        const bool act = (li >> 3) == r;
#pragma unroll
        for (int cx = 0; cx < NXR; ++cx)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t v = *(lds_cu32ptr)((rb + (uint32_t)((4 * cx + e) << 4)) ^ sw);
            xr[cx][e] = (r == 0 || act) ? v : xr[cx][e];
          }
      }
    }
#pragma unroll
    for (int cx = 0; cx < NXR; ++cx) {
      const int ci = PK ? 2 * cx + par : cx;         // (PK: run-time in the lane, the register index stays a constant)
      const int k0 = (wave * NCH + ci) * 32 + 2 * kq;
      uint32_t d[4];
      if constexpr (staged) {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = on ? xr[cx][e] : 0u;
      } else if (!TG_XR_XT || p.x_tc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t idx = p.x_tc ? tc_a_index(xi, k0 + 8 * e, p.k >> 4) : (int64_t)xi * p.k + k0 + 8 * e;
          const uint32_t v = *reinterpret_cast<const uint32_t*>(xb + idx * 2);
          d[e] = on ? v : 0u;
        }
      } else {
        // row-major x: the lane's dwords kq, kq + 4, kq + 8, kq + 12 of the chunk's 64 bytes as ONE 16-byte load of dwords 4 kq ... 4 kq + 3 and
        // a 4 x 4 transpose across the four 16-lane rows (tg_common.cuh: transpose_rows4) -- a quarter of the gather's requests (round 6)
        const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((int64_t)xi * p.k + (wave * NCH + ci) * 32) * 2 + 16 * kq);
        const u32x4 vt = transpose_rows4(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = on ? vt[e] : 0u;
      }
      xr[cx] = u32x4{__builtin_amdgcn_perm(d[1], d[0], 0x05040100u), __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                     __builtin_amdgcn_perm(d[1], d[0], 0x07060302u), __builtin_amdgcn_perm(d[3], d[2], 0x07060302u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) gsum = dot2_ones<DT>(d[e], gsum);
      if constexpr (!QMX) {  // (mx4 has no zero point: no sums)
        // chunks of the slice seen so far: all of them up to cx (unpacked) / the pairs up to cx (packed)
        constexpr int CPL = PK ? (CPG > 1 ? CPG / 2 : 1) : CPG;  // register sets per group in this lane
        if (cx % CPL == CPL - 1) {  // (compile-time) the group is complete: add the other k-quads' (and the other parity's) shares
          gsum = tgl::rows16_sum(gsum);  // (row swaps / a DPP rotation: tg_common.cuh)
          gsum = tgl::halves32_sum(gsum);
          if constexpr (PK && CPG > 1) gsum += tgl::lane_xor<8>(gsum, (int)lane_p);
          if constexpr (ZM) {
            // every lane of row i holds the group's sum now: its three 16-bit parts into this lane's slots (k-quad kq)
            const int gidx = cx / CPL;   // (a constant after unrolling; packed rows: cx counts chunk PAIRS)
            const uint16_t ph = DT::from_f32(gsum);
            const float r1 = gsum - DT::lo_f32(ph);
            const uint16_t pm = DT::from_f32(r1);
            const uint16_t pl = DT::from_f32(r1 - DT::lo_f32(pm));
            const uint16_t part[3] = {ph, pm, pl};
#pragma unroll
            for (int pp = 0; pp < 3; ++pp) {
              const int kbs = pp / ZR, rep = pp % ZR;       // (compile-time after unrolling)
              const int j = rep * NG + gidx, dw = j >> 1;
              const uint32_t ins = (j & 1) ? ((xza[dw] & 0x0000ffffu) | ((uint32_t)part[pp] << 16)) : ((xza[dw] & 0xffff0000u) | (uint32_t)part[pp]);
              xza[dw] = kq == kbs ? ins : xza[dw];
            }
            gsum = 0.f;
            continue;
          }
          // the group this lane's sum belongs to: packed with one chunk per group, the two halves of a row hold DIFFERENT groups
          const int cg = (PK && CPG == 1) ? 2 * cx + par : (PK ? 2 * cx : cx);
          const uint32_t g = (uint32_t)(((wave * NCH + cg) * 32) >> p.gshift);
          if constexpr (PK && CPG == 1) {
            if (lane_p < 16u) *(lds_fptr)(lds_xs + (g * 16u + (uint32_t)lrow) * 4u) = gsum;
          } else {
            // (packed: rows 8 ... 15 of the staged sums are written as zeros -- they feed accumulator rows that are never stored)
            if (lane_p < 16u) *(lds_fptr)(lds_xs + (g * 16u + lane_p) * 4u) = (PK && lane_p >= 8u) ? 0.f : gsum;
          }
          gsum = 0.f;
        }
      }
    }
  };

  // ---- prologue: first item's table (buffer 0) and activations, the first R super-tiles ----
  Rows rcur = rows_of(cur);
  bool x_staged = false;
  if constexpr (XLDS) {
    x_staged = !p.x_tc;  // (wave-uniform)
    if (x_staged) {      // the first problem's activations, in front of the ring: vector memory returns in request order
      const char* xs = p.x + (int64_t)cur.b * p.stride_x + (int64_t)(wave * NCH * 32) * 2 + lane * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j) xg[j] = *reinterpret_cast<const u32x4*>(xs + (int64_t)min(j, p.m - 1) * p.k * 2);
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    __builtin_amdgcn_sched_barrier(0);
    issue(rcur, j, ring[j], true);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!QMX) {
    const uint32_t hw = lhw;
#pragma unroll
    for (int a = 0; a < 16; ++a) build_step(0u, a, hw);
  }
  if constexpr (XLDS) {
    if (x_staged) x_prepare(cur.b, std::true_type{});
    else x_prepare(cur.b, std::false_type{});
  } else {
    x_prepare(cur.b, std::false_type{});
  }
  __syncthreads();

  // lookup address = byte << 8 | column << 2 | buffer << 16: one v_perm_b32 of the word with (column << 2 | buffer << 8)
  uint32_t colreg[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) colreg[u] = (uint32_t)((32 * u + 16 * lb + ln) * 4);
  uint32_t buf = 0u;
  f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int it = it_begin; it < it_end; ++it) {
    const bool has_next = it + 1 < it_end;
    Item inext = cur;
    if (has_next) {
      inext.rb = cur.rb + 1;
      if (inext.rb == p.rblocks) { inext.rb = 0; inext.b = cur.b + 1; }
    }
    Rows rnext = rows_of(inext);
    if constexpr (QMX) e_request(inext, enext);
    else if (lut_loaded) lut_request(inext);  // (the last item asks for its own rows again)
    const int row0 = cur.rb * 64;

    f32x4 acc[4];
    float yacc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) yacc[t][r] = 0.f;
    }
    uint32_t gq[4] = {0u, 0u, 0u, 0u};  // scale | zero of the current group, packed as loaded
    // a finished group gi of tile pair u: y += scale * acc + zero * sum(x); the sums come from LDS here (two reads per group)
    uint32_t zb[4][(NG + 1) / 2 > 0 ? (NG + 1) / 2 : 1];  // ZM: zero[gidx] of this lane's row in tile t, two groups per register
    auto finalize_pair = [&](int u, int gi) {
      if constexpr (ZM) {
#pragma unroll
        for (int t = 2 * u; t < 2 * u + 2; ++t) {
          const float gs = DT::lo_f32(gq[t]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            yacc[t][r] = __builtin_fmaf(gs, acc[t][r], yacc[t][r]);
            asm volatile("" : "+v"(yacc[t][r]));  // (kept HERE: without a memory operation in the update hipcc sinks it behind the loop and
                                                   //  keeps every group's accumulators alive: 300-600 bytes of scratch)
          }
          // the group's zero for the MFMA behind the loop: low half of the pair register for an even group, high half for an odd one
          zb[t][gi >> 1] = (gi & 1) ? __builtin_amdgcn_perm(gq[t], zb[t][gi >> 1], 0x07060100u) : (gq[t] >> 16);
        }
        return;
      }
      const uint32_t g = (uint32_t)(((wave * NCH + gi * CPG) * 32) >> p.gshift);
      const f32x4 xsv = *(lds_cf32x4ptr)(lds_xs + uni(g * 64u) + (uint32_t)((lane >> 4) * 16));
#pragma unroll
      for (int t = 2 * u; t < 2 * u + 2; ++t) {
        const float gs = DT::lo_f32(gq[t]), gz = DT::hi_f32(gq[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          yacc[t][r] = __builtin_fmaf(gz, xsv[r], __builtin_fmaf(gs, acc[t][r], yacc[t][r]));
          // (one table: no LDS store is left in the main loop, and without one the compiler sinks every group's update -- its LDS
          //  read of the sums first -- behind the loop and keeps all the groups' accumulators alive until then: 500 bytes of scratch)
          if constexpr (ONE_TABLE) asm volatile("" : "+v"(yacc[t][r]));
        }
      }
    };
    uint32_t hw = 0u;

    // ---- main loop, fully unrolled (every register index is a constant).  Stage = (chunk ci, tile pair u): 8 table lookups, one
    // swap per looked-up register pair, two MFMAs.  Software-pipelined by hand: the lookups of stage st + 1 are issued before stage st
    // is consumed (a workgroup has the CU to itself, two waves per SIMD: nobody else hides the LDS latency); lgkmcnt counts to 15, so
    // one stage of 8 reads ahead is what the counter can express.
    constexpr int NSTG = 2 * NCH;
    u32x4 pv[2], pw[2];  // looked-up registers of two stages: V (quad 2 a), W (quad 2 a + 1)
    auto look = [&](int st) {
      const int ci = st >> 1, u = st & 1, l = ci / CPS, c = ci % CPS;
      const Slot& sl = ring[l % R];
      const uint32_t wv = sl.w[u][c], ww = sl.w[u][CPS + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t av = __builtin_amdgcn_perm(wv, colreg[u], 0x0c010400u + ((uint32_t)j << 8));
        const uint32_t aw = __builtin_amdgcn_perm(ww, colreg[u], 0x0c010400u + ((uint32_t)j << 8));
        if constexpr (XR_ABL == 1) { pv[st & 1][j] = av; pw[st & 1][j] = aw; }  // ablation: no lookups
        else { pv[st & 1][j] = *(lds_cu32ptr)(av); pw[st & 1][j] = *(lds_cu32ptr)(aw); }
      }
    };
    constexpr bool AHEAD = WAVES != 16;  // (sixteen waves: four per SIMD hide the lookup latency; one stage's registers less)
    u32x4 xodd;                         // PK: the odd chunk's A operand (the register set rotated by 8 lanes within every 16-lane row)
    auto x_of = [&](int ci, int u) -> u32x4 {
      if constexpr (!PK) return xr[ci];
      else {
        if ((ci & 1) == 0) return xr[ci >> 1];
        if (u == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) xodd[j] = (uint32_t)__builtin_amdgcn_mov_dpp((int)xr[ci >> 1][j], 0x128 /* row_ror:8 */, 0xf, 0xf, false);
        }
        return xodd;
      }
    };
    if constexpr (!QMX && AHEAD) look(0);
    xr_static_for<NSTG>([&](auto ST) {
      constexpr int st = decltype(ST)::value;
      constexpr int ci = st >> 1, u = st & 1, l = ci / CPS, c = ci % CPS;
      Slot& sl = ring[l % R];
      if constexpr (QMX) {
        // mx4 stage: swap the lane's word pair into (tile 2 u, tile 2 u + 1) at quad kb, convert each word with its row's scale
        // of this chunk, two MFMAs into accumulators that run through the whole slice
        const auto sw = __builtin_amdgcn_permlane16_swap(sl.w[u][c], sl.w[u][CPS + c], false, false);
        const u32x4 b0 = mx4_cvt_word(sw[0], e_scale(ecur[2 * u], ci));
        const u32x4 b1 = mx4_cvt_word(sw[1], e_scale(ecur[2 * u + 1], ci));
        acc[2 * u] = DT::mfma(xr[ci], b0, acc[2 * u]);
        acc[2 * u + 1] = DT::mfma(xr[ci], b1, acc[2 * u + 1]);
        if (u == 1 && c == CPS - 1) {
          if (l + R < NST) issue(rcur, l + R, sl, true);
          else issue(rnext, l + R - NST, sl, has_next);
        }
        return;
      }
      const bool gfirst = ci % CPG == 0;
      const int gi = ci / CPG;
      if constexpr (AHEAD) {
        if (st + 1 < NSTG) look(st + 1);
      } else {
        look(st);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (gfirst) {
        if (ci > 0) finalize_pair(u, gi - 1);  // the previous group of this pair's tiles, behind the next stage's lookups
        const int gg = GPS == 1 ? 0 : c / CPG;
#pragma unroll
        for (int t = 2 * u; t < 2 * u + 2; ++t) gq[t] = sl.q[t][gg];
      }
      {
        u32x4 b0, b1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (XR_ABL == 9) {  // ablation: no lane swaps (wrong operands: timing only)
            b0[j] = pv[st & 1][j];
            b1[j] = pw[st & 1][j];
            continue;
          }
          const auto sw = __builtin_amdgcn_permlane16_swap(pv[st & 1][j], pw[st & 1][j], false, false);
          b0[j] = sw[0];
          b1[j] = sw[1];
        }
        const u32x4 xa = x_of(ci, u);
        acc[2 * u] = DT::mfma(xa, b0, gfirst ? zero4 : acc[2 * u]);
        acc[2 * u + 1] = DT::mfma(xa, b1, gfirst ? zero4 : acc[2 * u + 1]);
      }
      if (u == 1) {
        // the next item's table, one step per chunk of the slice's second half
        if (!ONE_TABLE && ci >= NCH / 2 && XR_ABL != 5) {
          if (ci == NCH / 2) hw = lhw;
          // the 16 build steps spread evenly over the NCH / 2 chunks of the half (k = 4096: two per chunk, 8192: one, 14336: 16 over 28)
          constexpr int J = ci - NCH / 2, S0 = (J * 16) / (NCH / 2), S1 = ((J + 1) * 16) / (NCH / 2);
#pragma unroll
          for (int e = S0; e < S1; ++e) build_step(buf ^ 1u, e, hw);
        }
        // refill: position l + R of this item, or of the next one (the stage's lookups of this slot were issued a stage ago)
        if (c == CPS - 1) {
          if (l + R < NST) issue(rcur, l + R, sl, true);
          else issue(rnext, l + R - NST, sl, has_next);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (QMX) {  // the slice's sums as they are; the next item's exponents become the current ones
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) yacc[t][r] = acc[t][r];
        ecur[t] = enext[t];
      }
    } else {
      finalize_pair(0, NCH / CPG - 1);
      finalize_pair(1, NCH / CPG - 1);
      if constexpr (ZM) {
        // y[a][row] += sum over (part, group) of X_part[gidx][a] zero[gidx][row]: one MFMA per tile.  Slot j of a k-quad holds group
        // j % NG for j < ZR NG: with an even NG a B dword is one of the zb registers as it stands
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          u32x4 bz;
#pragma unroll
          for (int dw = 0; dw < 4; ++dw) {
            const int j0 = 2 * dw, j1 = 2 * dw + 1;
            const bool v0 = j0 < ZR * NG, v1 = j1 < ZR * NG;
            const int g0 = j0 % NG, g1 = j1 % NG;
            if (!v0 && !v1) bz[dw] = 0u;
            else if (v0 && v1 && (g0 & 1) == 0 && g1 == g0 + 1) bz[dw] = zb[t][g0 >> 1];
            else {
              // low half = zero[g0] (half g0 & 1 of zb[g0 / 2]), high half = zero[g1]; an invalid slot = 0 (selector byte 0x0c)
              const uint32_t lo_sel = v0 ? ((g0 & 1) ? 0x0302u : 0x0100u) : 0x0c0cu;
              const uint32_t hi_sel = v1 ? ((g1 & 1) ? 0x0706u : 0x0504u) : 0x0c0cu;
              bz[dw] = __builtin_amdgcn_perm(zb[t][v1 ? g1 >> 1 : 0], zb[t][v0 ? g0 >> 1 : 0], (hi_sel << 16) | lo_sel);
            }
          }
          f32x4 yv = {yacc[t][0], yacc[t][1], yacc[t][2], yacc[t][3]};
          yv = DT::mfma(xza, bz, yv);
#pragma unroll
          for (int r = 0; r < 4; ++r) yacc[t][r] = yv[r];
        }
      }
    }

    const bool new_problem = has_next && inext.b != cur.b;

    // ---- split-K tail: the partial sums of the 8 waves meet in LDS (over the finished table) and are added in wave order ----
    if constexpr (XR_ABL == 6) {  // ablation: no split-K tail (one barrier per item, nothing stored)
      __syncthreads();
      if (yacc[0][0] == 123.f) *reinterpret_cast<float*>(p.y) = yacc[1][1] + yacc[2][2] + yacc[3][3];
      if (new_problem) x_prepare(inext.b, std::false_type{});
      rcur = rnext; cur = inext; buf ^= 1u; colreg[0] ^= 0x100u; colreg[1] ^= 0x100u;
      continue;
    }
    // The tail's per-thread indices are re-derived from a lane id read HERE (v_mbcnt, opaque): derived from the kernel's `tid` they
    // are loop-invariant, get hoisted in front of the item loop, spilled around the 256-register main loop and reloaded from
    // scratch behind s_waitcnt vmcnt(0) -- which drains the weight ring once per item.
    uint32_t lane_t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    const int tid_t = wave * 64 + (int)lane_t;
    if constexpr (ONE_TABLE) {
      // ---- WV = 4 (at most 8 rows): the sums of rows 0 ... 7 sit in lanes 0 ... 31 (k-quads 0, 1); handed over through a region of
      // their own, [wave][r][t][32 lanes rotated by 16 t] f32 = 8 KiB, so that the table can be rebuilt while they are summed ----
      const uint32_t lds_dump = lds_xs + (uint32_t)p.ngroups * 64u;
      if (lane_t < 32u) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const uint32_t pl = lds_dump + (uint32_t)(wave * 2048) + ((lane_t + 16u * (uint32_t)t) & 31u) * 4u;
#pragma unroll
          for (int r = 0; r < 4; ++r) *(lds_fptr)(pl + (uint32_t)((r * 4 + t) * 128)) = yacc[t][r];
        }
      }
      __syncthreads();  // every wave is done with the table and has handed over its sums
      {
        // wave W: activation rows W and W + 4 (register r = W of k-quads 0 and 1), lane = weight row of the item: four 4-byte reads
        // per output in k-slice order, one 128-byte line of y per wave and row
        const int t2 = (int)(lane_t >> 4), n2 = (int)(lane_t & 15u);
        const uint32_t pa = lds_dump + (uint32_t)(((wave * 4 + t2) * 32 + ((n2 + 16 * t2) & 31)) * 4);
        const uint32_t pb = lds_dump + (uint32_t)(((wave * 4 + t2) * 32 + ((16 + n2 + 16 * t2) & 31)) * 4);
        f32x2 a0, a1, b0, b1;
        asm volatile(
            "ds_read2st64_b32 %0, %4 offset1:8\n\t"
            "ds_read2st64_b32 %1, %4 offset0:16 offset1:24\n\t"
            "ds_read2st64_b32 %2, %5 offset1:8\n\t"
            "ds_read2st64_b32 %3, %5 offset0:16 offset1:24\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1)
            : "v"(pa), "v"(pb)
            : "memory");
        const float sums[2] = {((a0[0] + a0[1]) + a1[0]) + a1[1], ((b0[0] + b0[1]) + b1[0]) + b1[1]};
        char* yb = p.y + (int64_t)cur.b * p.stride_y;
        const int row = row0 + (int)lane_t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int a2 = wave + 4 * h;
          if (a2 < p.m && (XR_ABL != 7 || sums[h] == 123.456f)) {
            uint16_t oa = DT::from_f32(sums[h]);
            if (p.bias)
              oa = DT::from_f32(DT::lo_f32(oa) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + (int64_t)cur.b * p.stride_bias + ((int64_t)a2 * p.bias_row_stride + row) * 2)));
            *reinterpret_cast<uint16_t*>(yb + (p.y_tc ? tc_a_index(a2, row, p.y_tiles) : (int64_t)a2 * p.wrows + row) * 2) = oa;
          }
        }
      }
      // the next item's table (its LUT rows were requested when this item started) and, at a problem boundary, its activations
      if (has_next && XR_ABL != 5) {
        const uint32_t hwn = lhw;
#pragma unroll
        for (int a = 0; a < 16; ++a) build_step(0u, a, hwn);
      }
      if (new_problem) x_prepare(inext.b, std::false_type{});
      __syncthreads();  // table, sums and hand-over region are free / ready for the next item
      rcur = rnext;
      cur = inext;
      continue;
    }
    if constexpr (!QMX) __syncthreads();  // every wave is done with this item's table; the next item's table is complete (mx4: no table)
    const uint32_t lds_red = QMX ? 0u : buf * TABLE;
    if constexpr (WAVES == 8) {
      // dump layout [wave][r][t][64 lanes, rotated by 16 t]: the tail below reads whole 64-row lines of one activation row with
      // 32 / 64 lanes at a time, and the rotation puts the four tiles' 64-byte runs on different banks
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t pl = lds_red + (uint32_t)(wave * 4096) + ((lane_t + 16u * (uint32_t)t) & 63u) * 4u;
#pragma unroll
        for (int r = 0; r < 4; ++r) *(lds_fptr)(pl + (uint32_t)((r * 4 + t) * 256)) = yacc[t][r];
      }
    } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) *(lds_fptr)(lds_red + (uint32_t)((((wave * 4 + t) * 4 + r) * 64 + (int)lane_t) * 4)) = yacc[t][r];
    }
    if (new_problem) x_prepare(inext.b, std::false_type{});  // (every wave is behind its last use of the old registers and sums)
    __syncthreads();
    {
      // this thread's two outputs o = tid, tid + 512: (tile t, register r, lane l) -> activation row a, weight row; their eight
      // partial sums each: eight ds_read2st64_b32 issued together, spelled out (at the kernel's register limit the compiler's own
      // schedule was a dependent read -> wait -> add round per partial sum: sixteen LDS round trips with the CU otherwise idle)
      const int l = tid_t & 63, q0 = tid_t >> 6;  // (o + 512 >> 6 = q0 + 8: tile t + 2, the same r and l)
      const int t0 = q0 >> 2, r = q0 & 3;
      const int a = r + 4 * (l >> 4);
      const uint32_t pa = lds_red + (uint32_t)(((t0 * 4 + r) * 64 + l) * 4);
      f32x2 v0, v1, v2, v3, w0, w1, w2, w3;
      if constexpr (WAVES == 16) {
        // 1024 threads, ONE output each (tile t0 = q0 >> 2 covers all four tiles): its sixteen partial sums, 4096 bytes apart
        asm volatile(
            "ds_read2st64_b32 %0, %8 offset1:16\n\t"
            "ds_read2st64_b32 %1, %8 offset0:32 offset1:48\n\t"
            "ds_read2st64_b32 %2, %8 offset0:64 offset1:80\n\t"
            "ds_read2st64_b32 %3, %8 offset0:96 offset1:112\n\t"
            "ds_read2st64_b32 %4, %8 offset0:128 offset1:144\n\t"
            "ds_read2st64_b32 %5, %8 offset0:160 offset1:176\n\t"
            "ds_read2st64_b32 %6, %8 offset0:192 offset1:208\n\t"
            "ds_read2st64_b32 %7, %8 offset0:224 offset1:240\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
            : "v"(pa)
            : "memory");
        float sum = ((((((v0[0] + v0[1]) + v1[0]) + v1[1]) + v2[0]) + v2[1]) + v3[0]) + v3[1];
        sum = ((((((((sum + w0[0]) + w0[1]) + w1[0]) + w1[1]) + w2[0]) + w2[1]) + w3[0]) + w3[1]);
        char* yb = p.y + (int64_t)cur.b * p.stride_y;
        const int row = row0 + 16 * t0 + (l & 15);
        if (a < p.m && (XR_ABL != 7 || sum == 123.456f)) {
          uint16_t o16 = DT::from_f32(sum);
          if (p.bias)
            o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + (int64_t)cur.b * p.stride_bias + ((int64_t)a * p.bias_row_stride + row) * 2)));
          *reinterpret_cast<uint16_t*>(yb + (p.y_tc ? tc_a_index(a, row, p.y_tiles) : (int64_t)a * p.wrows + row) * 2) = o16;
        }
      } else {
        // 512 threads; every wave stores WHOLE 128-byte lines of y (the 64 weight rows of the item in one activation row): a partial
        // line costs the memory side what a whole one does, and the tile-per-wave mapping this replaces wrote 64 32-byte pieces per
        // item -- m = 4 / 8 / 16 differed by nothing but these stores (same-box 73.2 / 71.7 / 69.0 %).  The sums are the same additions
        // in the same (wave) order as before: the same bits.
        char* yb = p.y + (int64_t)cur.b * p.stride_y;
        if (p.m > 8) {
          // wave W: activation rows 2 W, 2 W + 1; lane: row pair rp = lane & 31 of row a = 2 W + (lane >> 5): two adjacent weight
          // rows = adjacent lanes of the dump = one 8-byte LDS read per k-slice, one 4-byte store
          const int rp = (int)(lane_t & 31u), a2 = 2 * wave + (int)(lane_t >> 5);
          const int t2 = rp >> 3, n2 = 2 * (rp & 7), r2 = a2 & 3, kb2 = a2 >> 2;
          const uint32_t pa2 = lds_red + (uint32_t)(((r2 * 4 + t2) * 64 + ((16 * kb2 + n2 + 16 * t2) & 63)) * 4);
          f32x4 q0v, q1v, q2v, q3v;  // (k-slice w: [0], [1] = the two rows; slice w + 1: [2], [3])
          asm volatile(
              "ds_read2st64_b64 %0, %4 offset1:8\n\t"
              "ds_read2st64_b64 %1, %4 offset0:16 offset1:24\n\t"
              "ds_read2st64_b64 %2, %4 offset0:32 offset1:40\n\t"
              "ds_read2st64_b64 %3, %4 offset0:48 offset1:56\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(q0v), "=&v"(q1v), "=&v"(q2v), "=&v"(q3v)
              : "v"(pa2)
              : "memory");
          const float sa = ((((((q0v[0] + q0v[2]) + q1v[0]) + q1v[2]) + q2v[0]) + q2v[2]) + q3v[0]) + q3v[2];
          const float sb = ((((((q0v[1] + q0v[3]) + q1v[1]) + q1v[3]) + q2v[1]) + q2v[3]) + q3v[1]) + q3v[3];
          const int row = row0 + 2 * rp;  // (even; the host guarantees wrows % 64 == 0)
          if (a2 < p.m && p.y_f32) {  // a k-window of a longer contraction: the unrounded sums
            *reinterpret_cast<f32x2*>(yb + ((int64_t)a2 * p.wrows + row) * 4) = f32x2{sa, sb};
          } else if (a2 < p.m && (XR_ABL != 7 || sa == 123.456f)) {  // (ablation 7: no output stores)
            uint16_t oa = DT::from_f32(sa), ob = DT::from_f32(sb);
            if (p.bias) {  // rounded sum + bias, rounded again: bit-identical to the reference module's separate `y + bias`
              const uint32_t bv = *reinterpret_cast<const uint32_t*>(p.bias + (int64_t)cur.b * p.stride_bias + ((int64_t)a2 * p.bias_row_stride + row) * 2);
              oa = DT::from_f32(DT::lo_f32(oa) + DT::lo_f32(bv));
              ob = DT::from_f32(DT::lo_f32(ob) + DT::hi_f32(bv));
            }
            *reinterpret_cast<uint32_t*>(yb + (p.y_tc ? tc_a_index(a2, row, p.y_tiles) : (int64_t)a2 * p.wrows + row) * 2) = (uint32_t)oa | ((uint32_t)ob << 16);
          }
        } else {
          // at most 8 activation rows: wave W owns row W, lane = weight row of the item: eight 4-byte LDS reads, one 2-byte store per
          // lane = one 128-byte line per wave
          const int a2 = wave, t2 = (int)(lane_t >> 4), n2 = (int)(lane_t & 15u), r2 = a2 & 3, kb2 = a2 >> 2;
          const uint32_t pa2 = lds_red + (uint32_t)(((r2 * 4 + t2) * 64 + ((16 * kb2 + n2 + 16 * t2) & 63)) * 4);
          f32x2 v0, v1, v2, v3;
          asm volatile(
              "ds_read2st64_b32 %0, %4 offset1:16\n\t"
              "ds_read2st64_b32 %1, %4 offset0:32 offset1:48\n\t"
              "ds_read2st64_b32 %2, %4 offset0:64 offset1:80\n\t"
              "ds_read2st64_b32 %3, %4 offset0:96 offset1:112\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
              : "v"(pa2)
              : "memory");
          const float sa = ((((((v0[0] + v0[1]) + v1[0]) + v1[1]) + v2[0]) + v2[1]) + v3[0]) + v3[1];
          const int row = row0 + (int)lane_t;
          if (a2 < p.m && p.y_f32) {
            *reinterpret_cast<float*>(yb + ((int64_t)a2 * p.wrows + row) * 4) = sa;
          } else if (a2 < p.m && (XR_ABL != 7 || sa == 123.456f)) {
            uint16_t oa = DT::from_f32(sa);
            if (p.bias)
              oa = DT::from_f32(DT::lo_f32(oa) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + (int64_t)cur.b * p.stride_bias + ((int64_t)a2 * p.bias_row_stride + row) * 2)));
            *reinterpret_cast<uint16_t*>(yb + (p.y_tc ? tc_a_index(a2, row, p.y_tiles) : (int64_t)a2 * p.wrows + row) * 2) = oa;
          }
        }
      }  // WAVES == 8
    }
    if constexpr (XR_ABL != 8)  // (ablation 8: without this barrier -- a race, timing only)
    __syncthreads();  // the partial sums are consumed before the next item's build steps write into this buffer
    rcur = rnext;
    cur = inext;
    buf ^= 1u;
#pragma unroll
    for (int u = 0; u < 2; ++u) colreg[u] ^= 0x100u;
  }
}
