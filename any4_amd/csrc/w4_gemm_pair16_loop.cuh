// w4_gemm_pair16_loop.cuh -- w4_gemm_pair16_kernel's register-resident-activation path (XREG) as a LOOP over 16-row tiles: ONE layer per
// launch with 5 ... 16 activation rows and MORE 16-row tiles than compute units (4096 < rows, k = 4096, innerKTiles 4).
//
// Why: w4_gemm_pair16_kernel is one tile per workgroup.  Past 256 tiles its launch runs in rounds -- every workgroup of the second round
// pays the whole prologue again (128 KiB of activations through its CU's 64-byte-per-clock vector-memory path, the HBM latency of its
// first weights) behind a workgroup dispatch: 5120 rows at 16 rows 14.9 us per graph node against 7.0 at 4096.  The alternative so far
// was w4_gemm_xr_kernel with one workgroup per 64-row item: a 12.6 us floor (an item is 6.5 us of one CU) on as few as 80 CUs.
// Here a workgroup owns a contiguous range of tiles (like w4_gemv_kernel): the activations are loaded and arranged ONCE, tile i + 1's
// weights are requested before tile i is consumed (two register sets), its table is built from LUT rows staged in LDS into the other 32
// columns of the pair table, the split-K partial sums go through two alternating 16 KiB regions: one barrier per tile.
//
// Same contract and numerics as w4_gemm_pair16.cuh (TG_NUM_FAST, group-scaled; reference TinyGemmImpl.cuh:23-345 with BLayout_TC_int4,
// MatrixLayoutB.cuh:686-1101): lane (n, q) of a wave holds weight row n of the tile in the B operand; the A operand's lane i holds
// activation row 4 (i & 3) + (i >> 2) (quad-contiguous loads, w4_gemm_pair16.cuh XQ), so accumulator register r is activation row 4 r + q.
#pragma once

struct Pair16LoopParams {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  const char* bias;
  int64_t bias_row_stride;
  int32_t m, wrows, k;
  int32_t ntiles;        // packed.size(0): 8-row tiles
  int32_t ksuper;        // 64
  int32_t gshift, ngroups, qtype;
  int32_t tbase, trem;   // workgroup b owns tbase + (b < trem) 16-row tiles, starting at tile b * tbase + min(b, trem)
  int32_t lds_lut;       // LDS byte offset of the staged LUT rows (row-wise LUT, more than one tile per workgroup): 32 bytes per row, 544 per tile
  int32_t lds_red;       // ... of the two partial-sum regions, 16 KiB each
  int32_t epilogue;      // TG_EPI_SWIGLU: a tile is 8 gate + 8 up rows, y is [m][wrows / 2]
  const char* norm_w;    // fused LlamaRMSNorm of the activations (tg_w4_gemm.norm_weight), nullptr = off
  float norm_eps;
  int32_t lds_nrm;       // LDS byte offset of the sums of squares, f32 [16 waves][16 rows]
};

template <typename DT, int CPG, bool NORM>
__global__ void __launch_bounds__(1024) w4_gemm_pair16_loop_kernel(const Pair16LoopParams p) {
  constexpr int I = 4, CPS = I / 2, CH = 4, NCHK = CH * CPS;  // a wave's k-slice: four super-tiles = eight 32-k chunks (k = 4096 over 16 waves)
  static_assert(NCHK % CPG == 0, "a slice is whole groups");
  asm volatile("" ::"s"(p.x), "s"(p.w), "s"(p.qinfo), "s"(p.lut), "s"(p.y), "s"(p.bias), "s"(p.bias_row_stride), "s"(p.m), "s"(p.wrows), "s"(p.k),
               "s"(p.ntiles), "s"(p.ksuper), "s"(p.gshift), "s"(p.ngroups), "s"(p.qtype), "s"(p.tbase), "s"(p.trem), "s"(p.lds_lut), "s"(p.lds_red), "s"(p.epilogue), "s"(p.norm_w), "s"(p.norm_eps), "s"(p.lds_nrm));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int bx = blockIdx.x;
  const int t0 = bx * p.tbase + min(bx, p.trem);
  const int nt = p.tbase + (bx < p.trem ? 1 : 0);
  if (nt <= 0) return;
  const bool rowwise = p.qtype == TG_Q_ANY4_ROWWISE;
  const bool lutq = rowwise || p.qtype == TG_Q_ANY4_GLOBAL;

  // ---- requests: the first tile's LUT row, every tile's LUT rows (for LDS), the activations, the first tile's weights ----
  const int tcol = tid & 31;  // table column of a tile: 16 copy + row
  uint32_t lp[8];
  if (lutq) {
    const int trow = min(t0 * 16 + (tcol & 15), p.wrows - 1);
    const char* lsrc = p.lut + (rowwise ? (int64_t)trow * 32 : 0);
    const u32x4 l0 = reinterpret_cast<const u32x4*>(lsrc)[0];
    const u32x4 l1 = reinterpret_cast<const u32x4*>(lsrc)[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { lp[j] = l0[j]; lp[4 + j] = l1[j]; }
  } else {
#pragma unroll
    for (int e = 0; e < 16; e += 2) lp[e >> 1] = DT::pack2((float)(e - 8), (float)(e - 7));  // int4: code - 8, exact
  }
  const bool stage_lut = rowwise && nt > 1;  // (host: nt <= 32, so the 1024 threads hold one 16-byte half row each)
  u32x4 lstage = {0u, 0u, 0u, 0u};
  if (stage_lut) {
    const int idx = min(tid, nt * 32 - 1);
    const int row = min(t0 * 16 + (idx >> 1), p.wrows - 1);
    lstage = *reinterpret_cast<const u32x4*>(p.lut + (int64_t)row * 32 + (idx & 1) * 16);
  }
  // activations: lane 4 a + b reads quarter b of row a's 64-byte chunk (quad-contiguous), rows >= m read row m - 1 again
  uint32_t xoff = (uint32_t)((min(lane >> 2, p.m - 1) * p.k + 8 * (lane & 3)) * 2);
  u32x4 xf[NCHK];
#pragma unroll
  for (int c = 0; c < NCHK; ++c) {
    asm volatile("" : "+v"(xoff));
    xf[c] = *reinterpret_cast<const u32x4*>(p.x + (uint32_t)__builtin_amdgcn_readfirstlane((wave * NCHK + c) * 64) + xoff);
  }
  // fused LlamaRMSNorm (dg_add_rmsnorm's formula, as in w4_gemv.cuh): y = rs * sum_k w_k x'_k with x'_k = RNE16(x_k g_k) and rs = rsqrt(mean(x^2)
  // + eps) applied to the f32 sum in the output store.  The squares of this lane's 64 values -> the row's four lanes (one quad) -> one
  // partial per wave and row in LDS, added over the 16 waves by the threads that store the outputs.
  constexpr bool norm = NORM;
  // the norm weights of the whole k (8 KiB) through LDS: one 16-byte load per thread of the first 8 waves now, read back piece by piece behind
  // the prologue's barrier (32 registers of them in flight next to the activations spilled).  They borrow the SECOND partial-sum region:
  // it is first written behind the first tile's barrier, when every wave is through its norm block.
  u32x4 gstage = {0u, 0u, 0u, 0u};
  if constexpr (NORM) {
    if (tid < 512) gstage = reinterpret_cast<const u32x4*>(p.norm_w)[tid];
  }
  // weights of a tile: row n of the tile, quad q; the scale | zero word of every group of the slice
  struct W {
    uint32_t w[CH][CPS];
    uint32_t qv[NCHK / CPG];
  };
  auto w_request = [&](W& r, int tile) {
    const int wrow = min(tile * 16 + n, p.wrows - 1);
    const int t8 = min(wrow >> 3, p.ntiles - 1);
    const char* wl = p.w + ((int64_t)t8 * p.ksuper * 32 + (4 * (wrow & 7) + q)) * (2 * I);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wl + (int64_t)(wave * CH + j) * (64 * I)));
      r.w[j][0] = v[0]; r.w[j][1] = v[1];
    }
#pragma unroll
    for (int gi = 0; gi < NCHK / CPG; ++gi) {
      const int g = (((wave * NCHK + gi * CPG)) * 32) >> p.gshift;
      r.qv[gi] = *reinterpret_cast<const uint32_t*>(p.qinfo + ((int64_t)g * p.wrows + wrow) * 4);
    }
  };
  W wa, wb;
  w_request(wa, t0);

  // ---- pair table of a tile in columns 32 par ... 32 par + 31: thread = (column, high nibble (tid >> 5) & 15, half tid >> 9 of the low nibbles) ----
  auto build_table = [&](int par) {
    const int hi = (tid >> 5) & 15, half = tid >> 9;
    uint32_t hw = lp[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) hw = ((hi >> 1) == j) ? lp[j] : hw;
    const uint32_t hsel = (hi & 1) ? 0x07060000u : 0x05040000u;
    const uint32_t base = (uint32_t)((hi * 16 + half * 8) * 256 + (32 * par + tcol) * 4);
    const uint32_t hm = half ? 0xffffffffu : 0u;
    uint32_t lq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lq[j] = (lp[4 + j] & hm) | (lp[j] & ~hm);
#pragma unroll
    for (int a = 0; a < 8; ++a) ((lds_u32ptr)base)[a * 64] = __builtin_amdgcn_perm(hw, lq[a >> 1], hsel | ((a & 1) ? 0x0302u : 0x0100u));
  };
  build_table(0);
  // (a tile's 16 rows at a pitch of 544 bytes, rows 8 ... 15 shifted by 16 bytes: the 16-byte reads of next_table -- 16 rows per wave,
  //  32 bytes apart -- then touch 64 distinct banks instead of colliding two by two: 32768 bank-conflict cycles per launch before)
  if constexpr (NORM) {
    if (tid < 512) *(lds_u32x4ptr)((uint32_t)p.lds_red + 16384u + (uint32_t)tid * 16u) = gstage;
  }
  if (stage_lut && tid < nt * 32) {
    const uint32_t row = (uint32_t)tid >> 1, ti = row >> 4, i = row & 15u;
    *(lds_u32x4ptr)((uint32_t)p.lds_lut + ti * 544u + i * 32u + (i >> 3) * 16u + ((uint32_t)tid & 1u) * 16u) = lstage;
  }
  __syncthreads();

  // fused LlamaRMSNorm (dg_add_rmsnorm's formula, as in w4_gemv.cuh): y = rs * sum_k w_k x'_k with x'_k = RNE16(x_k g_k) and rs = rsqrt(mean(x^2)
  // + eps) applied to the f32 sum in the output store.  The squares of this lane's 64 values -> the row's four lanes (one quad) -> one
  // partial per wave and row in LDS, added over the 16 waves by the threads that store the outputs.
  if constexpr (NORM) {
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < NCHK; ++c) {
      const u32x4 gw = *(lds_cu32x4ptr)((uint32_t)p.lds_red + 16384u + (uint32_t)(((wave * NCHK + c) * 32 + 8 * (lane & 3)) * 2));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t xv = xf[c][e], g = gw[e];
        if constexpr (std::is_same<DT, BF16>::value) v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xv), __builtin_bit_cast(bf16x2, xv), v, false);
        else v = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, xv), __builtin_bit_cast(f16x2, xv), v, false);
        xf[c][e] = DT::pack2(DT::lo_f32(xv) * DT::lo_f32(g), DT::hi_f32(xv) * DT::hi_f32(g));
      }
    }
    v += tgl::lane_xor<1>(v, lane);
    v += tgl::lane_xor<2>(v, lane);
    if ((lane & 3) == 0) *(lds_fptr)((uint32_t)p.lds_nrm + (uint32_t)((wave * 16 + (lane >> 2)) * 4)) = v;  // (rows >= m: row m - 1 again, never read)
  }

  // ---- the A operands: a 4 x 4 dword transpose over the lane bits 4, 5 and two bit exchanges inside the quad (w4_gemm_pair16.cuh, XQ) ----
  {
    const bool l1 = (lane & 2) != 0, l0 = (lane & 1) != 0;
    auto qx = [&](uint32_t& x, uint32_t& y, auto CTRL, bool hi) {
      constexpr int ctrl = decltype(CTRL)::value;
      const uint32_t yx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, ctrl, 0xf, 0xf, false);
      const uint32_t xx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xf, 0xf, false);
      const uint32_t nx = hi ? yx : x, ny = hi ? y : xx;
      x = nx; y = ny;
    };
#pragma unroll
    for (int c = 0; c < NCHK; ++c) {
      const auto s02 = __builtin_amdgcn_permlane32_swap(xf[c][0], xf[c][2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(xf[c][1], xf[c][3], false, false);
      const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
      const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
      uint32_t d0 = t01[0], d1 = t01[1], d2 = t23[0], d3 = t23[1];
      qx(d0, d2, std::integral_constant<int, 0x4E>{}, l1);
      qx(d1, d3, std::integral_constant<int, 0x4E>{}, l1);
      qx(d0, d1, std::integral_constant<int, 0xB1>{}, l0);
      qx(d2, d3, std::integral_constant<int, 0xB1>{}, l0);
      xf[c] = u32x4{__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u),
                    __builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
    }
  }

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  const uint32_t one2 = DT::pack2(1.f, 1.f);
  const u32x4 ones = {one2, one2, one2, one2};
  uint32_t colreg = (uint32_t)((16 * (q & 1) + n) * 4);

  // one tile: 8 chunks x (4 lookups, the tile's MFMA, the activation sums' MFMA against ones), a scale / zero update per group
  auto consume = [&](const W& r, float (&yacc)[4]) {
    f32x4_t acc = zero4, xs = zero4;
#pragma unroll
    for (int u = 0; u < NCHK; ++u) {
      const uint32_t w = r.w[u / CPS][u % CPS];
      u32x4 bf;
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[e] = *(lds_cu32ptr)(__builtin_amdgcn_perm(w, colreg, 0x0c0c0400u + ((uint32_t)e << 8)));
      const bool gfirst = u % CPG == 0, glast = u % CPG == CPG - 1;
      acc = mfma16<DT>(xf[u], bf, gfirst ? zero4 : acc);
      xs = mfma16<DT>(xf[u], ones, gfirst ? zero4 : xs);
      if (glast) {
        const uint32_t qv = r.qv[u / CPG];
        const float gs = DT::lo_f32(qv), gz = DT::hi_f32(qv);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) yacc[rr] = __builtin_fmaf(gz, xs[rr], __builtin_fmaf(gs, acc[rr], yacc[rr]));
      }
      if (u % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the tile's split-K tail: partial sums -> region `par`, (the caller's barrier), 256 threads add the 16 waves in wave order and store
  auto dump = [&](const float (&yacc)[4], int par) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) *(lds_fptr)((uint32_t)p.lds_red + (uint32_t)(par * 16384 + ((wave * 4 + rr) * 64 + lane) * 4)) = yacc[rr];
  };
#ifndef TG_P16L_SPREAD
#define TG_P16L_SPREAD 0  // (developer A/B: no difference -- 8192 / 16384 rows 9.6-9.8 / 14.4-14.8 us either way, profiles/r05_ab_p16_loop_spread.txt; the waves 0 ... 3 keep the wave-order sum of w4_gemm_pair16_kernel)
#endif
  float rsn = NORM ? 0.f : 1.f;  // 1 / rms of this thread's activation row (set with the first tile's store)
  auto store = [&](int tile, int par) {
    if (TG_P16L_SPREAD && p.epilogue == 0 && !norm) {
      // every wave takes 16 of the tile's 256 outputs: lane = 16 part + o -- four partial sums per lane (waves 4 part ... 4 part + 3, in wave
      // order), the four parts across the lane rows by two row swaps (tg_common.cuh) -- instead of the waves 0 ... 3 reading 16 partial sums
      // per output while the other twelve run ahead to the next barrier and wait there
      const int o = wave * 16 + (lane & 15), part = lane >> 4;
      const int rr = o >> 6, l = o & 63;
      const int a = 4 * rr + (l >> 4), row = tile * 16 + (l & 15);
      const uint32_t src = (uint32_t)p.lds_red + (uint32_t)(par * 16384 + ((part * 16 + rr) * 64 + l) * 4);
      float sum = *(lds_fptr)(src);
#pragma unroll
      for (int i = 1; i < 4; ++i) sum += *(lds_fptr)(src + (uint32_t)(i * 4 * 64 * 4));
      sum = tgl::halves32_sum(tgl::rows16_sum(sum));
      if (lane < 16 && a < p.m && row < p.wrows) {
        uint16_t o16 = DT::from_f32(sum);
        if (p.bias)  // rounded sum + bias, rounded again: the reference module's separate `y + bias` (modules.py:221-222)
          o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + ((int64_t)a * p.bias_row_stride + row) * 2)));
        *reinterpret_cast<uint16_t*>(p.y + ((int64_t)a * p.wrows + row) * 2) = o16;
      }
      return;
    }
    if (tid < 256) {
      int lt = tid & 63;
      asm volatile("" : "+v"(lt));  // (opaque per tile: the 64-bit row offsets derived from it are loop-invariant, get hoisted and -- in the norm
                                     //  instantiations -- spilled: a scratch reload in every tile's store)
      const int rr = (tid >> 6) & 3, l = lt;
      const int a = 4 * rr + (l >> 4), row = tile * 16 + (l & 15);  // (the A operand's rows are rotated: register rr of lane (n, q) = row 4 rr + q)
      if (a < p.m && row < p.wrows) {
        float sum = 0.f;
#pragma unroll
        for (int w16 = 0; w16 < 16; ++w16) sum += *(lds_fptr)((uint32_t)p.lds_red + (uint32_t)(par * 16384 + ((w16 * 4 + rr) * 64 + l) * 4));
        if constexpr (NORM) {
          if (rsn == 0.f) {  // the first tile: this thread's activation row is the same for every tile
            float tot = 0.f;
#pragma unroll 4
            for (int w16 = 0; w16 < 16; ++w16) tot += *(lds_fptr)((uint32_t)p.lds_nrm + (uint32_t)((w16 * 16 + a) * 4));
            rsn = rsqrtf(tot * (1.0f / (float)p.k) + p.norm_eps);
          }
          sum *= rsn;
        }
        if (p.epilogue == TG_EPI_SWIGLU) {
          // the tile is one block of 8 gate + 8 up rows: lane l + 8 holds the up row of gate row l (w4_gemm_pair16.cuh, dg_swiglu's formula)
          if ((l & 15) < 8) {
            float up = 0.f;
#pragma unroll
            for (int w16 = 0; w16 < 16; ++w16) up += *(lds_fptr)((uint32_t)p.lds_red + (uint32_t)(par * 16384 + ((w16 * 4 + rr) * 64 + l + 8) * 4));
            up *= rsn;
            *reinterpret_cast<uint16_t*>(p.y + ((int64_t)a * (p.wrows >> 1) + tile * 8 + (l & 7)) * 2) = swiglu16<DT>(sum, up);
          }
        } else {
          uint16_t o16 = DT::from_f32(sum);
          if (p.bias)  // rounded sum + bias, rounded again: the reference module's separate `y + bias` (modules.py:221-222)
            o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + ((int64_t)a * p.bias_row_stride + row) * 2)));
          *reinterpret_cast<uint16_t*>(p.y + ((int64_t)a * p.wrows + row) * 2) = o16;
        }
      }
    }
  };
  // tile ti + 1's table into the other 32 columns, from the LUT rows staged in LDS (int4 / one global LUT: the columns never change)
  auto next_table = [&](int ti) {
    if (rowwise) {
      const uint32_t lrow = (uint32_t)p.lds_lut + (uint32_t)((ti + 1) * 544 + (tcol & 15) * 32 + ((tcol & 15) >> 3) * 16);
      const u32x4 l0 = *(lds_cu32x4ptr)(lrow), l1 = *(lds_cu32x4ptr)(lrow + 16u);
#pragma unroll
      for (int j = 0; j < 4; ++j) { lp[j] = l0[j]; lp[4 + j] = l1[j]; }
      build_table((ti + 1) & 1);
    }
  };

  for (int ti = 0; ti < nt; ti += 2) {
    float yacc[4];
    // ---- even tile: set A ----
    if (ti + 1 < nt) w_request(wb, t0 + ti + 1);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) yacc[rr] = 0.f;
    consume(wa, yacc);
    dump(yacc, 0);
    if (ti + 1 < nt) next_table(ti);
    __syncthreads();
    store(t0 + ti, 0);
    if (ti + 1 >= nt) break;
    if (rowwise) colreg ^= 128u;
    // ---- odd tile: set B ----
    if (ti + 2 < nt) w_request(wa, t0 + ti + 2);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) yacc[rr] = 0.f;
    consume(wb, yacc);
    dump(yacc, 1);
    if (ti + 2 < nt) next_table(ti + 1);
    __syncthreads();
    store(t0 + ti + 1, 1);
    if (rowwise) colreg ^= 128u;
  }
}
