// w4_gemm_pair16.cuh -- the pair-table W4A16 kernel for SMALL launches (one Any4Linear.forward: a single 4096 x 4096 layer is
// 64 work items of w4_gemm_pair_kernel -- a quarter of the chip, each wave walking 16 KiB through a two-deep ring).
//
// Same contract and numerics as w4_gemm_pair.cuh (TG_NUM_FAST, group-scaled), Bint4 weights (reference TinyGemmImpl.cuh:23-345
// with BLayout_TC_int4, MatrixLayoutB.cuh:686-1101, Dequantization.cuh:55-178, 331-351), other decomposition:
//
//   workgroup  = 16 weight rows (two 8-row tiles of the layout) x the whole k, 16 waves, split-K 16: a 4096-row layer is 256
//                workgroups, one per CU; nothing persistent, no ring: a wave requests its WHOLE k-slice (k = 4096: 4 super-tiles
//                = 8 packed words per lane) before it does anything else, so the launch costs one memory round trip, and the
//                dependent chain of a wave (lookup -> MFMA, 8 steps) is short.
//   MFMA       = v_mfma_f32_16x16x32: lane (n = lane & 15, q = lane >> 4) holds, in ONE packed word of the reference layout, the 8
//                codes of weight row n at k = 2 q + {0, 16, 1, 17, 8, 24, 9, 25} of a 32-k chunk (TinyGemmConvertB.cu:252-308) --
//                exactly one B operand.  A operand = activations, lane (row i = lane & 15, k-quad q): the 16-byte piece (chunk, q)
//                of the byte-order staging of w4_gemm_pair.cuh.  D[i][n]: lane (n, q) holds activation rows 4 q + r of ITS row.
//   table      = [256 byte values][64 columns] x 4 bytes at LDS address 0 as in the large kernel (address = byte << 8 | column
//                << 2, one v_perm_b32); only columns 16 (q & 1) + n are used: the two lanes of a 32-lane LDS access group that
//                share a weight row read different copies, a group touches 32 distinct banks.
//   activations= m <= 16 rows staged whole in LDS (byte order) with their per-group sums.
//   TPW        = 16-row MFMA tiles per workgroup.  Only 1 is instantiated: 2 (table columns 32 t + 16 (q & 1) + n, one X fragment
//                read per step for both tiles; would keep 6144-row layers at one round of workgroups) spills 80-150 registers at
//                the 128-VGPR budget of a 1024-thread workgroup; m = 1 launches of more than one round go to the stream kernel.
//   long k     = a slice of more than CH super-tiles per wave (k = 14336: 14) is walked in blocks of CH with the NEXT block's
//                words requested before the current one is consumed (two register sets).
//   XREG       = 5 ... 16 activation rows: the A operands never pass through LDS.  A wave only needs the activations of ITS k-slice:
//                lane (row i = lane & 15, quarter e = lane >> 4) loads the 16 bytes e of row i's 64-byte chunk (a wave-load = 16 rows
//                x 64 contiguous bytes, L2 hits after the first workgroup), a 4 x 4 dword transpose across the four lane quarters
//                (two v_permlane32_swap + two v_permlane16_swap) gives lane (i, q) the dwords q, q + 4, q + 8, q + 12 of the chunk,
//                four v_perm_b32 put them into the packed words' byte order: 8 vector ops per chunk instead of a trip through LDS --
//                and no LDS for activations at all, so 16 rows x any k run in ONE pass (the LDS path stages 16 x 4096 in two parts
//                with two workgroup barriers each; 16 x 14336 in eight).  Rows >= m load row m - 1 again (same addresses: no extra
//                traffic); their accumulator rows are never stored.  The per-group activation sums of the zero-point term come
//                from the matrix core as well: one more MFMA per chunk against an all-ones B operand leaves sum_k x[4 q + r][k] in
//                the accumulator layout of the lane that needs it (no shuffles, no LDS).
#pragma once
#ifndef P16_XSCHED
#define P16_XSCHED 2
#endif
#ifndef P16_ABL
#define P16_ABL 0  // developer ablations (0 in the product)
#endif

struct Pair16Params {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  int32_t m, wrows, k;
  int32_t ntiles;    // packed.size(0): 8-row tiles
  int32_t ksuper;    // packed.size(1)
  int32_t gshift, ngroups, qtype;
  int32_t phases;    // the k range is walked in `phases` equal parts (activation rows of one part staged at a time; 1: whole k)
  int32_t ksuper_p;  // super-tiles per phase
  int32_t spw;       // k super-tiles per wave and phase (whole groups)
  int32_t gch_mask;  // 32-k chunks per group - 1
  int32_t x_pitch;   // bytes per staged activation row (one phase of it)
  int32_t lds_x, lds_xs;  // LDS byte offsets: staged activations (m rows + a 16-byte zero piece), sums f32 [groups of a phase][16]
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y;
  const char* bias;
  int64_t stride_bias;
  int64_t bias_row_stride;  // elements between the bias rows of consecutive activation rows (0: one row for all; wrows: a residual)
  const char* norm_w;       // fused RMSNorm of the activations (tg_w4_gemm.norm_weight; phases == 1 only), nullptr = off
  float norm_eps;
  int32_t epilogue;         // TG_EPI_SWIGLU: rows in blocks of 8 gate + 8 up, y is [m][wrows / 2]
  int32_t x_tc, y_tc;  // 1: activations / output in A-fragment order (tc_a_index, w4_gemm_pair.cuh)
  int32_t y_tiles;     // ceil(wrows / 16)
#if GEMV_TRACE
  unsigned long long* trace;  // developer builds (-DGEMV_TRACE=1): [workgroup][8] s_memrealtime stamps
#endif
};
#if GEMV_TRACE
#define P16_STAMP(i) do { if (p.trace && tid == 0) tr[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define P16_STAMP(i) do { } while (0)
#endif

// I   = innerKTiles of the Bint4 layout (2, 4, 8): I / 2 words per lane and super-tile (one per 32-k chunk)
// CPG = 32-k chunks per quantisation group (1, 2, 4, 8): a full block of CH super-tiles then has its group boundaries at fixed
//       places of the unrolled code (no branches between the steps)
// XREG / CH: see the header; CH = super-tiles requested at once (4: k = 4096 at I = 4 is the whole slice; XREG with longer slices: 2,
//       two register sets of words and activation fragments)
// XTC  = XREG with the activations in A-fragment order (tg_w4_gemm.x_layout, the `f16TC` ops): the four dwords of a lane's piece come
//        from four places of the fragment tensor (tc_a_index) -- four 4-byte loads instead of one 16-byte load, the same registers after
template <typename DT, int I, bool QMX, int CPG, int TPW = 1, bool NORM = false, bool XREG = false, int CH = 4, bool XTC = false>
__global__ void __launch_bounds__(1024) w4_gemm_pair16_kernel(const Pair16Params p) {
  static_assert(!NORM || !QMX, "fused RMSNorm borrows the activation-sum area");
  static_assert(!XREG || (TPW == 1 && !NORM), "register-resident activations: one tile per workgroup, no fused norm");
  constexpr int WAVES = 16;
  constexpr int NT = WAVES * 64;
  constexpr int CPS = I / 2;  // 32-k chunks (= words per lane) of a super-tile
  constexpr bool STATIC_G = (CH * CPS) % CPG == 0;
  const uint32_t lds_x = (uint32_t)p.lds_x, lds_xs = (uint32_t)p.lds_xs;

  if constexpr (XREG) {
    // every kernel argument the prologue needs, touched here: the compiler otherwise fetches them where they are first used --
    // several dependent scalar-memory round trips in front of the first vector load (w4_gemv.cuh)
    asm volatile("" ::"s"(p.x), "s"(p.w), "s"(p.qinfo), "s"(p.lut), "s"(p.m), "s"(p.wrows), "s"(p.k), "s"(p.ksuper), "s"(p.spw),
                 "s"(p.stride_x), "s"(p.stride_w), "s"(p.stride_qinfo), "s"(p.stride_lut), "s"(p.gshift), "s"(p.ngroups), "s"(p.qtype));
  }
  const int tid = threadIdx.x;
#if GEMV_TRACE
  unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  P16_STAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int b = blockIdx.y;
  // (pairing the tiles 2 j, 2 j + 1 -- which share every line of scale | zero words -- on one XCD, as w4_gemm_stream.cuh / w4_gemv.cuh
  //  do, measured SLOWER here: m = 16 8.2 -> 8.6 us per graph node, m = 8 equal, profiles/r05_ab_xcd_ranges.txt)
  const int row0 = blockIdx.x * (16 * TPW);

  // this wave's slice of phase ph: super-tiles [ph ksuper_p + wave spw, + nl)
  const int nl = max(min(p.spw, p.ksuper_p - wave * p.spw), 0);
  int s_begin = wave * p.spw;

  // ---- requests, in the order they are consumed: LUT row of this thread's table column, activations, weights ----
  const int tcol = tid & 31;                     // table column (of tile t: + 32 t) = 16 copy + row
  uint32_t lp[TPW][8];
  if (p.qtype == TG_Q_ANY4_GLOBAL || p.qtype == TG_Q_ANY4_ROWWISE) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int trow = min(row0 + 16 * t + (tcol & 15), p.wrows - 1);
      const char* lsrc = p.lut + (int64_t)b * p.stride_lut + (p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)trow * 32 : 0);
      const u32x4 l0 = reinterpret_cast<const u32x4*>(lsrc)[0];
      const u32x4 l1 = reinterpret_cast<const u32x4*>(lsrc)[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { lp[t][j] = l0[j]; lp[t][4 + j] = l1[j]; }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      float v0, v1;
      if (p.qtype == TG_Q_INT4) {
        v0 = (float)(e - 8);
        v1 = (float)(e - 7);
      } else {  // fp4 e2m1 (mx4)
        const int e1 = e + 1;
        v0 = (e & 8 ? -1.f : 1.f) * ((e & 7) < 5 ? 0.5f * (e & 7) : ((e & 7) == 5 ? 3.f : (e & 7) == 6 ? 4.f : 6.f));
        v1 = (e1 & 8 ? -1.f : 1.f) * ((e1 & 7) < 5 ? 0.5f * (e1 & 7) : ((e1 & 7) == 5 ? 3.f : (e1 & 7) == 6 ? 4.f : 6.f));
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) lp[t][e >> 1] = DT::pack2(v0, v1);
    }
  }

  // activations: chunk (row a, 32 k) per thread, 64 bytes
  const int nch = (p.ksuper_p * 16 * I) >> 5;  // chunks per row and phase
  const int xtotal = p.m * nch;
  const char* xb = p.x + (int64_t)b * p.stride_x;
  uint32_t xd[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xd[j] = 0u;
  auto x_load = [&](int xi, int ph) {
    const int a = p.m == 1 ? 0 : xi / nch, ch = xi - a * nch + ph * nch;  // (m = 1: no division on the launch's critical path)
    if (p.x_tc) {
      tc_a_load_chunk(xb, a, ch, p.k >> 4, xd);
      return;
    }
    const u32x4* src = reinterpret_cast<const u32x4*>(xb + ((int64_t)a * p.k + ch * 32) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = src[j];
      xd[4 * j] = v[0]; xd[4 * j + 1] = v[1]; xd[4 * j + 2] = v[2]; xd[4 * j + 3] = v[3];
    }
  };
  if constexpr (!XREG)
    if (tid < xtotal) x_load(tid, 0);

  // weights of this lane: row n of each of the workgroup's TPW 16-row tiles, quad q
  int wrow[TPW];
  const char* wl[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    wrow[t] = min(row0 + 16 * t + n, p.wrows - 1);
    const int nt = min(wrow[t] >> 3, p.ntiles - 1);
    wl[t] = p.w + (int64_t)b * p.stride_w + ((int64_t)nt * p.ksuper * 32 + (4 * (wrow[t] & 7) + q)) * (2 * I);
  }
  const char* qb = p.qinfo + (int64_t)b * p.stride_qinfo;
  // two register sets where they fit without spills (one tile per workgroup, groups of >= 128): set B is only used when a
  // wave's slice is longer than one block of CH super-tiles (see the main loop)
  constexpr bool PIPE = TPW == 1 && (XREG ? CH < 4 : CPG >= 4);  // (XREG with CH = 4: the host only sends slices of one block)
  // XREG: this lane's activation fragments of a block, one per 32-k chunk (as loaded: the 16 bytes (lane >> 4) of row lane & 15's
  // chunk; after x_arrange: the A operand of the chunk)
  constexpr int NXF = XREG ? CH * CPS : 1;
  u32x4 xfA[NXF], xfB[PIPE ? NXF : 1];
  // address = (wave-uniform chunk base, SGPRs) + (per-lane 32-bit offset: the host checks m k 2 < 4 GiB): the saddr form, no 64-bit
  // vector address per load
  // XQ: the four lanes of a QUAD read one aligned 64-byte chunk of one row (lane 4 a + b: quarter b of row a) instead of lane (row, quarter)
  // = 16 q + i.  The vector-memory path coalesces per quad of adjacent lanes: with 16 different rows in adjacent lanes every lane's 16
  // bytes were a request of their own, and a workgroup's 128 KiB of activations took 3 us instead of 0.8 (tools/ubench/x_broadcast.hip:
  // 4.74 vs 2.47 us per node -- rows per wave-load, order within the quad, XCD-private copies, rotations: all irrelevant).  Two more bit
  // exchanges in x_arrange then put quarter b into register b of lane (b' = dword, row), and the A operand's row index becomes a ROTATED
  // row number: lane i holds activation row 4 (i & 3) + (i >> 2), so accumulator register r of lane (n, q) is row 4 r + q (not 4 q + r).
#ifndef P16_XQUAD
#define P16_XQUAD 1
#endif
  constexpr bool XQ = XREG && !XTC && P16_XQUAD;
  uint32_t xoff = XQ ? (uint32_t)((min(lane >> 2, p.m - 1) * p.k + 8 * (lane & 3)) * 2) : (uint32_t)((min(n, p.m - 1) * p.k + 8 * q) * 2);
  auto x_request = [&](u32x4 (&xf)[NXF], int l0) {
    if constexpr (XREG) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int s = l0 + j < nl ? s_begin + l0 + j : 0;  // (past the slice: a cached request, never consumed -- as w_request)
#pragma unroll
        for (int jc = 0; jc < CPS; ++jc) {
          if constexpr (XTC) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              xf[j * CPS + jc][e] = *reinterpret_cast<const uint32_t*>(xb + tc_a_index(min(n, p.m - 1), (s * CPS + jc) * 32 + 8 * q + 2 * e, p.k >> 4) * 2);
            continue;
          }
          asm volatile("" : "+v"(xoff));  // (opaque at its use: the zero-extension stays next to the load, w4_gemm_pair.cuh `pin`)
          xf[j * CPS + jc] = *reinterpret_cast<const u32x4*>(xb + (uint32_t)__builtin_amdgcn_readfirstlane((s * CPS + jc) * 64) + xoff);
        }
      }
    }
  };
  auto x_arrange = [&](u32x4 (&xf)[NXF]) {
    if constexpr (XREG) {
#pragma unroll
      for (int c = 0; c < NXF; ++c) {
        // dword j of quarter e -> dword e of quarter j (a 4 x 4 transpose over the lane bits 4, 5): lane (i, q) then holds
        // d[e] = (x[2 q + 8 e], x[2 q + 8 e + 1]) of the chunk
        const auto s02 = __builtin_amdgcn_permlane32_swap(xf[c][0], xf[c][2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(xf[c][1], xf[c][3], false, false);
        const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        uint32_t d0 = t01[0], d1 = t01[1], d2 = t23[0], d3 = t23[1];
        if constexpr (XQ) {
          // register bit 1 <-> lane bit 1, register bit 0 <-> lane bit 0 (within the quad: a DPP quad permute and a select each)
          auto qx = [&](uint32_t& x, uint32_t& y, auto CTRL, bool hi) {
            constexpr int ctrl = decltype(CTRL)::value;
            const uint32_t yx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)y, ctrl, 0xf, 0xf, false);
            const uint32_t xx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xf, 0xf, false);
            const uint32_t nx = hi ? yx : x, ny = hi ? y : xx;
            x = nx; y = ny;
          };
          const bool l1 = (lane & 2) != 0, l0 = (lane & 1) != 0;
          qx(d0, d2, std::integral_constant<int, 0x4E>{}, l1);  // quad_perm [2,3,0,1]
          qx(d1, d3, std::integral_constant<int, 0x4E>{}, l1);
          qx(d0, d1, std::integral_constant<int, 0xB1>{}, l0);  // quad_perm [1,0,3,2]
          qx(d2, d3, std::integral_constant<int, 0xB1>{}, l0);
        }
        xf[c] = u32x4{__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u),
                      __builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u)};
      }
    }
  };
  uint32_t wregA[TPW][CH][CPS], wregB[PIPE ? TPW : 1][PIPE ? CH : 1][PIPE ? CPS : 1];
  uint32_t qregA[TPW][CH][CPS], qregB[PIPE ? TPW : 1][PIPE ? CH : 1][PIPE ? CPS : 1];  // scale | zero word (or mx4 exponent byte) of the group of every chunk
  // Requests the block of CH super-tiles at slice position l0.  Positions past the slice are still requested (the number of loads
  // in flight stays the same on every path) but every lane reads the operand's first bytes: one cached request, never consumed.
  auto w_request = [&](uint32_t (&wreg)[TPW][CH][CPS], uint32_t (&qreg)[TPW][CH][CPS], int l0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int l = l0 + j;
      const bool valid = l < nl;  // wave-uniform
      const int s = valid ? s_begin + l : 0;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const char* src = valid ? wl[t] + (int64_t)s * (64 * I) : p.w;
        if constexpr (I == 2) {
          wreg[t][j][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src));
        } else if constexpr (I == 4) {
          const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src));
          wreg[t][j][0] = v[0]; wreg[t][j][1] = v[1];
        } else {
          const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
          wreg[t][j][0] = v[0]; wreg[t][j][1] = v[1]; wreg[t][j][2] = v[2]; wreg[t][j][3] = v[3];
        }
#pragma unroll
        for (int jc = 0; jc < CPS; ++jc) {
          // a block starts on a group boundary (slices are whole groups, CH CPS is a multiple of CPG): only the chunks that
          // start a group need their scale | zero word
          if (STATIC_G && (j * CPS + jc) % CPG != 0) continue;
          const int g = ((s * CPS + jc) * 32) >> p.gshift;
          if constexpr (QMX) qreg[t][j][jc] = *reinterpret_cast<const uint8_t*>(valid ? qb + (int64_t)wrow[t] * p.ngroups + g : qb);
          else qreg[t][j][jc] = *reinterpret_cast<const uint32_t*>(valid ? qb + ((int64_t)g * p.wrows + wrow[t]) * 4 : qb);
        }
      }
    }
  };
  // XREG: the activations FIRST.  Vector memory returns in request order per wave and a CU's vector-memory path moves 64 bytes per clock:
  // the 128 KiB of activations a workgroup needs at 16 rows (four times its 32 KiB of weights) are ~1 us of that path.  Requested behind
  // the weights they could only be delivered after the weights' HBM latency; in front of them they arrive while the weights are in flight:
  // 4096^2 per graph node at m = 16 / 12 / 9: 8.09 -> 7.71 / 7.36 -> 6.98 / 6.79 -> 6.49 us (profiles/r05_ab_p16_prologue.txt).
#ifndef P16_XFIRST
#define P16_XFIRST 1
#endif
  if (P16_ABL != 1) {
    if constexpr (XREG && P16_XFIRST) { x_request(xfA, 0); w_request(wregA, qregA, 0); }
    else { w_request(wregA, qregA, 0); x_request(xfA, 0); }
  }
  else {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int jc = 0; jc < CPS; ++jc) { wregA[t][j][jc] = (uint32_t)(tid * 977 + j * 13 + jc + t); qregA[t][j][jc] = 0x3c003c00u; }
  }

  // ---- stage the activations (byte order) and their group sums; build the table ----
  auto x_store = [&](int xi, bool on) {
    const int a = on && p.m > 1 ? xi / nch : 0, ch = on ? xi - a * nch : 0;
    if (on) {
      const uint32_t dst = lds_x + (uint32_t)(a * p.x_pitch + ch * 64);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        u32x4 o;
        o[0] = __builtin_amdgcn_perm(xd[qq + 4], xd[qq], 0x05040100u);       // x[2q]     x[2q+8]
        o[1] = __builtin_amdgcn_perm(xd[qq + 12], xd[qq + 8], 0x05040100u);  // x[2q+16]  x[2q+24]
        o[2] = __builtin_amdgcn_perm(xd[qq + 4], xd[qq], 0x07060302u);       // x[2q+1]   x[2q+9]
        o[3] = __builtin_amdgcn_perm(xd[qq + 12], xd[qq + 8], 0x07060302u);  // x[2q+17]  x[2q+25]
        *(lds_u32x4ptr)(dst + (uint32_t)(qq * 16)) = o;
      }
    }
    if constexpr (!QMX) {
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) sum = dot2_ones<DT>(xd[j], sum);
      for (int o = 1; o <= p.gch_mask; o <<= 1) sum += __shfl_xor(sum, o);
      if (on && (ch & p.gch_mask) == 0) *(lds_fptr)(lds_xs + (uint32_t)(((ch >> (p.gshift - 5)) * 16 + a) * 4)) = sum;
    }
  };
  const int gpp = p.ngroups / p.phases;  // groups per phase
  auto x_stage = [&](int ph, bool pre) {  // pre: the first batch of chunks is already in xd
    if constexpr (NORM) {
      // LlamaRMSNorm of the rows on the way into LDS (tg_w4_gemm.norm_weight; the host only asks for it with one phase and at
      // most NT chunks, one per thread): sum of squares per chunk, over the 64 chunks of a wave (one row: k % 2048 == 0), one
      // partial per wave in the (not yet written) activation-sum area; per row they are added in wave order (deterministic)
      const bool on = tid < xtotal;
      if (!pre) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xd[j] = 0u;
        if (on) x_load(tid, ph);
      }
      // (the chunk's norm weights are requested before the reduction: one L2 round trip less on the launch's critical path)
      const int a = (on && p.m > 1) ? tid / nch : 0;
      u32x4 gw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gw[j] = reinterpret_cast<const u32x4*>(p.norm_w + (on ? tid - a * nch : 0) * 64)[j];
      float ss = chunk_sumsq<DT>(xd);
      ss = tgl::wave_sum(ss);
      if (lane == 0) *(lds_fptr)(lds_xs + (uint32_t)(wave * 4)) = ss;
      __syncthreads();
      ss = 0.f;
      for (int w0 = a * (nch >> 6); w0 < (a + 1) * (nch >> 6); ++w0) ss += *(lds_fptr)(lds_xs + (uint32_t)(w0 * 4));
      __syncthreads();  // the partial sums are read before x_store's group sums land in the same area
      if (on) chunk_rmsnorm<DT>(xd, rsqrtf(ss * (1.0f / (float)p.k) + p.norm_eps), gw);
      x_store(tid, on);
    } else
    for (int it0 = 0; it0 < (P16_ABL == 3 ? 0 : xtotal); it0 += NT) {
      const int xi = it0 + tid;
      const bool on = xi < xtotal;
      if (it0 > 0 || !pre) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xd[j] = 0u;
        if (on) x_load(xi, ph);
      }
      x_store(xi, on);
    }
    if constexpr (!QMX)
      for (int idx = tid; idx < gpp * 16; idx += NT)
        if ((idx & 15) >= p.m) *(lds_fptr)(lds_xs + (uint32_t)(idx * 4)) = 0.f;
    if (tid == 0) *(lds_u32x4ptr)(lds_x + (uint32_t)(p.m * p.x_pitch)) = u32x4{0, 0, 0, 0};  // zero piece for padding rows
  };
  P16_STAMP(1);
  if constexpr (!XREG) x_stage(0, true);
  // (mx4: no table -- the weights are converted in registers by v_cvt_scalef32_pk_bf16_fp4, w4_gemm_pair.cuh: mx4_cvt_word)
#pragma unroll
  for (int t = 0; t < (QMX ? 0 : TPW); ++t) {
    // thread = (column tcol, high nibble (tid >> 5) & 15, half tid >> 9 of the low nibbles): entries (lut[lo], lut[hi])
    const int hi = (tid >> 5) & 15, half = tid >> 9;
    uint32_t hw = lp[t][0];
#pragma unroll
    for (int j = 1; j < 8; ++j) hw = ((hi >> 1) == j) ? lp[t][j] : hw;
    const uint32_t hsel = (hi & 1) ? 0x07060000u : 0x05040000u;  // the high half of the entry: value `hi` of the pair hw
    const uint32_t base = (uint32_t)((hi * 16 + half * 8) * 256 + (32 * t + tcol) * 4);
    // (masks, not `half ? lp[4 + j] : lp[j]`: a select between two array elements becomes a dynamically indexed private array,
    //  which the compiler then moves to static LDS -- and this kernel's table must start at LDS address 0)
    const uint32_t hm = half ? 0xffffffffu : 0u;
    uint32_t lq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lq[j] = (lp[t][4 + j] & hm) | (lp[t][j] & ~hm);
#pragma unroll
    for (int a = 0; a < (P16_ABL == 4 ? 1 : 8); ++a) {
      const uint32_t e = __builtin_amdgcn_perm(hw, lq[a >> 1], hsel | ((a & 1) ? 0x0302u : 0x0100u));
      ((lds_u32ptr)base)[a * 64] = e;  // (one LDS pointer + constant offsets: immediate offset fields, no address arithmetic per store)
    }
  }
#ifndef P16_ARRANGE_EARLY
#define P16_ARRANGE_EARLY 0
#endif
  // (developer A/B) XREG, the whole slice in one block: the fragment transposes in front of the barrier instead of behind it -- no
  // difference (7.71 vs 7.67 us): the stamps say the LAST wave's activations land ~3.8 us after the requests whatever the order. Every
  // CU of an XCD pulls the same 128 KiB of x out of that XCD's L2 -- 4 MiB per XCD and launch at ~2 TB/s: the launch is bound by that
  // broadcast, not by the weights (profiles/r05_p16_trace_xfirst.txt)
  constexpr bool ARRANGE_EARLY = XREG && CH == 4 && P16_XFIRST && P16_ARRANGE_EARLY && P16_ABL != 6;
  if constexpr (ARRANGE_EARLY) x_arrange(xfA);
  P16_STAMP(2);
#ifndef P16_ASM_BARRIER
#define P16_ASM_BARRIER 0
#endif
  if constexpr (XREG && P16_ASM_BARRIER) {
    // (developer A/B: `__syncthreads()` waits vmcnt(0) -- for every weight and activation request of the wave -- where only the table's LDS
    //  stores have to be done; spelled out without that wait the launch measured 0.35 us SLOWER in either request order: not used)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  } else {
    __syncthreads();
  }
  P16_STAMP(3);

  // ---- main loop ----
  const bool a_on = n < p.m;  // (as the A operand's row index: lane (i = n, q))
  const uint32_t xzero = lds_x + (uint32_t)(p.m * p.x_pitch);
  const uint32_t xrow = a_on ? lds_x + (uint32_t)(n * p.x_pitch + q * 16) : xzero;
  const uint32_t xmask = a_on ? 0xffffffffu : 0u;
  uint32_t colreg[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) colreg[t] = (uint32_t)((32 * t + 16 * (q & 1) + n) * 4);
  // this lane's accumulator rows are activation rows 4 q + r
  const uint32_t xs_lane = lds_xs + (uint32_t)(4 * q * 4);

  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t acc[TPW];
  float yacc[TPW][4];
  float gs[TPW], gz[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = zero4;
    gs[t] = gz[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) yacc[t][r] = 0.f;
  }
  f32x4 xsv = {0.f, 0.f, 0.f, 0.f};
  f32x4_t xsacc = zero4;  // XREG: the group's activation sums, rows 4 q + r (an MFMA against ones)
  const uint32_t one2 = DT::pack2(1.f, 1.f);
  const u32x4 ones = {one2, one2, one2, one2};

  if (P16_ABL == 2) {  // loads consumed, nothing computed
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int jc = 0; jc < CPS; ++jc) yacc[t][0] += u2f(wregA[t][j][jc] ^ qregA[t][j][jc]);
  }
  int chunk_ph = 0;  // first chunk of the current phase: LDS holds chunks / groups relative to it
  // one 32-k chunk of every tile: 4 lookups + one MFMA per tile, the X fragment read once
  auto step = [&](const uint32_t (&wreg)[TPW][CH][CPS], const uint32_t (&qreg)[TPW][CH][CPS], const u32x4 (&xfr)[NXF], int j, int jc, int chunk, bool gfirst, bool glast) {
    u32x4 xf;
    if constexpr (XREG) xf = xfr[j * CPS + jc];
    else xf = *(lds_cu32x4ptr)(xrow + ((uint32_t)((chunk - chunk_ph) * 64) & xmask));
    u32x4 bf[TPW];
    if constexpr (QMX) {  // the group's scale inside the conversion: accumulators run through the whole slice, nothing per group
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        if (gfirst) {
          const uint32_t qv = qreg[t][j][jc];
          const float sc = u2f(qv == 0u ? 0x00400000u : (qv << 23));  // Dequantization.cuh:331-339; e = 255: inf ...
          gs[t] = __builtin_fmaf(sc, 0.f, sc);                         // ... -> NaN
        }
        bf[t] = mx4_cvt_word(wreg[t][j][jc], gs[t]);
        acc[t] = mfma16<DT>(xf, bf[t], acc[t]);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const uint32_t w = wreg[t][j][jc];
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[t][e] = *(lds_cu32ptr)(__builtin_amdgcn_perm(w, colreg[t], 0x0c0c0400u + ((uint32_t)e << 8)));
    }
    if (gfirst) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const uint32_t qv = qreg[t][j][jc];
        if constexpr (QMX) {
          gs[t] = u2f(qv == 255u ? 0x7fc00000u : (qv == 0u ? 0x00400000u : (qv << 23)));  // Dequantization.cuh:331-339
        } else {
          gs[t] = DT::lo_f32(qv);
          gz[t] = DT::hi_f32(qv);
        }
      }
      if constexpr (!QMX && !XREG) xsv = *(lds_cf32x4ptr)(xs_lane + (uint32_t)(((((chunk - chunk_ph) * 32) >> p.gshift) * 16) * 4));
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = mfma16<DT>(xf, bf[t], gfirst ? zero4 : acc[t]);
    if constexpr (XREG && !QMX && P16_ABL != 5) {  // (ablation 5: no activation sums)
      xsacc = mfma16<DT>(xf, ones, gfirst ? zero4 : xsacc);
      if (glast) xsv = f32x4{xsacc[0], xsacc[1], xsacc[2], xsacc[3]};
    }
    if (glast) {
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          yacc[t][r] = __builtin_fmaf(gs[t], acc[t][r], yacc[t][r]);
          if constexpr (!QMX) yacc[t][r] = __builtin_fmaf(gz[t], xsv[r], yacc[t][r]);
        }
    }
  };
  // the block of CH super-tiles at slice position l0 (its words are in wreg / qreg)
  auto consume_block = [&](const uint32_t (&wreg)[TPW][CH][CPS], const uint32_t (&qreg)[TPW][CH][CPS], const u32x4 (&xfr)[NXF], int l0) {
    if (STATIC_G && l0 + CH <= nl) {
      // a whole block: the slice starts on a group boundary and CH CPS is a multiple of CPG, so step u starts a group iff
      // u % CPG == 0 -- straight-line code
      const int chunk0 = (s_begin + l0) * CPS;
#pragma unroll
      for (int u = 0; u < CH * CPS; ++u) {
        step(wreg, qreg, xfr, u / CPS, u % CPS, chunk0 + u, u % CPG == 0, u % CPG == CPG - 1);
        // XREG: the fragments occupy 4 registers per chunk -- keep the scheduler from hoisting every chunk's lookups to the top of the
        // block (4 more each); two chunks' lookups in flight per wave, four waves per SIMD hide the rest
        if constexpr (XREG && P16_XSCHED > 0)
          if (u % P16_XSCHED == P16_XSCHED - 1) __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (l0 + j < nl) {
        const int s = s_begin + l0 + j;
#pragma unroll
        for (int jc = 0; jc < CPS; ++jc) {
          const int chunk = s * CPS + jc;
          step(wreg, qreg, xfr, j, jc, chunk, (chunk & p.gch_mask) == 0, (chunk & p.gch_mask) == p.gch_mask);
        }
      }
    }
  };
  for (int ph = 0; ph < (XREG ? 1 : p.phases); ++ph) {
    if (!XREG && ph > 0) {
      // the next part of k: its first weights are requested before the activations are re-staged (two barriers: every wave is
      // done with the previous part's activations / the new ones are visible)
      s_begin = ph * p.ksuper_p + wave * p.spw;
      w_request(wregA, qregA, 0);
      __syncthreads();
      x_stage(ph, false);
      __syncthreads();
    }
    chunk_ph = ph * p.ksuper_p * CPS;
    if (P16_ABL == 2) continue;
    if (nl <= CH || (XREG && CH == 4)) {  // (wave-uniform) the whole slice was requested up front (XREG, CH = 4: always -- the host's choice)
      if (P16_ABL != 6 && !ARRANGE_EARLY) x_arrange(xfA);  // (ablation 6: fragments used as loaded)
#if GEMV_TRACE
      asm volatile("" ::"v"(xfA[0]), "v"(xfA[NXF - 1]));
#endif
      P16_STAMP(4);
      if (nl > 0) consume_block(wregA, qregA, xfA, 0);
      continue;
    }
    if constexpr (PIPE) {
      // long slices: two blocks per turn, the next block always requested before the current one is consumed
      for (int l0 = 0; l0 < nl; l0 += 2 * CH) {
#ifndef P16_XFIRST_PIPE
#define P16_XFIRST_PIPE 0
#endif
        if constexpr (XREG && P16_XFIRST_PIPE) { x_request(xfB, l0 + CH); w_request(wregB, qregB, l0 + CH); }
        else { w_request(wregB, qregB, l0 + CH); x_request(xfB, l0 + CH); }
        x_arrange(xfA);
        consume_block(wregA, qregA, xfA, l0);
        if constexpr (XREG && P16_XFIRST_PIPE) { x_request(xfA, l0 + 2 * CH); w_request(wregA, qregA, l0 + 2 * CH); }
        else { w_request(wregA, qregA, l0 + 2 * CH); x_request(xfA, l0 + 2 * CH); }
        x_arrange(xfB);
        if (l0 + CH < nl) consume_block(wregB, qregB, xfB, l0 + CH);
      }
    } else {
      for (int l0 = 0; l0 < nl; l0 += CH) {
        if (l0 > 0) { w_request(wregA, qregA, l0); x_request(xfA, l0); }
        x_arrange(xfA);
        consume_block(wregA, qregA, xfA, l0);
      }
    }
  }  // phases

  if constexpr (QMX) {  // the slice's sums as they are
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) yacc[t][r] = acc[t][r];
  }
  // ---- split-K tail: partial sums of the 16 waves meet in the (now unused) table's LDS, added in wave order ----
#if GEMV_TRACE
  asm volatile("" ::"v"(yacc[0][0]), "v"(yacc[0][3]));
#endif
  P16_STAMP(5);
  __syncthreads();
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(lds_fptr)((uint32_t)((((t * WAVES + wave) * 4 + r) * 64 + lane) * 4)) = yacc[t][r];
  __syncthreads();
  P16_STAMP(6);
#if GEMV_TRACE
  if (p.trace && tid == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) p.trace[(size_t)blockIdx.x * 8 + i] = tr[i];
  }
#endif
  if (tid < 256 * TPW) {
    const int t = tid >> 8, r = (tid >> 6) & 3, l = tid & 63;
    const int a = XQ ? 4 * r + (l >> 4) : 4 * (l >> 4) + r, row = row0 + 16 * t + (l & 15);  // (XQ: the A operand's rows are rotated)
    if (a < p.m && row < p.wrows) {
      float sum = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < WAVES; ++w8) sum += *(lds_fptr)((uint32_t)((((t * WAVES + w8) * 4 + r) * 64 + l) * 4));
      if (p.epilogue == TG_EPI_SWIGLU) {
        // rows come in blocks of 8 gate + 8 up (this workgroup's 16 rows are one block): lane l + 8 holds the up row of gate row l
        if ((l & 15) < 8) {
          float up = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < WAVES; ++w8) up += *(lds_fptr)((uint32_t)((((t * WAVES + w8) * 4 + r) * 64 + l + 8) * 4));
          *reinterpret_cast<uint16_t*>(p.y + (int64_t)b * p.stride_y + ((int64_t)a * (p.wrows >> 1) + ((row >> 4) << 3) + (row & 7)) * 2) = swiglu16<DT>(sum, up);
        }
        return;
      }
      uint16_t o16 = DT::from_f32(sum);
      if (p.bias)  // rounded sum + bias, rounded again: the reference module's separate `y + bias` (modules.py:221-222)
        o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + (int64_t)b * p.stride_bias + ((int64_t)a * p.bias_row_stride + row) * 2)));
      *reinterpret_cast<uint16_t*>(p.y + (int64_t)b * p.stride_y + (p.y_tc ? tc_a_index(a, row, p.y_tiles) : (int64_t)a * p.wrows + row) * 2) = o16;
    }
  }
}
