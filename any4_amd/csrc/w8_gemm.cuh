// w8_gemm.cuh -- W8A16 small-batch GEMM for gfx950 (included by tinygemm_hip.hip).
//
// Replaces {A,B}Layout_TC_int8 (reference MatrixLayoutA.cuh int8 layout, MatrixLayoutB.cuh:1104-1327) and
// convert_i8x4_to_f16x2x2 (Dequantization.cuh:262-330) under tinygemm_m16n8k16_chunk_kernel: the stored byte b
// means b - 128, w = RNE16(fma(b - 128, scale, zero)), contraction in fp32 on the MFMA.
//
// Packed words are consumed as stored (bit-identical to the reference's layouts):
//   Bint8 [nT][kS][32][I]  : lane t, k-tile kt -> ONE word, bytes = k 2q, 2q+8, 2q+1, 2q+9 (q = t % 4) of row t / 4
//   Aint8 [mT][kO][32][2I] : lane t, k-tile kt -> TWO words, bytes = (m0,k0)(m1,k0)(m0,k0+1)(m1,k0+1) and the same at
//                            k0 + 8 (m0 = t / 4, m1 = m0 + 8, k0 = 2q)
// Mapping: MFMA 16x16x32, W = A operand.  Lane (i = lane & 15, Q = lane >> 4) owns row i of the 16-row tile and
// the q = Q words; one K-slot = two k-tiles = the 8 k values {2Q, 2Q+1, 2Q+8, 2Q+9} + {0, 16}, so the matching X
// fragment of lane (c, Q) is four dwords of activation row c at byte offsets 4Q + {0, 16, 32, 48} of the slot.
// A step = 4 k-tiles (64 k, two K-slots): 16 B (B side, I = 4) or 2 x 16 B (A side, I = 2) of packed bytes per lane.
// Same wave-tile / split-K / no-predication structure as w4_gemm.cuh (clamped addresses; out-of-range lanes get
// scale = zero = 0 and contribute exact zeros).  A correct, HBM-streaming kernel; not tuned like the 4-bit path.
#pragma once

#ifndef W8_RING
#define W8_RING 2   // steps in flight per wave (1 / 2 / 4 / 8 measured: 7.7 / 7.4 / 7.4 / 7.7 us at one row: VALU-bound at ~ 190 instructions per step, not latency-bound)
#endif
template <typename DT, bool LAYOUT_A, int I, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES <= 8 ? 2 : 1) w8_gemm_kernel(const GemmParams p) {
  __shared__ f32x4 s_red[WAVES * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, Q = lane >> 4;

  const int slice = wave & (p.splitk - 1);
  const int rt = blockIdx.x * (WAVES >> p.sk_shift) + (wave >> p.sk_shift);
  const bool rt_ok = rt < p.rowtiles;
  const int ct = blockIdx.y;
  const int64_t b = blockIdx.z;
  const char* xb = p.x + b * p.stride_x;
  const uint32_t* wb = reinterpret_cast<const uint32_t*>(p.w + b * p.stride_w);
  const uint32_t* qb = reinterpret_cast<const uint32_t*>(p.qinfo + b * p.stride_qinfo);
  char* yb = p.y + b * p.stride_y;

  const int row0 = rt * 16, row = row0 + i;
  const int row_c = min(row, p.wrows - 1);
  // packed tile and lane-in-tile of this row
  const int tile = LAYOUT_A ? rt : 2 * rt + (i >> 3);
  const int tile_c = min(tile, p.ntiles - 1);
  const int t = 4 * (i & 7) + Q;
  const bool row_ok = rt_ok && row < p.wrows && tile < p.ntiles;
  constexpr int WPT = LAYOUT_A ? 2 : 1;  // words per k-tile and lane
  const int ktiles = p.k >> 4;           // k % 32 == 0 is checked by the host
  const uint32_t lane_words = (uint32_t)((tile_c * p.ksuper * 32 + t) * (I * WPT));  // word index of (tile, super 0, t)

  const int xrow = min(ct * 16 + i, p.m - 1);
  const bool xcol = ct * 16 + i < p.m;
  const char* xlane = xb + ((int64_t)xrow * p.k) * 2 + 16 * Q;   // (dwords 4Q ... 4Q + 3 of a K-slot's 64 bytes: transposed across the Q rows at the consumer)

  const int nsteps_total = (ktiles + 3) >> 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // everything one step needs, so that the loads of step s + splitk are in flight while step s is computed
  struct Step {
    uint32_t w0[4], w1[4];  // packed words of the 4 k-tiles (1 per k-tile on the B side, 2 on the A side)
    uint32_t q[2];          // scale|zero of the two K-slots
    u32x4 x[2];             // X fragments of the two K-slots
  };
  auto load_step = [&](int s, Step& st) {
    // one vector load per k super-tile (I k-tiles, I * WPT consecutive words of this lane); a clamped (repeated) super-tile
    // only feeds K-slots whose scale is forced to zero below
    constexpr int WPS = I * WPT;   // words per super-tile and lane: 1, 2 or 4
    constexpr int SPS = 4 / I;     // super-tiles per step (I = 4: 1, I = 2: 2, I = 1: 4)
    uint32_t w[4 * WPT];
#pragma unroll
    for (int u = 0; u < SPS; ++u) {
      const int sup = min(s * SPS + u, p.ksuper - 1);
      const uint32_t* src = wb + lane_words + (uint32_t)(sup * 32 * WPS);
      if constexpr (WPS == 4) {
        const u32x4 v = LAYOUT_A ? *reinterpret_cast<const u32x4*>(src) : __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
        w[u * 4 + 0] = v[0]; w[u * 4 + 1] = v[1]; w[u * 4 + 2] = v[2]; w[u * 4 + 3] = v[3];
      } else if constexpr (WPS == 2) {
        const u32x2 v = LAYOUT_A ? *reinterpret_cast<const u32x2*>(src) : __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src));
        w[u * 2 + 0] = v[0]; w[u * 2 + 1] = v[1];
      } else {
        w[u] = __builtin_nontemporal_load(src);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // k-tile j of the step: word j (B side) or words 2j, 2j + 1 (A side)
      st.w0[j] = w[j * WPT];
      if constexpr (LAYOUT_A) st.w1[j] = w[j * WPT + 1];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kt0 = 4 * s + 2 * h;
      const int kt0_c = min(kt0, ktiles - 2);
      // (every load of a step is unconditional per lane, with clamped addresses, and nothing of it is touched before compute_step: a load
      //  under a branch or a select right behind it made hipcc wait vmcnt(0) inside load_step -- every step then exposed the whole memory
      //  latency whatever the ring depth: 0.86 us per step and wave, a one-row 4096^2 layer at the 16-bit layer's time)
      st.q[h] = qb[(uint32_t)(((kt0_c << 4) >> p.gshift) * p.wrows + row_c)];
      st.x[h] = *reinterpret_cast<const u32x4*>(xlane + (int64_t)kt0_c * 32);   // (16-byte aligned: the host checks x)
    }
  };
  auto compute_step = [&](int s, const Step& st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // two K-slots of two k-tiles each
      // out-of-range rows / k-tiles: scale = zero = 0, the lane contributes exact zeros; activation rows beyond m: zeros
      const uint32_t qh = (row_ok && 4 * s + 2 * h < ktiles) ? st.q[h] : 0u;
      const float sc = DT::lo_f32(qh), zp = DT::hi_f32(qh);
      u32x4 a;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // the four bytes of this lane's row in k-tile 2h + j of the step, in k order 2Q, 2Q+1, 2Q+8, 2Q+9
        uint32_t b0, b1, b2, b3;
        if constexpr (LAYOUT_A) {
          const uint32_t u0 = st.w0[2 * h + j] >> ((i >> 3) * 8), u1 = st.w1[2 * h + j] >> ((i >> 3) * 8);
          b0 = u0 & 0xffu; b1 = (u0 >> 16) & 0xffu; b2 = u1 & 0xffu; b3 = (u1 >> 16) & 0xffu;
        } else {
          const uint32_t u = st.w0[2 * h + j];
          b0 = u & 0xffu; b1 = (u >> 16) & 0xffu; b2 = (u >> 8) & 0xffu; b3 = u >> 24;
        }
        const float f0 = __builtin_fmaf((float)b0 - 128.f, sc, zp), f1 = __builtin_fmaf((float)b1 - 128.f, sc, zp);
        const float f2 = __builtin_fmaf((float)b2 - 128.f, sc, zp), f3 = __builtin_fmaf((float)b3 - 128.f, sc, zp);
        a[2 * j] = DT::pack2(f0, f1);
        a[2 * j + 1] = DT::pack2(f2, f3);
      }
      const u32x4 xt = transpose_rows4(st.x[h]);
      const u32x4 xv = {xcol ? xt[0] : 0u, xcol ? xt[1] : 0u, xcol ? xt[2] : 0u, xcol ? xt[3] : 0u};
      acc = DT::mfma(a, xv, acc);
    }
  };
  if (rt_ok && slice < nsteps_total) {
    // a ring of W8_RING steps in flight per wave (round 6)
    Step ring[W8_RING];
    // this wave's steps: slice, slice + splitk, ...: nw of them.  Rounds of W8_RING steps whose refills are all in range run without a branch
    // around a load (exact vmcnt); the last round(s) only consume -- a refill past the end is real work for the CU's vector-memory path, which
    // this kernel's 4-byte activation loads load 4 x as much as the weights do (unconditional clamped refills throughout: 8 rows 14.9 -> 18.9 us)
    const int nw = (nsteps_total - slice + p.splitk - 1) >> p.sk_shift;
#pragma unroll
    for (int j = 0; j < W8_RING; ++j) load_step(slice + j * p.splitk, ring[j]);
    int base = 0;
    for (; base + 2 * W8_RING <= nw; base += W8_RING) {
#pragma unroll
      for (int j = 0; j < W8_RING; ++j) {
        compute_step(slice + (base + j) * p.splitk, ring[j]);
        load_step(slice + (base + W8_RING + j) * p.splitk, ring[j]);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < W8_RING; ++j) {
        const int jj = base + r * W8_RING + j;
        if (jj < nw) {
          compute_step(slice + jj * p.splitk, ring[j]);
          if (jj + W8_RING < nw) load_step(slice + (jj + W8_RING) * p.splitk, ring[j]);
        }
      }
    }
  }

  // ---- split-K tail (as in w4_gemm.cuh): lane (c = lane & 15, Q) holds y[act row c][rows row0 + 4Q .. +3] ----
  if (p.splitk > 1) {
    s_red[wave * 64 + lane] = acc;
    __syncthreads();
    if (slice != 0) return;
    for (int o = 1; o < p.splitk; ++o) acc += s_red[(wave + o) * 64 + lane];
  }
  const int col = ct * 16 + i;
  const int rowg = row0 + 4 * Q;
  if (rt_ok && col < p.m && rowg < p.wrows) {
    store_rows4<DT>(yb, p.bias ? p.bias + b * p.stride_bias + (int64_t)col * p.bias_row_stride * 2 : nullptr, (int64_t)col * p.wrows + rowg, rowg, acc);
  }
}
