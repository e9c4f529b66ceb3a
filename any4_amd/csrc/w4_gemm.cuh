// w4_gemm.cuh -- the W4A16 small-batch GEMM kernel for gfx950 (included by tinygemm_hip.hip).
//
// Replaces tinygemm_m16n8k16_chunk_kernel (reference TinyGemmImpl.cuh:23-345) together with
// {A,B}Layout_TC_int4 (MatrixLayoutA.cuh:375-816, MatrixLayoutB.cuh:686-1101), {A,B}Layout_RM and
// the converters of Dequantization.cuh.
//
// Work decomposition ("wave-tile"):
//   * a 16-row weight tile x the whole k is the unit of output; `splitk` waves share one tile
//     (wave `slice` takes steps slice, slice + splitk, ...), a workgroup of WAVES waves therefore
//     holds WAVES / splitk tiles.  splitk = 1 (many tiles: streaming) needs no barrier and no LDS
//     reduction at all; splitk = WAVES (one matrix of 256 tiles on 256 CUs: latency) is the classic
//     split-K workgroup.
//   * a step = one 16-byte non-temporal load of packed weights per lane (1 KiB per wave) plus the
//     matching scale|zero word and X fragments; DEPTH steps are kept in flight per wave.
//   * every wave owns a private f32 LUT image in LDS, [16 entries][64 lanes]: lane l only ever
//     touches bank l % 32, so the data-dependent lookups are conflict-free and need no barrier
//     (a lane reads only what it wrote itself).  The lookup address (entry << 8 | lane << 2 | base)
//     is produced by ONE v_perm_b32 per nibble.
//   * no per-lane predication in the loop: out-of-range lanes load from clamped (valid) addresses
//     and get scale = zero = 0, so they contribute exact zeros.
#pragma once

// (GemmParams and the CANON_* values live in tg_common.cuh: the launch paths of the kernel families are separate translation units)


// Word transpose between the four 16-lane rows of a wave so that every lane ends up with the
// four words (q = 0..3) of ONE k-chunk.  See DESIGN.md "canonical chunk".
template <int CANON>
__device__ __forceinline__ void canonicalize(u32x4& w) {
  if constexpr (CANON == CANON_PAIR) {
    // lane holds (q=2p, j0) (q=2p, j1) (q=2p+1, j0) (q=2p+1, j1); the partner row holds the other p.
    auto r0 = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
    w = u32x4{r0[0], r1[0], r0[1], r1[1]};  // (q0, q1, q2, q3) of one chunk
  } else if constexpr (CANON == CANON_QUAD) {
    // lane row Q holds words j = 0..3 of q = Q: 4x4 transpose across the rows
    auto r0 = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
    auto s0 = __builtin_amdgcn_permlane32_swap(r0[0], r1[0], false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(r0[1], r1[1], false, false);
    w = u32x4{s0[0], s1[0], s0[1], s1[1]};
  }
}

// One prefetch slot = everything a lane needs for one step.
template <int NMMA>
struct Slot {
  u32x4 w;        // 4 packed words
  uint32_t q;     // scale|zero pair (or mx4 exponent byte)
  u32x4 x[NMMA];  // X fragments
};

template <typename DT, bool LAYOUT_A, int CANON, bool QMX, int WAVES, int DEPTH, int MINW, int ABL = 0>
__global__ void __launch_bounds__(WAVES * 64, MINW) w4_gemm_kernel(const GemmParams p) {
  constexpr int CHUNK = LAYOUT_A ? 16 : 32;  // k covered by one lane per step
  constexpr int KSTEP = 4 * CHUNK;           // k covered by one wave per step
  constexpr int NMMA = LAYOUT_A ? 2 : 4;     // MFMAs per step
  constexpr int WPL = (CANON == CANON_NONE) ? 1 : (CANON == CANON_PAIR ? 2 : 4);  // words per lane-row in the packed layout

  // per wave: [entry][lane] f32 LUT, one bank column per lane; 4 KiB-aligned so that the entry
  // index occupies address bits 8..11 exactly
  __shared__ __attribute__((aligned(4096))) float s_tab[WAVES * 16 * 64];
  __shared__ f32x4 s_red[WAVES * 64];  // split-K partial tiles (splitk > 1 only)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15;  // weight row within the tile (A operand) / activation row (B operand)
  const int Q = lane >> 4;  // k-chunk selector
  const int r = i & 7;

  const int slice = wave & (p.splitk - 1);
  const int rt = blockIdx.x * (WAVES >> p.sk_shift) + (wave >> p.sk_shift);
  const bool rt_ok = rt < p.rowtiles;  // wave-uniform
  const int ct = blockIdx.y;
  const int64_t b = blockIdx.z;
  const char* xb = p.x + b * p.stride_x;
  const char* wb = p.w + b * p.stride_w;
  const char* qb = p.qinfo + b * p.stride_qinfo;
  const char* lb = p.lut ? p.lut + b * p.stride_lut : nullptr;
  char* yb = p.y + b * p.stride_y;

  const int row0 = rt * 16;
  const int row = row0 + i;
  const bool row_ok = rt_ok && row < p.wrows;
  const int row_c = min(row, p.wrows - 1);

  // ---- per-lane addressing (32-bit byte offsets from wave-uniform bases; invalid lanes clamp) ----
  const int tile = LAYOUT_A ? rt : 2 * rt + (i >> 3);
  const bool tile_ok = rt_ok && tile < p.ntiles;
  const int tile_c = min(tile, p.ntiles - 1);
  int toff;
  if constexpr (CANON == CANON_NONE) toff = 4 * r;
  else if constexpr (CANON == CANON_PAIR) toff = 4 * r + 2 * (Q & 1);
  else toff = 4 * r + Q;
  const uint32_t wlane = (uint32_t)((tile_c * p.ksuper * 32 + toff) * WPL * 4);  // bytes; matrix < 4 GiB
  const uint32_t wstep = 32u * WPL * 4u;                                         // bytes per k super-tile

  const int xrow = min(ct * 16 + i, p.m - 1);
  const bool xcol = ct * 16 + i < p.m;
  const uint32_t xlane = (uint32_t)((xrow * p.k + Q * CHUNK) * 2);

  const int nsteps_total = (p.k + KSTEP - 1) / KSTEP;
  // this wave owns steps slice, slice + splitk, ...
  const int nsteps = rt_ok ? ((nsteps_total - slice + p.splitk - 1) >> p.sk_shift) : 0;

  auto load_slot = [&](int j, Slot<NMMA>& sl) {
    const int s = slice + (j << p.sk_shift);
    int ks;
    if constexpr (CANON == CANON_NONE) ks = 4 * s + Q;
    else if constexpr (CANON == CANON_PAIR) ks = 2 * s + (Q >> 1);
    else ks = s;
    ks = min(ks, p.ksuper - 1);
    if constexpr (ABL == 3 || ABL == 9) sl.w = u32x4{(uint32_t)ks, 1u, 2u, 3u};  // ablation: no weight stream
    // (Aint4: lanes i and i + 8 read the same words -- keep them cacheable, see w4_gemm_stream.cuh)
    else if constexpr (LAYOUT_A) sl.w = *reinterpret_cast<const u32x4*>(wb + (wlane + (uint32_t)ks * wstep));
    else sl.w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb + (wlane + (uint32_t)ks * wstep)));
    const int kk = s * KSTEP + Q * CHUNK;  // first k of this lane's canonical chunk
    const bool ok = tile_ok && row_ok && kk < p.k;
    const int kk_c = min(kk, p.k - CHUNK);
    const int g = kk_c >> p.gshift;
    uint32_t q;
    if constexpr (QMX) {
      q = reinterpret_cast<const uint8_t*>(qb)[(uint32_t)(row_c * p.ngroups + g)];
      q = ok ? q : 256u;  // 256 = "this lane contributes zeros"
    } else {
      q = reinterpret_cast<const uint32_t*>(qb)[(uint32_t)(g * p.wrows + row_c)];
      q = ok ? q : 0u;    // scale = zero = 0
    }
    sl.q = q;
    const uint32_t xo = xlane + (uint32_t)((s * KSTEP - (kk - kk_c)) * 2);
#pragma unroll
    for (int h = 0; h < NMMA; ++h) {
      if constexpr (ABL == 2) sl.x[h] = u32x4{xo, 1u, 2u, 3u};  // ablation: no X loads
      else if (xcol) sl.x[h] = *reinterpret_cast<const u32x4*>(xb + (xo + 16u * h));
      else sl.x[h] = u32x4{0u, 0u, 0u, 0u};  // unused MFMA column: zero operand (its output is never stored)
    }
  };

  Slot<NMMA> slots[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nsteps) load_slot(d, slots[d]);

  // ---- this wave's f32 LUT image, built while the first weight loads are in flight ----
  float* tab = s_tab + wave * (16 * 64);
  {
    float tv[16];
    if (p.qtype == TG_Q_INT4) {
#pragma unroll
      for (int e = 0; e < 16; ++e) tv[e] = (float)(e - 8);
    } else if (p.qtype == TG_Q_MX4) {
#pragma unroll
      for (int e = 0; e < 16; ++e) tv[e] = (e & 8 ? -1.f : 1.f) * ((e & 7) < 5 ? 0.5f * (e & 7) : ((e & 7) == 5 ? 3.f : (e & 7) == 6 ? 4.f : 6.f));
    } else {
      // 16 entries x 16 bit = 32 bytes per LUT row; global LUT: the same row for every lane
      const char* lrow = lb + (p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)row_c * 32 : 0);
      const u32x4 l0 = reinterpret_cast<const u32x4*>(lrow)[0];
      const u32x4 l1 = reinterpret_cast<const u32x4*>(lrow)[1];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t pair = e < 8 ? l0[e >> 1] : l1[(e - 8) >> 1];
        tv[e] = (e & 1) ? DT::hi_f32(pair) : DT::lo_f32(pair);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) tab[e * 64 + lane] = tv[e];
  }

  // LDS byte address of entry e for this lane = tabbase | e << 8 | lane << 2, tabbase % 4096 == 0.
  // One v_perm_b32 builds it: byte 0 <- lane*4, byte 1 <- (tabbase bits 12..15) << 4 | nibble,
  // bytes 2,3 <- tabbase bytes 2,3.
  const uint32_t tabbase = (uint32_t)reinterpret_cast<uintptr_t>(tab);  // low 32 bits of an LDS generic address = LDS offset
  const uint32_t lane4 = (uint32_t)lane * 4u | (tabbase & 0xffff0000u);
  const uint32_t kmask = __builtin_amdgcn_readfirstlane(((tabbase >> 12) & 0xfu) * 0x10101010u);
  const uint32_t sh = LAYOUT_A ? (uint32_t)(i >> 3) * 4u : 0u;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

  auto lookup = [&](uint32_t src, int byte) -> float {
    const uint32_t addr = __builtin_amdgcn_perm(src, lane4, 0x03020400u + ((uint32_t)byte << 8));
    if constexpr (ABL == 1) return u2f(addr);  // ablation: no LDS lookup
    return *(lds_cfptr)(addr);
  };

  auto process = [&](const Slot<NMMA>& sl) {
    if constexpr (ABL == 4) {  // ablation: stream only, no dequant / MFMA
      acc[0] += u2f(sl.w[0] ^ sl.w[1] ^ sl.w[2] ^ sl.w[3] ^ sl.q ^ sl.x[0][0] ^ sl.x[NMMA - 1][3]);
      return;
    }
    u32x4 w = sl.w;
    canonicalize<CANON>(w);
    float s, z;
    if constexpr (QMX) {
      // e8m0: 2^(e-127), 255 -> NaN (reference Dequantization.cuh:331-339); 256 = padding lane
      const uint32_t e = sl.q;
      s = u2f(e == 255u ? 0x7fc00000u : (e == 0u ? 0x00400000u : (e << 23)));
      s = e == 256u ? 0.f : s;
      z = 0.f;
    } else {
      s = DT::lo_f32(sl.q);
      z = DT::hi_f32(sl.q);
    }
    if constexpr (!LAYOUT_A) {
      // Bint4 word: nibble p holds v[e], p = {0,4,1,5,2,6,3,7}[e]; v[e] is k = 2q + 8(e>>1) + (e&1)
      uint32_t wa[4], wb4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        wa[q] = (w[q] & 0x0f0f0f0fu) | kmask;          // bytes: v0 v4 v1 v5
        wb4[q] = ((w[q] >> 4) & 0x0f0f0f0fu) | kmask;  // bytes: v2 v6 v3 v7
      }
      if constexpr (ABL >= 8) {
        // experiment: all 32 lookups in flight before the first fma
        float f[32];
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t src = (h & 1) ? wb4[q] : wa[q];
            f[h * 8 + q * 2] = lookup(src, h >> 1);
            f[h * 8 + q * 2 + 1] = lookup(src, (h >> 1) + 2);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          u32x4 a;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            a[q] = DT::pack2(__builtin_fmaf(f[h * 8 + q * 2], s, z), __builtin_fmaf(f[h * 8 + q * 2 + 1], s, z));
          acc = DT::mfma(a, sl.x[h], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        return;
      }
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        u32x4 a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t src = (h & 1) ? wb4[q] : wa[q];
          const float f0 = lookup(src, h >> 1);        // v[2h]   -> k = 8h + 2q
          const float f1 = lookup(src, (h >> 1) + 2);  // v[2h+1] -> k = 8h + 2q + 1
          a[q] = DT::pack2(__builtin_fmaf(f0, s, z), __builtin_fmaf(f1, s, z));
        }
        acc = DT::mfma(a, sl.x[h], acc);
      }
    } else {
      // Aint4 word: low nibbles = row r (k0 k2 k1 k3 in bytes 0..3), high nibbles = row r + 8
      uint32_t ws[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) ws[q] = ((w[q] >> sh) & 0x0f0f0f0fu) | kmask;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x4 a;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float f0 = lookup(ws[q], h);      // k = 8h + 2q
          const float f1 = lookup(ws[q], h + 2);  // k = 8h + 2q + 1
          a[q] = DT::pack2(__builtin_fmaf(f0, s, z), __builtin_fmaf(f1, s, z));
        }
        acc = DT::mfma(a, sl.x[h], acc);
      }
    }
  };

  // Slot d is consumed, then immediately refilled with step j + DEPTH: DEPTH - 1 steps stay in
  // flight while one is being dequantised.  All branches here are wave-uniform.
  for (int jb = 0; jb < nsteps; jb += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int j = jb + d;
      if (j < nsteps) {
        process(slots[d]);
        if (j + DEPTH < nsteps) load_slot(j + DEPTH, slots[d]);
      }
    }
  }

  // ---- split-K tail ----
  // MFMA C/D: lane (c = lane & 15, Q) holds y[act row c][weight rows row0 + 4Q .. +3]
  if (p.splitk > 1) {
    s_red[wave * 64 + lane] = acc;
    __syncthreads();
    if (slice != 0) return;
    for (int o = 1; o < p.splitk; ++o) {  // fixed order: deterministic
      const f32x4 v = s_red[(wave + o) * 64 + lane];
      acc += v;
    }
  }
  const int col = ct * 16 + i;
  const int rowg = row0 + 4 * Q;
  if (rt_ok && col < p.m && rowg < p.wrows) {  // wrows % 8 == 0 and rowg % 4 == 0: all four rows valid
    store_rows4<DT>(yb, p.bias ? p.bias + b * p.stride_bias + (int64_t)col * p.bias_row_stride * 2 : nullptr, (int64_t)col * p.wrows + rowg, rowg, acc);
  }
}
