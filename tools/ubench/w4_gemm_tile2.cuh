// w4_gemm_tile2.cuh (developer experiment, NOT shipped: measured equal at 128 x 128 tiles and 10-15 % slower at 128 x 64) -- the tile GEMM of w4_gemm_tile.cuh with TWELVE waves and two roles (round 6, int4 / any4 only):
//
//   waves 0 ... 3   CONSUME: request the x tiles (LDS-DMA: the only vector-memory traffic of these waves, so the hand-counted vmcnt is exact),
//                   fragment reads and MFMAs -- with 768 threads a lane has 170 registers: the fragments of the NEXT 32-k block are read while
//                   the MFMAs of the current one run (at sixteen waves = 128 registers the 128 x 128 tile had room for one block's fragments:
//                   every block exposed the LDS latency in front of its 16 MFMAs)
//   waves 4 ... 11  dequantise: packed words and scale / zero words in register rings (plain loads: hipcc counts them), the per-(row, group)
//                   tables (four entries per thread and group), the lookups, the w tile
//
// Same tile shapes, LDS layout (TileLds), operand mapping, swizzles, split-K and schedule as w4_gemm_tile_kernel: step s: x of step s + DX
// requested; the words of step s + 1 dequantised into the other w buffer; the tables of the group that starts at step s + 2 built; the MFMAs
// of step s; ONE barrier per step.
#pragma once

template <typename DT, int BM, int BN, int DX = 3, int KS = 1>
__global__ void __launch_bounds__(768) w4_gemm_tile2_kernel(const TileParams p) {
  constexpr int NCW = 4, NDW = 8;
  constexpr int WN = BN / 2, NT = WN / 16, WMR = BM / 2, MT = WMR / 16;
  constexpr int XPW = BM / 32;           // x DMA instructions (1 KiB = 8 rows) per super-tile of each consumer wave
  constexpr int WPT = BN / 8 / NDW;      // packed words per dequantising thread and super-tile
  constexpr int NW = WPT * KS;
  constexpr int TPT = BN * 4 / (NDW * 64) > 0 ? BN * 4 / (NDW * 64) : 1;   // table (row, entry quad) items per dequantising thread (BN = 128: 1)
  constexpr bool THALF = BN * 4 < NDW * 64;                                 // BN = 64: only half the dequantising threads build tables
  constexpr int PWD = 4;
  using L = TileLds<BM, BN, DX, KS>;
  constexpr int KSH = KS == 1 ? 0 : 1;
  constexpr int NST = L::NST;
  constexpr int NSUB = L::NSUB;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nsplit = p.splits > 1 ? p.splits : 1;
  const int ntot = p.tiles_m * p.tiles_n * nsplit;
  int tile;
  {
    const int b = blockIdx.x, q = ntot >> 3, r = ntot & 7, xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tns = tile / p.tiles_m, tm = tile - tns * p.tiles_m;
  const int tn = tns / nsplit, split = tns - tn * nsplit;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ksuper_l = p.ksuper / nsplit;
  const int ks0 = split * ksuper_l;
  const int ksteps = ksuper_l >> KSH;
  const int last = ksteps - 1;
  const int gshift = p.gshift;
  const int spg_shift = gshift > 6 + KSH ? gshift - 6 - KSH : 0;
  const int nsub = gshift < 6 + KSH ? 1 << (6 + KSH - gshift) : 1;
  const int ngroups = (ksuper_l << 6) >> gshift;
  const int g0 = (ks0 << 6) >> gshift;
  auto new_group = [&](int step) { return nsub > 1 || step == 0 || ((step >> spg_shift) != ((step - 1) >> spg_shift)); };

  if (wave_all >= NCW) {
    // =================================== dequantising waves: words, scale / zero, tables, lookups, w tile ===================================
    const int dw = wave_all - NCW;
    const int dtid = dw * 64 + lane;
    const int ntiles8 = p.wrows >> 3;
    const int drow8 = lane >> 3, dword = lane & 7, di = dword >> 1, dj = dword & 1;
    const uint32_t* wsrc[WPT];
    uint32_t dst0[WPT][4];
    uint32_t tab_off[NW];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int t8 = dw * WPT + u;
      int gt = (n0 >> 3) + t8;
      gt = gt < ntiles8 ? gt : ntiles8 - 1;
      wsrc[u] = reinterpret_cast<const uint32_t*>(p.w) + (((int64_t)gt * p.ksuper + ks0) * 32 + 4 * drow8 + di) * 2 + dj;
      const int row = t8 * 8 + drow8;
#pragma unroll
      for (int h = 0; h < 4; ++h) dst0[u][h] = (uint32_t)(row * 128 + 4 * di) + (((uint32_t)(4 * dj + h) ^ (uint32_t)((row >> 1) & 7)) << 4);
#pragma unroll
      for (int pl = 0; pl < KS; ++pl) {
        const int sub = nsub > 1 ? (((2 * pl + dj) * 32) >> gshift) : 0;
        tab_off[u * KS + pl] = (uint32_t)(row * 32 + sub * (BN * 32));
      }
    }
    // tables: thread -> (row trow, entries 4 e4 ... 4 e4 + 3)
    const bool tact = !THALF || dtid < BN * 4;
    const int trow = (dtid >> 2) % BN, e4 = (dtid & 3) * 4;
    float lv[4];
    const uint32_t* qsrc;
    {
      int gr = n0 + trow;
      gr = gr < p.wrows ? gr : p.wrows - 1;
      qsrc = reinterpret_cast<const uint32_t*>(p.qinfo) + (int64_t)g0 * p.wrows + gr;
      if (p.qtype == TG_Q_INT4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) lv[e] = (float)(e4 + e - 8);
      } else {
        const u32x2 pr = *reinterpret_cast<const u32x2*>(p.lut + ((p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)gr * 16 : 0) + e4) * 2);
        lv[0] = DT::lo_f32(pr[0]); lv[1] = DT::hi_f32(pr[0]); lv[2] = DT::lo_f32(pr[1]); lv[3] = DT::hi_f32(pr[1]);
      }
    }
    uint32_t ring[PWD][NW];
    uint32_t szr[PWD][NSUB];     // scale | zero of the groups of steps t + 1 ... (slot = step % PWD)
    auto load_words = [&](int step, uint32_t (&dst)[NW]) {
      const int c = step < last ? step : last;
#pragma unroll
      for (int u = 0; u < WPT; ++u)
#pragma unroll
        for (int pl = 0; pl < KS; ++pl) dst[u * KS + pl] = __builtin_nontemporal_load(wsrc[u] + (int64_t)(c * KS + pl) * 64);
    };
    auto load_sz = [&](int step, uint32_t (&dst)[NSUB]) {
      const int c = step < last ? step : last;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        int g = ((c * (64 * KS)) >> gshift) + (sub < nsub ? sub : 0);
        g = g < ngroups ? g : ngroups - 1;
        dst[sub] = qsrc[(int64_t)g * p.wrows];
      }
    };
    auto build_tables = [&](int step, const uint32_t (&sz)[NSUB]) {
      if (!tact) return;
      char* tb = lds + L::T_OFF + ((step >> spg_shift) & 1) * L::T_BUF + trow * 32 + (dtid & 3) * 8;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        if (sub < nsub) {
          const float sc = DT::lo_f32(sz[sub]), z = DT::hi_f32(sz[sub]);
          u32x2 o = {DT::pack2(__builtin_fmaf(lv[0], sc, z), __builtin_fmaf(lv[1], sc, z)), DT::pack2(__builtin_fmaf(lv[2], sc, z), __builtin_fmaf(lv[3], sc, z))};
          *reinterpret_cast<u32x2*>(tb + sub * (BN * 32)) = o;
        }
      }
    };
    auto dequant = [&](int step, const uint32_t (&wd)[NW]) {
      const char* tab0 = lds + L::T_OFF + ((step >> spg_shift) & 1) * L::T_BUF;
      char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
      uint32_t v[NW][4];
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const char* tab = tab0 + tab_off[u];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const uint32_t c0 = (wd[u] >> (4 * h)) & 15u, c1 = (wd[u] >> (16 + 4 * h)) & 15u;
          v[u][h] = (uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c0) | ((uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c1) << 16);
        }
      }
#pragma unroll
      for (int u = 0; u < WPT; ++u)
#pragma unroll
        for (int pl = 0; pl < KS; ++pl)
#pragma unroll
          for (int h = 0; h < 4; ++h) *reinterpret_cast<uint32_t*>(bst + pl * L::B_PLANE + dst0[u][h]) = v[u * KS + pl][h];
    };
    {
      uint32_t s0[NSUB], s1[NSUB];
      load_sz(0, s0);
      load_sz(1, s1);
#pragma unroll
      for (int j = 0; j < PWD; ++j) load_words(j, ring[j]);
#pragma unroll
      for (int j = 0; j < PWD; ++j) load_sz(2 + j, szr[(2 + j) % PWD]);
      build_tables(0, s0);
      if (ksteps > 1 && new_group(1)) build_tables(1, s1);
    }
    tile_barrier();            // (tables of steps 0 and 1 built)
    dequant(0, ring[0]);
    load_words(PWD, ring[0]);
    tile_barrier();
    for (int s = 0; s < ksteps; s += PWD) {
#pragma unroll
      for (int j = 0; j < PWD; ++j) {
        if (s + j >= ksteps) break;
        // step t = s + j: the words of step t + 1 (slot (j + 1) % PWD) -> w stage; the tables of step t + 2 (scale / zero slot (j + 2) % PWD)
        if (s + j + 1 < ksteps) dequant(s + j + 1, ring[(j + 1) % PWD]);
        load_words(s + j + 1 + PWD, ring[(j + 1) % PWD]);
        if (s + j + 2 < ksteps && new_group(s + j + 2)) build_tables(s + j + 2, szr[(j + 2) % PWD]);
        load_sz(s + j + 2 + PWD, szr[(j + 2) % PWD]);
        tile_barrier();
      }
    }
    return;
  }

  // =================================== consumer waves: x DMA, fragment reads, MFMAs ===================================
  const int wave = wave_all;
  const char* xsrc[XPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int row = (wave * XPW + q) * 8 + (lane >> 3);
    int mr = m0 + row;
    mr = mr < p.m ? mr : p.m - 1;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    xsrc[q] = p.x + ((int64_t)mr * p.x_pitch + ks0 * 64) * 2 + chunk * 16;
  }
  auto dma = [&](int step) {
    const int c = step < last ? step : last;
    char* adst = lds + L::A_OFF + (step % NST) * L::A_STAGE + wave * XPW * 1024;
#pragma unroll
    for (int pl = 0; pl < KS; ++pl)
#pragma unroll
      for (int q = 0; q < XPW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + (int64_t)(c * KS + pl) * 128),
                                         (__attribute__((address_space(3))) void*)(adst + pl * L::A_PLANE + q * 1024), 16, 0, 0);
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, kq = lane >> 4;
  uint32_t a_base[2], b_base[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int ra = wm * WMR + fi, rb = wn * WN + fi;
    a_base[kb] = (uint32_t)(ra * 128) + (((uint32_t)(4 * kb + kq) ^ (uint32_t)((ra >> 1) & 7)) << 4);
    b_base[kb] = (uint32_t)(rb * 128) + (((uint32_t)(4 * kb + kq) ^ (uint32_t)((rb >> 1) & 7)) << 4);
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NB = 2 * KS;     // 32-k blocks per step
  auto mma = [&](int step) {
    const char* ast = lds + L::A_OFF + (step % NST) * L::A_STAGE;
    const char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
    u32x4 wf[2][NT], xf[2][MT];
    auto read = [&](int blk, int slot) {
      const int pl = blk >> 1, kb = blk & 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[slot][t] = *reinterpret_cast<const u32x4*>(bst + pl * L::B_PLANE + b_base[kb] + t * 2048);
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[slot][t] = *reinterpret_cast<const u32x4*>(ast + pl * L::A_PLANE + a_base[kb] + t * 2048);
    };
    read(0, 0);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      if (blk + 1 < NB) read(blk + 1, (blk + 1) & 1);      // the next block's fragments under this block's MFMAs
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = DT::mfma(wf[blk & 1][a], xf[blk & 1][b], acc[a][b]);
    }
  };
#pragma unroll
  for (int t = 0; t < DX; ++t) dma(t);
  tile_wait_vm<0>();
  tile_barrier();
  tile_barrier();
  for (int s = 0; s < ksteps; ++s) {
    dma(s + DX);
    mma(s);
    tile_wait_vm<(DX - 1) * XPW * KS>();
    tile_barrier();
  }
  tile_wait_vm<0>();
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int mr = m0 + wm * WMR + b * 16 + fi;
    if (mr >= p.m) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int nr = n0 + wn * WN + a * 16 + 4 * kq;
      if (nr >= p.wrows) continue;
      if (nsplit > 1) *reinterpret_cast<f32x4*>(p.part + ((int64_t)split * p.m + mr) * p.wrows + nr) = acc[a][b];
      else store_rows4<DT>(p.y, p.bias, (int64_t)mr * p.wrows + nr, nr, acc[a][b]);
    }
  }
}
