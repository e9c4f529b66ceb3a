// w4_gemm_tile.cuh -- the W4A16 GEMM for MANY activation rows (a prefill through the modules: m > 64): an LDS-tiled MFMA GEMM that
// dequantises the Bint4 weights on the way into its B tile.
//
// Reference: the one kernel of the reference walks any m, 16 rows per block and pass over the weights (TinyGemmImpl.cuh:379-392); its
// arithmetic per weight is w = RNE16(fma(f32(lut[row][code]), f32(scale[g][row]), f32(zero[g][row]))) (MatrixLayoutB.cuh:1042-1046;
// int4: lut = code - 8, Dequantization.cuh:136-178), products summed in f32, one rounding of the result.  This kernel computes exactly
// those weights (bit-equal to tg_dequant_w4 and the oracle) and contracts them with v_mfma_f32_16x16x32_{bf16,f16}.
//
//   tile       BM = 128 activation rows x BN = 64 / 128 weight rows x BK = 64 (one super-tile of innerKTiles 4); 12 or 16 waves, ONE ROLE
//              each: waves 0 ... 3 CONSUME (2 (m) x 2 (n): a wave owns 64 x BN/2 outputs; fragment reads and MFMAs only), the next 4
//              request the x tiles (LDS-DMA), the last 4 / 8 request the packed words and scale / zero, build the tables and
//              dequantise.  Every role is a chain of dependent LDS / memory round trips of 0.2-0.3 us per step; one wave doing two of
//              them pays their SUM (all roles on every wave: 2300 cycles per step, 261 TFLOP/s at m = 512), side by side on the
//              SIMDs they overlap.
//   operands   the WEIGHTS are the MFMA's A operand (lane (i = lane & 15, kq = lane >> 4): weight row i, k = 8 kq ... 8 kq + 7), the
//              activations its B operand, so that D[i = weight row][j = activation row] puts FOUR CONSECUTIVE weight rows of one
//              activation row into a lane's accumulator registers: 8-byte stores of y[m][n ... n + 3].
//   x tile     global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers, no ds_write), three
//              stages; rows of 128 bytes whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7 ON THE SOURCE ADDRESS (the DMA writes
//              lane-linear), so that the ds_read_b128 of a fragment (16 rows x one chunk) touches all 64 banks once.
//   w tile     every thread takes whole packed words (8 codes of ONE weight row: k = 2 i + {0, 1} + 8 h, h = 0 ... 3, of a 32-k run,
//              TinyGemmConvertB.cu:252-308) -- each word is loaded exactly once per tile -- looks its codes up in the per-(row, group)
//              table of FINAL 16-bit values (16 entries, built once per group with the reference's fma: one 2-byte LDS read per weight
//              instead of a select tree and an fma) and writes four 4-byte pieces into the swizzled tile.  Lane = (row & 7, word):
//              a 32-lane LDS access group touches 8 tables = 8 x 8 banks, and 32 distinct banks of the tile.
//   pipeline   step s: DMA of x for step s + 2, packed words of step s + 2 into registers, dequantisation of step s + 1 into the other
//              w buffer, the tables of the group that starts at step s + 2, the MFMAs of step s; one barrier per step.
//   grid       one workgroup per tile; tile index -> (n tile, m tile) with the m tiles of one n tile on ONE XCD (block b runs on XCD
//              b % 8: observed, used for speed only), so that the packed weights of an n tile leave HBM once.
#pragma once
#ifndef TILE_ABL
#define TILE_ABL 0  // developer ablations (timing only, wrong results), bit mask: 1 no dequantisation, 2 no MFMA stage, 4 no LDS-DMA, 8 no table builds
#endif

struct TileParams {
  const char* x;       // [m][k] 16-bit, row-major
  const char* w;       // Bint4 words, innerKTiles 4: [wrows / 8][k / 64][32][2] uint32
  const char* qinfo;   // [k / g][wrows][2] 16-bit (scale, zero)
  const char* lut;     // [wrows][16] (row-wise) / [16] (global) 16-bit, nullptr for int4
  char* y;             // [m][wrows] 16-bit
  const char* bias;    // optional [wrows] 16-bit
  int32_t m, wrows, k, ksuper, gshift, qtype;
  int32_t tiles_m, tiles_n;
};

template <int BM, int BN, int DX, int EW>
struct TileLds {
  // x of step u (L2-resident after its first touch) is requested DX steps ahead (live DX + 1 steps); the packed words and scale / zero come
  // from HBM (every byte is read once): DW = DX + 1 + EW and DW + 1 steps ahead, consumed one / two steps early (live DW steps each;
  // the scale / zero ring has one more slot: the prologue requests steps 0 ... DW at once)
  static constexpr int NST = DX + 1;
  static constexpr int DW = DX + 1 + EW;
  static constexpr int A_STAGE = BM * 128;           // BM rows x 64 k x 2 bytes
  static constexpr int B_STAGE = BN * 128;
  static constexpr int T_BUF = 2 * BN * 32;          // [sub-group of the step: 2][row][16 entries] 16-bit
  static constexpr int W_STAGE = BN * 32;            // the packed words of one step: BN rows x 64 k x 4 bits
  static constexpr int Q_STAGE = 2 * BN * 4;         // (scale, zero) of the step's group(s): [sub-group][row] 4 bytes
  static constexpr int A_OFF = 0;
  static constexpr int B_OFF = NST * A_STAGE;
  static constexpr int T_OFF = B_OFF + 2 * B_STAGE;
  static constexpr int W_OFF = T_OFF + 2 * T_BUF;
  static constexpr int Q_OFF = W_OFF + DW * W_STAGE;
  static constexpr int BYTES = Q_OFF + (DW + 1) * Q_STAGE;
  static_assert(BYTES <= 160 * 1024, "LDS");
};

// s_waitcnt vmcnt(N) with a literal N (the LDS-DMA queue is counted by hand: hipcc would drain it, cdna_hip_programming.md section 5)
template <int N>
__device__ __forceinline__ void tile_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tile_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename DT, int BM, int BN, int NPW = 12, int DX = 3, int EW = 2>
__global__ void __launch_bounds__(256 + 64 * NPW) w4_gemm_tile_kernel(const TileParams p) {
  constexpr int WN = BN / 2;             // weight rows of a consumer wave
  constexpr int NT = WN / 16;            // its 16-row tiles
  constexpr int MT = BM / 32;            // 16-row tiles of its BM / 2 activation rows
  constexpr int XPW = BM / 32;           // x DMA instructions (1 KiB = 8 rows) per step of the producer waves 0 ... 3
  constexpr int NDW = NPW - 4;           // producer waves 4 ... NPW - 1 dequantise (and request the packed words and scale / zero); 0 ... 3 only request x
  constexpr int WPT = BN / 8 / NDW;      // 8-row tiles per dequantising wave = packed words per thread and step
  constexpr int RPW = BN / NDW;          // table rows per dequantising wave
  constexpr int TQ = 64 / RPW;           // threads per table row
  constexpr int EPT = 16 / TQ;           // table entries per thread
  static_assert(NPW == 8 || NPW == 12, "4 waves that request x + 4 or 8 that dequantise");
  static_assert(WPT >= 1 && EPT >= 2 && EPT <= 8, "tile / producer split");
  using L = TileLds<BM, BN, DX, EW>;
  static_assert(BM == 64 || BM == 128, "activation rows per tile");
  constexpr int NST = L::NST, DW = L::DW, DQ = DW + 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wave_all >= 4;
  const int wave = producer ? wave_all - 4 : wave_all;     // index within the role
  // ---- tile of this workgroup: consecutive tiles on ONE XCD (block b runs on XCD b % 8) ----
  const int ntot = p.tiles_m * p.tiles_n;
  int tile;
  {
    const int b = blockIdx.x, q = ntot >> 3, r = ntot & 7, xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for every ntot (cdna_hip_programming.md T1)
  }
  const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ksteps = p.ksuper;                                 // one super-tile of 64 k per step
  const int gshift = p.gshift;
  const int spg_shift = gshift > 6 ? gshift - 6 : 0;           // steps per group (g = 128: 2, 256: 4; g <= 64: 1)
  const int nsub = gshift == 5 ? 2 : 1;                        // groups per step (g = 32: 2)
  const int ntiles8 = p.wrows >> 3;

  // ---- per-thread constants of the producer roles ----
  // (a) x DMA: wave w, instruction q: rows (XPW w + q) 8 ... + 8; lane: row + (lane >> 3), LDS slot lane & 7 <- global chunk slot ^ f(row)
  const char* xsrc[XPW];
#pragma unroll
  for (int q = 0; q < XPW; ++q) {
    const int row = ((wave & 3) * XPW + q) * 8 + (lane >> 3);
    int mr = m0 + row;
    mr = mr < p.m ? mr : p.m - 1;                               // (rows beyond m: a valid row, never stored)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    xsrc[q] = p.x + ((int64_t)mr * p.k) * 2 + chunk * 16;
  }
  // (b) words DMA: BN / 32 full-wave instructions per step (1 KiB = four 8-row tiles x 256 contiguous bytes), one each for the
  //     "cold" producer waves 4 ... 4 + BN / 32 - 1: lane l: tile 4 c + (l >> 4), 16-byte piece l & 15.  (c) scale / zero DMA: (BN / 64) nsub
  //     instructions of 64 rows x 4 bytes, one each for the first of the cold waves.  A wave's requests retire IN ORDER (vmcnt), so the
  //     waves that request x (L2 hits, needed two steps later) never wait behind an HBM miss: measured, one 8- / 16-lane words and scale /
  //     zero request on every wave cost 0.1 us per step each.
  constexpr int NWD = BN / 32;           // cold waves
  static_assert(NWD <= NDW, "the packed words are requested by the first BN / 32 dequantising waves");
  const int nqd = (BN / 64) * nsub;      // scale / zero requests per step (<= NWD)
  const bool x_wave = wave < 4, w_wave = wave >= 4 && wave - 4 < NWD;
  const int cold = w_wave ? wave - 4 : 0;
  const bool q_wave = w_wave && cold < nqd;
  const char* wsrc;
  {
    int gt = (n0 >> 3) + 4 * cold + (lane >> 4);
    gt = gt < ntiles8 ? gt : ntiles8 - 1;
    wsrc = p.w + ((int64_t)gt * p.ksuper) * 256 + (lane & 15) * 16;
  }
  const char* qsrc;
  const int q_idx = q_wave ? cold : 0, q_sub = q_idx / (BN / 64), q_half = q_idx % (BN / 64);
  {
    int gr = n0 + 64 * q_half + lane;
    gr = gr < p.wrows ? gr : p.wrows - 1;
    qsrc = p.qinfo + (int64_t)gr * 4;
  }
  const int ngroups = p.k >> gshift;
  // (d) dequantisation: word u of the thread: 8-row tile (wave * WPT + u) of the BN rows, row (lane & 7), word (lane >> 3) = 2 i + j
  const int dw = wave >= 4 ? wave - 4 : 0;    // index among the dequantising waves
  const int drow8 = lane & 7, dword = lane >> 3, di = dword >> 1, dj = dword & 1;
  uint32_t wrd_off[WPT];     // byte offset of the thread's word in a words stage
  uint32_t dst_off[WPT];     // byte offset of the thread's 4-byte piece h = 0 in a w stage (without the chunk term)
  uint32_t tab_off[WPT];     // byte offset of its row's table in a table buffer (sub-group 0)
  uint32_t dswz[WPT];
#pragma unroll
  for (int u = 0; u < WPT; ++u) {
    const int t8 = dw * WPT + u;
    const int row = t8 * 8 + drow8;
    wrd_off[u] = (uint32_t)(t8 * 256 + (4 * drow8 + di) * 8 + dj * 4);
    dst_off[u] = (uint32_t)(row * 128 + 4 * di);
    dswz[u] = (uint32_t)((row >> 1) & 7);
    tab_off[u] = (uint32_t)(row * 32);
  }
  // (e) tables: row RPW wave + lane / TQ, entries EPT (lane % TQ) ... + EPT - 1
  const int t_row = RPW * dw + lane / TQ, t_e0 = (lane % TQ) * EPT;
  float lv[EPT];
  if (producer && wave >= 4) {
    int gr = n0 + t_row;
    gr = gr < p.wrows ? gr : p.wrows - 1;
    if (p.qtype == TG_Q_INT4) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) lv[e] = (float)(t_e0 + e - 8);
    } else {
      const uint32_t* lp = reinterpret_cast<const uint32_t*>(p.lut + ((p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)gr * 16 : 0) + t_e0) * 2);
#pragma unroll
      for (int e = 0; e < EPT; e += 2) {
        const uint32_t pr = lp[e / 2];
        lv[e] = DT::lo_f32(pr);
        lv[e + 1] = DT::hi_f32(pr);
      }
    }
  }

  // One step's LDS-DMA requests of this wave: x of step sx (waves 0 ... 3), packed words of step sw and scale / zero of step sq (cold
  // waves); each clamped to the last step: a tail step re-requests data nobody reads, so that every step issues the same number of
  // requests per wave and ONE literal vmcnt per role fits every step
  auto dma = [&](int sx, int sw, int sq) {
    const int last = ksteps - 1;
    const int cx = sx < last ? sx : last, cw = sw < last ? sw : last, cq = sq < last ? sq : last;
    if (x_wave && !(TILE_ABL & 16)) {
      char* adst = lds + L::A_OFF + (sx % NST) * L::A_STAGE + wave * XPW * 1024;
#pragma unroll
      for (int q = 0; q < XPW; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + (int64_t)cx * 128),
                                         (__attribute__((address_space(3))) void*)(adst + q * 1024), 16, 0, 0);
    }
    if (w_wave && !(TILE_ABL & 32))
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (int64_t)cw * 256),
                                       (__attribute__((address_space(3))) void*)(lds + L::W_OFF + (sw % DW) * L::W_STAGE + cold * 1024), 16, 0, 2);
    if (q_wave && !(TILE_ABL & 64)) {
      int g = ((cq * 64) >> gshift) + q_sub;
      g = g < ngroups ? g : ngroups - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + (int64_t)g * p.wrows * 4),
                                       (__attribute__((address_space(3))) void*)(lds + L::Q_OFF + (sq % (DW + 1)) * L::Q_STAGE + (q_sub * BN + 64 * q_half) * 4), 4, 0, 0);
    }
  };
  // end of step t: x(u) must have landed for u <= t + 1 (requested at u - DX: the last DX - 1 steps' requests may be in flight);
  // words(u) for u <= t + 2 (requested at u - DW: DW - 2 in flight), scale / zero(u) for u <= t + 3 (requested at u - DW - 1: DW - 2)
  auto wait_dma = [&]() {
    if (x_wave) tile_wait_vm<(DX - 1) * XPW>();
    else if (q_wave) tile_wait_vm<2 * (DW - 2)>();
    else if (w_wave) tile_wait_vm<DW - 2>();
  };
  auto build_tables = [&](int step) {   // the tables of the group(s) of k-step `step` into buffer (step >> spg_shift) & 1
    const uint32_t tb = L::T_OFF + (uint32_t)(((step >> spg_shift) & 1) * L::T_BUF);
    const char* qst = lds + L::Q_OFF + (step % (DW + 1)) * L::Q_STAGE;
    for (int sub = 0; sub < nsub; ++sub) {
      const uint32_t sz = *reinterpret_cast<const uint32_t*>(qst + (sub * BN + t_row) * 4);
      const float sc = DT::lo_f32(sz), z = DT::hi_f32(sz);
      uint32_t o[EPT / 2];
#pragma unroll
      for (int e = 0; e < EPT; e += 2) o[e / 2] = DT::pack2(__builtin_fmaf(lv[e], sc, z), __builtin_fmaf(lv[e + 1], sc, z));
      char* dst = lds + tb + sub * (BN * 32) + t_row * 32 + t_e0 * 2;
      if constexpr (EPT == 2) *reinterpret_cast<uint32_t*>(dst) = o[0];
      else if constexpr (EPT == 4) *reinterpret_cast<u32x2*>(dst) = u32x2{o[0], o[1]};
      else *reinterpret_cast<u32x4*>(dst) = u32x4{o[0], o[1], o[2], o[3]};
    }
  };
  auto new_group = [&](int step) { return nsub == 2 || step == 0 || ((step >> spg_shift) != ((step - 1) >> spg_shift)); };
  auto dequant = [&](int step) {  // the words of k-step `step` -> w stage step & 1
    const uint32_t tb = L::T_OFF + (uint32_t)(((step >> spg_shift) & 1) * L::T_BUF) + (nsub == 2 ? dj * (BN * 32) : 0);
    const char* wst = lds + L::W_OFF + (step % DW) * L::W_STAGE;
    char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const uint32_t wd = *reinterpret_cast<const uint32_t*>(wst + wrd_off[u]);
      const char* tab = lds + tb + tab_off[u];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const uint32_t c0 = (wd >> (4 * h)) & 15u, c1 = (wd >> (16 + 4 * h)) & 15u;
        const uint32_t v = (uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c0) |
                           ((uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c1) << 16);
        const uint32_t chunk = (uint32_t)(4 * dj + h) ^ dswz[u];
        *reinterpret_cast<uint32_t*>(bst + dst_off[u] + chunk * 16) = v;
      }
    }
  };

  // ---- MFMA role: wave (wm, wn) ----
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, kq = lane >> 4;
  uint32_t a_off[MT], b_off[NT], a_swz[MT], b_swz[NT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int row = wm * (BM / 2) + t * 16 + fi;
    a_off[t] = (uint32_t)(row * 128);
    a_swz[t] = (uint32_t)((row >> 1) & 7);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = wn * WN + t * 16 + fi;
    b_off[t] = (uint32_t)(row * 128);
    b_swz[t] = (uint32_t)((row >> 1) & 7);
  }
  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto mma = [&](int step) {
    const char* ast = lds + L::A_OFF + (step % NST) * L::A_STAGE;
    const char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
    u32x4 wf[2][NT], xf[2][MT];      // every fragment of the step requested up front: the second half's reads land under the first half's MFMAs
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wf[kb][t] = *reinterpret_cast<const u32x4*>(bst + b_off[t] + (((uint32_t)(4 * kb + kq) ^ b_swz[t]) * 16));
#pragma unroll
      for (int t = 0; t < MT; ++t) xf[kb][t] = *reinterpret_cast<const u32x4*>(ast + a_off[t] + (((uint32_t)(4 * kb + kq) ^ a_swz[t]) * 16));
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = DT::mfma(wf[kb][a], xf[kb][b], acc[a][b]);
  };

  // Schedule (u = a k-step).  Data of step u is consumed: x by the MFMAs in step u; the dequantised weights by the MFMAs in step u,
  // so the dequantisation runs in step u - 1 and needs the words and the tables a barrier earlier: tables built in step u - 2 from
  // scale / zero that landed by the end of step u - 3.  A DMA issued in step t has landed for everyone after the barrier of step
  // t + DX - 1 (vmcnt leaves the requests of the last DX - 1 steps outstanding).  So step t requests x(t + DX); the packed words and
  // scale / zero are HBM misses (~0.4 us idle, 1-2 us under load: several steps of this loop) and are requested DW = DX + 1 + EW and
  // DW + 1 steps ahead by waves that request nothing else (wait_dma).
  // Both roles pass the same barriers: three in the prologue, one per step.
  if (producer) {
    // ---- prologue: "steps" -DQ ... -1 (requests of steps before 0 go to step 0: duplicates of the same bytes) ----
#pragma unroll
    for (int t = -DQ; t < 0; ++t) dma(t + DX > 0 ? t + DX : 0, t + DW > 0 ? t + DW : 0, t + DQ);
    tile_wait_vm<0>();
    tile_barrier();
    if (!x_wave) {
      build_tables(0);
      if (ksteps > 1 && new_group(1)) build_tables(1);
    }
    tile_barrier();
    if (!x_wave) dequant(0);
    tile_barrier();
    for (int s = 0; s < ksteps; ++s) {
      if (!(TILE_ABL & 4)) dma(s + DX, s + DW, s + DQ);
      if (!x_wave) {
        if (!(TILE_ABL & 1) && s + 1 < ksteps) dequant(s + 1);
        if (!(TILE_ABL & 8) && s + 2 < ksteps && new_group(s + 2)) build_tables(s + 2);
      }
      if (!(TILE_ABL & 4)) wait_dma();
      tile_barrier();
    }
    tile_wait_vm<0>();
    return;
  }
  tile_barrier();
  tile_barrier();
  tile_barrier();
  for (int s = 0; s < ksteps; ++s) {
    if (!(TILE_ABL & 2)) mma(s);
    tile_barrier();
  }

  // ---- store: lane (j = activation row fi of tile b, weight rows 4 kq ... 4 kq + 3 of tile a) ----
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int mr = m0 + wm * (BM / 2) + b * 16 + fi;
    if (mr >= p.m) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int nr = n0 + wn * WN + a * 16 + 4 * kq;
      if (nr >= p.wrows) continue;   // (wrows is a multiple of 8: a group of four rows is inside or outside)
      store_rows4<DT>(p.y, p.bias, (int64_t)mr * p.wrows + nr, nr, acc[a][b]);
    }
  }
}
