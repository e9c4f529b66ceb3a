// w4_gemm_tile.cuh -- the W4A16 GEMM for MANY activation rows (a prefill through the modules: m > 64): an LDS-tiled MFMA GEMM that
// dequantises the Bint4 weights on the way into its B tile.
//
// Reference: the one kernel of the reference walks any m, 16 rows per block and pass over the weights (TinyGemmImpl.cuh:379-392); its
// arithmetic per weight is w = RNE16(fma(f32(lut[row][code]), f32(scale[g][row]), f32(zero[g][row]))) (MatrixLayoutB.cuh:1042-1046;
// int4: lut = code - 8, Dequantization.cuh:136-178), products summed in f32, one rounding of the result.  This kernel computes exactly
// those weights (bit-equal to tg_dequant_w4 and the oracle) and contracts them with v_mfma_f32_16x16x32_{bf16,f16}.
//
//   tile       BM = 128 activation rows x BN = 64 / 128 weight rows x BK = 64 (one super-tile of innerKTiles 4); 16 waves, ONE ROLE each:
//                waves 0 ... 3   CONSUME: 2 (m) x 2 (n), a wave owns 64 x BN/2 outputs -- fragment reads and MFMAs -- and build the
//                                per-(row, group) tables (their scale / zero words come by plain loads four steps ahead: these waves have
//                                no other memory traffic, so hipcc's own vmcnt bookkeeping is exact)
//                waves 4 ... 7   request the x tiles (LDS-DMA), nothing else
//                waves 8 ... 15  dequantise: packed words by plain non-temporal loads into a register ring four steps ahead, table
//                                lookups, the w tile
//              Every role is a chain of dependent LDS / memory round trips per step; a wave that carries two of them pays their SUM
//              (all roles on every wave: 2300 cycles per step, 261 TFLOP/s at m = 512; words and scale / zero through LDS-DMA on the
//              dequantising waves, tables built there too: 1400 cycles, 0.37 PFLOP/s -- the ablations are in DESIGN.md section 9).
//   operands   the WEIGHTS are the MFMA's A operand (lane (i = lane & 15, kq = lane >> 4): weight row i, k = 8 kq ... 8 kq + 7), the
//              activations its B operand, so that D[i = weight row][j = activation row] puts FOUR CONSECUTIVE weight rows of one
//              activation row into a lane's accumulator registers: 8-byte stores of y[m][n ... n + 3].
//   x tile     global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers, no ds_write), DX + 1
//              stages; rows of 128 bytes whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7 ON THE SOURCE ADDRESS (the DMA writes
//              lane-linear), so that the ds_read_b128 of a fragment (16 rows x one chunk) touches all 64 banks once.  The DMA queue is
//              counted by hand (s_waitcnt vmcnt(N) with a literal N; hipcc would drain it at every barrier).
//   w tile     every thread takes whole packed words (8 codes of ONE weight row: k = 2 i + {0, 1} + 8 h, h = 0 ... 3, of a 32-k run,
//              TinyGemmConvertB.cu:252-308) -- each word is loaded exactly once per tile -- looks its codes up in the per-(row, group)
//              table of FINAL 16-bit values (16 entries, built once per group with the reference's fma: one 2-byte LDS read per weight
//              instead of a select tree and an fma) and writes four 4-byte pieces into the swizzled tile.  Lane = 8 row + word (the word's
//              place in the packed layout: contiguous wave-loads): a 32-lane LDS access group touches 4 tables = 4 x 8 of the 32 banks a
//              4-byte access sees.
//   pipeline   step s: x of step s + DX requested; the words of step s + 1 dequantised into the other w buffer; the tables of the group
//              that starts at step s + 2 built; the MFMAs of step s; ONE barrier per step.
//   grid       one workgroup per (tile, k-split); index -> (n tile, split, m tile) with the m tiles of one n tile and k range on ONE XCD
//              (block b runs on XCD b % 8: observed, used for speed only), so that the packed weights of an n tile leave HBM once.
//   split-K    TileParams::splits > 1: the workgroup covers ksuper / splits super-tiles from split * (ksuper / splits) on (a whole number of
//              steps and of quantisation groups: the host's job) and stores its f32 accumulators to part[split][m][rows];
//              tile_split_sum_kernel (below) adds the splits in order, applies the bias and rounds once.  Why: a tile's k-steps are a chain
//              of dependent LDS round trips, 0.6-0.9 us per 128 k whatever the tile holds -- a launch with 64 tiles takes as long as one with
//              256 (tg_tile.hip chooses the splits; profiles/r06_tile_splitk.txt).
//   mx4        template flag QMX: the table entries are fp4[code] * 2^(e - 127) (exact in bf16; e = 255: NaN), the exponents one byte per row
//              and 32-k group.  A flavour of its own because as a run-time branch it cost the other formats registers (128 x 128 tile: spill).
#pragma once
#ifndef TILE_ABL
#define TILE_ABL 0  // developer ablations (timing only, wrong results), bit mask: 1 no dequantisation, 2 no MFMA stage, 4 no LDS-DMA, 8 no table builds, 16 no lookups, 32 no w-tile writes, 64 no word loads
#endif

struct TileParams {
  const char* x;       // [m][k] 16-bit, row-major
  const char* w;       // Bint4 words, innerKTiles 4: [wrows / 8][k / 64][32][2] uint32
  const char* qinfo;   // [k / g][wrows][2] 16-bit (scale, zero); mx4: [wrows][k / 32] uint8 exponents
  const char* lut;     // [wrows][16] (row-wise) / [16] (global) 16-bit, nullptr for int4
  char* y;             // [m][wrows] 16-bit
  const char* bias;    // optional [wrows] 16-bit
  int32_t m, wrows, k, ksuper, gshift, qtype;
  int32_t tiles_m, tiles_n;
  // split-K (few tiles: m <= 256 against 256 CUs): `splits` workgroups per tile, each over ksuper / splits super-tiles, f32 partial tiles
  // [split][m][wrows] in `part`; tile_split_sum_kernel adds them in split order.  splits <= 1: the kernel stores y itself
  int32_t splits;
  float* part;
  int64_t x_pitch;     // elements between activation rows (>= k)
};

template <int BM, int BN, int DX, int KS>
struct TileLds {
  static constexpr int NST = DX + 1;                 // x of step u is requested DX steps ahead: live DX + 1 steps
  static constexpr int A_PLANE = BM * 128;           // BM rows x 64 k x 2 bytes: one super-tile of a step
  static constexpr int B_PLANE = BN * 128;
  static constexpr int A_STAGE = KS * A_PLANE;       // a step = KS super-tiles (KS x 64 k between two barriers)
  static constexpr int B_STAGE = KS * B_PLANE;
  static constexpr int NSUB = 2 * KS;                // quantisation groups per step at most (g = 32)
  static constexpr int T_BUF = NSUB * BN * 32;       // [sub-group of the step][row][16 entries] 16-bit
  static constexpr int A_OFF = 0;
  static constexpr int B_OFF = NST * A_STAGE;
  static constexpr int T_OFF = B_OFF + 2 * B_STAGE;
  static constexpr int BYTES = T_OFF + 2 * T_BUF;
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

// NDW = dequantising waves: 8 (one / two words per thread at BN = 64 / 128; 16 waves = 128 registers per lane: the consumers' 64 accumulator
// registers at BN = 128 fit because a step's fragments are read one k32 block at a time and their LDS offsets are two registers + constants;
// with all fragments up front they spilled 1 KiB per lane) or 4 (12 waves; measured 10 % slower at BN = 128)
// KS  = 64-k super-tiles per step (= per barrier): 1, or 2 where the LDS allows (128 x 64 tiles): every role's chain of dependent LDS /
//       memory round trips is paid once per step whatever the step holds -- waves wait 54 % of their cycles at KS = 1 (SQ_WAIT_ANY), the
//       LDS array is 41 % busy -- so two super-tiles per step nearly halve the time per k
// NCW = consuming waves: 4 (2 x 2, shipped) or 8 (4 (m) x 2 (n): two per SIMD, one's fragment reads under the other's MFMAs; with 16 waves per
//       workgroup that leaves 4 dequantising waves, which then bound the 128 x 128 tile: 62.7 vs 55 us at m = 1024 -- developer A/B only)
// QMX = mx4 weights (a template parameter: as a run-time branch its code cost the other formats 10 % at 128 x 64 tiles and spilled the 128 x 128 tile: 106 -> 207 us)
// W8  = int8 weights (tinygemm_y_f16RM_x_f16RM_w_int8TC at many rows; TinyGemm_int8.cu:216-399): 0 = 4-bit, 1 = Bint8 words of innerKTiles 2
//       ([n / 8][k / 32][32][2]: lane t, k-tile kt -> one word, bytes = k 2q, 2q + 8, 2q + 1, 2q + 9 of row t / 4, q = t % 4;
//       MatrixLayoutB.cuh:1104-1327), 2 = Aint8 words of innerKTiles 2 ([n / 16][k / 32][32][4]: lane t, k-tile kt -> two words, bytes =
//       (m0, k0) (m1, k0) (m0, k0 + 1) (m1, k0 + 1), the second word at k0 + 8; m0 = t / 4, m1 = m0 + 8, k0 = 2q).  No tables: the
//       dequantising waves compute w = RNE16(fma(byte - 128, scale, zero)) (Dequantization.cuh:262-330) in the vector ALU -- 2.75 operations
//       per weight against one LDS lookup -- from scale / zero words they prefetch themselves; one quantisation group per step (g >= 64 KS).
template <typename DT, int BM, int BN, int DX = 3, int NDW = 8, int KS = 1, int NCW = 4, bool QMX = false, int W8 = 0>
__global__ void __launch_bounds__(64 * (NCW + 4 + NDW)) w4_gemm_tile_kernel(const TileParams p) {
  constexpr int WN = BN / 2;             // weight rows of a consumer wave
  constexpr int NT = WN / 16;            // its 16-row tiles
  constexpr int WMR = BM / (NCW / 2);    // activation rows of a consumer wave
  constexpr int MT = WMR / 16;           // its 16-row tiles
  constexpr int XPW = BM / 32;           // x DMA instructions (1 KiB = 8 rows) per step of each of the four x waves
  constexpr int WPT = BN / 8 / NDW;      // 8-row tiles per dequantising wave = packed words per thread and step
  constexpr int RPT = BN * 4 / (NCW * 64);  // table rows per consumer thread (BN rows x 4 entry quads / the consumers' threads)
  constexpr int TRS = NCW * 16;          // table rows one pass of the consumers' threads covers
  static_assert(NCW == 4 || NCW == 8, "consuming waves");
  static_assert(RPT >= 1, "table rows per consumer thread");
  constexpr int PW = 4;                  // the register ring of scale / zero: steps ahead
#ifndef TILE_PWD
#define TILE_PWD 4
#endif
  constexpr int PWD = TILE_PWD;          // the register ring of the packed words: steps ahead (their requests queue behind DX steps of x DMA
                                         // requests in the CU's vector-memory path; 4 / 8 / 12 steps ahead measured equal)
  using L = TileLds<BM, BN, DX, KS>;
  constexpr int KSH = KS == 1 ? 0 : 1;   // log2(KS)
  static_assert(KS == 1 || KS == 2, "super-tiles per step");
  static_assert(BM == 64 || BM == 128 || BM == 256, "activation rows per tile");
  static_assert(BN == 64 || BN == 128, "weight rows per tile");
  constexpr int NST = L::NST;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int lane = threadIdx.x & 63, wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ---- tile of this workgroup: consecutive tiles on ONE XCD (block b runs on XCD b % 8) ----
  const int nsplit = p.splits > 1 ? p.splits : 1;
  const int ntot = p.tiles_m * p.tiles_n * nsplit;
  int tile;
  {
    const int b = blockIdx.x, q = ntot >> 3, r = ntot & 7, xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for every ntot (cdna_hip_programming.md T1)
  }
  // (n tile, split, m tile): the m tiles of one n tile and k range next to each other
  const int tns = tile / p.tiles_m, tm = tile - tns * p.tiles_m;
  const int tn = tns / nsplit, split = tns - tn * nsplit;
  const int m0 = tm * BM, n0 = tn * BN;
  const int ksuper_l = p.ksuper / nsplit;                      // this workgroup's super-tiles: split * ksuper_l ... (host: divides, a multiple of KS
  const int ks0 = split * ksuper_l;                            // and of the quantisation group)
  const int ksteps = ksuper_l >> KSH;                          // KS super-tiles of 64 k per step
  const int last = ksteps - 1;
  const int gshift = p.gshift;
  const int spg_shift = gshift > 6 + KSH ? gshift - 6 - KSH : 0;        // log2(steps per group)            (KS = 1: g = 128: 2, 256: 4)
  const int nsub = gshift < 6 + KSH ? 1 << (6 + KSH - gshift) : 1;      // groups per step                  (KS = 1: g = 32: 2)
  const int ngroups = (ksuper_l << 6) >> gshift;              // groups of this workgroup's k range, counted from g0
  const int g0 = (ks0 << 6) >> gshift;
  auto new_group = [&](int step) { return nsub > 1 || step == 0 || ((step >> spg_shift) != ((step - 1) >> spg_shift)); };
  // Schedule (u = a k-step).  Data of step u is consumed by the MFMAs in step u: x(u) lands by the end of step u - 1 (requested DX steps
  // ahead; vmcnt leaves the requests of the last DX - 1 steps in flight); the w tile of step u is written in step u - 1 from tables
  // built in step u - 2.  Every wave passes the same barriers: two in the prologue, one per step.

  if (wave_all >= NCW + 4) {
    // =================================== dequantising waves ===================================
    const int dw = wave_all - (NCW + 4);                                // 0 ... NDW - 1: owns the 8-row tiles dw * WPT ... of the BN rows
    if constexpr (W8 != 0) {
      // ---- int8: a wave-load is 512 consecutive bytes = one (8-row tile, 64 k) block of Bint8 / one (16-row tile, 32 k) block of Aint8 ----
      constexpr int UW = (W8 == 1 ? BN / 8 : BN / 16 * 2) / NDW;       // blocks per wave and super-tile
      constexpr int NW8 = UW * KS;
      constexpr int RW = W8 == 1 ? 1 : 2;                              // weight rows a lane's bytes belong to
      const int64_t k32 = (int64_t)p.ksuper * 2;                        // 32-k super-tiles of the packed layout (innerKTiles 2)
      const u32x2* wsrc[UW];
      uint32_t dst[UW][RW][2];    // byte offset in a plane of a w stage of (row rr, the piece of word / k-tile j); the second piece of it: ^ 16 in the chunk
      const uint32_t* qrow[UW][RW];
      int kt_or_sup;              // B: the lane's 32-k half of the block; A: its k-tile within the block's 32 k
#pragma unroll
      for (int u = 0; u < UW; ++u) {
        const int blk = dw * UW + u;
        if constexpr (W8 == 1) {
          const int sup = lane >> 5, t = lane & 31, q = t & 3;
          kt_or_sup = sup;
          int gt = (n0 >> 3) + blk;
          gt = gt < (p.wrows >> 3) ? gt : (p.wrows >> 3) - 1;
          wsrc[u] = reinterpret_cast<const u32x2*>(p.w) + ((int64_t)gt * k32 + 2 * ks0 + sup) * 32 + t;
          const int row = blk * 8 + (t >> 2);
          int gr = n0 + row;
          gr = gr < p.wrows ? gr : p.wrows - 1;
          qrow[u][0] = reinterpret_cast<const uint32_t*>(p.qinfo) + (int64_t)g0 * p.wrows + gr;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)   // word kt: k = 32 sup + 16 kt + 2q (+ 1) and + 8: chunks 4 sup + 2 kt and + 1, byte 4 q
            dst[u][0][kt] = (uint32_t)(row * 128 + 4 * q) + (((uint32_t)(4 * sup + 2 * kt) ^ (uint32_t)((row >> 1) & 7)) << 4);
        } else {
          const int t16 = blk >> 1, ko = blk & 1, t = lane >> 1, kt = lane & 1, q = t & 3;
          kt_or_sup = kt;
          int gt = (n0 >> 4) + t16;
          gt = gt < (p.wrows >> 4) ? gt : (p.wrows >> 4) - 1;
          wsrc[u] = reinterpret_cast<const u32x2*>(p.w) + (((int64_t)gt * k32 + 2 * ks0 + ko) * 32 + t) * 2 + kt;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int row = t16 * 16 + (t >> 2) + 8 * rr;
            int gr = n0 + row;
            gr = gr < p.wrows ? gr : p.wrows - 1;
            qrow[u][rr] = reinterpret_cast<const uint32_t*>(p.qinfo) + (int64_t)g0 * p.wrows + gr;
#pragma unroll
            for (int j = 0; j < 2; ++j)    // word j: k = 32 ko + 16 kt + 2q + 8 j (+ 1): chunk 4 ko + 2 kt + j, byte 4 q
              dst[u][rr][j] = (uint32_t)(row * 128 + 4 * q) + (((uint32_t)(4 * ko + 2 * kt + j) ^ (uint32_t)((row >> 1) & 7)) << 4);
          }
        }
      }
      (void)kt_or_sup;
      u32x2 ring[PWD][NW8];
      uint32_t szq[PWD][UW][RW];
      auto load_step = [&](int step, u32x2 (&wd)[NW8], uint32_t (&sz)[UW][RW]) {
        const int c = step < last ? step : last;
        int g = (c * (64 * KS)) >> gshift;
        g = g < ngroups ? g : ngroups - 1;
#pragma unroll
        for (int u = 0; u < UW; ++u) {
#pragma unroll
          for (int pl = 0; pl < KS; ++pl) wd[u * KS + pl] = __builtin_nontemporal_load(wsrc[u] + (int64_t)(c * KS + pl) * (W8 == 1 ? 64 : 128));
#pragma unroll
          for (int rr = 0; rr < RW; ++rr) sz[u][rr] = qrow[u][rr][(int64_t)g * p.wrows];
        }
      };
      auto deq = [&](uint32_t b, float sc, float z) { return __builtin_fmaf((float)b - 128.f, sc, z); };
      auto dequant8 = [&](int step, const u32x2 (&wd)[NW8], const uint32_t (&sz)[UW][RW]) {
        char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
#pragma unroll
        for (int u = 0; u < UW; ++u)
#pragma unroll
          for (int pl = 0; pl < KS; ++pl) {
            const u32x2 v = wd[u * KS + pl];
            char* pb = bst + pl * L::B_PLANE;
            if constexpr (W8 == 1) {
              const float sc = DT::lo_f32(sz[u][0]), z = DT::hi_f32(sz[u][0]);
#pragma unroll
              for (int kt = 0; kt < 2; ++kt) {   // bytes: k 2q, 2q + 8, 2q + 1, 2q + 9
                const uint32_t w = v[kt];
                *reinterpret_cast<uint32_t*>(pb + dst[u][0][kt]) = DT::pack2(deq(w & 0xffu, sc, z), deq((w >> 16) & 0xffu, sc, z));
                *reinterpret_cast<uint32_t*>(pb + (dst[u][0][kt] ^ 16u)) = DT::pack2(deq((w >> 8) & 0xffu, sc, z), deq(w >> 24, sc, z));
              }
            } else {
              const float s0 = DT::lo_f32(sz[u][0]), z0 = DT::hi_f32(sz[u][0]), s1 = DT::lo_f32(sz[u][1]), z1 = DT::hi_f32(sz[u][1]);
#pragma unroll
              for (int j = 0; j < 2; ++j) {      // bytes: (m0, k0) (m1, k0) (m0, k0 + 1) (m1, k0 + 1)
                const uint32_t w = v[j];
                *reinterpret_cast<uint32_t*>(pb + dst[u][0][j]) = DT::pack2(deq(w & 0xffu, s0, z0), deq((w >> 16) & 0xffu, s0, z0));
                *reinterpret_cast<uint32_t*>(pb + dst[u][1][j]) = DT::pack2(deq((w >> 8) & 0xffu, s1, z1), deq(w >> 24, s1, z1));
              }
            }
          }
      };
#pragma unroll
      for (int j = 0; j < PWD; ++j) load_step(j, ring[j], szq[j]);
      tile_barrier();
      dequant8(0, ring[0], szq[0]);
      load_step(PWD, ring[0], szq[0]);
      tile_barrier();
      for (int s = 0; s < ksteps; s += PWD) {
#pragma unroll
        for (int j = 0; j < PWD; ++j) {
          if (s + j >= ksteps) break;
          if (s + j + 1 < ksteps) dequant8(s + j + 1, ring[(j + 1) % PWD], szq[(j + 1) % PWD]);
          load_step(s + j + 1 + PWD, ring[(j + 1) % PWD], szq[(j + 1) % PWD]);
          tile_barrier();
        }
      }
      return;
    } else {
    const int ntiles8 = p.wrows >> 3;
    // lane = 8 (row of the tile) + (word 2 i + j of the row) = the word's own position in the tile's 256-byte block of the packed layout: a
    // wave-load is 64 CONSECUTIVE dwords (with lanes = (row & 3, word, row >> 2) a quad of adjacent lanes touched two 64-byte chunks: 8 x the
    // requests, and the eight wave-loads per step cost 5 of 32 us), and a 32-lane LDS access group (4-byte accesses: 32 banks) still looks up
    // in FOUR tables of 8 banks each (with eight rows per group, rows r and r + 4 met in the same banks: every lookup twice as long)
    const int drow8 = lane >> 3, dword = lane & 7, di = dword >> 1, dj = dword & 1;   // row of the tile, word 2 i + j of the row
    constexpr int NW = WPT * KS;   // words per thread and step: word u * KS + pl = 8-row tile u of the wave, super-tile pl of the step
    const uint32_t* wsrc[WPT];
    uint32_t dst0[WPT][4];     // byte offset of the thread's 4-byte piece h in a plane of a w stage
    uint32_t tab_off[NW];      // byte offset of its row's table in a table buffer (sub-group of the word's 32-k run)
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
        const int sub = nsub > 1 ? (((2 * pl + dj) * 32) >> gshift) : 0;   // the word's 32-k run starts (2 pl + j) 32 k into the step
        tab_off[u * KS + pl] = (uint32_t)(row * 32 + sub * (BN * 32));
      }
    }
    uint32_t ring[PWD][NW];     // words of steps t ... t + PW - 1 (slot = step % PWD)
    auto load_words = [&](int step, uint32_t (&dst)[NW]) {
      const int c = step < last ? step : last;                 // (past the end: the last step again, never used)
#pragma unroll
      for (int u = 0; u < WPT; ++u)
#pragma unroll
        for (int pl = 0; pl < KS; ++pl) {
          if constexpr (TILE_ABL & 64) dst[u * KS + pl] = (uint32_t)(c * 0x9e3779b9u) + (uint32_t)(uintptr_t)wsrc[u];
          else dst[u * KS + pl] = __builtin_nontemporal_load(wsrc[u] + (int64_t)(c * KS + pl) * 64);
        }
    };
    auto dequant = [&](int step, const uint32_t (&wd)[NW]) {   // the words of step `step` -> w stage step & 1
      const char* tab0 = lds + L::T_OFF + ((step >> spg_shift) & 1) * L::T_BUF;
      char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE;
      uint32_t v[NW][4];
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const char* tab = tab0 + tab_off[u];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const uint32_t c0 = (wd[u] >> (4 * h)) & 15u, c1 = (wd[u] >> (16 + 4 * h)) & 15u;
          if constexpr (TILE_ABL & 16) v[u][h] = c0 | (c1 << 16) | (uint32_t)(uintptr_t)tab;
          else v[u][h] = (uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c0) | ((uint32_t) * reinterpret_cast<const uint16_t*>(tab + 2 * c1) << 16);
        }
      }
#pragma unroll
      for (int u = 0; u < WPT; ++u)
#pragma unroll
        for (int pl = 0; pl < KS; ++pl)
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            if constexpr (TILE_ABL & 32) { if (v[u * KS + pl][h] == 0x12345u) *reinterpret_cast<uint32_t*>(bst) = 1u; }
            else *reinterpret_cast<uint32_t*>(bst + pl * L::B_PLANE + dst0[u][h]) = v[u * KS + pl][h];
          }
    };
#pragma unroll
    for (int j = 0; j < PWD; ++j) load_words(j, ring[j]);
    tile_barrier();            // (tables of steps 0 and 1 built)
    dequant(0, ring[0]);
    load_words(PWD, ring[0]);
    tile_barrier();
    for (int s = 0; s < ksteps; s += PWD) {
#pragma unroll
      for (int j = 0; j < PWD; ++j) {
        if (s + j >= ksteps) break;
        // step t = s + j: the words of step t + 1 are in slot (j + 1) % PWD; refilled with step t + 1 + PWD
        if (!(TILE_ABL & 1) && s + j + 1 < ksteps) dequant(s + j + 1, ring[(j + 1) % PWD]);
        load_words(s + j + 1 + PWD, ring[(j + 1) % PWD]);
        tile_barrier();
      }
    }
    return;
    }  // W8 == 0
  }

  if (wave_all >= NCW) {
    // =================================== x waves: LDS-DMA only ===================================
    const int xw = wave_all - NCW;
    const char* xsrc[XPW];
#pragma unroll
    for (int q = 0; q < XPW; ++q) {
      const int row = (xw * XPW + q) * 8 + (lane >> 3);
      int mr = m0 + row;
      mr = mr < p.m ? mr : p.m - 1;                             // (rows beyond m: a valid row, never stored)
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);          // LDS slot lane & 7 <- global chunk slot ^ f(row)
      xsrc[q] = p.x + ((int64_t)mr * p.x_pitch + ks0 * 64) * 2 + chunk * 16;
    }
    auto dma = [&](int step) {                                  // (past the end: the last step again, into a stage nobody reads)
      const int c = step < last ? step : last;
      char* adst = lds + L::A_OFF + (step % NST) * L::A_STAGE + xw * XPW * 1024;
#pragma unroll
      for (int pl = 0; pl < KS; ++pl)
#pragma unroll
        for (int q = 0; q < XPW; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[q] + (int64_t)(c * KS + pl) * 128),
                                           (__attribute__((address_space(3))) void*)(adst + pl * L::A_PLANE + q * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < DX; ++t) dma(t);
    tile_wait_vm<0>();
    tile_barrier();
    tile_barrier();
    for (int s = 0; s < ksteps; ++s) {
      if (!(TILE_ABL & 4)) {
        dma(s + DX);
        tile_wait_vm<(DX - 1) * XPW * KS>();
      }
      tile_barrier();
    }
    tile_wait_vm<0>();
    return;
  }

  // =================================== consumer waves: MFMAs and tables ===================================
  const int wave = wave_all, tid = wave * 64 + lane;
  // tables: row (tid >> 2) + TRS v, entries 4 (tid & 3) ... + 3
  float lv[RPT][4];
  const uint32_t* qsrc[RPT];
#pragma unroll
  for (int v = 0; v < RPT; ++v) {
    int gr = n0 + (tid >> 2) + TRS * v;
    gr = gr < p.wrows ? gr : p.wrows - 1;
    // (mx4, bf16 only: one exponent byte per row and 32-k group, row-major; the table entry is fp4[code] * 2^(e - 127), exact)
    if constexpr (QMX) qsrc[v] = reinterpret_cast<const uint32_t*>(p.qinfo + (int64_t)gr * (p.k >> 5) + g0);
    else qsrc[v] = reinterpret_cast<const uint32_t*>(p.qinfo) + (int64_t)g0 * p.wrows + gr;
    const int e4 = (tid & 3) * 4;
    if constexpr (W8 != 0) {      // (int8: no tables)
#pragma unroll
      for (int e = 0; e < 4; ++e) lv[v][e] = 0.f;
    } else if (p.qtype == TG_Q_INT4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) lv[v][e] = (float)(e4 + e - 8);
    } else if (QMX) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = e4 + e, mag = c & 7;
        lv[v][e] = (c & 8 ? -1.f : 1.f) * (mag < 5 ? 0.5f * mag : (mag == 5 ? 3.f : mag == 6 ? 4.f : 6.f));   // fp4-e2m1 (FloatDefs.cuh:18-34)
      }
    } else {
      const u32x2 pr = *reinterpret_cast<const u32x2*>(p.lut + ((p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)gr * 16 : 0) + e4) * 2);
      lv[v][0] = DT::lo_f32(pr[0]); lv[v][1] = DT::hi_f32(pr[0]); lv[v][2] = DT::lo_f32(pr[1]); lv[v][3] = DT::hi_f32(pr[1]);
    }
  }
  constexpr int NSUB = L::NSUB;
  uint32_t szr[PW][NSUB][RPT]; // scale | zero of the groups of steps t ... t + PW - 1 (slot = step % PW; up to NSUB sub-groups per step)
  auto load_sz = [&](int step, uint32_t (&dst)[NSUB][RPT]) {
    const int c = step < last ? step : last;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      int g = ((c * (64 * KS)) >> gshift) + (sub < nsub ? sub : 0);
      g = g < ngroups ? g : ngroups - 1;
      if constexpr (QMX) {
#pragma unroll
        for (int v = 0; v < RPT; ++v) {
          // scale = 2^(e - 127) as bf16 bits (Dequantization.cuh:331-346: 255 -> NaN, 0 -> the subnormal 2^-127), zero = -0.0: fma(v, s, -0) = v * s
          const uint32_t q = reinterpret_cast<const uint8_t*>(qsrc[v])[g];
          dst[sub][v] = (q == 255u ? 0x7fc0u : (q == 0u ? 0x0040u : (q << 7))) | 0x80000000u;
        }
      } else {
#pragma unroll
        for (int v = 0; v < RPT; ++v) dst[sub][v] = qsrc[v][(int64_t)g * p.wrows];
      }
    }
  };
  auto build_tables = [&](int step, const uint32_t (&sz)[NSUB][RPT]) {   // the tables of the group(s) of k-step `step` into buffer (step >> spg_shift) & 1
    char* tb = lds + L::T_OFF + ((step >> spg_shift) & 1) * L::T_BUF + (tid >> 2) * 32 + (tid & 3) * 8;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (sub < nsub) {
#pragma unroll
        for (int v = 0; v < RPT; ++v) {
          const float sc = DT::lo_f32(sz[sub][v]), z = DT::hi_f32(sz[sub][v]);
          u32x2 o = {DT::pack2(__builtin_fmaf(lv[v][0], sc, z), __builtin_fmaf(lv[v][1], sc, z)),
                     DT::pack2(__builtin_fmaf(lv[v][2], sc, z), __builtin_fmaf(lv[v][3], sc, z))};
          *reinterpret_cast<u32x2*>(tb + sub * (BN * 32) + v * (TRS * 32)) = o;
        }
      }
    }
  };
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 15, kq = lane >> 4;
  // byte offset of this lane's fragment of 16-row tile 0 in a stage, per k32 block; tile t is 16 rows = 2048 bytes further (the swizzle
  // term (row >> 1) & 7 does not change with 16 t): a constant in the ds_read's offset field
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
  auto mma = [&](int step) {
#pragma unroll
    for (int pl = 0; pl < KS; ++pl) {
      const char* ast = lds + L::A_OFF + (step % NST) * L::A_STAGE + pl * L::A_PLANE;
      const char* bst = lds + L::B_OFF + (step & 1) * L::B_STAGE + pl * L::B_PLANE;
      if constexpr (NT * MT <= 8) {
        u32x4 wf[2][NT], xf[2][MT];  // every fragment of the super-tile requested up front: the second half's reads land under the first half's MFMAs
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[kb][t] = *reinterpret_cast<const u32x4*>(bst + b_base[kb] + t * 2048);
#pragma unroll
          for (int t = 0; t < MT; ++t) xf[kb][t] = *reinterpret_cast<const u32x4*>(ast + a_base[kb] + t * 2048);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = DT::mfma(wf[kb][a], xf[kb][b], acc[a][b]);
      } else {                       // (64 accumulator registers: one half's fragments at a time, 128 registers per lane at 16 waves)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          u32x4 wf[NT], xf[MT];
#pragma unroll
          for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const u32x4*>(bst + b_base[kb] + t * 2048);
#pragma unroll
          for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const u32x4*>(ast + a_base[kb] + t * 2048);
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = DT::mfma(wf[a], xf[b], acc[a][b]);
        }
      }
    }
  };

  // ---- prologue: the tables of steps 0 and 1; scale / zero of steps 2 ... PW + 1 into the ring (int8: no tables) ----
  if constexpr (W8 == 0) {
    uint32_t s0[NSUB][RPT], s1[NSUB][RPT];
    load_sz(0, s0);
    load_sz(1, s1);
#pragma unroll
    for (int j = 0; j < PW; ++j) load_sz(2 + j, szr[(2 + j) % PW]);
    build_tables(0, s0);
    if (ksteps > 1 && new_group(1)) build_tables(1, s1);
  }
  tile_barrier();
  tile_barrier();
  for (int s = 0; s < ksteps; s += PW) {
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      if (s + j >= ksteps) break;
      // step t = s + j: the MFMAs of step t; the tables of step t + 2 (slot (j + 2) % PW), refilled with step t + 2 + PW
      if (!(TILE_ABL & 2)) mma(s + j);
      if constexpr (W8 == 0) {
        if (!(TILE_ABL & 8) && s + j + 2 < ksteps && new_group(s + j + 2)) build_tables(s + j + 2, szr[(j + 2) % PW]);
        load_sz(s + j + 2 + PW, szr[(j + 2) % PW]);
      }
      tile_barrier();
    }
  }

  // ---- store: lane (j = activation row fi of tile b, weight rows 4 kq ... 4 kq + 3 of tile a) ----
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int mr = m0 + wm * WMR + b * 16 + fi;
    if (mr >= p.m) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int nr = n0 + wn * WN + a * 16 + 4 * kq;
      if (nr >= p.wrows) continue;   // (wrows is a multiple of 8: a group of four rows is inside or outside)
      if (nsplit > 1) *reinterpret_cast<f32x4*>(p.part + ((int64_t)split * p.m + mr) * p.wrows + nr) = acc[a][b];
      else store_rows4<DT>(p.y, p.bias, (int64_t)mr * p.wrows + nr, nr, acc[a][b]);
    }
  }
}

// y[a][row ... row + 3] = RNE16(sum of the k-splits' f32 partial tiles, in split order) (+ bias, rounded again as everywhere)
template <typename DT>
__global__ void __launch_bounds__(256) tile_split_sum_kernel(const float* __restrict__ part, int splits, int64_t part_stride, char* __restrict__ y,
                                                             const char* __restrict__ bias, int wrows, int64_t quads) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= quads) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(part + i * 4);
  for (int s = 1; s < splits; ++s) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + s * part_stride + i * 4);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  store_rows4<DT>(y, bias, i * 4, (int)((i * 4) % wrows), acc);
}
