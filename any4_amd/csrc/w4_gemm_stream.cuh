// w4_gemm_stream.cuh -- the streaming W4A16 GEMM kernel for gfx950 ("lane owns group").
//
// Same contract and numerics as w4_gemm.cuh (the split-K latency kernel); this one is used whenever a
// launch has enough 16-row tiles to give every wave its own tile or a large k-slice of one, and the
// quantisation group is at least one unit (Bint4: g >= 128, Aint4: g >= 64).
//
// Idea: the dequantised weight  w = RNE16( fma_f32(lut[row][code], scale[g,row], zero[g,row]) )
// (reference MatrixLayoutB.cuh:1042-1046 / MatrixLayoutA.cuh:747-754) takes only 16 distinct
// values per (row, quantisation group).  If ONE lane walks a contiguous k-span of ONE row, it can
// compute those 16 final 16-bit values once per group -- bit-identical to the per-element fma --
// park them in its private LDS column, and every weight element then costs one v_perm_b32 (the LDS
// address) and one LDS read; two reads are merged into an MFMA operand register by one v_or_b32.
// No per-element fma / cvt, no cross-lane traffic.
//
// Mapping: MFMA 16x16x32, W = A operand.  Lane (i = lane & 15, Q = lane >> 4) owns row i of the
// tile and the Q-th quarter of the wave's k-slice, walked in units of 64 packed bytes (B: 128 k,
// A: 64 k).  The K-slot of lane-row Q in every MFMA is "the next 8 k of quarter Q"; the X fragment
// of lane (c, Q) is the matching 16 contiguous bytes of activation row c.
//
// Activations go through LDS so that X never sits in the in-order vector-memory return queue (the
// only VM waits in the loop are for data requested one whole unit earlier).  Two staging modes:
//   PRIVX = false: every wave of a workgroup walks the same k sequence, so the X slab of one
//                  unit-step ([quarters][slices][act rows][UNIT]) is staged once per workgroup,
//                  double buffered, one barrier per unit;
//   PRIVX = true : (m = 1) every wave stages its own slab (one 16-byte load per lane per unit): no barrier in
//                  the main loop; the waves of a workgroup are the k-slices of one tile (split-K 1..8);
//   XRES  = true : (m >= 2, split-K 1, m * k * 2 <= ~96 KiB) the whole activation block is staged ONCE per
//                  workgroup ([quarters][act rows][k / 4], p.xslab_bytes = bytes per row) and stays resident
//                  while the 16 waves walk p.tiles_per_wave tiles each: one barrier per workgroup instead of one per unit.
//
// LDS (dynamic, sized by the host; no static LDS, checked on the host, so the base is 0):
//   [WAVES x 4 KiB lookup tables][X slabs][split-K partial tiles, only when splitk > 1].
// Wave w's table starts at byte w * 4096.  With WAVES == 1 (the stacked m = 1 launches) the table sits at 0
// and a lookup address is just (nibble << 8 | lane << 2); otherwise address bits 12..15 (the table
// select) are OR-ed into the nibble bytes before the v_perm_b32.
#pragma once
#ifndef TG_STREAM_UNCOND
#define TG_STREAM_UNCOND 0
#endif


struct StreamParams {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  int32_t m, wrows, k;
  int32_t ntiles;     // packed.size(0)
  int32_t ksuper;     // packed.size(1)
  int32_t gshift;     // log2(group)
  int32_t ngroups;    // k / group
  int32_t qtype;
  int32_t splitk, sk_shift;
  int32_t rowtiles;   // ceil(wrows / 16)
  int32_t units_per_lane;   // NU: units walked by every lane (a multiple of group / UNIT)
  int32_t upg_mask;         // (units per quantisation group) - 1
  int32_t xslab_bytes;      // bytes of one staged X slab
  int32_t red_off;          // LDS byte offset of the split-K partial tiles (unused when splitk == 1)
  int32_t tiles_per_wave;   // resident-X launches: consecutive tile groups walked by one workgroup (else 1)
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y;
  const char* bias;   // optional [wrows] 16-bit, added after the first rounding (see store_rows4)
  int64_t stride_bias;
  int64_t bias_row_stride;  // elements between the bias rows of consecutive activation rows (0: one row for all)
  int32_t dry;        // host-side only: report the kernel family instead of launching (tg_gemm_w4_plan)
};

// WPL = packed words per (k super-tile, lane-row) entry: Bint4: I/2, Aint4: I
// XL  = 16-byte X pieces staged per thread and unit (host picks the smallest that covers the slab)
// LK  = 1 (Bint4): the lookups of one MFMA step as ONE block of hand-written LDS instructions (do_chunk) in which the X fragment
//       is read under an EXEC mask of the lanes whose MFMA column is a real activation row (m = 1: 4 of 64 lanes; the others keep
//       the zeros their fragment registers were initialised with): a 16-byte LDS read of all 64 lanes costs eight LDS cycles,
//       a third of this kernel's LDS time at m = 1 (SQ_LDS_IDX_ACTIVE: 2.8 cycles per wave and weight, of which 2 are the lookups).
template <typename DT, bool LAYOUT_A, int WPL, bool QMX, int WAVES, int MINW, int XL, bool PRIVX, int ABL = 0, bool XRES = false, int LK = 0>
__global__ void __launch_bounds__(WAVES * 64, MINW) w4_gemm_stream_kernel(const StreamParams p) {
  static_assert(LK == 0 || !LAYOUT_A, "the hand-written lookup block exists for Bint4 weights");
  constexpr int CHUNK = LAYOUT_A ? 16 : 32;  // k per chunk (one packed word per q)
  constexpr int UNIT = 4 * CHUNK;            // k per unit = 64 packed bytes per lane
  constexpr int NMMA = LAYOUT_A ? 2 : 4;     // MFMAs per chunk
  constexpr int NP = 4 / WPL;                // 16*WPL-byte pieces per unit
  constexpr int XROW = UNIT * 2 + 16;        // bytes per staged X row; +16 rotates rows over the LDS banks
  constexpr int PPR = UNIT * 2 / 16;         // 16-byte pieces per staged X row
  constexpr int NSTAGE = PRIVX ? 64 : WAVES * 64;  // threads sharing one slab

  // LDS is addressed from offset 0: the kernel has no static LDS, which the host verifies before the first launch
  // (prepare_lds_kernel: hipFuncGetAttributes().sharedSizeBytes == 0).
  constexpr uint32_t lds_x0 = WAVES * 4096u;
  const uint32_t lds_red = (uint32_t)p.red_off;  // split-K tiles live behind the X slabs (only allocated when splitk > 1)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15;
  const int Q = lane >> 4;
  const int r = i & 7;

  const int slice = wave & (p.splitk - 1);
  const int ct = blockIdx.y;
  const int64_t b = blockIdx.z;
  const char* xb = p.x + b * p.stride_x;
  const char* wb = p.w + b * p.stride_w;
  const char* qb = p.qinfo + b * p.stride_qinfo;
  const char* lb = p.lut ? p.lut + b * p.stride_lut : nullptr;
  char* yb = p.y + b * p.stride_y;

  const uint32_t wks = 32u * WPL * 4u;  // bytes per k super-tile
  const int nunits = p.k / UNIT + (p.k % UNIT ? 1 : 0);
  const int NU = p.units_per_lane;
  const int u_first = (slice * 4 + Q) * NU;  // this lane's first unit

  // ---- X staging: thread -> (staged row, 16-byte piece); staged row = (quarter, slice, act row c) ----
  const int mrows = min(p.m - ct * 16, 16);
  // shared slab: rows of every k-slice of the workgroup; private slab: only this wave's slice
  const int xrows = (PRIVX || XRES) ? mrows * 4 : mrows * 4 * p.splitk;
  const uint32_t lds_x = PRIVX ? lds_x0 + (uint32_t)(wave * 2 * p.xslab_bytes) : lds_x0;
  uint32_t xs_rowbase[XL];  // global byte offset of the activation row this thread stages from
  uint32_t xs_in[XL];       // byte offset inside that row at unit-step 0
  uint32_t xs_loff[XL];     // LDS byte offset inside one slab
  bool xs_on[XL];
#pragma unroll
  for (int j = 0; j < XL; ++j) {
    const int pid = (PRIVX ? lane : tid) + j * NSTAGE;
    const int srow = pid / PPR, pc = pid % PPR;
    xs_on[j] = srow < xrows;
    // staged row order [quarter][slice][act row]: the 16 lanes (c = 0..15) of one lane-row read CONSECUTIVE slab rows,
    // whose +16-byte rotation spreads them over the LDS banks (measured: halves SQ_LDS_BANK_CONFLICT of the Aint4
    // shared-slab launches against the [act row][quarter] order)
    // (private slabs exist for m = 1 only, where both orders coincide: keep the shift-only form there)
    const int sc = PRIVX ? srow >> 2 : srow % mrows, st_ = PRIVX ? 0 : srow / mrows;
    const int ssl = PRIVX ? slice : st_ & (p.splitk - 1);
    const int sq = PRIVX ? srow & 3 : st_ >> p.sk_shift;
    const int xr = min(ct * 16 + sc, p.m - 1);
    xs_rowbase[j] = (uint32_t)(xr * p.k * 2);
    xs_in[j] = (uint32_t)((((ssl * 4 + sq) * NU) * UNIT + pc * 8) * 2);
    xs_loff[j] = (uint32_t)(srow * XROW + pc * 16);
  }
  const uint32_t xrow_last = (uint32_t)((p.k - 8) * 2);  // clamp: last valid 16-byte piece of a row
  auto stage_load = [&](int u, u32x4 (&R)[XL]) {
#pragma unroll
    for (int j = 0; j < XL; ++j) {
      if (xs_on[j]) {
        // clamp inside the activation row (padding units read valid memory; their weights are zero)
        const uint32_t inrow = min(xs_in[j] + (uint32_t)(u * UNIT * 2), xrow_last);
        R[j] = *reinterpret_cast<const u32x4*>(xb + (xs_rowbase[j] + inrow));
      }
    }
  };
  auto stage_store = [&](int buf, const u32x4 (&R)[XL]) {
#pragma unroll
    for (int j = 0; j < XL; ++j)
      if (xs_on[j]) *(lds_u32x4ptr)(lds_x + (uint32_t)(buf * p.xslab_bytes) + xs_loff[j]) = R[j];
  };
  // this lane's fragment row in a staged slab.  MFMA columns >= m are never stored; they read the all-zero row
  // behind the staged rows (zero operands keep the multipliers of the unused columns quiet -> less power, more clock).
  const int frow = i >= mrows ? xrows : (PRIVX ? i * 4 + Q : (XRES ? Q * mrows + i : ((Q << p.sk_shift) + slice) * mrows + i));
  const uint32_t xfrag = lds_x + (uint32_t)(frow * (XRES ? p.xslab_bytes : XROW));
  if constexpr (XRES) {
    // resident X: stage every (act row, quarter) k-span once; the zero row (one unit long) sits behind them
    const int ppr = NU * (UNIT / 8);  // 16-byte pieces per staged row
    for (int pid = tid; pid < xrows * ppr; pid += WAVES * 64) {
      const int srow = pid / ppr, pc = pid - srow * ppr;  // srow = quarter * mrows + act row
      const int xr = min(ct * 16 + srow % mrows, p.m - 1);
      const uint32_t inrow = min((uint32_t)((((srow / mrows) * NU) * UNIT + pc * 8) * 2), xrow_last);
      *(lds_u32x4ptr)(lds_x + (uint32_t)(srow * p.xslab_bytes + pc * 16)) =
          *reinterpret_cast<const u32x4*>(xb + ((uint32_t)(xr * p.k * 2) + inrow));
    }
    if (tid < XROW / 16) *(lds_u32x4ptr)(lds_x + (uint32_t)(xrows * p.xslab_bytes + tid * 16)) = u32x4{0, 0, 0, 0};
    __syncthreads();
  } else {
    const int zt = PRIVX ? lane : tid;
    if (zt < 2 * (XROW / 16))
      *(lds_u32x4ptr)(lds_x + (uint32_t)((zt / (XROW / 16)) * p.xslab_bytes + xrows * XROW + (zt % (XROW / 16)) * 16)) = u32x4{0, 0, 0, 0};
  }

  // ---- tile loop: a resident-X workgroup walks p.tiles_per_wave consecutive groups of WAVES tiles (the X block
  // stays in LDS, waves do not wait for each other between tiles); every other mode handles one tile per wave ----
  const int tiles_per_wave = XRES ? p.tiles_per_wave : 1;
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
  // One 16-row tile per workgroup: the 128-byte line of scale | zero words of a group covers 32 rows = the tiles 2 j and 2 j + 1, and
  // workgroup b runs on XCD b % 8 (observed; used for speed only) -- neighbouring tiles on neighbouring workgroups fetched every
  // such line from HBM TWICE, once per XCD's L2 (round 4 counters: HBM read 1.064 x the algorithmic bytes = the scale-zero bytes
  // again).  Tile pairs therefore go to the workgroups b and b + 8: the same XCD, dispatched together.
  int bx = (int)blockIdx.x;
  if ((WAVES >> p.sk_shift) == 1 && tiles_per_wave == 1 && (gridDim.x & 15u) == 0u) bx = (((bx >> 4) * 8 + (bx & 7)) << 1) + ((bx >> 3) & 1);
  const int rt = (bx * tiles_per_wave + tt) * (WAVES >> p.sk_shift) + (wave >> p.sk_shift);
  const bool rt_ok = rt < p.rowtiles;
  const int row0 = rt * 16;
  const int row = row0 + i;
  const int row_c = min(row, p.wrows - 1);
  const int tile = LAYOUT_A ? rt : 2 * rt + (i >> 3);
  const bool lane_ok = rt_ok && row < p.wrows && tile < p.ntiles;
  const int tile_c = min(tile, p.ntiles - 1);
  // packed-weight addressing: piece pc of unit U lives at k super-tile NP * U + pc
  const uint32_t wrow = (uint32_t)((tile_c * p.ksuper * 32 + 4 * r) * WPL * 4);

  // ---- raw LUT of this lane's row: 16 x 16 bit, kept packed in 8 registers ----
  u32x4 lut0, lut1;
  if (p.qtype == TG_Q_INT4 || p.qtype == TG_Q_MX4) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      float v0, v1;
      if (p.qtype == TG_Q_INT4) {
        v0 = (float)(e - 8);
        v1 = (float)(e - 7);
      } else {
        const int e1 = e + 1;
        v0 = (e & 8 ? -1.f : 1.f) * ((e & 7) < 5 ? 0.5f * (e & 7) : ((e & 7) == 5 ? 3.f : (e & 7) == 6 ? 4.f : 6.f));
        v1 = (e1 & 8 ? -1.f : 1.f) * ((e1 & 7) < 5 ? 0.5f * (e1 & 7) : ((e1 & 7) == 5 ? 3.f : (e1 & 7) == 6 ? 4.f : 6.f));
      }
      const uint32_t pr = DT::pack2(v0, v1);  // exact: small integers / fp4 values
      if (e < 8) lut0[e >> 1] = pr;
      else lut1[(e - 8) >> 1] = pr;
    }
  } else {
    const char* lrow = lb + (p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)row_c * 32 : 0);
    lut0 = reinterpret_cast<const u32x4*>(lrow)[0];
    lut1 = reinterpret_cast<const u32x4*>(lrow)[1];
  }

  const uint32_t lane4 = (uint32_t)lane * 4u;
  const uint32_t sh = LAYOUT_A ? (uint32_t)(i >> 3) * 4u : 0u;

  // scale|zero word of the group containing the first k of unit U
  auto load_q = [&](int U) -> uint32_t {
    const int g = min(U * UNIT, p.k - 1) >> p.gshift;
    if constexpr (QMX) return reinterpret_cast<const uint8_t*>(qb)[(uint32_t)(row_c * p.ngroups + g)];
    else return reinterpret_cast<const uint32_t*>(qb)[(uint32_t)(g * p.wrows + row_c)];
  };

  auto load_unit = [&](int U, u32x4 (&L)[4]) {
    const int Uc = min(U, nunits - 1);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int ks = min(NP * Uc + pc, p.ksuper - 1);
      const char* src = wb + (wrow + (uint32_t)ks * wks);
#pragma unroll
      for (int v = 0; v < WPL; ++v) {
        if constexpr (ABL == 3) L[pc * WPL + v] = u32x4{(uint32_t)ks, 1u, 2u, (uint32_t)v};  // ablation: no weight stream
        // Aint4: lanes i and i + 8 read the same words (rows m0 / m0 + 8 share a word), the second read must hit the
        // cache -- a non-temporal load would go to HBM twice (measured: 2.65 -> 2.11 us per 4096^2 layer)
        else if constexpr (LAYOUT_A) L[pc * WPL + v] = *reinterpret_cast<const u32x4*>(src + 16 * v);
        else L[pc * WPL + v] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + 16 * v));
      }
    }
  };

  auto word = [&](const u32x4 (&L)[4], int q, int c) -> uint32_t {
    if constexpr (WPL == 1) return L[c][q];
    else if constexpr (WPL == 2) return L[2 * (c >> 1) + (q >> 1)][2 * (q & 1) + (c & 1)];
    else return L[q][c];
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // LK: the X fragment registers (zero in the lanes whose MFMA column is padding, for good) and the lanes that read them
  u32x4 xz = {0u, 0u, 0u, 0u};
  const uint64_t xexec = (uint64_t)((1u << mrows) - 1u) * 0x0001000100010001ull;

  constexpr bool ONE = WAVES == 1;  // single-wave workgroup: table at LDS offset 0, no table-select bits
  const uint32_t tabbase = ONE ? 0u : (uint32_t)wave * 4096u;
  const uint32_t kmask = ONE ? 0u : __builtin_amdgcn_readfirstlane(((tabbase >> 12) & 0xfu) * 0x10101010u);
  const uint32_t tabcol = tabbase + lane4;
  {
    // 16 final 16-bit weights of (row, group) -> this lane's LDS column; slot = value << 16
    auto build_table = [&](uint32_t q, bool ok) {
      float s, z;
      if constexpr (QMX) {
        s = u2f(q == 255u ? 0x7fc00000u : (q == 0u ? 0x00400000u : (q << 23)));  // Dequantization.cuh:331-339
        z = 0.f;
      } else {
        s = DT::lo_f32(q);
        z = DT::hi_f32(q);
      }
      if (!ok) s = z = 0.f;  // padding lanes contribute exact zeros
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const uint32_t raw = e < 8 ? lut0[e >> 1] : lut1[(e - 8) >> 1];
        // two scalar v_fma_f32 (full rate); left to itself the compiler SLP-vectorises the pair into v_pk_fma_f32 plus
        // operand moves, which measured 2-3 % slower
        float f0, f1;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f0) : "v"(DT::lo_f32(raw)), "v"(s), "v"(z));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(f1) : "v"(DT::hi_f32(raw)), "v"(s), "v"(z));
        const uint32_t pr = DT::pack2(f0, f1);
        *(lds_u32ptr)(tabcol + (uint32_t)e * 256u) = pr << 16;
        *(lds_u32ptr)(tabcol + (uint32_t)(e + 1) * 256u) = pr & 0xffff0000u;
      }
    };

    // two lookups -> one operand register.  A table slot is the dword (value << 16): a 32-bit read
    // yields the value already in the HIGH half, a 16-bit read at slot + 2 yields it in the LOW half,
    // so the merge is one VOP2 v_or_b32.  Address = (table select | nibble) << 8 | lane << 2.
    auto look2 = [&](uint32_t src, int byte_lo, int byte_hi) -> uint32_t {
      const uint32_t a0 = __builtin_amdgcn_perm(src, lane4, 0x0c0c0400u + ((uint32_t)byte_lo << 8));
      const uint32_t a1 = __builtin_amdgcn_perm(src, lane4, 0x0c0c0400u + ((uint32_t)byte_hi << 8));
      if constexpr (ABL == 1) return a0 ^ a1;  // ablation: no LDS lookups
      const uint32_t lo = *(lds_cu16ptr)(a0 + 2u);
      const uint32_t hi = *(lds_cu32ptr)(a1);
      return lo | hi;
    };

    // chunk c of the unit whose words are in L; X fragments come from the staged slab `xbuf`
    auto do_chunk = [&](const u32x4 (&L)[4], int c, uint32_t xbuf) {
      const uint32_t xa = xbuf + (uint32_t)(c * CHUNK * 2);
      if constexpr (ABL == 4) {  // ablation: stream only
        acc[0] += u2f(L[c][0] ^ L[c][1] ^ L[c][2] ^ L[c][3] ^ (*(lds_cu32ptr)(xa)));
        return;
      }
      if constexpr (!LAYOUT_A) {
        uint32_t wa[4], wb4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t w = word(L, q, c);
          wa[q] = ONE ? (w & 0x0f0f0f0fu) : ((w & 0x0f0f0f0fu) | kmask);                 // bytes: v0 v4 v1 v5
          wb4[q] = ONE ? ((w >> 4) & 0x0f0f0f0fu) : (((w >> 4) & 0x0f0f0f0fu) | kmask);  // bytes: v2 v6 v3 v7
        }
        if constexpr (LK) {
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            uint32_t al[4], ah[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t src = (h & 1) ? wb4[q] : wa[q];
              al[q] = __builtin_amdgcn_perm(src, lane4, 0x0c0c0400u + ((uint32_t)(h >> 1) << 8));
              ah[q] = __builtin_amdgcn_perm(src, lane4, 0x0c0c0400u + ((uint32_t)((h >> 1) + 2) << 8));
            }
            uint32_t a0, a1, a2, a3, t0, t1, t2, t3;
            uint64_t sv;
            // (LDS returns in order and the block ends with lgkmcnt(0): nothing it issued is outstanding when the compiler's own
            //  bookkeeping resumes; the data of a read arrives >= 64 cycles after its issue, long after the previous MFMA has
            //  read the operand registers it overwrites.  The d16 load forms would make the merges unnecessary, but with SRAM ECC
            //  enabled (this part) they clear the other half of the register instead of preserving it.)
            asm volatile(
                "ds_read_u16 %0, %10 offset:2\n\t"
                "ds_read_b32 %4, %11\n\t"
                "ds_read_u16 %1, %12 offset:2\n\t"
                "ds_read_b32 %5, %13\n\t"
                "ds_read_u16 %2, %14 offset:2\n\t"
                "ds_read_b32 %6, %15\n\t"
                "ds_read_u16 %3, %16 offset:2\n\t"
                "ds_read_b32 %7, %17\n\t"
                "s_mov_b64 %9, exec\n\t"
                "s_mov_b64 exec, %19\n\t"
                "ds_read_b128 %8, %18\n\t"
                "s_mov_b64 exec, %9\n\t"
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_or_b32 %0, %0, %4\n\t"
                "v_or_b32 %1, %1, %5\n\t"
                "v_or_b32 %2, %2, %6\n\t"
                "v_or_b32 %3, %3, %7"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(xz), "=&s"(sv)
                : "v"(al[0]), "v"(ah[0]), "v"(al[1]), "v"(ah[1]), "v"(al[2]), "v"(ah[2]), "v"(al[3]), "v"(ah[3]), "v"(xa + 16u * h), "s"(xexec)
                : "memory");
            acc = DT::mfma(u32x4{a0, a1, a2, a3}, xz, acc);
          }
        } else
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          u32x4 a;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t src = (h & 1) ? wb4[q] : wa[q];
            a[q] = look2(src, h >> 1, (h >> 1) + 2);  // k = 8h + 2q, 8h + 2q + 1
          }
          acc = DT::mfma(a, *(lds_cu32x4ptr)(xa + 16u * h), acc);
        }
      } else {
        uint32_t ws[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ws[q] = ONE ? ((word(L, q, c) >> sh) & 0x0f0f0f0fu) : (((word(L, q, c) >> sh) & 0x0f0f0f0fu) | kmask);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          u32x4 a;
#pragma unroll
          for (int q = 0; q < 4; ++q) a[q] = look2(ws[q], h, h + 2);
          acc = DT::mfma(a, *(lds_cu32x4ptr)(xa + 16u * h), acc);
        }
      }
    };

    // ---- prologue: unit 0 words + scale, X slab 0 staged, X slab 1 in registers ----
    u32x4 L0[4], L1[4];   // packed words: unit being consumed / unit in flight (ping-pong)
    uint32_t q0, q1 = 0;  // scale|zero word of the same two units
    u32x4 XR[XL];         // X pieces of the unit after next, on their way to LDS
    load_unit(u_first, L0);
    q0 = load_q(u_first);
    if constexpr (!XRES) {
      stage_load(0, XR);
      stage_store(0, XR);
      if (NU > 1) stage_load(1, XR);
      if constexpr (!PRIVX) __syncthreads();
    }

    // All control flow below is workgroup-uniform (NU, splitk, group size), barriers included.
    auto do_unit = [&](int u, const u32x4 (&Lc)[4], uint32_t qc, u32x4 (&Ln)[4], uint32_t& qn) {
      const int U = u_first + u;
#if TG_STREAM_UNCOND
      // (the next unit's words and scale are requested unconditionally -- load_unit / load_q clamp their addresses, the unit past the end is
      //  never consumed: with the requests under the branch hipcc cannot count them and waits vmcnt(0) in front of build_table below,
      //  i.e. for the requests it has just issued)
      load_unit(U + 1, Ln);
      qn = load_q(U + 1);
#endif
      if (u + 1 < NU) {
#if !TG_STREAM_UNCOND
        load_unit(U + 1, Ln);
        qn = load_q(U + 1);
#endif
        if constexpr (!XRES) {
          stage_store((u + 1) & 1, XR);           // slab u+1 (requested one unit ago) -> LDS
          if (u + 2 < NU) stage_load(u + 2, XR);  // request slab u+2
        }
      }
      if ((u & p.upg_mask) == 0) build_table(qc, lane_ok && U * UNIT < p.k);  // a quantisation group starts here
      // resident X: unit u of this lane's k-span (the zero row is only one unit long)
      const uint32_t xbuf = XRES ? xfrag + (i < mrows ? (uint32_t)(u * UNIT * 2) : 0u) : xfrag + (uint32_t)((u & 1) * p.xslab_bytes);
#pragma unroll
      for (int c = 0; c < 4; ++c) do_chunk(Lc, c, xbuf);
      // shared slab: slab u+1 visible to everyone, everyone done with slab u.  Private slab: a wave's
      // DS operations execute in order, nothing to wait for.
      if constexpr (!PRIVX && !XRES) __syncthreads();
    };
    for (int u = 0; u < NU; u += 2) {
      do_unit(u, L0, q0, L1, q1);
      if (u + 1 < NU) do_unit(u + 1, L1, q1, L0, q0);
    }
  }

  // ---- split-K tail (as in w4_gemm.cuh) ----
  if (p.splitk > 1) {
    *(lds_f32x4ptr)(lds_red + (uint32_t)((wave * 64 + lane) * 16)) = acc;
    __syncthreads();
    if (slice != 0) return;
    for (int o = 1; o < p.splitk; ++o) acc += *(lds_cf32x4ptr)(lds_red + (uint32_t)(((wave + o) * 64 + lane) * 16));
  }
  const int col = ct * 16 + i;
  const int rowg = row0 + 4 * Q;
  if (rt_ok && col < p.m && rowg < p.wrows) {
    store_rows4<DT>(yb, p.bias ? p.bias + b * p.stride_bias + (int64_t)col * p.bias_row_stride * 2 : nullptr, (int64_t)col * p.wrows + rowg, rowg, acc);
  }
  }  // tile loop
}
