// w4_gemm_pair.cuh -- the persistent "pair-table" W4A16 GEMM kernel for gfx950: Bint4 weights (described first) and, with
// the template flag LA, Aint4 weights; plus its activation pre-pass w4_xprep_kernel.
//
// Same contract as w4_gemm.cuh / w4_gemm_stream.cuh (reference TinyGemmImpl.cuh:23-345 with {A,B}Layout_TC_int4,
// MatrixLayoutB.cuh:686-1101 / MatrixLayoutA.cuh, and the converters of Dequantization.cuh:55-178, 331-351), with the
// "group-scaled" numerics described below.  Variants (template flags, all in this one kernel):
//   XG   activations too large to stage whole: pre-arranged in a caller workspace, streamed per wave through a 1 KiB LDS buffer
//   LA   Aint4 weights: v_mfma_f32_16x16x32, duplicated table, activations straight from the workspace into the MFMA operand
//   LB   Bint4 weights on the same 16x16x32 structure (32-row items, up to 16 activation rows in one pass: m = 9 ... 16): one
//        packed word = one B operand, no nibble shuffling
//   QMX  mx4: e8m0 exponents fetched as 16-byte blocks per row
//   MR   1 = the m = 1 specialisation (one accumulator register per tile, group sums as running differences), 4 / 16 = general
//   NORM LlamaRMSNorm of the activations fused into the staging (tg_w4_gemm.norm_weight; own instantiations: the extra staging
//        code costs the m = 1 kernel registers it does not have)
//   x_tc / y_tc (run time): activations / output in the reference's A-fragment order instead of row-major
//   p.epilogue / p.bias_row_stride (run time): SwiGLU of gate / up row pairs, residual add in the output store
//
// Why another kernel: a per-element LDS lookup (w4_gemm_stream.cuh) costs one LDS access per 4-bit weight and the LDS
// serves 32 addresses per clock -- at the HBM roofline a CU has to dequantise ~20 weights per clock, which leaves the LDS
// 62 % busy with lookups alone (measured: 68 % LDS-busy, 61 % VALU-busy at 57 % of the HBM roofline).  A packed byte holds
// TWO codes of one weight row, so a 256-entry table of code PAIRS per row turns two lookups and a merge into one lookup.
// Such a table can only depend on the row (a per-(row, group) pair table would cost more to build than it saves), so the
// per-group affine map moves behind the contraction:
//
//   y[a][row] = sum_g ( scale[g,row] * sum_{k in g} x[a][k] * lut[row][code[row][k]]  +  zero[g,row] * sum_{k in g} x[a][k] )
//
// with every product and every sum in f32: the MFMA contracts x with the RAW LUT values (16-bit x 16-bit products are exact
// in f32), the accumulator of a group is scaled once, and sum_{k in g} x is computed once per workgroup.  Compared with the
// reference (w = RNE16(fma(lut, scale, zero)) per element, MatrixLayoutB.cuh:1042-1046) this skips the rounding of every
// dequantised weight to 16 bits: the result differs from the reference's by at most the reference's own per-weight rounding
// (|dy| <= 2^-9 sum_k |x_k w_k| worst case, ~2^-9 sqrt(sum_k x_k^2 w_k^2) typically: below one output ulp for k <= 16K)
// and is the closer of the two to the unrounded sum.  mx4 (scale = 2^e, zero = 0) loses nothing: its weights are exact.
//
//   workgroup  = 8 waves, 64 weight rows (two 32-row MFMA tiles), the whole k; wave w walks the k-slice w (split-K 8, the
//                partial sums meet in LDS and are added in wave order: deterministic).
//   MFMA       = v_mfma_f32_32x32x16: A operand = activations (act row a = lane & 31, k-slot h = lane >> 5),
//                B operand = weights (weight row c = lane & 31, same k-slot), D[a][c]: lane (c, h) holds act rows
//                (r & 3) + 8 (r >> 2) + 4 h of ITS weight row, so scale / zero are per-lane scalars.
//   table      = LDS [256 byte values][64 columns] x 4 bytes at LDS address 0; column = weight row of the workgroup, entry =
//                (lut[byte & 15], lut[byte >> 4]).  A 32-lane access group touches 32 distinct columns = 32 distinct banks:
//                conflict-free data-dependent reads, the address is ONE v_perm_b32 (byte << 8 | column << 2).
//   weights    = lane (c, h) reads the 4 I bytes (I words) of row c in k super-tile s that belong to lane-quads
//                q = 2 h, 2 h + 1 of the reference layout: the 16 lanes of 8 rows x 2 h cover one whole 256-byte (I = 4)
//                super-tile, so a wave-load touches 4 (8) fully used segments.  One word = 8 codes = one MFMA step.
//   activations= staged once per workgroup into LDS in "byte order": byte j of a packed word of quad q holds the codes of
//                k = {2q, 2q+16, 2q+1, 2q+17}[j] (low nibble) and that + 8 (high nibble) of a 32-k chunk
//                (TinyGemmConvertB.cu:252-308), so the X fragment of (chunk, q) is the 8 values
//                x[2q, 2q+8, 2q+16, 2q+24, 2q+1, 2q+9, 2q+17, 2q+25] as one 16-byte piece.
//   persistent = a workgroup walks a contiguous range of work items (item = one 64-row block of one problem of the batch):
//                the register ring of R super-tiles of packed words (+ scale|zero words) a wave keeps in flight from HBM
//                runs across item boundaries, and the LUT rows of the next item are requested at the start of the current
//                one, so the weight stream never drains while a table is rebuilt.  Two barriers per item.
//
// ABL (template parameter, always 0 in the shipped library; developer builds set it with -DTG_PAIR_ABL=<n>, dev/README.md) stubs one
// stage out so that its cost can be read off a same-box A/B: 1 no table lookups, 3 no weight loads, 4 no MFMA, 5 no activation
// reads from LDS, 6 stream only (loads, table build, staging, barriers), 7 no per-group work, 8 no scale | zero loads,
// 9 / 10 workspace variant without the activation loads / their LDS store, 11 no activation-sum staging.
#pragma once

// Element (r, c) of a matrix kept in the reference's m16n8k16 A-fragment order [ceil(rows/16)][ctiles = ceil(cols/16)][32][8]
// (TinyGemmConvertA.cu:19-141: lane t = 4 (r & 7) + (c & 7) / 2 holds (r, c0) (r, c0+1) (r+8, c0) (r+8, c0+1) and the same at
// c0 + 8): the "TC" activations / outputs of tinygemm_y_f16TC_x_f16TC_w_*TC with the weights on the right.
__device__ __forceinline__ int64_t tc_a_index(int r, int c, int ctiles) {
  const int t = (r & 7) * 4 + ((c & 7) >> 1);
  const int j = (c & 1) + 2 * ((r >> 3) & 1) + 4 * ((c >> 3) & 1);
  return (((int64_t)(r >> 4) * ctiles + (c >> 4)) * 32 + t) * 8 + j;
}
// the 32 k of chunk ch of row a (16 dwords in k order) from A-fragment-order activations
__device__ __forceinline__ void tc_a_load_chunk(const char* xb, int a, int ch, int ktiles, uint32_t (&d)[16]) {
#pragma unroll
  for (int dw = 0; dw < 16; ++dw) d[dw] = *reinterpret_cast<const uint32_t*>(xb + tc_a_index(a, ch * 32 + 2 * dw, ktiles) * 2);
}

// mx4 on gfx950 without a table: v_cvt_scalef32_pk_bf16_fp4 converts the two fp4-e2m1 codes of one byte of a packed word into a pair
// of bf16 values times an f32 scale -- the dequantised weights (fp4[code] * 2^(e - 127), exact) in ONE vector instruction per two
// weights, no LDS lookup, and with the group's scale already inside the operand no per-group accumulator update either.  Checked
// against the e2m1 table for every byte value, every byte position and scales from 2^-127 (denormal) to 2^127 and NaN (e = 255):
// tools/ubench/mx4_cvt_probe.hip.
__device__ __forceinline__ u32x4 mx4_cvt_word(uint32_t w, float scale) {
  u32x4 r;
  r[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 0));
  r[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 1));
  r[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 2));
  r[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 3));
  return r;
}

// ... one byte (sel = 0 ... 3, a constant after unrolling: the switch folds)
__device__ __forceinline__ uint32_t mx4_cvt_byte(uint32_t w, float scale, int sel) {
  switch (sel) {
    case 0: return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 0));
    case 1: return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 1));
    case 2: return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 2));
    default: return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 3));
  }
}

#ifndef TG_PAIR_M1_DOT
#define TG_PAIR_M1_DOT 1  // the m = 1 specialisation contracts with v_dot2_f32_bf16 per lane instead of the MFMA (0: MFMA, for A/B builds)
#endif
template <typename DT>
__device__ __forceinline__ float dot2_pair(uint32_t a, uint32_t b, float acc) {
  if constexpr (std::is_same<DT, BF16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), acc, false);
}

struct PairParams {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  int32_t m, wrows, k;
  int32_t ntiles;   // packed.size(0): 8-row groups
  int32_t ksuper;   // packed.size(1)
  int32_t gshift;   // log2(group)
  int32_t ngroups;  // k / group
  int32_t qtype;
  int32_t spw;      // k super-tiles per wave (a multiple of the super-tiles per group)
  int32_t nsg_shift; // log2(super-tiles per group), 0 when a group is at most one super-tile
  int32_t gch_mask; // (32-k chunks per group) - 1
  int32_t x_pitch;  // bytes per staged activation row
  int32_t lds_x;    // LDS byte offsets: staged activations (m rows, then a zero piece of one super-tile = 32 I bytes)
  int32_t lds_xs;   //                   per-group activation sums, f32 [ngroups][xs_rows]
  int32_t lds_red;  //                   split-K partial sums, f32 [8 waves][2 tiles][rused][red_lanes]
  int32_t rused;    // accumulator registers that hold real activation rows (m < 4: m, else MREGS)
  int32_t xs_rows;  // rows of the activation sums kept per group (a power of two): 4 when m <= 4 (lane half 1 then holds no real row), else 2 * MREGS
  int32_t red_lanes;  // lanes whose partial sums are exchanged: 32 when m <= 4, else 64
  int32_t red_alias;  // 1: the partial sums reuse the table's LDS (m > 4: two more barriers per item), lds_red = 0
  int32_t rblocks;  // 64-row blocks per problem
  int32_t cblocks;  // activation-row passes per problem (ceil(m / (2 MREGS)))
  int32_t items;    // rblocks * cblocks * batch
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y;
  const char* bias;   // optional [wrows] 16-bit, added after the first rounding
  int64_t stride_bias;
  int64_t bias_row_stride;  // elements between the bias rows of consecutive activation rows: 0 = one row for all, wrows = a residual
  const char* norm_w;       // fused RMSNorm of the activations in the staging (tg_w4_gemm.norm_weight), 16-bit [k]; nullptr = off
  float norm_eps;
  int32_t epilogue;         // TG_EPI_SWIGLU: rows in blocks of 8 gate + 8 up, y is [m][wrows / 2]
  int32_t dry;        // host-side only: report the kernel family instead of launching (tg_gemm_w4_plan)
  // XG variant (activation block too large to stage whole): activations pre-arranged by w4_xprep_kernel in the caller's workspace
  const char* xp;     // [problem][pass][k super-tile][row of the pass][32 I bytes in byte order]
  const char* xsum;   // f32 [problem][pass][group][xs_rows]
  int64_t stride_xp, stride_xsum;  // bytes per problem
  int32_t xw_pitch;   // bytes per activation row in a wave's LDS buffer (32 I + 16: rotates rows over the banks)
  int32_t xw_bytes;   // bytes of one wave's buffer
  int32_t chunk;      // XG: consecutive work items a workgroup takes before it moves on by (workgroups x chunk) items; >= 1
  int32_t x_tc, y_tc; // 1: activations / output in A-fragment order (tc_a_index) instead of row-major; y_tiles = ceil(wrows/16)
  int32_t y_tiles;
};

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename DT>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
  if constexpr (std::is_same<DT, BF16>::value)
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
template <typename DT>
__device__ __forceinline__ f32x4_t mfma16(u32x4 a, u32x4 b, f32x4_t c) {
  if constexpr (std::is_same<DT, BF16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {  // (mask & a) | (~mask & b)
  uint32_t d;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask), "v"(a), "v"(b));
  return d;
}

template <typename DT>
__device__ __forceinline__ float dot2_ones(uint32_t pair, float acc) {
  if constexpr (std::is_same<DT, BF16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, pair), __builtin_bit_cast(bf16x2, 0x3f803f80u), acc, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, pair), __builtin_bit_cast(f16x2, 0x3c003c00u), acc, false);
}

// sum of squares of the 32 values of a staged chunk (16 packed pairs), f32
template <typename DT>
__device__ __forceinline__ float chunk_sumsq(const uint32_t (&d)[16]) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if constexpr (std::is_same<DT, BF16>::value)
      s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, d[j]), __builtin_bit_cast(bf16x2, d[j]), s, false);
    else
      s = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, d[j]), __builtin_bit_cast(f16x2, d[j]), s, false);
  }
  return s;
}
// LlamaRMSNorm of a staged chunk: x' = RNE16(RNE16(x rs) g), g = the chunk's 32 norm weights (64 bytes at gsrc): the formula
// and rounding points of dg_add_rmsnorm (decode_glue.cuh)
template <typename DT>
__device__ __forceinline__ void chunk_rmsnorm(uint32_t (&d)[16], float rs, const u32x4 (&gw)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x4 g = gw[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t v = d[4 * j + e];
      const float lo = DT::to_f32(DT::from_f32(DT::lo_f32(v) * rs)) * DT::lo_f32(g[e]);
      const float hi = DT::to_f32(DT::from_f32(DT::hi_f32(v) * rs)) * DT::hi_f32(g[e]);
      d[4 * j + e] = DT::pack2(lo, hi);
    }
  }
}
// SwiGLU of two 16-bit GEMM outputs (dg_swiglu's formula): RNE16(RNE16(silu(g)) u)
template <typename DT>
__device__ __forceinline__ uint16_t swiglu16(float gsum, float usum) {
  const float g = DT::to_f32(DT::from_f32(gsum)), u = DT::to_f32(DT::from_f32(usum));
  return DT::from_f32(DT::to_f32(DT::from_f32(g / (1.f + __expf(-g)))) * u);
}

// I     = innerKTiles of the Bint4 layout (2, 4, 8): k super-tile = 16 I, I words per lane and super-tile
// GPS   = quantisation groups per super-tile (1 when group >= 16 I)
// MR    = 1: m = 1 (one accumulator register per tile is finalised and exchanged); else the accumulator registers that can
//         hold real activation rows: 4 -> m <= 8, 8 -> m <= 16, 16 -> m <= 32
// R     = super-tiles in flight per wave (register ring)
// XG    = activations come pre-arranged from the workspace (w4_xprep_kernel) and each wave stages its own k-slice of them, one
//         super-tile per ring slot, through a private LDS buffer (m <= 8 rows x 32 I bytes <= 1 KiB); else the workgroup stages
//         the whole activation block itself
// NSG   = super-tiles per quantisation group when that is 1 or R (group boundaries then sit at fixed places of the unrolled
//         round: no per-step tests, scale | zero words are only requested for the first super-tile of a group); 0 = any, tested at run time
// LA    = weights in the Aint4 layout (weightOnRight = false; I = 2, 4; workspace activations, one pass of m <= 16 rows).  A packed word holds
//         4 codes of row r and 4 of row r + 8 of a 16-row tile at k = 2 kq + {0, 1, 8, 9} of one 16-k tile
//         (TinyGemmConvertA.cu:172-255), lane t = 4 (r & 7) + kq.  Lane (n = lane & 15, kb = lane >> 4) of a wave takes
//         row n & 7 of 16-row tile n >> 3 at kq = kb: its word pair (two k-tiles) is the B operand of
//         v_mfma_f32_16x16x32 for the rows r (MFMA "tile" 0) and r + 8 (tile 1): 32 weight rows per workgroup.
//         The pair bytes are (code k, code k + 8): v_bfi of the word with itself shifted by one nibble.  Two lanes of a
//         32-lane LDS access group share a weight row, so the table holds every row twice (32 rows x 2 copies = the same
//         64 columns), the copy chosen by kb & 1: conflict-free.
template <typename DT, int I, int GPS, int MR, bool QMX, int R, int NSG = 0, int ABL = 0, bool XG = false, int LAY = 0, bool NORM = false>
__global__ void __launch_bounds__(512, 4) w4_gemm_pair_kernel(const PairParams p) {
  // LAY: 0 = Bint4 weights on 32x32x16 tiles (the description above), 1 = Aint4 weights (LA), 2 = Bint4 weights on 16x16x32 tiles (LB)
  constexpr bool LA = LAY == 1, LB = LAY == 2, T16 = LAY != 0;
  // m = 1 by per-lane dot products (every lane reads activation row 0's piece of ITS k-slot): a 32x32x16 MFMA spends 16384
  // multiplier slots on 512 useful products and, under the 1400 W cap, clock -- 4 v_dot2 per word instead: 76.3 -> 80.6 % on the
  // headline shape, int4 82.6 -> 84.4 %, global LUT 79.4 -> 84.0 % same-box.  Only with fixed group boundaries (NSG > 0): the
  // run-time group test around it compiles to 128 VGPRs + 200-500 bytes of scratch.
  // ABL == 100 (the one non-zero value in the shipped library: tg_w4_gemm.numerics = TG_NUM_FAST_MFMA) keeps the MFMA at m = 1
  constexpr bool DOT = TG_PAIR_M1_DOT && ABL != 100 && MR == 1 && !T16 && (QMX || NSG > 0);  // (mx4 has no group updates: any NSG)
  constexpr bool MXC = QMX;  // mx4: weights converted by v_cvt_scalef32_pk_bf16_fp4 (mx4_cvt_word), no table, no group updates
  static_assert(!NORM || (!XG && !T16 && !QMX), "fused RMSNorm: the workgroup stages the whole activation block itself");
  constexpr int WAVES = 8;
  constexpr int TILES = 2;              // MFMA tiles per workgroup (B side: 32 rows each; A side: 16 rows each, sharing their words)
  constexpr int WT = LA ? 1 : TILES;    // sets of packed words per ring slot
  constexpr int RW = T16 ? 32 : 32 * TILES;
  static_assert(!T16 || (XG && MR == 4 && (LB || I == 2 || I == 4)), "16x16x32 tiles: workspace activations, one pass of up to 16 rows");
  constexpr int NW = LB ? I / 2 : I;    // packed words per lane, word set and super-tile
  using acc_t = typename std::conditional<T16, f32x4_t, f32x16>::type;
  constexpr int CPS = I / 2;            // 32-k chunks per super-tile
  constexpr int CPG = CPS / GPS;        // chunks per group inside a super-tile (GPS > 1 only)
  constexpr int MREGS = MR == 1 ? 4 : MR;  // accumulator registers of a row set
  constexpr int RF = MR;                   // registers that are finalised per group and exchanged at the end
  constexpr int MA = T16 ? 16 : 2 * MREGS;  // activation rows a pass can hold
  // XG: 16-byte pieces per lane of one super-tile's activation block.  A side: the lane's own MFMA operand of every 32-k chunk,
  // straight from the workspace into registers (lane (row i, k-quad kb) of v_mfma_f32_16x16x32 needs exactly one piece per
  // chunk): no LDS round trip for the activations at all
  constexpr int NXW = T16 ? I / 2 : XG ? (MA * 2 * I + 63) / 64 : 1;
  static_assert(!XG || MR <= 4, "XG: one pass is at most 8 activation rows");

  // The pair table sits at LDS address 0 (a lookup address is just byte << 8 | column << 2): the kernel has no static LDS,
  // which the host verifies once per kernel (hipFuncGetAttributes().sharedSizeBytes == 0) before the first launch.
  constexpr uint32_t lds0 = 0u;
  const uint32_t lds_x = lds0 + (uint32_t)p.lds_x;
  const uint32_t lds_xs = lds0 + (uint32_t)p.lds_xs;
  const uint32_t lds_red = lds0 + (uint32_t)p.lds_red;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31;
  const int h = lane >> 5;
  const int xa = DOT ? 0 : T16 ? lane & 15 : c;       // activation row this lane supplies to the MFMA (X operand)
  const int xq = T16 ? lane >> 4 : 2 * h;   // first 16-byte piece (k-quad) of a 32-k chunk this lane reads
  const int hh = T16 ? lane >> 4 : h;       // this lane's accumulator registers r < 4 are activation rows 4 hh + r
  // weight row (inside the workgroup's block) of this lane in tile t
  auto wrow_local = [&](int t) -> int { return LA ? 16 * ((lane & 15) >> 3) + (lane & 7) + 8 * t : LB ? 16 * t + (lane & 15) : t * 32 + c; };
  const int tcol = tid & 63;  // table column this thread builds
  // ... and the weight row it belongs to (A side: column = 32 tile + 16 copy + n)
  const int tcol_row = LA ? 16 * ((tcol & 15) >> 3) + (tcol & 7) + 8 * (tcol >> 5) : LB ? 16 * (tcol >> 5) + (tcol & 15) : tcol;

  // ---- this workgroup's work items; item -> (problem b, activation pass ct, row block rb) ----
  // staged activations: a contiguous range (the block in LDS is re-staged only when the problem changes).
  // XG: items dealt round-robin, so that the workgroups running at one time read the activations of a few problems only
  // (with contiguous ranges every workgroup streams a different problem's block again and again: at m = 8, k = 4096 that is
  // 64 KiB x 64 workgroups per XCD = the whole L2, behind the weight stream -- measured as +50 % time)
  constexpr bool RR = XG;
  // XG: workgroup b takes the items [(j G + b) C, + C), j = 0, 1, ...: C = p.chunk consecutive items (one contiguous C x 128 KiB
  // of one problem's weights), then on by G C.  C = 1 is plain round-robin.
  const int chunk = RR ? p.chunk : 1;
  const int it_stride = RR ? (int)gridDim.x * chunk - (chunk - 1) : 1;  // the step from the last item of a chunk
  const int it_begin = RR ? (int)blockIdx.x * chunk : (int)(((int64_t)blockIdx.x * p.items) / gridDim.x);
  const int it_end = RR ? p.items : (int)(((int64_t)(blockIdx.x + 1) * p.items) / gridDim.x);
  const int per_problem = p.rblocks * p.cblocks;

  // ---- this wave's k-slice: super-tiles [s_begin, s_begin + nl) ----
  const int s_begin = wave * p.spw;
  const int nl = max(min(p.spw, p.ksuper - s_begin), 0);

  struct Item {
    int b, ct, rb;
  };
  auto decode = [&](int it) -> Item {
    const int b = it / per_problem, r = it - b * per_problem;
    const int ct = r / p.rblocks;
    return Item{b, ct, r - ct * p.rblocks};
  };
  // the item it_stride further: no division on the way (a wave-uniform integer division is ~25 vector instructions, and the
  // item loop needed four of them between its barriers)
  const int adv_b = it_stride / per_problem, adv_r = it_stride - adv_b * per_problem;
  auto advance = [&](const Item& e, bool big) -> Item {  // big: by it_stride, else by one item
    int b = e.b + (big ? adv_b : 0), r = e.ct * p.rblocks + e.rb + (big ? adv_r : 1);
    if (r >= per_problem) { r -= per_problem; ++b; }
    const int ct = p.cblocks == 1 ? 0 : r / p.rblocks;
    return Item{b, ct, r - ct * p.rblocks};
  };

  // ---- LUT rows: 16 values of this thread's table column as 8 packed 16-bit pairs, requested one item ahead ----
  uint32_t lp[8];
  auto lut_const = [&]() {
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
      lp[e >> 1] = DT::pack2(v0, v1);  // exact: small integers / fp4 values
    }
  };
  const bool lut_loaded = p.qtype == TG_Q_ANY4_GLOBAL || p.qtype == TG_Q_ANY4_ROWWISE;
  auto lut_request = [&](const Item& e) {  // a valid item
    const int lrow = min(e.rb * RW + tcol_row, p.wrows - 1);
    const char* lsrc = p.lut + (int64_t)e.b * p.stride_lut + (p.qtype == TG_Q_ANY4_ROWWISE ? (int64_t)lrow * 32 : 0);
    const u32x4 l0 = reinterpret_cast<const u32x4*>(lsrc)[0];
    const u32x4 l1 = reinterpret_cast<const u32x4*>(lsrc)[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { lp[j] = l0[j]; lp[4 + j] = l1[j]; }
  };

  // ---- ring of R super-tiles: packed words and scale|zero words (or mx4 exponent bytes) ----
  struct Slot {
    uint32_t w[WT][I];
    uint32_t q[TILES][GPS];
    u32x4 xw[NXW];  // XG: this lane's 16-byte pieces of the super-tile's activation block
  };
  Slot ring[R];
  // per-lane addressing of an item: byte offset of this lane's words in super-tile 0 of its rows (the host checks the matrix
  // is < 4 GiB) and its rows
  struct Rows {
    uint32_t wbase[WT];
    uint32_t qrow[TILES];
    uint32_t qrow4[TILES];  // byte offset of the row in a group's scale | zero words
    const char* wb;
    const char* qb;
    const char* xpb;  // XG: the pass's pre-arranged activations
    uint32_t xblk;    // XG: bytes of one super-tile's block (rows of the pass x 32 I)
    uint32_t xoff[NXW];  // XG: this lane's piece(s) inside a super-tile's block (A side: inside a chunk's block; lanes whose
                         // row is padding point at the block's zero row)
  };
  auto rows_of = [&](const Item& e) -> Rows {
    Rows r;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      const int row = min(e.rb * RW + wrow_local(t), p.wrows - 1);
      r.qrow[t] = (uint32_t)row;
      r.qrow4[t] = (uint32_t)row * 4u;
      if constexpr (!LA) {
        const int nt = min(row >> 3, p.ntiles - 1);
        r.wbase[t] = ((uint32_t)nt * (uint32_t)p.ksuper * 32u + (uint32_t)(4 * (row & 7) + (LB ? lane >> 4 : 2 * h))) * (uint32_t)(2 * I);
      }
    }
    if constexpr (LA) {  // [16-row tile][k super-tile][32 lanes][I words]
      const int mt = min(e.rb * 2 + ((lane & 15) >> 3), p.ntiles - 1);
      r.wbase[0] = ((uint32_t)mt * (uint32_t)p.ksuper * 32u + (uint32_t)(4 * (lane & 7) + (lane >> 4))) * (uint32_t)(4 * I);
    }
    r.wb = p.w + (int64_t)e.b * p.stride_w;
    r.qb = p.qinfo + (int64_t)e.b * p.stride_qinfo;
    if constexpr (XG) {
      const int rows = min(p.m - e.ct * MA, MA);
      r.xpb = p.xp + (int64_t)e.b * p.stride_xp + (int64_t)e.ct * MA * p.k * 2;
      r.xblk = (uint32_t)(rows * 32 * I);
      if constexpr (T16) {  // workspace [32-k chunk][k-quad][rows + 1][16 bytes]: row `rows` of every block is zero
        r.xblk = (uint32_t)((rows + 1) * 64);  // bytes of one chunk's block
        r.xoff[0] = (uint32_t)(((lane >> 4) * (rows + 1) + min(lane & 15, rows)) * 16);
      } else {
#pragma unroll
        for (int i = 0; i < NXW; ++i) r.xoff[i] = (uint32_t)(lane + 64 * i) * 16u < r.xblk ? (uint32_t)(lane + 64 * i) * 16u : 0u;
      }
    } else {
      r.xpb = nullptr;
      r.xblk = 0;
#pragma unroll
      for (int i = 0; i < NXW; ++i) r.xoff[i] = 0;
    }
    return r;
  };
  // Requests super-tile s of the rows `rw`.  `valid` is wave-uniform.  A request past the last item is still ISSUED (so that
  // the number of loads in flight is the same on every path and the compiler's vmcnt bookkeeping stays exact -- a conditional
  // refill makes it wait for every outstanding load at the next use, draining the ring each round) but every lane reads the
  // first bytes of the operand: one cached request, never consumed.
  // Every address below is (wave-uniform base that moves with s) + (per-lane offset fixed for the item): the base lives in
  // SGPRs, the loads take the saddr + VGPR-offset form and a request costs no vector ALU work.  An invalid request uses the
  // item's super-tile 0.
  // `pin` makes a per-lane offset opaque at its use: its zero-extension then stays next to the load (saddr form, 32-bit VGPR
  // offset) instead of being hoisted out of the loop as a 64-bit register pair that every load adds to its base.
  auto pin = [](uint32_t& v) -> uint32_t { asm volatile("" : "+v"(v)); return v; };
  auto issue = [&](Rows& rw, int s, Slot& sl, bool valid, bool needq) {
    // (readfirstlane keeps the moving part a scalar: otherwise loop strength reduction turns every address into a per-lane
    //  64-bit induction variable -- two vector adds and two more VGPRs per load)
    auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    const uint32_t sv = valid ? (uint32_t)s : 0u;
    if constexpr (XG) {
      // lanes beyond the block re-read its first piece (never stored): the load stays unconditional
#pragma unroll
      for (int i = 0; i < NXW; ++i) {
        if constexpr (T16) {  // chunk s CPS + i: a block of 4 k-quads x (rows + the zero row) pieces
          if constexpr (ABL == 9 || ABL == 13) sl.xw[i] = u32x4{rw.xoff[0], (uint32_t)s, 0x3f803f80u, 0x3f803f80u};  // ablation: no activation loads
          else if constexpr (ABL == 12) sl.xw[i] = *reinterpret_cast<const u32x4*>(rw.xpb + uni((uint32_t)i * rw.xblk) + pin(rw.xoff[0]));  // ablation: every chunk reads the first block (L1 hits)
          else sl.xw[i] = *reinterpret_cast<const u32x4*>(rw.xpb + uni((sv * CPS + (uint32_t)i) * rw.xblk) + pin(rw.xoff[0]));
          continue;
        }
        if constexpr (ABL == 9) sl.xw[i] = u32x4{rw.xoff[i], (uint32_t)s, 0x3f803f80u, 0x3f803f80u};  // ablation: no activation loads
        else sl.xw[i] = *reinterpret_cast<const u32x4*>(rw.xpb + uni(sv * rw.xblk) + pin(rw.xoff[i]));
      }
    }
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      if (t < WT) {  // (compile-time; the A side's two tiles share one set of words)
        const int tw = t < WT ? t : 0;
        const char* src = rw.wb + uni(sv * (uint32_t)((LA ? 128 : 64) * I)) + pin(rw.wbase[tw]);
        if constexpr (ABL == 3 || ABL == 13) {
#pragma unroll
          for (int j = 0; j < NW; ++j) sl.w[tw][j] = (uint32_t)(s * 7 + j + t);
        } else if constexpr (NW == 1) {
          sl.w[tw][0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(src));
        } else if constexpr (NW == 2) {
          const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(src));
          sl.w[tw][0] = v[0]; sl.w[tw][1] = v[1];
        } else {
#pragma unroll
          for (int v4 = 0; v4 < NW / 4; ++v4) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + v4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sl.w[tw][4 * v4 + j] = v[j];
          }
        }
      }
#pragma unroll
      for (int gg = 0; gg < GPS; ++gg) {
        if (!needq) continue;  // (compile-time per call site) not the first super-tile of its group
        const uint32_t g = (uint32_t)(((sv * CPS + gg * CPG) * 32) >> p.gshift);
        if constexpr (ABL == 8) sl.q[t][gg] = 0x3c003c00u + (uint32_t)s;  // ablation: no scale | zero loads
        else if constexpr (QMX) sl.q[t][gg] = 0u;  // mx4: the exponents come in 16-byte blocks per row, see e_request
        else sl.q[t][gg] = *reinterpret_cast<const uint32_t*>(rw.qb + uni(g * (uint32_t)p.wrows * 4u) + pin(rw.qrow4[t]));
      }
    }
  };

  // ---- mx4: e8m0 exponents [row][k / 32].  A lane owns a weight row, so a per-group byte load touches 32 different cache
  // lines per wave-instruction (measured: the kernel's bound at 41 % of the HBM roofline).  Instead every lane fetches the 16
  // exponent bytes of its row for the next 16 groups of the wave's k-slice in one load, one block ahead. ----
  constexpr int EBS = QMX ? 16 / GPS : 1;  // super-tiles per exponent block
  u32x4 ecur[TILES], enext[TILES];
  int eoff_next = 0;
  auto e_request = [&](Rows& rw, int blk, bool valid) {
    const int start = s_begin * GPS + blk * 16;          // first group of the block
    const int st = max(min(start & ~3, p.ngroups - 16), 0);  // 4-byte aligned; the last block of a row is moved back inside the row
    eoff_next = start - st;
#pragma unroll
    for (int t = 0; t < TILES; ++t)
      enext[t] = *reinterpret_cast<const u32x4*>(rw.qb + (uint32_t)__builtin_amdgcn_readfirstlane(valid ? st : 0) + rw.qrow[t] * (uint32_t)p.ngroups);
  };
  int ebase = 0;  // byte of ecur that holds group 0 of the wave's slice (block start and the shift above folded in)
  // the dword of ecur[t] that holds the exponents of slice group gi (wave-uniform); the GPS groups of one super-tile share it
  auto e_dword = [&](int t, int gi) -> uint32_t {
    const int dw = (gi + ebase) >> 2;
    // a wave-uniform dynamic element of the register vector: one relative move (nested selects on a uniform condition compile to
    // a tree of scalar branches inside the main loop)
    return ecur[t][dw & 3];
  };
  // 2^(e - 127) as f32 bits-wise: e << 23, e = 0 -> 2^-127 (a denormal), e = 255 -> NaN (Dequantization.cuh:331-339)
  auto e_scale = [&](uint32_t d, int gi) -> float {
    const uint32_t e23 = __builtin_amdgcn_ubfe(d, (uint32_t)(((gi + ebase) & 3) * 8), 8u) << 23;
    const float sc = u2f(e23 > 0x00400000u ? e23 : 0x00400000u);
    return __builtin_fmaf(sc, 0.f, sc);  // inf (e = 255) -> NaN, everything else unchanged
  };

  // ---- activation staging: chunk (row a, 32 k) -> LDS in byte order, and the per-group sums ----
  const int nch = p.k >> 5;
  auto x_load = [&](const char* xb, int a0, int xi, uint32_t (&d)[16]) {  // xb = the problem's activations, a0 = the pass's first row
    const int a = xi / nch, ch = xi - a * nch;
    if (p.x_tc) {
      tc_a_load_chunk(xb, a0 + a, ch, p.k >> 4, d);
      return;
    }
    const u32x4* src = reinterpret_cast<const u32x4*>(xb + ((int64_t)(a0 + a) * p.k + ch * 32) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = src[j];
      d[4 * j] = v[0]; d[4 * j + 1] = v[1]; d[4 * j + 2] = v[2]; d[4 * j + 3] = v[3];
    }
  };
  auto x_store = [&](int xi, bool on, const uint32_t (&d)[16]) {  // d holds zeros when !on
    const int a = on ? xi / nch : 0, ch = on ? xi - (xi / nch) * nch : 0;
    if (on) {
      const uint32_t dst = lds_x + (uint32_t)(a * p.x_pitch + ch * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u32x4 o;
        o[0] = __builtin_amdgcn_perm(d[q + 4], d[q], 0x05040100u);       // x[2q]     x[2q+8]
        o[1] = __builtin_amdgcn_perm(d[q + 12], d[q + 8], 0x05040100u);  // x[2q+16]  x[2q+24]
        o[2] = __builtin_amdgcn_perm(d[q + 4], d[q], 0x07060302u);       // x[2q+1]   x[2q+9]
        o[3] = __builtin_amdgcn_perm(d[q + 12], d[q + 8], 0x07060302u);  // x[2q+17]  x[2q+25]
        *(lds_u32x4ptr)(dst + (uint32_t)(q * 16)) = o;
      }
    }
    if constexpr (QMX) return;  // mx4 has no zero point: no activation sums (the host plans no LDS for them)
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum = dot2_ones<DT>(d[j], sum);
    // chunks of one group sit in adjacent lanes (k / 32 is a multiple of the chunks per group): fixed-shape butterfly
    for (int o = 1; o <= p.gch_mask; o <<= 1) sum += __shfl_xor(sum, o);
    if (on && (ch & p.gch_mask) == 0) *(lds_fptr)(lds_xs + (uint32_t)(((ch >> (p.gshift - 5)) * p.xs_rows + a) * 4)) = sum;
  };
  // stages activation rows [a0, a0 + mrows) of problem b; `pre` = the first batch of chunks is already in xd
  // With p.norm_w the rows pass through LlamaRMSNorm on the way (tg_w4_gemm.norm_weight): sum of squares per chunk, then over the
  // 64 chunks of a wave (one row: the host guarantees k % 2048 == 0), one partial per wave in the (not yet written) activation-sum
  // area, added per row in wave order (deterministic); then scale, multiply by the norm weights, store.
  auto x_stage = [&](int b, int a0, int mrows, bool pre, uint32_t (&xd)[16]) {
    const char* xb = p.x + (int64_t)b * p.stride_x;
    const int xtotal = mrows * nch;
    if constexpr (NORM) {  // (xtotal <= 512: a block that is staged whole is at most ~14 KiB = 224 chunks; the host checks)
      const bool on = tid < xtotal;
      if (!pre) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xd[j] = 0u;
        if (on) x_load(xb, a0, tid, xd);
      }
      // (the chunk's norm weights are requested before the reduction: one L2 round trip less on the launch's critical path)
      const int a = on ? tid / nch : 0;
      u32x4 gw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) gw[j] = reinterpret_cast<const u32x4*>(p.norm_w + (on ? tid - a * nch : 0) * 64)[j];
      float ss = chunk_sumsq<DT>(xd);
      ss = tgl::wave_sum(ss);
      if (lane == 0) *(lds_fptr)(lds_xs + (uint32_t)(wave * 4)) = ss;
      __syncthreads();
      ss = 0.f;
      for (int w0 = a * (nch >> 6); w0 < (a + 1) * (nch >> 6); ++w0) ss += *(lds_fptr)(lds_xs + (uint32_t)(w0 * 4));
      __syncthreads();  // every thread has its row's sum before x_store's group sums land in the same area
      if (on) chunk_rmsnorm<DT>(xd, rsqrtf(ss * (1.0f / (float)p.k) + p.norm_eps), gw);
      pre = true;
    }
    for (int it0 = 0; it0 < xtotal; it0 += 512) {
      const int xi = it0 + tid;
      const bool on = xi < xtotal;
      if (!(pre && it0 == 0)) {
#pragma unroll
        for (int j = 0; j < 16; ++j) xd[j] = 0u;
        if (on) x_load(xb, a0, xi, xd);
      }
      x_store(xi, on, xd);
    }
    // rows a >= mrows of the sums stay zero
    if constexpr (!QMX)
      for (int idx = tid; idx < p.ngroups * p.xs_rows; idx += 512)
        if ((idx & (p.xs_rows - 1)) >= mrows) *(lds_fptr)(lds_xs + (uint32_t)(idx * 4)) = 0.f;
    // the zero piece behind the staged rows is one super-tile long: lanes whose A-operand row is padding read it with the same
    // immediate offsets as the real rows
    if (tid < CPS * 4) *(lds_u32x4ptr)(lds_x + (uint32_t)(mrows * p.x_pitch + tid * 16)) = u32x4{0, 0, 0, 0};
  };

  // XG: the pass's per-group sums (computed by w4_xprep_kernel) -> LDS, rows >= mrows zero, and the zero piece
  // Items are dealt round-robin, so every item stages another layer's sums: one L2 round trip at the item's start.  They are
  // requested BEFORE the next item's LUT rows (vector memory returns in order: behind those cold HBM loads the sums would wait a
  // DRAM latency per item; same-box A/B +0.5 % at 4096^2, +1.5 % at 8192^2).  On the A side (registers to spare) the first
  // 2 x 512 sums are requested one item ahead into registers (xs_request), like the LUT rows: +0.5-1.7 %; on the B side that
  // costs 7 more spilled registers and 5 %.
  constexpr int NXS = T16 ? 2 : 0;
  float xsn[2] = {0.f, 0.f};
  auto xs_request = [&](const Item& e) {  // a valid item
    if constexpr (XG && !QMX) {
      const float* src = reinterpret_cast<const float*>(p.xsum + (int64_t)e.b * p.stride_xsum) + (int64_t)e.ct * p.ngroups * p.xs_rows;
      const int total = p.ngroups * p.xs_rows;
#pragma unroll
      for (int j = 0; j < NXS; ++j) xsn[j] = src[min(tid + 512 * j, total - 1)];
    }
  };
  auto xs_stage = [&](int b, int ct, int mrows) {
    const float* src = reinterpret_cast<const float*>(p.xsum + (int64_t)b * p.stride_xsum) + (int64_t)ct * p.ngroups * p.xs_rows;
    if constexpr (!QMX) {
      const int total = p.ngroups * p.xs_rows;
#pragma unroll
      for (int j = 0; j < NXS; ++j) {
        const int idx = tid + 512 * j;
        if (idx < total) *(lds_fptr)(lds_xs + (uint32_t)(idx * 4)) = (idx & (p.xs_rows - 1)) < mrows ? xsn[j] : 0.f;
      }
      for (int idx = tid + 512 * NXS; idx < total; idx += 512)
        *(lds_fptr)(lds_xs + (uint32_t)(idx * 4)) = (idx & (p.xs_rows - 1)) < mrows ? src[idx] : 0.f;
    }
    if (tid < CPS * 4) *(lds_u32x4ptr)(lds_x + (uint32_t)(WAVES * p.xw_bytes + tid * 16)) = u32x4{0, 0, 0, 0};
  };

  // ---- requests before the first item, in the order the prologue consumes them (vector memory returns in order): its LUT
  // rows, the first batch of its activation chunks, then its first R super-tiles (one by one: the scheduler must not reorder
  // them, the ring is consumed in slot order) ----
  if (it_begin >= it_end) return;
  const Item first = decode(it_begin);
  if (lut_loaded) lut_request(first);
  else lut_const();
  xs_request(first);
  int staged_b = XG ? -1 : first.b, staged_ct = first.ct;  // which activation block the LDS holds (XG: staged by the first item)
  uint32_t xd0[16];
  if constexpr (!XG) {
    const int mrows0 = min(p.m - first.ct * MA, MA);
#pragma unroll
    for (int j = 0; j < 16; ++j) xd0[j] = 0u;
    if (tid < mrows0 * nch) x_load(p.x + (int64_t)first.b * p.stride_x, first.ct * MA, tid, xd0);
  }
  Rows rcur = rows_of(first);
  if constexpr (QMX) e_request(rcur, 0, nl > 0);
#pragma unroll
  for (int j = 0; j < R; ++j) {
    __builtin_amdgcn_sched_barrier(0);
    issue(rcur, s_begin + j, ring[j], j < nl, NSG == 0 || j % NSG == 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!XG) x_stage(first.b, first.ct * MA, min(p.m - first.ct * MA, MA), true, xd0);

  uint32_t colreg[TILES];
#pragma unroll
  for (int t = 0; t < TILES; ++t) colreg[t] = (uint32_t)((T16 ? 32 * t + 16 * ((lane >> 4) & 1) + (lane & 15) : t * 32 + c) * 4);

  int table_b = -1;  // the problem whose LUT the table in LDS was built from
  Item inext = first;
  int cpos = 0;  // position inside the chunk
  for (int it = it_begin, step = 1; it < it_end; it += step) {
    const bool big = cpos + 1 >= chunk;
    step = big ? it_stride : 1;
    cpos = big ? 0 : cpos + 1;
    const Item cur = inext;
    const int row0 = cur.rb * RW;
    const int a0 = cur.ct * MA;
    const int mrows = min(p.m - a0, MA);
    const bool has_next = it + step < it_end;
    if (has_next) inext = advance(cur, big);  // (the last item asks for its own rows again)
    Rows rnext = rows_of(inext);

    // ---- pair table of this item: thread = (column, high nibbles 2 wave and 2 wave + 1).  The previous item's lookups are
    // all behind the barrier that ended it.  Only a per-row LUT changes from item to item: int4 / mx4 / one global LUT keep the
    // first item's table (unless the split-K partial sums reuse its LDS) -- 32 table stores per thread and item less. ----
    if (!MXC && (it == it_begin || p.qtype == TG_Q_ANY4_ROWWISE || p.red_alias ||
        (p.qtype == TG_Q_ANY4_GLOBAL && cur.b != table_b))) {  // (a global LUT is one per PROBLEM of the batch)
      table_b = cur.b;
      uint32_t hw = lp[0];
#pragma unroll
      for (int j = 1; j < 8; ++j) hw = (wave == j) ? lp[j] : hw;
      const uint32_t base = lds0 + (uint32_t)(wave * 2 * 16 * 256 + tcol * 4);
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const uint32_t e0 = __builtin_amdgcn_perm(hw, lp[a >> 1], (a & 1) ? 0x05040302u : 0x05040100u);
        const uint32_t e1 = __builtin_amdgcn_perm(hw, lp[a >> 1], (a & 1) ? 0x07060302u : 0x07060100u);
        // (indexing ONE LDS pointer: constant offsets go into the instructions' offset fields and the pair becomes a
        //  ds_write2st64_b32; integer address arithmetic cast to a pointer per store cost a v_or_b32 + ds_write_b32 each)
        const lds_u32ptr tb = (lds_u32ptr)base;
        tb[a * 64] = e0;
        tb[(16 + a) * 64] = e1;
      }
    }
    // ---- activations: only when the activation block changes (staged: the first item's block was staged above; XG: the sums
    // of this item were requested one item ago) ----
    if (cur.b != staged_b || cur.ct != staged_ct) {
      staged_b = cur.b;
      staged_ct = cur.ct;
      if constexpr (XG) {
        if (ABL != 11) xs_stage(cur.b, cur.ct, mrows);
      } else {
        uint32_t xd[16];
        x_stage(cur.b, a0, mrows, false, xd);
      }
    }
    // the next item's LUT rows (and activation sums) travel while this item is computed (the last item re-reads its own)
    if (lut_loaded) lut_request(inext);
    xs_request(inext);
    __syncthreads();  // table and activations visible (and every thread is done with the previous item's partial sums)

    // ---- main loop of the item ----
    const bool a_on = xa < mrows;  // this lane's A-operand row is a real activation row; the others read the zero piece
    const uint32_t xzero = XG ? lds_x + (uint32_t)(WAVES * p.xw_bytes) : lds_x + (uint32_t)(mrows * p.x_pitch);
    const uint32_t xs_mask = 4 * hh < p.xs_rows ? 0xffffffffu : 0u;
    const uint32_t xs_lane = 4 * hh < p.xs_rows ? lds_xs + (uint32_t)(4 * hh * 4) : xzero;
    const uint32_t xwbuf = lds_x + (uint32_t)(wave * p.xw_bytes);  // XG: this wave's activation buffer (one super-tile)
    const uint32_t xrow = !a_on ? xzero : XG ? xwbuf + (uint32_t)(xa * p.xw_pitch + xq * 16) : lds_x + (uint32_t)(xa * p.x_pitch + xq * 16);
    const uint32_t xmask = a_on && !XG ? 0xffffffffu : 0u;  // lanes on the zero piece (and XG lanes: one buffer) never move
    // XG: where this lane's 16 bytes of a super-tile's block go (piece = lane; 2 I pieces per activation row)
    const uint32_t xw_dst = xwbuf + (uint32_t)((lane / (2 * I)) * p.xw_pitch + (lane % (2 * I)) * 16);

    acc_t acc[TILES];
    float yacc[TILES][RF];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int r = 0; r < (T16 ? 4 : 16); ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int r = 0; r < RF; ++r) yacc[t][r] = 0.f;
    }
    float gs[TILES], gz[TILES];  // scale / zero of the current group
    float xsv[RF];               // activation sums of the current group for this lane's accumulator rows
#pragma unroll
    for (int t = 0; t < TILES; ++t) gs[t] = gz[t] = 0.f;
#pragma unroll
    for (int r = 0; r < RF; ++r) xsv[r] = 0.f;

    auto unpack_q = [&](uint32_t q, float& s, float& z) {
      if constexpr (QMX) {
        s = u2f(q == 255u ? 0x7fc00000u : (q == 0u ? 0x00400000u : (q << 23)));  // Dequantization.cuh:331-339
        z = 0.f;
      } else {
        s = DT::lo_f32(q);
        z = DOT && h ? 0.f : DT::hi_f32(q);  // (DOT: both lane halves of a row accumulate, the zero-point term only once)
      }
    };
    // m = 1: the accumulator tuples run through the whole slice (cleared per item) and a group's sum is taken as a difference;
    // otherwise a group's first MFMA takes a zero C operand
    constexpr bool DIFF = MR == 1;
    float dacc[TILES] = {0.f, 0.f};  // DOT: this lane's running sum over its k-slot
    float prev[TILES][RF];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int r = 0; r < RF; ++r) prev[t][r] = 0.f;
    // a finished group: y += scale * acc + zero * sum(x).  The accumulator tuple is only ever written by MFMAs -- a group's
    // first MFMA takes a zero C operand instead of the tuple being cleared element-wise (partial writes of the 16-register
    // tuple made the compiler copy it around).  The finalize of a group runs BEHIND the lookups of the next step (`pending`),
    // so the MFMA results are ready when it reads them.
    auto finalize = [&]() {
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        if constexpr (DOT) {
          const float d = dacc[t] - prev[t][0];
          prev[t][0] = dacc[t];
          yacc[t][0] = __builtin_fmaf(gz[t], xsv[0], __builtin_fmaf(gs[t], d, yacc[t][0]));
          continue;
        }
        if constexpr (DIFF && RF == 1 && !QMX) {
          // m = 1: four single-register instructions per tile, spelled out.  Left to the compiler the two tiles' updates become
          // v_pk_* on register PAIRS; at the 128-VGPR budget the only aligned pair it finds overlaps an accumulator tuple, which
          // it then moves out of the way and back (17 v_mov_b64 per group and wave).
          float a0 = acc[t][0];
          asm volatile("" : "+v"(a0));  // a copy made by the compiler (it knows the MFMA -> VALU read hazard), not by the asm below
          float d;
          asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a0), "v"(prev[t][0]));
          prev[t][0] = a0;
          asm("v_fmac_f32 %0, %1, %2" : "+v"(yacc[t][0]) : "v"(gs[t]), "v"(d));
          asm("v_fmac_f32 %0, %1, %2" : "+v"(yacc[t][0]) : "v"(gz[t]), "v"(xsv[0]));
          asm volatile("" : "+v"(acc[t]));
          asm volatile("" ::"v"(acc[t][1]), "v"(acc[t][2]), "v"(acc[t][3]));
          continue;
        }
#pragma unroll
        for (int r = 0; r < RF; ++r) {
          float d = acc[t][r];
          if constexpr (DIFF) {  // running accumulator: this group's sum is what was added since the last group ended
            d -= prev[t][r];
            prev[t][r] = acc[t][r];
          }
          yacc[t][r] = __builtin_fmaf(gs[t], d, yacc[t][r]);
          if constexpr (!QMX) yacc[t][r] = __builtin_fmaf(gz[t], xsv[r], yacc[t][r]);  // mx4 has no zero point
        }
        // keep the accumulator one opaque 16-register value: when only element 0 is read (m = 1) the compiler's
        // sub-register liveness otherwise scatters the MFMA chain over several overlapping tuples and spills.  (Not with
        // zero-C group starts: there the pin made the compiler copy the finished tuple, 16 v_mov_b64 per group.)
        if constexpr (DIFF) asm volatile("" : "+v"(acc[t]));
        if constexpr (RF == 1) asm volatile("" ::"v"(acc[t][1]), "v"(acc[t][2]), "v"(acc[t][3]));  // same purpose: as live as with m > 1
      }
    };
    bool pending = false;  // wave-uniform
    acc_t zero16;
#pragma unroll
    for (int r = 0; r < (T16 ? 4 : 16); ++r) zero16[r] = 0.f;
    // one super-tile: 2 CPS MFMA steps, step = (chunk jc, quad pair qq) for both tiles: 8 table lookups + one X piece
    uint32_t edw[TILES] = {0u, 0u};  // mx4: the exponent dword of the current super-tile
    auto consume = [&](int s, const Slot& sl, int j_slot) {
      const uint32_t xst = xrow + ((uint32_t)(s * CPS * 64) & xmask);  // this lane's X pieces of the super-tile
      if constexpr (XG && !T16) {
        // the wave's own DS operations execute in order: the reads of the previous super-tile are behind us, the reads below
        // follow this store; no barrier
        // every lane stores (the buffer has a row for each: rows past the pass's are never read): a store under a lane mask
        // is control flow, and around it the compiler copied the accumulator tuples
#pragma unroll
        for (int i = 0; i < NXW; ++i)
          if constexpr (ABL != 10) *(lds_u32x4ptr)(xw_dst + (uint32_t)(i * (64 / (2 * I)) * p.xw_pitch)) = sl.xw[i];
      }
      if constexpr (ABL == 6) {  // ablation: stream only
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
          for (int j = 0; j < I; ++j) acc[t][0] += u2f(sl.w[t][j]);
        acc[0][1] += u2f(sl.q[0][0] ^ sl.q[1][0]);
        return;
      }
#pragma unroll
      // B side: a step = half a 32-k chunk (quads 2 h + qq); A side: a whole chunk (one word pair = two 16-k tiles)
      for (int u = 0; u < (T16 ? CPS : 2 * CPS); ++u) {
        const int jc = T16 ? u : u >> 1, qq = T16 ? 0 : u & 1;
        const bool qfirst = T16 || qq == 0, qlast = T16 || qq == 1;
        const int chunk = s * CPS + jc;
        // group boundaries: static when a super-tile holds several groups, else a wave-uniform runtime test
        constexpr bool STATIC_G = GPS > 1 || NSG > 0;
        const int gpos = NSG > 0 ? j_slot % NSG : 0;  // (compile-time) position of this super-tile in its group
        const bool gfirst = qfirst && (GPS > 1 ? jc % CPG == 0 : NSG > 0 ? (gpos == 0 && jc == 0) : (chunk & p.gch_mask) == 0);
        const bool glast = qlast && (GPS > 1 ? jc % CPG == CPG - 1 : NSG > 0 ? (gpos == NSG - 1 && jc == CPS - 1) : (chunk & p.gch_mask) == p.gch_mask);
        u32x4 xf;
        u32x4 bf[TILES];
        if constexpr (ABL == 5) xf = u32x4{xrow, (uint32_t)s, (uint32_t)jc, (uint32_t)qq};  // ablation: no X reads
        else if constexpr (T16) xf = sl.xw[jc];
        else xf = *(lds_cu32x4ptr)(xst + (uint32_t)(jc * 64 + 16 * qq));
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          if constexpr (MXC) {
            // the step's group scale first (a group is one 32-k chunk), then one conversion per pair byte
            if (gfirst) {
              const int gg0 = GPS == 1 ? 0 : jc / CPG;
              const int gi = (s - s_begin) * GPS + gg0;
              if (gg0 == 0) edw[t] = e_dword(t, gi);
              gs[t] = e_scale(edw[t], gi);
            }
          }
          if constexpr (LA) {
            // tile 0 = the low nibbles (rows r), tile 1 = the high nibbles (rows r + 8) of the chunk's two words; pair byte =
            // (code k, code k + 8): bytes 0 / 2 of the low-nibble word, 1 / 3 of the high-nibble word; operand order
            // (k, k+8) (k+16, k+24) (k+1, k+9) (k+17, k+25) = the activation pieces' order
            uint32_t uw[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const uint32_t w = sl.w[0][2 * jc + e];
              // (mask & w) | (~mask & shifted w): one v_bfi_b32 (the compiler's own choice for the C expression is three ops)
              uw[e] = t == 0 ? bfi(0x0f0f0f0fu, w, w >> 4) : bfi(0xf0f0f0f0u, w, w << 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if constexpr (MXC) {
                bf[t][j] = mx4_cvt_byte(uw[j & 1], gs[t], t + 2 * (j >> 1));
                continue;
              }
              const uint32_t addr = __builtin_amdgcn_perm(uw[j & 1], colreg[t], 0x0c0c0400u + ((uint32_t)(t + 2 * (j >> 1)) << 8));
              if constexpr (ABL == 1) bf[t][j] = addr;
              else bf[t][j] = *(lds_cu32ptr)(addr);
            }
          } else if constexpr (LB) {
            // one packed word = the 8 codes of this lane's row at k = 2 q + {0, 16, 1, 17, 8, 24, 9, 25} of the chunk = one B
            // operand of v_mfma_f32_16x16x32, in the activation pieces' order (as in w4_gemm_pair16.cuh)
            const uint32_t w = sl.w[t][jc];
            if constexpr (MXC) {
              bf[t] = mx4_cvt_word(w, gs[t]);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const uint32_t addr = __builtin_amdgcn_perm(w, colreg[t], 0x0c0c0400u + ((uint32_t)j << 8));
                if constexpr (ABL == 1) bf[t][j] = addr;
                else bf[t][j] = *(lds_cu32ptr)(addr);
              }
            }
          } else if constexpr (MXC) {
            bf[t] = mx4_cvt_word(sl.w[t][qq * CPS + jc], gs[t]);
          } else {
            const uint32_t w = sl.w[t < WT ? t : 0][qq * CPS + jc];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t addr = __builtin_amdgcn_perm(w, colreg[t], 0x0c0c0400u + ((uint32_t)j << 8));
              if constexpr (ABL == 1) bf[t][j] = addr;  // ablation: no lookups
              else bf[t][j] = *(lds_cu32ptr)(addr);
            }
          }
        }
        // the previous step ended a group: its finalize runs here, behind this step's lookups (MFMA results ready, no stall)
        if constexpr (MXC) {
          // (nothing per group: the scale is inside the operands, the accumulators run through the whole slice)
        } else if constexpr (STATIC_G) {
          // fixed boundaries: the step that starts a group finalises the previous one (before the first group of an item
          // the accumulators and scales are zero: it adds nothing)
          if (gfirst) finalize();
        } else if (qfirst && pending) {
          finalize();
          pending = false;
        }
        // the finished group's accumulators must be dead before the next group's first MFMA: if the scheduler sinks the
        // finalize below it, the two groups need two accumulator tuples (32 VGPRs more)
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 7 && gfirst && !MXC) {  // a group starts: its scale | zero and (not mx4) its activation sums
          const int gg = GPS == 1 ? 0 : jc / CPG;
#pragma unroll
          for (int t = 0; t < TILES; ++t) {
            if constexpr (QMX) {
              const int gi = (s - s_begin) * GPS + gg;
              if (gg == 0) edw[t] = e_dword(t, gi);
              gs[t] = e_scale(edw[t], gi);
            } else {
              unpack_q(sl.q[t][gg], gs[t], gz[t]);
            }
          }
          if constexpr (!QMX) {
            // lanes whose accumulator rows are all padding (lane half 1 when m <= 4) read the zero piece behind the staged rows
            // (branch-free: a data-dependent branch in this loop makes the compiler copy accumulators around it)
            const uint32_t xsa = xs_lane + ((uint32_t)((((chunk * 32) >> p.gshift) * p.xs_rows) * 4) & xs_mask);
            if constexpr (RF == 1) {
              // m = 1: every lane reads row 0's sum (a wave-uniform address: one v_mov); the lanes of half 1 hold activation
              // row 4 in this register, which is never stored
              xsv[0] = *(const __attribute__((address_space(3))) float*)(lds_xs + (uint32_t)((((chunk * 32) >> p.gshift) * p.xs_rows) * 4));
            } else {
#pragma unroll
              for (int r4 = 0; r4 < MREGS / 4; ++r4) {
                const f32x4 vv = *(lds_cf32x4ptr)(xsa + ((uint32_t)(r4 * 32) & xs_mask));
                xsv[4 * r4] = vv[0]; xsv[4 * r4 + 1] = vv[1]; xsv[4 * r4 + 2] = vv[2]; xsv[4 * r4 + 3] = vv[3];
              }
            }
          }
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          if constexpr (DOT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dacc[t] = dot2_pair<DT>(bf[t][j], xf[j], dacc[t]);
          } else
          if constexpr (ABL == 4) acc[t][0] += u2f(bf[t][0] ^ bf[t][1] ^ bf[t][2] ^ bf[t][3] ^ xf[0] ^ xf[1] ^ xf[2] ^ xf[3]);  // ablation: no MFMA
          else if constexpr (T16) acc[t] = mfma16<DT>(xf, bf[t], (ABL != 7 && gfirst && !MXC) ? zero16 : acc[t]);
          else if (ABL != 7 && !DIFF && gfirst && !MXC) acc[t] = mfma32<DT>(xf, bf[t], zero16);
          else acc[t] = mfma32<DT>(xf, bf[t], acc[t]);
        }
        if constexpr (STATIC_G && !DOT) {
#pragma unroll
          for (int t = 0; t < TILES; ++t) asm volatile("" : "+v"(acc[t]));  // see finalize()
        }
        if (ABL != 7 && !STATIC_G && glast && !MXC) pending = true;
        // keep the scheduler from hoisting the lookups of later steps above this point: it would trade the 4-waves-per-SIMD
        // register budget for instruction-level parallelism the other waves already provide
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // Rounds of R slice positions; position l lives in ring[l % R].  All rounds but the last refill from this item; the
    // last round (always executed, also by waves with an empty slice) refills slot j with position j of the NEXT item, so
    // the weight stream never drains -- and the compiler knows that those R refills were issued after the LUT request
    // above, so the next table build waits with vmcnt(loads of one round) instead of draining the ring.
    const int rounds = max((nl + R - 1) / R, 1);
    int l0 = 0;
    // mx4: the slice is walked in blocks of EBS super-tiles (16 exponent bytes per row); at the start of a block the next
    // block's exponents are requested -- for the last block, the first block of the next item
    const int nblk = QMX ? max((nl + EBS - 1) / EBS, 1) : 1;
    for (int blk = 0; blk < nblk; ++blk) {
      const bool lastb = blk + 1 == nblk;
      if constexpr (QMX) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) ecur[t] = enext[t];
        ebase = eoff_next - blk * 16;
        e_request(lastb ? rnext : rcur, lastb ? 0 : blk + 1, lastb ? has_next && nl > 0 : true);
      }
      const int rend = lastb ? rounds - 1 : (blk + 1) * (EBS / R);  // rounds before the peeled one / of this block
      for (int rd = l0 / R; rd < rend; ++rd, l0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          consume(s_begin + l0 + j, ring[j], j);
          issue(rcur, s_begin + l0 + j + R, ring[j], l0 + j + R < nl, NSG == 0 || j % NSG == 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (l0 + j < nl) consume(s_begin + l0 + j, ring[j], j);
      issue(rnext, s_begin + j, ring[j], has_next && j < nl, NSG == 0 || j % NSG == 0);
    }
    if constexpr (MXC) {  // the slice's sums as they are
#pragma unroll
      for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < RF; ++r) yacc[t][r] = DOT ? dacc[t] : acc[t][r];
    } else if (pending || GPS > 1 || NSG > 0) finalize();  // the last group of the slice
    if constexpr (DOT) {  // the two k-slots of a row live in lanes c and c + 32: lane half 0 gets the sum
#pragma unroll
      for (int t = 0; t < TILES; ++t) yacc[t][0] += __shfl_xor(yacc[t][0], 32);
    }
    if constexpr (ABL == 6) yacc[0][0] += acc[0][0] + acc[0][1] + acc[1][0];
    if constexpr (ABL == 7) { yacc[0][0] = acc[0][0]; yacc[1][0] = acc[1][0]; }

    // ---- split-K tail: the partial sums of the 8 waves meet in LDS and are added in wave order ----
    if (p.red_alias) __syncthreads();  // the partial sums overwrite the table: every wave must be done with its lookups
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int r = 0; r < RF; ++r)
        if (r < p.rused && lane < p.red_lanes)
          *(lds_fptr)(lds_red + (uint32_t)((((wave * TILES + t) * p.rused + r) * p.red_lanes + lane) * 4)) = yacc[t][r];
    }
    __syncthreads();  // partial sums visible; every wave is done with this item's table
    {
      char* yb = p.y + (int64_t)cur.b * p.stride_y;
      static_assert(TILES == 2, "the output index split below assumes two tiles");
      const int rl_shift = p.red_lanes == 64 ? 6 : 5;  // red_lanes is 32 or 64
      for (int o = tid; o < TILES * p.rused * p.red_lanes; o += 512) {
        // (no integer divisions: three of them per output were ~50 vector instructions on the one wave the workgroup's next
        //  barrier then waits for)
        const int l = o & (p.red_lanes - 1), q = o >> rl_shift;
        const int t = q >= p.rused ? 1 : 0, r = q - t * p.rused;
        const int a = T16 ? r + 4 * (l >> 4) : (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const int row = row0 + (LA ? 16 * ((l & 15) >> 3) + (l & 7) + 8 * t : LB ? 16 * t + (l & 15) : t * 32 + (l & 31));
        if (a < mrows && row < p.wrows) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < WAVES; ++w) sum += *(lds_fptr)(lds_red + (uint32_t)((((w * TILES + t) * p.rused + r) * p.red_lanes + l) * 4));
          if (!T16 && p.epilogue == TG_EPI_SWIGLU) {
            // rows come in blocks of 8 gate + 8 up: lane l holds gate row `row`, lane l + 8 (same tile, same half) its up row
            if ((l & 15) < 8) {
              float up = 0.f;
#pragma unroll
              for (int w = 0; w < WAVES; ++w) up += *(lds_fptr)(lds_red + (uint32_t)((((w * TILES + t) * p.rused + r) * p.red_lanes + l + 8) * 4));
              *reinterpret_cast<uint16_t*>(yb + ((int64_t)(a0 + a) * (p.wrows >> 1) + ((row >> 4) << 3) + (row & 7)) * 2) = swiglu16<DT>(sum, up);
            }
            continue;
          }
          uint16_t o16 = DT::from_f32(sum);
          if (p.bias)  // rounded sum + bias, rounded again: bit-identical to the reference module's separate `y + bias`
            o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(*reinterpret_cast<const uint16_t*>(p.bias + (int64_t)cur.b * p.stride_bias + ((int64_t)(a0 + a) * p.bias_row_stride + row) * 2)));
          *reinterpret_cast<uint16_t*>(yb + (p.y_tc ? tc_a_index(a0 + a, row, p.y_tiles) : (int64_t)(a0 + a) * p.wrows + row) * 2) = o16;
        }
      }
    }
    if (p.red_alias) __syncthreads();  // ... and the next table must not overwrite partial sums that are still being read
    rcur = rnext;
  }
}

// ---- XG pre-pass: activations -> workspace in the order the pair kernel's waves consume them, plus the per-group sums ----
// xp   [problem][pass][k super-tile][row of the pass][32 I bytes]: the 64 bytes of a 32-k chunk in "byte order" (see above)
// xsum f32 [problem][pass][group][xs_rows]
// thread = one (activation row, 32-k chunk); the sums are formed exactly as the staging path of the kernel forms them
// (16 two-element dot products in k order, then a butterfly over the chunks of the group).
struct XPrepParams {
  const char* x;
  char* xp;
  char* xsum;
  int32_t m, k, ma, cps, gshift, gch_mask, ngroups, xs_rows;
  int32_t la;  // 1: A-side order [32-k chunk][k-quad][rows + 1]: the last row of every block is zero
  int32_t x_tc;  // 1: the activations come in A-fragment order (tc_a_index)
  int64_t stride_x, stride_xp, stride_xsum;
};

template <typename DT>
__global__ void __launch_bounds__(256) w4_xprep_kernel(const XPrepParams p) {
  const int nch = p.k >> 5;
  const int xi = blockIdx.x * 256 + threadIdx.x;
  const bool on = xi < p.m * nch;
  const int a = on ? xi / nch : 0, ch = on ? xi - a * nch : 0;
  const int b = blockIdx.y;
  uint32_t d[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) d[j] = 0u;
  if (on && p.x_tc) {
    tc_a_load_chunk(p.x + (int64_t)b * p.stride_x, a, ch, p.k >> 4, d);
  } else if (on) {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.x + (int64_t)b * p.stride_x + ((int64_t)a * p.k + ch * 32) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x4 v = src[j];
      d[4 * j] = v[0]; d[4 * j + 1] = v[1]; d[4 * j + 2] = v[2]; d[4 * j + 3] = v[3];
    }
  }
  const int ct = a / p.ma, ar = a - ct * p.ma;
  const int rows = min(p.m - ct * p.ma, p.ma);
  if (on) {
    const int s = ch / p.cps, jc = ch - s * p.cps;
    char* dst = p.xp + (int64_t)b * p.stride_xp + (int64_t)ct * p.ma * p.k * 2 +
                (p.la ? ((int64_t)ch * 4 * (rows + 1) + ar) * 16 : ((int64_t)(s * rows + ar) * p.cps + jc) * 64);
    const int qstride = p.la ? rows + 1 : 1;  // in 16-byte pieces
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32x4 o;
      o[0] = __builtin_amdgcn_perm(d[q + 4], d[q], 0x05040100u);
      o[1] = __builtin_amdgcn_perm(d[q + 12], d[q + 8], 0x05040100u);
      o[2] = __builtin_amdgcn_perm(d[q + 4], d[q], 0x07060302u);
      o[3] = __builtin_amdgcn_perm(d[q + 12], d[q + 8], 0x07060302u);
      reinterpret_cast<u32x4*>(dst)[q * qstride] = o;
    }
    if (p.la && ar == 0)  // the zero row of this chunk's blocks
#pragma unroll
      for (int q = 0; q < 4; ++q) reinterpret_cast<u32x4*>(dst)[q * qstride + rows] = u32x4{0, 0, 0, 0};
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) sum = dot2_ones<DT>(d[j], sum);
  for (int o = 1; o <= p.gch_mask; o <<= 1) sum += __shfl_xor(sum, o);
  if (on && (ch & p.gch_mask) == 0)
    reinterpret_cast<float*>(p.xsum + (int64_t)b * p.stride_xsum)[((int64_t)ct * p.ngroups + (ch >> (p.gshift - 5))) * p.xs_rows + ar] = sum;
}
