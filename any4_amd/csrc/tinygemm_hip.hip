// tinygemm_hip.hip -- hand-written gfx950 (MI355X / CDNA4) kernels + the C ABI of
// include/tinygemm_hip.h.  Written for wave64 + v_mfma_f32_16x16x32_{bf16,f16}; there is no
// CUDA path, no hipify output and no multi-backend dispatch in this file.
//
// Reference behaviour being replaced (facebookresearch/any4 @ 2025-07-18, file:line):
//   tinygemm_lib/TinyGemmImpl.cuh:23-345      the split-K tile kernel
//   tinygemm_lib/MatrixLayoutA.cuh:375-816    weights-as-A int4 load + dequant
//   tinygemm_lib/MatrixLayoutB.cuh:686-1101   weights-as-B int4 load + dequant
//   tinygemm_lib/Dequantization.cuh:17-178, 331-351
//   tinygemm_lib/TinyGemm_int4.cu:294-548     host validation / dispatch
//   tinygemm_lib/TinyGemm_bf16.cu:163-327     16-bit weights
//   tinygemm_lib/TinyGemmConvert{A,B}.cu      layout / packing kernels
//   tinygemm_lib/TinyGemmDequantize.cu:19-58  debug dequant op
//
// Design notes live in DESIGN.md; the short version of the GEMM kernel:
//   * one workgroup = one 16-row weight tile, split-K over its waves (step = one 16-byte
//     packed-weight load per lane = 1 KiB per wave, streamed with non-temporal loads);
//   * W is the MFMA A operand (16 weight rows x 32 k), X the B operand (32 k x 16 activation
//     rows).  The reference's packed words are consumed AS STORED: a 2x2 / 4x4 word transpose
//     between the four 16-lane rows of the wave (v_permlane16_swap / v_permlane32_swap) gives
//     every lane a contiguous k-chunk, so the X fragment is one contiguous 16-byte load;
//   * the 16-entry LUT lives in LDS as f32, one private bank column per lane
//     ([16 entries][64 lanes]) so the 8 data-dependent lookups per word never conflict;
//     the lookup address is built with one v_perm_b32 per nibble;
//   * dequant = v_fma_f32(lut, scale, zero) then v_cvt_pk_bf16_f32 (RNE): the f32 product of
//     two 16-bit floats is exact, so this equals the reference's single-rounding bf16 fma;
//   * fp32 partial tiles meet in LDS, a fixed-order sum gives a deterministic result.

#include <stddef.h>
#include "tg_common.cuh"

namespace {

#include "w8_gemm.cuh"

// ---- 16-bit weights (reference TinyGemm_bf16.cu) ---------------------------------------------
// Same tile/split-K structure; the A operand is gathered dword-wise from the fragment-order
// tensor (each dword = two adjacent k of one row), no dequantisation.
struct F16GemmParams {
  const char* x;
  const char* w;
  char* y;
  int32_t m, wrows, k;
  int32_t ktiles_padded;  // k-tiles present in the TC tensor (size(1) * I)
  int32_t inner;          // I
};

template <typename DT, bool LAYOUT_A, int WAVES, int I>
__global__ void __launch_bounds__(WAVES * 64) f16_gemm_kernel(const F16GemmParams p) {
  __shared__ f32x4 s_red[WAVES * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, Q = lane >> 4, r = i & 7;
  const int rt = blockIdx.x, ct = blockIdx.y;
  const int row0 = rt * 16, row = row0 + i;
  const bool row_ok = row < p.wrows;
  // Lane (i, Q) reads the fragment words of ITS OWN lane slot t = 4 (i & 7) + Q of the m16n8k16 layouts, as stored:
  // per k-tile the dwords (k0,k1) and (k0+8,k0+9) with k0 = 2Q, so one K-slot (two k-tiles) is the 8 k values
  // {2Q, 2Q+1, 2Q+8, 2Q+9} + {0, 16} and the X fragment is four dwords at byte offsets 4Q + {0, 16, 32, 48} of the slot
  // (the mapping of w8_gemm.cuh).  One 16-byte (A16, B16 I = 2) or 8-byte (B16 I = 1) load per k-tile pair / k-tile.
  const uint32_t* wd = reinterpret_cast<const uint32_t*>(p.w);
  const int xrow = min(ct * 16 + i, p.m - 1);
  const bool xcol = ct * 16 + i < p.m;
  const int ktiles = p.k >> 4;  // k % 32 == 0
  const int nsteps_total = ktiles >> 1;
  const int t = 4 * r + Q;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // A ring of F16_RING steps in flight per wave.  Every load is unconditional per lane with clamped addresses and nothing of a step is touched
  // before it is consumed (round 6: a load under a lane mask makes hipcc wait vmcnt(0) right behind it -- every step then exposed the whole
  // memory latency); rows / columns beyond the problem are masked at the consumer.
  constexpr int F16_RING = 4;
  struct Step { u32x4 w0, w1, x; };
  const int rt_c = rt;  // (row tiles are never out of range: the grid is exact; rows beyond wrows within the last tile are masked below)
  auto load_step = [&](int s, Step& st) {
    const int sc = min(s, nsteps_total - 1);
    if constexpr (LAYOUT_A) {
      // [mT][kT][32][8 halfs] = 4 dwords per lane slot: (m0;k0,k1) (m1;k0,k1) (m0;k8,k9) (m1;k8,k9)
      st.w0 = *reinterpret_cast<const u32x4*>(wd + (((int64_t)rt_c * p.ktiles_padded + 2 * sc) * 32 + t) * 4);
      st.w1 = *reinterpret_cast<const u32x4*>(wd + (((int64_t)rt_c * p.ktiles_padded + 2 * sc + 1) * 32 + t) * 4);
    } else {
      // [nT][kT/I][32][4 I halfs]: per k-tile the dwords (k0,k1) (k8,k9)
      const int tile = min(2 * rt + (i >> 3), (p.wrows + 7) / 8 - 1);
      if constexpr (I == 2) {
        st.w0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wd + (((int64_t)tile * (p.ktiles_padded / 2) + sc) * 32 + t) * 4));
        st.w1 = st.w0;
      } else {
        const u32x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wd + (((int64_t)tile * p.ktiles_padded + 2 * sc) * 32 + t) * 2));
        const u32x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wd + (((int64_t)tile * p.ktiles_padded + 2 * sc + 1) * 32 + t) * 2));
        st.w0 = u32x4{v0[0], v0[1], v1[0], v1[1]};
        st.w1 = st.w0;
      }
    }
    st.x = *reinterpret_cast<const u32x4*>(p.x + ((int64_t)xrow * p.k + 32 * sc) * 2 + 16 * Q);   // dwords 4Q ... 4Q + 3: transposed at the consumer
  };
  auto compute_step = [&](const Step& st) {
    u32x4 a;
    if constexpr (LAYOUT_A) {
      const int h = i >> 3;
      a = u32x4{h ? st.w0[1] : st.w0[0], h ? st.w0[3] : st.w0[2], h ? st.w1[1] : st.w1[0], h ? st.w1[3] : st.w1[2]};
    } else {
      a = st.w0;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = row_ok ? a[e] : 0u;
    const u32x4 xt = transpose_rows4(st.x);
    const u32x4 xv = {xcol ? xt[0] : 0u, xcol ? xt[1] : 0u, xcol ? xt[2] : 0u, xcol ? xt[3] : 0u};
    acc = DT::mfma(a, xv, acc);
  };
  {
    // this wave's steps: wave, wave + WAVES, ...: nw of them.  Rounds of F16_RING steps whose refills are all in range run without a branch
    // around a load (exact vmcnt); the last round(s) only consume (a refill past the end would be real work for the vector-memory path).
    const int nw = (nsteps_total - wave + WAVES - 1) / WAVES;
    Step ring[F16_RING];
#pragma unroll
    for (int j = 0; j < F16_RING; ++j) load_step(wave + j * WAVES, ring[j]);   // (clamped: a wave with fewer steps loads its last one again)
    int base = 0;
    for (; base + 2 * F16_RING <= nw; base += F16_RING) {
#pragma unroll
      for (int j = 0; j < F16_RING; ++j) {
        compute_step(ring[j]);
        load_step(wave + (base + F16_RING + j) * WAVES, ring[j]);
      }
    }
    // the remainder: fewer than two rounds; refills only where a step exists
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < F16_RING; ++j) {
        const int jj = base + r * F16_RING + j;
        if (jj < nw) {
          compute_step(ring[j]);
          if (jj + F16_RING < nw) load_step(wave + (jj + F16_RING) * WAVES, ring[j]);
        }
      }
    }
  }
  s_red[wave * 64 + lane] = acc;
  __syncthreads();
  if (tid < 256) {
    const int c = tid >> 4, rr = tid & 15;
    const float* red = reinterpret_cast<const float*>(s_red);
    const int src = (((rr >> 2) * 16 + c) << 2) + (rr & 3);
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) sum += red[wv * 256 + src];
    const int col = ct * 16 + c, rowg = row0 + rr;
    if (col < p.m && rowg < p.wrows) reinterpret_cast<uint16_t*>(p.y)[(int64_t)col * p.wrows + rowg] = DT::from_f32(sum);
  }
}

// ---- packing kernels (integer only, bit-exact) -----------------------------------------------
// One workgroup stages a [ROWS x KB] tile of codes as bytes in LDS with fully coalesced 16-byte
// reads of the int32 input, then every thread assembles output words from four 2-byte LDS reads
// and writes them contiguously (the packed tile is contiguous in the output tensor).

// Bint4: tile = 8 rows (one n-tile) x KB k.   ref TinyGemmConvertB.cu:252-308
template <int I>
__global__ void __launch_bounds__(256) pack_Bint4_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                        int64_t n, int64_t k, int64_t ksuper) {
  constexpr int KB = 512;  // k per workgroup; multiple of 16*I for I <= 8
  // codes are staged as full 32-bit values: the reference ORs the shifted UNMASKED inputs (TinyGemmConvertB.cu:302-303),
  // so out-of-range codes must reach the pack expression untouched for the words to stay bit-identical
  __shared__ uint32_t s_codes[8][KB + 4];
  const int tid = threadIdx.x;
  const int64_t nT = blockIdx.y;
  const int64_t kb0 = (int64_t)blockIdx.x * KB;
  // load: 8 rows x 512 ints = 1024 x int4
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 256 + tid;
    const int rr = idx >> 7, c4 = idx & 127;
    const int64_t row = nT * 8 + rr, kk = kb0 + c4 * 4;
    int4 v = {0, 0, 0, 0};
    if (row < n && kk < k) v = *reinterpret_cast<const int4*>(in + row * k + kk);  // k % 32 == 0 -> whole int4 in range
    *reinterpret_cast<int4*>(&s_codes[rr][c4 * 4]) = v;
  }
  __syncthreads();
  // words of this tile: [kS_local][t][j], KB/(16 I) super-tiles x 32 x I/2 = KB words
  constexpr int WORDS = KB;  // 8 rows * KB / 8
#pragma unroll
  for (int it = 0; it < WORDS / 256; ++it) {
    const int wi = it * 256 + tid;
    const int j = wi % (I / 2);
    const int t = (wi / (I / 2)) & 31;
    const int ksl = wi / (16 * I);
    const int64_t ks = kb0 / (16 * I) + ksl;
    if (ks >= ksuper) continue;
    const int rr = t >> 2, q = t & 3;
    const int kl = (ksl * I + 2 * j) * 16 + 2 * q;
    const uint32_t* src = &s_codes[rr][kl];
    uint32_t v[8];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      v[2 * pr] = src[8 * pr];
      v[2 * pr + 1] = src[8 * pr + 1];
    }
    const uint32_t pack = (v[7] << 28) | (v[5] << 24) | (v[3] << 20) | (v[1] << 16) | (v[6] << 12) | (v[4] << 8) | (v[2] << 4) | v[0];
    out[((nT * ksuper + ks) * 32 + t) * (I / 2) + j] = (int32_t)pack;
  }
}

// Aint4: tile = 16 rows (one m-tile) x KB k.   ref TinyGemmConvertA.cu:226-285
template <int I>
__global__ void __launch_bounds__(256) pack_Aint4_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                        int64_t m, int64_t k, int64_t ksuper) {
  constexpr int KB = 256;
  __shared__ uint32_t s_codes[16][KB + 4];  // full 32-bit codes, see pack_Bint4_kernel
  const int tid = threadIdx.x;
  const int64_t mT = blockIdx.y;
  const int64_t kb0 = (int64_t)blockIdx.x * KB;
  const bool vec_ok = (k & 3) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 256 + tid;
    const int rr = idx >> 6, c4 = idx & 63;
    const int64_t row = mT * 16 + rr, kk = kb0 + c4 * 4;
    int4 v = {0, 0, 0, 0};
    if (row < m) {
      if (vec_ok && kk + 3 < k) {
        v = *reinterpret_cast<const int4*>(in + row * k + kk);
      } else {
        if (kk < k) v.x = in[row * k + kk];
        if (kk + 1 < k) v.y = in[row * k + kk + 1];
        if (kk + 2 < k) v.z = in[row * k + kk + 2];
        if (kk + 3 < k) v.w = in[row * k + kk + 3];
      }
    }
    *reinterpret_cast<int4*>(&s_codes[rr][c4 * 4]) = v;
  }
  __syncthreads();
  // words of this tile: [kS_local][t][inner]: KB/16 k-tiles x 32 = 512 words
  constexpr int WORDS = KB * 2;
#pragma unroll
  for (int it = 0; it < WORDS / 256; ++it) {
    const int wi = it * 256 + tid;
    const int inner = wi % I;
    const int t = (wi / I) & 31;
    const int ksl = wi / (32 * I);
    const int64_t ks = kb0 / (16 * I) + ksl;
    if (ks >= ksuper) continue;
    const int m0 = t >> 2, q = t & 3;
    const int kl = (ksl * I + inner) * 16 + 2 * q;
    const uint32_t v0 = s_codes[m0][kl], v1 = s_codes[m0][kl + 1];              // (m0,k0) (m0,k1)
    const uint32_t v2 = s_codes[m0 + 8][kl], v3 = s_codes[m0 + 8][kl + 1];      // (m1,k0) (m1,k1)
    const uint32_t v4 = s_codes[m0][kl + 8], v5 = s_codes[m0][kl + 9];          // (m0,k2) (m0,k3)
    const uint32_t v6 = s_codes[m0 + 8][kl + 8], v7 = s_codes[m0 + 8][kl + 9];  // (m1,k2) (m1,k3)
    const uint32_t pack = (v7 << 28) | (v5 << 24) | (v3 << 20) | (v1 << 16) | (v6 << 12) | (v4 << 8) | (v2 << 4) | v0;
    out[((mT * ksuper + ks) * 32 + t) * I + inner] = (int32_t)pack;
  }
}

// ---- unpack: one thread per code; the index arithmetic is the packers' read backwards (TinyGemmConvertA.cu:226-285,
// TinyGemmConvertB.cu:252-308: pack = v7<<28 | v5<<24 | v3<<20 | v1<<16 | v6<<12 | v4<<8 | v2<<4 | v0) ----
__global__ void __launch_bounds__(256) unpack_int4_kernel(const uint32_t* __restrict__ packed, int32_t* __restrict__ codes, int layout_a,
                                                          int64_t rows, int64_t k, int I, int64_t ksuper) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * k) return;
  const int64_t r = idx / k, kk = idx - r * k;
  const int64_t kt = kk >> 4;
  const int kq = (int)(kk & 15);
  int64_t word;
  int v;
  if (layout_a) {  // v = [(m0,k0),(m0,k1),(m1,k0),(m1,k1),(m0,k2),(m0,k3),(m1,k2),(m1,k3)], k0 = 2 (t % 4), k2 = k0 + 8
    const int rr = (int)(r & 15), m0 = rr & 7, hi = rr >> 3;
    const int t = 4 * m0 + ((kq & 7) >> 1);
    v = (kq >> 3) * 4 + hi * 2 + (kq & 1);
    word = (((r >> 4) * ksuper + kt / I) * 32 + t) * I + kt % I;
  } else {         // word j of a lane: k-tiles 2j (v0..v3) and 2j + 1 (v4..v7) of the super-tile, k = base + 2 (t % 4) + {0, 1, 8, 9}
    const int t = 4 * (int)(r & 7) + ((kq & 7) >> 1);
    const int ktl = (int)(kt % I);
    v = (ktl & 1) * 4 + (kq & 1) + 2 * (kq >> 3);
    word = (((r >> 3) * ksuper + kt / I) * 32 + t) * (I / 2) + (ktl >> 1);
  }
  const int shift = (v & 1) * 16 + (v >> 1) * 4;
  codes[idx] = (int32_t)((packed[word] >> shift) & 15u);
}

// ---- dequantise a Bint4-packed weight matrix into row-major 16-bit values (what a GEMM library multiplies for MANY activation rows) ----
// w[r][k] = RNE16(fma(f32(lut[r][code]), f32(scale[g][r]), f32(zero[g][r]))) -- the reference's dequantisation, element for element
// (MatrixLayoutB.cuh:1042-1046; int4: lut = code - 8, Dequantization.cuh:136-178).  Thread = (row, 64-k super-tile of innerKTiles 4 /
// 32-k of 2 / 128-k of 8): its words are 4 lanes x I / 2 words = 8 I contiguous bytes of the packed layout (ConvertB.cu:252-308), its
// output 32 I contiguous bytes of the row.
template <typename DT, int I, int CHK>
__global__ void __launch_bounds__(256) dequant_w4_kernel(const uint32_t* __restrict__ packed, const uint16_t* __restrict__ qinfo, const uint16_t* __restrict__ lut,
                                                        uint16_t* __restrict__ out, int64_t rows, int64_t wrows_q, int64_t k, int64_t ksuper, int gshift, int qtype) {
  constexpr int W = I / 2;   // words per lane of the packed layout = 32-k runs per super-tile
  // lane = (super-tile, word column j, run h of 8 consecutive k) of a 512-k chunk: a quad of lanes writes 64 contiguous bytes, the 4 W lanes
  // of a super-tile 32 I contiguous bytes, consecutive super-tiles follow: whole lines per wave-store; the 4 words a lane needs (lanes
  // 0 ... 3 of its row, column j) are the same for the four h -- one request per quad.
  // A WAVE is CHK consecutive 512-k chunks of ONE row (host: k a multiple of 512 CHK): at most 16 quantisation groups per chunk.  Their
  // dequantised tables -- 16 values RNE16(fma(lut[e], scale, zero)) per group, one fma per lane and round of 64 -- go to the wave's own LDS,
  // and every weight is then ONE 2-byte LDS read at table + 2 code (a 32-byte table is 8 banks: different entries never collide, equal
  // ones broadcast) instead of an 8-way select and an fma per element (~150 vector ops per 16 bytes of output: 31 us for a 4096 x 4096
  // matrix against 14 with one chunk per wave; CHK = 4: every load of the wave's 2048 k in flight before the first table is built).
  __shared__ uint16_t tables[4][CHK][16][16];  // [wave][chunk][group of the chunk][entry]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // blockIdx.y (+ 65535 blockIdx.z) = the row, blockIdx.x = a 256-thread piece of it: no 64-bit division by the run-time row length
  const int per_row = (int)(k >> 3) / CHK;             // threads per row (a multiple of 64)
  const int64_t r = (int64_t)blockIdx.y + (int64_t)blockIdx.z * 65535;
  const int tr = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (r >= rows || tr - lane >= per_row) return;       // (wave-uniform)
  const int64_t k0w = (int64_t)(tr - lane) * 8 * CHK;  // first k of the wave
  const int ngw = (512 >> gshift) > 0 ? (512 >> gshift) : 1;   // groups of a chunk (g = 256 / 128 / 64 / 32: 2 / 4 / 8 / 16)
  // ---- requests: the packed words of every chunk, then the table inputs ----
  uint32_t wd[CHK][4];
  int hh[CHK];
  const uint16_t* tb[CHK];
#pragma unroll
  for (int c = 0; c < CHK; ++c) {
    const int t = (int)((k0w >> 3) + c * 64 + lane);   // this lane's 8-k run of the row
    const int h = t & 3, j = (t >> 2) % W;
    const int64_t s = t / (4 * W);
    const uint32_t* src = packed + (((r >> 3) * ksuper + s) * 32 + 4 * (r & 7)) * W + j;
#pragma unroll
    for (int i = 0; i < 4; ++i) wd[c][i] = src[i * W];
    hh[c] = h;
    const int64_t k0 = s * (16 * I) + j * 32;
    tb[c] = &tables[wave][c][(int)((k0 >> gshift) - ((k0w + c * 512) >> gshift))][0];
  }
  for (int t = lane; t < CHK * ngw * 16; t += 64) {
    const int e = t & 15, cg = t >> 4, c = cg / ngw, gw = cg - c * ngw;
    float lv;
    if (qtype == TG_Q_INT4) lv = (float)(e - 8);
    else lv = DT::lo_f32((uint32_t)lut[(qtype == TG_Q_ANY4_ROWWISE ? r * 16 : 0) + e]);
    const uint32_t sz = reinterpret_cast<const uint32_t*>(qinfo)[(((k0w + c * 512) >> gshift) + gw) * wrows_q + r];
    tables[wave][c][gw][e] = DT::from_f32(__builtin_fmaf(lv, DT::lo_f32(sz), DT::hi_f32(sz)));
  }
  // (the region is the wave's own and a wave's LDS operations execute in order: no barrier)
  // word i holds k = 2 i + {0, 1, 8, 9, 16, 17, 24, 25} of the run of 32 in the nibbles (v & 1) * 16 + (v >> 1) * 4, v = 0 ... 7: the pair
  // (k, k + 1) = (2 i + 8 h, 2 i + 8 h + 1) sits at bits 4 h and 16 + 4 h
#pragma unroll
  for (int c = 0; c < CHK; ++c) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t c0 = (wd[c][i] >> (hh[c] * 4)) & 15u, c1 = (wd[c][i] >> (16 + hh[c] * 4)) & 15u;
      o[i] = (uint32_t)tb[c][c0] | ((uint32_t)tb[c][c1] << 16);
    }
    *reinterpret_cast<u32x4*>(out + r * k + k0w + (int64_t)(c * 64 + lane) * 8) = o;
  }
}

// ---- 16-bit fragment-order conversions (pure data movement) ------------------------------------
// ref TinyGemmConvertA.cu:19-141 / 442-546 and TinyGemmConvertB.cu:20-66 / 136-176
__global__ void __launch_bounds__(256) to_A16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                    int64_t m, int64_t k, int64_t mTiles, int64_t kTiles) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= mTiles * kTiles * 32) return;
  const int t = gid & 31;
  const int64_t kT = (gid >> 5) % kTiles, mT = (gid >> 5) / kTiles;
  const int64_t m0 = mT * 16 + (t >> 2), m1 = m0 + 8;
  const int64_t k0 = kT * 16 + (t & 3) * 2;
  uint16_t v[8];
  auto at = [&](int64_t rr, int64_t cc) -> uint16_t { return (rr < m && cc < k) ? in[rr * k + cc] : (uint16_t)0; };
  v[0] = at(m0, k0); v[1] = at(m0, k0 + 1); v[2] = at(m1, k0); v[3] = at(m1, k0 + 1);
  v[4] = at(m0, k0 + 8); v[5] = at(m0, k0 + 9); v[6] = at(m1, k0 + 8); v[7] = at(m1, k0 + 9);
  u32x4 o = {v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16)};
  *reinterpret_cast<u32x4*>(out + gid * 8) = o;
}

__global__ void __launch_bounds__(256) from_A16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                      int64_t m, int64_t k, int64_t mTiles, int64_t kTiles) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= mTiles * kTiles * 32) return;
  const int t = gid & 31;
  const int64_t kT = (gid >> 5) % kTiles, mT = (gid >> 5) / kTiles;
  const u32x4 o = *reinterpret_cast<const u32x4*>(in + gid * 8);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t rr = mT * 16 + (t >> 2) + 8 * ((e >> 1) & 1);
    const int64_t cc = kT * 16 + (t & 3) * 2 + 8 * (e >> 2) + (e & 1);
    const uint16_t val = (uint16_t)(o[e >> 1] >> (16 * (e & 1)));
    if (rr < m && cc < k) out[rr * k + cc] = val;
  }
}

__global__ void __launch_bounds__(256) to_B16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                    int64_t n, int64_t k, int64_t nTiles, int64_t totalK, int inner) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nTiles * totalK * 32) return;
  const int t = gid & 31;
  const int64_t kT = (gid >> 5) % totalK, nT = (gid >> 5) / totalK;
  const int64_t n0 = nT * 8 + (t >> 2);
  const int64_t k0 = kT * 16 + (t & 3) * 2;
  auto at = [&](int64_t cc) -> uint32_t { return (n0 < n && cc < k) ? in[n0 * k + cc] : 0u; };
  u32x2 o = {at(k0) | (at(k0 + 1) << 16), at(k0 + 8) | (at(k0 + 9) << 16)};
  uint16_t* dst = out + ((nT * (totalK / inner) + kT / inner) * 32 + t) * (4 * inner) + (kT % inner) * 4;
  *reinterpret_cast<u32x2*>(dst) = o;
}

__global__ void __launch_bounds__(256) from_B16_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                      int64_t n, int64_t k, int64_t nTiles, int64_t kTiles,
                                                      int64_t outerK, int inner) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= nTiles * kTiles * 32) return;
  const int t = gid & 31;
  const int64_t kT = (gid >> 5) % kTiles, nT = (gid >> 5) / kTiles;
  const int64_t n0 = nT * 8 + (t >> 2);
  if (n0 >= n) return;
  const uint16_t* src = in + ((nT * outerK + kT / inner) * 32 + t) * (4 * inner) + (kT % inner) * 4;
  const u32x2 o = *reinterpret_cast<const u32x2*>(src);
  const int64_t k0 = kT * 16 + (t & 3) * 2;
  const int64_t ks[4] = {k0, k0 + 1, k0 + 8, k0 + 9};
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (ks[e] < k) out[n0 * k + ks[e]] = (uint16_t)(o[e >> 1] >> (16 * (e & 1)));
}

// debug op, ref TinyGemmDequantize.cu:19-34 (grid-stride, one word -> 8 bf16 = 16 bytes)
__global__ void __launch_bounds__(256) dequant_int4_kernel(const int32_t* __restrict__ in, u32x4* __restrict__ out, int64_t count) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < count; idx += (int64_t)gridDim.x * 256) {
    const uint32_t w = (uint32_t)in[idx];
    u32x4 o;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const float lo = (float)((int)((w >> (4 * ii)) & 0xfu) - 8);
      const float hi = (float)((int)((w >> (4 * ii + 16)) & 0xfu) - 8);
      o[ii] = BF16::pack2(lo, hi);
    }
    out[idx] = o;
  }
}


#ifdef TG_DEV  // developer builds only (-DTG_DEV): geometry override through the environment, never in the shipped library
int g_dbg_variant = 0;
#else
constexpr int g_dbg_variant = 0;
#endif

// Launch geometry.  Streaming shapes (many tiles) use 8-wave workgroups, two per CU, and the
// smallest split-K that still puts >= ~16 waves on every CU; a single small matrix (one tile per
// CU) uses one 16-wave workgroup per tile with split-K 16.
struct Geometry {
  int waves, splitk, sk_shift;
};

inline Geometry pick_geometry(int64_t rowtiles, int64_t coltiles, int64_t batch, int64_t nsteps) {
  const int64_t tiles = rowtiles * coltiles * batch;
  const int64_t want_waves = 256 * 16;  // 256 CUs x 16 waves
  Geometry g;
  if (g_dbg_variant > 0) {  // developer override: variant = 100 * waves + splitk
    g.waves = g_dbg_variant / 100;
    g.splitk = g_dbg_variant % 100;
  } else if (tiles * 8 <= want_waves) {
    g.waves = 16;
    g.splitk = 16;
  } else {
    g.waves = 8;
    int sk = 1;
    while (sk < 8 && tiles * sk * 2 <= want_waves) sk *= 2;
    g.splitk = sk;
  }
  while (g.splitk > 1 && g.splitk > nsteps) g.splitk >>= 1;
  g.sk_shift = 0;
  while ((1 << g.sk_shift) < g.splitk) ++g.sk_shift;
  return g;
}


// ---- int8 packers (reference TinyGemmConvertB.cu:366-411, TinyGemmConvertA.cu:337-397): one thread per output word.
// The OR of the shifted 32-bit inputs is kept exactly as written there (inputs above 255 bleed into higher bytes).
template <int I>
__global__ void __launch_bounds__(256) pack_Bint8_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t n,
                                                         int64_t k, int64_t ksuper, int64_t total) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int j = (int)(o % I), t = (int)((o / I) % 32);
    const int64_t ks_ = (o / (I * 32)) % ksuper, nt = o / (I * 32 * ksuper);
    const int64_t n0 = nt * 8 + t / 4, kb = (ks_ * I + j) * 16 + (t % 4) * 2;
    uint32_t v[4] = {0u, 0u, 0u, 0u};
    if (n0 < n) {
      const int32_t* r = in + n0 * k;
      if (kb < k) v[0] = (uint32_t)r[kb];
      if (kb + 1 < k) v[1] = (uint32_t)r[kb + 1];
      if (kb + 8 < k) v[2] = (uint32_t)r[kb + 8];
      if (kb + 9 < k) v[3] = (uint32_t)r[kb + 9];
    }
    out[o] = (int32_t)((v[3] << 24) | (v[1] << 16) | (v[2] << 8) | v[0]);
  }
}

template <int I>
__global__ void __launch_bounds__(256) pack_Aint8_kernel(const int32_t* __restrict__ in, int32_t* __restrict__ out, int64_t m,
                                                         int64_t k, int64_t kouter, int64_t total) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int w = (int)(o % 2), j = (int)((o / 2) % I), t = (int)((o / (2 * I)) % 32);
    const int64_t ko = (o / (2 * I * 32)) % kouter, mt = o / (2 * I * 32 * kouter);
    const int64_t m0 = mt * 16 + t / 4, m1 = m0 + 8;
    const int64_t ka = (ko * I + j) * 16 + (t % 4) * 2 + 8 * w;  // word 0: k0, k0+1; word 1: k0+8, k0+9
    uint32_t v0 = 0u, v1 = 0u, v2 = 0u, v3 = 0u;                 // (m0,ka) (m0,ka+1) (m1,ka) (m1,ka+1)
    if (m0 < m && ka < k) v0 = (uint32_t)in[m0 * k + ka];
    if (m0 < m && ka + 1 < k) v1 = (uint32_t)in[m0 * k + ka + 1];
    if (m1 < m && ka < k) v2 = (uint32_t)in[m1 * k + ka];
    if (m1 < m && ka + 1 < k) v3 = (uint32_t)in[m1 * k + ka + 1];
    out[o] = (int32_t)((v3 << 24) | (v1 << 16) | (v2 << 8) | v0);
  }
}

template <typename DT, bool LAYOUT_A, int I>
int launch_w8(GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int WAVES = 8;
  const int64_t tiles = (int64_t)p.rowtiles * coltiles * batch;
  const int nsteps = (p.k / 16 + 3) / 4;
  int sk = 1;
  while (sk < WAVES && tiles * sk < 256 * 16 && sk * 2 <= nsteps) sk *= 2;
  p.splitk = sk;
  p.sk_shift = 0;
  while ((1 << p.sk_shift) < sk) ++p.sk_shift;
  const int tpb = WAVES / sk;
  dim3 grid((unsigned)((p.rowtiles + tpb - 1) / tpb), (unsigned)coltiles, (unsigned)batch);
  hipLaunchKernelGGL((w8_gemm_kernel<DT, LAYOUT_A, I, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
  return launch_status();
}


// Dispatch of one validated 4-bit GEMM call to a kernel family (each family's launch path is its own translation unit, tg_common.cuh).
// dt = TG_BF16 / TG_F16, canon = words per lane-quad of the packed layout (CANON_*), qmx = mx4.
int launch_w4(int dt, bool LAYOUT_A, int canon, bool QMX, GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  const int KSTEP = LAYOUT_A ? 64 : 128;
  const Geometry g = pick_geometry(p.rowtiles, coltiles, batch, (p.k + KSTEP - 1) / KSTEP);
  p.splitk = g.splitk;
  p.sk_shift = g.sk_shift;
  const int tpb = g.waves / g.splitk;
  dim3 grid((unsigned)((p.rowtiles + tpb - 1) / tpb), (unsigned)coltiles, (unsigned)batch);
  // Streaming shapes go to the lane-owns-group kernel when the quantisation group covers at least one
  // unit of its walk (Bint4: g >= 128, Aint4: g >= 64).  TG_STREAM=0 forces the split-K kernel.
#ifdef TG_DEV
  static const int use_stream = getenv("TG_STREAM") ? atoi(getenv("TG_STREAM")) : 1;
#else
  constexpr int use_stream = 1;
#endif
  const int WPL = canon == CANON_NONE ? 1 : (canon == CANON_PAIR ? 2 : 4);
  // TG_NUM_FAST, weights on the B side: the pair-table kernel (group-scaled numerics) whenever its LDS plan fits
  // (mx4 in BOTH numerics: its dequantised weights, fp4 * 2^(e - 127), are exact 16-bit values however they are formed, so the
  //  pair-table kernels -- which convert them with v_cvt_scalef32_pk_bf16_fp4 -- ARE the reference arithmetic for it)
  if (p.numerics == TG_NUM_FAST || p.numerics == TG_NUM_FAST_MFMA || QMX) {
    int rc;
    if (!LAYOUT_A) {
      // one layer per launch with up to 4 activation rows (a decode step's GEMMs): its own kernel (w4_gemv.cuh)
      rc = p.numerics == TG_NUM_FAST_MFMA ? (int)TG_PAIR_NA : tgx::gemv(dt, 2 * WPL, QMX, p, batch, st);  // (w4_gemv contracts with v_dot2)
      if (rc != TG_PAIR_NA) return rc;
      // STACKED launches of one activation row in the default numerics: the contraction tg_m1_default_contraction() names
      // (TG_M1_DEFAULT_MFMA; TG_NUM_FAST_DOT2 / TG_NUM_FAST_MFMA pin either one for A/B runs -- p.dot2 is set by the entry point)
      if (TG_M1_DEFAULT_MFMA && p.numerics == TG_NUM_FAST && p.m == 1 && !p.dot2 && !QMX) p.numerics = TG_NUM_FAST_MFMA;
      rc = tgx::pair_xr(dt, 2 * WPL, QMX, p, batch, st);
      if (rc != TG_PAIR_NA) return rc;
      p.ws_need = 0;
      // one layer per launch with more 16-row tiles than CUs (5 ... 16 rows, k = 4096): in front of the persistent kernel, which would
      // take it from 192 64-row items on (12288 rows: 15.6 us per graph node)
      rc = tgx::pair16_loop(dt, 2 * WPL, QMX, p, batch, st);
      if (rc != TG_PAIR_NA) return rc;
    }
    if (LAYOUT_A) rc = tgx::pair_a(dt, WPL, QMX, p, batch, st);
    else rc = tgx::pair(dt, 2 * WPL, QMX, p, batch, st);
    if (rc != TG_PAIR_NA) return rc;
    p.ws_need = 0;
    if (!LAYOUT_A) {
      if (p.m > 8) {
        rc = tgx::pair_b16(dt, 2 * WPL, QMX, p, batch, st);
        if (rc != TG_PAIR_NA) return rc;
        p.ws_need = 0;
      }
      rc = tgx::pair16(dt, 2 * WPL, QMX, p, batch, st);
      if (rc != TG_PAIR_NA) return rc;
    }
  }
  if (p.x_tc || p.y_tc) return TG_E_LAYOUT;  // only the pair-table kernels read / write fragment order themselves
  if (p.norm_w || p.epilogue) return TG_E_FUSION;  // ... and only they carry the fused norm / SwiGLU stages
#ifdef TG_DEV_MIN  // developer A/B builds carry the pair-table kernels only (a third of the build time)
  return TG_E_SHAPE;
#else
  // m = 1 always streams: with private X slabs its split-K variants beat the latency kernel down to one matrix
  if (use_stream && (g.waves == 8 || p.m == 1 || use_stream == 2) && (1 << p.gshift) >= (LAYOUT_A ? 64 : 128)) {
    return tgx::stream(dt, LAYOUT_A, WPL, QMX, p, coltiles, batch, st);
  }
  if (p.dry) return TG_PLAN_SPLITK;
  return tgx::splitk(dt, LAYOUT_A, canon, QMX, g.waves, p, grid, st);
#endif
}

}  // namespace


extern "C" {

int tg_abi_version(void) { return TG_ABI_VERSION; }
int tg_m1_default_contraction(void) { return TG_M1_DEFAULT_MFMA ? 1 : 0; }

const char* tg_error_string(int code) {
  switch (code) {
    case 0: return "ok";
    case TG_E_NULL: return "a required tensor is missing (null data pointer)";
    case TG_E_INNER_K: return "innerKTiles is not valid for this layout (Aint4/A: 1,2,4; Bint4: 2,4,8; B16: 1,2; A16: 1)";
    case TG_E_K_DIV: return "k must be a multiple of 32 and of innerKTiles * 16 (isEvenDivisor(k, 32), isEvenDivisor(kTiles, innerKTiles))";
    case TG_E_GROUP: return "qGroupSize must be 32, 64, 128 or 256 and divide k";
    case TG_E_DTYPE: return "activation dtype must be bfloat16 or float16 (mx4: bfloat16 only)";
    case TG_E_QTYPE: return "unknown 4-bit quantization type";
    case TG_E_SHAPE: return "inconsistent or non-positive sizes";
    case TG_E_ALIGN: return "device buffers must be 16-byte aligned";
    case TG_E_DEVICE: return "could not select the requested device";
    case TG_E_SIZE: return "an operand is too large for the kernels' 32-bit byte offsets (activations, packed weights or quantisation info of one problem must stay below 2 GiB; at most 65535 16-row activation tiles)";
    case TG_E_INTERNAL: return "internal error: a kernel that addresses LDS from offset 0 was built with static LDS";
    case TG_E_STRUCT: return "tg_w4_gemm.struct_bytes must be sizeof(struct tg_w4_gemm) of the header the caller was built with (at least the ABI-1 prefix, at most this library's struct)";
    case TG_E_FUSION: return "no kernel with the requested fused stage (norm_weight / epilogue) for this problem: run that stage as its own launch (include/decode_glue_hip.h)";
    case TG_E_LAYOUT: return "fragment-order activations / outputs (x_layout, y_layout) are not available for this problem: convert with tg_convert_{from,to}_A16 around a row-major call";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown tinygemm error";
  }
}

int tg_convert_to_Bint4(const int32_t* in, int64_t n, int64_t k, int I, int32_t* out, int device, tg_stream_t stream) {
  if (!in || !out) return TG_E_NULL;
  if (!(I == 2 || I == 4 || I == 8)) return TG_E_INNER_K;  // ConvertB.cu:327
  if (n <= 0 || k <= 0) return TG_E_SHAPE;
  if (k % (I * 16) != 0) return TG_E_K_DIV;               // ConvertB.cu:337
  if (!aligned16(in)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t nTiles = cdiv(n, 8), ksuper = k / (I * 16);
  dim3 grid((unsigned)cdiv(k, 512), (unsigned)nTiles);
  hipStream_t st = (hipStream_t)stream;
  if (I == 2) hipLaunchKernelGGL(pack_Bint4_kernel<2>, grid, dim3(256), 0, st, in, out, n, k, ksuper);
  else if (I == 4) hipLaunchKernelGGL(pack_Bint4_kernel<4>, grid, dim3(256), 0, st, in, out, n, k, ksuper);
  else hipLaunchKernelGGL(pack_Bint4_kernel<8>, grid, dim3(256), 0, st, in, out, n, k, ksuper);
  return launch_status();
}

int tg_convert_to_Aint4(const int32_t* in, int64_t m, int64_t k, int I, int32_t* out, int device, tg_stream_t stream) {
  if (!in || !out) return TG_E_NULL;
  if (!(I == 1 || I == 2 || I == 4)) return TG_E_INNER_K;  // ConvertA.cu:299
  if (m <= 0 || k <= 0) return TG_E_SHAPE;
  if (!aligned16(in)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t mTiles = cdiv(m, 16), ksuper = cdiv(k, I * 16);
  dim3 grid((unsigned)cdiv(ksuper * I * 16, 256), (unsigned)mTiles);
  hipStream_t st = (hipStream_t)stream;
  if (I == 1) hipLaunchKernelGGL(pack_Aint4_kernel<1>, grid, dim3(256), 0, st, in, out, m, k, ksuper);
  else if (I == 2) hipLaunchKernelGGL(pack_Aint4_kernel<2>, grid, dim3(256), 0, st, in, out, m, k, ksuper);
  else hipLaunchKernelGGL(pack_Aint4_kernel<4>, grid, dim3(256), 0, st, in, out, m, k, ksuper);
  return launch_status();
}

int tg_unpack_int4(const int32_t* packed, int layout_a, int64_t rows, int64_t k, int I, int32_t* codes, int device, tg_stream_t stream) {
  if (!packed || !codes) return TG_E_NULL;
  if (layout_a ? !(I == 1 || I == 2 || I == 4) : !(I == 2 || I == 4 || I == 8)) return TG_E_INNER_K;
  if (rows <= 0 || k <= 0 || rows * k / 256 > INT32_MAX) return TG_E_SHAPE;
  if (!layout_a && k % (16 * I) != 0) return TG_E_K_DIV;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t ksuper = cdiv(k, 16 * I);
  hipLaunchKernelGGL(unpack_int4_kernel, dim3((unsigned)cdiv(rows * k, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint32_t*>(packed), codes, layout_a, rows, k, I, ksuper);
  return launch_status();
}

int tg_dequant_w4(const void* packed, const void* qinfo, const void* lut, int64_t wrows, int64_t k, int group, int qtype, int dtype, int I, void* out,
                  int device, tg_stream_t stream) {
  return tg_dequant_w4_panel(packed, qinfo, lut, wrows, wrows, k, group, qtype, dtype, I, out, device, stream);
}

int tg_dequant_w4_panel(const void* packed, const void* qinfo, const void* lut, int64_t wrows, int64_t wrows_q, int64_t k, int group, int qtype,
                        int dtype, int I, void* out, int device, tg_stream_t stream) {
  if (!packed || !qinfo || !out) return TG_E_NULL;
  if (wrows_q < wrows) return TG_E_SHAPE;
  if (!(qtype == TG_Q_INT4 || qtype == TG_Q_ANY4_GLOBAL || qtype == TG_Q_ANY4_ROWWISE)) return TG_E_QTYPE;
  if (qtype != TG_Q_INT4 && !lut) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (!(I == 2 || I == 4 || I == 8)) return TG_E_INNER_K;
  if (wrows <= 0 || k <= 0 || wrows % 8 != 0 || wrows > INT32_MAX || k > INT32_MAX) return TG_E_SHAPE;
  if (k % (16 * I) != 0 || k % 32 != 0) return TG_E_K_DIV;
  if (k % 512 != 0) return TG_E_K_DIV;   // (a wave of the kernel is 512 consecutive k of one row)
  if (!(group == 32 || group == 64 || group == 128 || group == 256) || k % group != 0) return TG_E_GROUP;
  if (!aligned16(packed) || !aligned16(out) || (reinterpret_cast<uintptr_t>(qinfo) & 3u) || (lut && !aligned16(lut))) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t ksuper = k / (16 * I);
  const int gshift = group == 32 ? 5 : group == 64 ? 6 : group == 128 ? 7 : 8;
  const int chk = k % 2048 == 0 ? 4 : 1;   // chunks of 512 k per wave
  const unsigned bs = k / 8 / chk < 256 ? (unsigned)(k / 8 / chk) : 256u;   // (a row's threads: a multiple of 64)
  const dim3 grid((unsigned)cdiv(k / 8 / chk, 256), (unsigned)(wrows < 65535 ? wrows : 65535), (unsigned)cdiv(wrows, 65535));
#define TG_DQ2(DTT, I_, C_)                                                                                                                       \
  hipLaunchKernelGGL((dequant_w4_kernel<DTT, I_, C_>), grid, dim3(bs), 0, (hipStream_t)stream, (const uint32_t*)packed, (const uint16_t*)qinfo, \
                     (const uint16_t*)lut, (uint16_t*)out, wrows, wrows_q, k, ksuper, gshift, qtype)
#define TG_DQ(DTT, I_) do { if (chk == 4) TG_DQ2(DTT, I_, 4); else TG_DQ2(DTT, I_, 1); } while (0)
  if (dtype == TG_BF16) { if (I == 2) TG_DQ(BF16, 2); else if (I == 4) TG_DQ(BF16, 4); else TG_DQ(BF16, 8); }
  else { if (I == 2) TG_DQ(F16, 2); else if (I == 4) TG_DQ(F16, 4); else TG_DQ(F16, 8); }
#undef TG_DQ2
#undef TG_DQ
  return launch_status();
}

int tg_convert_to_A16(const void* rm, int64_t m, int64_t k, void* tc, int device, tg_stream_t stream) {
  if (!rm || !tc) return TG_E_NULL;
  if (m <= 0 || k <= 0) return TG_E_SHAPE;
  if (!aligned16(tc)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t mT = cdiv(m, 16), kT = cdiv(k, 16);
  hipLaunchKernelGGL(to_A16_kernel, dim3((unsigned)cdiv(mT * kT * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)rm, (uint16_t*)tc, m, k, mT, kT);
  return launch_status();
}

int tg_convert_from_A16(const void* tc, int64_t m, int64_t k, void* rm, int device, tg_stream_t stream) {
  if (!rm || !tc) return TG_E_NULL;
  if (m <= 0 || k <= 0) return TG_E_SHAPE;
  if (!aligned16(tc)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t mT = cdiv(m, 16), kT = cdiv(k, 16);
  hipLaunchKernelGGL(from_A16_kernel, dim3((unsigned)cdiv(mT * kT * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)tc, (uint16_t*)rm, m, k, mT, kT);
  return launch_status();
}

int tg_convert_to_B16(const void* rm, int64_t n, int64_t k, int I, void* tc, int device, tg_stream_t stream) {
  if (!rm || !tc) return TG_E_NULL;
  if (!(I == 1 || I == 2)) return TG_E_INNER_K;  // ConvertB.cu:84
  if (n <= 0 || k <= 0) return TG_E_SHAPE;
  if (!aligned16(tc)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t nT = cdiv(n, 8), totalK = cdiv(k, 16 * I) * I;
  hipLaunchKernelGGL(to_B16_kernel, dim3((unsigned)cdiv(nT * totalK * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)rm, (uint16_t*)tc, n, k, nT, totalK, I);
  return launch_status();
}

int tg_convert_from_B16(const void* tc, int64_t n, int64_t k, int I, void* rm, int device, tg_stream_t stream) {
  if (!rm || !tc) return TG_E_NULL;
  if (!(I == 1 || I == 2)) return TG_E_INNER_K;
  if (n <= 0 || k <= 0) return TG_E_SHAPE;
  if (!aligned16(tc)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t nT = cdiv(n, 8), kT = cdiv(k, 16), outerK = cdiv(k, 16 * I);
  hipLaunchKernelGGL(from_B16_kernel, dim3((unsigned)cdiv(nT * kT * 32, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)tc, (uint16_t*)rm, n, k, nT, kT, outerK, I);
  return launch_status();
}

int tg_dequant_int4(const int32_t* in, int64_t count, void* out_bf16, int device, tg_stream_t stream) {
  if (!in || !out_bf16) return TG_E_NULL;
  if (count <= 0) return TG_E_SHAPE;
  if (!aligned16(out_bf16)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t blocks = cdiv(count, 256);
  hipLaunchKernelGGL(dequant_int4_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0,
                     (hipStream_t)stream, in, (u32x4*)out_bf16, count);
  return launch_status();
}

// The caller's struct, as long as IT says it is (tg_w4_gemm.struct_bytes), into a zero-filled struct of this library's length:
// fields the caller's header did not have read as zero = off.  Never reads past struct_bytes.
static int take_args(const tg_w4_gemm* a, tg_w4_gemm* full) {
  if (!a) return TG_E_NULL;
  const size_t prefix = offsetof(tg_w4_gemm, stride_y) + sizeof(a->stride_y);  // ABI 1
  const size_t n = a->struct_bytes;
  if (n < prefix || n > sizeof(tg_w4_gemm) || a->struct_reserved != 0) return TG_E_STRUCT;
  memset(full, 0, sizeof(*full));
  memcpy(full, a, n);
  return 0;
}

// dry: 0 launch, 1 report the kernel family (tg_gemm_w4_plan), 2 report the workspace the fastest kernel wants
static int gemm_w4_impl(const tg_w4_gemm* caller, int device, tg_stream_t stream, int dry, int64_t* ws_need = nullptr) {
  tg_w4_gemm full;
  const int src = take_args(caller, &full);
  if (src != 0) return src;
  const tg_w4_gemm* a = &full;
  if (!a->x || !a->w || !a->qinfo || !a->y) return TG_E_NULL;
  if (a->qtype < TG_Q_INT4 || a->qtype > TG_Q_MX4) return TG_E_QTYPE;
  if ((a->qtype == TG_Q_ANY4_GLOBAL || a->qtype == TG_Q_ANY4_ROWWISE) && !a->lut) return TG_E_NULL;
  if (!(a->dtype == TG_BF16 || a->dtype == TG_F16)) return TG_E_DTYPE;
  if (a->qtype == TG_Q_MX4 && a->dtype != TG_BF16) return TG_E_DTYPE;  // TinyGemm_int4.cu:758,782
  if (a->m <= 0 || a->wrows <= 0 || a->k <= 0 || a->m > INT32_MAX || a->wrows > INT32_MAX || a->k > INT32_MAX) return TG_E_SHAPE;
  int I = a->inner_k_tiles;
  bool on_right = a->w_on_right != 0;
  if (on_right ? !(I == 2 || I == 4 || I == 8) : !(I == 1 || I == 2 || I == 4)) return TG_E_INNER_K;
  // TinyGemmImpl.cuh:370-376: kTiles % innerKTiles == 0, k % 32 == 0
  if (a->k % 32 != 0 || a->k % (16 * I) != 0) return TG_E_K_DIV;
  const int g = a->group;
  if (!(g == 32 || g == 64 || g == 128 || g == 256) || a->k % g != 0) return TG_E_GROUP;  // TinyGemm_int4.cu:379-387
  if (a->wrows % (on_right ? 8 : 16) != 0) return TG_E_SHAPE;
  if (!(a->w_format == TG_WFMT_M16N8K16 || a->w_format == TG_WFMT_ROWS) || (a->w_format && on_right) || a->reserved6 != 0) return TG_E_SHAPE;
  if (a->w_format == TG_WFMT_ROWS) {
    // the A-shaped tensor holds Bint4 words (rows padded to 16): from here on this IS a weights-on-the-right call -- both sides
    // produce [activation row][weight row] (TinyGemm_int4.cu:450-456)
    on_right = true;
    I = a->k % 64 == 0 ? 4 : 2;
  }
  const int rows_per_tile = on_right ? 8 : 16;
  if (!aligned16(a->x) || !aligned16(a->w) || (reinterpret_cast<uintptr_t>(a->qinfo) & 3u)) return TG_E_ALIGN;
  if (a->lut && !aligned16(a->lut)) return TG_E_ALIGN;          // LUT rows are read as two 16-byte vectors
  if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 7u)) return TG_E_ALIGN;
  if (!(a->numerics == TG_NUM_FAST || a->numerics == TG_NUM_REFERENCE || a->numerics == TG_NUM_FAST_MFMA || a->numerics == TG_NUM_FAST_DOT2) || a->reserved != 0) return TG_E_SHAPE;
  if (a->workspace && (!aligned16(a->workspace) || a->workspace_bytes < 0)) return TG_E_ALIGN;
  if (!(a->x_layout == TG_LAYOUT_RM || a->x_layout == TG_LAYOUT_TC_A) || !(a->y_layout == TG_LAYOUT_RM || a->y_layout == TG_LAYOUT_TC_A)) return TG_E_LAYOUT;
  if ((a->x_layout || a->y_layout) && (!on_right || a->m % 16 != 0 || a->bias)) return TG_E_LAYOUT;
  if (a->bias_row_stride < 0 || (a->bias_row_stride && !a->bias) || (a->bias_row_stride & 3)) return TG_E_SHAPE;
  if (!(a->epilogue == TG_EPI_NONE || a->epilogue == TG_EPI_SWIGLU)) return TG_E_SHAPE;
  if (a->norm_weight && !aligned16(a->norm_weight)) return TG_E_ALIGN;
  // the fused stages exist in the TG_NUM_FAST pair-table kernels only (row-major operands)
  if ((a->norm_weight || a->epilogue) && (a->numerics == TG_NUM_REFERENCE || a->x_layout || a->y_layout)) return TG_E_FUSION;
  if (a->norm_weight && a->k % 2048 != 0) return TG_E_FUSION;
  if (a->epilogue == TG_EPI_SWIGLU && (!on_right || a->bias || a->wrows % 16 != 0)) return TG_E_FUSION;
  const int batch = a->batch > 1 ? a->batch : 1;
  if (batch > 1 && ((a->stride_x | a->stride_w | a->stride_lut) & 15)) return TG_E_ALIGN;
  if (batch > 1 && a->bias && (a->stride_bias & 7)) return TG_E_ALIGN;
  // the kernels address one problem's operands with 32-bit byte offsets
  if (a->m * a->k * 2 >= (int64_t)1 << 31 || a->wrows * a->k / 2 >= (int64_t)1 << 31 ||
      (a->k / a->group) * a->wrows * 4 >= (int64_t)1 << 31 || cdiv(a->m, 16) > 65535)
    return TG_E_SIZE;

  GemmParams p;
  p.x = (const char*)a->x;
  p.w = (const char*)a->w;
  p.qinfo = (const char*)a->qinfo;
  p.lut = (const char*)a->lut;
  p.y = (char*)a->y;
  p.m = (int32_t)a->m;
  p.wrows = (int32_t)a->wrows;
  p.k = (int32_t)a->k;
  p.ntiles = (int32_t)(a->wrows / rows_per_tile);
  p.ksuper = (int32_t)(a->k / (16 * I));
  p.gshift = g == 32 ? 5 : g == 64 ? 6 : g == 128 ? 7 : 8;
  p.ngroups = (int32_t)(a->k / g);
  p.qtype = a->qtype;
  p.dbg = 0;
  p.dry = dry != 0;
  p.numerics = a->numerics == TG_NUM_FAST_DOT2 ? (int)TG_NUM_FAST : a->numerics;
  p.dot2 = a->numerics == TG_NUM_FAST_DOT2;
  p.ws = (char*)a->workspace;
  p.ws_bytes = a->workspace ? a->workspace_bytes : 0;
  p.ws_query = dry == 2;
  p.ws_need = 0;
  p.x_tc = a->x_layout == TG_LAYOUT_TC_A;
  p.y_tc = a->y_layout == TG_LAYOUT_TC_A;
#ifdef TG_DEV
  {
    static const int env_dbg = getenv("TG_DBG") ? atoi(getenv("TG_DBG")) : 0;
    static const int env_var = getenv("TG_VARIANT") ? atoi(getenv("TG_VARIANT")) : 0;
    p.dbg = env_dbg;
    g_dbg_variant = env_var;
  }
#endif
  p.bias = (const char*)a->bias;
  p.stride_bias = batch > 1 ? a->stride_bias : 0;
  p.bias_row_stride = a->bias_row_stride;
  p.norm_w = (const char*)a->norm_weight;
  p.norm_eps = a->norm_eps;
  p.epilogue = a->epilogue;
  p.stride_x = batch > 1 ? a->stride_x : 0;
  p.stride_w = batch > 1 ? a->stride_w : 0;
  p.stride_qinfo = batch > 1 ? a->stride_qinfo : 0;
  p.stride_lut = batch > 1 ? a->stride_lut : 0;
  p.stride_y = batch > 1 ? a->stride_y : 0;

  DeviceScope ds(dry ? -1 : device);
  if (!dry && !ds.ok) return TG_E_DEVICE;
  hipStream_t st = (hipStream_t)stream;
  p.rowtiles = (int32_t)cdiv(a->wrows, 16);
  const int64_t coltiles = cdiv(a->m, 16);
  // packed words per lane-quad in the layout decide the in-register transpose
  const int canon = on_right ? (I == 2 ? CANON_NONE : I == 4 ? CANON_PAIR : CANON_QUAD)
                             : (I == 1 ? CANON_NONE : I == 2 ? CANON_PAIR : CANON_QUAD);
  // MANY activation rows (a prefill through the modules, a wide decode batch): the LDS-tiled MFMA GEMM that dequantises the weights once per
  // 128-row tile of m instead of once per 16 rows (w4_gemm_tile.cuh; the reference's weights bit for bit, so it serves both numerics
  // settings).  From 65 rows; ONE layer from 17 rows when the caller's workspace allows a split-K launch (tg_tile.hip).  4096^2 at m = 128 / 256 / 1024:
  // 17.9 / 25.5 / 52 us against 37.8 / 44.5 / 166 on the stream kernel and 55 / 102 / 427 in 16-row blocks.
  if (on_right && a->m >= TG_TILE_MIN_M_SPLIT) {
    const int trc = tgx::tile(a->dtype, I, a->qtype == TG_Q_MX4, p, batch, st);
    if (trc != TG_PAIR_NA) {
      if (ws_need) *ws_need = p.ws_need;
      return trc;
    }
    p.ws_need = 0;
  }
  // More than 16 activation rows in the default numerics (row-major operands, weights on the B side): the group-scaled kernels hold at
  // most one 16-row MFMA tile of activations, so the call is issued as ceil(m / 16) launches of up to 16 rows each on the caller's stream
  // (the reference's own grid walks the 16-row tiles of m the same way and re-reads the weights per tile, TinyGemmImpl.cuh:379-392).
  // Stacked 4096^2 layers at m = 17 ... 32: 5.2-5.3 us per layer on the reference-numerics stream kernel (22 % of the roofline for
  // ONE pass over the weights) against 3.0-3.3 here; one layer per graph node at m = 32 / 64: 15.3 / 35.5 us against 14 / 28.
  // Up to 64 rows: from 128 rows on the stream kernel's ONE launch (its 16-row tiles of m run concurrently and share the weights in
  // L2) wins -- one 4096^2 layer per graph node at m = 128 / 256 / 1024: 37.8 / 44.5 / 166 us against 55 / 102 / 427 in blocks
  // (profiles/r05_row_blocks_large_m.txt).
#ifndef TG_ROW_BLOCKS
#define TG_ROW_BLOCKS 1
#endif
#ifndef TG_ROW_BLOCKS_MAX_M
#define TG_ROW_BLOCKS_MAX_M 64
#endif
  if (TG_ROW_BLOCKS && on_right && a->m > 16 && a->m <= TG_ROW_BLOCKS_MAX_M && (p.numerics == TG_NUM_FAST || p.numerics == TG_NUM_FAST_MFMA) && !p.x_tc && !p.y_tc &&
      !p.norm_w && !p.epilogue) {
    // decided on a dry pass over the two block shapes of the call (16 rows, the ragged last block): both on a group-scaled kernel, or the
    // whole call stays on the path below
    auto block_plan = [&](int mb) {
      GemmParams q = p;
      q.m = mb;
      q.dry = true;
      q.ws_need = 0;
      return launch_w4(a->dtype, false, canon, a->qtype == TG_Q_MX4, q, 1, batch, st);
    };
    auto group_scaled = [](int rc) { return rc == TG_PLAN_PAIR || rc == TG_PLAN_PAIR_XR || rc == TG_PLAN_GEMV; };
    const int last = (int)(a->m % 16 ? a->m % 16 : 16);
    const int r16 = block_plan(16), rl = last == 16 ? r16 : block_plan(last);
    if (group_scaled(r16) && group_scaled(rl)) {
      int64_t need = 0;
      for (int64_t m0 = 0; m0 < a->m; m0 += 16) {
        GemmParams q = p;
        q.m = (int32_t)(a->m - m0 < 16 ? a->m - m0 : 16);
        q.x = p.x + m0 * a->k * 2;
        q.y = p.y + m0 * a->wrows * 2;
        if (q.bias && q.bias_row_stride) q.bias = p.bias + m0 * q.bias_row_stride * 2;
        q.ws_need = 0;
        const int rc = launch_w4(a->dtype, false, canon, a->qtype == TG_Q_MX4, q, 1, batch, st);
        if (rc < 0) return rc;
        if (rc == TG_PLAN_PAIR || rc == TG_PLAN_PAIR_XR) need = q.ws_need > need ? q.ws_need : need;
      }
      if (ws_need) *ws_need = need;
      return dry ? r16 : 0;
    }
    p.ws_need = 0;
  }
  const int rc = launch_w4(a->dtype, !on_right, canon, a->qtype == TG_Q_MX4, p, coltiles, batch, st);
  if (ws_need) *ws_need = (rc == TG_PLAN_PAIR || rc == TG_PLAN_PAIR_XR) ? p.ws_need : 0;
  return rc;
}

int tg_gemm_w4(const tg_w4_gemm* a, int device, tg_stream_t stream) { return gemm_w4_impl(a, device, stream, 0); }

int tg_gemm_w4_plan(const tg_w4_gemm* a, int device) { return gemm_w4_impl(a, device, nullptr, 1); }

int64_t tg_gemm_w4_workspace_bytes(const tg_w4_gemm* a) {
  int64_t need = 0;
  const int rc = gemm_w4_impl(a, -1, nullptr, 2, &need);
  return rc < 0 ? rc : need;
}

int tg_convert_to_Bint8(const int32_t* in, int64_t n, int64_t k, int I, int32_t* out, int device, tg_stream_t stream) {
  if (!in || !out) return TG_E_NULL;
  if (!(I == 1 || I == 2 || I == 4)) return TG_E_INNER_K;  // ConvertB.cu:428
  if (n <= 0 || k <= 0) return TG_E_SHAPE;
  if (k % (I * 16) != 0) return TG_E_K_DIV;                // ConvertB.cu:438
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t ksuper = k / (I * 16), total = cdiv(n, 8) * ksuper * 32 * I;
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (I == 1) hipLaunchKernelGGL(pack_Bint8_kernel<1>, dim3(blocks), dim3(256), 0, st, in, out, n, k, ksuper, total);
  else if (I == 2) hipLaunchKernelGGL(pack_Bint8_kernel<2>, dim3(blocks), dim3(256), 0, st, in, out, n, k, ksuper, total);
  else hipLaunchKernelGGL(pack_Bint8_kernel<4>, dim3(blocks), dim3(256), 0, st, in, out, n, k, ksuper, total);
  return launch_status();
}

int tg_convert_to_Aint8(const int32_t* in, int64_t m, int64_t k, int I, int32_t* out, int device, tg_stream_t stream) {
  if (!in || !out) return TG_E_NULL;
  if (!(I == 1 || I == 2)) return TG_E_INNER_K;  // ConvertA.cu:413
  if (m <= 0 || k <= 0) return TG_E_SHAPE;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t kouter = cdiv(cdiv(k, 16), I), total = cdiv(m, 16) * kouter * 32 * I * 2;
  const unsigned blocks = (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (I == 1) hipLaunchKernelGGL(pack_Aint8_kernel<1>, dim3(blocks), dim3(256), 0, st, in, out, m, k, kouter, total);
  else hipLaunchKernelGGL(pack_Aint8_kernel<2>, dim3(blocks), dim3(256), 0, st, in, out, m, k, kouter, total);
  return launch_status();
}

// dry: 0 launch, 2 report the workspace the fastest kernel wants (tg_gemm_w8_workspace_bytes)
static int gemm_w8_impl(const tg_w4_gemm* caller, int device, tg_stream_t stream, int dry, int64_t* ws_need) {
  tg_w4_gemm full;
  const int src = take_args(caller, &full);
  if (src != 0) return src;
  const tg_w4_gemm* a = &full;
  if (!a->x || !a->w || !a->qinfo || !a->y) return TG_E_NULL;
  if (a->qtype != TG_Q_INT8) return TG_E_QTYPE;
  if (!(a->dtype == TG_BF16 || a->dtype == TG_F16)) return TG_E_DTYPE;
  if (a->m <= 0 || a->wrows <= 0 || a->k <= 0 || a->m > INT32_MAX || a->wrows > INT32_MAX || a->k > INT32_MAX) return TG_E_SHAPE;
  const int I = a->inner_k_tiles;
  const bool on_right = a->w_on_right != 0;
  if (on_right ? !(I == 1 || I == 2 || I == 4) : !(I == 1 || I == 2)) return TG_E_INNER_K;  // TinyGemm_int8.cu:262, 286
  if (a->k % 32 != 0 || a->k % (16 * I) != 0) return TG_E_K_DIV;                            // TinyGemmImpl.cuh:370-376
  const int g = a->group;
  if (!(g == 32 || g == 64 || g == 128 || g == 256) || a->k % g != 0) return TG_E_GROUP;     // TinyGemm_int8.cu:293-301
  const int rows_per_tile = on_right ? 8 : 16;
  if (a->wrows % rows_per_tile != 0) return TG_E_SHAPE;
  // (x: 16-byte loads of the activation fragments / LDS-DMA of the tile flavour; rows are k * 2 bytes with k % 32 == 0)
  if ((reinterpret_cast<uintptr_t>(a->x) & 15u) || (a->batch > 1 && (a->stride_x & 15)) || (reinterpret_cast<uintptr_t>(a->w) & 3u) ||
      (reinterpret_cast<uintptr_t>(a->qinfo) & 3u) || (reinterpret_cast<uintptr_t>(a->y) & 7u))
    return TG_E_ALIGN;
  if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 7u)) return TG_E_ALIGN;
  if (a->reserved != 0) return TG_E_SHAPE;
  if (a->norm_weight || a->epilogue) return TG_E_FUSION;
  if (a->bias_row_stride < 0 || (a->bias_row_stride && !a->bias) || (a->bias_row_stride & 3)) return TG_E_SHAPE;
  if (a->m * a->k * 2 >= (int64_t)1 << 31 || a->wrows * a->k >= (int64_t)1 << 31 ||
      (a->k / a->group) * a->wrows * 4 >= (int64_t)1 << 31 || cdiv(a->m, 16) > 65535)
    return TG_E_SIZE;
  const int batch = a->batch > 1 ? a->batch : 1;
  if (batch > 1 && a->bias && (a->stride_bias & 7)) return TG_E_ALIGN;
  GemmParams p;
  p.x = (const char*)a->x; p.w = (const char*)a->w; p.qinfo = (const char*)a->qinfo; p.lut = nullptr; p.y = (char*)a->y;
  p.bias = (const char*)a->bias; p.stride_bias = batch > 1 ? a->stride_bias : 0; p.numerics = TG_NUM_REFERENCE;
  p.bias_row_stride = a->bias_row_stride; p.norm_w = nullptr; p.norm_eps = 0.f; p.epilogue = 0;
  p.m = (int32_t)a->m; p.wrows = (int32_t)a->wrows; p.k = (int32_t)a->k;
  p.ntiles = (int32_t)(a->wrows / rows_per_tile);
  p.ksuper = (int32_t)(a->k / (16 * I));
  p.gshift = g == 32 ? 5 : g == 64 ? 6 : g == 128 ? 7 : 8;
  p.ngroups = (int32_t)(a->k / g);
  p.qtype = a->qtype; p.dbg = 0; p.dry = dry != 0;
  p.stride_x = batch > 1 ? a->stride_x : 0; p.stride_w = batch > 1 ? a->stride_w : 0;
  p.stride_qinfo = batch > 1 ? a->stride_qinfo : 0; p.stride_lut = 0; p.stride_y = batch > 1 ? a->stride_y : 0;
  if (a->workspace && (!aligned16(a->workspace) || a->workspace_bytes < 0)) return TG_E_ALIGN;
  p.ws = (char*)a->workspace; p.ws_bytes = a->workspace ? a->workspace_bytes : 0; p.ws_need = 0; p.ws_query = dry == 2;
  p.x_tc = p.y_tc = 0;
  DeviceScope ds(dry ? -1 : device);
  if (!dry && !ds.ok) return TG_E_DEVICE;
  hipStream_t st = (hipStream_t)stream;
  p.rowtiles = (int32_t)cdiv(a->wrows, 16);
  const int64_t coltiles = cdiv(a->m, 16);
#ifdef TG_DEV_MIN
  (void)coltiles; (void)st;
  return TG_E_SHAPE;
#else
  // many activation rows: the LDS-tiled MFMA GEMM's int8 flavour (w4_gemm_tile.cuh; the same weights bit for bit), split-K with the caller's workspace
  {
    const int trc = tgx::tile_w8(a->dtype, on_right, I, p, batch, st);
    if (ws_need) *ws_need = trc == TG_PAIR_NA ? 0 : p.ws_need;
    if (trc != TG_PAIR_NA) return trc == TG_PLAN_TILE ? 0 : trc;
    if (dry) return 0;
  }
#define TG_W8(DTT)                                                                                    \
  do {                                                                                                \
    if (on_right) {                                                                                   \
      if (I == 1) return launch_w8<DTT, false, 1>(p, coltiles, batch, st);                           \
      if (I == 2) return launch_w8<DTT, false, 2>(p, coltiles, batch, st);                           \
      return launch_w8<DTT, false, 4>(p, coltiles, batch, st);                                       \
    }                                                                                                 \
    if (I == 1) return launch_w8<DTT, true, 1>(p, coltiles, batch, st);                              \
    return launch_w8<DTT, true, 2>(p, coltiles, batch, st);                                          \
  } while (0)
  if (a->dtype == TG_BF16) TG_W8(BF16);
  TG_W8(F16);
#undef TG_W8
#endif
}

int tg_gemm_w8(const tg_w4_gemm* a, int device, tg_stream_t stream) { return gemm_w8_impl(a, device, stream, 0, nullptr); }

int64_t tg_gemm_w8_workspace_bytes(const tg_w4_gemm* a) {
  int64_t need = 0;
  const int rc = gemm_w8_impl(a, -1, nullptr, 2, &need);
  return rc < 0 ? rc : need;
}

int tg_gemm_f16(const void* x, const void* w, void* y, int64_t m, int64_t wrows, int64_t k, int dtype,
                int w_on_right, int I, int device, tg_stream_t stream) {
  if (!x || !w || !y) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (m <= 0 || wrows <= 0 || k <= 0 || m > INT32_MAX || wrows > INT32_MAX || k > INT32_MAX) return TG_E_SHAPE;
  if (w_on_right ? !(I == 1 || I == 2) : I != 1) return TG_E_INNER_K;
  if (k % 32 != 0) return TG_E_K_DIV;  // TinyGemmImpl.cuh:376
  if (wrows % (w_on_right ? 8 : 16) != 0) return TG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w)) return TG_E_ALIGN;
  F16GemmParams p;
  p.x = (const char*)x;
  p.w = (const char*)w;
  p.y = (char*)y;
  p.m = (int32_t)m;
  p.wrows = (int32_t)wrows;
  p.k = (int32_t)k;
  p.inner = I;
  p.ktiles_padded = (int32_t)(cdiv(k, 16 * I) * I);
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(wrows, 16), (unsigned)cdiv(m, 16));
  constexpr int WAVES = 8;
  if (dtype == TG_BF16) {
    if (w_on_right && I == 2) hipLaunchKernelGGL((f16_gemm_kernel<BF16, false, WAVES, 2>), grid, dim3(WAVES * 64), 0, st, p);
    else if (w_on_right) hipLaunchKernelGGL((f16_gemm_kernel<BF16, false, WAVES, 1>), grid, dim3(WAVES * 64), 0, st, p);
    else hipLaunchKernelGGL((f16_gemm_kernel<BF16, true, WAVES, 1>), grid, dim3(WAVES * 64), 0, st, p);
  } else {
    if (w_on_right && I == 2) hipLaunchKernelGGL((f16_gemm_kernel<F16, false, WAVES, 2>), grid, dim3(WAVES * 64), 0, st, p);
    else if (w_on_right) hipLaunchKernelGGL((f16_gemm_kernel<F16, false, WAVES, 1>), grid, dim3(WAVES * 64), 0, st, p);
    else hipLaunchKernelGGL((f16_gemm_kernel<F16, true, WAVES, 1>), grid, dim3(WAVES * 64), 0, st, p);
  }
  return launch_status();
}

}  // extern "C"

#include "decode_glue.cuh"
#include "peer_gather.cuh"
