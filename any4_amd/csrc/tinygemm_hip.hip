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

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include <utility>

#include "../../include/tinygemm_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- 16-bit float traits ---------------------------------------------------------------------
struct BF16 {
  static __device__ __forceinline__ float to_f32(uint16_t h) { return u2f(((uint32_t)h) << 16); }
  static __device__ __forceinline__ float lo_f32(uint32_t pair) { return u2f(pair << 16); }
  static __device__ __forceinline__ float hi_f32(uint32_t pair) { return u2f(pair & 0xffff0000u); }
  // round-to-nearest-even pack; lowers to v_cvt_pk_bf16_f32 on gfx950
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static __device__ __forceinline__ uint16_t from_f32(float a) { return (uint16_t)(pack2(a, 0.f) & 0xffffu); }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct F16 {
  static __device__ __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ float lo_f32(uint32_t pair) { return to_f32((uint16_t)(pair & 0xffffu)); }
  static __device__ __forceinline__ float hi_f32(uint32_t pair) { return to_f32((uint16_t)(pair >> 16)); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ uint16_t from_f32(float a) { return (uint16_t)(pack2(a, 0.f) & 0xffffu); }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// fp4-e2m1 values in code order (reference FloatDefs.cuh:18-34)
__device__ const float kMX4Values[16] = {0.0f,  0.5f,  1.0f,  1.5f,  2.0f,  3.0f,  4.0f,  6.0f,
                                         -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f};

// Output store of four consecutive weight rows of one activation row.  With a bias the sum is first rounded to 16 bits and
// the bias added in a second rounded step: bit-identical to the reference module's separate `y + bias` (modules.py:221-222).
template <typename DT>
__device__ __forceinline__ void store_rows4(char* yb, const char* bias, int64_t elem, int rowg, f32x4 acc) {
  u32x2 o = {DT::pack2(acc[0], acc[1]), DT::pack2(acc[2], acc[3])};
  if (bias) {
    const u32x2 bv = *reinterpret_cast<const u32x2*>(bias + (int64_t)rowg * 2);
    o[0] = DT::pack2(DT::lo_f32(o[0]) + DT::lo_f32(bv[0]), DT::hi_f32(o[0]) + DT::hi_f32(bv[0]));
    o[1] = DT::pack2(DT::lo_f32(o[1]) + DT::lo_f32(bv[1]), DT::hi_f32(o[1]) + DT::hi_f32(bv[1]));
  }
  *reinterpret_cast<u32x2*>(yb + elem * 2) = o;
}

#include "w4_gemm.cuh"
#include "w4_gemm_stream.cuh"
#include "w4_gemm_pair.cuh"
#include "w4_gemm_pair16.cuh"
#include "w4_gemm_xr.cuh"
#include "w8_gemm.cuh"

#ifndef STREAM_MINW
#define STREAM_MINW 4
#endif

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

template <typename DT, bool LAYOUT_A, int WAVES>
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
  for (int s = wave; s < nsteps_total; s += WAVES) {
    u32x4 a = {0, 0, 0, 0}, xv = {0, 0, 0, 0};
    if (row_ok) {
      if constexpr (LAYOUT_A) {
        // [mT][kT][32][8 halfs] = 4 dwords per lane slot: (m0;k0,k1) (m1;k0,k1) (m0;k8,k9) (m1;k8,k9)
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(wd + (((int64_t)rt * p.ktiles_padded + 2 * s) * 32 + t) * 4);
        const u32x4 v1 = *reinterpret_cast<const u32x4*>(wd + (((int64_t)rt * p.ktiles_padded + 2 * s + 1) * 32 + t) * 4);
        const int h = i >> 3;
        a = u32x4{h ? v0[1] : v0[0], h ? v0[3] : v0[2], h ? v1[1] : v1[0], h ? v1[3] : v1[2]};
      } else {
        // [nT][kT/I][32][4 I halfs]: per k-tile the dwords (k0,k1) (k8,k9)
        const int tile = 2 * rt + (i >> 3);
        if (p.inner == 2) {
          a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wd + (((int64_t)tile * (p.ktiles_padded / 2) + s) * 32 + t) * 4));
        } else {
          const u32x2 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wd + (((int64_t)tile * p.ktiles_padded + 2 * s) * 32 + t) * 2));
          const u32x2 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(wd + (((int64_t)tile * p.ktiles_padded + 2 * s + 1) * 32 + t) * 2));
          a = u32x4{v0[0], v0[1], v1[0], v1[1]};
        }
      }
    }
    if (xcol) {
      const char* xp = p.x + ((int64_t)xrow * p.k + 32 * s) * 2 + 4 * Q;
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] = *reinterpret_cast<const uint32_t*>(xp + 16 * e);
    }
    acc = DT::mfma(a, xv, acc);
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

// ---- host side ---------------------------------------------------------------------------------

struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    if (prev == device) prev = -1;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Kernels that address LDS from offset 0 (lookup tables at the start of the dynamic region) and / or need more than 64 KiB
// of dynamic LDS: once per device, check that the kernel has no static LDS (the dynamic region then starts at 0) and raise
// its dynamic-LDS limit.  State = one write-once bit per device and kernel; racing threads repeat the same idempotent calls.
template <auto KERN>
int prepare_lds_kernel() {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return TG_E_DEVICE;
  if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1u)) return 0;
  hipFuncAttributes fa;
  hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(KERN));
  if (e != hipSuccess) return (int)e;
  if (fa.sharedSizeBytes != 0) return TG_E_INTERNAL;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
  return 0;
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

// ---- streaming kernel launch ---------------------------------------------------------------------
// LDS per workgroup: lookup tables (4 KiB per wave and row set) + two X slabs (+ split-K tiles).
template <bool LAYOUT_A>
inline unsigned stream_lds_bytes(int sw, int mrows, int sk, bool privx = false) {
  const unsigned nr = 1u;  // lookup tables per wave
  const unsigned unit = LAYOUT_A ? 64u : 128u;
  // shared slab: rows of all k-slices; private slabs: one per wave with the rows of its own slice; + the all-zero row
  const unsigned slab = (unsigned)(mrows * 4 * (privx ? 1 : sk) + 1) * (unit * 2u + 16u);
  return (unsigned)sw * nr * 4096u + 2u * slab * (privx ? (unsigned)sw : 1u) + (sk > 1 ? (unsigned)sw * nr * 1024u : 0u);
}

template <typename DT, bool LAYOUT_A, int WPL, bool QMX, int SW, bool privx = (SW == 1)>
int launch_stream_sw(StreamParams& sp, int sk, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int UNIT = LAYOUT_A ? 64 : 128;
  constexpr unsigned NR = 1u;
  const int nunits = (sp.k + UNIT - 1) / UNIT;
  const int upg = (1 << sp.gshift) / UNIT;  // units per quantisation group (>= 1)
  const int mrows = sp.m < 16 ? sp.m : 16;
  int nu = (nunits + 4 * sk - 1) / (4 * sk);
  nu = (nu + upg - 1) / upg * upg;
  sp.splitk = sk;
  sp.sk_shift = 0;
  while ((1 << sp.sk_shift) < sk) ++sp.sk_shift;
  sp.units_per_lane = nu;
  sp.upg_mask = upg - 1;
  const int xrows = mrows * 4 * (privx ? 1 : sk);
  sp.xslab_bytes = (xrows + 1) * (UNIT * 2 + 16);
  sp.red_off = (int32_t)(SW * NR * 4096u + 2u * (unsigned)sp.xslab_bytes * (privx ? SW : 1));
  const int pieces = xrows * (UNIT * 2 / 16);
  // privx: every wave stages its own X slab (no barrier in the main loop)
  const int nstage = privx ? 64 : SW * 64;
  const int xl = pieces <= nstage ? 1 : (pieces <= 2 * nstage ? 2 : 4);
  const unsigned lds = stream_lds_bytes<LAYOUT_A>(SW, mrows, sk, privx);
  const int tpb = SW / sk;
  dim3 grid((unsigned)((sp.rowtiles + tpb - 1) / tpb), (unsigned)coltiles, (unsigned)batch);
#define TG_LAUNCH_STREAM(XL)                                                                              \
  do {                                                                                                    \
    constexpr auto kern = w4_gemm_stream_kernel<DT, LAYOUT_A, WPL, QMX, SW, STREAM_MINW, XL, privx>;      \
    if (sp.dry) return TG_PLAN_STREAM;                                                                    \
    const int prc = prepare_lds_kernel<kern>();                                                           \
    if (prc != 0) return prc;                                                                             \
    hipLaunchKernelGGL(kern, grid, dim3(SW * 64), lds, st, sp);                                           \
  } while (0)
  if constexpr (privx && SW > 1) {
    if (xl != 1) return TG_E_SHAPE;  // private slabs with split-K are only instantiated for one piece per lane (m = 1)
    TG_LAUNCH_STREAM(1);
  } else if constexpr (SW == 1) {
    if (xl == 1) TG_LAUNCH_STREAM(1);
    else TG_LAUNCH_STREAM(2);
  } else {
    if (xl == 1) TG_LAUNCH_STREAM(1);
    else if (xl == 2) TG_LAUNCH_STREAM(2);
    else TG_LAUNCH_STREAM(4);
  }
#undef TG_LAUNCH_STREAM
  return launch_status();
}

// Resident-X launch: 16-wave workgroups, the whole [mrows][k] activation block staged once per workgroup.
template <typename DT, bool LAYOUT_A, int WPL, bool QMX>
int launch_stream_xres(StreamParams& sp, int64_t coltiles, int64_t batch, unsigned lds, hipStream_t st) {
  // one workgroup walks up to 4 consecutive groups of 16 tiles of its layer (X staged once, tile-granularity tail
  // amortised) as long as that leaves at least two workgroups per CU
  int tpw = 4;
  while (tpw > 1 && ((sp.rowtiles + 16 * tpw - 1) / (16 * tpw)) * coltiles * batch < 512) tpw >>= 1;
  sp.tiles_per_wave = tpw;
  constexpr auto kern = w4_gemm_stream_kernel<DT, LAYOUT_A, WPL, QMX, 16, STREAM_MINW, 1, false, 0, true>;
  if (sp.dry) return TG_PLAN_STREAM;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc;
  dim3 grid((unsigned)((sp.rowtiles + 16 * tpw - 1) / (16 * tpw)), (unsigned)coltiles, (unsigned)batch);
  hipLaunchKernelGGL(kern, grid, dim3(16 * 64), lds, st, sp);
  return launch_status();
}

template <typename DT, bool LAYOUT_A, int WPL, bool QMX>
int launch_stream(const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int UNIT = LAYOUT_A ? 64 : 128;
  constexpr int RPW = 16;  // weight rows per wave
  StreamParams sp;
  sp.x = p.x; sp.w = p.w; sp.qinfo = p.qinfo; sp.lut = p.lut; sp.y = p.y;
  sp.m = p.m; sp.wrows = p.wrows; sp.k = p.k; sp.ntiles = p.ntiles; sp.ksuper = p.ksuper;
  sp.gshift = p.gshift; sp.ngroups = p.ngroups; sp.qtype = p.qtype;
  sp.rowtiles = (p.wrows + RPW - 1) / RPW;
  sp.tiles_per_wave = 1;
  sp.stride_x = p.stride_x; sp.stride_w = p.stride_w; sp.stride_qinfo = p.stride_qinfo;
  sp.stride_lut = p.stride_lut; sp.stride_y = p.stride_y;
  sp.bias = p.bias; sp.stride_bias = p.stride_bias; sp.bias_row_stride = p.bias_row_stride; sp.dry = p.dry;
  const int mrows = p.m < 16 ? p.m : 16;
  const int nunits = (p.k + UNIT - 1) / UNIT;
  const int upg = (1 << p.gshift) / UNIT;
  // split-K: aim for at least two rounds of 16 waves on every CU; the X slab limits act rows * splitk to 16
  const int64_t wave_tiles = (int64_t)sp.rowtiles * coltiles * batch;
  int sk = 1;
  // (m = 1, private slabs: one round of 16 waves per CU is enough -- measured on the Llama-3-8B shapes)
  const int64_t want = mrows == 1 ? 256 * 16 : 2 * 256 * 16;
  while (sk < 8 && wave_tiles * sk < want && nunits >= 8 * sk * upg && mrows * sk * 2 <= 16) sk *= 2;
#ifdef TG_DEV
  static const int sk_env = getenv("TG_SK") ? atoi(getenv("TG_SK")) : 0;  // developer override
  if (sk_env > 0) sk = sk_env;
#endif
  // m == 1: every wave stages its own X slab (no barrier in the main loop); a workgroup is the sk waves of one tile
#ifdef TG_DEV
  static const int xres_env = getenv("TG_XRES") ? atoi(getenv("TG_XRES")) : 1;  // developer knob: 0 off, 2 also for m = 1
#else
  constexpr int xres_env = 1;
#endif
  if (mrows == 1 && xres_env != 2) {
    switch (sk) {
      case 1: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 1>(sp, 1, coltiles, batch, st);
      case 2: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 2, true>(sp, 2, coltiles, batch, st);
      case 4: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 4, true>(sp, 4, coltiles, batch, st);
      default: return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 8, true>(sp, 8, coltiles, batch, st);
    }
  }
  // m >= 2, one tile per wave: keep the whole activation block resident in LDS when it fits next to 16 lookup
  // tables (m = 8 at k = 4096 does: 66 KiB + 64 KiB) -- one barrier per workgroup instead of one per unit
  // (measured: wins for m >= 8 at k = 4096 and for m >= 2 at k = 8192; the 16-wave workgroup costs ~15 % in tile-granularity
  //  tail against 4-wave workgroups, which the small slabs of m <= 4 at k = 4096 do not pay back)
  if (sk == 1 && xres_env && (mrows * UNIT >= 1024 || sp.k >= 8192 || xres_env == 2)) {
    const int nu = (int)(((nunits + 3) / 4 + upg - 1) / upg * upg);
    const unsigned xrow = (unsigned)(nu * UNIT * 2 + 16);
    const unsigned lds = 16u * 4096u + (unsigned)(mrows * 4) * xrow + (unsigned)(UNIT * 2 + 16);
    if (lds <= 160u * 1024u) {
      sp.splitk = 1; sp.sk_shift = 0; sp.units_per_lane = nu; sp.upg_mask = upg - 1;
      sp.xslab_bytes = (int32_t)xrow; sp.red_off = 0;
      return launch_stream_xres<DT, LAYOUT_A, WPL, QMX>(sp, coltiles, batch, lds, st);
    }
  }
  // otherwise 4-wave workgroups while their LDS footprint lets 16 waves live on a CU and the X slab is small;
  // X slabs of 8 KiB or more per unit (Bint4: m >= 8, Aint4: m = 16): 8-wave workgroups halve the staging work per wave
  const int sk4 = sk < 4 ? sk : 4;
  if (mrows * UNIT < 1024 && 160u * 1024u / stream_lds_bytes<LAYOUT_A>(4, mrows, sk4) >= 4)
    return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 4>(sp, sk4, coltiles, batch, st);
  return launch_stream_sw<DT, LAYOUT_A, WPL, QMX, 8>(sp, sk, coltiles, batch, st);
}


// ---- pair-table kernel launch (w4_gemm_pair.cuh) ---------------------------------------------------
// Returns TG_PAIR_NA when the shape does not fit this kernel's LDS plan (the caller then takes another kernel).
enum { TG_PAIR_NA = -100 };

// Tuning constants of the pair-table launches, each with the measurement that set it.  The shipped library
// always uses these values; only developer builds (-DTG_DEV / -DTG_DEV_MIN, dev/build_variant.sh) may override one with -D<NAME>=<v>.
#if !defined(TG_DEV) && !defined(TG_DEV_MIN)
#if defined(TG_PAIR_R) || defined(TG_PAIR_ABL) || defined(TG_PAIR_MR1) || defined(TG_PAIR_NSG2) || defined(TG_PAIR_MR1_GPS) || defined(TG_PAIR_RA) ||   \
    defined(TG_PAIR_RA1) || defined(TG_PAIR_RB16) || defined(TG_B16_CHUNK) || defined(TG_PAIR_MIN_ITEMS) || defined(TG_XG_CHUNK) || defined(TG_PAIR_WGS) || \
    defined(TG_PAIR_NSG2_M1) || defined(TG_PAIR_FORCE_XG) || defined(TG_XR_MIN_M) || defined(TG_XR_R) || defined(TG_XR_R8K) || defined(TG_XR_RMX)
#error "the TG_PAIR_* / TG_XG_* / TG_B16_* tuning constants can only be overridden in developer builds (-DTG_DEV or -DTG_DEV_MIN)"
#endif
#endif
#ifndef TG_PAIR_R
#define TG_PAIR_R 2            // super-tiles a wave keeps in flight (2, 3, 4 measured equal; 5 spills)
#endif
#ifndef TG_PAIR_ABL
#define TG_PAIR_ABL 0          // ablation stub of w4_gemm_pair.cuh (its header lists them)
#endif
#ifndef TG_PAIR_MR1
#define TG_PAIR_MR1 1          // 1: m = 1 runs the one-register specialisation (+2-3 %), 4: the general m <= 8 kernel
#endif
#ifndef TG_PAIR_NSG2
#define TG_PAIR_NSG2 1         // group boundaries at fixed places of the unrolled round when a group is one round of the ring
#endif
#ifndef TG_PAIR_NSG2_M1
#define TG_PAIR_NSG2_M1 1      // ... also in the m = 1 specialisation (74.8 -> 75.8 % once its group update was spelled out)
#endif
#ifndef TG_PAIR_MR1_GPS
#define TG_PAIR_MR1_GPS 1      // the m = 1 specialisation is used up to this many groups per super-tile (it spills beyond)
#endif
#ifndef TG_PAIR_RA
#define TG_PAIR_RA 2           // ring depth of the A-side kernels (4 / 6 / 8 measured 8.2 / 21 / 40 us against 6.8)
#endif
#ifndef TG_PAIR_RA1
#define TG_PAIR_RA1 1          // ... with several groups per super-tile
#endif
#ifndef TG_PAIR_RB16
#define TG_PAIR_RB16 2         // ring depth of the 16x16x32 kernels for Bint4 weights, m = 9 ... 16 (3 / 4: 41 % against 44 %)
#endif
#ifndef TG_B16_CHUNK
#define TG_B16_CHUNK 4         // consecutive 32-row work items per workgroup visit of those kernels (1 / 4 / 8 within 1 %)
#endif
#ifndef TG_PAIR_MIN_ITEMS
#define TG_PAIR_MIN_ITEMS 192  // fewer work items: the launch is latency-bound, w4_gemm_pair16_kernel / the reference kernels take
                               // it (measured per hipGraph node, one layer, m = 1: 14336 x 4096 = 224 items 12.3 us here against
                               // 18.3 us on pair16 and 13.8 us on the stream kernel; 6144 x 4096 = 96 items 10.7 against 9.9 / 8.2)
#endif
#ifndef TG_XG_CHUNK
#define TG_XG_CHUNK 4          // consecutive work items per workgroup visit in the workspace variant of Bint4 weights (1: plain
                               // round-robin): m = 8: 4096^2 66.1 -> 66.5 %, 8192^2 67.3 -> 68.6 %; Aint4 weights keep 1
#endif
#ifndef TG_XR_MIN_M
#define TG_XR_MIN_M 2          // activation rows from which the register-resident-activation kernel (w4_gemm_xr.cuh) takes stacked launches
                               // (same-box A/B against the kernels it replaces, 4096^2: m = 2 71.9 vs 69.7 %, 4: 68.9 vs 66.9, 8: 66.2 vs 62.7,
                               //  16: 65.2 vs 46.0; m = 1 stays on the 32x32x16 kernel, 77 %)
#endif
#ifndef TG_XR_RMX
#define TG_XR_RMX 8            // ... for mx4 (no lookups: latency-bound; the whole slice in flight: m = 16 75.4 -> 79.0 %, m = 2 81.7 -> 85.3 %)
#endif
#ifndef TG_XR_R8K
#define TG_XR_R8K 2            // ... at k = 8192 (128 registers of activations per lane: 4 in flight spill 25)
#endif
#ifndef TG_XR_R
#define TG_XR_R 4              // super-tiles a wave of that kernel keeps in flight
#endif
#ifndef TG_PAIR_WGS
#define TG_PAIR_WGS 512        // persistent workgroups: two per CU (768 / 1024: +4 % / +1 % time)
#endif
template <typename DT, int I, int GPS, int MR, bool QMX, int NSG, bool XG = false, int LA = 0, bool NORM = false>
int launch_pair_k(PairParams& pp, unsigned lds, hipStream_t st) {
#ifdef TG_DEV_MIN  // developer builds: only the headline instantiation (fast A/B builds)
#ifndef TG_DEV_GPS
#define TG_DEV_GPS 1
#endif
#ifndef TG_DEV_QMX
#define TG_DEV_QMX false
#endif
#ifndef TG_DEV_MR
#define TG_DEV_MR TG_PAIR_MR1
#endif
#ifndef TG_DEV_LA
#define TG_DEV_LA 0
#endif
  if constexpr (!(std::is_same<DT, BF16>::value && I == 4 && GPS == TG_DEV_GPS && MR == TG_DEV_MR && QMX == TG_DEV_QMX && NSG == TG_DEV_MIN && LA == TG_DEV_LA && !NORM)) return TG_PAIR_NA;
  else {
#endif
  if constexpr (QMX && !std::is_same<DT, BF16>::value) return TG_E_DTYPE;  // mx4 is bf16-only (TinyGemm_int4.cu:758)
  else {
  // several groups per super-tile (group 32 / 64 with wide super-tiles): more per-slot state, one slot in flight fits the
  // 128-VGPR budget without spills (ring depth measured irrelevant between 2 and 4)
  // (mx4 on the 32x32x16 tiles converts its weights in registers and has no per-group state in the slots: the usual depth)
  constexpr int RING = (QMX && LA == 0) ? TG_PAIR_R : (MR == 1 && NSG == 4 && LA == 0) ? 4 : GPS > 1 ? (LA ? TG_PAIR_RA1 : 1) : LA == 1 ? TG_PAIR_RA : LA == 2 ? TG_PAIR_RB16 : TG_PAIR_R;
  constexpr auto kern = w4_gemm_pair_kernel<DT, I, GPS, MR, QMX, RING, NSG, TG_PAIR_ABL, XG, LA, NORM>;
  if (pp.dry) return TG_PLAN_PAIR;
  const int prc = prepare_lds_kernel<kern>();
  if (prc != 0) return prc;
  const unsigned wgs = (unsigned)(pp.items < TG_PAIR_WGS ? pp.items : TG_PAIR_WGS);
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(512), lds, st, pp);
  return launch_status();
  }
#ifdef TG_DEV_MIN
  }
#endif
}

// The activation block of one pass does not fit next to the table (m = 8 at k = 4096, m = 1 at k >= 8192): the XG variant
// takes the activations pre-arranged from a caller-provided workspace (w4_xprep_kernel, one small launch in front).
// Workspace = [batch][m k 2 bytes] arranged activations, then [batch][passes][groups][xs_rows] f32 sums.
template <typename DT>
int launch_xprep(const PairParams& pp, int I, int ma, int64_t batch, hipStream_t st, int la = 0) {
  XPrepParams xq;
  xq.la = la;
  xq.x = pp.x; xq.xp = const_cast<char*>(pp.xp); xq.xsum = const_cast<char*>(pp.xsum);
  xq.x_tc = pp.x_tc;
  xq.m = pp.m; xq.k = pp.k; xq.ma = ma; xq.cps = I / 2; xq.gshift = pp.gshift; xq.gch_mask = pp.gch_mask;
  xq.ngroups = pp.ngroups; xq.xs_rows = pp.xs_rows;
  xq.stride_x = pp.stride_x; xq.stride_xp = pp.stride_xp; xq.stride_xsum = pp.stride_xsum;
  const int64_t chunks = (int64_t)pp.m * (pp.k / 32);
  hipLaunchKernelGGL(w4_xprep_kernel<DT>, dim3((unsigned)cdiv(chunks, 256), (unsigned)batch), dim3(256), 0, st, xq);
  return launch_status();
}

// m = 1 has its own specialisation (one accumulator register finalised per group, taken as a running difference) -- except
// with several groups per super-tile, where the general kernel's zero-C group starts compile without spills; `norm`: the
// instantiations with LlamaRMSNorm fused into the activation staging (staged activations, m <= 8, not mx4)
template <typename DT, int I, int GPS, bool QMX, int NSG>
int launch_pair_m(PairParams& pp, unsigned lds, hipStream_t st, bool xg, int m, int mregs, bool norm) {
  const bool m1 = m == 1 && TG_PAIR_MR1 == 1 && (QMX || GPS <= TG_PAIR_MR1_GPS);  // (mx4: no per-group state, the specialisation fits at any GPS)
  if (xg) return m1 ? launch_pair_k<DT, I, GPS, 1, QMX, NSG, true>(pp, lds, st) : launch_pair_k<DT, I, GPS, 4, QMX, NSG, true>(pp, lds, st);
  if (norm) {
    if constexpr (QMX) return TG_PAIR_NA;
    else {
      if (mregs != 4) return TG_PAIR_NA;
      return m1 ? launch_pair_k<DT, I, GPS, 1, false, NSG, false, false, true>(pp, lds, st)
                : launch_pair_k<DT, I, GPS, 4, false, NSG, false, false, true>(pp, lds, st);
    }
  }
  if (m1) return launch_pair_k<DT, I, GPS, 1, QMX, NSG>(pp, lds, st);
  return mregs == 4 ? launch_pair_k<DT, I, GPS, 4, QMX, NSG>(pp, lds, st) : launch_pair_k<DT, I, GPS, 16, QMX, NSG>(pp, lds, st);
}

template <typename DT, int I, bool QMX>
int launch_pair(GemmParams& p, int64_t batch, hipStream_t st) {
  constexpr int RW = 64;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  const int mregs = p.m <= 8 ? 4 : 16;  // accumulator registers of a row set (8 or 32 activation rows per pass)
  const int ma = 2 * mregs;
  PairParams pp;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;  // super-tiles per group
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  const int mrows = p.m < ma ? p.m : ma;
  pp.rused = mrows < 4 ? mrows : mregs;
  pp.xs_rows = mrows <= 4 ? 4 : ma;
  pp.red_lanes = mrows <= 4 ? 32 : 64;
  pp.x_pitch = p.k * 2 + 16;
  pp.lds_x = QMX ? 0 : 65536;  // mx4 converts its weights in registers (v_cvt_scalef32_pk_bf16_fp4): no table, the LDS starts with the activations
  pp.lds_xs = (pp.lds_x + mrows * pp.x_pitch + 32 * I + 15) & ~15;  // staged rows + a zero piece of one super-tile
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
  pp.red_alias = !QMX && mrows > 4;  // 16 KiB and more of partial sums: reuse the table's LDS instead
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  // mx4: exponent blocks of 16 bytes per row, read at 4-byte alignment (w4_gemm_pair.cuh, e_request)
  // (a slice that starts off a 4-byte boundary loses up to 3 bytes of its one block)
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  // fused RMSNorm: done in the workgroup's own staging of the whole activation block (its partial sums borrow the activation-sum
  // area, which mx4 does not have); the workspace variant would need it in the pre-pass
  if (p.norm_w && (QMX || lds > 80u * 1024u || p.m > ma)) return TG_PAIR_NA;
  bool xg = false;
#ifdef TG_PAIR_FORCE_XG  // developer A/B: the workspace variant also where the staged plan fits
  if (mregs == 4 && p.m <= ma) lds = 1u << 30;
#endif
  if (lds > 80u * 1024u) {  // two workgroups per CU
    // XG: every wave keeps one super-tile of the pass's activations (<= 8 rows) in a private buffer
    if (mregs != 4 || p.m > ma) return TG_PAIR_NA;
    pp.xw_pitch = 32 * I + 16;
    pp.xw_bytes = (I == 2 ? 16 : 8) * pp.xw_pitch;  // a row for every 2 I lanes of the wave's (unmasked) store
    pp.lds_xs = (pp.lds_x + 8 * pp.xw_bytes + 32 * I + 15) & ~15;  // 8 buffers + the zero piece
    pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
    lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
    if (lds > 80u * 1024u) pp.red_alias = 1;
    if (QMX && pp.red_alias) return TG_PAIR_NA;  // (no table to put the partial sums over; cannot happen: 8 one-KiB buffers + 16 KiB)
    if (pp.red_alias) {
      lds = (unsigned)pp.lds_red;
      pp.lds_red = 0;
    }
    if (lds > 80u * 1024u) return TG_PAIR_NA;
    pp.stride_xp = (int64_t)p.m * p.k * 2;
    pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
    const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
    p.ws_need = need;
    if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
    pp.xp = p.ws;
    pp.xsum = p.ws + batch * pp.stride_xp;
    xg = true;
  }
  pp.rblocks = (p.wrows + RW - 1) / RW;
  pp.cblocks = (p.m + ma - 1) / ma;
  const int64_t items = (int64_t)pp.rblocks * pp.cblocks * batch;
  if (items > INT32_MAX) return TG_PAIR_NA;
  // The kernel's unit of work is a 64-row block over the whole k (8 waves): a launch needs about one item per workgroup slot
  // (2 per CU) to fill the chip.  Smaller launches (one 4096-row layer = 64 items) are latency-bound and stay on the
  // split-K kernels, which spread one 16-row tile over up to 16 waves.
  if (items < TG_PAIR_MIN_ITEMS) { p.ws_need = 0; return TG_PAIR_NA; }
  pp.items = (int32_t)items;
  // XG item dealing: chunks of consecutive items once every workgroup still gets several chunks
  pp.chunk = items >= (int64_t)TG_PAIR_WGS * TG_XG_CHUNK * 4 ? TG_XG_CHUNK : 1;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = p.norm_w; pp.norm_eps = p.norm_eps; pp.epilogue = p.epilogue;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  if (xg && !p.dry) {
    const int rc = launch_xprep<DT>(pp, I, ma, batch, st);
    if (rc != 0) return rc;
  }
#define TG_PAIR_M(GPS_, NSG_) launch_pair_m<DT, I, GPS_, QMX, NSG_>(pp, lds, st, xg, p.m, mregs, p.norm_w != nullptr)
  if (gps == 1) {
    // group boundaries at fixed places of the unrolled round when a group is one super-tile or one whole round
    // (the m = 1 specialisation too since its group update is spelled out instruction by instruction: before that, fixed
    //  boundaries made the compiler scatter its accumulator chain over several register tuples and spill)
    // (a group of ONE super-tile, g = 64 at I = 4, also keeps the run-time test: its fixed-boundary build spills 27 registers,
    //  m = 8 50 % against 59 %)
    const bool fixed = TG_PAIR_NSG2 && (TG_PAIR_NSG2_M1 || !(p.m == 1 && TG_PAIR_MR1 == 1));
    if (fixed && nsg == TG_PAIR_R) return TG_PAIR_M(1, TG_PAIR_R);
    // m = 1, a group of ONE super-tile (g = 64 at innerKTiles 4): fixed boundaries too since the dot2 contraction freed the registers
    // (with the MFMA this build spilled 27; 77 -> 81 %), and a group of FOUR super-tiles (g = 256) as one round of a ring of four
    // (76.7 -> 84.0 %; a ring of four at g = 128 / 64 measured 1-1.5 points below the ring of two).  Only the m = 1 kernels are
    // instantiated for these (launch_pair_k directly: launch_pair_m would drag the general kernels in as well).
    if constexpr (!QMX) {
      if (fixed && p.m == 1 && TG_PAIR_MR1 == 1 && !p.norm_w && (nsg == 1 || nsg == 4)) {
        if (nsg == 1) return xg ? launch_pair_k<DT, I, 1, 1, false, 1, true>(pp, lds, st) : launch_pair_k<DT, I, 1, 1, false, 1>(pp, lds, st);
        if constexpr (I <= 4)  // (innerKTiles 8: four super-tiles would be g = 512)
          return xg ? launch_pair_k<DT, I, 1, 1, false, 4, true>(pp, lds, st) : launch_pair_k<DT, I, 1, 1, false, 4>(pp, lds, st);
      }
    }
    return TG_PAIR_M(1, 0);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return TG_PAIR_M(2, 0);
  }
  if constexpr (I >= 8) {
    if (gps == 4) return TG_PAIR_M(4, 0);
  }
#undef TG_PAIR_M
  return TG_PAIR_NA;
}

// Small launches of Bint4 weights (one layer per call): w4_gemm_pair16_kernel, 16 weight rows per workgroup, the whole k-slice
// of a wave requested up front.  Taken when the launch is too small for the persistent kernel (or its LDS plan does not fit)
// and the activations (m <= 16 rows) fit in LDS next to the table.
template <typename DT, int I, bool QMX>
int launch_pair16(const GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (QMX && !std::is_same<DT, BF16>::value) return TG_E_DTYPE;  // mx4 is bf16-only (TinyGemm_int4.cu:758)
  else {
#ifdef TG_DEV_MIN
  if constexpr (!(std::is_same<DT, BF16>::value && I == 4 && !QMX)) return TG_PAIR_NA;
#endif
  if (p.m > 16 || batch > 65535) return TG_PAIR_NA;
  // m = 1 and more than one round of workgroups (one per CU): the streaming kernel's split-K launches are faster there
  // (per hipGraph node, 6144 x 4096: 8.6 us against 10.2 us; 14336 x 4096: 13.8 against 18.3)
  // (not when a fused stage is asked for: only the pair-table kernels have them)
  if (p.m == 1 && !p.x_tc && !p.y_tc && !p.norm_w && !p.epilogue && (int64_t)((p.wrows + 15) / 16) * batch > 256 && (1 << p.gshift) >= 128) return TG_PAIR_NA;
  if (p.norm_w && (QMX || (int64_t)p.m * p.k > 32768)) return TG_PAIR_NA;  // the norm pass: one 32-k chunk per thread
  const int g = 1 << p.gshift;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  Pair16Params pp;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  pp.gch_mask = g / 32 - 1;
  pp.lds_x = 65536;
  const int64_t wgs = (int64_t)((p.wrows + 15) / 16) * batch;
  // one workgroup per CU may take the whole LDS; a launch of more than two rounds of workgroups should fit two per CU
  const unsigned lds_limit = (wgs <= 512 ? 160u : 80u) * 1024u;
  // activation rows that do not fit next to the table are staged one part of k at a time (whole groups per part)
  unsigned lds = 0;
  int phases = 1;
  for (; phases <= (wgs <= 512 && !p.norm_w ? 8 : 1); phases *= 2) {  // (the fused norm needs a row's whole k in one part)
    if (p.ksuper % (phases * nsg) != 0 || p.ngroups % phases != 0) return TG_PAIR_NA;
    const int kp = p.k / phases;
    pp.x_pitch = kp * 2 + 16;
    pp.lds_xs = (pp.lds_x + p.m * pp.x_pitch + 16 + 15) & ~15;
    lds = (unsigned)pp.lds_xs + (QMX ? 0u : (unsigned)(p.ngroups / phases) * 64u);
    if (lds <= lds_limit) break;
  }
  if (lds > lds_limit) return TG_PAIR_NA;
  pp.phases = phases;
  pp.ksuper_p = p.ksuper / phases;
  pp.spw = ((pp.ksuper_p / nsg + 15) / 16) * nsg;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = p.norm_w; pp.norm_eps = p.norm_eps; pp.epilogue = p.epilogue;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  if (p.dry) return TG_PLAN_PAIR;
  const dim3 grid((unsigned)((p.wrows + 15) / 16), (unsigned)batch);
#define TG_P16K(CPG_, NORM_)                                                 \
  do {                                                                      \
    constexpr auto kern = w4_gemm_pair16_kernel<DT, I, QMX, CPG_, 1, NORM_>; \
    const int prc = prepare_lds_kernel<kern>();                             \
    if (prc != 0) return prc;                                               \
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, pp);                \
  } while (0)
#define TG_P16(CPG_)                                 \
  do {                                               \
    if constexpr (!QMX) {                            \
      if (p.norm_w) { TG_P16K(CPG_, true); break; }  \
    }                                                \
    TG_P16K(CPG_, false);                            \
  } while (0)
  if constexpr (QMX) TG_P16(1);  // mx4: group = 32
  else if (g == 32) TG_P16(1);
  else if (g == 64) TG_P16(2);
  else if (g == 128) TG_P16(4);
  else TG_P16(8);
#undef TG_P16K
#undef TG_P16
  return launch_status();
  }
}

// Aint4 weights (weightOnRight = false) on the pair-table kernel: 32 weight rows per work item, v_mfma_f32_16x16x32,
// activations always through the workspace (one pass of at most 8 rows).
template <typename DT, int I, bool QMX>
int launch_pair_a(GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (I < 2) return TG_PAIR_NA;  // one 16-k tile per word set: no word pair for a 32-k MFMA step
  else {
  if (p.m > 16 || p.x_tc || p.y_tc || p.norm_w || p.epilogue) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  PairParams pp;
  pp.x_tc = pp.y_tc = 0; pp.y_tiles = 0;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = nullptr; pp.norm_eps = 0.f; pp.epilogue = 0;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  const int mrows = p.m;
  pp.rused = mrows < 4 ? mrows : 4;
  pp.xs_rows = mrows <= 4 ? 4 : mrows <= 8 ? 8 : 16;
  pp.red_lanes = mrows <= 8 ? 32 : 64;  // lanes 0..31 hold activation rows 0..7
  pp.x_pitch = 0;
  pp.lds_x = QMX ? 0 : 65536;
  pp.xw_pitch = 0;  // the lanes' MFMA operands come straight from the workspace: LDS only holds a zero piece here
  pp.xw_bytes = 0;
  pp.lds_xs = (pp.lds_x + 32 * I + 15) & ~15;
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;  // mx4: no zero point, no activation sums
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  pp.red_alias = lds > 80u * 1024u;
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  if (lds > 80u * 1024u) return TG_PAIR_NA;
  pp.stride_xp = (int64_t)(p.m + 1) * p.k * 2;  // + a zero row
  pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
  const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
  pp.rblocks = (p.wrows + 31) / 32;
  pp.cblocks = 1;
  const int64_t items = (int64_t)pp.rblocks * batch;
  if (items > INT32_MAX || items < TG_PAIR_MIN_ITEMS) return TG_PAIR_NA;
  p.ws_need = need;
  if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
  pp.xp = p.ws;
  pp.xsum = p.ws + batch * pp.stride_xp;
  pp.items = (int32_t)items;
  pp.chunk = 1;  // plain round-robin dealing (chunks of consecutive items measured slower for the 32-row items of this layout)
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  if (!p.dry) {
    const int rc = launch_xprep<DT>(pp, I, 16, batch, st, 1);
    if (rc != 0) return rc;
  }
  if (gps == 1) {
    if (TG_PAIR_NSG2 && nsg == 1) return launch_pair_k<DT, I, 1, 4, QMX, 1, true, true>(pp, lds, st);
    if (TG_PAIR_NSG2 && nsg == TG_PAIR_R) return launch_pair_k<DT, I, 1, 4, QMX, TG_PAIR_R, true, true>(pp, lds, st);
    return launch_pair_k<DT, I, 1, 4, QMX, 0, true, true>(pp, lds, st);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return launch_pair_k<DT, I, 2, 4, QMX, 0, true, true>(pp, lds, st);
  }
  return TG_PAIR_NA;
  }
}

// Bint4 weights with 9 ... 16 activation rows: the 16x16x32 structure of the A-side kernel (32-row work items, duplicated
// table, activations of one pass -- all m <= 16 rows -- straight from the workspace into the MFMA operand) on B-layout words:
// one packed word is one B operand, 4 vector ops per word.  (The 32x32x16 kernel holds 8 rows per pass; a second pass would
// stream the weights twice.)
template <typename DT, int I, bool QMX>
int launch_pair_b16(GemmParams& p, int64_t batch, hipStream_t st) {
  if (p.m > 16 || p.norm_w || p.epilogue) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int gps = g >= 16 * I ? 1 : (16 * I) / g;
  PairParams pp;
  pp.x_tc = p.x_tc; pp.y_tc = p.y_tc; pp.y_tiles = (p.wrows + 15) / 16;
  pp.bias_row_stride = p.bias_row_stride; pp.norm_w = nullptr; pp.norm_eps = 0.f; pp.epilogue = 0;
  pp.x = p.x; pp.w = p.w; pp.qinfo = p.qinfo; pp.lut = p.lut; pp.y = p.y;
  pp.m = p.m; pp.wrows = p.wrows; pp.k = p.k; pp.ntiles = p.ntiles; pp.ksuper = p.ksuper;
  pp.gshift = p.gshift; pp.ngroups = p.ngroups; pp.qtype = p.qtype;
  const int nsg = g >= 16 * I ? g / (16 * I) : 1;
  const int units = p.ksuper / nsg;
  pp.spw = ((units + 7) / 8) * nsg;
  pp.nsg_shift = 0;
  while ((1 << pp.nsg_shift) < nsg) ++pp.nsg_shift;
  pp.gch_mask = g / 32 - 1;
  if (QMX && (p.ngroups < 16 || p.ngroups % 4 != 0 || ((pp.spw * gps) % 4 != 0 && pp.spw * gps > 12))) return TG_PAIR_NA;
  const int mrows = p.m;
  pp.rused = mrows < 4 ? mrows : 4;
  pp.xs_rows = mrows <= 4 ? 4 : mrows <= 8 ? 8 : 16;
  pp.red_lanes = mrows <= 8 ? 32 : 64;  // lanes 0..31 hold activation rows 0..7
  pp.x_pitch = 0;
  pp.lds_x = QMX ? 0 : 65536;
  pp.xw_pitch = 0;
  pp.xw_bytes = 0;
  pp.lds_xs = (pp.lds_x + 32 * I + 15) & ~15;
  pp.lds_red = (pp.lds_xs + (QMX ? 0 : p.ngroups * pp.xs_rows * 4) + 15) & ~15;
  unsigned lds = (unsigned)pp.lds_red + (unsigned)(8 * 2 * pp.rused * pp.red_lanes * 4);
  pp.red_alias = lds > 80u * 1024u;
  if (pp.red_alias) {
    lds = (unsigned)pp.lds_red;
    pp.lds_red = 0;
  }
  if (lds > 80u * 1024u) return TG_PAIR_NA;
  pp.stride_xp = (int64_t)(p.m + 1) * p.k * 2;  // + a zero row
  pp.stride_xsum = ((int64_t)p.ngroups * pp.xs_rows * 4 + 15) & ~(int64_t)15;
  const int64_t need = batch * (pp.stride_xp + pp.stride_xsum);
  pp.rblocks = (p.wrows + 31) / 32;
  pp.cblocks = 1;
  const int64_t items = (int64_t)pp.rblocks * batch;
  if (items > INT32_MAX || items < 2 * TG_PAIR_MIN_ITEMS) return TG_PAIR_NA;  // (32-row items: two per 64-row item of the other kernel)
  p.ws_need = need;
  if (!p.ws_query && (p.ws == nullptr || p.ws_bytes < need)) return TG_PAIR_NA;
  pp.xp = p.ws;
  pp.xsum = p.ws + batch * pp.stride_xp;
  pp.items = (int32_t)items;
  pp.chunk = items >= (int64_t)TG_PAIR_WGS * TG_B16_CHUNK * 4 ? TG_B16_CHUNK : 1;
  pp.stride_x = p.stride_x; pp.stride_w = p.stride_w; pp.stride_qinfo = p.stride_qinfo;
  pp.stride_lut = p.stride_lut; pp.stride_y = p.stride_y;
  pp.bias = p.bias; pp.stride_bias = p.stride_bias; pp.dry = p.dry;
  if (!p.dry) {
    const int rc = launch_xprep<DT>(pp, I, 16, batch, st, 1);
    if (rc != 0) return rc;
  }
  if (gps == 1) {
    if (TG_PAIR_NSG2 && nsg == 1) return launch_pair_k<DT, I, 1, 4, QMX, 1, true, 2>(pp, lds, st);
    if (TG_PAIR_NSG2 && nsg == TG_PAIR_RB16) return launch_pair_k<DT, I, 1, 4, QMX, TG_PAIR_RB16, true, 2>(pp, lds, st);
    return launch_pair_k<DT, I, 1, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  if constexpr (I >= 4) {
    if (gps == 2) return launch_pair_k<DT, I, 2, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  if constexpr (I >= 8) {
    if (gps == 4) return launch_pair_k<DT, I, 4, 4, QMX, 0, true, 2>(pp, lds, st);
  }
  return TG_PAIR_NA;
}

// Bint4 weights, stacked launches, TG_XR_MIN_M ... 16 activation rows, k = 4096: w4_gemm_xr_kernel (one 8-wave workgroup per CU, the
// activations of a wave's k-slice resident in its registers, 64-row work items, two tables).  No workspace, no pre-pass.
template <typename DT, int I, bool QMX, int NCH>
int launch_pair_xr_n(GemmParams& p, int64_t batch, hipStream_t st) {
  if constexpr (I != 4 || (QMX && (NCH != 16 || !std::is_same<DT, BF16>::value))) return TG_PAIR_NA;  // (mx4: bf16, k = 4096)
  else {
#ifdef TG_DEV_MIN
  if constexpr (!std::is_same<DT, BF16>::value) return TG_PAIR_NA;
  else {
#endif
  if (p.m > 16 || p.m < TG_XR_MIN_M || p.norm_w || p.epilogue) return TG_PAIR_NA;
  if (p.ksuper * 16 * I != p.k || p.wrows % 64 != 0 || p.ntiles * 8 != p.wrows) return TG_PAIR_NA;
  const int g = 1 << p.gshift;
  const int cpg = g / 32 < NCH ? g / 32 : NCH;  // 32-k chunks per group inside a wave's slice
#ifdef TG_DEV_MIN
  if (cpg != (QMX ? 1 : 4)) return TG_PAIR_NA;
#endif
  if (QMX ? cpg != 1 : (cpg != 1 && cpg != 2 && cpg != 4 && cpg != 8)) return TG_PAIR_NA;  // g = 32, 64, 128, 256; mx4: g = 32
  if (QMX && p.ngroups % 16 != 0) return TG_PAIR_NA;  // 16-byte exponent blocks
  if (NCH > 16 && cpg == 2) return TG_PAIR_NA;          // (k = 8192, g = 64: that instantiation spills four registers)
  XrParams xp;
  xp.w = p.w; xp.qinfo = p.qinfo; xp.lut = p.lut; xp.y = p.y;
  xp.m = p.m; xp.wrows = p.wrows; xp.k = p.k; xp.ntiles = p.ntiles; xp.ksuper = p.ksuper;
  xp.gshift = p.gshift; xp.ngroups = p.ngroups; xp.qtype = p.qtype;
  xp.rblocks = (p.wrows + 63) / 64;
  const int64_t items = (int64_t)xp.rblocks * batch;
  if (items > INT32_MAX || items < 2 * 256) return TG_PAIR_NA;  // two items per workgroup at least
  xp.items = (int32_t)items;
  xp.lds_xs = 2 * 65536;
  const unsigned lds = QMX ? 32768u : (unsigned)xp.lds_xs + (unsigned)p.ngroups * 64u;  // two tables, the activation sums (mx4: the partial sums only)
  if (lds > 160u * 1024u) return TG_PAIR_NA;
  xp.x = p.x; xp.stride_x = p.stride_x; xp.x_tc = p.x_tc;  // (no pre-pass, no workspace: the kernel arranges the activations itself)
  p.ws_need = 0;
  xp.stride_w = p.stride_w; xp.stride_qinfo = p.stride_qinfo; xp.stride_lut = p.stride_lut; xp.stride_y = p.stride_y;
  xp.bias = p.bias; xp.stride_bias = p.stride_bias; xp.bias_row_stride = p.bias_row_stride;
  xp.y_tc = p.y_tc; xp.y_tiles = (p.wrows + 15) / 16; xp.dry = p.dry;
  if (p.dry) return TG_PLAN_PAIR_XR;
#define TG_XR_LAUNCH(CPG_)                                                  \
  do {                                                                      \
    constexpr auto kern = w4_gemm_xr_kernel<DT, I, NCH, CPG_, (NCH > 16 ? TG_XR_R8K : TG_XR_R)>; \
    const int prc = prepare_lds_kernel<kern>();                             \
    if (prc != 0) return prc;                                               \
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, xp);            \
  } while (0)
  if constexpr (QMX) {
    constexpr auto kern = w4_gemm_xr_kernel<DT, I, NCH, 1, TG_XR_RMX, true>;
    const int prc = prepare_lds_kernel<kern>();
    if (prc != 0) return prc;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, st, xp);
  } else {
#ifdef TG_DEV_MIN
  TG_XR_LAUNCH(4);
#else
  if (cpg == 1) TG_XR_LAUNCH(1);
  else if (cpg == 2) TG_XR_LAUNCH(2);
  else if (cpg == 4) TG_XR_LAUNCH(4);
  else TG_XR_LAUNCH(8);
#endif
  }
#undef TG_XR_LAUNCH
  return launch_status();
#ifdef TG_DEV_MIN
  }
#endif
  }
}

template <typename DT, int I, bool QMX>
int launch_pair_xr(GemmParams& p, int64_t batch, hipStream_t st) {
  if (p.k == 4096) return launch_pair_xr_n<DT, I, QMX, 16>(p, batch, st);
  // k = 8192: 128 registers of activations per lane leave room for two super-tiles in flight only -- faster than the 16x16x32
  // workspace kernel it replaces at 9 ... 16 rows (8192^2, m = 16: 62 vs 47-51 %), slower than the 32x32x16 one below that
  // (m = 8: 66 vs 70 %)
  if (p.k == 8192 && p.m >= 9) return launch_pair_xr_n<DT, I, QMX, 32>(p, batch, st);
  return TG_PAIR_NA;
}

template <typename DT, bool LAYOUT_A, int CANON, bool QMX>
int launch_w4(GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  constexpr int KSTEP = LAYOUT_A ? 64 : 128;
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
  constexpr int WPL = CANON == CANON_NONE ? 1 : (CANON == CANON_PAIR ? 2 : 4);
  // TG_NUM_FAST, weights on the B side: the pair-table kernel (group-scaled numerics) whenever its LDS plan fits
  // (mx4 in BOTH numerics: its dequantised weights, fp4 * 2^(e - 127), are exact 16-bit values however they are formed, so the
  //  pair-table kernels -- which convert them with v_cvt_scalef32_pk_bf16_fp4 -- ARE the reference arithmetic for it)
  if (p.numerics == TG_NUM_FAST || QMX) {
    int rc;
    if constexpr (!LAYOUT_A) {
      rc = launch_pair_xr<DT, 2 * WPL, QMX>(p, batch, st);
      if (rc != TG_PAIR_NA) return rc;
      p.ws_need = 0;
    }
    if constexpr (LAYOUT_A) rc = launch_pair_a<DT, WPL, QMX>(p, batch, st);
    else rc = launch_pair<DT, 2 * WPL, QMX>(p, batch, st);
    if (rc != TG_PAIR_NA) return rc;
    p.ws_need = 0;
    if constexpr (!LAYOUT_A) {
      if (p.m > 8) {
        rc = launch_pair_b16<DT, 2 * WPL, QMX>(p, batch, st);
        if (rc != TG_PAIR_NA) return rc;
        p.ws_need = 0;
      }
      rc = launch_pair16<DT, 2 * WPL, QMX>(p, batch, st);
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
    return launch_stream<DT, LAYOUT_A, WPL, QMX>(p, coltiles, batch, st);
  }
  if (p.dry) return TG_PLAN_SPLITK;
  if (g.waves == 16) {
    hipLaunchKernelGGL((w4_gemm_kernel<DT, LAYOUT_A, CANON, QMX, 16, 2, 4>), grid, dim3(16 * 64), 0, st, p);
  } else {
    hipLaunchKernelGGL((w4_gemm_kernel<DT, LAYOUT_A, CANON, QMX, 8, 2, 4>), grid, dim3(8 * 64), 0, st, p);
  }
  return launch_status();
#endif
}

template <typename DT, bool LAYOUT_A, int CANON>
int launch_w4_q(GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  return p.qtype == TG_Q_MX4 ? launch_w4<DT, LAYOUT_A, CANON, true>(p, coltiles, batch, st)
                             : launch_w4<DT, LAYOUT_A, CANON, false>(p, coltiles, batch, st);
}

template <typename DT, bool LAYOUT_A>
int launch_w4_c(GemmParams& p, int canon, int64_t coltiles, int64_t batch, hipStream_t st) {
  switch (canon) {
    case CANON_NONE: return launch_w4_q<DT, LAYOUT_A, CANON_NONE>(p, coltiles, batch, st);
    case CANON_PAIR: return launch_w4_q<DT, LAYOUT_A, CANON_PAIR>(p, coltiles, batch, st);
    default: return launch_w4_q<DT, LAYOUT_A, CANON_QUAD>(p, coltiles, batch, st);
  }
}

}  // namespace

extern "C" {

int tg_abi_version(void) { return TG_ABI_VERSION; }

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

// dry: 0 launch, 1 report the kernel family (tg_gemm_w4_plan), 2 report the workspace the fastest kernel wants
static int gemm_w4_impl(const tg_w4_gemm* a, int device, tg_stream_t stream, int dry, int64_t* ws_need = nullptr) {
  if (!a || !a->x || !a->w || !a->qinfo || !a->y) return TG_E_NULL;
  if (a->qtype < TG_Q_INT4 || a->qtype > TG_Q_MX4) return TG_E_QTYPE;
  if ((a->qtype == TG_Q_ANY4_GLOBAL || a->qtype == TG_Q_ANY4_ROWWISE) && !a->lut) return TG_E_NULL;
  if (!(a->dtype == TG_BF16 || a->dtype == TG_F16)) return TG_E_DTYPE;
  if (a->qtype == TG_Q_MX4 && a->dtype != TG_BF16) return TG_E_DTYPE;  // TinyGemm_int4.cu:758,782
  if (a->m <= 0 || a->wrows <= 0 || a->k <= 0 || a->m > INT32_MAX || a->wrows > INT32_MAX || a->k > INT32_MAX) return TG_E_SHAPE;
  const int I = a->inner_k_tiles;
  const bool on_right = a->w_on_right != 0;
  if (on_right ? !(I == 2 || I == 4 || I == 8) : !(I == 1 || I == 2 || I == 4)) return TG_E_INNER_K;
  // TinyGemmImpl.cuh:370-376: kTiles % innerKTiles == 0, k % 32 == 0
  if (a->k % 32 != 0 || a->k % (16 * I) != 0) return TG_E_K_DIV;
  const int g = a->group;
  if (!(g == 32 || g == 64 || g == 128 || g == 256) || a->k % g != 0) return TG_E_GROUP;  // TinyGemm_int4.cu:379-387
  const int rows_per_tile = on_right ? 8 : 16;
  if (a->wrows % rows_per_tile != 0) return TG_E_SHAPE;
  if (!aligned16(a->x) || !aligned16(a->w) || (reinterpret_cast<uintptr_t>(a->qinfo) & 3u)) return TG_E_ALIGN;
  if (a->lut && !aligned16(a->lut)) return TG_E_ALIGN;          // LUT rows are read as two 16-byte vectors
  if (a->bias && (reinterpret_cast<uintptr_t>(a->bias) & 7u)) return TG_E_ALIGN;
  if (!(a->numerics == TG_NUM_FAST || a->numerics == TG_NUM_REFERENCE) || a->reserved != 0) return TG_E_SHAPE;
  if (a->workspace && (!aligned16(a->workspace) || a->workspace_bytes < 0)) return TG_E_ALIGN;
  if (!(a->x_layout == TG_LAYOUT_RM || a->x_layout == TG_LAYOUT_TC_A) || !(a->y_layout == TG_LAYOUT_RM || a->y_layout == TG_LAYOUT_TC_A)) return TG_E_LAYOUT;
  if ((a->x_layout || a->y_layout) && (!a->w_on_right || a->m % 16 != 0 || a->bias)) return TG_E_LAYOUT;
  if (a->bias_row_stride < 0 || (a->bias_row_stride && !a->bias) || (a->bias_row_stride & 3)) return TG_E_SHAPE;
  if (!(a->epilogue == TG_EPI_NONE || a->epilogue == TG_EPI_SWIGLU)) return TG_E_SHAPE;
  if (a->norm_weight && !aligned16(a->norm_weight)) return TG_E_ALIGN;
  // the fused stages exist in the TG_NUM_FAST pair-table kernels only (row-major operands)
  if ((a->norm_weight || a->epilogue) && (a->numerics != TG_NUM_FAST || a->x_layout || a->y_layout)) return TG_E_FUSION;
  if (a->norm_weight && a->k % 2048 != 0) return TG_E_FUSION;
  if (a->epilogue == TG_EPI_SWIGLU && (!a->w_on_right || a->bias || a->wrows % 16 != 0)) return TG_E_FUSION;
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
  p.numerics = a->numerics;
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
  int rc;
  if (a->dtype == TG_BF16) rc = on_right ? launch_w4_c<BF16, false>(p, canon, coltiles, batch, st) : launch_w4_c<BF16, true>(p, canon, coltiles, batch, st);
  else rc = on_right ? launch_w4_c<F16, false>(p, canon, coltiles, batch, st) : launch_w4_c<F16, true>(p, canon, coltiles, batch, st);
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

int tg_gemm_w8(const tg_w4_gemm* a, int device, tg_stream_t stream) {
  if (!a || !a->x || !a->w || !a->qinfo || !a->y) return TG_E_NULL;
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
  if ((reinterpret_cast<uintptr_t>(a->x) & 3u) || (reinterpret_cast<uintptr_t>(a->w) & 3u) ||
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
  p.qtype = a->qtype; p.dbg = 0; p.dry = 0;
  p.stride_x = batch > 1 ? a->stride_x : 0; p.stride_w = batch > 1 ? a->stride_w : 0;
  p.stride_qinfo = batch > 1 ? a->stride_qinfo : 0; p.stride_lut = 0; p.stride_y = batch > 1 ? a->stride_y : 0;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  hipStream_t st = (hipStream_t)stream;
  p.rowtiles = (int32_t)cdiv(a->wrows, 16);
  const int64_t coltiles = cdiv(a->m, 16);
#ifdef TG_DEV_MIN
  (void)coltiles; (void)st;
  return TG_E_SHAPE;
#else
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
    if (w_on_right) hipLaunchKernelGGL((f16_gemm_kernel<BF16, false, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
    else hipLaunchKernelGGL((f16_gemm_kernel<BF16, true, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
  } else {
    if (w_on_right) hipLaunchKernelGGL((f16_gemm_kernel<F16, false, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
    else hipLaunchKernelGGL((f16_gemm_kernel<F16, true, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
  }
  return launch_status();
}

}  // extern "C"

#include "decode_glue.cuh"
#include "peer_gather.cuh"
