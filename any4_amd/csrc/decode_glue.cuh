// Non-GEMM kernels of one batch-1 decode step (include/decode_glue_hip.h).  Included at the end of
// tinygemm_hip.hip: shares its type traits (BF16 / F16), DeviceScope and launch_status().
//
// These are small HBM/L2-latency-bound kernels (a few KiB to a few hundred KiB per launch); what matters is that
// there are 5 of them per layer instead of ~45 torch launches, not their individual bandwidth.  Rounding points
// mirror the torch formulation in any4_amd/decode.py so the two paths can be compared in tests.
#include "../../include/decode_glue_hip.h"

namespace {

// (tg_common.cuh, tgl: DPP rotations and row swaps instead of six trips through the LDS crossbar)
__device__ __forceinline__ float wave_sum(float v) { return tgl::wave_sum(v); }
__device__ __forceinline__ float wave_max(float v) { return tgl::wave_max(v); }
// all threads of a 256-thread block get the reduction; `scratch` holds >= 4 floats and is reusable afterwards
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float* scratch) {
  v = MAX ? wave_max(v) : wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = scratch[0];
#pragma unroll
  for (int w = 1; w < 4; ++w) r = MAX ? fmaxf(r, scratch[w]) : r + scratch[w];
  return r;
}

template <typename DT>
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f[2 * q] = DT::lo_f32(v[q]);
    f[2 * q + 1] = DT::hi_f32(v[q]);
  }
}
template <typename DT>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{DT::pack2(f[0], f[1]), DT::pack2(f[2], f[3]), DT::pack2(f[4], f[5]), DT::pack2(f[6], f[7])};
}
template <typename DT>
__device__ __forceinline__ float round16(float a) { return DT::to_f32(DT::from_f32(a)); }

// ---- residual add + RMSNorm: one 256-thread block per row ------------------------------------------
template <typename DT>
__global__ void __launch_bounds__(256) add_rmsnorm_kernel(const u32x4* h, const u32x4* __restrict__ delta,
                                                          const u32x4* __restrict__ w, u32x4* h_out, u32x4* __restrict__ y,
                                                          int dim8, float inv_dim, float eps) {
  __shared__ float scratch[4];
  const int64_t base = (int64_t)blockIdx.x * dim8;
  float ss = 0.f;
  for (int v = threadIdx.x; v < dim8; v += 256) {
    float a[8];
    unpack8<DT>(h[base + v], a);
    if (delta) {
      float d[8];
      unpack8<DT>(delta[base + v], d);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = round16<DT>(a[e] + d[e]);
    }
    if (delta || h_out != h) h_out[base + v] = pack8<DT>(a);
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += a[e] * a[e];
  }
  if (!y) return;
  const float r = rsqrtf(block_reduce<false>(ss, scratch) * inv_dim + eps);
  for (int v = threadIdx.x; v < dim8; v += 256) {
    float a[8], g[8];
    unpack8<DT>(h_out[base + v], a);  // written by this same thread above (or the untouched input)
    unpack8<DT>(w[v], g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = round16<DT>(a[e] * r) * g[e];
    y[base + v] = pack8<DT>(a);
  }
}

// q . K[s] over d = 8 * d8 elements, fp32, in element order; the 16-byte loads are issued eight at a time (a
// position's key row is one dependent L2 round trip otherwise).
template <typename DT>
__device__ __forceinline__ float qk_dot(const float* qf, const u32x4* __restrict__ Krow, int d8) {
  float acc = 0.f;
  for (int v0 = 0; v0 < d8; v0 += 8) {
    u32x4 kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v0 + u < d8) kk[u] = Krow[v0 + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (v0 + u < d8) {
        float kf[8];
        unpack8<DT>(kk[u], kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(qf[(v0 + u) * 8 + e], kf[e], acc);
      }
    }
  }
  return acc;
}

// value contraction of one thread: acc[e] += p[s] * V[s][vc*8 + e] over s = part, part + nparts, ... (ascending).
// Eight independent 16-byte loads are kept in flight: the loop is latency-bound (one L2 round trip per position).
// `vnew` (LDS, may be null) supplies position S - 1 when the cache row was written by this very launch.
template <typename DT>
__device__ __forceinline__ void pv_accumulate(const u32x4* __restrict__ V, const float* vnew, const float* sc, float inv,
                                              int S, int d8, int vc, int part, int nparts, float (&acc)[8], int first = 0) {
  for (int s0 = part + first; s0 < S; s0 += nparts * 8) {
    u32x4 vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u * nparts;
      if (s < S && !(vnew && s == S - 1)) vv[u] = V[(int64_t)s * d8 + vc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u * nparts;
      if (s < S) {
        const float p = round16<DT>(sc[s] * inv);  // probabilities are cast to 16 bit before the matmul
        float vf[8];
        if (vnew && s == S - 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vf[e] = vnew[vc * 8 + e];
        } else {
          unpack8<DT>(vv[u], vf);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
      }
    }
  }
}

// ---- rotary embedding + KV-cache write: block = one head of one sequence, thread = one rotation pair ----
template <typename DT>
__global__ void rope_kv_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ cos, const float* __restrict__ sin,
                               const int64_t* __restrict__ pos_p, uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache,
                               uint16_t* __restrict__ v_cache, int hl, int kvl, int d, int64_t max_seq) {
  const int b = blockIdx.x, head = blockIdx.y, j = threadIdx.x, d2 = d >> 1;
  const int64_t pos = *pos_p;
  if (pos < 0 || pos >= max_seq) return;  // the position lives on the device (graph replays bypass the host check): never index the cache outside [0, max_seq)
  const uint16_t* src = qkv + ((int64_t)b * (hl + 2 * kvl) + head) * d;
  if (head >= hl + kvl) {  // v: plain copy into the cache
    uint16_t* dst = v_cache + (((int64_t)b * kvl + (head - hl - kvl)) * max_seq + pos) * d;
    dst[j] = src[j];
    dst[j + d2] = src[j + d2];
    return;
  }
  const float x1 = DT::to_f32(src[j]), x2 = DT::to_f32(src[j + d2]);
  const float c1 = cos[pos * d + j], c2 = cos[pos * d + j + d2], s1 = sin[pos * d + j], s2 = sin[pos * d + j + d2];
  // x * cos + rotate_half(x) * sin with each product and the sum rounded separately (as the torch ops do)
  const float o1 = __fadd_rn(__fmul_rn(x1, c1), __fmul_rn(-x2, s1));
  const float o2 = __fadd_rn(__fmul_rn(x2, c2), __fmul_rn(x1, s2));
  uint16_t* dst = head < hl ? q_out + ((int64_t)b * hl + head) * d
                            : k_cache + (((int64_t)b * kvl + (head - hl)) * max_seq + pos) * d;
  dst[j] = DT::from_f32(o1);
  dst[j + d2] = DT::from_f32(o2);
}

// ---- decode attention: block = one query head of one sequence --------------------------------------
template <typename DT>
__global__ void __launch_bounds__(256) decode_attn_kernel(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k_cache,
                                                          const uint16_t* __restrict__ v_cache, const int64_t* __restrict__ pos_p,
                                                          uint16_t* __restrict__ out, int hl, int kvl, int d, int64_t max_seq,
                                                          float scale) {
  extern __shared__ float sm[];  // [256 q] [4 scratch] [max_seq scores]; the partial outputs reuse the scores
  float* qf = sm;
  float* scratch = sm + 256;
  float* sc = sm + 260;
  const int b = blockIdx.x / hl, h = blockIdx.x % hl, kv = h / (hl / kvl), t = threadIdx.x;
  if (*pos_p < 0 || *pos_p >= max_seq) return;  // see rope_kv_kernel
  const int S = (int)(*pos_p) + 1, d8 = d >> 3;
  if (t < d) qf[t] = DT::to_f32(q[((int64_t)b * hl + h) * d + t]);
  __syncthreads();
  const u32x4* K = reinterpret_cast<const u32x4*>(k_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  const u32x4* V = reinterpret_cast<const u32x4*>(v_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  float mx = -INFINITY;
  for (int s = t; s < S; s += 256) {
    const float acc = qk_dot<DT>(qf, K + (int64_t)s * d8, d8);
    const float x = round16<DT>(acc) * scale;  // the score matrix is a 16-bit tensor in the torch formulation
    sc[s] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_reduce<true>(mx, scratch);
  float sum = 0.f;
  for (int s = t; s < S; s += 256) {
    const float e = __expf(sc[s] - mx);
    sc[s] = e;
    sum += e;
  }
  const float inv = 1.f / block_reduce<false>(sum, scratch);
  __syncthreads();
  // value contraction: thread = (8-wide column vc, position partition part)
  const int vc = t % d8, part = t / d8, nparts = 256 / d8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  pv_accumulate<DT>(V, nullptr, sc, inv, S, d8, vc, part, nparts, acc);
  __syncthreads();  // everyone is done reading the scores; reuse them as [nparts][d] partial outputs
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[part * d + vc * 8 + e] = acc[e];
  __syncthreads();
  if (t < d) {
    float o = 0.f;
    for (int p = 0; p < nparts; ++p) o += sc[p * d + t];
    out[((int64_t)b * hl + h) * d + t] = DT::from_f32(o);
  }
}

// ---- RoPE + KV-cache write + decode attention in one launch: block = one query head of one sequence ----
// Every block rotates its own q head and the k head of its KV group, writes k / v of the new token into the caches
// (the hl/kvl blocks of a group write identical bytes) and uses its LDS copy for position *pos, so nothing is read
// back from global memory within the launch.  Same arithmetic, in the same order, as rope_kv_kernel + decode_attn_kernel.
template <typename DT>
__global__ void __launch_bounds__(256) rope_attn_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ cos,
                                                        const float* __restrict__ sin, const int64_t* __restrict__ pos_p,
                                                        uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                        uint16_t* __restrict__ out, int hl, int kvl, int d, int64_t max_seq,
                                                        float scale) {
  extern __shared__ float sm[];  // [256 q] [256 k_new] [256 v_new] [4 scratch] [scores]
  float* qf = sm;
  float* kn = sm + 256;
  float* vn = sm + 512;
  float* scratch = sm + 768;
  float* sc = sm + 772;
  const int b = blockIdx.x / hl, h = blockIdx.x % hl, kv = h / (hl / kvl), t = threadIdx.x;
  const int64_t pos = *pos_p;
  if (pos < 0 || pos >= max_seq) return;  // the position lives on the device (graph replays bypass the host check): never index the cache outside [0, max_seq)
  const int S = (int)pos + 1, d8 = d >> 3, d2 = d >> 1;
  const uint16_t* row = qkv + (int64_t)b * (hl + 2 * kvl) * d;
  uint16_t* kdst = k_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * d;
  uint16_t* vdst = v_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * d;
  // Issue every cache read this thread will need first -- its key row and the first eight value vectors of its
  // partition -- so that the HBM round trips of qkv / K / V overlap instead of following each other (the KV cache of a
  // layer is cold: ~4 GB of weights have streamed through the caches since the previous token).
  const u32x4* K = reinterpret_cast<const u32x4*>(k_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  const u32x4* V = reinterpret_cast<const u32x4*>(v_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  const int vc = t % d8, part = t / d8, nparts = 256 / d8;
  const bool kpre = d8 <= 16 && t < S - 1;
  u32x4 kreg[16], vpre[8];
  if (kpre) {
#pragma unroll
    for (int v = 0; v < 16; ++v)
      if (v < d8) kreg[v] = K[(int64_t)t * d8 + v];
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int sv = part + u * nparts;
    if (sv < S - 1) vpre[u] = V[(int64_t)sv * d8 + vc];
  }
  if (t < d) {  // threads 0..d2-1 rotate q, d2..d-1 rotate k (one rotation pair each)
    const bool isk = t >= d2;
    const int j = isk ? t - d2 : t;
    const uint16_t* src = row + (isk ? (hl + kv) * d : h * d);
    const float x1 = DT::to_f32(src[j]), x2 = DT::to_f32(src[j + d2]);
    const float c1 = cos[pos * d + j], c2 = cos[pos * d + j + d2], s1 = sin[pos * d + j], s2 = sin[pos * d + j + d2];
    const uint16_t o1 = DT::from_f32(__fadd_rn(__fmul_rn(x1, c1), __fmul_rn(-x2, s1)));
    const uint16_t o2 = DT::from_f32(__fadd_rn(__fmul_rn(x2, c2), __fmul_rn(x1, s2)));
    float* dstf = isk ? kn : qf;
    dstf[j] = DT::to_f32(o1);
    dstf[j + d2] = DT::to_f32(o2);
    if (isk) { kdst[j] = o1; kdst[j + d2] = o2; }
    const uint16_t vraw = row[(hl + kvl + kv) * d + t];
    vn[t] = DT::to_f32(vraw);
    vdst[t] = vraw;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int s = t; s < S; s += 256) {
    float acc = 0.f;
    if (s == S - 1) {
      for (int e = 0; e < d; ++e) acc = fmaf(qf[e], kn[e], acc);
    } else if (kpre && s == t) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        if (v < d8) {
          float kf[8];
          unpack8<DT>(kreg[v], kf);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc = fmaf(qf[v * 8 + e], kf[e], acc);
        }
      }
    } else {
      acc = qk_dot<DT>(qf, K + (int64_t)s * d8, d8);
    }
    const float x = round16<DT>(acc) * scale;
    sc[s] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_reduce<true>(mx, scratch);
  float sum = 0.f;
  for (int s = t; s < S; s += 256) {
    const float e = __expf(sc[s] - mx);
    sc[s] = e;
    sum += e;
  }
  const float inv = 1.f / block_reduce<false>(sum, scratch);
  __syncthreads();
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 8; ++u) {  // first batch: the prefetched vectors (same order as pv_accumulate)
    const int sv = part + u * nparts;
    if (sv < S) {
      const float p = round16<DT>(sc[sv] * inv);
      float vf[8];
      if (sv == S - 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vf[e] = vn[vc * 8 + e];
      } else {
        unpack8<DT>(vpre[u], vf);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vf[e], acc[e]);
    }
  }
  pv_accumulate<DT>(V, vn, sc, inv, S, d8, vc, part, nparts, acc, nparts * 8);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[part * d + vc * 8 + e] = acc[e];
  __syncthreads();
  if (t < d) {
    float o = 0.f;
    for (int p = 0; p < nparts; ++p) o += sc[p * d + t];
    out[((int64_t)b * hl + h) * d + t] = DT::from_f32(o);
  }
}

// ---- split-S variant of rope_attn_kernel: grid (bs * hl, NS); block (bh, c) handles positions [c * CS, (c + 1) * CS)
// with CS = ceil((*pos + 1) / NS), writes (max, sum, unnormalised output) of its chunk to `part`, and the LAST block of a
// head to arrive (atomic counter, self-resetting, so the launch is replayable in a graph) combines the NS partials.
// One block per head leaves 7/8 of the CUs idle and walks a long context at ~40 ns per position; this one fills the GPU.
// Softmax statistics are combined flash-decoding style, so probabilities are normalised AFTER the value contraction
// (the single-block kernels round normalised probabilities to 16 bit first): same result within 16-bit rounding.
template <typename DT>
__global__ void __launch_bounds__(256) rope_attn_split_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ cos,
                                                              const float* __restrict__ sin, const int64_t* __restrict__ pos_p,
                                                              uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                              uint16_t* __restrict__ out, float* part, int* counters, int hl,
                                                              int kvl, int d, int64_t max_seq, float scale) {
  extern __shared__ float sm[];  // [256 q] [256 k_new] [256 v_new] [4 scratch] [scores / partial outputs]
  float* qf = sm;
  float* kn = sm + 256;
  float* vn = sm + 512;
  float* scratch = sm + 768;
  float* sc = sm + 772;
  __shared__ int s_last;
  const int bh = blockIdx.x, c = blockIdx.y, NS = gridDim.y;
  const int b = bh / hl, h = bh % hl, kv = h / (hl / kvl), t = threadIdx.x;
  const int64_t pos = *pos_p;
  if (pos < 0 || pos >= max_seq) return;  // the position lives on the device (graph replays bypass the host check): never index the cache outside [0, max_seq)
  const int S = (int)pos + 1, d8 = d >> 3, d2 = d >> 1;
  const int CS = (S + NS - 1) / NS;
  const int c0 = c * CS, c1 = min(S, c0 + CS);  // may be empty
  const bool owner = c0 <= (int)pos && (int)pos < c1;  // this block's chunk holds the new token
  const uint16_t* row = qkv + (int64_t)b * (hl + 2 * kvl) * d;
  if (t < d) {
    const bool isk = t >= d2;
    const int j = isk ? t - d2 : t;
    const uint16_t* src = row + (isk ? (hl + kv) * d : h * d);
    const float x1 = DT::to_f32(src[j]), x2 = DT::to_f32(src[j + d2]);
    const float cs1 = cos[pos * d + j], cs2 = cos[pos * d + j + d2], s1 = sin[pos * d + j], s2 = sin[pos * d + j + d2];
    const uint16_t o1 = DT::from_f32(__fadd_rn(__fmul_rn(x1, cs1), __fmul_rn(-x2, s1)));
    const uint16_t o2 = DT::from_f32(__fadd_rn(__fmul_rn(x2, cs2), __fmul_rn(x1, s2)));
    float* dstf = isk ? kn : qf;
    dstf[j] = DT::to_f32(o1);
    dstf[j + d2] = DT::to_f32(o2);
    const uint16_t vraw = row[(hl + kvl + kv) * d + t];
    vn[t] = DT::to_f32(vraw);
    if (owner) {
      uint16_t* kdst = k_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * d;
      uint16_t* vdst = v_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * d;
      if (isk) { kdst[j] = o1; kdst[j + d2] = o2; }
      vdst[t] = vraw;
    }
  }
  __syncthreads();
  const u32x4* K = reinterpret_cast<const u32x4*>(k_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  const u32x4* V = reinterpret_cast<const u32x4*>(v_cache + ((int64_t)b * kvl + kv) * max_seq * d);
  float mx = -INFINITY;
  for (int s = c0 + t; s < c1; s += 256) {
    float acc = 0.f;
    if (s == S - 1) {
      for (int e = 0; e < d; ++e) acc = fmaf(qf[e], kn[e], acc);
    } else {
      acc = qk_dot<DT>(qf, K + (int64_t)s * d8, d8);
    }
    const float x = round16<DT>(acc) * scale;
    sc[s - c0] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_reduce<true>(mx, scratch);
  float sum = 0.f;
  for (int s = c0 + t; s < c1; s += 256) {
    const float e = __expf(sc[s - c0] - mx);
    sc[s - c0] = e;
    sum += e;
  }
  sum = block_reduce<false>(sum, scratch);
  __syncthreads();
  const int vc = t % d8, prt = t / d8, nparts = 256 / d8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // positions are chunk-relative inside pv_accumulate: shift the row pointer, the new token is the chunk's last row
  pv_accumulate<DT>(V + (int64_t)c0 * d8, owner ? vn : nullptr, sc, 1.0f, c1 > c0 ? c1 - c0 : 0, d8, vc, prt, nparts, acc);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) sc[prt * d + vc * 8 + e] = acc[e];
  __syncthreads();
  float* mine = part + ((int64_t)bh * NS + c) * (d + 2);
  if (t < d) {
    float o = 0.f;
    for (int p = 0; p < nparts; ++p) o += sc[p * d + t];
    mine[2 + t] = o;
  }
  if (t == 0) { mine[0] = mx; mine[1] = sum; }
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = atomicAdd(&counters[bh], 1) == NS - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (t < d) {
    const float* base = part + (int64_t)bh * NS * (d + 2);
    float M = -INFINITY;
    for (int i = 0; i < NS; ++i) M = fmaxf(M, __hip_atomic_load(base + i * (d + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float L = 0.f, o = 0.f;
    for (int i = 0; i < NS; ++i) {
      const float mi = __hip_atomic_load(base + i * (d + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float w = mi == -INFINITY ? 0.f : __expf(mi - M);
      L += w * __hip_atomic_load(base + i * (d + 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o += w * __hip_atomic_load(base + i * (d + 2) + 2 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[((int64_t)b * hl + h) * d + t] = DT::from_f32(o / L);
  }
  if (t == 0) counters[bh] = 0;  // ready for the next launch / graph replay
}

template <typename DT>
__device__ __forceinline__ float dot2_16(uint32_t a, uint32_t b, float acc) {
  if constexpr (std::is_same<DT, BF16>::value)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), acc, false);
  else
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), acc, false);
}

// ---- RoPE + KV-cache write + decode attention, one launch, ONE barrier: block = one query head of one sequence, 8 waves ----
// What a batch-1 decode step needs from this node is latency, not bandwidth (at a few hundred cached positions the K and V
// rows of a head are < 100 KiB): rope_attn_kernel above walks five dependent phases with four barriers (11 us per layer in
// the Llama-3-8B step, profiles/); this one issues EVERY load of the launch -- q / k / v of the new token, the rotary table row,
// and the K and V rows of up to 256 cached positions -- right behind the read of `pos`, and then computes without another
// global round trip:
//   row layout   a K / V row is d / 8 lanes x 16 bytes; a wave-load covers 64 / (d / 8) rows, the block 8 times that per
//                iteration, NI iterations (256 positions) in flight per chunk.
//   scores       v_dot2 of the packed rotated q with the packed K piece, summed over the row's lanes with DPP (quad_perm /
//                row_half_mirror / row_mirror: no LDS traffic), rounded to 16 bits like the matmul of the torch formulation,
//                times scale.
//   softmax      per ROW GROUP (the d / 8 lanes that share rows): running max / sum / unnormalised value accumulator over the
//                group's rows (flash-decoding style; probabilities are not rounded to 16 bits before the value contraction --
//                as dg_rope_attn_split: the same result within 16-bit rounding); the 32 (64) groups meet ONCE in LDS.
//   RoPE         rotate_half pairs (j, j + d/2) sit d/16 lanes apart in the row: one lane exchange per value; the same products
//                and roundings as rope_kv_kernel (the cache rows written are bit-identical to it).
template <typename DT, int LPR>
__global__ void __launch_bounds__(512) rope_attn_online_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ cos,
                                                               const float* __restrict__ sin, const int64_t* __restrict__ pos_p,
                                                               uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                               uint16_t* __restrict__ out, int hl, int kvl, int64_t max_seq, float scale,
                                                               unsigned long long* trace, float* part, int* counters) {
  // gridDim.y = NS > 1: split over the sequence.  Block (head, c) takes the 32-row iterations c, c + NS, c + 2 NS, ... of the context
  // (a partition that does not depend on the position: the speculative requests below stay possible), writes its (max, sum,
  // unnormalised output) to `part`, and the last block of a head to arrive (self-resetting counter: replayable in a graph) combines
  // the NS partials.  One block per head walks a long context at one CU's pace (~50 GB/s: 19 us per layer at 1900 positions).
  constexpr int D = LPR * 8, RPW = 64 / LPR, NWV = 8, RPI = NWV * RPW, NI = 256 / RPI, NG = RPI;
  const int NS = gridDim.y, cblk = blockIdx.y;
  // first row of the block's local iteration jl
  auto row0_of = [&](int jl) -> int { return (jl * NS + cblk) * RPI; };
#if GEMV_TRACE
  unsigned long long tr[8];
#define DG_STAMP(n) tr[n] = __builtin_amdgcn_s_memrealtime()
#else
#define DG_STAMP(n)
#endif
  DG_STAMP(0);
  extern __shared__ float sm[];  // [8 waves][D + 2]: unnormalised accumulator, max, sum
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = lane / LPR, i = lane % LPR, grp = wave * RPW + g;
  // Eight KV groups (Llama-3-8B on one GPU), eight XCDs: the `rep` query heads of a group read the same K / V rows, and workgroup w
  // runs on XCD w % 8 (observed; used for speed only) -- head = rep (w % 8) + w / 8 puts a group's heads behind ONE L2, which then
  // fetches the rows once instead of once per XCD the group is spread over.
  const int b = blockIdx.x / hl, rep = hl / kvl;
  const int hb = blockIdx.x % hl;
#ifndef DG_ATTN_XCD_HEADS
#define DG_ATTN_XCD_HEADS 1
#endif
  const int h = (DG_ATTN_XCD_HEADS && kvl == 8 && hl == 8 * rep) ? rep * (hb & 7) + (hb >> 3) : hb;
  const int kv = h / rep;
  const uint16_t* row = qkv + (int64_t)b * (hl + 2 * kvl) * D;
  // (row r, piece i) of this head's K / V at byte ((r * LPR + i) << 4): a 32-bit offset on a scalar base (host: max_seq * d * 2 < 4 GiB)
  const char* K = reinterpret_cast<const char*>(k_cache + ((int64_t)b * kvl + kv) * max_seq * D);
  const char* V = reinterpret_cast<const char*>(v_cache + ((int64_t)b * kvl + kv) * max_seq * D);
  auto piece = [&](const char* base, int r) -> u32x4 { return *reinterpret_cast<const u32x4*>(base + (((uint32_t)r * LPR + (uint32_t)i) << 4)); };
  // the first NSPEC iterations' rows are requested BEFORE the position is known (rows past it are valid memory and masked later):
  // the read of `pos` is a dependent round trip the K / V requests of a short context need not wait for
  // (all of the first chunk: same box, Llama-3-8B decode, 2 -> 8 iterations: 620 -> 620 tokens/s at position 136, 544 -> 553 at 900,
  //  481 -> 494 at 1900.  Requesting the NEXT chunk before the current one is consumed -- a second register set, 211 VGPRs -- measured
  //  SLOWER than this: 550 / 480 tokens/s at 900 / 1900; profiles/r04_ab_attention_speculation.txt)
#ifndef DG_ATTN_NSPEC
#define DG_ATTN_NSPEC 8
#endif
  constexpr int NSPEC = NI < DG_ATTN_NSPEC ? NI : DG_ATTN_NSPEC;
  u32x4 kk[NI], vv[NI];
#pragma unroll
  for (int it = 0; it < NSPEC; ++it) {
    const int rc = min(row0_of(it) + grp, (int)max_seq - 1);
    kk[it] = piece(K, rc);
    vv[it] = piece(V, rc);
  }
  const int64_t pos = *pos_p;
  if (pos < 0 || pos >= max_seq) return;  // the position lives on the device (graph replays bypass the host check)
  const int S = (int)pos + 1;
  DG_STAMP(1);
  // ---- every load of the launch (positions past the end re-read position 0; the new token's row comes from qkv) ----
  const u32x4 qraw = reinterpret_cast<const u32x4*>(row + h * D)[i];
  const u32x4 kraw = reinterpret_cast<const u32x4*>(row + (hl + kv) * D)[i];
  const u32x4 vraw = reinterpret_cast<const u32x4*>(row + (hl + kvl + kv) * D)[i];
  const f32x4 c0 = reinterpret_cast<const f32x4*>(cos + pos * D)[2 * i], c1 = reinterpret_cast<const f32x4*>(cos + pos * D)[2 * i + 1];
  const f32x4 s0 = reinterpret_cast<const f32x4*>(sin + pos * D)[2 * i], s1 = reinterpret_cast<const f32x4*>(sin + pos * D)[2 * i + 1];
  auto request = [&](int base, int first) {
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      if (it < first || row0_of(base / RPI + it) >= S) continue;  // (wave-uniform) iterations past the last position request nothing
      const int r = row0_of(base / RPI + it) + grp;
      const int rc = r < S - 1 ? r : 0;
      kk[it] = piece(K, rc);
      vv[it] = piece(V, rc);
    }
  };
  request(0, NSPEC);
  DG_STAMP(2);
  // ---- rotary embedding of q and k (this lane's 8 elements; the partner elements j +- d/2 are LPR/2 lanes away) ----
  float cf[8], sf[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { cf[e] = c0[e]; cf[4 + e] = c1[e]; sf[e] = s0[e]; sf[4 + e] = s1[e]; }
  const bool lower = i < LPR / 2;
  auto rotate = [&](const u32x4& raw) -> u32x4 {
    float x[8], o[8];
    unpack8<DT>(raw, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float partner;
      if constexpr (LPR == 16) partner = tgl::lane_xor<8>(x[e], lane);  // (one DPP move)
      else partner = __shfl_xor(x[e], LPR / 2, 64);
      // rope_kv_kernel: o1 = x1 c1 + (-x2) s1 (lower half), o2 = x2 c2 + x1 s2 (upper half); every product and the sum rounded once
      // (two rounded products and a rounded sum, never an fma: hipcc contracts __fmul_rn / __fadd_rn across a lambda, so the
      //  products are made opaque)
      float p1 = x[e] * cf[e], p2 = (lower ? -partner : partner) * sf[e];
      asm volatile("" : "+v"(p1), "+v"(p2));
      o[e] = p1 + p2;
    }
    return pack8<DT>(o);
  };
  const u32x4 qp = rotate(qraw), kn = rotate(kraw);
  DG_STAMP(3);
  if (h % rep == 0 && grp == 0 && cblk == 0) {  // one row group of the KV group's first head writes the new token's cache rows
    reinterpret_cast<u32x4*>(k_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * D)[i] = kn;
    reinterpret_cast<u32x4*>(v_cache + (((int64_t)b * kvl + kv) * max_seq + pos) * D)[i] = vraw;
  }
  auto dpp_add = [](float v, auto ctrl) -> float {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  // ---- this group's rows: running max m, sum l, unnormalised accumulator acc[8] (the lane's 8 elements of the value row) ----
  float m = -INFINITY, l = 0.f, acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int base = 0; row0_of(base / RPI) < S; base += NI * RPI) {  // (base counts the block's own rows: local iteration base / RPI)
    if (base > 0) request(base, 0);
    float x[NI];
    float cm = -INFINITY;
#pragma unroll
    for (int it = 0; it < NI; ++it) {
      const int r = row0_of(base / RPI + it) + grp;
      const u32x4 kr = r == S - 1 ? kn : kk[it];
      // (literal indices: with a loop variable hipcc (ROCm 7.2) fed dword 0 of both vectors to all four v_dot2)
      float d2 = dot2_16<DT>(kr[0], qp[0], 0.f);
      d2 = dot2_16<DT>(kr[1], qp[1], d2);
      d2 = dot2_16<DT>(kr[2], qp[2], d2);
      d2 = dot2_16<DT>(kr[3], qp[3], d2);
#ifdef DG_ATTN_SHFL
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) d2 += __shfl_xor(d2, o, 64);
#else
      d2 = dpp_add(d2, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
      d2 = dpp_add(d2, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
      d2 = dpp_add(d2, std::integral_constant<int, 0x141>{});   // row_half_mirror: the other quad of the 8 lanes
      if constexpr (LPR == 16) d2 = dpp_add(d2, std::integral_constant<int, 0x140>{});  // row_mirror: the other half of the 16 lanes
#endif
      x[it] = r < S ? round16<DT>(d2) * scale : -INFINITY;
#ifdef DG_ATTN_DUMP
      if (blockIdx.x == 0 && i == 0 && r < S) reinterpret_cast<float*>(v_cache + (((int64_t)kvl - 1) * max_seq + (max_seq - 1)) * D)[r] = x[it];
#endif
      cm = fmaxf(cm, x[it]);
    }
    const float mn = fmaxf(m, cm);
    if (base == 0) { DG_STAMP(4); }
    if (mn > -INFINITY) {  // (uniform within the row group; a group without rows so far keeps its zeros)
      const float alpha = __expf(m - mn);  // exp(-inf) = 0 for the first chunk with rows
      l *= alpha;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int r = row0_of(base / RPI + it) + grp;
        const float pr = __expf(x[it] - mn);  // 0 for rows past the end
        l += pr;
        float vf[8];
        // (rows past the end re-read position 0, which at S = 1 is the row this launch is writing: never let its bytes in)
        unpack8<DT>(r == S - 1 ? vraw : r < S ? vv[it] : u32x4{0u, 0u, 0u, 0u}, vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(pr, vf[e], acc[e]);
      }
      m = mn;
    }
  }
  DG_STAMP(5);
  // ---- the groups meet: out[e] = sum_g exp(m_g - M) acc_g[e] / sum_g exp(m_g - M) l_g.  First the RPW groups of a wave through
  // lane exchanges (no barrier), then the 8 waves once through LDS ----
  auto meet = [&](auto OO) {
    constexpr int o = decltype(OO)::value;
    auto other = [&](float v) -> float {
      if constexpr (o >= 16) return tgl::lane_xor<o>(v, lane);  // (one row swap and a select instead of a trip through the LDS crossbar)
      else return __shfl_xor(v, o, 64);
    };
    const float mo = other(m), lo = other(l);
    const float mn = fmaxf(m, mo);
    // (both sides empty: keep the zeros, never exp(-inf - -inf))
    const float wa = mn > -INFINITY ? __expf(m - mn) : 0.f, wb = mn > -INFINITY ? __expf(mo - mn) : 0.f;
    l = l * wa + lo * wb;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] * wa + other(acc[e]) * wb;
    m = mn;
  };
  if constexpr (LPR <= 8) meet(std::integral_constant<int, 8>{});
  meet(std::integral_constant<int, 16>{});
  meet(std::integral_constant<int, 32>{});
  float* mine = sm + wave * (D + 2);
  if (g == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) mine[i * 8 + e] = acc[e];
    if (i == 0) { mine[D] = m; mine[D + 1] = l; }
  }
  __syncthreads();
  DG_STAMP(6);
  float Mb = -INFINITY, num = 0.f, den = 0.f;
  if (t < D) {
#pragma unroll
    for (int w = 0; w < NWV; ++w) Mb = fmaxf(Mb, sm[w * (D + 2) + D]);
    if (Mb > -INFINITY) {  // (a block of a split launch may have no rows)
#pragma unroll
      for (int w = 0; w < NWV; ++w) {
        const float wt = __expf(sm[w * (D + 2) + D] - Mb);
        num = fmaf(wt, sm[w * (D + 2) + t], num);
        den = fmaf(wt, sm[w * (D + 2) + D + 1], den);
      }
    }
    if (NS == 1) out[((int64_t)b * hl + h) * D + t] = DT::from_f32(num / den);
  }
  if (NS > 1) {  // (uniform over the launch)
    const int bh = blockIdx.x;
    float* mine = part + ((int64_t)bh * NS + cblk) * (D + 2);
    // No device-scope fence here: __threadfence() writes back and invalidates the XCD's whole L2 and costs ~12 us per block on this
    // 8-XCD part (measured: NS = 2 / 4 / 8 made the decode step 20 / 30 / 45 % slower at ANY context length).  Instead every value that
    // crosses blocks is written and read with agent-scope atomics (write-through stores, cache-bypassing loads), a thread's stores
    // are complete (vmcnt(0), the workgroup barrier's release) before thread 0 increments the head's counter, and the counter itself
    // is an agent-scope atomic: the last block to arrive sees every partial.  (The hardware guide's valid form "sc1 stores AND
    // sc1 loads on both sides, every writing wave drained before the flag": no acquire is needed because no plain load ever reads
    // these words.  tests/test_gpu_decode.py::test_rope_attn_split_many_back_to_back_launches pins it under uneven load.)
    if (t < D) __hip_atomic_store(mine + 2 + t, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) {
      __hip_atomic_store(mine, Mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 1, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    __shared__ int s_last;
    if (t == 0) s_last = __hip_atomic_fetch_add(&counters[bh], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NS - 1;
    __syncthreads();
    if (s_last) {
      if (t < D) {
        const float* base = part + (int64_t)bh * NS * (D + 2);
        float M = -INFINITY;
        for (int c2 = 0; c2 < NS; ++c2) M = fmaxf(M, __hip_atomic_load(base + c2 * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float L = 0.f, o = 0.f;
        for (int c2 = 0; c2 < NS; ++c2) {
          const float mi = __hip_atomic_load(base + c2 * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float w = mi == -INFINITY ? 0.f : __expf(mi - M);
          L += w * __hip_atomic_load(base + c2 * (D + 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          o += w * __hip_atomic_load(base + c2 * (D + 2) + 2 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out[((int64_t)b * hl + h) * D + t] = DT::from_f32(o / L);
      }
      if (t == 0) __hip_atomic_store(&counters[bh], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch / graph replay
    }
  }
#if GEMV_TRACE
  DG_STAMP(7);
  if (t == 0 && trace) {
#pragma unroll
    for (int n = 0; n < 8; ++n) trace[(size_t)blockIdx.x * 8 + n] = tr[n];
  }
#endif
#undef DG_STAMP
}

// ---- y[a][row] = RNE16(sum_k x[a][k] w[row][k]) for ROW-MAJOR 16-bit weights [n][k] and 1 ... 4 activation rows: the LM head of the
// decode step (the reference leaves it an nn.Linear, quantize.py:34-36; 1.05 GB at Llama-3-8B's vocabulary -- hipBLASLt streams it at
// 5.8 TB/s, this kernel's only job is to stream it faster).  A wave owns a weight row at a time: lane l holds the 16-byte pieces
// l, l + 64, ... of the row (every wave-load is 1 KiB contiguous) and the same pieces of the activation rows in registers for the
// whole launch; NR rows per wave are in flight. k = 512 KP (KP pieces per lane); f32 accumulation in one fixed order per row. ----
template <typename DT, int M, int KP, int NR>
__global__ void __launch_bounds__(512) linear16_gemv_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                            uint16_t* __restrict__ y, int64_t n, int64_t rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  const int64_t r0 = wv * rows_per_wave, r1 = min(r0 + rows_per_wave, n);
  if (r0 >= r1) return;
  constexpr int K = KP * 512;
  u32x4 xr[M][KP];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int j = 0; j < KP; ++j) xr[a][j] = reinterpret_cast<const u32x4*>(x + (int64_t)a * K)[lane + 64 * j];
  for (int64_t r = r0; r < r1; r += NR) {
    u32x4 wr[NR][KP];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int64_t rr = min(r + i, r1 - 1);  // (rows past the range re-read its last row: every load unconditional)
#pragma unroll
      for (int j = 0; j < KP; ++j) wr[i][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w + rr * K) + lane + 64 * j);
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      float acc[M];
#pragma unroll
      for (int a = 0; a < M; ++a) {
        acc[a] = 0.f;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          acc[a] = dot2_16<DT>(wr[i][j][0], xr[a][j][0], acc[a]);
          acc[a] = dot2_16<DT>(wr[i][j][1], xr[a][j][1], acc[a]);
          acc[a] = dot2_16<DT>(wr[i][j][2], xr[a][j][2], acc[a]);
          acc[a] = dot2_16<DT>(wr[i][j][3], xr[a][j][3], acc[a]);
        }
        acc[a] = wave_sum(acc[a]);
      }
      if (lane == 0 && r + i < r1) {
#pragma unroll
        for (int a = 0; a < M; ++a) y[(int64_t)a * n + r + i] = DT::from_f32(acc[a]);
      }
    }
  }
}

#if GEMV_TRACE
unsigned long long* g_attn_trace = nullptr;
#endif
// ---- SwiGLU ----------------------------------------------------------------------------------------
template <typename DT>
__global__ void __launch_bounds__(256) swiglu_kernel(const u32x4* __restrict__ gu, u32x4* __restrict__ out, int64_t il8, int64_t total8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / il8, j = i % il8;
    float g[8], u[8];
    unpack8<DT>(gu[b * 2 * il8 + j], g);
    unpack8<DT>(gu[b * 2 * il8 + il8 + j], u);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = round16<DT>(g[e] / (1.f + __expf(-g[e]))) * u[e];
    out[i] = pack8<DT>(g);
  }
}

}  // namespace

extern "C" {

int dg_add_rmsnorm(const void* h, const void* delta, const void* w, void* h_out, void* y, int64_t rows, int64_t dim,
                   float eps, int dtype, int device, tg_stream_t stream) {
  if (!h || !h_out || (y && !w)) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (rows <= 0 || dim <= 0 || dim % 8 != 0 || dim > 16384 || rows > INT32_MAX) return TG_E_SHAPE;
  if (!aligned16(h) || !aligned16(h_out) || (delta && !aligned16(delta)) || (y && (!aligned16(y) || !aligned16(w)))) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  auto kern = dtype == TG_BF16 ? add_rmsnorm_kernel<BF16> : add_rmsnorm_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const u32x4*)h, (const u32x4*)delta,
                     (const u32x4*)w, (u32x4*)h_out, (u32x4*)y, (int)(dim / 8), 1.0f / (float)dim, eps);
  return launch_status();
}

int dg_rope_kv(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* q_out, void* k_cache,
               void* v_cache, int64_t bs, int hl, int kvl, int d, int64_t max_seq, int dtype, int device, tg_stream_t stream) {
  if (!qkv || !cos || !sin || !pos || !q_out || !k_cache || !v_cache) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || bs > 65535 || hl <= 0 || kvl <= 0 || hl % kvl != 0 || d <= 0 || d % 2 != 0 || d > 256 || max_seq <= 0) return TG_E_SHAPE;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  auto kern = dtype == TG_BF16 ? rope_kv_kernel<BF16> : rope_kv_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)bs, (unsigned)(hl + 2 * kvl)), dim3(d / 2), 0, (hipStream_t)stream,
                     (const uint16_t*)qkv, cos, sin, pos, (uint16_t*)q_out, (uint16_t*)k_cache, (uint16_t*)v_cache, hl, kvl, d, max_seq);
  return launch_status();
}

int dg_decode_attn(const void* q, const void* k_cache, const void* v_cache, const int64_t* pos, void* out, int64_t bs,
                   int hl, int kvl, int d, int64_t max_seq, float scale, int dtype, int device, tg_stream_t stream) {
  if (!q || !k_cache || !v_cache || !pos || !out) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || hl <= 0 || kvl <= 0 || hl % kvl != 0 || d < 8 || d % 8 != 0 || d > 256 || (256 % (d / 8)) != 0 ||
      max_seq <= 0 || max_seq > 8192 || bs * hl > INT32_MAX)
    return TG_E_SHAPE;
  if (!aligned16(k_cache) || !aligned16(v_cache)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t sc_floats = max_seq > (256 / (d / 8)) * (int64_t)d ? max_seq : (256 / (d / 8)) * (int64_t)d;
  const unsigned lds = (unsigned)((260 + sc_floats) * sizeof(float));
  auto kern = dtype == TG_BF16 ? decode_attn_kernel<BF16> : decode_attn_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(bs * hl)), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)q,
                     (const uint16_t*)k_cache, (const uint16_t*)v_cache, pos, (uint16_t*)out, hl, kvl, d, max_seq, scale);
  return launch_status();
}

int dg_rope_attn(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache, void* v_cache,
                 void* out, int64_t bs, int hl, int kvl, int d, int64_t max_seq, float scale, int dtype, int device,
                 tg_stream_t stream) {
  if (!qkv || !cos || !sin || !pos || !k_cache || !v_cache || !out) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || hl <= 0 || kvl <= 0 || hl % kvl != 0 || d < 8 || d % 8 != 0 || d > 256 || (256 % (d / 8)) != 0 ||
      max_seq <= 0 || max_seq > 8192 || bs * hl > INT32_MAX)
    return TG_E_SHAPE;
  if (!aligned16(k_cache) || !aligned16(v_cache)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t sc_floats = max_seq > (256 / (d / 8)) * (int64_t)d ? max_seq : (256 / (d / 8)) * (int64_t)d;
  const unsigned lds = (unsigned)((772 + sc_floats) * sizeof(float));
  auto kern = dtype == TG_BF16 ? rope_attn_kernel<BF16> : rope_attn_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(bs * hl)), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)qkv, cos, sin, pos,
                     (uint16_t*)k_cache, (uint16_t*)v_cache, (uint16_t*)out, hl, kvl, d, max_seq, scale);
  return launch_status();
}

#if GEMV_TRACE
TG_API void tg_dev_attn_trace(unsigned long long* buf) { g_attn_trace = buf; }  // developer builds: [blocks][8] stamps of the last launch
#endif

int dg_rope_attn_online(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache, void* v_cache,
                        void* out, int64_t bs, int hl, int kvl, int d, int64_t max_seq, float scale, int dtype, int device,
                        tg_stream_t stream) {
  if (!qkv || !cos || !sin || !pos || !k_cache || !v_cache || !out) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || hl <= 0 || kvl <= 0 || hl % kvl != 0 || !(d == 64 || d == 128) || max_seq <= 0 || bs * hl > INT32_MAX ||
      max_seq * d * 2 >= ((int64_t)1 << 32))
    return TG_E_SHAPE;
  if (!aligned16(qkv) || !aligned16(cos) || !aligned16(sin) || !aligned16(k_cache) || !aligned16(v_cache)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const unsigned lds = (unsigned)(8 * (d + 2) * sizeof(float));
#define DG_ONLINE(DTT, LPR_)                                                                                                        \
  hipLaunchKernelGGL((rope_attn_online_kernel<DTT, LPR_>), dim3((unsigned)(bs * hl)), dim3(512), lds, (hipStream_t)stream,          \
                     (const uint16_t*)qkv, cos, sin, pos, (uint16_t*)k_cache, (uint16_t*)v_cache, (uint16_t*)out, hl, kvl, max_seq, scale, trace, \
                     (float*)nullptr, (int*)nullptr)
  unsigned long long* trace = nullptr;
#if GEMV_TRACE
  trace = g_attn_trace;
#endif
  if (dtype == TG_BF16) {
    if (d == 128) DG_ONLINE(BF16, 16); else DG_ONLINE(BF16, 8);
  } else {
    if (d == 128) DG_ONLINE(F16, 16); else DG_ONLINE(F16, 8);
  }
#undef DG_ONLINE
  return launch_status();
}

int dg_rope_attn_split(const void* qkv, const float* cos, const float* sin, const int64_t* pos, void* k_cache, void* v_cache,
                       void* out, void* scratch, int64_t scratch_bytes, int64_t bs, int hl, int kvl, int d, int64_t max_seq,
                       float scale, int nsplit, int dtype, int device, tg_stream_t stream) {
  if (!qkv || !cos || !sin || !pos || !k_cache || !v_cache || !out || !scratch) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || hl <= 0 || kvl <= 0 || hl % kvl != 0 || d < 8 || d % 8 != 0 || d > 256 || (256 % (d / 8)) != 0 ||
      max_seq <= 0 || max_seq > 65536 || bs * hl > INT32_MAX || nsplit < 1 || nsplit > 64)
    return TG_E_SHAPE;
  if (scratch_bytes < dg_rope_attn_split_scratch_bytes(bs, hl, d, nsplit)) return TG_E_SHAPE;
  if (!aligned16(k_cache) || !aligned16(v_cache) || !aligned16(scratch)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t cs = (max_seq + nsplit - 1) / nsplit;
  const int64_t sc_floats = cs > (256 / (d / 8)) * (int64_t)d ? cs : (256 / (d / 8)) * (int64_t)d;
  const unsigned lds = (unsigned)((772 + sc_floats) * sizeof(float));
  int* counters = reinterpret_cast<int*>(scratch);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + ((bs * hl * 4 + 15) / 16) * 16);
  if ((d == 64 || d == 128) && max_seq * d * 2 < ((int64_t)1 << 32) && aligned16(qkv) && aligned16(cos) && aligned16(sin)) {
    // the one-barrier kernel, split over the sequence (same scratch layout: counters, then [head][chunk][max, sum, d outputs])
    const unsigned lds1 = (unsigned)(8 * (d + 2) * sizeof(float));
    const dim3 grid((unsigned)(bs * hl), (unsigned)nsplit);
#define DG_ONLINE_SPLIT(DTT, LPR_)                                                                                                   \
  hipLaunchKernelGGL((rope_attn_online_kernel<DTT, LPR_>), grid, dim3(512), lds1, (hipStream_t)stream, (const uint16_t*)qkv, cos, sin, \
                     pos, (uint16_t*)k_cache, (uint16_t*)v_cache, (uint16_t*)out, hl, kvl, max_seq, scale, (unsigned long long*)nullptr, part, counters)
    if (dtype == TG_BF16) {
      if (d == 128) DG_ONLINE_SPLIT(BF16, 16); else DG_ONLINE_SPLIT(BF16, 8);
    } else {
      if (d == 128) DG_ONLINE_SPLIT(F16, 16); else DG_ONLINE_SPLIT(F16, 8);
    }
#undef DG_ONLINE_SPLIT
    return launch_status();
  }
  if (lds > 64u * 1024u) return TG_E_SHAPE;
  auto kern = dtype == TG_BF16 ? rope_attn_split_kernel<BF16> : rope_attn_split_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(bs * hl), (unsigned)nsplit), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)qkv,
                     cos, sin, pos, (uint16_t*)k_cache, (uint16_t*)v_cache, (uint16_t*)out, part, counters, hl, kvl, d, max_seq, scale);
  return launch_status();
}

int64_t dg_rope_attn_split_scratch_bytes(int64_t bs, int hl, int d, int nsplit) {
  return ((bs * hl * 4 + 15) / 16) * 16 + bs * hl * (int64_t)nsplit * (d + 2) * 4;
}

int dg_linear16(const void* x, const void* w, void* y, int64_t m, int64_t n, int64_t k, int dtype, int device, tg_stream_t stream) {
  if (!x || !w || !y) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (m < 1 || m > 4 || n <= 0 || !(k == 2048 || k == 4096 || k == 8192) || (k == 8192 && m > 2)) return TG_E_SHAPE;  // (what is instantiated: the caller falls back to its GEMM)
  if (!aligned16(x) || !aligned16(w)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  // persistent waves: 256 CUs x 2 workgroups x 8 waves, contiguous row ranges
  const int64_t waves = 2 * 8 * (int64_t)cu_count();
  const int64_t rpw = (n + waves - 1) / waves;
  const unsigned blocks = (unsigned)((n + rpw * 8 - 1) / (rpw * 8));
#define DG_L16(DTT, M_, KP_, NR_)                                                                                              \
  hipLaunchKernelGGL((linear16_gemv_kernel<DTT, M_, KP_, NR_>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const uint16_t*)x, \
                     (const uint16_t*)w, (uint16_t*)y, n, rpw)
#define DG_L16_M(DTT, KP_, NR_)                                                          \
  do {                                                                                   \
    if (m == 1) DG_L16(DTT, 1, KP_, NR_); else if (m == 2) DG_L16(DTT, 2, KP_, NR_);      \
    else if (m == 3) DG_L16(DTT, 3, KP_, NR_); else DG_L16(DTT, 4, KP_, NR_);             \
  } while (0)
#define DG_L16_K(DTT)                                                                    \
  do {                                                                                   \
    if (k == 2048) DG_L16_M(DTT, 4, 4); else if (k == 4096) DG_L16_M(DTT, 8, 2);           \
    else if (m == 1) DG_L16(DTT, 1, 16, 1); else DG_L16(DTT, 2, 16, 1);  /* (k = 8192: one or two rows fit the registers) */ \
  } while (0)
  if (dtype == TG_BF16) DG_L16_K(BF16); else DG_L16_K(F16);
#undef DG_L16_K
#undef DG_L16_M
#undef DG_L16
  return launch_status();
}

int dg_swiglu(const void* gu, void* out, int64_t bs, int64_t il, int dtype, int device, tg_stream_t stream) {
  if (!gu || !out) return TG_E_NULL;
  if (!(dtype == TG_BF16 || dtype == TG_F16)) return TG_E_DTYPE;
  if (bs <= 0 || il <= 0 || il % 8 != 0) return TG_E_SHAPE;
  if (!aligned16(gu) || !aligned16(out)) return TG_E_ALIGN;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  const int64_t total8 = bs * (il / 8);
  const int64_t blocks = cdiv(total8, 256);
  auto kern = dtype == TG_BF16 ? swiglu_kernel<BF16> : swiglu_kernel<F16>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)gu, (u32x4*)out, il / 8, total8);
  return launch_status();
}

}  // extern "C"
