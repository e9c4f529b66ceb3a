// fused_qkv_attn.hip -- prices the decode layer's q/k/v-projection -> attention seam on MI355X BEFORE the fused kernel is built.
//
//   hipcc --offload-arch=gfx950 -O3 -o variants/fused_qkv_attn tools/ubench/fused_qkv_attn.hip && variants/fused_qkv_attn
//
// Llama-3-8B at batch 1: the q/k/v GEMV (6144 x 4096, 13.6 MB of 4-bit weights) is followed by one query's attention over the
// cache (8 KV groups, ~0.7 MB).  Head group g needs only the 768 projection rows of ITS group (4 q heads, k, v).  Two hipGraphs
// of L "layers" (each layer also has a trailing 8 MiB GEMV-shaped node standing for o-proj, so that the chain has the real
// producer -> consumer -> producer shape):
//   split   node A: 256 workgroups stream 12 MiB, write 24 values each (row-major vector)       node B: 8 workgroups (one per
//           group): read the cache slice (80 KiB), then the group's 768 values, "attend", write 512 outputs      node C: o-proj
//   fused   node AB: workgroup b serves group b % 8 (what runs on XCD b % 8 in practice -- used for speed only); every workgroup
//           publishes its 24 values as 8-byte {tag, 2 x 16-bit} granules (write-through, relaxed agent-scope atomics: the data is
//           the flag); local workgroup 0 of the group requests its cache slice BEFORE it waits, sweeps the group's 384 granules
//           until every tag equals the epoch, attends, writes 512 outputs, clears the tags.                      node C: o-proj
// The epoch lives in device memory and is advanced by the last node of the graph (a graph replays with frozen arguments).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  const u32x4* w;       // this node's weights
  const uint32_t* x_in; // 8 KiB activation vector of the previous node
  uint32_t* x_out;      // split: [6144] 16-bit pairs -> 3072 dwords; o-proj: 4096 x 16 bit
  u64* gran;            // fused: [8][384] granules
  const u32x4* cache;   // [8][80 KiB]
  uint32_t* attn_out;   // [8][256] dwords (512 x 16 bit)
  unsigned* epoch;      // device-resident, advanced once per graph replay
  unsigned* err;
  int pieces;           // 16-byte pieces per thread of the weight stream
  int fused;
};

__device__ __forceinline__ float stream(const P& p, int b, int t, const u32x4 xv) {
  const u32x4* base = p.w + (size_t)b * p.pieces * 512 + t;
  u32x4 ring[8];
  const int n = p.pieces;
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = __builtin_nontemporal_load(base + (size_t)(i < n ? i : n - 1) * 512);
  float acc = 0.f;
  for (int i = 0; i < n; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u32x4 v = ring[j];
      acc += __builtin_bit_cast(float, (v[0] ^ xv[0]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[1] ^ xv[1]) & 0x3fffffffu) +
             __builtin_bit_cast(float, (v[2] ^ xv[2]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[3] ^ xv[3]) & 0x3fffffffu);
      int nx = i + j + 8;
      nx = nx < n ? nx : n - 1;
      ring[j] = __builtin_nontemporal_load(base + (size_t)nx * 512);
    }
  }
  return acc;
}

// the attention-shaped part of one group: 80 KiB of cache per workgroup (10 pieces per thread, requested by `cache_request`), the
// group's 768 values (384 dwords) in LDS, a dependent two-pass reduction (scores -> weights -> values), 256 dwords out
__device__ __forceinline__ void attend(const P& p, int g, int t, const u32x4 (&cv)[10], const uint32_t* qkv_lds, float* red) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) s += __builtin_bit_cast(float, (cv[i][0] ^ qkv_lds[(t + i) % 384]) & 0x3fffffffu);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  float m = 0.f;
  for (int i = 0; i < 8; ++i) m += red[i];
  float o2 = 0.f;
#pragma unroll
  for (int i = 5; i < 10; ++i) o2 += __builtin_bit_cast(float, (cv[i][1] ^ __builtin_bit_cast(uint32_t, m)) & 0x3fffffffu);
  if (t < 256) p.attn_out[g * 256 + t] = __builtin_bit_cast(uint32_t, o2) | 0x00010001u;
}

__global__ void __launch_bounds__(512) proj_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);
  uint32_t* qkv_lds = reinterpret_cast<uint32_t*>(smem + 256);
  const int t = threadIdx.x, b = blockIdx.x;
  const int g = b & 7, li = b >> 3;
  const u32x4 xv = reinterpret_cast<const u32x4*>(p.x_in)[t];
  const bool consumer = p.fused && li == 0;
  u32x4 cv[10];
  if (consumer) {  // the cache slice does not depend on this layer's projection: requested up front, lands under the weight stream
#pragma unroll
    for (int i = 0; i < 10; ++i) cv[i] = p.cache[((size_t)g * 10 + i) * 512 + t];
  }
  float acc = stream(p, b, t, xv);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += red[i];
  const uint32_t val = (__builtin_bit_cast(uint32_t, s) & 0x7fff7fffu) | 0x00010001u;
  if (!p.fused) {
    if (t < 12) p.x_out[b * 12 + t] = val + t;   // 24 16-bit values per workgroup, row-major
    return;
  }
  const unsigned ep = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 12) __hip_atomic_store(p.gran + (size_t)g * 384 + li * 12 + t, ((u64)ep << 32) | (val + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!consumer) return;
  // ---- the group's consumer: one wave sweeps the 384 granules (6 per lane) until every tag equals the epoch ----
  if (t < 64) {
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const u64 x = __hip_atomic_load(p.gran + (size_t)g * 384 + k * 64 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qkv_lds[k * 64 + t] = (uint32_t)x;
        ok &= (unsigned)(x >> 32) == ep;
      }
      if (__all(ok)) break;
      if (++spins > 400000u) { *p.err = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)   // ready for the next replay of THIS node: tags cleared by the only reader
      __hip_atomic_store(p.gran + (size_t)g * 384 + k * 64 + t, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  attend(p, g, t, cv, qkv_lds, red);
}

__global__ void __launch_bounds__(512) attn_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);
  uint32_t* qkv_lds = reinterpret_cast<uint32_t*>(smem + 256);
  const int t = threadIdx.x, g = blockIdx.x;
  u32x4 cv[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) cv[i] = p.cache[((size_t)g * 10 + i) * 512 + t];
  if (t < 384) qkv_lds[t] = p.x_in[g * 384 + t];   // (row-major vector: the group's rows gathered -- one contiguous run here)
  __syncthreads();
  attend(p, g, t, cv, qkv_lds, red);
}

__global__ void __launch_bounds__(512) oproj_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, b = blockIdx.x;
  const u32x4 xv = reinterpret_cast<const u32x4*>(p.x_in)[t & 255];   // attention output: 4096 x 16 bit
  float acc = stream(p, b, t, xv);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t < 8) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += red[i];
    p.x_out[b * 8 + t] = __builtin_bit_cast(uint32_t, s) | 0x00010001u;
  }
}

__global__ void bump_epoch(unsigned* e) { *e = *e + 1u; }

int main() {
  const int L = 32;
  std::vector<u32x4*> wq(L), wo(L);
  for (int i = 0; i < L; ++i) {
    CHECK(hipMalloc(&wq[i], 12u << 20)); CHECK(hipMemset(wq[i], 0x11 + i, 12u << 20));
    CHECK(hipMalloc(&wo[i], 8u << 20)); CHECK(hipMemset(wo[i], 0x31 + i, 8u << 20));
  }
  uint32_t *x, *qkv, *ao; u64* gran; u32x4* cache; unsigned *epoch, *err;
  CHECK(hipMalloc(&x, 8192)); CHECK(hipMemset(x, 1, 8192));
  CHECK(hipMalloc(&qkv, 3072 * 4)); CHECK(hipMemset(qkv, 1, 3072 * 4));
  CHECK(hipMalloc(&ao, 8 * 256 * 4 * 2)); CHECK(hipMemset(ao, 1, 8 * 256 * 4 * 2));
  CHECK(hipMalloc(&gran, (size_t)L * 8 * 384 * 8)); CHECK(hipMemset(gran, 0, (size_t)L * 8 * 384 * 8));
  CHECK(hipMalloc(&cache, (size_t)L * 8 * 10 * 512 * 16)); CHECK(hipMemset(cache, 3, (size_t)L * 8 * 10 * 512 * 16));
  CHECK(hipMalloc(&epoch, 64)); CHECK(hipMalloc(&err, 64));
  unsigned one = 1, zero = 0;
  CHECK(hipMemcpy(epoch, &one, 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(err, &zero, 4, hipMemcpyHostToDevice));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int fused = 0; fused < 2; ++fused) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < L; ++i) {
      P p{};
      p.w = wq[i]; p.x_in = x; p.x_out = qkv; p.gran = gran + (size_t)i * 8 * 384; p.cache = cache + (size_t)i * 8 * 10 * 512;
      p.attn_out = ao; p.epoch = epoch; p.err = err; p.pieces = (12 << 20) / 16 / 512 / 256; p.fused = fused;
      hipLaunchKernelGGL(proj_kernel, dim3(256), dim3(512), 4096, st, p);
      if (!fused) { P q = p; q.x_in = qkv; hipLaunchKernelGGL(attn_kernel, dim3(8), dim3(512), 4096, st, q); }
      P o{}; o.w = wo[i]; o.x_in = ao; o.x_out = x; o.pieces = (8 << 20) / 16 / 512 / 256;
      hipLaunchKernelGGL(oproj_kernel, dim3(256), dim3(512), 4096, st, o);
    }
    hipLaunchKernelGGL(bump_epoch, dim3(1), dim3(1), 0, st, epoch);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    const int reps = 30;
    CHECK(hipEventRecord(e0, st));
    for (int it = 0; it < reps; ++it) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned e; CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("%-44s %7.2f us per layer%s\n", fused ? "fused  (proj + in-kernel hand-over + attend, o-proj)" : "split  (proj, attend, o-proj: three nodes)",
           ms * 1e3 / reps / L, e ? "   [a sweep gave up]" : "");
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  return 0;
}
