// overlap_chain.hip -- can consecutive dependent weight-streaming nodes of a decode step OVERLAP across the kernel boundary?
//
//   hipcc --offload-arch=gfx950 -O3 -o variants/overlap_chain tools/ubench/overlap_chain.hip && variants/overlap_chain
//
// graph_chain.hip measured what a dependent node costs when every node waits for its predecessor at the kernel boundary
// (2.8 us before the first byte + bytes / 6.3 TB/s).  Here node i is launched when node i - 2 has finished (two streams,
// captured alternately: two parallel chains in the graph), requests its weights at once, and only then waits -- on a counter
// in memory that the workgroups of node i - 1 bump after their write-through output stores -- for its input vector.
// Every node is 256 workgroups of 512 threads with < 80 KiB of LDS and <= 128 VGPRs, so two nodes are always co-resident
// (never three: node i + 2 sits behind node i in its stream): no deadlock by construction; every spin is bounded anyway.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  const u32x4* w;
  const uint32_t* x_in;
  uint32_t* x_out;
  unsigned* ctr_in;   // bumped once per workgroup of the producer (null: no wait -- the kernel boundary orders)
  unsigned* ctr_out;  // bumped by this node
  unsigned* err;
  int pieces;
  unsigned expect;
  int lds_pad;
};

__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

template <int R>
__global__ void __launch_bounds__(512, 4) node_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, b = blockIdx.x;
  const u32x4* base = p.w + (size_t)b * p.pieces * 512 + t;
  u32x4 ring[R];
  const int n = p.pieces;
#pragma unroll
  for (int i = 0; i < R; ++i) ring[i] = ld_nt(base + (size_t)i * 512);
  if (p.ctr_in) {  // wait for the producer: ONE lane polls, relaxed; then one agent acquire; then the workgroup
    if (t == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(p.ctr_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.expect) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > 20000u) { *p.err = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  const u32x4 xv = reinterpret_cast<const u32x4*>(p.x_in)[t];
  float acc = 0.f;
  for (int i = 0; i < n; i += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u32x4 v = ring[j];
      acc += __builtin_bit_cast(float, (v[0] ^ xv[0]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[1] ^ xv[1]) & 0x3fffffffu) +
             __builtin_bit_cast(float, (v[2] ^ xv[2]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[3] ^ xv[3]) & 0x3fffffffu);
      int nx = i + j + R;
      nx = nx < n ? nx : n - 1;
      ring[j] = ld_nt(base + (size_t)nx * 512);
    }
  }
  uint32_t junk = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) junk ^= ring[j][0];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t < 8) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += red[i];
    const uint32_t o = __builtin_bit_cast(uint32_t, s) | 0x00010001u | (junk == 0x12345u ? 2u : 0u);
    // write-through (agent scope) so that the consumer needs no release fence from us
    __hip_atomic_store(p.x_out + b * 8 + t, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (p.ctr_out && t < 64) {  // the storing wave: its stores have left, then one bump per workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) __hip_atomic_fetch_add(p.ctr_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  const int N = 64;
  std::vector<u32x4*> w(N);
  const size_t maxbytes = 64u << 20;
  for (int i = 0; i < N; ++i) { CHECK(hipMalloc(&w[i], maxbytes)); CHECK(hipMemset(w[i], 0x11 + i, maxbytes)); }
  uint32_t* x[2];
  CHECK(hipMalloc(&x[0], 8192)); CHECK(hipMalloc(&x[1], 8192));
  CHECK(hipMemset(x[0], 1, 8192)); CHECK(hipMemset(x[1], 1, 8192));
  unsigned* ctr; CHECK(hipMalloc(&ctr, (N + 2) * 64));  // one counter per node, 64 bytes apart
  unsigned* err; CHECK(hipMalloc(&err, 64)); CHECK(hipMemset(err, 0, 64));
  hipStream_t s0, s1; CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&fork)); CHECK(hipEventCreate(&join));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  struct Mode { const char* name; size_t bytes; int overlap; unsigned lds; };
  std::vector<Mode> modes = {
      {"8 MiB  boundary-ordered", 8u << 20, 0, 64}, {"8 MiB  overlapped (two chains + counters)", 8u << 20, 1, 64},
      {"8 MiB  overlapped, 72 KiB LDS per workgroup", 8u << 20, 1, 72 * 1024},
      {"16 MiB boundary-ordered", 16u << 20, 0, 64}, {"16 MiB overlapped", 16u << 20, 1, 64},
      {"32 MiB boundary-ordered", 32u << 20, 0, 64}, {"32 MiB overlapped", 32u << 20, 1, 64},
      {"64 MiB boundary-ordered", 64u << 20, 0, 64}, {"64 MiB overlapped", 64u << 20, 1, 64},
  };
  // eager first: two streams, no graph (is it the graph that serialises the two chains?)
  for (int overlap = 0; overlap < 2; ++overlap) {
    CHECK(hipMemset(ctr, 0, (N + 2) * 64));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, s0));
    if (overlap) { CHECK(hipEventRecord(fork, s0)); CHECK(hipStreamWaitEvent(s1, fork, 0)); }
    for (int i = 0; i < N; ++i) {
      P p;
      p.w = w[i]; p.x_in = x[i & 1]; p.x_out = x[(i + 1) & 1];
      p.pieces = 4; p.err = err; p.expect = 256; p.lds_pad = 0;
      p.ctr_in = (overlap && i > 0) ? ctr + (i - 1) * 16 : nullptr;
      p.ctr_out = overlap ? ctr + i * 16 : nullptr;
      hipLaunchKernelGGL(node_kernel<4>, dim3(256), dim3(512), 64, (overlap && (i & 1)) ? s1 : s0, p);
    }
    if (overlap) { CHECK(hipEventRecord(join, s1)); CHECK(hipStreamWaitEvent(s0, join, 0)); }
    CHECK(hipEventRecord(e1, s0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("eager 8 MiB %-36s %7.2f us per node%s\n", overlap ? "two streams + counters" : "one stream", ms * 1e3 / N, herr ? "   [a spin gave up]" : "");
    fflush(stdout);
    CHECK(hipMemset(err, 0, 64));
  }
  for (const Mode& m : modes) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
    CHECK(hipMemsetAsync(ctr, 0, (N + 2) * 64, s0));
    if (m.overlap) { CHECK(hipEventRecord(fork, s0)); CHECK(hipStreamWaitEvent(s1, fork, 0)); }
    for (int i = 0; i < N; ++i) {
      P p;
      p.w = w[i]; p.x_in = x[i & 1]; p.x_out = x[(i + 1) & 1];
      p.pieces = (int)(m.bytes / 16 / 512 / 256);
      p.err = err; p.expect = 256; p.lds_pad = 0;
      p.ctr_in = (m.overlap && i > 0) ? ctr + (i - 1) * 16 : nullptr;
      p.ctr_out = m.overlap ? ctr + i * 16 : nullptr;
      hipStream_t st = (m.overlap && (i & 1)) ? s1 : s0;
      if (p.pieces < 8) hipLaunchKernelGGL(node_kernel<4>, dim3(256), dim3(512), m.lds, st, p);
      else hipLaunchKernelGGL(node_kernel<8>, dim3(256), dim3(512), m.lds, st, p);
    }
    if (m.overlap) { CHECK(hipEventRecord(join, s1)); CHECK(hipStreamWaitEvent(s0, join, 0)); }
    CHECK(hipStreamEndCapture(s0, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) CHECK(hipGraphLaunch(ge, s0));
    CHECK(hipStreamSynchronize(s0));
    const int reps = 5;
    CHECK(hipEventRecord(e0, s0));
    for (int it = 0; it < reps; ++it) CHECK(hipGraphLaunch(ge, s0));
    CHECK(hipEventRecord(e1, s0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const double us = ms * 1e3 / reps / N;
    printf("%-48s %7.2f us per node   %6.2f TB/s%s\n", m.name, us, m.bytes / us / 1e6, herr ? "   [a spin gave up]" : "");
    fflush(stdout);
    CHECK(hipMemset(err, 0, 64));
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  return 0;
}
