// grid_barrier.hip -- what does it cost to hand a vector from ALL workgroups of one stage to ALL workgroups of the next INSIDE one
// kernel on MI355X (8 XCDs, one L2 each)?  The number a persistent whole-layer decode kernel would pay per stage instead of the
// 2.8 us of a dependent hipGraph node (graph_chain.hip).
//
//   hipcc --offload-arch=gfx950 -O3 -o variants/grid_barrier tools/ubench/grid_barrier.hip && variants/grid_barrier
//
// One workgroup of 512 threads per CU (96 KiB of LDS each: never two on a CU, all co-resident), R rounds.  In a round every
// workgroup writes its 16 values of a 4096-value vector, all workgroups meet, every workgroup reads the WHOLE vector (what a GEMV
// stage needs of its predecessor) and checks it.  Variants of the hand-over:
//   fence    plain stores, __threadfence(), atomic counter, spin, __threadfence(), plain loads   (the textbook grid barrier)
//   atomics  agent-scope atomic stores (write-through), vmcnt(0), atomic counter, spin, agent-scope atomic loads -- no fence
//            (rope_attn_online_kernel's split hand-over, any4_amd/csrc/decode_glue.cuh)
//   + work   the same with 8 MiB of cold weights streamed per round (32 KiB per workgroup), requested BEFORE the barrier of the
//            previous round is passed: the overlap a graph of dependent nodes cannot have
// Every spin is bounded (an error flag is raised instead of a hang).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  float* vec;          // [2][4096] double-buffered vector
  unsigned* counter;   // monotonically increasing arrival counter
  unsigned* err;
  const u32x4* w;      // [rounds][wgs][4][512] cold weights (work variants)
  float* sink;
  int rounds, wgs, mode, work;
};

template <int MODE>  // 0 fence, 1 atomics
__global__ void __launch_bounds__(512) barrier_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, b = blockIdx.x, n = p.wgs * 16;
  float acc = 0.f;
  u32x4 ring[4];
  if (p.work) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = __builtin_nontemporal_load(p.w + ((size_t)b * 4 + i) * 512 + t);
  }
  for (int r = 0; r < p.rounds; ++r) {
    float* cur = p.vec + (size_t)(r & 1) * n;
    // ---- this workgroup's 16 values of round r (depend on what it read in round r - 1) ----
    if (t < 16) {
      const float v = (float)(r * 4096 + b * 16 + t) + acc * 0.f;
      if (MODE == 0) cur[b * 16 + t] = v;
      else __hip_atomic_store(cur + b * 16 + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 0) __threadfence();
    else __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    // ---- all workgroups meet ----
    if (t == 0) {
      __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(r + 1) * (unsigned)p.wgs;
      unsigned spins = 0;
      while (__hip_atomic_load(p.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000u) { *p.err = 1; break; }
      }
    }
    __syncthreads();
    if (MODE == 0) __threadfence();
    // ---- the next stage's weights are requested before the vector is read (work variants): they do not depend on it ----
    u32x4 nxt[4];
    if (p.work) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc += __builtin_bit_cast(float, ring[i][0] ^ ring[i][1] ^ ring[i][2] ^ ring[i][3]);
        const int rr = r + 1 < p.rounds ? r + 1 : r;
        nxt[i] = __builtin_nontemporal_load(p.w + (((size_t)rr * p.wgs + b) * 4 + i) * 512 + t);
      }
    }
    // ---- every workgroup reads the whole vector ----
    for (int i = t; i < n; i += 512) {
      float v;
      if (MODE == 0) v = cur[i];
      else v = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)(r * 4096 + i)) *p.err = 2;
      sx[i] = v;
    }
    __syncthreads();
    acc += sx[(t * 7) % n];
    if (p.work) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ring[i] = nxt[i];
    }
    __syncthreads();
  }
  if (acc == 123.456f) p.sink[b] = acc;
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int wgs = prop.multiProcessorCount, rounds = 200;
  P p{};
  CHECK(hipMalloc(&p.vec, 2 * 4096 * sizeof(float) * 2));
  CHECK(hipMalloc(&p.counter, 64));
  CHECK(hipMalloc(&p.err, 64));
  CHECK(hipMalloc(&p.sink, 4096));
  const size_t wbytes = (size_t)rounds * wgs * 4 * 512 * 16;
  CHECK(hipMalloc((void**)&p.w, wbytes));
  CHECK(hipMemset((void*)p.w, 1, wbytes));
  p.rounds = rounds; p.wgs = wgs;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(barrier_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(barrier_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  printf("%d workgroups (one per CU), %d rounds per launch\n", wgs, rounds);
  for (int work = 0; work < 2; ++work)
    for (int mode = 0; mode < 2; ++mode) {
      p.work = work; p.mode = mode;
      float best = 1e30f;
      unsigned err = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(p.counter, 0, 64));
        CHECK(hipMemset(p.err, 0, 64));
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(wgs), dim3(512), 96 * 1024, 0, p);
        else hipLaunchKernelGGL(barrier_kernel<1>, dim3(wgs), dim3(512), 96 * 1024, 0, p);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned e;
        CHECK(hipMemcpy(&e, p.err, 4, hipMemcpyDeviceToHost));
        err |= e;
      }
      printf("%-10s %-22s %8.2f us per round%s\n", mode == 0 ? "fence" : "atomics", work ? "+ 8 MiB cold per round" : "hand-over only", best * 1000.f / rounds,
             err == 1 ? "   [a spin gave up]" : err ? "   [WRONG VALUES READ]" : "");
      fflush(stdout);
    }
  return 0;
}
