// Developer microbenchmark: cost of one ds_read_b128 wave-instruction for a given per-lane address pattern
// (which lanes the LDS serves together is not documented; this measures it).  Not part of the product.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void __launch_bounds__(1024) k(const uint32_t* addrs, uint32_t* out, int iters) {
  __shared__ float tab[16 * 1024];
  for (int e = threadIdx.x; e < 16 * 1024; e += 1024) tab[e] = (float)e;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t a = addrs[lane] + wave * 4096u;  // every wave its own 4 KiB window (same bank phase)
  uint32_t s = 0;
  for (int it = 0; it < iters; ++it) {
    u32x4 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(a));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 16; ++i) s ^= r[i][0];
  }
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
static float run(const std::vector<uint32_t>& h) {
  uint32_t *d, *o; hipMalloc(&d, 256); hipMalloc(&o, 256 * 1024 * 4);
  hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, o, 10);
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, o, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(d); hipFree(o);
  return ms * 1e6f / (16.0f * 2000 * 16);  // ns per wave-instruction per CU
}
int main() {
  const int RS = 272;  // slab row stride of the Bint4 stream kernel
  printf("linear lane*16: %.2f ns\n", run([] { std::vector<uint32_t> v(64); for (int l = 0; l < 64; ++l) v[l] = l * 16; return v; }()));
  printf("all same      : %.2f ns\n", run(std::vector<uint32_t>(64, 0)));
  for (int m : {1, 2, 4, 8, 12, 16}) {
    printf("m=%2d rows [Q][m], zero row at slot z: ", m);
    for (int z = 0; z < 16; ++z) {
      std::vector<uint32_t> v(64);
      const uint32_t zbase = 3584;  // 256-aligned zero area
      for (int l = 0; l < 64; ++l) { int j = l & 15, Q = l >> 4; v[l] = j < m ? (Q * m + j) * RS : zbase + z * 16; }
      printf("%.2f ", run(v));
    }
    printf("\n");
  }
  // old order [act row][Q] for comparison (m = 8, zero row behind the rows)
  { std::vector<uint32_t> v(64); for (int l = 0; l < 64; ++l) { int j = l & 15, Q = l >> 4; v[l] = j < 8 ? (j * 4 + Q) * RS : 32 * RS; } printf("m= 8 old order [row][Q], zero row after: %.2f ns\n", run(v)); }
  return 0;
}
