// Developer micro-benchmark: LDS STORE throughput of one 8-wave workgroup per CU on gfx950, for the table build / split-K tail of the
// pair-table kernels (not part of the product).  Every wave writes 8 KiB per round (its own region of a 64 KiB buffer), ROUNDS
// rounds, with: ds_write_b32 | ds_write2st64_b32 | ds_write_b64 | ds_write_b128 | ds_write_addtid_b32 (address = M0 + imm + 4 lane).
// Also checks ds_write_addtid_b32's addressing beyond 64 KiB (M0[15:0] + 16-bit immediate) by reading the buffer back.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_store.hip -o /tmp/lds_store && /tmp/lds_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) uint32_t* lds_u32ptr;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) u32x2* lds_u32x2ptr;
typedef __attribute__((address_space(3))) u32x4* lds_u32x4ptr;

template <int KIND>
__global__ void __launch_bounds__(512) k(uint32_t* out, uint64_t* cyc, int rounds, uint32_t seed) {
  extern __shared__ uint32_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = seed * (i + 1) + threadIdx.x;
  __syncthreads();
  const uint32_t base = (uint32_t)(wave * 8192 + lane * 4);
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    // (every kind as inline asm: plain C++ stores that the next round overwrites are dead-store-eliminated -- the first version of
    //  this benchmark reported 475-504 GB/s for kinds that wrote nothing)
    if constexpr (KIND == 0) {  // 32 x ds_write_b32, 256 bytes apart
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(v[i]), "n"(i * 256) : "memory");
    } else if constexpr (KIND == 1) {  // 16 x ds_write2st64_b32
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base), "v"(v[2 * i]), "v"(v[2 * i + 1]), "n"(2 * i), "n"(2 * i + 1) : "memory");
    } else if constexpr (KIND == 2) {  // 16 x ds_write_b64 (lane-contiguous 8 bytes)
      const uint32_t b2 = (uint32_t)(wave * 8192 + lane * 8);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const u32x2 d = {v[2 * i], v[2 * i + 1]};
        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(b2), "v"(d), "n"(i * 512) : "memory");
      }
    } else if constexpr (KIND == 3) {  // 8 x ds_write_b128
      const uint32_t b4 = (uint32_t)(wave * 8192 + lane * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x4 d = {v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(b4), "v"(d), "n"(i * 1024) : "memory");
      }
    } else {  // 32 x ds_write_addtid_b32: M0 = wave base, immediate = 256 i
      asm volatile("s_mov_b32 m0, %0" ::"s"(wave * 8192) : "memory");
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(v[i]), "n"(i * 256) : "memory");
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] += 0x01010101u;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_readcyclecounter();
  __syncthreads();
  uint32_t s = 0;
  for (int i = 0; i < 32; ++i) s += smem[wave * 2048 + i * 64 + lane];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// addressing check of ds_write_addtid_b32 across the 64 KiB line: M0 = 0xFFFC (65532), immediates up to 65284
__global__ void __launch_bounds__(64) addr_check(uint32_t* out) {
  extern __shared__ uint32_t smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 40960; i += 64) smem[i] = 0u;
  __syncthreads();
  const uint32_t val = 0xabc00000u + lane;
  asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:65284\n\tds_write_addtid_b32 %1 offset:4\n\ts_waitcnt lgkmcnt(0)" ::"s"(65532), "v"(val) : "memory");
  asm volatile("s_mov_b32 m0, %0\n\tds_write_addtid_b32 %1 offset:1024\n\ts_waitcnt lgkmcnt(0)" ::"s"(0x12340000 + 512), "v"(val + 0x100) : "memory");  // (only M0[15:0] counts)
  __syncthreads();
  out[lane] = smem[(65532 + 65284) / 4 + lane];
  out[64 + lane] = smem[(65532 + 4) / 4 + lane];
  out[128 + lane] = smem[(512 + 1024) / 4 + lane];
}

template <int KIND>
static void run(const char* name, uint32_t* out, uint64_t* cyc, int rounds) {
  hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<KIND><<<256, 512, 65536>>>(out, cyc, rounds, 3u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<KIND><<<256, 512, 65536>>>(out, cyc, rounds, 5u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[2048];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < 2048; ++i) mx = h[i] > mx ? h[i] : mx;
  // s_memtime ticks at 100 MHz on this part; report bytes per clock from the event time at the clock the chip ran
  printf("%-22s %8.3f ms  %6.1f GB/s per CU   (%d rounds x 64 KiB per CU)\n", name, ms, rounds * 65536.0 / (ms * 1e-3) / 1e9, rounds);
}

int main() {
  uint32_t* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2048 * 8);
  const int rounds = 20000;
  run<0>("ds_write_b32", out, cyc, rounds);
  run<1>("ds_write2st64_b32", out, cyc, rounds);
  run<2>("ds_write_b64", out, cyc, rounds);
  run<3>("ds_write_b128", out, cyc, rounds);
  run<4>("ds_write_addtid_b32", out, cyc, rounds);
  hipFuncSetAttribute((const void*)addr_check, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  addr_check<<<1, 64, 163840>>>(out);
  uint32_t h[192];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += (h[l] != 0xabc00000u + l) + (h[64 + l] != 0xabc00000u + l) + (h[128 + l] != 0xabc00100u + l);
  printf("ds_write_addtid_b32 addressing (M0[15:0] + imm16 + 4 lane, beyond 64 KiB): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
  return bad != 0;
}
