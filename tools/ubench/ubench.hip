// Developer micro-benchmarks for instruction throughput on gfx950 (not part of the product).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef const __attribute__((address_space(3))) float* lds_cfptr;

#define N_UNROLL 32
template <int KIND>
__global__ void __launch_bounds__(1024) k(uint32_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  __shared__ float tab[16 * 64 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = 0; e < 16; ++e) tab[wave * 1024 + e * 64 + lane] = (float)e;
  __syncthreads();
  uint32_t v[N_UNROLL];
  float f[N_UNROLL];
  for (int i = 0; i < N_UNROLL; ++i) { v[i] = seed * (i + 1) + threadIdx.x; f[i] = (float)v[i]; }
  const uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>(&tab[wave * 1024]) | (lane * 4);
  f32x4 acc = {0, 0, 0, 0};
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N_UNROLL; ++i) {
      if (KIND == 0) v[i] = __builtin_amdgcn_perm(v[i], base, 0x03020400u);
      if (KIND == 1) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);
      if (KIND == 2) { if (i & 1) { f32x2 a = {f[i - 1], f[i]}; f32x2 r = a * f32x2{1.0001f, 1.0002f} + f32x2{0.5f, 0.25f}; f[i - 1] = r[0]; f[i] = r[1]; } }
      if (KIND == 3) { if (i & 1) { f32x2 a = {f[i - 1], f[i]}; v[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2)); f[i] = __builtin_bit_cast(float, v[i]) ; } }
      if (KIND == 4) { uint32_t a = (v[i] & 0xf00u) | base; f[i] = *(lds_cfptr)(a); v[i] = __builtin_bit_cast(uint32_t, f[i]) + v[i]; }
      if (KIND == 5) v[i] = (v[i] & 0x0f0f0f0fu) | seed;
      if (KIND == 7) { u32x4 r = *(volatile __attribute__((address_space(3))) u32x4*)(uintptr_t)(((wave * 1024 + lane * 4) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r[0]; }
      if (KIND == 8) { if ((lane & 15) == 0) { u32x4 r = *(volatile __attribute__((address_space(3))) u32x4*)(uintptr_t)(((wave * 1024 + lane * 4) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r[0]; } }
      if (KIND == 9) { u32x4 r = *(volatile __attribute__((address_space(3))) u32x4*)(uintptr_t)(((wave * 1024 + (lane >> 4) * 4) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r[0]; }
      if (KIND == 10) { if (lane < 16) { u32x4 r = *(volatile __attribute__((address_space(3))) u32x4*)(uintptr_t)(((wave * 1024 + lane * 4) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r[0]; } }
      if (KIND == 11) { if ((lane & 15) < 4) { u32x4 r = *(volatile __attribute__((address_space(3))) u32x4*)(uintptr_t)(((wave * 1024 + lane * 4) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r[0]; } }
      if (KIND == 12) { uint32_t r = *(volatile __attribute__((address_space(3))) uint32_t*)(uintptr_t)(((wave * 1024 + lane) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r; }
      if (KIND == 13) { if ((lane & 15) == 0) { uint32_t r = *(volatile __attribute__((address_space(3))) uint32_t*)(uintptr_t)(((wave * 1024 + lane) * 4 + (i & 3) * 1024) & 0xffff); v[i] ^= r; } }
      if (KIND == 14) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(v[i]) : "v"(v[(i + 1) % N_UNROLL]));
      if (KIND == 15) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(v[i]) : "v"(v[i]), "v"(base));
      if (KIND == 16) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(v[i]) : "v"(8u), "v"(v[i]));
      if (KIND == 6) { if ((i & 7) == 7) { u32x4 a = {v[i], v[i-1], v[i-2], v[i-3]}; u32x4 b = {v[i-4], v[i-5], v[i-6], v[i-7]}; acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0); } }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < N_UNROLL; ++i) s += v[i] + __builtin_bit_cast(uint32_t, f[i]);
  s += __builtin_bit_cast(uint32_t, acc[0] + acc[1] + acc[2] + acc[3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

// LDS read throughput: 16 independent reads in flight per wave, one s_waitcnt per block of 16.
template <int WIDTH, int MASK>
__global__ void __launch_bounds__(1024) kl(uint32_t* out, uint64_t* cyc, int iters, uint32_t seed) {
  __shared__ float tab[16 * 64 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = 0; e < 16; ++e) tab[wave * 1024 + e * 64 + lane] = (float)e;
  __syncthreads();
  uint32_t addr = (uint32_t)reinterpret_cast<uintptr_t>(&tab[0]) + wave * 4096;
  if (MASK == 2) addr += (lane >> 4) * 16;            // broadcast within each 16-lane row
  else addr += lane * (WIDTH == 128 ? 16 : WIDTH == 64 ? 8 : 4);
  const bool on = MASK == 1 ? (lane & 15) == 0 : MASK == 3 ? lane < 16 : MASK == 4 ? (lane & 15) < 4 : true;
  uint32_t s = 0;
  uint64_t t0 = __builtin_readcyclecounter();
  if (on) {
    for (int it = 0; it < iters; ++it) {
      u32x4 r[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (WIDTH == 128) asm volatile("ds_read_b128 %0, %1" : "=v"(r[i]) : "v"(addr + (i & 3) * 1024));
        if (WIDTH == 64) { uint64_t t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(addr + (i & 3) * 1024)); r[i][0] = (uint32_t)t; }
        if (WIDTH == 32) asm volatile("ds_read_b32 %0, %1" : "=v"(r[i][0]) : "v"(addr + (i & 3) * 1024));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) s ^= r[i][0];
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int WIDTH, int MASK>
void runl(const char* name) {
  uint32_t* out; uint64_t* cyc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kl<WIDTH, MASK>), dim3(blocks), dim3(1024), 0, 0, out, cyc, 10, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kl<WIDTH, MASK>), dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr_per_cu = 16.0 * iters * 16;
  printf("%-34s wall %.3f ms -> %.2f ns per wave-instr per CU\n", name, ms, ms * 1e6 / instr_per_cu);
  hipFree(out); hipFree(cyc);
}

template <int KIND>
void run(const char* name, int per_iter_instr) {
  uint32_t* out; uint64_t* cyc;
  const int blocks = 256, iters = 2000;
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, out, cyc, 10, 3u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters, 3u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256 * 16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256 * 16; ++i) avg += h[i]; avg /= 256 * 16;
  // 16 waves per CU = 4 per SIMD; per SIMD instrs = 4 waves * iters * per_iter_instr
  double instr_per_simd = 4.0 * iters * per_iter_instr;
  printf("%-28s wall %.3f ms  counter/wave %.0f  -> %.2f counter-ticks per wave-instr per SIMD, %.2f ns per instr per SIMD\n", name, ms, avg,
         avg / instr_per_simd, ms * 1e6 / instr_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("clockRate %d kHz, CUs %d\n", pr.clockRate, pr.multiProcessorCount);
  run<0>("v_perm_b32", 32);
  run<1>("v_fma_f32", 32);
  run<2>("v_pk_fma_f32", 16);
  run<3>("v_cvt_pk_bf16_f32", 16);
  run<4>("and_or+ds_read_b32+add", 32);
  run<5>("v_and_or", 32);
  run<6>("mfma16x16x32bf16", 4);
  run<14>("v_mov_b32_sdwa preserve", 32);
  run<15>("v_or_b32_sdwa byte sel", 32);
  run<16>("v_lshlrev_b32_sdwa byte sel", 32);
  runl<128, 0>("ds_read_b128 full");
  runl<128, 1>("ds_read_b128 lanes%16==0");
  runl<128, 2>("ds_read_b128 bcast per 16-lane row");
  runl<128, 3>("ds_read_b128 lanes<16");
  runl<128, 4>("ds_read_b128 lanes%16<4");
  runl<64, 0>("ds_read_b64 full");
  runl<64, 1>("ds_read_b64 lanes%16==0");
  runl<32, 0>("ds_read_b32 full");
  runl<32, 1>("ds_read_b32 lanes%16==0");
  return 0;
}
