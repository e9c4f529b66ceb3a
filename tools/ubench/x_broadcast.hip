// x_broadcast.hip -- what does it cost when EVERY compute unit reads the same 128 KiB activation block (16 rows x 4096 x 2 bytes) at the
// start of a launch, and does the access pattern matter?
//
//   hipcc --offload-arch=gfx950 -O3 -o variants/x_broadcast tools/ubench/x_broadcast.hip && variants/x_broadcast
//
// A hipGraph of 64 dependent nodes, 256 workgroups x 1024 threads (16 waves), every workgroup reads all of x with 8 loads of 16 bytes per
// lane and writes one dword.  Patterns (which 16 bytes lane L of wave w reads in load j):
//   rows     w4_gemm_pair16_kernel's XREG: row L & 15, k-slice w (512 bytes), chunk j (64 bytes), quarter L >> 4: a wave-load touches 16
//            rows x 64 bytes, 8 KiB apart
//   rows-rot the same with the chunk order rotated by the row: chunk (j + (L & 15)) & 7
//   runs     contiguous: wave w reads bytes [(w * 8 + j) * 1024, + 1024) -- what a pre-arranged x would allow (or a transposing reader)
//   runs-cu  `runs` with the start rotated by the workgroup index (every CU starts somewhere else in x)
//   rows-cu  `rows` with the k-slice -> wave assignment rotated by the workgroup index
//   rows8x128 / rows4x256   a wave-load touches 8 rows x 128 bytes / 4 rows x 256 bytes (a lane bit <-> register bit exchange away from `rows`)
// and the same for 64 KiB (8 rows: rows 8..15 re-read rows 0..7).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) node(const char* __restrict__ x, uint32_t* __restrict__ out, int rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
  u32x4 v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t off;
    const int r = (lane & 15) % rows, q = lane >> 4;
    if (MODE == 0) off = r * 8192 + w * 512 + j * 64 + q * 16;
    else if (MODE == 1) off = r * 8192 + w * 512 + ((j + r) & 7) * 64 + q * 16;
    else if (MODE == 2) off = (((w * 8 + j) * 1024) % (rows * 8192)) + lane * 16;
    else if (MODE == 3) off = ((((w * 8 + j + b * 5) & 127) * 1024) % (rows * 8192)) + lane * 16;
    else if (MODE == 4) off = r * 8192 + ((w + b) & 15) * 512 + j * 64 + q * 16;
    else if (MODE == 5) off = ((8 * (j & 1) + (lane & 7)) % rows) * 8192 + w * 512 + (2 * (j >> 1) + ((lane >> 3) & 1)) * 64 + q * 16;   // 8 rows x 128 bytes per wave-load
    else if (MODE == 6) off = ((4 * (j & 3) + (lane & 3)) % rows) * 8192 + w * 512 + (4 * (j >> 2) + ((lane >> 2) & 3)) * 64 + q * 16;   // 4 rows x 256 bytes
    else if (MODE == 7) off = ((2 * j + (lane >> 5)) % rows) * 8192 + w * 512 + (lane & 31) * 16;                                          // 2 rows x 512 bytes: the wave's slice of rows 2 j, 2 j + 1
    else if (MODE == 8) off = (((lane & 15) + w) % rows) * 8192 + w * 512 + j * 64 + q * 16;                                               // `rows`, the row order rotated by the wave
    else if (MODE == 9) off = (w % rows) * 8192 + ((j + w) & 7) * 1024 + lane * 16;                                                        // `runs` (wave w = row w), start rotated by the wave
    else if (MODE == 10) off = r * 8192 + w * 512 + j * 64 + q * 16 + ((b & 7) * 131072);                                                  // `rows`, a private copy of x per XCD
    else if (MODE == 11) off = ((lane >> 2) % rows) * 8192 + w * 512 + j * 64 + (lane & 3) * 16;                                           // 16 rows x 64 bytes, the four lanes of a QUAD contiguous
    else if (MODE == 12) off = ((lane >> 2) % rows) * 8192 + w * 512 + j * 64 + ((lane & 1) * 2 + ((lane >> 1) & 1)) * 16;                // ... the quad's 64 bytes in the order 0, 2, 1, 3
    else if (MODE == 13) off = (w * 8 + j) * 1024 + (lane & 31) * 32 + (lane >> 5) * 16;                                                   // a 1 KiB run, adjacent lanes 32 bytes apart (the weight loads of w4_gemm_xr / w4_gemv: lane = row, 16 of its 32 bytes)
    else if (MODE == 14) off = (w * 8 + j) * 1024 + (lane & 15) * 64 + (lane >> 4) * 16;                                                   // a 1 KiB run, adjacent lanes 64 bytes apart
    else off = (w * 8 + j) * 1024 + (lane >> 1) * 32 + (lane & 1) * 16;                                                                    // a 1 KiB run, PAIRS of lanes contiguous (32 bytes), pairs in order
    v[j] = *reinterpret_cast<const u32x4*>(x + off);
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
  if (s == 0x12345678u) out[b] = s;
}

int main() {
  char* x; uint32_t* out;
  CHECK(hipMalloc(&x, 2 << 20)); CHECK(hipMemset(x, 1, 2 << 20));
  CHECK(hipMalloc(&out, 4096));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const char* names[16] = {"rows", "rows-rot", "runs", "runs-cu", "rows-cu", "rows8x128", "rows4x256", "rows2x512", "rows-wrot", "runs-wrot", "rows-xcdcopy", "rows-quad", "rows-quad-0213", "run-stride32", "run-stride64", "run-pairs"};
  for (int rows : {16}) {
    for (int mode = 0; mode < 16; ++mode) {
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 64; ++i) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(node<0>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 1: hipLaunchKernelGGL(node<1>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 2: hipLaunchKernelGGL(node<2>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 3: hipLaunchKernelGGL(node<3>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 4: hipLaunchKernelGGL(node<4>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 5: hipLaunchKernelGGL(node<5>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 6: hipLaunchKernelGGL(node<6>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 7: hipLaunchKernelGGL(node<7>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 8: hipLaunchKernelGGL(node<8>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 9: hipLaunchKernelGGL(node<9>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 10: hipLaunchKernelGGL(node<10>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 11: hipLaunchKernelGGL(node<11>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 12: hipLaunchKernelGGL(node<12>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 13: hipLaunchKernelGGL(node<13>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          case 14: hipLaunchKernelGGL(node<14>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
          default: hipLaunchKernelGGL(node<15>, dim3(256), dim3(1024), 0, st, x, out, rows); break;
        }
      }
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int it = 0; it < 3; ++it) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      for (int it = 0; it < 20; ++it) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%2d rows (%3d KiB per CU)  %-9s %6.2f us per node\n", rows, rows * 8, names[mode], ms * 1e3 / 20 / 64);
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
  }
  return 0;
}
