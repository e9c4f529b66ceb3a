#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void empty_k(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void touch_k(const uint4* w, uint4* out, int n16) {
  // every lane streams n16 16-byte pieces (strided by grid), like a GEMV weight stream
  uint4 acc = {0, 0, 0, 0};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
    uint4 v = w[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;
}
int main() {
  const size_t FL = 1ull << 30;
  char* flush; hipMalloc(&flush, FL);
  uint4* w; hipMalloc(&w, 64 << 20); hipMemset(w, 1, 64 << 20);
  uint4* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 6; ++mode) {
    float tot = 0; const int iters = 30;
    for (int it = 0; it < iters + 5; ++it) {
      hipMemsetAsync(flush, it, FL, 0);
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(empty_k, dim3(256), dim3(64), 0, 0, nullptr);
      if (mode == 1) hipLaunchKernelGGL(empty_k, dim3(256), dim3(1024), 0, 0, nullptr);
      if (mode == 2) hipLaunchKernelGGL(touch_k, dim3(256), dim3(1024), 0, 0, w, out, (9 << 20) / 16);
      if (mode == 3) hipLaunchKernelGGL(touch_k, dim3(2048), dim3(256), 0, 0, w, out, (9 << 20) / 16);
      if (mode == 4) hipLaunchKernelGGL(touch_k, dim3(256), dim3(1024), 0, 0, w, out, (32 << 20) / 16);
      if (mode == 5) {}
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it >= 5) tot += ms;
    }
    const char* names[] = {"empty 256x64", "empty 256x1024", "stream 9 MiB 256x1024", "stream 9 MiB 2048x256", "stream 32 MiB 256x1024", "no kernel (event pair only)"};
    printf("%-28s %.2f us\n", names[mode], tot / iters * 1e3);
  }
  return 0;
}
