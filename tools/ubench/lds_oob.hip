// Probe: what does a ds_read beyond the workgroup's LDS allocation return on gfx950?  (ISA: out-of-range LDS reads return 0.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_oob.hip -o /tmp/lds_oob ; prints the values read at several offsets.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) const u32x4* lds_cu32x4ptr;
typedef __attribute__((address_space(3))) uint32_t* lds_u32ptr;
__global__ void probe(uint32_t* out, uint32_t lds_bytes) {
  extern __shared__ uint32_t dyn[];
  for (uint32_t i = threadIdx.x; i < lds_bytes / 4; i += blockDim.x) ((lds_u32ptr)0)[i] = 0xdead0000u + i;
  __syncthreads();
  const uint32_t offs[6] = {0u, lds_bytes - 16u, lds_bytes, lds_bytes + 64u, lds_bytes + 4096u, 163840u - 16u};
  for (int j = 0; j < 6; ++j) {
    const u32x4 v = *(lds_cu32x4ptr)(offs[j]);
    if (threadIdx.x == 0) { out[4 * j] = v[0]; out[4 * j + 1] = v[1]; out[4 * j + 2] = v[2]; out[4 * j + 3] = v[3]; }
  }
}
int main() {
  uint32_t* d; hipMalloc(&d, 256); uint32_t h[24];
  for (uint32_t lds : {1024u, 65536u, 81920u}) {
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // two workgroups so that a neighbour's allocation may sit behind ours
    hipLaunchKernelGGL(probe, dim3(512), dim3(256), lds, 0, d, lds);
    hipMemcpy(h, d, 96, hipMemcpyDeviceToHost);
    printf("lds=%u:", lds);
    for (int j = 0; j < 6; ++j) printf("  [%d] %08x %08x %08x %08x", j, h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]);
    printf("\n");
  }
  return 0;
}
