// graph_chain.hip -- what does ONE dependent weight-streaming node of a batch-1 decode step cost on MI355X?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_chain tools/ubench/graph_chain.hip && /tmp/graph_chain
//
// A hipGraph of N dependent "GEMV-shaped" kernels: 256 workgroups (one per CU) x 512 threads, every workgroup streams its own
// contiguous slice of a weight buffer with non-temporal 16-byte loads (everything it can request up front, 8 loads per lane in
// flight), reads the 8 KiB activation vector the PREVIOUS node wrote, reduces and writes 16 outputs.  Reported: microseconds
// per node, for
//   cold   every node streams a different buffer (HBM)
//   warm   every node streams the same buffer (what the caches keep across a kernel boundary: the upper bound of any prefetch)
//   pf     cold, but node i also touches the first PF bytes per workgroup of node i + 1's slice after its own loads are issued
//          (a prefetch into whatever cache level survives the boundary)
// and for an empty node and a node that only does the activation hand-off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  const u32x4* w;       // this node's weights
  const u32x4* w_next;  // next node's weights (prefetch target) or null
  const uint32_t* x_in;
  uint32_t* x_out;
  int pieces;           // 16-byte pieces per thread
  int pf_pieces;        // prefetch pieces per thread (<= 4)
  int pf_stride;        // pieces per thread of the next node's slice
  int nt;
  int wg_stride;        // pieces per thread between the slices of consecutive workgroups (0: all read one slice -> cache-hot)
};

__device__ __forceinline__ u32x4 ld_nt(const u32x4* p) { return __builtin_nontemporal_load(p); }

template <int R>
__global__ void __launch_bounds__(512) node_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, b = blockIdx.x;
  // slice of this workgroup: pieces * 512 consecutive 16-byte pieces; lane-contiguous within a load
  const u32x4* base = p.w + (size_t)b * p.wg_stride * 512 + t;
  u32x4 ring[R];
  const int n = p.pieces;  // a multiple of R
#pragma unroll
  for (int i = 0; i < R; ++i) ring[i] = p.nt ? ld_nt(base + (size_t)i * 512) : base[(size_t)i * 512];
  // activation vector of the previous node: 4096 x 2 bytes = 512 threads x 16 bytes
  const u32x4 xv = reinterpret_cast<const u32x4*>(p.x_in)[t];
  u32x4 pfv[4] = {};
  if (p.w_next) {  // (uniform)
    const u32x4* nb = p.w_next + (size_t)b * p.pf_stride * 512 + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) pfv[i] = nb[(size_t)(i < p.pf_pieces ? i : 0) * 512];
  }
  float acc = 0.f;
  for (int i = 0; i < n; i += R) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const u32x4 v = ring[j];
      acc += __builtin_bit_cast(float, (v[0] ^ xv[0]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[1] ^ xv[1]) & 0x3fffffffu) +
             __builtin_bit_cast(float, (v[2] ^ xv[2]) & 0x3fffffffu) + __builtin_bit_cast(float, (v[3] ^ xv[3]) & 0x3fffffffu);
      int nx = i + j + R;  // unconditional refill (a conditional one makes hipcc wait vmcnt(0) at the next use): past the end re-reads the last piece
      nx = nx < n ? nx : n - 1;
      ring[j] = p.nt ? ld_nt(base + (size_t)nx * 512) : base[(size_t)nx * 512];
    }
  }
  uint32_t junk = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) junk ^= ring[j][0];
#pragma unroll
  for (int i = 0; i < 4; ++i) junk ^= pfv[i][1];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((t & 63) == 0) red[t >> 6] = acc;
  __syncthreads();
  if (t < 8) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += red[i];
    p.x_out[(b & 255) * 8 + t] = __builtin_bit_cast(uint32_t, s) | 0x00010001u | (junk == 0x12345u ? 2u : 0u);
  }
}

// touches its slice like dg_prefetch / the prefetch blocks of rope_attn_online_kernel: 4-byte reads of every 16-byte piece, default policy or nt
template <int NT>
__global__ void __launch_bounds__(512) touch_kernel(const P p) {
  const int t = threadIdx.x, b = blockIdx.x;
  const u32x4* base = p.w + (size_t)b * p.wg_stride * 512 + t;
  uint32_t acc = 0;
  for (int i = 0; i < p.pieces; ++i) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(base + (size_t)i * 512);
    acc ^= NT ? __builtin_nontemporal_load(q) : *q;
  }
  if (acc == 0x12345u) p.x_out[0] = acc;
}

__global__ void __launch_bounds__(512) empty_kernel(const P p) {
  if (p.pieces == -1) p.x_out[0] = 0;
}

int main(int argc, char** argv) {
  const int N = 64;
  std::vector<u32x4*> w(N + 1);
  const size_t maxbytes = 64u << 20;
  for (int i = 0; i <= N; ++i) { CHECK(hipMalloc(&w[i], maxbytes)); CHECK(hipMemset(w[i], 0x11 + i, maxbytes)); }
  uint32_t* x[2];
  CHECK(hipMalloc(&x[0], 8192)); CHECK(hipMalloc(&x[1], 8192));
  CHECK(hipMemset(x[0], 1, 8192)); CHECK(hipMemset(x[1], 1, 8192));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(node_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  struct Mode { const char* name; int kind; size_t bytes; int warm; size_t pf; int nt; unsigned lds; int ring = 8; int wgs = 256; int touch = 0; };
  std::vector<Mode> modes = {
      {"empty node", 0, 0, 0, 0, 1, 0},
      {"hand-off only (x in, x out)", 1, 0, 0, 0, 1, 64},
      {"hand-off only, 64 KiB dynamic LDS", 1, 0, 0, 0, 1, 65536},
      {"8 MiB cold nt", 1, 8u << 20, 0, 0, 1, 64},
      {"8 MiB cold nt, 64 KiB LDS", 1, 8u << 20, 0, 0, 1, 65536},
      {"8 MiB cold default policy", 1, 8u << 20, 0, 0, 0, 64},
      {"8 MiB warm (same buffer) nt", 1, 8u << 20, 1, 0, 1, 64},
      {"8 MiB warm (same buffer) default", 1, 8u << 20, 1, 0, 0, 64},
      {"8 MiB cold default + prefetch next 8 MiB", 1, 8u << 20, 0, 8u << 20, 0, 64},
      {"8 MiB cold nt + prefetch next 8 MiB", 1, 8u << 20, 0, 8u << 20, 1, 64},
      {"16 MiB cold nt", 1, 16u << 20, 0, 0, 1, 64},
      {"32 MiB cold nt", 1, 32u << 20, 0, 0, 1, 64},
      {"32 MiB cold nt + prefetch next 8 MiB", 1, 32u << 20, 0, 8u << 20, 1, 64},
      {"32 MiB cold default + prefetch next 8 MiB", 1, 32u << 20, 0, 8u << 20, 0, 64},
      {"64 MiB cold nt", 1, 64u << 20, 0, 0, 1, 64},
      {"64 MiB cold default", 1, 64u << 20, 0, 0, 0, 64},
      {"64 MiB cold nt + prefetch next 8 MiB", 1, 64u << 20, 0, 8u << 20, 1, 64},
      // (round 5) 16 loads of 16 bytes per lane in flight instead of 8: is a long node latency-bound at 64 KiB per CU in flight?
      {"16 MiB cold nt, 16 loads in flight", 1, 16u << 20, 0, 0, 1, 64, 16},
      {"32 MiB cold nt, 16 loads in flight", 1, 32u << 20, 0, 0, 1, 64, 16},
      {"64 MiB cold nt, 16 loads in flight", 1, 64u << 20, 0, 0, 1, 64, 16},
      // (round 5) two workgroups per CU, 8 loads in flight each
      {"8 MiB cold nt, 512 workgroups", 1, 8u << 20, 0, 0, 1, 64, 8, 512},
      {"32 MiB cold nt, 512 workgroups", 1, 32u << 20, 0, 0, 1, 64, 8, 512},
      {"64 MiB cold nt, 512 workgroups", 1, 64u << 20, 0, 0, 1, 64, 8, 512},
      {"64 MiB cold nt, 1024 workgroups", 1, 64u << 20, 0, 0, 1, 64, 8, 1024},
      // (round 5) a touch-only node (4-byte reads of every 16-byte piece, same slices) in FRONT of every streaming node: the PAIR's time
      {"touch (default) + 8 MiB nt stream: the pair", 1, 8u << 20, 0, 0, 1, 64, 8, 256, 1},
      {"touch (nt) + 8 MiB nt stream: the pair", 1, 8u << 20, 0, 0, 1, 64, 8, 256, 2},
      {"touch (default) + 8 MiB default stream: the pair", 1, 8u << 20, 0, 0, 0, 64, 8, 256, 1},
      {"touch only (default), 8 MiB", 2, 8u << 20, 0, 0, 0, 64, 8, 256, 1},
  };
  for (const Mode& m : modes) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
      P p;
      p.w = m.warm ? w[0] : w[i];
      p.w_next = m.pf ? w[i + 1] : nullptr;
      p.x_in = x[i & 1]; p.x_out = x[(i + 1) & 1];
      p.pieces = (int)(m.bytes / 16 / 512 / m.wgs);
      p.pf_pieces = (int)(m.pf / 16 / 512 / 256);
      p.pf_stride = p.pieces;
      p.wg_stride = p.pieces;
      p.nt = m.nt;
      if (m.touch == 1) hipLaunchKernelGGL(touch_kernel<0>, dim3(m.wgs), dim3(512), 0, st, p);
      if (m.touch == 2) hipLaunchKernelGGL(touch_kernel<1>, dim3(m.wgs), dim3(512), 0, st, p);
      if (m.kind == 2) continue;
      if (m.kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, st, p);
      else if (p.pieces == 0) { p.pieces = 4; p.wg_stride = 0; p.w = w[0]; hipLaunchKernelGGL(node_kernel<4>, dim3(m.wgs), dim3(512), m.lds, st, p); }  // hand-off only: every workgroup reads the same 32 KiB
      else if (p.pieces < 8) hipLaunchKernelGGL(node_kernel<4>, dim3(m.wgs), dim3(512), m.lds, st, p);
      else if (m.ring == 16 && p.pieces % 16 == 0) hipLaunchKernelGGL(node_kernel<16>, dim3(m.wgs), dim3(512), m.lds, st, p);
      else hipLaunchKernelGGL(node_kernel<8>, dim3(m.wgs), dim3(512), m.lds, st, p);
    }
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int it = 0; it < 3; ++it) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipStreamSynchronize(st));
    const int reps = 20;
    CHECK(hipEventRecord(e0, st));
    for (int it = 0; it < reps; ++it) CHECK(hipGraphLaunch(ge, st));
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps / N;
    printf("%-46s %7.2f us per node", m.name, us);
    if (m.bytes) printf("   %6.2f TB/s", m.bytes / us / 1e6);
    printf("\n");
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  return 0;
}
