// Developer harness of w8_gemm_kernel (any4_amd/csrc/w8_gemm.cuh) at few activation rows: timing over rotating weight sets, one launch per layer
// (random words: timing only; parity is tests/test_gpu_parity.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DW8_RING=..] [-DW8_ABL=..] -o variants/w8_gemv tools/ubench/w8_gemv.hip
//   variants/w8_gemv [m n k side(0 = Aint8 I=2, 1 = Bint8 I=4, 2 = Bint8 I=2) waves_splitk]
#include "../../any4_amd/csrc/tg_common.cuh"
namespace {
#include "../../any4_amd/csrc/w8_gemm.cuh"
}
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

#ifndef WV_
#define WV_ 8
#endif
template <bool LA, int I>
static void go(GemmParams& p, hipStream_t st) {
#ifndef WV_
#define WV_ 8
#endif
  constexpr int WAVES = WV_;
  const int tpb = WAVES / p.splitk;
  dim3 grid((unsigned)((p.rowtiles + tpb - 1) / tpb), (unsigned)((p.m + 15) / 16), 1);
  hipLaunchKernelGGL((w8_gemm_kernel<BF16, LA, I, WAVES>), grid, dim3(WAVES * 64), 0, st, p);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 1, n = argc > 2 ? atoi(argv[2]) : 4096, k = argc > 3 ? atoi(argv[3]) : 4096, side = argc > 4 ? atoi(argv[4]) : 0;
  const int sk_arg = argc > 5 ? atoi(argv[5]) : 0;
  const int SETS = 6, g = 128;
  const size_t wbytes = (size_t)n * k;
  char *dw, *dq, *dx, *dy;
  CK(hipMalloc(&dw, wbytes * SETS)); CK(hipMalloc(&dq, (size_t)(k / g) * n * 4)); CK(hipMalloc(&dx, (size_t)m * k * 2)); CK(hipMalloc(&dy, (size_t)m * n * 2));
  std::vector<uint32_t> h(wbytes / 4 * SETS);
  uint32_t s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
  CK(hipMemcpy(dw, h.data(), wbytes * SETS, hipMemcpyHostToDevice));
  CK(hipMemset(dq, 0x3c, (size_t)(k / g) * n * 4)); CK(hipMemset(dx, 0x3c, (size_t)m * k * 2));
  GemmParams p{};
  p.x = dx; p.w = dw; p.qinfo = dq; p.y = dy; p.m = m; p.wrows = n; p.k = k;
  const int I = side == 1 ? 4 : 2;
  p.ntiles = side == 0 ? n / 16 : n / 8; p.ksuper = k / (16 * I); p.gshift = 7; p.ngroups = k / g; p.qtype = TG_Q_INT8;
  p.rowtiles = (n + 15) / 16;
  const int nsteps = (k / 16 + 3) / 4;
  int sk = 1;
  while (sk < WV_ && (int64_t)p.rowtiles * ((m + 15) / 16) * sk < 256 * 16 && sk * 2 <= nsteps) sk *= 2;
  if (sk_arg) sk = sk_arg;
  p.splitk = sk; p.sk_shift = 0; while ((1 << p.sk_shift) < sk) ++p.sk_shift;
  auto launch = [&](int i) { GemmParams q = p; q.w = dw + (size_t)(i % SETS) * wbytes; if (side == 0) go<true, 2>(q, 0); else if (side == 1) go<false, 4>(q, 0); else go<false, 2>(q, 0); };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 6; ++i) launch(i);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 120; ++i) launch(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  m=%d n=%d k=%d side=%d splitk=%d: %.2f us per launch  %.2f TB/s\n", m, n, k, side, sk, ms * 1e3 / 120, (double)wbytes / (ms * 1e-3 / 120) * 1e-12);
  }
  return 0;
}
