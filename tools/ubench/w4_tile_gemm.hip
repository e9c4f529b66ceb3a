// Developer harness of w4_gemm_tile_kernel (any4_amd/csrc/w4_gemm_tile.cuh): correctness against a CPU restatement of the reference's
// arithmetic on random inputs (a sample of activation rows, every weight row) and timing over rotating weight sets.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o variants/w4_tile_gemm tools/ubench/w4_tile_gemm.hip
//   variants/w4_tile_gemm [m n k g qtype(0 int4, 1 global, 2 rowwise) bn]
#include "../../any4_amd/csrc/tg_common.cuh"
namespace {
#include "../../any4_amd/csrc/w4_gemm_tile.cuh"
#include "w4_gemm_tile2.cuh"
}
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

#ifndef NPW_
#define NPW_ 12
#endif
#ifndef DX_
#define DX_ 3
#endif
#ifndef EW_
#define EW_ 2
#endif
#ifndef BM_
#define BM_ 128
#endif
#ifndef KS_
#define KS_ 1
#endif
#ifndef NCW128_
#define NCW128_ 4
#endif
#ifndef NDW128_
#define NDW128_ 8
#endif
template <int BN>
static int launch(const TileParams& p, hipStream_t st) {
  constexpr int ndw = BN == 64 ? 8 : NDW128_;
  constexpr int ks = BN == 64 ? KS_ : 1;
  constexpr int dx = (ks == 2 || BM_ == 256) ? 2 : DX_;
  constexpr int ncw = BN == 64 ? 4 : NCW128_;
#ifdef TILE2
  constexpr auto kern = w4_gemm_tile2_kernel<BF16, BM_, BN, dx, ks>;
  constexpr int nthreads = 768;
#else
  constexpr auto kern = w4_gemm_tile_kernel<BF16, BM_, BN, dx, ndw, ks, ncw>;
  constexpr int nthreads = 64 * (ncw + 4 + ndw);
#endif
  static bool prepared = false;
  if (!prepared) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); prepared = true; }
  constexpr unsigned lds_bytes = TileLds<BM_, BN, dx, ks>::BYTES;
  const int ns = p.splits > 1 ? p.splits : 1;
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n * ns), dim3(nthreads), lds_bytes, st, p);
  if (ns > 1) {
    const int64_t quads = (int64_t)p.m * p.wrows / 4;
    hipLaunchKernelGGL(tile_split_sum_kernel<BF16>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, p.part, ns, (int64_t)p.m * p.wrows, p.y, p.bias, p.wrows, quads);
  }
  return 0;
}

int main(int argc, char** argv) {
  int m = argc > 1 ? atoi(argv[1]) : 512, n = argc > 2 ? atoi(argv[2]) : 4096, k = argc > 3 ? atoi(argv[3]) : 4096;
  int g = argc > 4 ? atoi(argv[4]) : 128, qtype = argc > 5 ? atoi(argv[5]) : 2, bn = argc > 6 ? atoi(argv[6]) : 64;
  const int splits = argc > 7 ? atoi(argv[7]) : 1;
  const int xpad = argc > 8 ? atoi(argv[8]) : 0;   // extra elements per activation row (pitch experiment)
  const size_t xp = (size_t)k + xpad;
  const int SETS = 6;
  std::mt19937 rng(123);
  std::uniform_int_distribution<uint32_t> u32;
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  const int ksuper = k / 64, ngroups = k / g;
  std::vector<uint32_t> w((size_t)SETS * (n / 8) * ksuper * 64);
  for (auto& v : w) v = u32(rng);
  std::vector<uint16_t> lut((size_t)n * 16), qinfo((size_t)ngroups * n * 2), x((size_t)m * xp), y((size_t)m * n), bias(n);
  for (auto& v : lut) v = f2bf(nd(rng));
  for (size_t i = 0; i < (size_t)ngroups * n; ++i) { qinfo[2 * i] = f2bf(ud(rng) * 0.02f + 0.005f); qinfo[2 * i + 1] = f2bf(nd(rng) * 0.01f); }
  for (auto& v : x) v = f2bf(nd(rng));
  for (auto& v : bias) v = f2bf(nd(rng));
  char *dw, *dl, *dq, *dx, *dy, *db;
  CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dl, lut.size() * 2)); CK(hipMalloc(&dq, qinfo.size() * 2));
  const int xrot = getenv("XROT") ? atoi(getenv("XROT")) : 1;   // distinct copies of the activations, rotated per launch (cold x)
  CK(hipMalloc(&dx, x.size() * 2 * xrot)); CK(hipMalloc(&dy, y.size() * 2)); CK(hipMalloc(&db, bias.size() * 2));
  CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, lut.data(), lut.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dq, qinfo.data(), qinfo.size() * 2, hipMemcpyHostToDevice)); for (int c = 0; c < xrot; ++c) CK(hipMemcpy(dx + (size_t)c * x.size() * 2, x.data(), x.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), bias.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0xff, y.size() * 2));
  TileParams p{};
  p.x = dx; p.w = dw; p.qinfo = dq; p.lut = qtype == 0 ? nullptr : dl; p.y = dy; p.bias = nullptr;
  p.m = m; p.wrows = n; p.k = k; p.ksuper = ksuper; p.qtype = qtype;
  p.gshift = 0; while ((1 << p.gshift) < g) ++p.gshift;
  p.tiles_m = (m + BM_ - 1) / BM_; p.tiles_n = (n + bn - 1) / bn;
  p.splits = splits; p.part = nullptr; p.x_pitch = (int64_t)xp;
  if (splits > 1) CK(hipMalloc(&p.part, (size_t)splits * m * n * 4));
  auto go = [&](const TileParams& pp) { return bn == 64 ? launch<64>(pp, 0) : launch<128>(pp, 0); };
  go(p);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
  // ---- check: sampled activation rows against the reference's arithmetic ----
  std::vector<float> wdq((size_t)n * k);
  for (int r = 0; r < n; ++r)
    for (int kk = 0; kk < k; ++kk) {
      const int kt = kk >> 4, kqq = kk & 15, t = 4 * (r & 7) + ((kqq & 7) >> 1), ktl = kt % 4;
      const int v = (ktl & 1) * 4 + (kqq & 1) + 2 * (kqq >> 3);
      const size_t word = (((size_t)(r >> 3) * ksuper + kt / 4) * 32 + t) * 2 + (ktl >> 1);
      const int shift = (v & 1) * 16 + (v >> 1) * 4;
      const int code = (w[word] >> shift) & 15;
      const float lvv = qtype == 0 ? (float)(code - 8) : bf2f(lut[(qtype == 2 ? (size_t)r * 16 : 0) + code]);
      const size_t qi = ((size_t)(kk / g) * n + r) * 2;
      wdq[(size_t)r * k + kk] = bf2f(f2bf(fmaf(lvv, bf2f(qinfo[qi]), bf2f(qinfo[qi + 1]))));
    }
  double worst = 0; long bad = 0, checked = 0;
  std::vector<int> rows;
  for (int r = 0; r < m; r += (m > 64 ? 29 : 1)) rows.push_back(r);
  rows.push_back(m - 1);
  for (int r : rows)
    for (int c = 0; c < n; ++c) {
      double s = 0, sa = 0;
      for (int kk = 0; kk < k; ++kk) { const double pr = (double)bf2f(x[(size_t)r * xp + kk]) * wdq[(size_t)c * k + kk]; s += pr; sa += fabs(pr); }
      const double got = bf2f(y[(size_t)r * n + c]);
      const double tol = fabs(s) * 0.0045 + sa * 4e-6 + 1e-30;
      const double e = fabs(got - s);
      if (!(e <= tol)) { if (bad < 5) printf("MISMATCH y[%d][%d] = %g want %g (tol %g)\n", r, c, got, s, tol); ++bad; }
      worst = fmax(worst, e / tol); ++checked;
    }
  printf("check m=%d n=%d k=%d g=%d qtype=%d bn=%d splits=%d: %ld / %ld bad, worst err/tol %.3f\n", m, n, k, g, qtype, bn, splits, bad, checked, worst);
  // ---- timing: rotating weight sets ----
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 60;
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 6; ++i) { TileParams q = p; q.w = dw + (size_t)(i % SETS) * (n / 8) * ksuper * 256; go(q); }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) { TileParams q = p; q.w = dw + (size_t)(i % SETS) * (n / 8) * ksuper * 256; q.x = dx + (size_t)(i % xrot) * x.size() * 2; go(q); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("  %.2f us per launch  %.1f TFLOP/s\n", us, 2.0 * m * n * k / us * 1e-6);
  }
  return bad ? 1 : 0;
}
