// probe: v_cvt_scalef32_pk_bf16_fp4 semantics on gfx950 (byte select, nibble order, scale handling incl. NaN / denormal / 0)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const uint32_t* src, const float* scale, uint32_t* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w = src[i];
  float s = scale[i];
  bf16x2 r0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 0);
  bf16x2 r1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 1);
  bf16x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 2);
  bf16x2 r3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, s, 3);
  out[4 * i + 0] = __builtin_bit_cast(uint32_t, r0);
  out[4 * i + 1] = __builtin_bit_cast(uint32_t, r1);
  out[4 * i + 2] = __builtin_bit_cast(uint32_t, r2);
  out[4 * i + 3] = __builtin_bit_cast(uint32_t, r3);
}
static float fp4v(int c) { static const float v[8] = {0, .5f, 1, 1.5f, 2, 3, 4, 6}; float x = v[c & 7]; return (c & 8) ? -x : x; }
static uint16_t bf16_rne(float f) { uint32_t u; memcpy(&u, &f, 4); if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0; uint32_t r = u + 0x7fff + ((u >> 16) & 1); return r >> 16; }
int main() {
  const int NS = 10;
  uint32_t es[NS] = {127, 120, 130, 1, 0, 254, 255, 100, 200, 126};
  int n = 256 * NS;
  uint32_t* hs = new uint32_t[n]; float* hsc = new float[n];
  for (int s = 0; s < NS; ++s) for (int b = 0; b < 256; ++b) {
    hs[s * 256 + b] = b | ((b ^ 0x5a) << 8) | (((b * 7) & 255) << 16) | ((255 - b) << 24);
    uint32_t e = es[s]; uint32_t bits = e == 255 ? 0x7fc00000u : (e == 0 ? 0x00400000u : e << 23);
    memcpy(&hsc[s * 256 + b], &bits, 4);
  }
  uint32_t *ds, *dout; float* dsc;
  hipMalloc(&ds, n * 4); hipMalloc(&dsc, n * 4); hipMalloc(&dout, n * 16);
  hipMemcpy(ds, hs, n * 4, hipMemcpyHostToDevice); hipMemcpy(dsc, hsc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, ds, dsc, dout, n);
  uint32_t* ho = new uint32_t[4 * n];
  hipMemcpy(ho, dout, n * 16, hipMemcpyDeviceToHost);
  for (int s = 0; s < NS; ++s) {
    int bad = 0, nanlo = 0; uint32_t first = 0, want0 = 0; int fb = -1, fj = -1;
    for (int b = 0; b < 256; ++b) for (int j = 0; j < 4; ++j) {
      uint32_t w = hs[s * 256 + b]; uint32_t byte = (w >> (8 * j)) & 255;
      float sc = hsc[s * 256 + b];
      uint16_t lo = bf16_rne(fp4v(byte & 15) * sc), hi = bf16_rne(fp4v(byte >> 4) * sc);
      if (es[s] == 255) { lo = hi = 0x7fc0; }
      uint32_t want = lo | ((uint32_t)hi << 16), got = ho[4 * (s * 256 + b) + j];
      bool ok = got == want;
      if (es[s] == 255) ok = ((got & 0x7f80) == 0x7f80 && (got & 0x7f)) && (((got >> 16) & 0x7f80) == 0x7f80 && ((got >> 16) & 0x7f));
      if (!ok) { if (!bad) { first = got; want0 = want; fb = b; fj = j; } ++bad; }
    }
    printf("e=%3u: %d mismatches of 1024", es[s], bad);
    if (bad) printf("  first: byte-src b=%d j=%d got %08x want %08x", fb, fj, first, want0);
    printf("\n");
  }
  return 0;
}
