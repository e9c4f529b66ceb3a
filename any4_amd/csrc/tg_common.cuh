// tg_common.cuh -- what every translation unit of libtinygemm_hip.so shares: type traits, host helpers, the parameter block the
// C ABI fills in (GemmParams) and the entry points of the kernel families' launch paths (namespace tgx).
//
// The library is built from one translation unit per kernel family (any4_amd/build.py compiles them in parallel):
//   tinygemm_hip.hip   the C ABI, validation, dispatch (launch_w4), packers / converters, 16-bit and 8-bit weights, decode glue
//   tg_splitk.hip      w4_gemm_kernel            (w4_gemm.cuh)
//   tg_stream.hip      w4_gemm_stream_kernel     (w4_gemm_stream.cuh)      compiled once per 16-bit type (-DTG_TU_F16)
//   tg_pair.hip        w4_gemm_pair_kernel       (w4_gemm_pair.cuh)        compiled once per 16-bit type
//   tg_pair16.hip      w4_gemm_pair16_kernel     (w4_gemm_pair16.cuh)
//   tg_xr.hip          w4_gemm_xr_kernel         (w4_gemm_xr.cuh)
//   tg_gemv.hip        w4_gemv_kernel            (w4_gemv.cuh)
//   tg_tile.hip        w4_gemm_tile_kernel       (w4_gemm_tile.cuh)
// Kernels and their helpers stay in each unit's anonymous namespace (one device code object per unit, no symbol shared between
// them); only GemmParams and the tgx:: functions cross unit boundaries.
#pragma once
#ifndef GEMV_TRACE
#define GEMV_TRACE 0  // developer builds (-DGEMV_TRACE=1): s_memrealtime stamps of kernel phases (dev/gemv_trace.py)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include <utility>

#include "../../include/tinygemm_hip.h"

struct GemmParams {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  int32_t m, wrows, k;
  int32_t ntiles;    // packed.size(0): 8-row (Bint4) or 16-row (Aint4) tiles
  int32_t ksuper;    // packed.size(1)
  int32_t gshift;    // log2(group)
  int32_t ngroups;   // k / group
  int32_t qtype;
  int32_t splitk;    // waves per tile (power of two, <= WAVES)
  int32_t sk_shift;  // log2(splitk)
  int32_t rowtiles;  // ceil(wrows / 16)
  int32_t dbg;       // developer ablation flags (0 in production)
  int32_t numerics;  // TG_NUM_* (host-side dispatch only)
  int32_t dot2;      // host-side only: TG_NUM_FAST_DOT2 was asked for (never promote a stacked m = 1 launch to the matrix-core contraction)
  int32_t dry;       // host-side only: report the kernel family instead of launching (tg_gemm_w4_plan)
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y;
  const char* bias;   // optional [wrows] 16-bit, added after the first rounding (see store_rows4)
  int64_t stride_bias;
  int64_t bias_row_stride;  // elements between the bias rows of consecutive activation rows (0: one row for all; wrows: a residual)
  const char* norm_w;       // fused RMSNorm of the activations (pair-table kernels only) / host-side dispatch
  float norm_eps;
  int32_t epilogue;         // TG_EPI_* (pair-table kernels only)
  // host-side only: the caller's workspace (pair kernel, XG variant) and the planner's answer to "how much would help"
  char* ws;
  int64_t ws_bytes, ws_need;
  int32_t ws_query;
  int32_t x_tc, y_tc;  // fragment-order activations / output (pair-table kernels only)
};

// Lane exchanges WITHOUT the LDS crossbar.  hipcc turns every `__shfl_xor` into ds_bpermute_b32 -- an LDS-pipeline round trip (address
// register, lgkmcnt wait, ~100 cycles) -- and a butterfly is four to six of them back to back, usually on a launch's critical path in
// front of its first barrier (dev/gemv_trace.py: the 16-lane step sums were 0.6 us of a 5.8 us launch).  gfx950 has the data paths in
// the vector ALU: DPP operands (quad permutes, rotations within a 16-lane row) and v_permlane16_swap / v_permlane32_swap across rows.
// TG_LANE_SHFL=1 (developer A/B) restores the shuffles.
#ifndef TG_LANE_SHFL
#define TG_LANE_SHFL 0
#endif
namespace tgl {
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// the two halves of a pair of 16-lane rows / of the wave, side by side: [0] = (r0, r0, r2, r2) / (lo, lo), [1] = (r1, r1, r3, r3) / (hi, hi)
__device__ __forceinline__ void rows16(float v, float& even, float& odd) {
  const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  even = __builtin_bit_cast(float, (unsigned)sw[0]);
  odd = __builtin_bit_cast(float, (unsigned)sw[1]);
}
__device__ __forceinline__ void halves32(float v, float& lo, float& hi) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  lo = __builtin_bit_cast(float, (unsigned)sw[0]);
  hi = __builtin_bit_cast(float, (unsigned)sw[1]);
}
// the value lane ^ O holds (O = 1, 2, 8: one DPP move; 16, 32: one row swap and a select on this lane's side; `lane` = lane id)
template <int O>
__device__ __forceinline__ float lane_xor(float v, int lane) {
  static_assert(O == 1 || O == 2 || O == 8 || O == 16 || O == 32, "xor by 4 has no DPP form on gfx9: rotate (row_sum) or shuffle");
#if TG_LANE_SHFL
  return __shfl_xor(v, O, 64);
#else
  if constexpr (O == 1) return dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  else if constexpr (O == 2) return dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  else if constexpr (O == 8) return dpp<0x128>(v);  // row_ror:8
  else if constexpr (O == 16) { float a, b; rows16(v, a, b); return (lane & 16) ? a : b; }
  else { float a, b; halves32(v, a, b); return (lane & 32) ? a : b; }
#endif
}
// every lane of a 16-lane row gets the row's sum (rotations by 8, 4, 2, 1: the same bits in all 16 lanes)
__device__ __forceinline__ float row_sum(float v) {
#if TG_LANE_SHFL
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  v += dpp<0x128>(v);
  v += dpp<0x124>(v);
  v += dpp<0x122>(v);
  v += dpp<0x121>(v);
  return v;
#endif
}
__device__ __forceinline__ float row_max(float v) {
#if TG_LANE_SHFL
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#else
  v = fmaxf(v, dpp<0x128>(v));
  v = fmaxf(v, dpp<0x124>(v));
  v = fmaxf(v, dpp<0x122>(v));
  v = fmaxf(v, dpp<0x121>(v));
  return v;
#endif
}
// ... of the rows 2 j, 2 j + 1 / of the whole wave (every lane the same bits)
__device__ __forceinline__ float rows16_sum(float v) {
#if TG_LANE_SHFL
  return v + __shfl_xor(v, 16, 64);
#else
  float a, b; rows16(v, a, b); return a + b;
#endif
}
__device__ __forceinline__ float halves32_sum(float v) {
#if TG_LANE_SHFL
  return v + __shfl_xor(v, 32, 64);
#else
  float a, b; halves32(v, a, b); return a + b;
#endif
}
__device__ __forceinline__ float wave_sum(float v) { return halves32_sum(rows16_sum(row_sum(v))); }
__device__ __forceinline__ float wave_max(float v) {
#if TG_LANE_SHFL
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#else
  v = row_max(v);
  float a, b;
  rows16(v, a, b); v = fmaxf(a, b);
  halves32(v, a, b); return fmaxf(a, b);
#endif
}
}  // namespace tgl

enum { CANON_NONE = 0, CANON_PAIR = 1, CANON_QUAD = 2 };

// Returned by a family's launch path when the shape does not fit its plan (the caller then takes another kernel).
enum { TG_PAIR_NA = -100 };

// launch paths of the kernel families (dt = TG_BF16 / TG_F16; the other arguments as in the templates they wrap).  The two largest
// families are compiled once per 16-bit type.
namespace tgx {
int pair_bf16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair_f16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair_a_bf16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair_a_f16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair_b16_bf16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair_b16_f16(int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int stream_bf16(bool layout_a, int wpl, bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st);
int stream_f16(bool layout_a, int wpl, bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st);
int pair_xr(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int pair16(int dt, int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st);
int pair16_loop(int dt, int I, bool qmx, const GemmParams& p, int64_t batch, hipStream_t st);  // w4_gemm_pair16_loop.cuh: one layer, more 16-row tiles than CUs
int splitk(int dt, bool layout_a, int canon, bool qmx, int waves, const GemmParams& p, dim3 grid, hipStream_t st);
int gemv(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);
int tile(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st);  // w4_gemm_tile.cuh: many activation rows
int tile_w8(int dt, bool on_right, int I, GemmParams& p, int64_t batch, hipStream_t st);  // ... int8 weights (tg_gemm_w8)
inline int pair(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? pair_bf16(I, qmx, p, batch, st) : pair_f16(I, qmx, p, batch, st);
}
inline int pair_a(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? pair_a_bf16(I, qmx, p, batch, st) : pair_a_f16(I, qmx, p, batch, st);
}
inline int pair_b16(int dt, int I, bool qmx, GemmParams& p, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? pair_b16_bf16(I, qmx, p, batch, st) : pair_b16_f16(I, qmx, p, batch, st);
}
inline int stream(int dt, bool layout_a, int wpl, bool qmx, const GemmParams& p, int64_t coltiles, int64_t batch, hipStream_t st) {
  return dt == TG_BF16 ? stream_bf16(layout_a, wpl, qmx, p, coltiles, batch, st) : stream_f16(layout_a, wpl, qmx, p, coltiles, batch, st);
}
}  // namespace tgx

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// LDS pointers (address space 3: ds_* instructions with constant offsets)
typedef const __attribute__((address_space(3))) float* lds_cfptr;
typedef __attribute__((address_space(3))) float* lds_fptr;
typedef const __attribute__((address_space(3))) uint16_t* lds_cu16ptr;
typedef const __attribute__((address_space(3))) uint32_t* lds_cu32ptr;
typedef __attribute__((address_space(3))) uint32_t* lds_u32ptr;
typedef const __attribute__((address_space(3))) u32x4* lds_cu32x4ptr;
typedef __attribute__((address_space(3))) u32x4* lds_u32x4ptr;
typedef const __attribute__((address_space(3))) f32x4* lds_cf32x4ptr;
typedef __attribute__((address_space(3))) f32x4* lds_f32x4ptr;

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- 16-bit float traits ---------------------------------------------------------------------
struct BF16 {
  static __device__ __forceinline__ float to_f32(uint16_t h) { return u2f(((uint32_t)h) << 16); }
  static __device__ __forceinline__ float lo_f32(uint32_t pair) { return u2f(pair << 16); }
  static __device__ __forceinline__ float hi_f32(uint32_t pair) { return u2f(pair & 0xffff0000u); }
  // round-to-nearest-even pack; lowers to v_cvt_pk_bf16_f32 on gfx950
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  }
  static __device__ __forceinline__ uint16_t from_f32(float a) { return (uint16_t)(pack2(a, 0.f) & 0xffffu); }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
struct F16 {
  static __device__ __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ float lo_f32(uint32_t pair) { return to_f32((uint16_t)(pair & 0xffffu)); }
  static __device__ __forceinline__ float hi_f32(uint32_t pair) { return to_f32((uint16_t)(pair >> 16)); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
  }
  static __device__ __forceinline__ uint16_t from_f32(float a) { return (uint16_t)(pack2(a, 0.f) & 0xffffu); }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// fp4-e2m1 values in code order (reference FloatDefs.cuh:18-34)
__device__ const float kMX4Values[16] = {0.0f,  0.5f,  1.0f,  1.5f,  2.0f,  3.0f,  4.0f,  6.0f,
                                         -0.0f, -0.5f, -1.0f, -1.5f, -2.0f, -3.0f, -4.0f, -6.0f};

// Output store of four consecutive weight rows of one activation row.  With a bias the sum is first rounded to 16 bits and
// the bias added in a second rounded step: bit-identical to the reference module's separate `y + bias` (modules.py:221-222).
template <typename DT>
__device__ __forceinline__ void store_rows4(char* yb, const char* bias, int64_t elem, int rowg, f32x4 acc) {
  u32x2 o = {DT::pack2(acc[0], acc[1]), DT::pack2(acc[2], acc[3])};
  if (bias) {
    const u32x2 bv = *reinterpret_cast<const u32x2*>(bias + (int64_t)rowg * 2);
    o[0] = DT::pack2(DT::lo_f32(o[0]) + DT::lo_f32(bv[0]), DT::hi_f32(o[0]) + DT::hi_f32(bv[0]));
    o[1] = DT::pack2(DT::lo_f32(o[1]) + DT::lo_f32(bv[1]), DT::hi_f32(o[1]) + DT::hi_f32(bv[1]));
  }
  *reinterpret_cast<u32x2*>(yb + elem * 2) = o;
}


// 4 x 4 transpose across the four 16-lane rows of a wave: out[e] in row Q = in[Q] of row e (lane = i + 16 row).  Two v_permlane16_swap
// (odd rows of the first operand <-> even rows of the second) and two v_permlane32_swap (upper half of the first <-> lower half of the second).
// Use: the MFMA 16x16x32 fragment of lane (i, Q) is the dwords Q, Q + 4, Q + 8, Q + 12 of a 64-byte block of row i -- one 16-byte load per
// lane of dwords 4Q ... 4Q + 3 and this transpose instead of four 4-byte loads (a quarter of the requests in the vector-memory path).
__device__ __forceinline__ u32x4 transpose_rows4(u32x4 v) {
  const auto s01 = __builtin_amdgcn_permlane16_swap(v[0], v[1], false, false);
  const auto s23 = __builtin_amdgcn_permlane16_swap(v[2], v[3], false, false);
  const auto t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
  const auto t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
  return u32x4{t02[0], t13[0], t02[1], t13[1]};
}

struct DeviceScope {
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int device) {
    if (device < 0) return;
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != device && hipSetDevice(device) != hipSuccess) ok = false;
    if (prev == device) prev = -1;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Kernels that address LDS from offset 0 (lookup tables at the start of the dynamic region) and / or need more than 64 KiB
// of dynamic LDS: once per device, check that the kernel has no static LDS (the dynamic region then starts at 0) and raise
// its dynamic-LDS limit.  State = one write-once bit per device and kernel; racing threads repeat the same idempotent calls.
template <auto KERN>
int prepare_lds_kernel() {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return TG_E_DEVICE;
  if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1u)) return 0;
  hipFuncAttributes fa;
  hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(KERN));
  if (e != hipSuccess) return (int)e;
  if (fa.sharedSizeBytes != 0) return TG_E_INTERNAL;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERN), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  if (dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
  return 0;
}


// compute units of the current device (write-once cache per device index; racing threads store the same value)
inline int cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cache[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cache[dev].store(v, std::memory_order_relaxed);
  return v;
}


// Tuning constants of the pair-table launches, each with the measurement that set it.  The shipped library
// always uses these values; only developer builds (-DTG_DEV / -DTG_DEV_MIN, dev/build_variant.sh) may override one with -D<NAME>=<v>.
#if !defined(TG_DEV) && !defined(TG_DEV_MIN)
#if defined(TG_PAIR_R) || defined(TG_PAIR_ABL) || defined(TG_PAIR_MR1) || defined(TG_PAIR_NSG2) || defined(TG_PAIR_MR1_GPS) || defined(TG_PAIR_RA) ||   \
    defined(TG_PAIR_RA1) || defined(TG_PAIR_RB16) || defined(TG_B16_CHUNK) || defined(TG_PAIR_MIN_ITEMS) || defined(TG_XG_CHUNK) || defined(TG_PAIR_WGS) || \
    defined(TG_PAIR_NSG2_M1) || defined(TG_PAIR_FORCE_XG) || defined(TG_XR_MIN_M) || defined(TG_XR_R) || defined(TG_XR_R8K) || defined(TG_XR_RMX)
#error "the TG_PAIR_* / TG_XG_* / TG_B16_* tuning constants can only be overridden in developer builds (-DTG_DEV or -DTG_DEV_MIN)"
#endif
#endif
#ifndef TG_PAIR_R
#define TG_PAIR_R 2            // super-tiles a wave keeps in flight (2, 3, 4 measured equal; 5 spills)
#endif
#ifndef TG_PAIR_ABL
#define TG_PAIR_ABL 0          // ablation stub of w4_gemm_pair.cuh (its header lists them)
#endif
#ifndef TG_PAIR_MR1
#define TG_PAIR_MR1 1          // 1: m = 1 runs the one-register specialisation (+2-3 %), 4: the general m <= 8 kernel
#endif
#ifndef TG_PAIR_NSG2
#define TG_PAIR_NSG2 1         // group boundaries at fixed places of the unrolled round when a group is one round of the ring
#endif
#ifndef TG_PAIR_NSG2_M1
#define TG_PAIR_NSG2_M1 1      // ... also in the m = 1 specialisation (74.8 -> 75.8 % once its group update was spelled out)
#endif
#ifndef TG_PAIR_MR1_GPS
#define TG_PAIR_MR1_GPS 1      // the m = 1 specialisation is used up to this many groups per super-tile (it spills beyond)
#endif
#ifndef TG_PAIR_RA
#define TG_PAIR_RA 2           // ring depth of the A-side kernels (4 / 6 / 8 measured 8.2 / 21 / 40 us against 6.8)
#endif
#ifndef TG_PAIR_RA1
#define TG_PAIR_RA1 1          // ... with several groups per super-tile
#endif
#ifndef TG_PAIR_RB16
#define TG_PAIR_RB16 2         // ring depth of the 16x16x32 kernels for Bint4 weights, m = 9 ... 16 (3 / 4: 41 % against 44 %)
#endif
#ifndef TG_B16_CHUNK
#define TG_B16_CHUNK 4         // consecutive 32-row work items per workgroup visit of those kernels (1 / 4 / 8 within 1 %)
#endif
#ifndef TG_TILE_W8_MIN_M
#define TG_TILE_W8_MIN_M 17  // int8 weights: activation rows from which tg_gemm_w8 takes the tile GEMM (tg_tile.hip): 14.3 us per 4096^2 layer; the 16-row kernel: 8.6 us up to 6 rows
#endif
#ifndef TG_TILE_MIN_M_SPLIT
#define TG_TILE_MIN_M_SPLIT 17  // ... and with a split-K launch of ONE layer (caller's workspace; tg_tile.hip): 12.6-14.8 us at 17 ... 64 rows against 7.5 us per 16-row pass
#endif
#ifndef TG_TILE_MIN_M
#define TG_TILE_MIN_M 65  // activation rows from which a call takes the LDS-tiled MFMA GEMM (w4_gemm_tile.cuh): beyond the 64 rows the row blocks of the group-scaled kernels cover
#endif
#ifndef TG_M1_DEFAULT_MFMA
#define TG_M1_DEFAULT_MFMA 0  // which contraction a STACKED m = 1 launch takes under TG_NUM_FAST: 0 = per-lane v_dot2, 1 = the matrix core (tg_m1_default_contraction())
#endif
#ifndef TG_PAIR_MIN_ITEMS
#define TG_PAIR_MIN_ITEMS 192  // fewer work items: the launch is latency-bound, w4_gemm_pair16_kernel / the reference kernels take
                               // it (measured per hipGraph node, one layer, m = 1: 14336 x 4096 = 224 items 12.3 us here against
                               // 18.3 us on pair16 and 13.8 us on the stream kernel; 6144 x 4096 = 96 items 10.7 against 9.9 / 8.2)
#endif
#ifndef TG_XG_CHUNK
#define TG_XG_CHUNK 4          // consecutive work items per workgroup visit in the workspace variant of Bint4 weights (1: plain
                               // round-robin): m = 8: 4096^2 66.1 -> 66.5 %, 8192^2 67.3 -> 68.6 %; Aint4 weights keep 1
#endif
#ifndef TG_XR_MIN_M
#define TG_XR_MIN_M 2          // activation rows from which the register-resident-activation kernel (w4_gemm_xr.cuh) takes stacked launches
                               // (same-box A/B against the kernels it replaces, 4096^2: m = 2 71.9 vs 69.7 %, 4: 68.9 vs 66.9, 8: 66.2 vs 62.7,
                               //  16: 65.2 vs 46.0; m = 1 stays on the 32x32x16 kernel, 77 %)
#endif
#ifndef TG_XR_RMX
#define TG_XR_RMX 8            // ... for mx4 (no lookups: latency-bound; the whole slice in flight: m = 16 75.4 -> 79.0 %, m = 2 81.7 -> 85.3 %)
#endif
#ifndef TG_XR_R8K
#define TG_XR_R8K 2            // ... at k = 8192 (128 registers of activations per lane: 4 in flight spill 25)
#endif
#ifndef TG_XR_R
#define TG_XR_R 4              // super-tiles a wave of that kernel keeps in flight
#endif
#ifndef TG_PAIR_WGS
#define TG_PAIR_WGS 512        // persistent workgroups: two per CU (768 / 1024: +4 % / +1 % time)
#endif

}  // namespace
