// Pricing of a PERSISTENT decode layer on MI355X: a chain of four weight-streaming GEMV-shaped ops with the byte
// volumes of a Llama-3-8B layer's 4-bit linears (o-proj 8.4 MB -> gate_up 58.7 MB -> down 29.4 MB -> q/k/v 12.6 MB, every op needs ALL
// outputs of the one before), run three ways on the same data:
//   launches   one launch per op, replayed from a hipGraph (what any4_amd/decode.py issues; x read from global memory)
//   persist    ONE launch for all layers; ops hand their outputs over inside the launch as 8-byte {data, tag} granules written with
//              agent-scope (sc1) stores and swept by every workgroup with sc1 loads (MI355X_MICROARCH.md, rows allgather / handoff-1to1);
//              the weight requests of an op start AFTER its input vector has been gathered
//   runahead   the same, but a wave requests the first RING loads of the NEXT op's weights BEFORE it waits for the hand-over (weights do
//              not depend on activations): row prefetch-credit
// The arithmetic is a placeholder that keeps the op memory-bound (weights as bf16 pairs, v_dot2 against the staged vector); what is priced
// is streaming + hand-over structure.  Every spin is bounded (a give-up sets an error word and the launch ends).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o variants/persistent_chain tools/ubench/persistent_chain.hip && variants/persistent_chain [layers]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#ifndef RING_
#define RING_ 16
#endif
constexpr int NWG = 256, NT = 512, NWAVE = 8, RING = RING_;
#ifndef GW_
#define GW_ 2
#endif
constexpr int GW = GW_;        // waves of a workgroup that sweep the hand-over granules; they do NOT run ahead (a wave's vector-memory results return in
                               // request order: a poll behind eight HBM misses would wait for them)
constexpr int MAXG = NWG * NWAVE * 8;   // granules per hand-over array: every wave owns ceil(values / 2) <= 8 consecutive ones (pads are zero)
constexpr int NOPS = 4;

struct Op {
  int rows, row_bytes, in_len, out_len;   // out_len = values handed over (gate_up: 14336 after SwiGLU; others: rows)
};
__constant__ Op c_ops[NOPS];

struct Params {
  const char* w[NOPS];          // per op: [layers][rows][row_bytes]
  unsigned long long* g[NOPS];  // per op: granules of its outputs, [out_len / 2] x {bf16x2 data, tag}
  uint16_t* y[NOPS];            // launches variant: plain outputs [out_len]
  const uint16_t* x0;           // input of the first op
  uint32_t* err;
  unsigned long long* stamps;   // workgroup 0, wave 0: s_memrealtime (100 MHz) after every op's publish and after every gather
  int layers;
  int runahead;
};

__device__ __forceinline__ float dot8(u32x4 w, u32x4 x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w[i]), __builtin_bit_cast(bf16x2, x[i]), acc, false);
  return acc;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// One op's stream for one wave: rows [r0, r0 + nr), each row_bytes long; a flat sequence of 1 KiB wave-loads, RING of them in flight.
struct Stream {
  const char* base;   // first byte of the wave's rows
  int loads;          // 1 KiB wave-loads in all
  int lpr;            // wave-loads per row
};
__device__ __forceinline__ Stream stream_of(const Op& op, const char* w, int layer, int wg, int wave, int& r0, int& nr) {
  const int per_wave = op.rows / (NWG * NWAVE);
  nr = per_wave;
  r0 = (wg * NWAVE + wave) * per_wave;
  Stream s;
  s.base = w + ((int64_t)layer * op.rows + r0) * op.row_bytes;
  s.lpr = op.row_bytes / 1024;
  s.loads = nr * s.lpr;
  return s;
}

template <bool PERSIST>
__device__ __forceinline__ void run_op(const Params& p, int opi, int layer, int wg, int wave, int lane, char* lds_x, u32x4 (&ring)[RING], bool ring_primed, uint16_t* outs) {
  const Op op = c_ops[opi];
  int r0, nr;
  const Stream s = stream_of(op, p.w[opi], layer, wg, wave, r0, nr);
  const char* lp = s.base + lane * 16;
  if (!ring_primed) {
#pragma unroll
    for (int j = 0; j < RING; ++j) ring[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lp + (int64_t)(j < s.loads ? j : s.loads - 1) * 1024));
  }
  float acc = 0.f;
  int row = 0, inrow = 0;
  for (int i0 = 0; i0 < s.loads; i0 += RING) {
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      const int i = i0 + j;
      if (i < s.loads) {
        const u32x4 wv = ring[j];
        const int nxt = i + RING;
        ring[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lp + (int64_t)(nxt < s.loads ? nxt : s.loads - 1) * 1024));
        const u32x4 xv = *reinterpret_cast<const u32x4*>(lds_x + (inrow * 64 + lane) * 16);
        acc = dot8(wv, xv, acc);
        if (++inrow == s.lpr) {
          const float t = wave_sum(acc);
          if (lane == 0) outs[row] = f2bf(t);
          acc = 0.f; inrow = 0; ++row;
        }
      }
    }
  }
}

// the first RING loads of (opi, layer) for this wave, before its input is known
__device__ __forceinline__ void prime(const Params& p, int opi, int layer, int wg, int wave, int lane, u32x4 (&ring)[RING]) {
  const Op op = c_ops[opi];
  int r0, nr;
  const Stream s = stream_of(op, p.w[opi], layer, wg, wave, r0, nr);
  const char* lp = s.base + lane * 16;
#pragma unroll
  for (int j = 0; j < RING; ++j) ring[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(lp + (int64_t)(j < s.loads ? j : s.loads - 1) * 1024));
}

__global__ void __launch_bounds__(NT) persistent_kernel(const Params p) {
  __shared__ __attribute__((aligned(16))) char lds_x[14336 * 2];
  __shared__ uint16_t outs_s[NWAVE][16];
  const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 ring[RING];
  // the first op's input: plain global memory
  for (int i = threadIdx.x; i < c_ops[0].in_len / 8; i += NT) reinterpret_cast<u32x4*>(lds_x)[i] = reinterpret_cast<const u32x4*>(p.x0)[i];
  __syncthreads();
  bool primed = false, dead = false;
  uint32_t epoch = 0;
  for (int layer = 0; layer < p.layers; ++layer) {
    for (int opi = 0; opi < NOPS; ++opi) {
      ++epoch;
      run_op<true>(p, opi, layer, wg, wave, lane, lds_x, ring, primed, outs_s[wave]);
      primed = false;
      // ---- publish this wave's outputs as granules (two values + tag) ----
      const Op op = c_ops[opi];
      const int per_wave = op.rows / (NWG * NWAVE);
      // (gate_up: rows = 28672 -> 14336 handed over: a wave's 14 rows become 7 values; the others hand over every row)
      const int nvals = per_wave * op.out_len / op.rows, ng = (nvals + 1) / 2;
      __builtin_amdgcn_wave_barrier();
      if (lane < ng) {
        const uint32_t a = outs_s[wave][2 * lane], b = 2 * lane + 1 < nvals ? outs_s[wave][2 * lane + 1] : 0u;
        const unsigned long long gv = ((unsigned long long)epoch << 32) | (b << 16) | a;
        __hip_atomic_store(p.g[opi] + (wg * NWAVE + wave) * ng + lane, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (wg == 0 && threadIdx.x == 0) p.stamps[(layer * NOPS + opi) * 2] = __builtin_amdgcn_s_memrealtime();
      const bool last = layer == p.layers - 1 && opi == NOPS - 1;
      if (last) break;
      const int nopi = (opi + 1) % NOPS, nlayer = opi + 1 == NOPS ? layer + 1 : layer;
      if (p.runahead && wave >= GW) { prime(p, nopi, nlayer, wg, wave, lane, ring); primed = true; }
      // ---- gather the next op's input: in_len values = in_len / 2 granules, all waves share the sweep ----
      __syncthreads();   // (everyone is done reading lds_x of this op)
      const int ngr = c_ops[nopi].in_len / 2;
      const unsigned long long* src = p.g[opi];
      if (wave < GW) {
        // a PASS = 16 granule loads per lane in flight, then the tag compares; a pass with a missing granule is repeated (bounded)
        for (int base = 0; base < ngr; base += GW * 64 * 16) {
          int spins = 0;
          for (;;) {
            unsigned long long gv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int i = base + j * (GW * 64) + (int)threadIdx.x;
              gv[j] = __hip_atomic_load(src + (i < ngr ? i : ngr - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 16; ++j) ok = ok && (uint32_t)(gv[j] >> 32) == epoch;
            if (ok || dead) {
#pragma unroll
              for (int j = 0; j < 16; ++j) {
                const int i = base + j * (GW * 64) + (int)threadIdx.x;
                if (i < ngr) reinterpret_cast<uint32_t*>(lds_x)[i] = (uint32_t)gv[j];
              }
              break;
            }
            if (++spins > (1 << 12)) { atomicOr(p.err, 1u); dead = true; }   // (bounded: then this thread never spins again)
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
      __syncthreads();
      if (wg == 0 && threadIdx.x == 0) p.stamps[(layer * NOPS + opi) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// launches variant: one op of one layer per launch; x from global memory (plain), outputs plain
__global__ void __launch_bounds__(NT) op_kernel(const Params p, int opi, int layer, const uint16_t* x, uint16_t* y) {
  __shared__ __attribute__((aligned(16))) char lds_x[14336 * 2];
  __shared__ uint16_t outs_s[NWAVE][16];
  const int wg = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 ring[RING];
  const Op op = c_ops[opi];
  for (int i = threadIdx.x; i < op.in_len / 8; i += NT) reinterpret_cast<u32x4*>(lds_x)[i] = reinterpret_cast<const u32x4*>(x)[i];
  __syncthreads();
  run_op<false>(p, opi, layer, wg, wave, lane, lds_x, ring, false, outs_s[wave]);
  const int per_wave = op.rows / (NWG * NWAVE);
  const int nvals = per_wave * op.out_len / op.rows, ng = (nvals + 1) / 2;
  __builtin_amdgcn_wave_barrier();
  if (lane < nvals) y[(wg * NWAVE + wave) * ng * 2 + lane] = outs_s[wave][lane];   // (the granule arrays' padded order; pads stay zero)
}

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 16;
  // rows, row bytes, input length, values handed over
  const Op ops[NOPS] = {{4096, 2048, 4096, 4096}, {28672, 2048, 4096, 14336}, {4096, 7168, 14336, 4096}, {6144, 2048, 4096, 6144}};
  // (the next op gathers ITS in_len values from the previous op's granules: o-proj after q/k/v takes the first 4096 of 6144)
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_ops), ops, sizeof(ops)));
  Params p{};
  p.layers = layers;
  size_t total = 0;
  for (int i = 0; i < NOPS; ++i) {
    const size_t bytes = (size_t)layers * ops[i].rows * ops[i].row_bytes;
    total += bytes;
    char* w;
    CK(hipMalloc(&w, bytes));
    std::vector<uint16_t> h(bytes / 2);
    uint32_t s = 12345u + i;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u | ((s >> 16) & 0x00ffu)) ^ (uint16_t)((s >> 9) & 0x8000u); }  // bf16 ~ +-0.0078 .. 0.0156 magnitudes
    CK(hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice));
    p.w[i] = w;
    CK(hipMalloc(&p.g[i], (size_t)MAXG * 8));
    CK(hipMemset(p.g[i], 0, (size_t)MAXG * 8));
    CK(hipMalloc(&p.y[i], (size_t)MAXG * 4));
    CK(hipMemset(p.y[i], 0, (size_t)MAXG * 4));
  }
  uint16_t* x0;
  CK(hipMalloc(&x0, 14336 * 2));
  {
    std::vector<uint16_t> h(14336);
    uint32_t s = 777u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)(0x3f00u | ((s >> 16) & 0x7fu)); }
    CK(hipMemcpy(x0, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  }
  p.x0 = x0;
  CK(hipMalloc(&p.stamps, 8 * 2 * NOPS * 64));
  CK(hipMalloc(&p.err, 4));
  CK(hipMemset(p.err, 0, 4));
  printf("%d layers, %.1f MB of weights per layer, %d workgroups x %d threads, ring %d x 16 B per lane\n", layers, total / 1e6 / layers, NWG, NT, RING);

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // ---- launches, from a graph ----
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < layers; ++l)
    for (int i = 0; i < NOPS; ++i) {
      const uint16_t* xin = (l == 0 && i == 0) ? x0 : p.y[(i + NOPS - 1) % NOPS];
      hipLaunchKernelGGL(op_kernel, dim3(NWG), dim3(NT), 0, st, p, i, l, xin, p.y[i]);
    }
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("launches (hipGraph, %d nodes)      %8.2f us per layer\n", layers * NOPS, ms * 1e3 / 5 / layers);
  }
  const int last_ng = NWG * NWAVE * ((ops[NOPS - 1].rows / (NWG * NWAVE) * ops[NOPS - 1].out_len / ops[NOPS - 1].rows + 1) / 2);
  std::vector<uint16_t> ref((size_t)last_ng * 2);
  CK(hipMemcpy(ref.data(), p.y[NOPS - 1], ref.size() * 2, hipMemcpyDeviceToHost));

  // ---- persistent: hand-overs inside the launch ----
  for (int ra = 0; ra < 2; ++ra) {
    p.runahead = ra;
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < NOPS; ++i) CK(hipMemsetAsync(p.g[i], 0, (size_t)MAXG * 8, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(persistent_kernel, dim3(NWG), dim3(NT), 0, st, p);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      uint32_t err = 0;
      CK(hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost));
      printf("persistent %-9s                %8.2f us per layer%s\n", ra ? "run-ahead" : "at-edge", ms * 1e3 / layers, err ? "   [a spin gave up]" : "");
      if (err) { CK(hipMemset(p.err, 0, 4)); }
    }
    {  // per-op spans seen by workgroup 0 (average over the layers but the first): op = gather end -> publish, hand-over = publish -> gather end
      std::vector<unsigned long long> stv((size_t)2 * NOPS * layers);
      CK(hipMemcpy(stv.data(), p.stamps, stv.size() * 8, hipMemcpyDeviceToHost));
      double opus[NOPS] = {0, 0, 0, 0}, hous[NOPS] = {0, 0, 0, 0};
      for (int l = 1; l < layers - 1; ++l)
        for (int i = 0; i < NOPS; ++i) {
          const size_t me = (size_t)(l * NOPS + i) * 2, prev = me - 2;
          opus[i] += (double)(stv[me] - stv[prev + 1]) / 100.0;          // this op's publish - previous gather's end
          hous[i] += (double)(stv[me + 1] - stv[me]) / 100.0;            // gather behind this op
        }
      const char* names[NOPS] = {"o-proj", "gate_up", "down", "q/k/v"};
      printf("   workgroup 0:");
      for (int i = 0; i < NOPS; ++i) printf("  %s %.2f + hand-over %.2f us", names[i], opus[i] / (layers - 2), hous[i] / (layers - 2));
      printf("\n");
    }
    // the last op's outputs against the launches variant (same arithmetic, same order)
    std::vector<unsigned long long> gl((size_t)last_ng);
    CK(hipMemcpy(gl.data(), p.g[NOPS - 1], gl.size() * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (size_t i = 0; i < gl.size(); ++i) {
      const uint16_t a = (uint16_t)gl[i], b = (uint16_t)(gl[i] >> 16);
      if (a != ref[2 * i] || b != ref[2 * i + 1]) ++bad;
    }
    printf("   last op's outputs vs the launches variant: %ld of %zu granules differ\n", bad, gl.size());
  }
  return 0;
}
