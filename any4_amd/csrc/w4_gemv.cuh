// w4_gemv.cuh -- the W4A16 kernel for ONE layer per launch with 1 ... 4 (matrix-core contraction: ... 8) activation rows: what a batch-1 decode step issues
// four times per decoder layer (BASELINE config 5; the reference times it through HuggingFace's LlamaDecoderLayer,
// benchmark.py:113-215) and what Any4Linear.forward / Int4Linear.forward issue at batch 1 (modules.py:207-227, 56-80).
//
// Same contract and numerics as w4_gemm_pair.cuh (TG_NUM_FAST: the per-group affine map applied to f32 partial sums; reference
// TinyGemmImpl.cuh:23-345 with BLayout_TC_int4, MatrixLayoutB.cuh:686-1101, Dequantization.cuh:55-178), Bint4 weights at
// innerKTiles 4, int4 / any4 (global or row-wise LUT).  What is different is the decomposition, chosen for a launch that is
// latency-bound (tools/ubench/graph_chain.hip: a dependent graph node costs 2.8 us before its first byte and streams at
// ~6.3 TB/s after it, so a 9 MB layer is 4.1 us of which 1.4 are bandwidth):
//
//   grid       = one 8-wave workgroup per CU, each owns a CONTIGUOUS range of 8-row tiles of the packed layout and the whole k:
//                every CU streams from the first cycle whatever the layer's row count (4096 rows = 2 tiles per CU, 6144 = 3,
//                28672 = 14), nothing is persistent, nothing is reduced across workgroups.
//   pass       = P = 8, 16 or 32 weight rows of the range at a time (8 / 16 for ranges of one / two tiles, 16 and several passes for
//                longer ones; 32 only where k forces it: tg_gemv.hip).  lane = (row of the pass, sub-slot); a sub-slot is (k super-tile offset, half of the
//                super-tile's lane-quads): P = 32 -> 2 sub-slots (the halves), P = 16 -> 4 (2 super-tiles), P = 8 -> 8 (4).
//                A wave-load is therefore always whole 256-byte super-tile blocks of the packed layout, and the lanes of a
//                32-lane LDS access group always use 32 distinct table columns (column = lane & 31): conflict-free lookups.
//   step       = one 16-byte load per lane = the 32 codes of ONE row in ONE half of ONE 64-k super-tile: 16 pair lookups
//                (v_perm_b32 address + ds_read_b32), 4 activation pieces per activation row (ds_read_b128, broadcast), 16
//                v_dot2_f32_bf16 per activation row, then ONE scale / zero update: y += scale * dot + zero * sum(x of the step)
//                -- the group-scaled sum of w4_gemm_pair.cuh regrouped per step (a step never straddles a group for g >= 64;
//                g = 32 splits it by 32-k chunk, GPS = 2).  This is the contraction of groups of 32 / 64 and of what the matrix-core
//                variant (MF, below) declines; on the power-capped STACKED kernel v_dot2 is 6 % faster than the 32x32x16 MFMA at
//                m = 1 -- in this latency-bound launch the 16x16x32 MFMA of MF measured faster from one row on.
//   split-K    = the 8 waves of a workgroup split k; per pass their partial sums meet in LDS, added in wave order by P * m
//                threads (deterministic), with the epilogue (bias / residual add, SwiGLU of gate / up row pairs) in that store.
//   ring       = D steps per lane in flight (registers), running ACROSS passes (a pass occupies a whole number of rounds of D
//                slots; slots past its last step re-read that step and are skipped); refills are unconditional (a conditional
//                refill makes hipcc wait vmcnt(0) at the next use), the launch's last round is peeled without refills.
//   tables     = [256 byte values][64 columns] x 4 bytes at LDS address 0 as in the other pair-table kernels (address = byte << 8
//                | column << 2, one v_perm_b32); a pass uses 32 columns, so the other 32 hold the NEXT pass's table (row-wise
//                LUT), built from an LDS copy of the range's LUT rows before the pass's only barrier.
//   activations= all m rows staged once in the w4_gemm_pair.cuh byte order, one 16-byte piece (8 values) per thread, with the
//                f32 sums of every (super-tile, half, chunk) next to them; optionally LlamaRMSNorm'ed on the way (NORM,
//                dg_add_rmsnorm's rounding points): every WAVE adds the squares of the whole row itself (k / 512 loads per lane
//                from L2, one butterfly) -- no barrier, no LDS round trip in front of the staging.
//   latency    = what a launch of this kind is made of (dev/gemv_trace.py, s_memrealtime stamps inside a decode step): ~1.2 us
//                from the previous kernel's last wave to this one's first, then the time to ISSUE the first loads, one memory
//                round trip (~1 us), the steps, the split-K tail.  Hence: no integer division and no dependent scalar load in
//                front of the first requests, the residual / bias values requested up front, one inlined copy of the pass tail.
#pragma once

#ifndef GEMV_TRACE
#define GEMV_TRACE 0  // developer builds: s_memrealtime stamps of workgroup phases into GemvParams.trace
#endif

struct GemvParams {
  const char* x;
  const char* w;
  const char* qinfo;
  const char* lut;
  char* y;
  const char* bias;
  const char* norm_w;
  int64_t stride_x, stride_w, stride_qinfo, stride_lut, stride_y, stride_bias;
  int64_t bias_row_stride;
  int32_t m, wrows, k;
  int32_t ntiles;    // packed.size(0): 8-row tiles
  int32_t ksuper;    // packed.size(1): 64-k super-tiles
  int32_t qtype;
  int32_t sg_shift;  // log2(super-tiles per quantisation group), g >= 64 (GPS = 1)
  int32_t P, p_shift;  // weight rows per pass (8, 16, 32) and its log2
  int32_t unit;      // tiles are dealt to workgroups in units of this many (2 with the SwiGLU epilogue: a gate / up block):
  int32_t ubase, urem;  // workgroup b owns ubase + (b < urem ? uextra : 0) units, starting at unit b * ubase + min(b, urem) * uextra
  int32_t uextra;       // 1; more when the first `urem` workgroups are the FIRST on their CU and should stream longer (two workgroups per CU)
  int32_t uh;           // 0; else (two workgroups per CU, urem = CUs): the pairs (b, b + urem) with b < uh move ONE more unit from the second workgroup to the first
  int32_t spw;       // k super-tiles per wave
  int32_t spp;       // steps per pass and wave
  int32_t rounds;    // ceil(spp / D): rounds of D ring slots a pass occupies
  int32_t x_pitch, xs_pitch;  // bytes per staged activation row / per row of its sums
  int32_t lds_lut, lds_x, lds_xs, lds_red, lds_nrm;  // LDS byte offsets (lds_nrm: 8 x m f32 partial sums of squares)
  float norm_eps;
  int32_t epilogue;
  int32_t xcd4;      // 1: four consecutive row ranges per XCD (no remainder ranges, workgroups a multiple of 32)
  int32_t cm;        // 1: chunk-mode staging (5 ... 8 activation rows on the matrix-core path, k = 2048 / 4096: w4_gemv_kernel, `cm`)
  unsigned long long* trace;
};

// DT = BF16 / F16, M = activation rows (1 ... 4; MF: 3 ... 8), GPS = groups per super-tile half-step (1: g >= 64, 2: g = 32), D = ring depth,
// NORM = RMSNorm fused into the staging
// MF  = the contraction on the matrix core (built for 3 ... 8 rows, where four v_dot2 per packed word and ROW make the vector ALU the bound
//       -- gate_up of Llama-3-8B at 4 rows: 30.8 us against 17.9 at one --, measured faster from ONE row on: tg_gemv.hip, TG_GEMV_MF_MIN_M).  16-row passes only (P = 16, the host's choice): lane (n = lane & 15, sub =
//       lane >> 4) holds the words of weight row n at the k-subset `sub` of the step (super-tile sub >> 1, quads 2 (sub & 1) + qq) -- one B
//       operand of v_mfma_f32_16x16x32 per looked-up word; the A operand of the same lane is the matching 16-byte piece of activation
//       row lane & 15 (rows >= M read row M - 1: their accumulator rows are never stored), so ONE LDS read per word serves all rows.
//       D[a][n]: lane (n, sub) holds rows 4 sub + r.  A step (two super-tiles) lies inside one quantisation group (g >= 128, the host
//       checks): one scale / zero update per step, the step's activation sums [step][16 rows] come from the staging.
#ifndef TG_GEMV_MF_ONES
#define TG_GEMV_MF_ONES 1  // MF: a step's activation sums (the zero-point term) from the matrix core -- four more MFMAs per step against an all-ones B operand leave sum_k x[4 sub + r][k] in
                           // the lane that needs it -- instead of from the staging (8 two-element dot products per piece, a 16-lane butterfly per row = four LDS-latency
                           // shuffles on the launch's critical path in front of the first barrier, an LDS write, and a 16-byte LDS read per step); 0: developer A/B
#endif
template <typename DT, int M, int GPS, int D, bool NORM, bool MF = false>
__global__ void __launch_bounds__(512, 2) w4_gemv_kernel(const GemvParams p) {
  static_assert(!MF || GPS == 1, "matrix-core contraction: groups of at least two super-tiles");
#ifndef TG_GEMV_MF_ONES_MIN_M
#define TG_GEMV_MF_ONES_MIN_M 5  // (one row in a long layer: the four extra MFMAs per step cost more than the staging saves -- 28672 x 4096 15.6 vs 15.0 us)
#endif
  constexpr bool MFS = MF && TG_GEMV_MF_ONES && M >= TG_GEMV_MF_ONES_MIN_M;
  constexpr int NW = 8, NT = NW * 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int b = 0;  // (ONE problem per launch, the host's condition: no stride arithmetic on the launch's critical path)
#if GEMV_TRACE
  unsigned long long tr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) tr[i] = 0;
  tr[0] = __builtin_amdgcn_s_memrealtime();
#endif

  // Every kernel argument into scalar registers NOW: left alone, hipcc loads them where they are first used -- four dependent
  // scalar-memory round trips (~0.3 us each from the cold argument segment) in front of the first weight request.
  asm volatile("" ::"s"(p.x), "s"(p.w), "s"(p.qinfo), "s"(p.lut), "s"(p.y), "s"(p.bias), "s"(p.norm_w), "s"(p.stride_x), "s"(p.stride_w),
               "s"(p.stride_qinfo), "s"(p.stride_lut), "s"(p.stride_y), "s"(p.stride_bias), "s"(p.bias_row_stride));
  asm volatile("" ::"s"(p.m), "s"(p.wrows), "s"(p.k), "s"(p.ksuper), "s"(p.qtype), "s"(p.sg_shift), "s"(p.P), "s"(p.p_shift), "s"(p.unit),
               "s"(p.ubase), "s"(p.urem), "s"(p.spw), "s"(p.spp), "s"(p.rounds), "s"(p.x_pitch), "s"(p.xs_pitch), "s"(p.lds_lut),
               "s"(p.lds_x), "s"(p.lds_xs), "s"(p.lds_red), "s"(p.lds_nrm), "s"(p.norm_eps), "s"(p.epilogue), "s"(p.xcd4), "s"(p.cm), "s"(p.uextra), "s"(p.uh));
#if GEMV_TRACE
  tr[6] = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- this workgroup's tiles, this lane's place in a pass ----
  // Four CONSECUTIVE ranges go to the workgroups b, b + 8, b + 16, b + 24 -- one XCD (workgroup b runs on XCD b % 8: observed, used
  // for speed only): a 128-byte line of scale | zero words covers 32 weight rows, ranges of 16 / 24 / 112 rows (Llama-3-8B's
  // projections over 256 CUs) share such lines with their neighbours, and neighbours on different XCDs fetch them from HBM twice.
  // (`xcd4`: the host's test of the grid -- gridDim itself is a hidden kernel argument, i.e. one more scalar-memory round trip at the very
  //  top of the launch)
  int bx = blockIdx.x;
  if (p.xcd4) bx = (((bx >> 5) * 8 + (bx & 7)) << 2) + ((bx >> 3) & 3);
  int t0 = bx * p.ubase + min(bx, p.urem) * p.uextra, nu = p.ubase + (bx < p.urem ? p.uextra : 0);
  if (p.uh) {  // (wave-uniform) first workgroups b < uh: one unit more; their partners b + urem: one less
    const int c = bx < p.urem ? bx : bx - p.urem;
    t0 += bx < p.urem ? min(c, p.uh) : p.uh - min(c, p.uh);
    nu += c < p.uh ? (bx < p.urem ? 1 : -1) : 0;
  }
  t0 *= p.unit;
  const int t1 = t0 + nu * p.unit;
  if (t0 >= t1) return;
  const int P = p.P, Pm = P - 1, tpp = P >> 3;        // tiles per pass
  const int passes = (t1 - t0 + tpp - 1) >> (p.p_shift - 3);
  const int row_l = lane & Pm;
  const int sub = lane >> p.p_shift;
  const int h = sub & 1, ss = sub >> 1;
  const int SS = 32 >> p.p_shift;                      // super-tiles per step
  const int s_begin = wave * p.spw;
  const int s_end = min(s_begin + p.spw, p.ksuper);
  const int spp = p.spp;                               // real steps of a pass
  const int rounds = p.rounds;                         // rounds of D ring slots a pass occupies: ceil(spp / D)

  const char* wb = p.w + (int64_t)b * p.stride_w;
  const char* qb = p.qinfo + (int64_t)b * p.stride_qinfo;
  const char* lb = p.lut + (int64_t)b * p.stride_lut;
  const char* xb = p.x + (int64_t)b * p.stride_x;

  // ---- the ring ----
  struct Slot {
    u32x4 w;
    uint32_t q[GPS];
  };
  Slot ring[D];
  // issue pointer: pass ip, slot ii of the pass (slots >= spp re-read the pass's last step); per-pass lane offsets
  int ip = 0, ii = 0;
  // Addressing: every address of the main loop = (wave-uniform part that moves with the step, scalar registers) + (per-lane
  // offset fixed for the pass).  The host guarantees ksuper % SS == 0 (it raises P otherwise), so a step's SS super-tiles are
  // all inside the matrix or all outside: no per-lane clamp, and (su + ss) >> sg == (su >> sg) + (ss >> sg) for the group index.
  uint32_t wlane, qlane;  // byte offsets of this lane's row and sub-slot in super-tile 0 / in group 0's scale | zero words
  const uint32_t qrow_bytes = (uint32_t)p.wrows * 4u;
  auto pass_lane = [&](int pass) {
    const int tile = min(t0 + pass * tpp + (row_l >> 3), t1 - 1);
    wlane = (uint32_t)tile * (uint32_t)p.ksuper * 256u + (uint32_t)((row_l & 7) * 32 + h * 16 + ss * 256);
    qlane = (uint32_t)(tile * 8 + (row_l & 7)) * 4u + (uint32_t)(GPS == 1 ? (ss >> p.sg_shift) : 2 * ss) * qrow_bytes;
  };
  pass_lane(0);
  const int slots = rounds * D;
  const int su_max = p.ksuper - SS;
  auto issue = [&](Slot& sl) {
    const int su = min(s_begin + min(ii, spp - 1) * SS, su_max);  // (scalar) first super-tile of the step; past the end: the last one
    sl.w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wb + (uint32_t)su * 256u + wlane));
    if constexpr (GPS == 1) {
      sl.q[0] = *reinterpret_cast<const uint32_t*>(qb + (uint32_t)(su >> p.sg_shift) * qrow_bytes + qlane);
    } else {
      sl.q[0] = *reinterpret_cast<const uint32_t*>(qb + (uint32_t)(2 * su) * qrow_bytes + qlane);
      sl.q[1] = *reinterpret_cast<const uint32_t*>(qb + (uint32_t)(2 * su + 1) * qrow_bytes + qlane);
    }
    if (++ii == slots) {  // (wave-uniform) the next slot belongs to the next pass; past the last pass the last slot is re-read
      ii = 0;
      if (ip + 1 < passes) pass_lane(++ip);
      else ii = slots - 1;
    }
  };

#ifndef TG_GEMV_PRE_BARRIER
#define TG_GEMV_PRE_BARRIER 1  // 1: the first weight requests behind the staging requests AND a workgroup barrier (below); developer A/B: 0 = no barrier
                               // (every wave requests its weights right behind its own staging requests), 2 = the weights FIRST
#endif
#if TG_GEMV_PRE_BARRIER == 2
#pragma unroll
  for (int j = 0; j < D; ++j) {
    __builtin_amdgcn_sched_barrier(0);
    issue(ring[j]);
  }
  __builtin_amdgcn_sched_barrier(0);
#endif
  // ---- requests: this thread's share of the activations, norm weights, LUT rows, bias values -- then, behind a workgroup barrier,
  // the first D steps of the weight stream (the CU's vector-memory path takes requests in arrival order: 64 KiB of weight
  // requests of the waves that got there first would sit in front of the last wave's 16 bytes of activations) ----
  // Activation staging: a 32-k chunk becomes four 16-byte pieces; piece q = the chunk's dwords q, q + 4, q + 8, q + 12
  // (x[2q], x[2q+1] | +8 | +16 | +24).  k <= 4096: thread t stages piece t of every row (four dword loads); larger k: thread t
  // stages chunk t (four 16-byte loads).  Lanes without a share re-read the last one: every load of this prologue is
  // unconditional per lane (a load under a lane mask made hipcc wait vmcnt(0) right behind it).
  const int npc = p.k >> 3;   // pieces per activation row
  const int nch = p.k >> 5;   // chunks per activation row (host: <= NT)
  const bool wide = npc > NT; // (wave-uniform)
  uint32_t xd[M][16];         // !wide: [0..3] = the piece's four dwords; wide: the chunk's 16 dwords
  uint32_t gwd[NORM ? 16 : 1];
  auto stage_load = [&](const char* base, uint32_t (&d)[16]) {
    if (!wide) {
      const int pc = min(tid, npc - 1);
      const char* src = base + (pc >> 2) * 64 + (pc & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = *reinterpret_cast<const uint32_t*>(src + j * 16);
#pragma unroll
      for (int j = 4; j < 16; ++j) d[j] = 0u;
    } else {
      const u32x4* src = reinterpret_cast<const u32x4*>(base + min(tid, nch - 1) * 64);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 v = src[j];
        d[4 * j] = v[0]; d[4 * j + 1] = v[1]; d[4 * j + 2] = v[2]; d[4 * j + 3] = v[3];
      }
    }
  };
  // Chunk mode (5 ... 8 rows, k = 2048 / 4096): thread t of round j owns the WHOLE 32-k chunk t % nch of row j (NT / nch) + t / nch --
  // four 16-byte loads per round instead of four 4-byte loads per ROW (M = 8: 8 wide loads per thread instead of 32 narrow ones;
  // the narrow form touches every 64-byte line of the block four times and was ~2 us of the launch's critical path), a wave's 64
  // chunks belong to ONE row.  xd[j] = the chunk of round j.
  bool cm = false;
  int cm_row = 0, cm_chunk = 0, cm_rpr = 1;  // first row of this thread, its chunk, rows per round
  if constexpr (M >= 5) {
    cm = p.cm != 0;
    cm_rpr = NT / nch;
    cm_row = tid / nch;
    cm_chunk = tid - cm_row * nch;
  }
  if (M >= 5 && cm) {
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (j * cm_rpr < M) {  // (wave-uniform)
        const int row = min(j * cm_rpr + cm_row, M - 1);
        const u32x4* src = reinterpret_cast<const u32x4*>(xb + (int64_t)row * p.k * 2 + cm_chunk * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32x4 v = src[q];
          xd[j][4 * q] = v[0]; xd[j][4 * q + 1] = v[1]; xd[j][4 * q + 2] = v[2]; xd[j][4 * q + 3] = v[3];
        }
      }
    }
    if constexpr (NORM) {
      const u32x4* src = reinterpret_cast<const u32x4*>(p.norm_w + cm_chunk * 64);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4 v = src[q];
        gwd[4 * q] = v[0]; gwd[4 * q + 1] = v[1]; gwd[4 * q + 2] = v[2]; gwd[4 * q + 3] = v[3];
      }
    }
  } else {
#pragma unroll
  for (int a = 0; a < M; ++a) stage_load(xb + (int64_t)a * p.k * 2, xd[a]);
  if constexpr (NORM) stage_load(p.norm_w, gwd);
  }
  // LUT: the 16 values of table column c of pass 0 as 8 packed pairs
  const int c = tid & 31;   // table column this thread builds
  const int hi = tid >> 5;  // ... for the bytes with this high nibble (16 x 32 = 512 threads)
  uint32_t lp[8];
  const bool rowwise = p.qtype == TG_Q_ANY4_ROWWISE;
  const int wg_rows = (t1 - t0) * 8;
  if (p.qtype == TG_Q_INT4) {
#pragma unroll
    for (int e = 0; e < 16; e += 2) lp[e >> 1] = DT::pack2((float)(e - 8), (float)(e - 7));
  } else {
    const int trow = min(t0 + ((c & Pm) >> 3), t1 - 1) * 8 + (c & 7);
    const char* lsrc = lb + (rowwise ? (int64_t)trow * 32 : 0);
    const u32x4 l0 = reinterpret_cast<const u32x4*>(lsrc)[0];
    const u32x4 l1 = reinterpret_cast<const u32x4*>(lsrc)[1];
#pragma unroll
    for (int j = 0; j < 4; ++j) { lp[j] = l0[j]; lp[4 + j] = l1[j]; }
  }
  // the range's LUT rows -> LDS (the tables of the passes after the first are built from there)
  const bool stage_lut = rowwise && passes > 1;
  u32x4 lstage = {0, 0, 0, 0};
  if (stage_lut) lstage = reinterpret_cast<const u32x4*>(lb + (int64_t)t0 * 256)[min(tid, wg_rows * 2 - 1)];
  // the bias / residual values of the first pass's outputs (thread = (activation row, row of the pass), as in pass_end)
  uint16_t res0 = 0;
  if (p.bias) {
    const int a = min(tid >> p.p_shift, M - 1), r = tid & Pm;
    const int row = min(t0 + (r >> 3), t1 - 1) * 8 + (r & 7);
    res0 = *reinterpret_cast<const uint16_t*>(p.bias + (int64_t)b * p.stride_bias + ((int64_t)a * p.bias_row_stride + row) * 2);
  }
  // the first D steps of the weight stream, behind everything the staging in front of the first barrier waits for (vector
  // memory returns in order: requested first, the staging would wait for HBM instead of L2) -- on every wave of the workgroup
#ifndef TG_GEMV_ASM_BARRIER
#define TG_GEMV_ASM_BARRIER 1
#endif
#if TG_GEMV_PRE_BARRIER == 1
#if TG_GEMV_ASM_BARRIER
  asm volatile("s_barrier" ::: "memory");  // (spelled out: in front of the builtin hipcc waits vmcnt(0) -- the staging loads would have to RETURN before the first weight request)
#else
  __builtin_amdgcn_s_barrier();
#endif
#endif
#if TG_GEMV_PRE_BARRIER != 2
#pragma unroll
  for (int j = 0; j < D; ++j) {
    __builtin_amdgcn_sched_barrier(0);
    issue(ring[j]);
  }
  __builtin_amdgcn_sched_barrier(0);
#endif
#if GEMV_TRACE
  tr[1] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- stage the activations (byte order of w4_gemm_pair.cuh) and their sums ----
  const uint32_t lds_x = (uint32_t)p.lds_x, lds_xs = (uint32_t)p.lds_xs, lds_red = (uint32_t)p.lds_red;
  // one piece -> LDS (byte order); returns the sum of its 8 values
  auto piece_store = [&](int a, int pc, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, bool on) -> float {
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);  // x[2q]     x[2q+8]
    o[1] = __builtin_amdgcn_perm(d3, d2, 0x05040100u);  // x[2q+16]  x[2q+24]
    o[2] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);  // x[2q+1]   x[2q+9]
    o[3] = __builtin_amdgcn_perm(d3, d2, 0x07060302u);  // x[2q+17]  x[2q+25]
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sum = dot2_ones<DT>(o[j], sum);
    if (on) *(lds_u32x4ptr)(lds_x + (uint32_t)(a * p.x_pitch + pc * 16)) = o;
    return sum;
  };
  // sums of a chunk's halves: [super-tile][half][chunk of the super-tile]
  auto xs_store = [&](int a, int ch, int hh, float sum) {
    *(lds_fptr)(lds_xs + (uint32_t)(a * p.xs_pitch + (((ch >> 1) * 2 + hh) * 2 + (ch & 1)) * 4)) = sum;
  };
  if (M >= 5 && cm) {
    // chunk mode: round j stages chunk cm_chunk of row j cm_rpr + cm_row: its four pieces, the step's sum (four consecutive chunks =
    // four consecutive threads), the row's sum of squares (a wave's chunks are one row's: every wave writes ALL M partials, zeros
    // for the rows it does not hold)
    float nrm[M];
#pragma unroll
    for (int a = 0; a < M; ++a) nrm[a] = 0.f;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (j * cm_rpr < M) {  // (wave-uniform)
        const int a = j * cm_rpr + cm_row;
        const bool on = a < M;
        if constexpr (NORM) {
          float v = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            if constexpr (std::is_same<DT, BF16>::value)
              v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xd[j][e]), __builtin_bit_cast(bf16x2, xd[j][e]), v, false);
            else
              v = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, xd[j][e]), __builtin_bit_cast(f16x2, xd[j][e]), v, false);
          }
          v = on ? v : 0.f;
          v = tgl::wave_sum(v);
#pragma unroll
          for (int a2 = 0; a2 < M; ++a2) nrm[a2] = a2 == a ? v : nrm[a2];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const uint32_t xv = xd[j][e], g = gwd[e];
            xd[j][e] = DT::pack2(DT::lo_f32(xv) * DT::lo_f32(g), DT::hi_f32(xv) * DT::hi_f32(g));
          }
        }
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += piece_store(on ? a : 0, 4 * cm_chunk + q, xd[j][q], xd[j][q + 4], xd[j][q + 8], xd[j][q + 12], on);
        if constexpr (!MFS) {
          sum = on ? sum : 0.f;
          sum += tgl::lane_xor<1>(sum, lane);
          sum += tgl::lane_xor<2>(sum, lane);
          if (on && (cm_chunk & 3) == 0) *(lds_fptr)(lds_xs + (uint32_t)((cm_chunk >> 2) * 64 + a * 4)) = sum;
        }
      }
    }
    if constexpr (NORM) {
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < M; ++a) *(lds_fptr)((uint32_t)p.lds_nrm + (uint32_t)((wave * M + a) * 4)) = nrm[a];
      }
    }
  } else {
#pragma unroll
  for (int a = 0; a < M; ++a) {
    const bool on = wide ? tid < nch : tid < npc;
    if constexpr (NORM) {
      // LlamaRMSNorm as  y = rs * sum_k w_k x'_k,  x'_k = RNE16(x_k g_k),  rs = rsqrt(mean(x^2) + eps)  applied to the f32 sum in the
      // output store: the launch does not wait for a reduction over the activations before it can stage them.  Squares of this
      // thread's share -> wave -> one partial per wave and row in LDS, added in wave order by the threads that store the outputs.
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (std::is_same<DT, BF16>::value)
          v = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xd[a][j]), __builtin_bit_cast(bf16x2, xd[a][j]), v, false);
        else
          v = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, xd[a][j]), __builtin_bit_cast(f16x2, xd[a][j]), v, false);
      }
      v = on ? v : 0.f;
      v = tgl::wave_sum(v);
      if (lane == 0) *(lds_fptr)((uint32_t)p.lds_nrm + (uint32_t)((wave * M + a) * 4)) = v;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t xv = xd[a][j], g = gwd[j];
        xd[a][j] = DT::pack2(DT::lo_f32(xv) * DT::lo_f32(g), DT::hi_f32(xv) * DT::hi_f32(g));
      }
    }
    if (MF && M <= 4 && wide) {  // (host: long k only with up to four rows)
      // a thread stages a 32-k chunk; a STEP's 128 k are four consecutive threads
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += piece_store(a, 4 * tid + q, xd[a][q], xd[a][q + 4], xd[a][q + 8], xd[a][q + 12], on);
      if constexpr (!MFS) {
        sum = on ? sum : 0.f;
        sum += tgl::lane_xor<1>(sum, lane);
        sum += tgl::lane_xor<2>(sum, lane);
        if (on && (tid & 3) == 0) *(lds_fptr)(lds_xs + (uint32_t)((tid >> 2) * 64 + a * 4)) = sum;
      }
    } else if (MF) {
      // the sums of a STEP's 128 k (the zero-point term is added per step): its 16 pieces are 16 consecutive threads
      float sum = piece_store(a, tid, xd[a][0], xd[a][1], xd[a][2], xd[a][3], on);
      if constexpr (!MFS) {
        sum = on ? sum : 0.f;
        sum = tgl::row_sum(sum);  // (tg_common.cuh: DPP rotations; four shuffles here were 0.6 us of the launch's critical path)
        if (on && (tid & 15) == 0) *(lds_fptr)(lds_xs + (uint32_t)((tid >> 4) * 64 + a * 4)) = sum;
      }
    } else if (!wide) {
      float sum = piece_store(a, tid, xd[a][0], xd[a][1], xd[a][2], xd[a][3], on);
      sum += tgl::lane_xor<1>(sum, lane);  // the two quads of a half sit in adjacent lanes
      if (on && (tid & 1) == 0) xs_store(a, tid >> 2, (tid >> 1) & 1, sum);
    } else {
      float sq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) sq[q] = piece_store(a, 4 * tid + q, xd[a][q], xd[a][q + 4], xd[a][q + 8], xd[a][q + 12], on);
      if (on) {
        xs_store(a, tid, 0, sq[0] + sq[1]);
        xs_store(a, tid, 1, sq[2] + sq[3]);
      }
    }
  }
  }  // !cm
  if (stage_lut && tid < wg_rows * 2) *(lds_u32x4ptr)((uint32_t)p.lds_lut + (uint32_t)tid * 16u) = lstage;
  if (stage_lut)
    for (int i = tid + NT; i < wg_rows * 2; i += NT)
      *(lds_u32x4ptr)((uint32_t)p.lds_lut + (uint32_t)i * 16u) = reinterpret_cast<const u32x4*>(lb + (int64_t)t0 * 256)[i];

  // ---- pair table of a pass: entry[byte][column] = (lut[byte & 15], lut[byte >> 4]); thread = (column c, high nibble hi) ----
  auto build_table = [&](int half) {
    uint32_t hsel = lp[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) hsel = (hi >> 1) == j ? lp[j] : hsel;
    const uint32_t hv = (hi & 1) ? (hsel & 0xffff0000u) : (hsel << 16);
    const lds_u32ptr tb = (lds_u32ptr)((uint32_t)(hi * 16 * 256 + (half * 32 + c) * 4));
#pragma unroll
    for (int lo = 0; lo < 16; ++lo) {
      const uint32_t lv = (lo & 1) ? (lp[lo >> 1] >> 16) : (lp[lo >> 1] & 0xffffu);
      tb[lo * 64] = lv | hv;
    }
  };
  build_table(0);
#if GEMV_TRACE
  tr[2] = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();
#if GEMV_TRACE
  tr[3] = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- main loop ----
  uint32_t colreg = (uint32_t)((lane & 31) * 4);
  float yacc[MF ? 4 : M];  // MF: rows 4 (lane >> 4) + r of weight row lane & 15
#pragma unroll
  for (int a = 0; a < (MF ? 4 : M); ++a) yacc[a] = 0.f;

  auto pass_end = [&](int cp) {
    const uint32_t par = (uint32_t)(cp & 1);
    if constexpr (MF) {
      // the matrix core has summed the k-subsets of the four lane quarters already: lane (n, sub) stores rows 4 sub + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 4 * (lane >> 4) + r;
        if (a < M) *(lds_fptr)(lds_red + (uint32_t)((((par * NW + wave) * M + a) * 32 + (lane & 15)) * 4)) = yacc[r];
        yacc[r] = 0.f;
      }
    } else {
    // the sub-slots of a row sit in lanes row + P i: lane `row` gets the wave's sum
    float v[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
      v[a] = yacc[a];
      yacc[a] = 0.f;
      v[a] = tgl::halves32_sum(v[a]);
      if (P <= 16) v[a] = tgl::rows16_sum(v[a]);
      if (P <= 8) v[a] += tgl::lane_xor<8>(v[a], lane);
    }
    if (lane < P) {
#pragma unroll
      for (int a = 0; a < M; ++a) *(lds_fptr)(lds_red + (uint32_t)((((par * NW + wave) * M + a) * 32 + lane) * 4)) = v[a];
    }
    }
    // the next pass's table goes into the other 32 columns (row-wise LUT only: the others never change)
    if (rowwise && cp + 1 < passes) {
      const int lrow = min((cp + 1) * P + (c & Pm), wg_rows - 1);
      const u32x4 l0 = *(lds_cu32x4ptr)((uint32_t)p.lds_lut + (uint32_t)lrow * 32u);
      const u32x4 l1 = *(lds_cu32x4ptr)((uint32_t)p.lds_lut + (uint32_t)lrow * 32u + 16u);
#pragma unroll
      for (int j = 0; j < 4; ++j) { lp[j] = l0[j]; lp[4 + j] = l1[j]; }
      build_table((cp + 1) & 1);
      colreg ^= 128u;
    }
    __syncthreads();
    if (tid < P * M) {
      const int a = tid >> p.p_shift, r = tid & Pm;
      const int tile = t0 + cp * tpp + (r >> 3);
      if (tile < t1) {
        const int row = tile * 8 + (r & 7);
        float sum = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NW; ++w8) sum += *(lds_fptr)(lds_red + (uint32_t)((((par * NW + w8) * M + a) * 32 + r) * 4));
        float rsn = 1.f;
        if constexpr (NORM) {
          float tot = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < NW; ++w8) tot += *(lds_fptr)((uint32_t)p.lds_nrm + (uint32_t)((w8 * M + a) * 4));
          rsn = rsqrtf(tot * (1.0f / (float)p.k) + p.norm_eps);
          sum *= rsn;
        }
        if (p.epilogue == TG_EPI_SWIGLU) {
          // rows come in blocks of 8 gate + 8 up (a block = two tiles of one pass): lane r + 8 holds the up row of gate row r
          if ((r & 15) < 8) {
            float up = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) up += *(lds_fptr)(lds_red + (uint32_t)((((par * NW + w8) * M + a) * 32 + r + 8) * 4));
            up *= rsn;
            *reinterpret_cast<uint16_t*>(p.y + (int64_t)b * p.stride_y + ((int64_t)a * (p.wrows >> 1) + ((row >> 4) << 3) + (row & 7)) * 2) = swiglu16<DT>(sum, up);
          }
        } else {
          uint16_t o16 = DT::from_f32(sum);
          if (p.bias) {  // rounded sum + bias, rounded again: the reference module's separate `y + bias` (modules.py:221-222)
            uint16_t bv = res0;  // (the first pass's were requested with the first loads)
            if (cp > 0) bv = *reinterpret_cast<const uint16_t*>(p.bias + (int64_t)b * p.stride_bias + ((int64_t)a * p.bias_row_stride + row) * 2);
            o16 = DT::from_f32(DT::lo_f32(o16) + DT::lo_f32(bv));
          }
          *reinterpret_cast<uint16_t*>(p.y + (int64_t)b * p.stride_y + ((int64_t)a * p.wrows + row) * 2) = o16;
        }
      }
    }
  };

  const uint32_t xlane = lds_x + (uint32_t)(ss * 128 + h * 32), xslane = lds_xs + (uint32_t)((ss * 2 + h) * 8);
  auto consume = [&](const Slot& sl, int i) {  // step i of the current pass (wave-uniform, < spp)
    const int su_l = s_begin + i * SS;
    const bool on = su_l < s_end;          // (wave-uniform: a step is inside the wave's slice or not)
    const int su = min(su_l, su_max);
    const uint32_t xa = xlane + (uint32_t)su * 128u;
    const uint32_t xsa = xslane + (uint32_t)su * 16u;
    if constexpr (MF) {
      const uint32_t xr = xa + (uint32_t)(min(lane & 15, M - 1) * p.x_pitch);
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      u32x4 e4[4], xf4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jc = u >> 1, qq = u & 1;
        const uint32_t w = sl.w[qq * 2 + jc];
#pragma unroll
        for (int j = 0; j < 4; ++j) e4[u][j] = *(lds_cu32ptr)(__builtin_amdgcn_perm(w, colreg, 0x0c0c0400u + ((uint32_t)j << 8)));
        xf4[u] = *(lds_cu32x4ptr)(xr + (uint32_t)(jc * 64 + qq * 16));
      }
      f32x4_t gs4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!MFS) {
        const f32x4 v = *(lds_cf32x4ptr)(lds_xs + (uint32_t)((su >> 1) * 64 + (lane >> 4) * 16));
        gs4 = f32x4_t{v[0], v[1], v[2], v[3]};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = mfma16<DT>(xf4[u], e4[u], acc);
      if constexpr (MFS) {
        const uint32_t one2 = DT::pack2(1.f, 1.f);
        const u32x4 ones = {one2, one2, one2, one2};
#pragma unroll
        for (int u = 0; u < 4; ++u) gs4 = mfma16<DT>(xf4[u], ones, gs4);
      }
      const float sc = on ? DT::lo_f32(sl.q[0]) : 0.f;
      const float zz = on ? DT::hi_f32(sl.q[0]) : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) yacc[r] = __builtin_fmaf(zz, gs4[r], __builtin_fmaf(sc, acc[r], yacc[r]));
      return;
    }
    float dsum[M][GPS];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int g = 0; g < GPS; ++g) dsum[a][g] = 0.f;
    // all 16 lookups and the 4 activation pieces per row of the step are requested before the first product: with two waves per
    // SIMD the LDS latency is otherwise paid once per packed word
    uint32_t e[4][4];
    u32x4 xf[M][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jc = u >> 1, qq = u & 1;  // 32-k chunk of the super-tile, quad of the half
      const uint32_t w = sl.w[qq * 2 + jc];
#pragma unroll
      for (int j = 0; j < 4; ++j) e[u][j] = *(lds_cu32ptr)(__builtin_amdgcn_perm(w, colreg, 0x0c0c0400u + ((uint32_t)j << 8)));
#pragma unroll
      for (int a = 0; a < M; ++a) xf[a][u] = *(lds_cu32x4ptr)(xa + (uint32_t)(a * p.x_pitch + jc * 64 + qq * 16));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int jc = u >> 1;
#pragma unroll
      for (int a = 0; a < M; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) dsum[a][GPS == 1 ? 0 : jc] = dot2_pair<DT>(e[u][j], xf[a][u][j], dsum[a][GPS == 1 ? 0 : jc]);
    }
#pragma unroll
    for (int g = 0; g < GPS; ++g) {
      const float sc = on ? DT::lo_f32(sl.q[g]) : 0.f;
      const float zz = on ? DT::hi_f32(sl.q[g]) : 0.f;
#pragma unroll
      for (int a = 0; a < M; ++a) {
        const f32x2 xs2 = *(const __attribute__((address_space(3))) f32x2*)(xsa + (uint32_t)(a * p.xs_pitch));
        const float xs = GPS == 1 ? xs2[0] + xs2[1] : xs2[g];
        yacc[a] = __builtin_fmaf(zz, xs, __builtin_fmaf(sc, dsum[a][g], yacc[a]));
      }
    }
  };

  // a pass = `rounds` rounds of D ring slots (slots >= spp are padding: requested, skipped); every round refills its slots with
  // the slots D further -- except the launch's last round
  for (int cp = 0; cp < passes; ++cp) {
    const bool last = cp + 1 == passes;
    const int nr = rounds - (last ? 1 : 0);
    for (int r = 0; r < nr; ++r) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if (r * D + j < spp) consume(ring[j], r * D + j);
        issue(ring[j]);
#if GEMV_TRACE
        if (cp == 0 && r == 0 && j == 0) tr[4] = __builtin_amdgcn_s_memrealtime();
#endif
      }
    }
    if (last) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        if ((rounds - 1) * D + j < spp) consume(ring[j], (rounds - 1) * D + j);
#if GEMV_TRACE
        if (cp == 0 && rounds == 1 && j == 0) tr[4] = __builtin_amdgcn_s_memrealtime();
#endif
      }
    }
    pass_end(cp);
  }
#if GEMV_TRACE
  tr[5] = __builtin_amdgcn_s_memrealtime();
  if (tid == 0 && p.trace) {
#pragma unroll
    for (int i = 0; i < 7; ++i) p.trace[(size_t)blockIdx.x * 8 + i] = tr[i];
  }
#endif
}
