// One-shot peer-write gather of row-sharded outputs (include/peer_gather_hip.h).  Included at the end of tinygemm_hip.hip:
// shares DeviceScope and launch_status().
//
// No counterpart in the reference (it has no multi-GPU code); the shape of the exchange is SURVEY.md 8(e): rank r owns the
// weight rows [r n/G, (r+1) n/G), so after its GEMM it holds y[:, r n/G : (r+1) n/G] and every rank needs all of y.
//
// One workgroup per destination rank p (G <= 16 workgroups): it stores this rank's [m][cols_local] slice into column block
// `rank` of p's gathered buffer with 16-byte stores (over xGMI for p != rank; every link of the point-to-point fabric carries
// one slice), fences at system scope, and raises flags_p[rank] to this call's sequence number.  The same workgroup then
// waits until flags_rank[p] has reached the sequence number: rank p's slice has landed HERE.  The wait is bounded
// (s_memrealtime, 100 MHz): a missing peer sets *status and has its slice filled with NaNs instead of hanging the GPU.
#include "../../include/peer_gather_hip.h"

namespace {

struct PeerGatherParams {
  const char* src;
  char* dst[TG_PEER_MAX_WORLD];
  uint32_t* flags[TG_PEER_MAX_WORLD];
  uint32_t* seq;     // [world] words of this rank: workgroup p keeps its own count (no cross-workgroup ordering needed)
  uint32_t* status;
  int32_t world, rank;
  int32_t m;
  int32_t row_bytes;        // cols_local * 2
  int64_t dst_pitch;        // world * row_bytes
  int64_t timeout_ticks;
};

__global__ void __launch_bounds__(512) peer_gather_kernel(const PeerGatherParams p) {
  const int peer = blockIdx.x;
  const uint32_t target = p.seq[peer] + 1u;
  // ---- this rank's slice -> column block `rank` of the peer's buffer ----
  const int pieces_per_row = p.row_bytes >> 4;
  const int total = p.m * pieces_per_row;
  char* dst = p.dst[peer] + (int64_t)p.rank * p.row_bytes;
  for (int i = threadIdx.x; i < total; i += 512) {
    const int r = i / pieces_per_row, c = i - r * pieces_per_row;
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.src + (int64_t)r * p.row_bytes + c * 16);
    *reinterpret_cast<u32x4*>(dst + (int64_t)r * p.dst_pitch + c * 16) = v;
  }
  __threadfence_system();  // every thread's stores are performed at system scope before the flag below
  __syncthreads();
  __shared__ int s_timed_out;
  if (threadIdx.x == 0) {
    __hip_atomic_store(p.flags[peer] + p.rank, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- wait for the peer's slice: its workgroup `rank` raises flags[rank][peer] ----
    uint32_t* mine = p.flags[p.rank] + peer;
    const uint64_t t0 = wall_clock64();
    bool ok = true;
    while ((int32_t)(__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
      if ((int64_t)(wall_clock64() - t0) > p.timeout_ticks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    s_timed_out = ok ? 0 : 1;
    p.seq[peer] = target;
  }
  __syncthreads();
  // A slice that did not arrive is made LOUD: its column block of this rank's own buffer is filled with NaN bit patterns
  // (0x7fff is a NaN in bf16 and in fp16), so a consumer that never looks at *status still cannot use stale numbers.
  if (s_timed_out) {
    char* own = p.dst[p.rank] + (int64_t)peer * p.row_bytes;
    for (int i = threadIdx.x; i < total; i += 512) {
      const int r = i / pieces_per_row, c = i - r * pieces_per_row;
      *reinterpret_cast<u32x4*>(own + (int64_t)r * p.dst_pitch + c * 16) = u32x4{0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};
    }
  }
}

}  // namespace

extern "C" {

int tg_peer_alloc(int device, int64_t bytes, void** ptr) {
  if (!ptr) return TG_E_NULL;
  if (bytes <= 0) return TG_E_SHAPE;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  void* q = nullptr;
  hipError_t e = hipExtMallocWithFlags(&q, (size_t)bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(q, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(q);
    return (int)e;
  }
  *ptr = q;
  return 0;
}

int tg_peer_free(int device, void* ptr) {
  if (!ptr) return TG_E_NULL;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  return (int)hipFree(ptr);
}

int tg_peer_export(int device, void* ptr, tg_peer_handle* out) {
  if (!ptr || !out) return TG_E_NULL;
  static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(tg_peer_handle), "IPC handle does not fit");
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) return (int)e;
  memset(out, 0, sizeof(*out));
  memcpy(out->bytes, &h, sizeof(h));
  return 0;
}

int tg_peer_open(int device, const tg_peer_handle* handle, void** ptr) {
  if (!handle || !ptr) return TG_E_NULL;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  hipIpcMemHandle_t h;
  memcpy(&h, handle->bytes, sizeof(h));
  void* q = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return (int)e;
  *ptr = q;
  return 0;
}

int tg_peer_close(int device, void* ptr) {
  if (!ptr) return TG_E_NULL;
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  return (int)hipIpcCloseMemHandle(ptr);
}

int tg_peer_gather_launch(const tg_peer_gather* a, int device, tg_stream_t stream) {
  if (!a || !a->src || !a->seq || !a->status) return TG_E_NULL;
  if (a->world < 1 || a->world > TG_PEER_MAX_WORLD || a->rank < 0 || a->rank >= a->world) return TG_E_SHAPE;
  if (a->m <= 0 || a->cols_local <= 0 || a->m > INT32_MAX || a->cols_local * 2 > INT32_MAX) return TG_E_SHAPE;
  if ((a->cols_local * 2) % 16 != 0) return TG_E_ALIGN;
  if (a->m * (a->cols_local / 8) > INT32_MAX) return TG_E_SIZE;
  if (!aligned16(a->src)) return TG_E_ALIGN;
  PeerGatherParams p;
  for (int r = 0; r < TG_PEER_MAX_WORLD; ++r) {
    p.dst[r] = nullptr;
    p.flags[r] = nullptr;
  }
  for (int r = 0; r < a->world; ++r) {
    if (!a->dst[r] || !a->flags[r]) return TG_E_NULL;
    if (!aligned16(a->dst[r])) return TG_E_ALIGN;
    p.dst[r] = (char*)a->dst[r];
    p.flags[r] = a->flags[r];
  }
  p.src = (const char*)a->src;
  p.seq = a->seq;
  p.status = a->status;
  p.world = a->world;
  p.rank = a->rank;
  p.m = (int32_t)a->m;
  p.row_bytes = (int32_t)(a->cols_local * 2);
  p.dst_pitch = (int64_t)a->world * p.row_bytes;
  p.timeout_ticks = (a->timeout_us > 0 ? a->timeout_us : 2000000) * 100;  // s_memrealtime counts at 100 MHz
  DeviceScope ds(device);
  if (!ds.ok) return TG_E_DEVICE;
  hipLaunchKernelGGL(peer_gather_kernel, dim3((unsigned)a->world), dim3(512), 0, (hipStream_t)stream, p);
  return launch_status();
}

}  // extern "C"
