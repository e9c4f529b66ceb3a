/*
 * peer_gather_hip.h -- one-shot peer-write gather of row-sharded GEMM outputs over xGMI.
 *
 * Not in the reference (facebookresearch/any4 has no distributed code): with the weight rows of a quantized Linear
 * partitioned across the G GPUs of one node (any4_amd/shard.py), every rank holds a [m][n/G] slice of y after its
 * tg_gemm_w4 and all ranks need the whole [m][n].  The payload is tiny (m * n/G * 2 bytes: 1-64 KiB), so the exchange is
 * latency-bound; a ring all-gather pays G-1 hops.  Here every rank STORES its slice straight into every peer's gathered
 * buffer (xGMI is point-to-point: one hop to each of the 7 peers, all links used at once), then raises a flag in every
 * peer's flag array; a rank is done when the flags of all peers have reached this call's sequence number.  One kernel
 * launch per rank, no host synchronisation, capturable in a hipGraph (the sequence number lives on the device).
 *
 * Memory: the gathered buffers and the flag arrays are allocated by tg_peer_alloc (uncached, so that a flag raised by a
 * peer is seen by a kernel that is already spinning on it) and shared between the processes of a node as IPC handles
 * (tg_peer_export / tg_peer_open); one process per GPU.  The caller alternates between TWO gathered buffers (call i uses
 * buffer i & 1): a rank can only start call i + 1 after every peer has entered call i, i.e. after every consumer of the
 * buffer of call i - 1 has been ordered before it on the peer's stream.
 *
 * Same conventions as tinygemm_hip.h: extern "C", plain pointers and sizes, 0 = ok, negative TG_E_* = a precondition
 * failed (nothing launched), positive = hipError_t.
 */
#ifndef PEER_GATHER_HIP_H_
#define PEER_GATHER_HIP_H_

#include "tinygemm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TG_PEER_MAX_WORLD 16

typedef struct tg_peer_handle {
  unsigned char bytes[64]; /* a hipIpcMemHandle_t */
} tg_peer_handle;

/* device memory that peers may write while a local kernel polls it (hipExtMallocWithFlags, uncached), zero-filled */
TG_API int tg_peer_alloc(int device, int64_t bytes, void** ptr);
TG_API int tg_peer_free(int device, void* ptr);
/* IPC handle of a tg_peer_alloc allocation (for the other processes of the node) / mapping of a peer's handle */
TG_API int tg_peer_export(int device, void* ptr, tg_peer_handle* out);
TG_API int tg_peer_open(int device, const tg_peer_handle* handle, void** ptr);
TG_API int tg_peer_close(int device, void* ptr);

typedef struct tg_peer_gather {
  const void* src;                  /* this rank's slice, row-major 16-bit [m][cols_local]                         */
  void* dst[TG_PEER_MAX_WORLD];     /* dst[r]: rank r's gathered buffer [m][world * cols_local] as mapped HERE     */
  uint32_t* flags[TG_PEER_MAX_WORLD]; /* flags[r]: rank r's flag array uint32[world] as mapped here                */
  uint32_t* seq;                    /* device word of this rank: calls so far; the kernel increments it            */
  uint32_t* status;                 /* device word of this rank: set to 1 when a peer's flag did not arrive (that
                                       peer's column block of dst[rank] is then filled with NaN bit patterns)      */
  int32_t world, rank;
  int64_t m, cols_local;            /* cols_local * 2 bytes must be a multiple of 16                               */
  int64_t timeout_us;               /* bound of the wait for the peers (<= 0: 2 s)                                 */
} tg_peer_gather;

/* one launch: copy `src` into column block `rank` of every dst[r], release, raise flags[r][rank], wait for flags[rank][*] */
TG_API int tg_peer_gather_launch(const tg_peer_gather* args, int device, tg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PEER_GATHER_HIP_H_ */
