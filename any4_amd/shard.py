"""Row-sharded (tensor-parallel over output features) quantized Linear for one node of MI355X GPUs.

Not in the reference (it has no distributed code): weight rows are independent units of
the tinygemm path (every 16-row tile has its own LUT rows and scale/zero columns), so rank r of G owns
rows [r*n/G, (r+1)*n/G) -- packed codes, lut[n/G,16] and scales_and_zeros[:, r*n/G:(r+1)*n/G, :] (dim 1:
the tensor is [k/g][n][2]) -- the activation is replicated, and ONE all-gather of the [m, n/G] partial
outputs per linear rebuilds y on every rank.  One process per GPU; `torch.distributed` backend "nccl" is
RCCL on ROCm, over xGMI inside a node.  The payload is tiny (m*n/G*2 bytes), i.e. latency-bound.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist


def row_range(n: int, rank: int, world: int, tile: int = 16):
    """Rows owned by `rank`; n must split into whole 16-row tiles per rank."""
    if n % (world * tile) != 0:
        raise ValueError(f"out_features={n} must be a multiple of world_size*{tile}={world * tile} to row-shard")
    per = n // world
    return rank * per, (rank + 1) * per


def shard_any4_params(codes: torch.Tensor, lut: torch.Tensor, scales_and_zeros: torch.Tensor, rank: int, world: int):
    """Slice UNPACKED codes [n][k], lut ([n][16] per-row or [16] global) and scales_and_zeros [k/g][n][2]."""
    lo, hi = row_range(codes.shape[0], rank, world)
    lut_local = lut[lo:hi].contiguous() if lut.dim() == 2 else lut
    return codes[lo:hi].contiguous(), lut_local, scales_and_zeros[:, lo:hi, :].contiguous()


def shard_mx4_params(codes: torch.Tensor, exponents: torch.Tensor, rank: int, world: int):
    lo, hi = row_range(codes.shape[0], rank, world)
    return codes[lo:hi].contiguous(), exponents[lo:hi].contiguous()


class _DeviceBytes:
    """__cuda_array_interface__ holder: lets torch view device memory owned by the C library (no copy, no ownership)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class PeerWriteGather:
    """Host side of include/peer_gather_hip.h: the one-shot gather of the [m, n/G] output slices of a row-sharded Linear.

    Every rank stores its slice straight into every peer's gathered buffer over xGMI and raises a flag there; one kernel
    launch per rank and call, no host synchronisation.  The result has the layout the consumer wants ([m, n], rank-major
    feature order), which RCCL's all_gather_into_tensor ([G, m, n/G]) only gives for m = 1.  One process per GPU; the
    buffers are uncached device memory of the C library, shared as IPC handles through `group` (any backend) once, here.
    Calls alternate between two gathered buffers (see the header for why two are enough); the tensor a call returns is
    overwritten by the call after the next one.
    """

    def __init__(self, m_max: int, cols_local: int, group=None, device=None, dtype=torch.bfloat16, timeout_us: int = 2_000_000,
                 graph_timeout_us: int = 500_000):
        """timeout_us: bound of a call's wait for its peers (eager calls: ranks may be seconds apart after a compile or a load).
        graph_timeout_us: the bound baked into launches recorded under stream capture.  Replays are started by independent host
        processes, so the bound has to cover the SKEW between the ranks' graph launches (a GC pause, sampling or tokenisation on
        one rank), not just the step: half a second by default -- a dead rank then turns a 2 ms token into 0.5 s before poll()
        raises, never into minutes.  A serving loop that keeps its ranks in lock step (a host barrier per step) may pass less;
        the caller must keep the launch skew below this value."""
        from . import _lib

        if dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("PeerWriteGather moves 16-bit outputs")
        if (cols_local * 2) % 16 != 0:
            raise ValueError("cols_local * 2 bytes must be a multiple of 16")
        self._L = _lib
        self.lib = _lib.load()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > _lib.TG_PEER_MAX_WORLD:
            raise ValueError(f"world size {self.world} > {_lib.TG_PEER_MAX_WORLD}")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dev_index = self.device.index
        self.m_max, self.cols_local, self.dtype, self.timeout_us = m_max, cols_local, dtype, timeout_us
        self.graph_timeout_us = min(graph_timeout_us, timeout_us)
        self.buf_bytes = (m_max * cols_local * self.world * 2 + 255) & ~255
        # control block: flags uint32[16] | seq uint32[16] | status uint32
        self.ctl_bytes = 256
        self._own = []
        self._data = self._alloc(2 * self.buf_bytes)
        self._ctl = self._alloc(self.ctl_bytes)
        handles = [None] * self.world
        mine = (bytes(self._export(self._data).bytes), bytes(self._export(self._ctl).bytes))
        dist.all_gather_object(handles, mine, group=group)
        self._opened = []
        self._peer_data, self._peer_ctl = [], []
        failure = None
        try:
            for r, (hd, hc) in enumerate(handles):
                if r == self.rank:
                    self._peer_data.append(self._data)
                    self._peer_ctl.append(self._ctl)
                else:
                    self._peer_data.append(self._open(hd))
                    self._peer_ctl.append(self._open(hc))
        except RuntimeError as e:  # e.g. no IPC / peer access between these two devices
            failure = f"rank {self.rank}: {e}"
        # every rank learns whether every rank mapped every buffer: either all go on or all raise (a rank that raised alone
        # would leave the others waiting in the barrier below)
        failures = [None] * self.world
        dist.all_gather_object(failures, failure, group=group)
        if any(f is not None for f in failures):
            self._release()
            raise RuntimeError("PeerWriteGather: could not map the peers' buffers: " + "; ".join(f for f in failures if f))
        self._calls = 0
        self._args = []
        for parity in range(2):
            a = _lib.PeerGather()
            for r in range(self.world):
                a.dst[r] = self._peer_data[r] + parity * self.buf_bytes
                a.flags[r] = self._peer_ctl[r]
            a.seq = self._ctl + 64
            a.status = self._ctl + 128
            a.world, a.rank, a.cols_local, a.timeout_us = self.world, self.rank, cols_local, timeout_us
            self._args.append(a)
        self._views = [torch.as_tensor(_DeviceBytes(self._data + parity * self.buf_bytes, self.buf_bytes), device=self.device)
                       for parity in range(2)]
        self._status = torch.as_tensor(_DeviceBytes(self._ctl + 128, 4), device=self.device).view(torch.int32)
        # host mirror of *status for poll(): an asynchronous copy + an event, never a device synchronisation
        self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._status_event = None
        dist.barrier(group=group)  # every rank has mapped every buffer before the first store into one

    def _alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._L.check(self.lib.tg_peer_alloc(self.dev_index, nbytes, ctypes.byref(p)), "tg_peer_alloc")
        self._own.append(p.value)
        return p.value

    def _export(self, ptr: int):
        h = self._L.PeerHandle()
        self._L.check(self.lib.tg_peer_export(self.dev_index, ptr, ctypes.byref(h)), "tg_peer_export")
        return h

    def _open(self, raw: bytes) -> int:
        h = self._L.PeerHandle()
        ctypes.memmove(h.bytes, raw, 64)
        p = ctypes.c_void_p()
        self._L.check(self.lib.tg_peer_open(self.dev_index, ctypes.byref(h), ctypes.byref(p)), "tg_peer_open")
        self._opened.append(p.value)
        return p.value

    def gather(self, y_local: torch.Tensor) -> torch.Tensor:
        """y_local [m, cols_local] (contiguous, on this device) -> [m, world * cols_local], on the current stream."""
        if y_local.dim() != 2 or y_local.shape[1] != self.cols_local or y_local.shape[0] > self.m_max:
            raise ValueError(f"expected [m <= {self.m_max}, {self.cols_local}], got {tuple(y_local.shape)}")
        if y_local.dtype != self.dtype or not y_local.is_contiguous() or y_local.device != self.device:
            raise ValueError("y_local must be a contiguous tensor of the gather's dtype on its device")
        m = y_local.shape[0]
        parity = self._calls & 1
        self._calls += 1
        a = self._args[parity]
        a.src, a.m = y_local.data_ptr(), m
        a.timeout_us = self.graph_timeout_us if torch.cuda.is_current_stream_capturing() else self.timeout_us
        self._L.check(self.lib.tg_peer_gather_launch(ctypes.byref(a), self.dev_index, torch.cuda.current_stream(self.device).cuda_stream),
                      "tg_peer_gather_launch")
        return self._views[parity][: m * self.world * self.cols_local * 2].view(self.dtype).view(m, self.world * self.cols_local)

    _TIMEOUT_MSG = ("PeerWriteGather: a peer's slice did not arrive within the timeout of an earlier call (its column block was "
                    "filled with NaNs): a rank is slow, dead, or the ranks disagree on the number of gather calls")

    def check(self) -> None:
        """Synchronises and raises if a peer's slice did not arrive within the timeout of some call."""
        torch.cuda.synchronize(self.device)
        if int(self._status.item()) != 0:
            raise RuntimeError(self._TIMEOUT_MSG)

    def poll(self) -> None:
        """Non-blocking health check for hot loops: raises if the status word read back by an EARLIER poll() shows a timeout,
        then queues the next asynchronous read-back on the current stream.  Never synchronises the device; a timeout is
        reported one or two poll() calls after it happened (the gathered data itself is already NaN by then)."""
        if self._status_event is not None and self._status_event.query():
            if int(self._status_host[0]) != 0:
                raise RuntimeError(self._TIMEOUT_MSG)
            self._status_event = None
        if self._status_event is None:
            self._status_host.copy_(self._status, non_blocking=True)
            self._status_event = torch.cuda.Event()
            self._status_event.record(torch.cuda.current_stream(self.device))

    def _release(self) -> None:
        for p in self._opened:
            self.lib.tg_peer_close(self.dev_index, p)
        for p in self._own:
            self.lib.tg_peer_free(self.dev_index, p)
        self._opened, self._own = [], []

    def close(self) -> None:
        torch.cuda.synchronize(self.device)
        if dist.is_initialized():
            dist.barrier(group=self.group)  # no peer is still storing into this rank's buffers
        self._release()


class RowShardedLinear(torch.nn.Module):
    """Wraps the rank-local quantized Linear (rows [lo, hi) only) and all-gathers its output.

    local        any module mapping [..., k] -> [..., n/G] (Any4Linear / Int4Linear built from the shard)
    out_features full n
    gather_output=False leaves the result sharded (for a following column-parallel consumer).
    gather="rccl" (all_gather_into_tensor) or "peer" (PeerWriteGather: one-shot stores into the peers' buffers, for inputs of
    at most peer_m_max rows; larger inputs take the RCCL collective).

    Output lifetime with gather="peer": the gather lands in a two-buffer ring that peers write into, so by default forward()
    returns a COPY (a fresh tensor, like the RCCL path; the payload is a few KiB).  alias_output=True returns the view into
    the ring instead: it is overwritten by the second following forward() of this layer and by remote stores of peers that
    are one call ahead, so it is only valid for a consumer enqueued on the SAME stream before the next forward() -- the decode
    harness's pattern.  Every `poll_every` forwards the gather's status word is checked without a device synchronisation
    (PeerWriteGather.poll); a peer that never arrived raises there, and its slice is NaN in the meantime.
    """

    def __init__(self, local: torch.nn.Module, out_features: int, group=None, gather_output: bool = True, gather: str = "rccl",
                 peer_m_max: int = 16, alias_output: bool = False, poll_every: int = 64):
        super().__init__()
        if gather not in ("rccl", "peer"):
            raise ValueError("gather must be 'rccl' or 'peer'")
        self.local = local
        self.out_features = out_features
        self.group = group
        self.gather_output = gather_output
        self.gather = gather
        self.peer_m_max = peer_m_max
        self.alias_output = alias_output
        self.poll_every = max(1, int(poll_every))
        self._forwards = 0
        self._peer = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y_local = self.local(x)
        if not self.gather_output or not dist.is_initialized():
            return y_local
        world = dist.get_world_size(self.group)
        if world == 1:
            return y_local
        y_local = y_local.contiguous()
        if y_local.shape[-1] * world != self.out_features:
            raise RuntimeError("local output width * world_size != out_features")
        rows = y_local.numel() // y_local.shape[-1]
        if self.gather == "peer" and y_local.is_cuda and rows <= self.peer_m_max:
            if self._peer is None:
                self._peer = PeerWriteGather(self.peer_m_max, y_local.shape[-1], group=self.group, device=y_local.device,
                                             dtype=y_local.dtype)
            out = self._peer.gather(y_local.view(rows, -1)).view(*y_local.shape[:-1], self.out_features)
            self._forwards += 1
            if self._forwards % self.poll_every == 0 and not torch.cuda.is_current_stream_capturing():
                self._peer.poll()
            return out if self.alias_output else out.clone()
        parts = torch.empty((world,) + tuple(y_local.shape), dtype=y_local.dtype, device=y_local.device)
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(parts, y_local, group=self.group)
        else:  # gloo (CPU tests)
            dist.all_gather(list(parts.unbind(0)), y_local, group=self.group)
        # [G, ..., n/G] -> [..., G*n/G] with rank-major feature order
        return parts.movedim(0, -2).reshape(*y_local.shape[:-1], self.out_features)


def build_row_sharded_any4(codes, lut, scales_and_zeros, bias, group_size, rank, world, device, dtype,
                           kernel="linear_y_f16RM_x_f16RM_W_any4TC", w_inner_k=4, group=None):
    """Construct the rank-local Any4Linear from full (unsharded) tensors and wrap it."""
    from .modules import Any4Linear

    n, k = codes.shape
    c, l, sz = shard_any4_params(codes, lut, scales_and_zeros, rank, world)
    lo, hi = row_range(n, rank, world)
    mod = Any4Linear(k, hi - lo, bias=bias is not None, device=device, dtype=dtype, group_size=group_size,
                     kernel=kernel, w_inner_k=w_inner_k, per_row=lut.dim() == 2)
    mod.weight.data = c.to(device)
    mod.lut.data = l.to(device=device, dtype=dtype)
    mod.scales_and_zeros.data = sz.to(device=device, dtype=dtype)
    if bias is not None:
        mod.bias.data = bias[lo:hi].to(device=device, dtype=dtype)
    mod.reshape_weight(w_inner_k)
    return RowShardedLinear(mod, n, group=group)
