"""Row-sharded (tensor-parallel over output features) quantized Linear for one node of MI355X GPUs.

Not in the reference (it has no distributed code): weight rows are independent units of
the tinygemm path (every 16-row tile has its own LUT rows and scale/zero columns), so rank r of G owns
rows [r*n/G, (r+1)*n/G) -- packed codes, lut[n/G,16] and scales_and_zeros[:, r*n/G:(r+1)*n/G, :] (dim 1:
the tensor is [k/g][n][2]) -- the activation is replicated, and ONE all-gather of the [m, n/G] partial
outputs per linear rebuilds y on every rank.  One process per GPU; `torch.distributed` backend "nccl" is
RCCL on ROCm, over xGMI inside a node.  The payload is tiny (m*n/G*2 bytes), i.e. latency-bound.
"""
from __future__ import annotations

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


class RowShardedLinear(torch.nn.Module):
    """Wraps the rank-local quantized Linear (rows [lo, hi) only) and all-gathers its output.

    local        any module mapping [..., k] -> [..., n/G] (Any4Linear / Int4Linear built from the shard)
    out_features full n
    gather_output=False leaves the result sharded (for a following column-parallel consumer).
    """

    def __init__(self, local: torch.nn.Module, out_features: int, group=None, gather_output: bool = True):
        super().__init__()
        self.local = local
        self.out_features = out_features
        self.group = group
        self.gather_output = gather_output

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
