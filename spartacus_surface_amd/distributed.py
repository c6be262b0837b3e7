"""Column sharding over the GPUs of one node.

Columns are independent (radsurf_interface.F90:105), so the path shards with no
data-path collective: every rank owns a contiguous range of columns together
with the matching packed layer range (istartlay / nlay,
radsurf_canopy_properties.F90:56-59), solves it with its own plan on its own
GPU, and the only exchange is the final gather of the results.  RCCL (like
NCCL) has no gatherv, so shards are padded to the largest shard and exchanged
with one all_gather per array; rank 0 strips the padding.

The functions take the process group's tensors as they come: CPU tensors with
the ``gloo`` backend (unit tests), device tensors with ``nccl`` (= RCCL).
"""
import copy

import numpy as np

from . import abi


def shard_ranges(nlay, world, cost_per_layer=1.0, cost_per_column=0.0):
    """Split columns [1..ncol] into ``world`` contiguous 1-based inclusive
    ranges balanced by sum(nlay * cost_per_layer + cost_per_column).  Empty
    ranges are (i, i-1)."""
    nlay = np.asarray(nlay, dtype=np.int64)
    ncol = int(nlay.size)
    w = nlay * float(cost_per_layer) + float(cost_per_column)
    if w.sum() <= 0:
        w = np.ones(ncol)
    cum = np.concatenate(([0.0], np.cumsum(w)))
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        j = int(np.searchsorted(cum, target, side="left"))
        j = min(max(j, bounds[-1]), ncol)
        bounds.append(j)
    bounds.append(ncol)
    return [(bounds[r] + 1, bounds[r + 1]) for r in range(world)]


def extract_shard(case, i0, i1):
    """Columns i0..i1 (1-based inclusive) of a case as a case of its own."""
    from .workloads import _finish
    cfg, cp, sw, lw = case["config"], case["canopy"], case["sw"], case["lw"]
    ncol = max(0, i1 - i0 + 1)
    c = slice(i0 - 1, i1)
    if ncol > 0:
        l0 = int(cp.istartlay[i0 - 1] - 1)
        l1 = int(cp.istartlay[i1 - 1] - 1 + cp.nlay[i1 - 1])
    else:
        l0 = l1 = 0
    lay = slice(l0, l1)
    cp2, sw2, lw2 = copy.copy(cp), copy.copy(sw), copy.copy(lw)
    cp2.set_layers(np.array(cp.nlay[c], dtype=np.int32))
    for n in abi.CANOPY_REAL_FIELDS + ("i_representation",):
        a = getattr(cp, n)
        if a is not None:
            setattr(cp2, n, a[c].copy() if a.shape[0] == cp.ncol else a[lay].copy())
    for obj, obj2, names in ((sw, sw2, abi.SW_FIELDS), (lw, lw2, abi.LW_FIELDS)):
        for n in names:
            a = getattr(obj, n)
            if a is not None:
                setattr(obj2, n, a[c].copy() if a.shape[0] == cp.ncol else a[lay].copy())
    extra = {k: v for k, v in case.items() if k not in ("config", "canopy", "sw", "lw")}
    return _finish(cfg, cp2, sw2, lw2, **extra), (l0, l1)


def gather_flat_outputs(local, ranges, layer_ranges, ncol, ntotlay, dist, device=None):
    """Gather {name: array} results of every rank's shard to full-size arrays.

    local        : this rank's {name: ndarray} (first dim = shard columns or layers)
    ranges       : [(i0, i1)] 1-based inclusive column range per rank
    layer_ranges : [(l0, l1)] 0-based half-open layer range per rank
    Returns the assembled dict on every rank (all_gather semantics)."""
    import torch
    world = dist.get_world_size()
    out = {}
    for name in sorted(local):
        a = np.ascontiguousarray(local[name], dtype=np.float64)
        leaf = name.split(".")[-1]
        per_col = name.startswith("bc_out.") or leaf in _PER_COLUMN
        sizes = [(r[1] - r[0] + 1) if per_col else (lr[1] - lr[0])
                 for r, lr in zip(ranges, layer_ranges)]
        width = int(np.prod(a.shape[1:])) if a.ndim > 1 else 1
        pad = max(sizes) * width
        buf = torch.zeros(pad, dtype=torch.float64, device=device)
        if a.size:
            buf[:a.size] = torch.from_numpy(a.reshape(-1)).to(buf.device)
        gathered = torch.empty(world * pad, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(gathered, buf)
        g = gathered.cpu().numpy().reshape(world, pad)
        full = np.zeros(((ncol if per_col else ntotlay),) + a.shape[1:])
        for r in range(world):
            n = sizes[r]
            if n == 0:
                continue
            start = (ranges[r][0] - 1) if per_col else layer_ranges[r][0]
            full[start:start + n] = g[r, :n * width].reshape((n,) + a.shape[1:])
        out[name] = full
    return out


_PER_COLUMN = {n for n, k in abi.FLUX_FIELDS if k in ("sc", "c")} | set(abi.BC_FIELDS)
_PER_LAYER = {n for n, k in abi.FLUX_FIELDS if k in ("sl", "l")}
