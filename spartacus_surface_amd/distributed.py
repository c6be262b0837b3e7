"""Column sharding over the GPUs of one node.

Columns are independent (radsurf_interface.F90:105), so the path shards with no
data-path collective: every rank owns a contiguous range of columns together
with the matching packed layer range (istartlay / nlay,
radsurf_canopy_properties.F90:56-59), solves it with its own plan on its own
GPU, and the only exchange is the final gather of the results: point-to-point
sends of every output array into its slice of rank 0's full-size array
(DeviceGather); gather_flat_outputs is the simple padded all_gather form the
CPU tests compare it with.

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


# --------------------------------------------------------------------------
# Final gather from device memory (the contract of the
# reference's output objects: radsurf/radsurf_canopy_flux.F90:27-91,
# radsurf/radsurf_boundary_conds_out.F90:24-39; what the driver writes out
# after its solver loop: driver/spartacus_surface_driver.F90:206-234, 250-265)
# --------------------------------------------------------------------------
OUTPUT_OBJECTS = ("sw_norm_dir", "sw_norm_diff", "lw_internal", "lw_norm")


class _DevArray:
    """CUDA-array-interface view of a device array owned by a plan."""

    def __init__(self, ptr, nelem):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


def output_names(plan):
    """Names of every output array the plan holds on the device: the four members of
    bc_out and every ALLOCATED member of the four flux objects, in a fixed order."""
    names = ["bc_out." + n for n in abi.BC_FIELDS]
    for obj in OUTPUT_OBJECTS:
        for n, _ in abi.FLUX_FIELDS:
            names.append("%s.%s" % (obj, n))
    have = []
    for n in names:
        try:
            plan.device_ptr(n)
        except Exception:       # member not allocated in this configuration
            continue
        have.append(n)
    return have


class HostArrays:
    """Stand-in for a plan whose "device" arrays are host arrays: lets the CPU test-suite
    (gloo, no GPU) drive DeviceGather's packing, the single gather and the unpacking.
    arrays: {name: ndarray with rows first}."""

    def __init__(self, arrays):
        self.arrays = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in arrays.items()}

    def device_ptr(self, name):
        a = self.arrays[name]           # KeyError = "member not allocated"
        return a.ctypes.data, int(a.size)

    def host_array(self, name):
        return self.arrays[name]

    def device(self):
        return -1


def _per_column(name):
    return name.startswith("bc_out.") or name.split(".")[-1] in _PER_COLUMN


class DeviceGather:
    """Gather of bc_out and the four flux objects of every rank's plan to rank
    ``root``, from device memory to device memory, WITHOUT intermediate buffers:
    per-column and per-layer ranges of a rank are contiguous in the full arrays
    (radsurf_canopy_properties.F90:56-59), so every peer sends each of its output
    arrays as it lies in its plan (views of ``ss_plan_device_ptr``) and the root
    receives it straight into the slice of the full-size array it belongs to - one
    batch of point-to-point operations (``batch_isend_irecv``: grouped send/recv in
    RCCL, each peer pushes over its own xGMI link into the root).  No padding to the
    largest shard, no packing pass on the senders, no unpacking pass on the root: the
    root holds the outputs once.  There is no reduction and no all-to-all: the ring
    bound of an all-reduce does not apply.

    names=None gathers everything; a subset (e.g. the four bc_out arrays that a
    host atmosphere model needs every step) gets its own instance.

    stage_host=True is for rehearsals of the N > 1 path where no RCCL is possible
    (several ranks on one GPU over ``gloo``): buffers go through host memory.
    """

    def __init__(self, plan, dist, ncol_local, ntotlay_local, names=None, root=0, stage_host=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.plan = plan
        self.root = root
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.names = list(names) if names is not None else output_names(plan)
        self.on_host = hasattr(plan, "host_array")            # HostArrays stand-in (CPU tests)
        self.stage_host = stage_host or self.on_host
        self.device = torch.device("cpu") if self.on_host else torch.device("cuda", plan.device())
        cdev = torch.device("cpu") if self.stage_host else self.device
        self.rows_local = (ncol_local, ntotlay_local)
        self._ptrs = None
        widths = self._make_views()
        # every rank learns every rank's (ncol, ntotlay) and the member widths (identical on ranks with data)
        meta = torch.tensor([ncol_local, ntotlay_local] + widths, dtype=torch.int64, device=cdev)
        allmeta = [torch.zeros_like(meta) for _ in range(self.world)]
        dist.all_gather(allmeta, meta)
        allmeta = [m.cpu().numpy() for m in allmeta]
        self.ncols = [int(m[0]) for m in allmeta]
        self.nlays = [int(m[1]) for m in allmeta]
        self.widths = [int(max(m[2 + i] for m in allmeta)) for i in range(len(self.names))]
        self.full = None
        if self.rank == root:
            self.full = {}
            tc, tl = sum(self.ncols), sum(self.nlays)
            for n, w in zip(self.names, self.widths):
                rows = tc if _per_column(n) else tl
                self.full[n] = torch.zeros(rows * w, dtype=torch.float64, device=cdev)
        self.bytes_per_rank = 8 * sum((ncol_local if _per_column(n) else ntotlay_local) * w
                                      for n, w in zip(self.names, self.widths))

    def _make_views(self):
        """Views of the plan's output arrays.  A plan may re-allocate them (its arrays grow when it is
        re-targeted): the addresses are re-queried at every call and the views rebuilt when they moved."""
        torch = self.torch
        ptrs = [self.plan.device_ptr(n) for n in self.names]
        if ptrs == self._ptrs:
            return None
        self._ptrs = ptrs
        self.views, widths = [], []
        for n, (ptr, nelem) in zip(self.names, ptrs):
            rows = self.rows_local[0] if _per_column(n) else self.rows_local[1]
            # logical size: rows x width (an array of an empty shard still has one element allocated)
            width = (nelem // rows) if rows > 0 else 0
            widths.append(width)
            used = rows * width
            if self.on_host:
                self.views.append(torch.from_numpy(self.plan.host_array(n).reshape(-1))[:used])
            else:
                self.views.append(torch.as_tensor(_DevArray(ptr, max(1, nelem)), device=self.device)[:used])
        return widths

    def __call__(self):
        """One batch of sends (peers) / receives into place (root).  The plan's streams must be idle
        (plan.synchronize())."""
        torch, dist = self.torch, self.dist
        self._make_views()
        if not self.stage_host:
            torch.cuda.synchronize(self.device)
        ops, keep = [], []
        if self.rank == self.root:
            c0 = l0 = 0
            for r in range(self.world):
                for i, n in enumerate(self.names):
                    rows = self.ncols[r] if _per_column(n) else self.nlays[r]
                    cnt = rows * self.widths[i]
                    if cnt == 0:
                        continue
                    start = (c0 if _per_column(n) else l0) * self.widths[i]
                    dst = self.full[n][start:start + cnt]
                    if r == self.rank:
                        dst.copy_(self.views[i])
                    else:
                        ops.append(dist.P2POp(dist.irecv, dst, r))
                c0 += self.ncols[r]
                l0 += self.nlays[r]
        else:
            for v in self.views:
                if v.numel():
                    t = v.cpu() if (self.stage_host and not self.on_host) else v
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, self.root))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        if not self.stage_host:
            torch.cuda.synchronize(self.device)
        return self.full if self.rank == self.root else None

    def to_host(self):
        """{name: ndarray (rows, width)} of the assembled arrays (root only)."""
        out = {}
        kinds = dict(abi.FLUX_FIELDS)
        for n, w in zip(self.names, self.widths):
            a = self.full[n].cpu().numpy()
            two_d = n.startswith("bc_out.") or kinds[n.split(".")[-1]] in ("sc", "sl")
            out[n] = a.reshape(-1, max(1, w)) if two_d else a
        return out
