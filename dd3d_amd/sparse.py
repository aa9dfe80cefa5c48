"""Sparse tensor plumbing for the dd3d hot path on MI355X: TensorField / SparseTensor / CoordinateManager.

These mirror the small part of MinkowskiEngine's object model that the reference touches
(models/MEConv.py:244-267: ``ME.TensorField(coordinates=, features=, device=)``, ``.C``, ``.F``, ``.shape``, ``.sparse()``,
``SparseTensor.C``, ``SparseTensor.dense(shape=, min_coordinate=)``).  All coordinate work runs in libdd3d_hip.so; the manager
builds a whole pyramid of maps (quantise -> stride 2 -> stride 4, kernel maps, pooling maps, per-cloud row ranges) with ONE
device->host read of the row counts.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

_I32 = torch.int32


class _Level:
    __slots__ = ("stride", "C", "table", "cap", "slot", "bound", "inv", "uidx", "seg_off", "M")

    def __init__(self):
        self.M = None
        self.seg_off = None


class CoordinateManager:
    """Coordinate maps of one batch.  ``coords`` float32 [N,4] = (batch index, c1, c2, c3); rows grouped by ascending batch
    index (what ``sparse_collate`` produces).  ``num_clouds``: batch size if the caller knows it (saves a device sync)."""

    MAX_LEVELS = 8

    def __init__(self, coords: torch.Tensor, num_clouds: Optional[int] = None, n_dev: Optional[torch.Tensor] = None):
        """``n_dev`` (int32[1] on the device, optional): only the first ``*n_dev`` rows of ``coords`` are points; ``coords.shape[0]``
        is then a bound.  With it (and without ever calling ``sync``) every map is built and consumed with device-side counts only,
        which is what lets a whole step be captured into a HIP graph."""
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coordinates must be [N,4] = (batch, c1, c2, c3)")
        coords = coords.detach().to(torch.float32).contiguous()
        if not coords.is_cuda:
            raise RuntimeError("dd3d_amd has no CPU path: coordinates must live on the GPU")
        self.device = coords.device
        self.N = int(coords.shape[0])
        self.coords = coords
        self._B = num_clouds
        self.n_dev = n_dev
        self.counts = torch.zeros(self.MAX_LEVELS, dtype=_I32, device=self.device)
        self.err = torch.zeros(1, dtype=_I32, device=self.device)
        self.levels: Dict[int, _Level] = {}
        self.kmaps: Dict[Tuple[int, int, int], Tuple[torch.Tensor, int]] = {}
        self.vox_off = self.vox_pts = None
        self._synced = False
        self._quantize()

    # ------------------------------------------------------------------ construction
    def _new_level(self, stride: int, bound: int) -> _Level:
        lv = _Level()
        lv.stride = stride
        lv.bound = bound
        lv.slot = len(self.levels)
        if lv.slot >= self.MAX_LEVELS:
            raise RuntimeError("too many coordinate levels")
        lv.cap = int(L.lib.dd3d_table_capacity(bound))
        dev = self.device
        lv.C = torch.empty((max(bound, 1), 4), dtype=_I32, device=dev)
        lv.table = torch.empty(lv.cap, dtype=_I32, device=dev)
        lv.inv = torch.empty(max(bound, 1), dtype=_I32, device=dev)
        lv.uidx = torch.empty(max(bound, 1), dtype=_I32, device=dev)
        self.levels[stride] = lv
        return lv

    def _unique(self, q: torch.Tensor, n_dev, bound: int, lv: _Level):
        nbytes = int(L.lib.dd3d_unique_ws_bytes(bound, lv.cap))
        ws = L.workspace(nbytes, self.device)
        L.call("dd3d_unique_rows", L.ptr(q), n_dev, bound, L.ptr(lv.table), lv.cap, L.ptr(lv.C), L.ptr(lv.inv), L.ptr(lv.uidx),
               self.counts[lv.slot:].data_ptr(), L.ptr(ws), ws.numel(), L.stream())

    def _quantize(self):
        N = self.N
        lv = self._new_level(1, N)
        q = torch.empty((max(N, 1), 4), dtype=_I32, device=self.device)
        L.call("dd3d_floor_coords", L.ptr(self.coords), N, L.ptr(q), L.stream())
        n_ptr = None if self.n_dev is None else self.n_dev.data_ptr()
        self._unique(q, n_ptr, N, lv)
        # voxel -> points CSR for the feature mean
        self.vox_off = torch.empty(N + 1, dtype=_I32, device=self.device)
        self.vox_pts = torch.empty(max(N, 1), dtype=_I32, device=self.device)
        ws = L.workspace(int(L.lib.dd3d_csr_ws_bytes(N)), self.device)
        L.call("dd3d_voxel_csr", L.ptr(lv.inv), n_ptr, N, self.counts[lv.slot:].data_ptr(), L.ptr(self.vox_off), L.ptr(self.vox_pts),
               L.ptr(ws), ws.numel(), L.stream())

    def _count_ptr(self, lv: _Level):
        return None if lv.M is not None else self.counts[lv.slot:].data_ptr()

    def count_ptr(self, stride: int) -> int:
        """Device address of the row count of the level at ``stride`` (always valid, synced or not)."""
        return self.counts[self.ensure_level(stride).slot:].data_ptr()

    def rows_bound(self, stride: int) -> int:
        return self._rows_bound(self.ensure_level(stride))

    def check(self):
        """Raise if a map kernel flagged an error (one device->host read; ``sync`` does the same)."""
        self._raise_on(int(self.err.item()))

    @staticmethod
    def _raise_on(err: int):
        if err & 1:
            raise RuntimeError("coordinates are not grouped by ascending batch index (or a batch index is out of range)")
        if err & 2:
            raise IndexError("dense(): a voxel falls outside the requested dense shape")
        if err & 16:
            raise RuntimeError("a cloud has more voxel rows than the cloud-window kernel gradient holds")

    def _rows_bound(self, lv: _Level) -> int:
        return lv.M if lv.M is not None else lv.bound

    def ensure_level(self, stride: int) -> _Level:
        """Coordinates at tensor stride ``stride`` (a power of two): unique(floor(c/stride)*stride), first-occurrence order."""
        if stride in self.levels:
            return self.levels[stride]
        if stride < 2 or stride % 2:
            raise ValueError("tensor stride must be 1, 2, 4, ...")
        src = self.ensure_level(stride // 2)
        bound = self._rows_bound(src)
        lv = self._new_level(stride, bound)
        q = torch.empty((max(bound, 1), 4), dtype=_I32, device=self.device)
        L.call("dd3d_stride_coords", L.ptr(src.C), self._count_ptr(src), bound, stride, L.ptr(q), L.stream())
        self._unique(q, self._count_ptr(src), bound, lv)
        self._synced = False
        return lv

    def kernel_map(self, in_stride: int, out_stride: int, kernel_size: int) -> Tuple[torch.Tensor, int]:
        """nbr[K, ld] int32: input row at C_out[o] + offset_k (SURVEY.md section 8a canonical offset order), or -1."""
        key = (in_stride, out_stride, kernel_size)
        if key not in self.kmaps:
            li, lo = self.ensure_level(in_stride), self.ensure_level(out_stride)
            ld = max(self._rows_bound(lo), 1)
            K = kernel_size ** 3
            nbr = torch.empty((K, ld), dtype=_I32, device=self.device)
            L.call("dd3d_kernel_map", L.ptr(lo.C), self._count_ptr(lo), self._rows_bound(lo), L.ptr(li.table), li.cap, L.ptr(li.C),
                   kernel_size, in_stride, L.ptr(nbr), ld, L.stream())
            self.kmaps[key] = (nbr, ld)
        return self.kmaps[key]

    def parent(self, out_stride: int) -> torch.Tensor:
        """Output row (at ``out_stride``) of every row of the next finer level."""
        return self.ensure_level(out_stride).inv

    def seg_off(self, stride: int) -> torch.Tensor:
        lv = self.ensure_level(stride)
        if lv.seg_off is None:
            B = self.B
            lv.seg_off = torch.empty(B + 1, dtype=_I32, device=self.device)
            L.call("dd3d_segment_offsets", L.ptr(lv.C), self._count_ptr(lv), self._rows_bound(lv), B, L.ptr(lv.seg_off), L.ptr(self.err), L.stream())
            self._synced = False
        return lv.seg_off

    # ------------------------------------------------------------------ host-visible sizes
    @property
    def B(self) -> int:
        if self._B is None:
            self._B = int(self.coords[:, 0].max().item()) + 1 if self.N else 0
        return self._B

    def sync(self):
        """ONE device->host read: row counts of all levels + the error word."""
        if self._synced:
            return
        host = torch.cat([self.counts, self.err]).tolist()
        for lv in self.levels.values():
            lv.M = int(host[lv.slot])
        self._synced = True
        self._raise_on(host[-1])

    def num_rows(self, stride: int) -> int:
        lv = self.ensure_level(stride)
        if lv.M is None:
            self.sync()
        return lv.M

    def coordinates(self, stride: int) -> torch.Tensor:
        return self.ensure_level(stride).C[: self.num_rows(stride)]

    def pairs(self, in_stride: int, out_stride: int, kernel_size: int):
        """Canonical pair lists (in_rows, out_rows, offs[K+1]) of a kernel map: grouped by offset, sorted by out_row."""
        nbr, ld = self.kernel_map(in_stride, out_stride, kernel_size)
        K = kernel_size ** 3
        M = self.num_rows(out_stride)
        in_rows = torch.empty(max(K * M, 1), dtype=_I32, device=self.device)
        out_rows = torch.empty(max(K * M, 1), dtype=_I32, device=self.device)
        offs = torch.empty(K + 1, dtype=_I32, device=self.device)
        ws = L.workspace(int(L.lib.dd3d_pairs_ws_bytes(M, K)), self.device)
        L.call("dd3d_kernel_map_pairs", L.ptr(nbr), ld, None, M, K, L.ptr(in_rows), L.ptr(out_rows), L.ptr(offs), L.ptr(ws), ws.numel(), L.stream())
        P = int(offs[-1].item())
        return in_rows[:P], out_rows[:P], offs


class CloudCoordinateManager(CoordinateManager):
    """The coordinate pyramid of MEConvExp (tensor strides 1, 2, 4; both k=3 kernel maps; both stride-2 pooling maps; per-cloud row
    ranges; optionally the dense() placement of stride 4) from ``dd3d_cloud_maps``: three launches, one workgroup per cloud, all
    hashing in LDS.  Needs the first point of every cloud (``pt_off`` int32[B+1] on the device; ``pt_off[B]`` = number of points, the
    rest of ``coords`` is ignored) and at most ``dd3d_cloud_maps_max_points()`` points per cloud.  Everything it produces is
    bit-identical to what ``CoordinateManager`` builds call by call; other strides / kernel sizes are not available here."""

    def __init__(self, coords: torch.Tensor, pt_off: torch.Tensor, num_clouds: int, max_points: int, dense_extent=None,
                 cloud_group: Optional[torch.Tensor] = None, num_groups: int = 1, group_cloud_off: Optional[torch.Tensor] = None,
                 out: Optional[dict] = None, gmin_floor: Optional[torch.Tensor] = None, flags: Optional[torch.Tensor] = None,
                 feats3: Optional[torch.Tensor] = None, F0_out: Optional[torch.Tensor] = None):
        """``feats3`` (optional, float32 [N, 3]): the points' features; the builder then also averages them per stride-1 voxel
        (``self.F0`` [N, 3], rows beyond the voxel count unwritten; ``F0_out``: a caller-owned buffer for it) -- dd3d_voxel_mean_fwd's bits
        without its launch.
        ``flags`` (optional, int32 [MAX_LEVELS + 1]): caller-owned count / error words that the caller has ALREADY cleared on this
        stream (dd3d_assemble_batch does it on the way): no fill launch here.
        ``gmin_floor`` (optional, int32 [num_groups, 3]): the minimum stride-4 corner of the clouds of every group that are NOT in this
        batch (0x7F7F7F7F = none): dense() then places by the minimum over the whole group, as if all its clouds were here.
        ``out`` (optional): caller-owned buffers for the outputs that outlive the step in the deferred-loss mode of the fused step --
        ``nbr1`` / ``nbr2`` int32 [27, >= N] and ``gro`` int32 [3, num_groups + 1]; their leading dimension is the table stride."""
        if coords.dim() != 2 or coords.shape[1] != 4:
            raise ValueError("coordinates must be [N,4] = (batch, c1, c2, c3)")
        coords = coords.detach().to(torch.float32).contiguous()
        if not coords.is_cuda:
            raise RuntimeError("dd3d_amd has no CPU path: coordinates must live on the GPU")
        if max_points > int(L.lib.dd3d_cloud_maps_max_points()):
            raise ValueError("clouds too large for the LDS map builder; use CoordinateManager")
        dev = self.device = coords.device
        self.N = int(coords.shape[0])
        self.coords, self._B, self.n_dev = coords, int(num_clouds), pt_off[num_clouds:]
        if flags is None:
            flags = torch.zeros(self.MAX_LEVELS + 1, dtype=_I32, device=dev)   # one fill: row counts + the error word
        elif flags.dtype != _I32 or flags.numel() != self.MAX_LEVELS + 1 or flags.device != dev:
            raise ValueError("flags must be int32 [MAX_LEVELS + 1] on the coordinates' device")
        self.counts, self.err = flags[:self.MAX_LEVELS], flags[self.MAX_LEVELS:]
        self.levels, self.kmaps, self._synced = {}, {}, False
        B, bound = self._B, max(self.N, 1)
        i32 = lambda *shape: torch.empty(shape, dtype=_I32, device=dev)
        C = [i32(bound, 4) for _ in range(3)]
        inv1, uidx1, par2, uidx2, par4, uidx4 = (i32(bound) for _ in range(6))
        self.vox_off, self.vox_pts = i32(bound + 1), i32(bound)
        out = out or {}
        nbr1 = out["nbr1"] if "nbr1" in out else i32(27, bound)
        nbr2 = out["nbr2"] if "nbr2" in out else i32(27, bound)
        if nbr1.shape[1] != bound or nbr2.shape[1] != bound or not (nbr1.is_contiguous() and nbr2.is_contiguous()):
            raise ValueError("caller-owned neighbour tables must be contiguous [27, N]")
        kid2, kid4 = i32(8, bound), i32(8, bound)
        seg = i32(3, B + 1)
        self.dense_tables = None
        E, gmin, site, vox_at = (0, 0, 0), None, None, None
        if dense_extent is not None:
            E = tuple(int(e) for e in dense_extent)
            gmin, site, vox_at = i32(num_groups, 3), i32(bound), i32(E[0] * E[1] * E[2], max(B, 1))
            self.dense_tables = (E, cloud_group, gmin, site, vox_at)
        self.group_row_off = (out["gro"] if "gro" in out else i32(3, num_groups + 1)) if group_cloud_off is not None else None   # rows of every group at strides 1, 2, 4
        ws = L.workspace(int(L.lib.dd3d_cloud_maps_ws_bytes(B)), dev)
        self.F0 = None
        extra = ()
        if feats3 is not None:
            if feats3.dtype != torch.float32 or tuple(feats3.shape) != (self.N, 3) or not feats3.is_contiguous() or feats3.device != dev:
                raise ValueError("feats3 must be contiguous float32 [N, 3] on the coordinates' device")
            self.F0 = F0_out[:bound] if F0_out is not None else torch.empty((bound, 3), dtype=torch.float32, device=dev)
            extra = (L.ptr(feats3), L.ptr(self.F0))
        L.call("dd3d_cloud_maps_f0" if extra else "dd3d_cloud_maps", L.ptr(coords), L.ptr(pt_off), B, int(max_points), self.N, L.ptr(cloud_group), int(num_groups), E[0], E[1], E[2],
               L.ptr(self.counts), L.ptr(seg), L.ptr(C[0]), L.ptr(C[1]), L.ptr(C[2]), L.ptr(inv1), L.ptr(uidx1), L.ptr(self.vox_off),
               L.ptr(self.vox_pts), L.ptr(par2), L.ptr(uidx2), L.ptr(par4), L.ptr(uidx4), L.ptr(nbr1), L.ptr(nbr2), L.ptr(kid2), L.ptr(kid4), bound,
               L.ptr(gmin), L.ptr(site), L.ptr(vox_at), L.ptr(group_cloud_off), L.ptr(self.group_row_off), L.ptr(gmin_floor), L.ptr(self.err), L.ptr(ws),
               ws.numel(), L.stream(), *extra)
        for slot, (stride, inv, uidx) in enumerate(((1, inv1, uidx1), (2, par2, uidx2), (4, par4, uidx4))):
            lv = _Level()
            lv.stride, lv.bound, lv.slot, lv.cap, lv.table = stride, self.N, slot, 0, None
            lv.C, lv.inv, lv.uidx, lv.seg_off = C[slot], inv, uidx, seg[slot]
            self.levels[stride] = lv
        self.kmaps = {(1, 1, 3): (nbr1, bound), (1, 2, 2): (kid2, bound), (2, 2, 3): (nbr2, bound), (2, 4, 2): (kid4, bound)}

    def ensure_level(self, stride: int) -> _Level:
        if stride not in self.levels:
            raise ValueError("CloudCoordinateManager holds tensor strides 1, 2, 4 only")
        return self.levels[stride]

    def kernel_map(self, in_stride: int, out_stride: int, kernel_size: int):
        key = (in_stride, out_stride, kernel_size)
        if key not in self.kmaps:
            raise ValueError("CloudCoordinateManager holds the k=3 maps at strides 1, 2 and the k=2 stride-2 pooling maps only")
        return self.kmaps[key]

    def seg_off(self, stride: int) -> torch.Tensor:
        return self.ensure_level(stride).seg_off


class SparseTensor:
    """Features ``F`` [M,C] on the voxels of ``coordinate_manager`` at ``tensor_stride``."""

    def __init__(self, features: torch.Tensor, coordinate_manager: CoordinateManager, tensor_stride: int = 1):
        self.F = features
        self.coordinate_manager = coordinate_manager
        self.tensor_stride = int(tensor_stride)

    @property
    def C(self) -> torch.Tensor:
        return self.coordinate_manager.coordinates(self.tensor_stride)

    @property
    def shape(self):
        return self.F.shape

    @property
    def device(self):
        return self.F.device

    def dense(self, shape=None, min_coordinate=None):
        """``SparseTensor.dense`` as the reference calls it (models/MEConv.py:254,203): returns (dense, min_coordinate, stride).
        Plain torch indexing on the GPU (differentiable), used by the ``--pytorch`` model; MEConvExp never builds this tensor."""
        ts = self.tensor_stride
        C = self.C.to(torch.int64)
        if min_coordinate is None:
            min_coordinate = C.min(0, keepdim=True)[0][:, 1:]
        minc = min_coordinate.reshape(-1).to(device=C.device, dtype=torch.int64)
        if bool((minc % ts != 0).any()):
            raise ValueError("min_coordinate must be a multiple of the tensor stride")
        idx = (C[:, 1:] - minc[None, :]) // ts
        if shape is None:
            ext = (idx.max(0)[0] + 1).tolist()
            shape = [int(C[:, 0].max().item()) + 1, self.F.shape[1]] + ext
        shape = list(shape)
        if shape[1] != self.F.shape[1]:
            shape[1] = self.F.shape[1]
        dense = torch.zeros(shape, dtype=self.F.dtype, device=self.F.device)
        dense[C[:, 0], :, idx[:, 0], idx[:, 1], idx[:, 2]] = self.F
        return dense, min_coordinate, torch.tensor([ts] * 3)


class TensorField:
    """``ME.TensorField(coordinates=, features=, device=)`` (models/MEConv.py:263-267): float coordinates with the batch index in
    column 0; ``sparse()`` quantises to voxels (floor) and averages the features of the points of each voxel."""

    def __init__(self, features=None, coordinates=None, device=None, num_clouds: Optional[int] = None, **_unused):
        if features is None or coordinates is None:
            raise ValueError("TensorField needs coordinates and features")
        dev = torch.device(device) if device is not None else coordinates.device
        self.C = coordinates.to(dev)
        self.F = features.to(dev)
        self.num_clouds = num_clouds
        self._manager: Optional[CoordinateManager] = None

    @property
    def shape(self):
        return self.F.shape

    @property
    def device(self):
        return self.F.device

    @property
    def coordinate_manager(self) -> CoordinateManager:
        if self._manager is None:
            self._manager = CoordinateManager(self.C, self.num_clouds)
        return self._manager

    def sparse(self) -> SparseTensor:
        from .ops import voxel_mean
        cm = self.coordinate_manager
        return SparseTensor(voxel_mean(self.F, cm), cm, 1)
