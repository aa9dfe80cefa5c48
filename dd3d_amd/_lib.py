"""ctypes binding of libdd3d_hip.so (include/dd3d.h).  The product path has NO fallback: if the library is missing the
import fails loudly, and calling a kernel without a GPU tensor raises."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DD3D_LIB: another build of the same library (tools/ab.sh comparison builds, the `make prof` build) without touching the product .so
LIB_PATH = os.environ.get("DD3D_LIB") or os.path.join(_HERE, "libdd3d_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C dd3d_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "dd3d_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH)

P, I64, I32, F32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class GSeg(C.Structure):
    """struct dd3d_gseg (include/dd3d.h)."""
    _fields_ = [("syn", P), ("real", P), ("dsyn", P), ("rows", I64), ("cols", I64), ("colmajor", I32), ("pad_", I32)]


_SIGS = {
    "dd3d_version": (I32, []),
    "dd3d_table_capacity": (I64, [I64]),
    "dd3d_unique_ws_bytes": (I64, [I64, I64]),
    "dd3d_floor_coords": (I32, [P, I64, P, P]),
    "dd3d_stride_coords": (I32, [P, P, I64, I32, P, P]),
    "dd3d_unique_rows": (I32, [P, P, I64, P, I64, P, P, P, P, P, I64, P]),
    "dd3d_quantize_unique": (I32, [P, I64, P, P, I64, P, P, P, P, P, I64, P]),
    "dd3d_csr_ws_bytes": (I64, [I64]),
    "dd3d_voxel_csr": (I32, [P, P, I64, P, P, P, P, I64, P]),
    "dd3d_kernel_map": (I32, [P, P, I64, P, I64, P, I32, I32, P, I64, P]),
    "dd3d_pairs_ws_bytes": (I64, [I64, I32]),
    "dd3d_kernel_map_pairs": (I32, [P, I64, P, I64, I32, P, P, P, P, I64, P]),
    "dd3d_segment_offsets": (I32, [P, P, I64, I32, P, P, P]),
    "dd3d_dense_sites": (I32, [P, P, I64, P, I32, I32, I32, I32, I32, I32, P, P, P, P, P]),
    "dd3d_gather_clouds": (I32, [P, P, I32, P, I32, I32, I64, I32, I64, F32, P, P, P, P, P, P]),
    "dd3d_assemble_batch": (I32, [P, I64, I32, P, I32, F32, F32, I32, F32, P, P, I32, P, I32, I32, I64, I32, I64, P, P, P, P, P, P, I32, P]),
    "dd3d_cloud_maps_max_points": (I32, []),
    "dd3d_cloud_maps_ws_bytes": (I64, [I32]),
    "dd3d_cloud_maps": (I32, [P, P, I32, I32, I64, P, I32, I32, I32, I32] + [P] * 17 + [I64, P, P, P, P, P, P, P, P, I64, P]),
    "dd3d_cloud_maps_f0": (I32, [P, P, I32, I32, I64, P, I32, I32, I32, I32] + [P] * 17 + [I64, P, P, P, P, P, P, P, P, I64, P, P, P]),
    "dd3d_conv_fwd": (I32, [P, I64, I32, P, I32, P, I64, I32, I64, P, I32, P, P]),
    "dd3d_syn_head": (I32, [P, I64, I32, P, I32, F32, F32, I32, F32, P, P, P]),
    "dd3d_sgd_scatter": (I32, [P, P, P, P, I64, I32, F32, F32, P, P]),
    "dd3d_populate_img": (I32, [P, I64, I64, I32, I32, P, P, P]),
    "dd3d_conv_schedule_tile_rows": (I32, [I64]),
    "dd3d_conv_schedule_blocks": (I64, [I64]),
    "dd3d_conv_schedule_ws_bytes": (I64, [I64]),
    "dd3d_conv_schedule": (I32, [P, I64, I32, I64, P, P, P, P, P, P, I64, P]),
    "dd3d_conv128_ordered": (I32, [P, I64, P, P, I64, I32, I64, P, I32, P, P, P, P, P, P]),
    "dd3d_conv128_split_ws_bytes": (I64, [I64]),
    "dd3d_conv128_split": (I32, [P, I64, P, P, I64, I32, I64, P, I32, P, P, I64, P]),
    "dd3d_conv_bwd_data": (I32, [P, I64, P, I32, P, I32, P, I64, I32, P, P]),
    "dd3d_conv_bwd_weight_ws_bytes": (I64, [I64, I32, I32, I32, I32]),
    "dd3d_conv_bwd_weight": (I32, [P, I32, P, I32, P, I64, I32, I64, P, I32, P, P, I64, P]),
    "dd3d_conv128_bwd_weight_clouds_max_rows": (I32, []),
    "dd3d_conv128_bwd_weight_clouds_ws_bytes": (I64, [I64, I32]),
    "dd3d_conv128_bwd_weight_clouds": (I32, [P, P, P, I64, I32, I64, P, I32, P, I32, P, P, P, I64, P]),
    "dd3d_conv_any_max_channels": (I32, []),
    "dd3d_conv_any_fwd": (I32, [P, I64, I32, P, I32, P, I64, I32, I64, P, P, P]),
    "dd3d_conv_any_bwd_data": (I32, [P, I64, P, I32, P, I32, P, I64, I32, P, P]),
    "dd3d_conv_any_bwd_weight_ws_bytes": (I64, [I32, I32, I32]),
    "dd3d_conv_any_bwd_weight": (I32, [P, I32, P, I32, P, I64, I32, I64, P, P, P, I64, P]),
    "dd3d_transpose_kernel": (I32, [P, I32, I32, I32, P, P]),
    "dd3d_voxel_mean_fwd": (I32, [P, I64, I32, P, P, I64, P, P, P]),
    "dd3d_voxel_mean_bwd": (I32, [P, I64, I32, P, P, I64, P, P]),
    "dd3d_inorm_fwd": (I32, [P, P, I32, I32, P, P, P]),
    "dd3d_inorm_bwd": (I32, [P, P, P, P, I32, I32, P, P]),
    "dd3d_inorm_arp_fwd": (I32, [P, P, I32, I32, P, P, P, P, P, I64, I32, P, P, P, P]),
    "dd3d_inorm_arp_fwd_y": (I32, [P, P, I32, I32, P, P, P, P, P, I64, I32, P, P, P, P]),
    "dd3d_affine_relu_pool_fwd": (I32, [P, P, P, I32, I32, P, I64, I32, I64, P, P, P, P]),
    "dd3d_avg_pool_fwd": (I32, [P, I64, I32, P, I64, I32, I64, P, P, P, P]),
    "dd3d_avg_pool_bwd": (I32, [P, P, P, I64, I64, I32, P, P]),
    "dd3d_global_avg_pool_fwd": (I32, [P, I64, P, I32, I32, P, P]),
    "dd3d_global_avg_pool_bwd": (I32, [P, I64, P, I32, I32, P, P]),
    "dd3d_arp_bwd_ws_bytes": (I64, [I64, I32]),
    "dd3d_affine_relu_pool_bwd": (I32, [P, P, P, P, P, P, I32, I32, I64, P, I32, P, P, P, P, P, P, P, I64, P]),
    "dd3d_arp_inorm_bwd_ws_bytes": (I64, [I32]),
    "dd3d_arp_inorm_bwd": (I32, [P, P, P, P, P, P, P, I32, P, I32, P, P, I32, P, P, I32, P, P, P, P, P, I64, P]),
    "dd3d_arp_inorm_bwd_y": (I32, [P, P, P, P, P, P, P, P, I32, P, I32, P, P, I32, P, P, I32, P, P, P, P, P, I64, P]),
    "dd3d_arp_inorm_bwd_static_max_clouds": (I32, []),
    "dd3d_arp_inorm_bwd_static": (I32, [P, P, P, P, P, P, P, I32, P, I32, P, P, I32, P, P, I32, P, P, P, P, P]),
    "dd3d_scale_rows_by_group": (I32, [P, P, P, I32, I64, I32, P, P]),
    "dd3d_sparse_linear_ws_bytes": (I64, [I64, I32]),
    "dd3d_sparse_linear_fwd": (I32, [P, P, P, I64, P, I32, I32, P, P, P, I32, P, P, P, I64, P]),
    "dd3d_classifier_head": (I32, [P, P, P, I64, P, I32, I32, P, P, P, I32, P, P, P, I32, I32, P, P, P, P, P, P, P, P, I64, P]),
    "dd3d_sparse_linear_bwd_x": (I32, [P, P, P, I64, P, I32, I32, P, P, P, I32, I32, P, P]),
    "dd3d_sparse_linear_bwd_w": (I32, [P, P, P, I32, I32, P, I32, I32, I32, P, P, I64, P]),
    "dd3d_classifier_layout": (I32, [P, I32, I32, I32, I32, I32, P, P]),
    "dd3d_softmax_ce": (I32, [P, P, P, P, I32, I32, I32, P, P, P, P, P]),
    "dd3d_softmax_ce_bwd2": (I32, [P, P, P, P, I32, I32, I32, P, P]),
    "dd3d_match_loss_ws_bytes": (I64, [P, I32, I32]),
    "dd3d_match_loss_stats_floats": (I64, [P, I32, I32]),
    "dd3d_match_loss_fwd": (I32, [P, I32, I32, I32, P, P, P, I64, P]),
    "dd3d_match_loss_bwd": (I32, [P, I32, I32, I32, P, P, P, I64, P]),
    "dd3d_match_loss_fwd_bwd": (I32, [P, I32, I32, I32, P, P, P, P, I64, P]),
    "dd3d_group_sum_rows": (I32, [P, P, I32, I32, P, P]),
    "dd3d_loss_total": (I32, [P, P, P, P, I32, P, P, P]),
    "dd3d_sgd_momentum_clamp": (I32, [P, P, P, I64, F32, F32, I32, I32, F32, F32, P]),
}

EXPORTS = tuple(_SIGS)

for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)  # AttributeError here = header and library disagree
    _f.restype, _f.argtypes = _r, _a


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: there is no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("dd3d_amd kernels need GPU tensors (no CPU fallback); got a tensor on " + str(t.device))
    if not t.is_contiguous():
        raise RuntimeError("dd3d_amd kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, name: str):
    if rc == 0:
        return
    if rc < 0:
        what = {-1: "bad argument", -2: "size out of range", -3: "workspace too small"}.get(rc, "argument error")
        raise RuntimeError(f"{name}: {what} (rc={rc})")
    raise RuntimeError(f"{name}: HIP error {rc}")


def call(name: str, *args):
    check(getattr(lib, name)(*args), name)


_ws_cache = {}
_ws_captured = {}      # key -> serial of the capture under which the block was last handed out (that graph keeps the address)
_capture_serial = [0]


def note_capture():
    """Call right before opening a graph capture: workspaces handed out under an EARLIER capture are frozen from here on (a block may
    still grow while the capture that first asked for it is being recorded -- stream-ordered reuse inside one recording is safe)."""
    _capture_serial[0] += 1
_ws_scope = ["ws"]


class ws_scope:
    """Workspaces requested inside ``with ws_scope(tag)`` come from a cache private to ``tag``: a captured HIP graph keeps the
    ADDRESSES of its workspaces, so they must never be regrown or shared with eager callers."""

    def __init__(self, tag: str):
        self.tag = tag

    def __enter__(self):
        _ws_scope.append(self.tag)

    def __exit__(self, *exc):
        _ws_scope.pop()


def drop_scope(tag: str):
    """Forget the workspaces of scope ``tag`` and of its sub-scopes (``tag`` + suffix)."""
    for k in [k for k in _ws_cache if k[2].startswith(tag)]:
        del _ws_cache[k]
        _ws_captured.pop(k, None)


_hip = None


def hip_last_error() -> str:
    """Name of the HIP runtime's pending error ("" if none), WITHOUT clearing it (hipPeekAtLastError of the runtime this process already
    has loaded): what a teardown prints so that a failure names itself."""
    global _hip
    try:
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so.7")   # by SONAME: the runtime torch and libdd3d_hip.so already share, never a second copy
            _hip.hipGetErrorName.restype = C.c_char_p
        rc = int(_hip.hipPeekAtLastError())
        return "" if rc == 0 else "%s (%d)" % ((_hip.hipGetErrorName(rc) or b"?").decode(), rc)
    except Exception:
        return ""


def workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """A per-(device, stream, tag) scratch tensor that only grows (stream-ordered reuse is safe within one stream)."""
    if tag == "ws":
        tag = _ws_scope[-1]
    key = (str(device), torch.cuda.current_stream(device).cuda_stream, tag)
    t = _ws_cache.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if t is None or t.numel() < nbytes:
        if t is not None and key in _ws_captured and not (capturing and _ws_captured[key] == _capture_serial[0]):
            # a FINISHED capture holds this block's ADDRESS: growing it would free memory that graph still writes on every replay
            raise RuntimeError("workspace of a captured HIP graph cannot grow (%d -> %d bytes, scope %r): drop the graph and its scope first "
                               "(FusedDistiller.invalidate / close)" % (t.numel(), nbytes, tag))
        t = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = t
    if capturing:   # (also a block the eager first step allocated: side streams keep their keys under capture)
        _ws_captured[key] = _capture_serial[0]
    return t


def true_divide(t: torch.Tensor, q: float) -> torch.Tensor:
    """t / q as the reference computes coordinates / quantization_size (models/MEConv.py:262 divides HOST tensors: a correctly rounded
    fp32 division).  torch's device kernel for ``tensor / python_scalar`` multiplies by the rounded reciprocal instead, which moves a
    point that lies within an ulp of a voxel face into the neighbouring voxel whenever the voxel size is not a power of two; dividing
    by a 0-d DEVICE tensor takes the true-division kernel.  The HIP kernels that build coordinates (dd3d_assemble_batch,
    dd3d_gather_clouds, dd3d_syn_head) divide the same way, so the eager and the captured step quantise identically."""
    if q == 1.0:
        return t.clone()
    return torch.div(t, torch.tensor(q, dtype=t.dtype, device=t.device))
