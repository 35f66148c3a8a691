"""ctypes binding of ``libfcp_hip.so`` (C ABI: ``include/fcp_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc
--offload-arch=gfx950``).  There is NO fallback: if the shared object is
missing or cannot be loaded, importing the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCP_LIB_DIR: tuning aid, loads an alternative build (e.g. build/abl4) of the same sources
LIB_PATH = os.path.join(os.environ.get("FCP_LIB_DIR", _HERE), "libfcp_hip.so")

FCP_ABI_VERSION = 2
FCP_OK = 0
FCP_ERR_INVALID_ARGUMENT = 1
FCP_ERR_SHAPE_MISMATCH = 2
FCP_ERR_ALLOC = 3
FCP_ERR_HIP = 4
FCP_ERR_UNSUPPORTED = 5
FCP_ERR_NO_DEVICE = 6
FLAG_HOST_ONLY = 1 << 31  # plan without device resources (layout queries only)


class FcpError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = "") -> None:
        self.status = status
        super().__init__(f"{what}: status {status}" + (f" ({detail})" if detail else ""))


class ColumnDesc(C.Structure):
    _fields_ = [
        ("form", C.c_int32), ("combiner", C.c_int32), ("dim", C.c_int32), ("id_source", C.c_int32),
        ("vocab", C.c_int64),
        ("table_input", C.c_int32), ("ids_input", C.c_int32), ("seg_input", C.c_int32),
        ("seg_kind", C.c_int32), ("seg_stride", C.c_int32),
        ("rows_source", C.c_int32), ("rows_arg", C.c_int32),
        ("n_boundaries", C.c_int32),
        ("boundaries", C.POINTER(C.c_float)),
        ("concat_group", C.c_int32), ("concat_slot", C.c_int32),
        ("xform_mode", C.c_int32), ("xform_n", C.c_int32),
        ("xform_lo", C.POINTER(C.c_int64)), ("xform_hi", C.POINTER(C.c_int64)),
        ("xform_substitute", C.c_int64), ("hash_buckets", C.c_int64),
    ]


class ColumnExt(C.Structure):
    """fcp_column_ext_t: per-column extensions (segment-id maps)"""
    _fields_ = [
        ("seg_map_n", C.c_int32), ("seg_map_sym", C.c_int32), ("seg_map_sym_slot", C.c_int32), ("reserved0", C.c_int32),
        ("seg_map_mul", C.c_int64 * 4), ("seg_map_div", C.c_int64), ("reserved1", C.c_int64 * 2),
    ]


class PlanDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_columns", C.c_int32),
        ("columns", C.POINTER(ColumnDesc)),
        ("n_host_inputs", C.c_int32),
        ("host_input_ranks", C.POINTER(C.c_int32)),
        ("host_input_elem_sizes", C.POINTER(C.c_int32)),
        ("n_device_inputs", C.c_int32), ("n_groups", C.c_int32), ("n_symbols", C.c_int32),
        ("layout", C.c_int32), ("device", C.c_int32),
        ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
        ("flags", C.c_uint32),
    ]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)


class ProcessArgs(C.Structure):
    _fields_ = [
        ("concated_inputs", C.c_void_p), ("concated_bytes", C.c_int64),
        ("concated_offsets", C.POINTER(C.c_int32)), ("concated_shapes", C.POINTER(C.c_int32)),
        ("input_ptrs", C.POINTER(C.c_void_p)), ("input_shapes", C.POINTER(C.c_int32)),
        ("symbols", C.POINTER(C.c_int32)),
        ("stream", C.c_void_p),
        ("malloc_temp", ALLOC_FN), ("malloc_temp_ctx", C.c_void_p),
        ("malloc_buff", ALLOC_FN), ("malloc_buff_ctx", C.c_void_p),
    ]


class ProcessResult(C.Structure):
    _fields_ = [
        ("output_ptrs", C.POINTER(C.c_void_p)), ("output_shapes", C.POINTER(C.c_int32)),
        ("output_row_strides", C.POINTER(C.c_int64)),
        ("group_ptrs", C.POINTER(C.c_void_p)), ("group_shapes", C.POINTER(C.c_int32)),
        ("buffer", C.c_void_p), ("buffer_bytes", C.c_int64),
    ]


class Placement(C.Structure):
    _fields_ = [("mode", C.c_int32), ("min_world", C.c_int32), ("bytes_per_gpu", C.c_int64)]


class PrivateStreamsStats(C.Structure):
    """fcp_private_streams_stats_t"""
    _fields_ = [("supervised_stream", C.c_void_p), ("requests", C.c_int64), ("lane_requests", C.c_int64), ("evaluations", C.c_int64),
                ("stream_order_us_per_mib", C.c_double), ("last_ratio", C.c_double), ("worst_ratio", C.c_double),
                ("keep_ratio", C.c_double), ("demoted", C.c_int32), ("evaluation_in_progress", C.c_int32)]


class StagerStats(C.Structure):
    """fcp_stager_stats_t"""
    _fields_ = [("calls", C.c_int64), ("copy_calls", C.c_int64), ("copy_calls_over_1ms", C.c_int64), ("fallback_switches", C.c_int64),
                ("requests_with_blocked_copy", C.c_int64), ("max_copy_call_us", C.c_double)]


class HostTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("elem_size", C.c_int32), ("rank", C.c_int32),
                ("dims", C.POINTER(C.c_int64))]


# every symbol include/fcp_hip.h declares
EXPORTS = [
    "fcp_abi_version", "fcp_status_string", "fcp_last_error",
    "fcp_concat_inputs_sizes", "fcp_concat_inputs",
    "fcp_plan_create", "fcp_plan_create_ex", "fcp_plan_create_from_file", "fcp_plan_counts", "fcp_plan_destroy", "fcp_plan_group_width",
    "fcp_plan_column_offset",
    "fcp_plan_arena_bytes", "fcp_plan_read_bad_ids", "fcp_plan_output_columns", "fcp_plan_table_bytes",
    "fcp_placement_decide", "fcp_plan_release_captures",
    "fcp_process_feature_columns", "fcp_concat_outputs", "fcp_concat_outputs_scatter", "fcp_concat_outputs_host",
    "fcp_shard_finalize", "fcp_comm_unique_id", "fcp_comm_create", "fcp_comm_destroy", "fcp_comm_rank",
    "fcp_shard_batch_slice", "fcp_shard_exchange", "fcp_shard_exchange_columns",
    "fcp_shard_step_create", "fcp_shard_step_run", "fcp_shard_step_destroy",
    "fcp_stager_create", "fcp_stager_create_ex", "fcp_stager_stage", "fcp_stager_stage_ex", "fcp_stager_stage_narrow", "fcp_stager_destroy", "fcp_stager_stats",
    "fcp_concat_inputs_ex_sizes", "fcp_concat_inputs_ex", "fcp_plan_file_stage_info",
    "fcp_pack_pool_create", "fcp_pack_pool_destroy", "fcp_concat_inputs_ex_pool",
    "fcp_graph_build", "fcp_graph_free", "fcp_placement_assign", "fcp_concat_outputs_scatter_strided",
    "fcp_plan_set_private_streams", "fcp_result_wait", "fcp_result_synchronize", "fcp_plan_set_request_order",
    "fcp_plan_probe_private_streams", "fcp_plan_private_streams_verdict", "fcp_plan_verify_private_streams",
    "fcp_plan_private_streams_stats",
]

_lib = None


def _build_in_tree() -> None:
    """`make -C recom_amd/csrc` (hipcc --offload-arch=gfx950) when the shared objects are missing,
    e.g. in a fresh checkout.  Silent no-op when there is no hipcc: load() then fails loudly."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")) or not shutil.which("make"):
        return
    # several ranks of one node may get here at once (torch.distributed.run): one builds, the others wait
    import fcntl
    try:
        with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(LIB_PATH):
                subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc")], stdout=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, OSError):
        pass


def load() -> C.CDLL:
    """Load libfcp_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch (device memory / stream plumbing of the Python host side) bundles its
    # own HIP runtime.  Load it first so the process ends up with ONE runtime: if
    # libfcp_hip.so pulled in /opt/rocm's copy before torch initialised its own, the
    # second runtime to start would not see the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:  # the C ABI itself does not need torch
        pass
    if not os.path.exists(LIB_PATH) and "FCP_LIB_DIR" not in os.environ:
        _build_in_tree()                     # fresh checkout on a box with hipcc: build, never fall back
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    L.fcp_abi_version.restype = C.c_int
    L.fcp_status_string.restype = C.c_char_p
    L.fcp_status_string.argtypes = [C.c_int]
    L.fcp_last_error.restype = C.c_char_p
    L.fcp_concat_inputs_sizes.argtypes = [C.POINTER(HostTensor), C.c_int32, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int32)]
    L.fcp_concat_inputs.argtypes = [C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p]
    L.fcp_plan_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(C.c_void_p)]
    if hasattr(L, "fcp_plan_create_ex"):
        L.fcp_plan_create_ex.argtypes = [C.POINTER(PlanDesc), C.POINTER(ColumnExt), C.POINTER(C.c_void_p)]
    L.fcp_plan_create_from_file.argtypes = [C.c_char_p, C.c_int32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.fcp_plan_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 5
    L.fcp_plan_destroy.argtypes = [C.c_void_p]
    L.fcp_plan_group_width.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fcp_plan_column_offset.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.fcp_plan_arena_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.fcp_plan_read_bad_ids.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    L.fcp_process_feature_columns.argtypes = [C.c_void_p, C.POINTER(ProcessArgs), C.POINTER(ProcessResult)]
    L.fcp_concat_outputs.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                     C.c_void_p]
    L.fcp_plan_output_columns.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int32]
    L.fcp_plan_table_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.fcp_plan_release_captures.argtypes = [C.c_void_p]
    L.fcp_placement_decide.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                       C.POINTER(Placement)]
    L.fcp_concat_outputs_scatter.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                             C.c_int32, C.c_void_p, C.c_void_p]
    L.fcp_concat_outputs_host.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                          C.c_int32, C.c_void_p, ALLOC_FN, C.c_void_p, C.c_int32, C.c_void_p]
    L.fcp_shard_finalize.argtypes = [C.c_void_p, C.POINTER(ProcessArgs), C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.fcp_comm_unique_id.argtypes = [C.c_void_p]
    L.fcp_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.fcp_comm_destroy.argtypes = [C.c_void_p]
    L.fcp_comm_rank.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.fcp_shard_batch_slice.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.fcp_shard_exchange.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.c_void_p]
    L.fcp_shard_exchange_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]
    L.fcp_shard_step_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p,
                                        C.POINTER(C.c_void_p)]
    L.fcp_shard_step_run.argtypes = [C.c_void_p, C.POINTER(ProcessArgs), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64)]
    L.fcp_shard_step_destroy.argtypes = [C.c_void_p]
    L.fcp_stager_stage_ex.argtypes = [C.c_void_p, C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int32)),
                                      C.POINTER(C.POINTER(C.c_int32))]
    L.fcp_stager_create_ex.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                       C.POINTER(C.c_void_p)]
    L.fcp_stager_create.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_void_p)]
    L.fcp_stager_stage.argtypes = [C.c_void_p, C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int32)),
                                   C.POINTER(C.POINTER(C.c_int32))]
    L.fcp_stager_stage_narrow.argtypes = [C.c_void_p, C.POINTER(HostTensor), C.c_int32, C.POINTER(C.c_uint8),
                                          C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                          C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
    L.fcp_stager_destroy.argtypes = [C.c_void_p]
    if hasattr(L, "fcp_stager_stats"):
        L.fcp_stager_stats.argtypes = [C.c_void_p, C.POINTER(StagerStats)]
    if hasattr(L, "fcp_pack_pool_create"):
        L.fcp_pack_pool_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
        L.fcp_pack_pool_destroy.argtypes = [C.c_void_p]
        L.fcp_concat_inputs_ex_pool.argtypes = [C.c_void_p, C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p]
    if "FCP_LIB_DIR" not in os.environ or hasattr(L, "fcp_concat_inputs_ex"):  # (an older A/B build may predate these)
        L.fcp_concat_inputs_ex_sizes.argtypes = [C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                                 C.POINTER(C.c_int32)]
        L.fcp_concat_inputs_ex.argtypes = [C.POINTER(HostTensor), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_void_p]
        L.fcp_plan_file_stage_info.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_int32,
                                               C.POINTER(C.c_int32)]
        L.fcp_graph_build.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                      C.POINTER(C.c_char_p)]
        L.fcp_graph_free.argtypes = [C.c_void_p]
        L.fcp_graph_free.restype = None
        L.fcp_placement_assign.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                           C.POINTER(Placement)]
        L.fcp_concat_outputs_scatter_strided.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                         C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    if hasattr(L, "fcp_plan_set_private_streams"):                              # (an older A/B build may predate these)
        L.fcp_plan_set_private_streams.argtypes = [C.c_void_p, C.c_int32, C.c_uint32]
        L.fcp_result_wait.argtypes = [C.c_void_p, C.c_void_p]
        L.fcp_result_synchronize.argtypes = [C.c_void_p]
    if hasattr(L, "fcp_plan_probe_private_streams"):
        L.fcp_plan_probe_private_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.fcp_plan_private_streams_verdict.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    if hasattr(L, "fcp_plan_verify_private_streams"):
        L.fcp_plan_verify_private_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.fcp_plan_private_streams_stats.argtypes = [C.c_void_p, C.POINTER(PrivateStreamsStats)]
    if hasattr(L, "fcp_plan_set_request_order"):
        L.fcp_plan_set_request_order.argtypes = [C.c_void_p, C.c_int32]
    if L.fcp_abi_version() != FCP_ABI_VERSION:
        raise ImportError("libfcp_hip.so ABI version mismatch; rebuild")
    _lib = L
    return L


def check(status: int, what: str) -> None:
    if status != FCP_OK:
        L = load()
        detail = L.fcp_last_error().decode(errors="replace") or L.fcp_status_string(status).decode()
        raise FcpError(status, what, detail)
