"""Placement gate: replicas, column shards or row shards.

The reference decides per table whether it may live on the GPU at all: a table larger
than ``max_table_size = 1 << 28`` bytes keeps its column on the CPU
(``check_table_size``, ``graph_optimizers/cuda_emitter.cc:1080-1094``;
``fc_optimize_pass.cc:71``; ``RECOM_CPU_GPU_CO_RUN``).  With 288 GB of HBM per MI355X the
question becomes whether the MODEL fits one GPU: if it does every GPU serves its own
requests on its own replica and nothing is exchanged; tables are sharded over the GPUs
of the node — with one RCCL all-to-all per request — only when their aggregate exceeds
one GPU (``BASELINE.json`` north star).  The decision itself is native
(``fcp_placement_decide``); this module feeds it a :class:`PlanSpec`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import lib as _lib
from .plan import FORM_GATHER, FORM_GATHER_SCATTER, FORM_SEGMENT_REDUCE, PlanSpec

MI355X_HBM_BYTES = 288 * 10**9          # HBM3E per GPU (MI355X_MICROARCH.md)
DEFAULT_RESERVE_BYTES = 8 << 30         # arenas, request blobs, RCCL buffers, runtime

REPLICATE, COLUMN_SHARD, ROW_SHARD, MIXED = 0, 1, 2, 3
MODE_NAMES = {REPLICATE: "replicas", COLUMN_SHARD: "column-sharded", ROW_SHARD: "row-sharded",
              MIXED: "mixed (whole tables where they fit, rows for the rest)"}


@dataclass
class Placement:
    mode: int
    min_world: int
    bytes_per_gpu: int
    total_bytes: int
    owners: Optional[Sequence[int]] = None      # per table input: the rank that holds it whole, or -1 = spread by rows

    @property
    def name(self) -> str:
        return MODE_NAMES[self.mode]


def table_bytes(spec: PlanSpec) -> np.ndarray:
    """Bytes of every table input of the plan (a shared table counted once)."""
    out = np.zeros(spec.n_device_inputs, np.int64)
    for c in spec.columns:
        if c.form in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
            out[c.table_input] = max(int(out[c.table_input]), int(c.vocab) * c.dim * 4)
    return out


def decide_placement(spec_or_bytes, world: int, hbm_bytes: Optional[int] = None,
                     reserve_bytes: int = DEFAULT_RESERVE_BYTES, prefer: str = "row") -> Placement:
    """``spec_or_bytes``: a :class:`PlanSpec` or the table sizes in bytes; ``prefer``: "row", "column" or "mixed" (whole
    tables wherever a table fits one GPU, rows only for those that do not: the fewest bytes on the wire).  Raises
    :class:`recom_amd.lib.FcpError` (``FCP_ERR_UNSUPPORTED``) when the tables do not fit ``world``
    GPUs; the message names the smallest world that would do."""
    L = _lib.load()
    tb = table_bytes(spec_or_bytes) if isinstance(spec_or_bytes, PlanSpec) else np.asarray(spec_or_bytes, np.int64)
    tb = np.ascontiguousarray(tb, np.int64)
    out = _lib.Placement()
    owners = np.zeros(len(tb), np.int32)
    _lib.check(L.fcp_placement_assign(tb.ctypes.data, len(tb), int(hbm_bytes or MI355X_HBM_BYTES), int(reserve_bytes),
                                      int(world), {"row": ROW_SHARD, "column": COLUMN_SHARD, "mixed": MIXED}[prefer],
                                      owners.ctypes.data, C.byref(out)),
               "fcp_placement_assign")
    return Placement(out.mode, out.min_world, out.bytes_per_gpu, int(tb.sum()), [int(o) for o in owners])


def device_hbm_bytes(device: int = 0) -> int:
    """Total memory of a GPU as the runtime reports it (the gate's `hbm_bytes` on a real box)."""
    import torch
    return int(torch.cuda.get_device_properties(device).total_memory)
