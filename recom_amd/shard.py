"""Tables sharded across the GPUs of one node (BASELINE.json configs[4]): by rows
(:class:`RowShardedPath`) or by columns (:class:`ColumnShardedPath`).

The reference has no multi-GPU code at all; this is the
MI355X-native addition for models whose tables exceed one GPU's 288 GB HBM.

Partitioning: rank ``g`` of ``world`` owns the table rows ``{r : r % world == g}``
(stored densely, local row ``r // world``).  Every rank receives the whole request
(ids are tiny next to table rows), gathers only the ids it owns and produces a
*partial* pooled sum ``P_g[rows, width]`` (``fcp`` plan with ``shard_rank`` /
``shard_world``).  One exchange step follows: an all-to-all of the partials
partitioned along the batch — rank ``h`` receives slice ``P_g[rows_h, :]`` from every
``g`` — realised with ``all_to_all_single`` (RCCL over xGMI: point-to-point, so all
7 links of a GPU carry traffic concurrently, unlike a ring all-reduce).  Rank ``h``
then adds its ``world`` slices in rank order (deterministic) and applies the mean
division (``fcp_shard_finalize``).  Columns with exactly one owner per row (dense
one-hot, scatter, passthrough) come out bit-identical to the single-GPU result.

When the tables fit one GPU no collective is used at all: requests are sharded
across replicas (``bench.py`` default).
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def _all_to_all(recv, send, out_splits, in_splits, group) -> None:
    """``all_to_all_single`` on the process group's backend.  RCCL (``nccl``) moves device
    tensors directly over xGMI.  Under ``gloo`` — only used to exercise the N > 1 control
    flow where every rank shares one GPU — device tensors are staged through the host."""
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = recv.cpu()
        dist.all_to_all_single(r, send.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=group)
        recv.copy_(r)
        return
    dist.all_to_all_single(recv, send, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def batch_slices(rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous split of the batch: (begin, count) per rank; the first
    ``rows % world`` ranks get one extra row."""
    base, extra = divmod(rows, world)
    out, begin = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((begin, cnt))
        begin += cnt
    return out


class RowShardedPath:
    """Exchange + finalize around a per-rank partial computation.

    ``partial_fn() -> tensor [rows, width]`` (this rank's partial sums, on the
    device the process group communicates from) and ``finalize_fn(slices[world,
    count, width], begin, count) -> tensor [count, width]`` are injected, so the same
    orchestration runs on GPUs (``FeatureColumnProcess`` + ``shard_finalize`` over
    RCCL) and in the CPU tests (oracle + NumPy over gloo).
    """

    def __init__(self, rank: int, world: int, group=None) -> None:
        self.rank, self.world, self.group = rank, world, group

    def exchange(self, partial):
        import torch
        import torch.distributed as dist
        rows, width = partial.shape
        sl = batch_slices(rows, self.world)
        begin, count = sl[self.rank]
        if self.world == 1:
            return partial.reshape(1, rows, width), begin, count
        recv = torch.empty((self.world * count, width), dtype=partial.dtype, device=partial.device)
        _all_to_all(recv, partial.contiguous(), [count] * self.world, [c for _, c in sl], self.group)
        return recv.view(self.world, count, width), begin, count

    def run(self, partial_fn: Callable, finalize_fn: Callable):
        partial = partial_fn()
        slices, begin, count = self.exchange(partial)
        return finalize_fn(slices, begin, count), begin, count

    def all_gather_batch(self, mine, rows: int):
        """Optional: give every rank the full [rows, width] result (slices are padded
        to the largest count so one all_gather suffices)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return mine
        sl = batch_slices(rows, self.world)
        cmax = max(c for _, c in sl)
        padded = torch.zeros((cmax, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        padded[:mine.shape[0]] = mine
        parts = [torch.empty_like(padded) for _ in range(self.world)]
        dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:c] for p, (_, c) in zip(parts, sl)], dim=0)


def assign_columns(spec, world: int, weights: Optional[Sequence[float]] = None) -> List[List[int]]:
    """Column (table-wise) sharding: split every concat group's columns, in concat
    order, into ``world`` contiguous ranges of about equal weight (default weight:
    ``dim`` — the HBM traffic per batch row, and the table bytes when vocabularies are
    alike).  Contiguous in concat order, so rank ``g``'s result is one column block of
    the group matrix.  Returns the column indices per rank (all groups)."""
    out: List[List[int]] = [[] for _ in range(world)]
    for g in range(spec.n_groups):
        members = [k for _, k in sorted((c.concat_slot, k) for k, c in enumerate(spec.columns) if c.concat_group == g)]
        if len(members) < world:
            raise ValueError(f"group {g}: {len(members)} columns cannot be split over {world} ranks")
        w = np.asarray([float(weights[k]) if weights is not None else float(spec.columns[k].dim) for k in members])
        acc = np.concatenate([[0.0], np.cumsum(w)])
        cuts = [0]
        for r in range(1, world):
            c = int(np.searchsorted(acc, acc[-1] * r / world, side="left"))
            c = min(max(c, cuts[-1] + 1), len(members) - (world - r))  # every rank gets >= 1 column
            cuts.append(c)
        cuts.append(len(members))
        for r in range(world):
            out[r].extend(members[cuts[r]:cuts[r + 1]])
    return out


class ColumnShardedPath:
    """Table-wise sharding (SURVEY.md §8e "cheaper alternative"): rank ``g`` owns whole
    columns, pools them for the whole batch, and one all-to-all partitioned along the
    batch hands rank ``h`` the rows ``rows_h`` of every rank's column block.  No
    partial sums cross the wire, so every column — pooled ones included — is
    bit-identical to the single-GPU result.  Valid while no single table exceeds one
    GPU; row sharding (:class:`RowShardedPath`) covers the rest."""

    def __init__(self, rank: int, world: int, group=None) -> None:
        self.rank, self.world, self.group = rank, world, group

    def exchange(self, block, widths: Sequence[int]):
        """``block [rows, widths[rank]]`` -> list of ``world`` tensors
        ``[count, widths[g]]`` (this rank's batch rows of every rank's block)."""
        import torch
        import torch.distributed as dist
        rows, width = block.shape
        assert width == widths[self.rank]
        sl = batch_slices(rows, self.world)
        begin, count = sl[self.rank]
        if self.world == 1:
            return [block], begin, count
        recv = torch.empty(count * int(sum(widths)), dtype=block.dtype, device=block.device)
        _all_to_all(recv, block.contiguous().view(-1), [count * int(w) for w in widths],
                    [c * width for _, c in sl], self.group)
        parts, pos = [], 0
        for w in widths:
            parts.append(recv[pos:pos + count * int(w)].view(count, int(w)))
            pos += count * int(w)
        return parts, begin, count

    def run(self, block_fn: Callable, widths: Sequence[int], concat_fn: Callable):
        parts, begin, count = self.exchange(block_fn(), widths)
        return concat_fn(parts), begin, count


class ColumnShardedFeatureColumns:
    """GPU implementation: one plan per rank over the columns it owns; the final
    ``[count, width]`` matrix is assembled by ``fcp_concat_outputs`` from the received
    column blocks."""

    def __init__(self, model, rank: int, world: int, device: int, group=None, assignment=None) -> None:
        import torch
        from .ops import FeatureColumnProcess
        self.torch = torch
        self.model = model
        self.assignment = assignment if assignment is not None else assign_columns(model.spec, world)
        self.sub = model.spec.column_subset(self.assignment[rank])
        self.dev = torch.device("cuda", device)
        self.op = FeatureColumnProcess(self.sub.spec, device)
        self.tables = self._tables(model)
        self.path = ColumnShardedPath(rank, world, group)
        self.widths = [[sum(model.spec.columns[k].dim for k in cols if model.spec.columns[k].concat_group == g)
                        for cols in self.assignment] for g in range(model.spec.n_groups)]

    def _tables(self, model):
        from .synth import hash_table_torch
        return [hash_table_torch(model.tables[i].seed, model.tables[i].vocab, model.tables[i].dim, self.dev)
                for i in self.sub.device_inputs]

    def request_inputs(self, inputs):
        """The host tensors of a full request this rank needs."""
        return [inputs[i] for i in self.sub.host_inputs]

    def __call__(self, d_blob, offsets, shapes, symbols, group: int = 0):
        from .ops import concat_outputs
        op, tables = self.op, self.tables

        def block():
            return op.groups_only(d_blob, offsets, shapes, tables, symbols)[group]

        return self.path.run(block, self.widths[group], lambda parts: concat_outputs(parts) if len(parts) > 1
                             else parts[0])


class ShardedFeatureColumns:
    """GPU implementation: one row-sharded plan per rank."""

    def __init__(self, model, rank: int, world: int, device: int, group=None) -> None:
        import torch
        from .ops import FeatureColumnProcess
        self.torch = torch
        self.model = model
        self.spec = model.spec.with_shard(rank, world)
        self.dev = torch.device("cuda", device)
        self.op = FeatureColumnProcess(self.spec, device)
        self.tables = model.torch_tables(self.dev, rank, world)
        self.path = RowShardedPath(rank, world, group)

    def __call__(self, d_blob, offsets, shapes, symbols, group: int = 0):
        op, tables = self.op, self.tables

        def partial():
            return op.groups_only(d_blob, offsets, shapes, tables, symbols)[group]

        def finalize(slices, begin, count):
            return op.shard_finalize(d_blob, offsets, shapes, tables, symbols, group, slices.contiguous(),
                                     self.path.world, begin, count)

        return self.path.run(partial, finalize)


class Communicator:
    """One RCCL communicator of this process' GPU, owned by ``libfcp_hip.so`` (``fcp_comm_*``).  The
    128-byte id is created on rank 0 and shipped with ``torch.distributed`` (any backend; a host-side
    channel, not the data path)."""

    def __init__(self, rank: int, world: int, device: int, dist=None) -> None:
        import ctypes as C
        import torch
        from . import lib as _lib
        self._L = _lib.load()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            _lib.check(self._L.fcp_comm_unique_id(ident), "fcp_comm_unique_id")
        if world > 1:
            t = torch.tensor(list(ident), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.cuda(device)
            dist.broadcast(t, src=0)
            ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        h = C.c_void_p()
        _lib.check(self._L.fcp_comm_create(ident, rank, world, device, C.byref(h)), "fcp_comm_create")
        self.handle, self.rank, self.world, self.device = h, rank, world, device

    def close(self) -> None:
        if getattr(self, "handle", None):
            self._L.fcp_comm_destroy(self.handle)
            self.handle = None


class NativeShardedStep:
    """The sharded serving step as ONE native call per request (``fcp_shard_step_run``): partial kernel ->
    grouped ncclSend / ncclRecv over xGMI -> ``fcp_shard_finalize`` (row mode) or concat (column mode), on
    one stream, from buffers the library owns.  Python prepares the request records once."""

    def __init__(self, model, comm: Communicator, mode: str = "row", group: int = 0, assignment=None) -> None:
        """``assignment`` (column mode): the columns every rank owns, as lists of column indices in concat order; default:
        contiguous ranges of about equal width (:func:`assign_columns`)."""
        import ctypes as C
        import torch
        from . import lib as _lib
        from .ops import Plan
        from .placement import COLUMN_SHARD, ROW_SHARD
        self.torch, self._lib, self._L, self.comm, self.mode = torch, _lib, _lib.load(), comm, mode
        self.dev = torch.device("cuda", comm.device)
        rank, world = comm.rank, comm.world
        if mode == "row":
            self.spec = model.spec.with_shard(rank, world)
            self.tables = model.torch_tables(self.dev, rank, world)
            self.host_inputs = list(range(model.spec.n_host_inputs))
            widths = None
        else:
            from .synth import hash_table_torch
            assignment = assignment if assignment is not None else assign_columns(model.spec, world)
            sub = model.spec.column_subset(assignment[rank])
            self.spec, self.host_inputs = sub.spec, sub.host_inputs
            self.tables = [hash_table_torch(model.tables[i].seed, model.tables[i].vocab, model.tables[i].dim, self.dev)
                           for i in sub.device_inputs]
            widths = np.asarray([sum(model.spec.columns[k].dim for k in cols if model.spec.columns[k].concat_group == group)
                                 for cols in assignment], np.int32)
        self.width = int(model.spec.group_width(group)) if mode == "row" else int(widths.sum())
        self.plan = Plan(self.spec, comm.device)
        self._tptrs = (C.c_void_p * max(1, len(self.tables)))(*[t.data_ptr() for t in self.tables])
        self._widths = widths
        self._step = None
        self._keep = []
        self._group = group

    def prepare(self, inputs, symbols):
        """Pack one request (this rank's share of its host tensors), put it in HBM and build the argument
        record ``run`` replays.  Returns an opaque request."""
        import ctypes as C
        from .ops import concat_inputs
        blob, offsets, shapes = concat_inputs([inputs[i] for i in self.host_inputs])
        d_blob = self.torch.from_numpy(blob).to(self.dev)
        sym = None if symbols is None else np.ascontiguousarray(symbols, np.int32)
        a = self._lib.ProcessArgs(
            d_blob.data_ptr() if blob.size else None, blob.nbytes, offsets.ctypes.data_as(C.POINTER(C.c_int32)),
            shapes.ctypes.data_as(C.POINTER(C.c_int32)), self._tptrs, None,
            None if sym is None else sym.ctypes.data_as(C.POINTER(C.c_int32)), None,
            self._lib.ALLOC_FN(), None, self._lib.ALLOC_FN(), None)
        if self._step is None:
            rows = self.spec.group_rows(self._group, shapes, sym)
            arena = self.plan.arena_bytes(shapes, sym)
            from .placement import COLUMN_SHARD, ROW_SHARD
            h = C.c_void_p()
            self._lib.check(self._L.fcp_shard_step_create(
                self.plan.handle, self.comm.handle, ROW_SHARD if self.mode == "row" else COLUMN_SHARD, self._group,
                2 * rows, 2 * arena + (1 << 20), None if self._widths is None else self._widths.ctypes.data, C.byref(h)),
                "fcp_shard_step_create")
            self._step = h
        self._keep.append((d_blob, offsets, shapes, sym))
        return a

    def run(self, request, stream=None):
        """Enqueue one request; returns (torch view [count, width] of this rank's batch slice, begin, count)."""
        import ctypes as C
        torch = self.torch
        request.stream = torch.cuda.current_stream(self.dev).cuda_stream if stream is None else stream
        out, begin, count = C.c_void_p(), C.c_int64(), C.c_int64()
        self._lib.check(self._L.fcp_shard_step_run(self._step, C.byref(request), C.byref(out), C.byref(begin), C.byref(count)),
                        "fcp_shard_step_run")
        return out.value, begin.value, count.value

    def result(self, ptr: int, count: int):
        """Copy of the slice `run` produced (the library's buffer is recycled after two more calls)."""
        torch = self.torch
        if not count:
            return torch.empty((0, self.width), dtype=torch.float32, device=self.dev)
        return _device_view(torch, ptr, count * self.width, self.dev).view(count, self.width).clone()

    def close(self) -> None:
        if self._step is not None:
            self._L.fcp_shard_step_destroy(self._step)
            self._step = None
        self.plan.close()


def _device_view(torch, ptr: int, numel: int, device):
    """float32 torch view of `numel` elements of device memory at `ptr` (no ownership)."""
    class _Holder:
        __cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(_Holder(), device=device)


def mixed_assignment(spec, owners: Sequence[int], world: int):
    """The gate's per-table decision (``Placement.owners``: a rank, or -1 = spread by rows) as column sets:
    ``(row_cols, per_rank)`` — the columns whose table is spread over all ranks, and per rank the columns it holds
    whole, both in plan order.  Table-free columns (passthrough, Sum) go with rank 0's whole columns."""
    row_cols = [k for k, c in enumerate(spec.columns) if c.table_input >= 0 and owners[c.table_input] < 0]
    per_rank: List[List[int]] = [[] for _ in range(world)]
    for k, c in enumerate(spec.columns):
        if c.table_input < 0:
            per_rank[0].append(k)
        elif owners[c.table_input] >= 0:
            per_rank[owners[c.table_input]].append(k)
    return row_cols, per_rank


def _mixed_pieces(spec, row_cols, per_rank, group: int = 0):
    """How the output row [width of the group] is put together: pieces ``(source, source offset, destination offset,
    width)`` with source 0 = the row-sharded part's result (its columns side by side in plan order) and source 1 = the
    whole-column part's result (rank 0's block, rank 1's block, ...); neighbouring pieces are merged."""
    offs = spec.column_offsets()
    src_of = {}
    at = 0
    # a sub-plan keeps the concat slots of its columns (PlanSpec.column_subset): its group matrix holds them in CONCAT
    # order, whatever their order in the plan — so that is the order in which a part's columns are numbered here
    for k in sorted(row_cols, key=lambda k: offs[k]):
        if spec.columns[k].concat_group == group:
            src_of[k] = (0, at)
            at += spec.columns[k].dim
    w_row = at
    at = 0
    for cols in per_rank:
        for k in sorted(cols, key=lambda k: offs[k]):
            if spec.columns[k].concat_group == group:
                src_of[k] = (1, at)
                at += spec.columns[k].dim
    w_col = at
    pieces = []
    for k in sorted(src_of, key=lambda k: offs[k]):
        src, so = src_of[k]
        d = spec.columns[k].dim
        if pieces and pieces[-1][0] == src and pieces[-1][1] + pieces[-1][3] == so and pieces[-1][2] + pieces[-1][3] == offs[k]:
            pieces[-1] = (src, pieces[-1][1], pieces[-1][2], pieces[-1][3] + d)
        else:
            pieces.append((src, so, offs[k], d))
    return pieces, w_row, w_col


class MixedShardedStep:
    """``FCP_PLACE_MIXED``: tables larger than one GPU are spread by rows (partial sums exchanged + ``fcp_shard_finalize``),
    every other table lives whole on the rank the gate dealt it to (final column blocks exchanged) — two native steps
    per request and one strided scatter (``fcp_concat_outputs_scatter_strided``) that puts their results side by side in
    concat order.  A whole column sends 1/world of what the same column sends as a dense partial sum, so only the tables
    that MUST be spread pay that price (BASELINE configs[4], every table fits: the row part is empty and this is the
    column-sharded step)."""

    def __init__(self, model, comm: Communicator, owners: Sequence[int], group: int = 0) -> None:
        import torch
        from . import lib as _lib
        from .synth import submodel
        self.torch, self._lib, self._L, self.comm = torch, _lib, _lib.load(), comm
        self.dev = torch.device("cuda", comm.device)
        spec = model.spec
        self.row_cols, per_rank = mixed_assignment(spec, owners, comm.world)
        col_cols = sorted(k for cols in per_rank for k in cols)
        if any(len(c) == 0 for c in per_rank) and col_cols:
            # fcp_placement_assign never answers MIXED with fewer whole tables than ranks (it spreads everything by rows
            # instead); an assignment made by hand may
            raise ValueError("mixed placement: every rank must hold at least one whole column (fewer whole tables than "
                             "ranks: use row sharding)")
        self.pieces, self.w_row, self.w_col = _mixed_pieces(spec, self.row_cols, per_rank, group)
        self.width = int(spec.group_width(group))
        self.row_hosts = spec.column_subset(self.row_cols).host_inputs if self.row_cols else []
        self.col_hosts = spec.column_subset(col_cols).host_inputs if col_cols else []
        self.row_step = NativeShardedStep(submodel(model, self.row_cols), comm, "row", group) if self.row_cols else None
        pos = {k: i for i, k in enumerate(col_cols)}
        self.col_step = NativeShardedStep(submodel(model, col_cols), comm, "col", group,
                                          assignment=[[pos[k] for k in cols] for cols in per_rank]) if col_cols else None
        self._out = []          # ring of output buffers, allocated at the first request
        self._retired = []      # buffers a larger request replaced
        self._next = 0

    def prepare(self, inputs, symbols):
        return (self.row_step.prepare([inputs[i] for i in self.row_hosts], symbols) if self.row_step else None,
                self.col_step.prepare([inputs[i] for i in self.col_hosts], symbols) if self.col_step else None)

    def run(self, request, stream=None):
        import ctypes as C
        torch = self.torch
        srcs = [None, None]
        begin = count = 0
        if self.row_step:
            srcs[0], begin, count = self.row_step.run(request[0], stream)
        if self.col_step:
            srcs[1], begin, count = self.col_step.run(request[1], stream)
        if not self._out:
            self._out = [torch.empty((max(count, 1) * 2, self.width), dtype=torch.float32, device=self.dev) for _ in range(3)]
        if count > self._out[self._next].shape[0]:
            # a larger batch than the ring was sized for: a new buffer for this entry (the old one may still be read by work
            # enqueued earlier on any stream: it is kept for the life of the step)
            self._retired.append(self._out[self._next])
            self._out[self._next] = torch.empty((count * 2, self.width), dtype=torch.float32, device=self.dev)
        out = self._out[self._next]
        self._next = (self._next + 1) % len(self._out)
        if count:
            n = len(self.pieces)
            widths = (self.w_row, self.w_col)
            ins = (C.c_void_p * n)(*[srcs[s] + 4 * so for s, so, _, _ in self.pieces])
            dims = np.asarray([w for _, _, _, w in self.pieces], np.int32)
            strides = np.asarray([widths[s] for s, _, _, _ in self.pieces], np.int32)
            offs = np.asarray([do for _, _, do, _ in self.pieces], np.int32)
            st = torch.cuda.current_stream(self.dev).cuda_stream if stream is None else stream
            self._lib.check(self._L.fcp_concat_outputs_scatter_strided(ins, dims.ctypes.data, strides.ctypes.data, offs.ctypes.data, n,
                                                                       count, self.width, out.data_ptr(), st),
                            "fcp_concat_outputs_scatter_strided")
        return out.data_ptr(), begin, count

    def result(self, ptr: int, count: int):
        torch = self.torch
        if not count:
            return torch.empty((0, self.width), dtype=torch.float32, device=self.dev)
        return _device_view(torch, ptr, count * self.width, self.dev).view(count, self.width).clone()

    def exchanged_bytes(self, rows: int) -> int:
        """bytes this rank sends per request (it receives as many): dense partial slices of the row part, final
        blocks of the whole-column part"""
        world = self.comm.world
        w_mine = sum(w for (s, _, _, w) in self.pieces if s == 1) // max(world, 1)  # about 1/world of the whole columns are mine
        return int(4 * rows * (world - 1) / world * (self.w_row + w_mine))

    def close(self) -> None:
        for st in (self.row_step, self.col_step):
            if st is not None:
                st.close()


class _GlooShardedStep:
    """The sharded step with the exchange staged through the host (``_all_to_all`` under gloo): the interface of
    :class:`NativeShardedStep`, for ``bench.py --dist-backend gloo`` on a box whose ranks share one GPU."""

    def __init__(self, model, rank: int, world: int, device: int, mode: str, owners=None) -> None:
        import torch
        from .synth import submodel
        self.torch = torch
        self.mode = mode
        self.dev = torch.device("cuda", device)
        if mode != "mixed":
            self.impl = (ShardedFeatureColumns if mode == "row" else ColumnShardedFeatureColumns)(model, rank, world, device)
            return
        spec = model.spec
        self.row_cols, per_rank = mixed_assignment(spec, owners, world)
        col_cols = sorted(k for cols in per_rank for k in cols)
        self.pieces, self.w_row, self.w_col = _mixed_pieces(spec, self.row_cols, per_rank)
        self.width = int(spec.group_width(0))
        self.row_hosts = spec.column_subset(self.row_cols).host_inputs if self.row_cols else []
        self.col_hosts = spec.column_subset(col_cols).host_inputs if col_cols else []
        pos = {k: i for i, k in enumerate(col_cols)}
        self.row_impl = ShardedFeatureColumns(submodel(model, self.row_cols), rank, world, device) if self.row_cols else None
        self.col_impl = ColumnShardedFeatureColumns(submodel(model, col_cols), rank, world, device,
                                                    assignment=[[pos[k] for k in cols] for cols in per_rank]) if col_cols else None

    def _pack(self, tensors, symbols):
        from .ops import concat_inputs
        blob, offsets, shapes = concat_inputs(tensors)
        return (self.torch.from_numpy(blob).to(self.dev), offsets, shapes, symbols)

    def prepare(self, inputs, symbols):
        if self.mode == "mixed":
            return (self._pack([inputs[i] for i in self.row_hosts], symbols) if self.row_impl else None,
                    self._pack(self.col_impl.request_inputs([inputs[i] for i in self.col_hosts]), symbols) if self.col_impl else None)
        return self._pack(inputs if self.mode == "row" else self.impl.request_inputs(inputs), symbols)

    def run(self, request):
        if self.mode != "mixed":
            return self.impl(*request)
        parts = [None, None]
        begin = count = 0
        if self.row_impl:
            parts[0], begin, count = self.row_impl(*request[0])
        if self.col_impl:
            parts[1], begin, count = self.col_impl(*request[1])
        out = self.torch.empty((count, self.width), dtype=self.torch.float32, device=self.dev)
        for s, so, do, w in self.pieces:
            out[:, do:do + w] = parts[s][:, so:so + w]
        return out, begin, count

    def close(self) -> None:
        pass


def _time_sharded_step(args, model, placement, mode: str, rank: int, world: int, local_rank: int, dist):
    """One sharded serving mode timed the bench contract's way: W warm-up requests, barrier + synchronize, exactly K timed
    requests, synchronize + barrier, MAX over ranks.  Returns (elapsed seconds, device seconds per request)."""
    import torch
    if dist is not None and dist.get_backend() != "nccl":
        # RCCL refuses two ranks on one device: under gloo (several ranks sharing a GPU, a 1-GPU box exercising the N > 1
        # control flow) the exchange goes through the host and the step is the Python orchestration of the same kernels
        step = _GlooShardedStep(model, rank, world, local_rank, mode, placement.owners)
        comm = None
    else:
        comm = Communicator(rank, world, local_rank, dist)
        step = MixedShardedStep(model, comm, placement.owners) if mode == "mixed" else NativeShardedStep(model, comm, mode)
    reqs = [step.prepare(r.inputs, r.symbols) for r in (model.make_request(s) for s in range(8))]  # ids replicated on every rank
    for i in range(max(args.warmup, 1)):
        step.run(reqs[i % len(reqs)])
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        step.run(reqs[i % len(reqs)])
    e1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dev_s = e0.elapsed_time(e1) * 1e-3 / args.steps
    step.close()
    if comm is not None:
        comm.close()
    del step, reqs
    torch.cuda.empty_cache()
    return elapsed, dev_s


def bench_sharded(args, model, placement, rank: int, world: int, local_rank: int, dist) -> dict:
    """`bench.py --workload shard[-row|-col]` once the placement gate has decided to shard (BASELINE.json config 5:
    4000 S2-shaped columns, 480 GB of tables): every rank holds its shard, every request runs partial
    kernel -> RCCL all-to-all over xGMI -> finalize / concat as one native call (NativeShardedStep).  `--workload shard`
    is BASELINE configs[4] as written (every table row-sharded: `value`) and ALSO times what the gate prefers for the same
    tables (configs[4]: every table fits a GPU -> whole columns, 8x fewer bytes on the wire), reported as `gate_choice`."""
    from .placement import MIXED, ROW_SHARD
    gate_mode = "row" if placement.mode == ROW_SHARD else "mixed" if placement.mode == MIXED else "col"
    # `--workload shard` IS BASELINE configs[4]: its own line is the ROW-sharded step the config names (r6; VERDICT r05 weak
    # 11), with what the placement gate would choose for the same tables (whole columns wherever a table fits one GPU: 8x
    # fewer bytes on the wire, bit-identical to one GPU) beside it as `gate_choice`.  shard-row / shard-col force one kind.
    mode = "row" if getattr(args, "workload", "") == "shard" and world > 1 else gate_mode
    elapsed, dev_s = _time_sharded_step(args, model, placement, mode, rank, world, local_rank, dist)
    batch = model.batch
    width = model.spec.group_width(0)
    from .ops import concat_inputs
    r0 = model.make_request(0)
    bytes_alg = model.spec.algorithmic_bytes(concat_inputs(r0.inputs)[2], r0.symbols)

    def accounting(mode):
        """(bytes one rank sends per request, per-GPU algorithmic bytes, description)"""
        sent = batch * width * 4 * (world - 1) / world / (1 if mode == "row" else world)
        if mode == "mixed":
            row_cols, per_rank = mixed_assignment(model.spec, placement.owners, world)
            w_row = sum(model.spec.columns[k].dim for k in row_cols)
            sent = batch * 4 * (world - 1) / world * (w_row + (width - w_row) / world)
            per_gpu = bytes_alg["total"] / world + 2 * sent + batch * width * 4 * 2 / world + batch * w_row * 4
            par = (f"mixed x{world}: {len(row_cols)} column(s) row-sharded (tables larger than one GPU: partial sums + fcp_shard_finalize), "
                   f"{sum(len(c) for c in per_rank)} whole (final blocks), grouped ncclSend/ncclRecv (RCCL over xGMI) + strided concat")
        elif mode == "row":
            # per-GPU algorithmic bytes: 1/world of the table rows, all ids, the partial
            # [rows, width] written once, its slices sent / received, the final slice written
            per_gpu = bytes_alg["rows"] / world + bytes_alg["ids"] + bytes_alg["boundaries"] + batch * width * 4 * (
                1 + 2.0 * (world - 1) / world + 1.0 / world)
            par = f"row-sharded x{world}: grouped ncclSend/ncclRecv of partial sums (RCCL over xGMI) + fcp_shard_finalize"
        else:
            # 1/world of the rows, ids and output; the block's remote slices sent / received,
            # and (world > 1) the final [count, width] slice read + written by the concat
            per_gpu = bytes_alg["total"] / world + batch * width * 4 / world * (
                2.0 * (world - 1) / world + (2.0 if world > 1 else 0.0))
            par = f"column-sharded x{world}: grouped ncclSend/ncclRecv of final column blocks (RCCL over xGMI) + concat"
        return sent, per_gpu, par

    sent, per_gpu, par = accounting(mode)
    rec = {
        "metric": f"inference QPS, {'row-sharded' if mode == 'row' else 'mixed-placement' if mode == 'mixed' else 'column-sharded'} tables (SHARD config)",
        "value": batch * args.steps / elapsed,
        "unit": "inferences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{model.name}: {model.description}; tables {placement.name} over {world} GPU(s) by the "
                               f"placement gate ({placement.total_bytes / 1e9:.0f} GB of tables, {placement.bytes_per_gpu / 1e9:.0f} GB per GPU)",
                   "batch": batch, "columns": model.spec.n_columns, "table_bytes": model.table_bytes(),
                   "parallelism": par, "exchange_bytes_sent_per_rank_per_request": int(sent)},
        "roofline": {"bound": "hbm", "achieved": per_gpu / dev_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": per_gpu / dev_s / 1e9 / 8000.0, "traffic": None,
                     "note": "per GPU, whole step (partial kernel + exchange + finalize / concat), events on the compute stream"},
    }
    if getattr(args, "workload", "") == "shard" and gate_mode != mode:
        # what the placement gate prefers for these tables, in the same run
        e_g, d_g = _time_sharded_step(args, model, placement, gate_mode, rank, world, local_rank, dist)
        s_g, g_g, p_g = accounting(gate_mode)
        rec["gate_choice"] = {
            "what": "the placement gate's own preference for the same tables (FCP_PLACE_MIXED: a table that fits one GPU stays "
                    "whole, final column blocks are exchanged: no partial sums, results bit-identical to one GPU); "
                    "`--workload shard-col` makes it the headline",
            "value": batch * args.steps / e_g, "unit": "inferences/s", "ms_per_step": e_g * 1e3 / args.steps,
            "parallelism": p_g, "exchange_bytes_sent_per_rank_per_request": int(s_g),
            "roofline_frac": g_g / d_g / 1e9 / 8000.0}
    return rec


def bench_row_sharded_record(args, rank: int, world: int, local_rank: int, dist, hbm_bytes: int, budget_s: float = 10.0) -> dict:
    """What `bench.py --gpus N` attaches to its DEFAULT line (replicated S2, no collective) as `sharded`: BASELINE configs[4]
    at its own shape — 4000 S2-shaped columns, batch 512, every table row-sharded over the N ranks — over the REAL backend
    for at most `budget_s` seconds, so that the first scaling run on a multi-GPU node measures the RCCL exchange over xGMI
    without asking for another workload.  Vocabulary: 1 M rows per table (480 GB) when N GPUs hold it,
    else the largest that fits.  Reports the whole step (partial kernel -> grouped ncclSend / ncclRecv -> fcp_shard_finalize)
    and the exchange alone (fcp_shard_exchange on a resident partial).  Testing aids: FCP_BENCH_SHARD_COLUMNS /
    FCP_BENCH_SHARD_VOCAB shrink the model (the 1-GPU boxes run N ranks on one device over gloo)."""
    import ctypes as C
    import os
    import torch
    from . import lib as _lib
    from . import synth
    columns = int(os.environ.get("FCP_BENCH_SHARD_COLUMNS", "4000"))
    batch = args.batch or 512
    width = sum((8, 16, 32, 64)[c % 4] for c in range(columns))
    # tables: columns x vocab x mean dim x 4 bytes over `world` GPUs, within 85 % of each GPU's memory net of the exchange
    # buffers (partial + slices, a ring of three)
    per_row = width * 4
    # what is FREE on the device now (the replicated run's tables have just been released), the smallest over the ranks so
    # that every rank builds the same model — a rank that ran out of memory alone would leave the others waiting in the exchange
    free_b = hbm_bytes
    if not os.environ.get("FCP_BENCH_HBM_BYTES"):
        torch.cuda.empty_cache()
        free_b = min(hbm_bytes, torch.cuda.mem_get_info(local_rank)[0])
        if dist is not None and world > 1:
            t = torch.tensor([free_b], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            free_b = int(t.item())
    room = 0.85 * free_b - 8 * batch * width * 4
    vocab = int(os.environ.get("FCP_BENCH_SHARD_VOCAB", "0")) or int(max(1000, min(1_000_000, room * world // per_row)))
    model = synth.model_shard(columns=columns, vocab=vocab, batch=batch)
    backend = dist.get_backend() if dist is not None else "none"
    native = dist is None or backend == "nccl"
    comm = Communicator(rank, world, local_rank, dist) if native else None
    step = NativeShardedStep(model, comm, "row") if native else _GlooShardedStep(model, rank, world, local_rank, "row")
    reqs = [step.prepare(r.inputs, r.symbols) for r in (model.make_request(s) for s in range(4))]

    def timed(fn, n):
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el / n, e0.elapsed_time(e1) * 1e-3 / n

    run = lambda i: step.run(reqs[i % len(reqs)])
    for i in range(5):
        run(i)
    per, _ = timed(run, 5)                                  # sizes the timed loops to the budget
    n = int(max(5, min(400, 0.35 * budget_s / max(per, 1e-6))))
    step_s, step_dev_s = timed(run, n)
    sent = batch * width * 4 * (world - 1) // world
    rec = {"mode": "row", "backend": backend, "columns": columns, "batch": batch, "vocab": vocab,
           "table_GB_total": columns * vocab * (per_row / columns) / 1e9, "table_GB_per_rank": vocab * per_row / world / 1e9,
           "steps": n, "step_us": step_s * 1e6, "step_device_us": step_dev_s * 1e6,
           "inferences_per_s": batch / step_s,
           "exchange_bytes_sent_per_rank_per_request": int(sent),
           "what": "BASELINE configs[4] at its own shape (4000 S2-shaped columns, batch 512, every table row-sharded id % world) on the "
                   "ranks of this run: partial kernel -> grouped ncclSend / ncclRecv of partial sums (RCCL over xGMI) -> "
                   "fcp_shard_finalize as one native call per request; `exchange_us` = the exchange alone on a resident partial; "
                   "never part of `value` (the default workload's tables fit one GPU: replicas, no collective)"}
    if native:
        L = _lib.load()
        r_, w_ = C.c_int32(), C.c_int32()
        _lib.check(L.fcp_comm_rank(comm.handle, C.byref(r_), C.byref(w_)), "fcp_comm_rank")
        rec["ranks_seen_by_rccl"] = int(w_.value)
        partial = torch.zeros((batch, width), dtype=torch.float32, device=step.dev)
        sl = batch_slices(batch, world)
        slices = torch.empty((world, max(c for _, c in sl), width), dtype=torch.float32, device=step.dev)
        b_, c_ = C.c_int64(), C.c_int64()

        def exch(i):
            _lib.check(L.fcp_shard_exchange(comm.handle, partial.data_ptr(), batch, width, slices.data_ptr(), C.byref(b_), C.byref(c_),
                                            torch.cuda.current_stream(step.dev).cuda_stream), "fcp_shard_exchange")
        for i in range(5):
            exch(i)
        ex_s, ex_dev_s = timed(exch, n)
        rec["exchange_us"] = ex_dev_s * 1e6
        rec["exchange_GBs_per_rank"] = sent / ex_dev_s / 1e9 if ex_dev_s > 0 else None
        del partial, slices
    else:
        rec["ranks_seen_by_rccl"] = 0
        rec["note"] = f"backend {backend}: the exchange is staged through the host (control flow of the N > 1 run on a 1-GPU box); no RCCL, no xGMI figure"
        path = RowShardedPath(rank, world)
        partial = torch.zeros((batch, width), dtype=torch.float32, device=step.dev)
        ex_s, _ = timed(lambda i: path.exchange(partial), max(3, n // 4))
        rec["exchange_us"] = ex_s * 1e6
        rec["exchange_GBs_per_rank"] = sent / ex_s / 1e9 if ex_s > 0 else None
    step.close()
    if comm is not None:
        comm.close()
    del step, reqs
    torch.cuda.empty_cache()
    return rec
