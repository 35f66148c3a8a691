"""Column plan: the static description of one model's fused feature-column stage.

This is the data the reference carries as *generated CUDA text* — one
``struct FCi`` per feature column, built by ``CudaEmitter::EmitFCCode``
(``graph_optimizers/cuda_emitter.cc:975-1178``) and assembled into ``KnlArgs`` /
``FusedKnl`` (``:2057-2137``).  Here it is a plain table of
:class:`ColumnSpec` records that pre-compiled gfx950 kernels interpret, so no
code generation or runtime compiler is needed.

The enums mirror ``include/fcp_hip.h`` one-to-one.  Pure Python, no GPU needed.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# --- enums (== include/fcp_hip.h) -------------------------------------------
FORM_GATHER = 1            # GatherV2(table, ids)              cuda_emitter.cc:250-293
FORM_SEGMENT_REDUCE = 2    # SparseSegment{Sum,Mean}WithNumSegments  :402-661, :768-962
FORM_GATHER_SCATTER = 3    # ScatterNd(rows, GatherV2(...))    :296-345
FORM_PASSTHROUGH = 4       # ConcatOutputs host_inputs         concat_outputs_op_gpu.cu.cc:186-216
FORM_BATCH_COL_REDUCTION = 5  # Sum(x, axis=1)                 cuda_emitter.cc:1180-1244
FORM_EXTERNAL = 6          # a concat slot filled by ConcatOutputs `host_inputs` (fcp_concat_outputs_host)

COMBINER_NONE, COMBINER_SUM, COMBINER_MEAN = 0, 1, 2
IDS_I32, IDS_I64, IDS_F32_BUCKETIZE = 0, 1, 2
SEG_NONE, SEG_IDS_I32, SEG_IDS_I64, SEG_CSR_I32 = 0, 1, 2, 3
ROWS_FROM_IDS, ROWS_FROM_SYMBOL, ROWS_FROM_INPUT_DIM0, ROWS_FROM_GROUP = 0, 1, 2, 3
STAGE_COPY, STAGE_NARROW_I64, STAGE_SEG_TO_CSR = 0, 1, 2  # fcp_stager_stage_ex modes
XFORM_NONE, XFORM_SELECT, XFORM_FILTER = 0, 1, 2   # id transforms (SelectValue / GatherIndiceValue family)
LAYOUT_CONCAT, LAYOUT_PER_COLUMN = 0, 1
FLAG_COUNT_BAD_IDS = 1

_ID_ELEM_SIZE = {IDS_I32: 4, IDS_I64: 8, IDS_F32_BUCKETIZE: 4}
_ID_NP_DTYPE = {IDS_I32: np.int32, IDS_I64: np.int64, IDS_F32_BUCKETIZE: np.float32}


@dataclass
class ColumnSpec:
    """One feature column (one FC subgraph of ``graph_info.cc:209-365``)."""

    form: int
    dim: int
    vocab: int = 0
    combiner: int = COMBINER_NONE
    id_source: int = IDS_I64
    table_input: int = -1
    ids_input: int = -1
    seg_input: int = -1
    seg_kind: int = SEG_NONE
    seg_stride: int = 1
    rows_source: int = ROWS_FROM_IDS
    rows_arg: int = 0
    boundaries: Optional[np.ndarray] = None
    concat_group: int = 0
    concat_slot: int = 0
    # id transform: closed integer intervals; SELECT substitutes ids outside them,
    # FILTER drops them (custom_ops/select_value, gather_indice_value, gather_value_gen_indice)
    xform_mode: int = XFORM_NONE
    xform_lo: Sequence[int] = ()
    xform_hi: Sequence[int] = ()
    xform_substitute: int = 0
    # > 0: integer ids are hashed into buckets first — Fingerprint64(decimal string) % hash_buckets, TensorFlow's
    # AsString -> StringToHashBucketFast (categorical_column_with_hash_bucket over integer features)
    hash_buckets: int = 0
    # Segment ids that are a function of several index coordinates: a SparseReshape between the SparseTensor and the
    # lookup, folded into the index expression as the reference does (EmitInputInline, cuda_emitter.cc:1874-1916):
    #   seg(i) = (sum_k idx[i*seg_stride + k] * seg_mul[k]) // seg_div    over the first len(seg_mul) coordinates;
    # one factor (seg_sym_slot: 0..3 = seg_mul[slot], 4 = seg_div) is multiplied by symbol seg_sym when seg_sym >= 0.
    # () = plain segment ids, idx[i*seg_stride].  (fcp_column_ext_t in include/fcp_hip.h)
    seg_mul: Sequence[int] = ()
    seg_div: int = 1
    seg_sym: int = -1
    seg_sym_slot: int = 0

    def validate(self) -> None:
        if self.form not in (1, 2, 3, 4, 5, 6):
            raise ValueError(f"bad form {self.form}")
        if self.dim <= 0:
            raise ValueError("dim must be positive")
        if (self.form == FORM_EXTERNAL) != (self.rows_source == ROWS_FROM_GROUP):
            raise ValueError("external slots (and only they) take their row count from their concat group")
        if self.form in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
            if self.vocab <= 0 or self.table_input < 0 or self.ids_input < 0:
                raise ValueError("lookup column needs vocab, table_input, ids_input")
            if self.id_source == IDS_F32_BUCKETIZE:
                if self.boundaries is None or len(self.boundaries) == 0:
                    raise ValueError("bucketize column needs boundaries")
        if self.form in (FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
            if self.seg_kind == SEG_NONE or self.seg_input < 0:
                raise ValueError("pooled/scatter column needs segment input")
            if self.rows_source == ROWS_FROM_IDS:
                raise ValueError("pooled/scatter column needs an explicit row count source")
        if self.form == FORM_SEGMENT_REDUCE and self.combiner not in (COMBINER_SUM, COMBINER_MEAN):
            raise ValueError("segment-reduce column needs sum or mean")
        if len(self.seg_mul):
            if self.form != FORM_SEGMENT_REDUCE or self.seg_kind not in (SEG_IDS_I32, SEG_IDS_I64):
                raise ValueError("a segment-id map needs a pooled column with segment ids")
            if len(self.seg_mul) > 4 or self.seg_stride < len(self.seg_mul) or self.seg_div < 1 or \
                    any(v < 0 for v in self.seg_mul):
                raise ValueError("bad segment-id map")
            if self.seg_sym >= 0 and not (self.seg_sym_slot == 4 or 0 <= self.seg_sym_slot < len(self.seg_mul)):
                raise ValueError("bad segment-id map symbol slot")
        if self.hash_buckets:
            if self.form not in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER) or self.hash_buckets < 0 or \
                    self.id_source == IDS_F32_BUCKETIZE:
                raise ValueError("hash_buckets applies to the integer ids of lookup columns")
        if self.xform_mode != XFORM_NONE:
            if self.form not in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
                raise ValueError("id transforms apply to lookup columns only")
            if self.xform_mode not in (XFORM_SELECT, XFORM_FILTER) or len(self.xform_lo) != len(self.xform_hi):
                raise ValueError("bad id transform")
            if any(lo > hi for lo, hi in zip(self.xform_lo, self.xform_hi)):
                raise ValueError("empty id transform interval")


@dataclass
class StageInfo:
    """The stage section of a plan file: what ``Addons>ConcatInputs`` does to each of its inputs while it packs
    (``fcp_concat_inputs_ex``; ``fcp_plan_file_stage_info``)."""
    modes: List[int]            # STAGE_* per ConcatInputs input
    rows_symbol: List[int]      # STAGE_SEG_TO_CSR inputs: index into the symbols vector of the row count, else -1
    symbols_input: int = -1     # which ConcatInputs input is the symbols vector (int32[n_symbols]), or -1

    def mode_args(self, inputs) -> List[int]:
        sym = None if self.symbols_input < 0 else [int(v) for v in inputs[self.symbols_input].reshape(-1)]
        return [sym[k] if m == STAGE_SEG_TO_CSR else 0 for m, k in zip(self.modes, self.rows_symbol)]


@dataclass
class PlanSpec:
    """All columns of a model + the ConcatInputs / FeatureColumnProcess attrs.

    ``host_input_ranks`` / ``host_input_elem_sizes`` are the ``ranks`` / ``T``
    attrs of ``Addons>ConcatInputs`` (``concat_inputs_ops.cc:82-88``);
    ``n_device_inputs`` is the length of ``input_types`` of
    ``Addons>FeatureColumnProcess`` (the embedding tables,
    ``feature_column_process_op_gpu.cu.cc:133-152``).
    """

    columns: List[ColumnSpec]
    host_input_ranks: List[int]
    host_input_elem_sizes: List[int]
    n_device_inputs: int
    n_groups: int = 1
    n_symbols: int = 0
    layout: int = LAYOUT_CONCAT
    shard_rank: int = 0
    shard_world: int = 1
    flags: int = 0

    # ---- static layout facts ------------------------------------------------
    def validate(self) -> None:
        if len(self.host_input_ranks) != len(self.host_input_elem_sizes):
            raise ValueError("host_input_ranks / elem_sizes length mismatch")
        seen = set()
        for k, c in enumerate(self.columns):
            c.validate()
            if not 0 <= c.concat_group < self.n_groups:
                raise ValueError(f"column {k}: concat_group out of range")
            key = (c.concat_group, c.concat_slot)
            if key in seen:
                raise ValueError(f"column {k}: duplicate concat slot {key}")
            seen.add(key)
            for idx in (c.ids_input, c.seg_input):
                if idx >= len(self.host_input_ranks):
                    raise ValueError(f"column {k}: host input index out of range")
            if c.table_input >= self.n_device_inputs:
                raise ValueError(f"column {k}: table_input out of range")
            if c.rows_source == ROWS_FROM_SYMBOL and not 0 <= c.rows_arg < self.n_symbols:
                raise ValueError(f"column {k}: symbol index out of range")

    @property
    def n_columns(self) -> int:
        return len(self.columns)

    @property
    def n_host_inputs(self) -> int:
        return len(self.host_input_ranks)

    def output_columns(self) -> List[int]:
        """Plan columns that are outputs of ``Addons>FeatureColumnProcess`` (all but external slots)."""
        return [k for k, c in enumerate(self.columns) if c.form != FORM_EXTERNAL]

    def group_width(self, group: int) -> int:
        return sum(c.dim for c in self.columns if c.concat_group == group)

    def column_offsets(self) -> List[int]:
        """Element offset of each column inside its concat group (prefix sums of
        ``embedd_dims`` in slot order, ``concat_outputs_op_gpu.cu.cc:74-79``)."""
        offs = [0] * len(self.columns)
        for g in range(self.n_groups):
            members = sorted((c.concat_slot, k) for k, c in enumerate(self.columns) if c.concat_group == g)
            acc = 0
            for _, k in members:
                offs[k] = acc
                acc += self.columns[k].dim
        return offs

    def shape_offsets(self) -> List[int]:
        out, acc = [], 0
        for r in self.host_input_ranks:
            out.append(acc)
            acc += r
        return out

    def to_dict(self) -> dict:
        """Plain-dict form (what the oracle wrapper consumes; keeps oracle/ free of
        product imports)."""
        d = dataclasses.asdict(self)
        for c, src in zip(d["columns"], self.columns):
            c["boundaries"] = None if src.boundaries is None else np.asarray(src.boundaries, np.float32)
            c["xform_lo"] = [int(v) for v in src.xform_lo]
            c["xform_hi"] = [int(v) for v in src.xform_hi]
            c["seg_mul"] = [int(v) for v in src.seg_mul]
        return d

    def with_shard(self, rank: int, world: int) -> "PlanSpec":
        return dataclasses.replace(self, shard_rank=rank, shard_world=world)

    def with_layout(self, layout: int) -> "PlanSpec":
        return dataclasses.replace(self, layout=layout)

    def narrowed(self) -> "tuple[PlanSpec, List[bool]]":
        """Plan for blobs staged with ``fcp_stager_stage_narrow``: every int64 id / segment-id
        input is declared int32 (half the PCIe bytes; ids and rows are < 2^31).  Returns
        the plan and the per-host-input narrow flags for the stager."""
        flags = [False] * self.n_host_inputs
        for c in self.columns:
            if c.form in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
                if c.id_source == IDS_I64 and c.vocab <= 0x7fffffff:
                    flags[c.ids_input] = True
                if c.seg_kind == SEG_IDS_I64:
                    flags[c.seg_input] = True
        # an input stays 8-byte unless every column reading it agrees; raw ids that are hashed or run through an
        # interval transform on the device keep their full width (the stager maps ids outside [0, 2^31) to -1,
        # which only the plain vocabulary check treats the same way)
        for c in self.columns:
            if c.form in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
                if c.id_source != IDS_I64 or c.vocab > 0x7fffffff or c.hash_buckets or c.xform_mode != XFORM_NONE:
                    if c.ids_input >= 0 and self.host_input_elem_sizes[c.ids_input] == 8:
                        flags[c.ids_input] = False
        cols = []
        for c in self.columns:
            r = {}
            if c.ids_input >= 0 and flags[c.ids_input] and c.id_source == IDS_I64:
                r["id_source"] = IDS_I32
            if c.seg_input >= 0 and flags[c.seg_input] and c.seg_kind == SEG_IDS_I64:
                r["seg_kind"] = SEG_IDS_I32
            cols.append(dataclasses.replace(c, **r) if r else c)
        sizes = [4 if f else e for f, e in zip(flags, self.host_input_elem_sizes)]
        return dataclasses.replace(self, columns=cols, host_input_elem_sizes=sizes), flags

    def staged(self, narrow: bool = True, csr: bool = True) -> "tuple[PlanSpec, List[int], List[int]]":
        """Plan for requests that go through ``fcp_stager_stage_ex``: int64 ids shipped as int32 (``narrowed()``) and the
        sorted row ids / SparseTensor indices of multi-hot columns turned into int32 CSR offsets by the host while it packs
        (``STAGE_SEG_TO_CSR``) — the device then needs neither the segment-offset pre-pass nor the in-block search, and the
        blob carries 4 bytes per ROW instead of 8-16 bytes per id for them.  Returns ``(plan, modes, rows_source)``:
        ``modes[i]`` is the stager mode of host input ``i``; ``rows_source[i]`` is, for converted inputs, the index of the
        column whose row count the stager needs (``rows_of_inputs`` evaluates it for a request), else -1."""
        spec, flags = self.narrowed() if narrow else (self, [False] * self.n_host_inputs)
        modes = [STAGE_NARROW_I64 if f else STAGE_COPY for f in flags]
        rows_col = [-1] * self.n_host_inputs
        if csr:
            users: dict = {}
            for k, c in enumerate(spec.columns):
                # pooled columns only: their segment ids are sorted (TF's SparseSegment* contract); a ScatterNd column
                # takes its row ids in any order (cuda_emitter.cc:296-345) and keeps them as they are
                # (a column whose segment ids are computed from several coordinates keeps its index matrix: the
                # pre-pass evaluates the map on the device)
                if c.form == FORM_SEGMENT_REDUCE and c.seg_kind in (SEG_IDS_I32, SEG_IDS_I64) and not len(c.seg_mul):
                    users.setdefault(c.seg_input, []).append(k)
            for i, ks in users.items():
                # an input is converted only if every reader is such a column with the same stride and the same row count
                # source, and nothing reads it as ids or as a passthrough payload
                c0 = self.columns[ks[0]]
                same = all((self.columns[k].seg_stride, self.columns[k].rows_source, self.columns[k].rows_arg) ==
                           (c0.seg_stride, c0.rows_source, c0.rows_arg) for k in ks)
                other = any(c.ids_input == i or (c.rows_source == ROWS_FROM_INPUT_DIM0 and c.rows_arg == i) for c in self.columns)
                readers = sum(1 for c in self.columns if c.seg_input == i)
                if same and not other and readers == len(ks) and c0.rows_source == ROWS_FROM_SYMBOL:
                    modes[i] = STAGE_SEG_TO_CSR
                    rows_col[i] = ks[0]
            cols = [dataclasses.replace(c, seg_kind=SEG_CSR_I32, seg_stride=1)
                    if c.seg_input >= 0 and modes[c.seg_input] == STAGE_SEG_TO_CSR else c for c in spec.columns]
            ranks = [1 if m == STAGE_SEG_TO_CSR else r for m, r in zip(modes, spec.host_input_ranks)]
            sizes = [4 if m == STAGE_SEG_TO_CSR else e for m, e in zip(modes, spec.host_input_elem_sizes)]
            spec = dataclasses.replace(spec, columns=cols, host_input_ranks=ranks, host_input_elem_sizes=sizes)
        spec.validate()
        return spec, modes, rows_col

    def staged_for_concat_inputs(self) -> "tuple[PlanSpec, StageInfo]":
        """The staged plan as ``Addons>ConcatInputs`` produces its blob in a rewritten graph (``python -m recom_amd.graph
        --staged``): ``staged()`` plus — when some input is converted to row offsets — the ``symbols`` vector as one more
        (last) ConcatInputs input, because the op needs the row counts and receives nothing but its input tensors.  The
        vector travels in the blob as an int32 tensor no column reads.  Returns the plan the kernels consume and the
        stage section of its plan file."""
        spec, modes, rows_col = self.staged()
        rows_symbol = [self.columns[k].rows_arg if k >= 0 else -1 for k in rows_col]
        symbols_input = -1
        if STAGE_SEG_TO_CSR in modes:
            symbols_input = spec.n_host_inputs
            spec = dataclasses.replace(spec, host_input_ranks=list(spec.host_input_ranks) + [1],
                                       host_input_elem_sizes=list(spec.host_input_elem_sizes) + [4])
            modes = modes + [STAGE_COPY]
            rows_symbol = rows_symbol + [-1]
        spec.validate()
        return spec, StageInfo(modes, rows_symbol, symbols_input)

    def column_subset(self, keep: Sequence[int]) -> "SubPlan":
        """Plan over the columns ``keep`` only (column-sharded serving: one such plan
        per GPU).  Host inputs and tables are renumbered to the ones those columns
        reference, in their original order; concat slots are kept, so the subset's
        group matrix is the matching column block of the full one."""
        keep = list(keep)
        host, dev = set(), set()
        for k in keep:
            c = self.columns[k]
            for i in (c.ids_input, c.seg_input):
                if i >= 0:
                    host.add(i)
            if c.rows_source == ROWS_FROM_INPUT_DIM0:
                host.add(c.rows_arg)
            if c.table_input >= 0:
                dev.add(c.table_input)
        host_l, dev_l = sorted(host), sorted(dev)
        hmap = {i: n for n, i in enumerate(host_l)}
        dmap = {i: n for n, i in enumerate(dev_l)}
        cols = []
        for k in keep:
            c = self.columns[k]
            cols.append(dataclasses.replace(
                c, ids_input=hmap.get(c.ids_input, -1), seg_input=hmap.get(c.seg_input, -1),
                table_input=dmap.get(c.table_input, -1),
                rows_arg=hmap[c.rows_arg] if c.rows_source == ROWS_FROM_INPUT_DIM0 else c.rows_arg))
        spec = dataclasses.replace(self, columns=cols, host_input_ranks=[self.host_input_ranks[i] for i in host_l],
                                   host_input_elem_sizes=[self.host_input_elem_sizes[i] for i in host_l],
                                   n_device_inputs=len(dev_l))
        return SubPlan(spec, keep, host_l, dev_l)

    # ---- roofline accounting --------------------------------
    def algorithmic_bytes(self, shapes: Sequence[int], symbols: Optional[Sequence[int]] = None) -> dict:
        """Algorithmic bytes of one request: table rows read + ids read + CSR
        offsets / segment ids read + bucketize boundaries + pooled output written
        once in concat layout.  No intermediate traffic is counted."""
        so = self.shape_offsets()

        def numel(i: int) -> int:
            n = 1
            for j in range(self.host_input_ranks[i]):
                n *= int(shapes[so[i] + j])
            return n

        rows_b = ids_b = seg_b = bnd_b = out_b = 0
        for c in self.columns:
            if c.form == FORM_EXTERNAL:      # written by fcp_concat_outputs_host, not by the fused kernel
                continue
            rows = self.column_rows(c, shapes, symbols)
            out_b += rows * c.dim * 4
            if c.form == FORM_PASSTHROUGH:
                rows_b += rows * c.dim * 4
            elif c.form == FORM_BATCH_COL_REDUCTION:
                rows_b += numel(c.ids_input) * 4
            else:
                nnz = numel(c.ids_input)
                rows_b += nnz * c.dim * 4
                ids_b += nnz * _ID_ELEM_SIZE[c.id_source]
                if c.seg_kind == SEG_CSR_I32:
                    seg_b += (rows + 1) * 4
                elif c.seg_kind == SEG_IDS_I32:
                    seg_b += nnz * 4
                elif c.seg_kind == SEG_IDS_I64:
                    seg_b += nnz * 8
                if c.id_source == IDS_F32_BUCKETIZE:
                    bnd_b += len(c.boundaries) * 4
        read = rows_b + ids_b + seg_b + bnd_b
        return {"rows": rows_b, "ids": ids_b, "segments": seg_b, "boundaries": bnd_b,
                "out": out_b, "read": read, "total": read + out_b}

    def column_rows(self, c: ColumnSpec, shapes: Sequence[int], symbols: Optional[Sequence[int]]) -> int:
        if c.rows_source == ROWS_FROM_GROUP:
            return self.group_rows(c.concat_group, shapes, symbols)
        so = self.shape_offsets()
        if c.rows_source == ROWS_FROM_IDS:
            n = 1
            for j in range(self.host_input_ranks[c.ids_input]):
                n *= int(shapes[so[c.ids_input] + j])
            return n
        if c.rows_source == ROWS_FROM_SYMBOL:
            if symbols is None:
                raise ValueError("plan needs symbols")
            return int(symbols[c.rows_arg])
        return int(shapes[so[c.rows_arg]])

    def group_rows(self, group: int, shapes: Sequence[int], symbols: Optional[Sequence[int]] = None) -> int:
        rows = None
        for c in self.columns:
            if c.concat_group != group or c.rows_source == ROWS_FROM_GROUP:
                continue
            r = self.column_rows(c, shapes, symbols)
            if rows is not None and r != rows:
                raise ValueError(f"group {group}: inconsistent row counts {rows} vs {r}")
            rows = r
        if rows is None:
            raise ValueError(f"group {group} has no columns")
        return rows


@dataclass
class SubPlan:
    """Result of :meth:`PlanSpec.column_subset`: the sub-plan and which columns /
    host inputs / tables of the full plan it uses (indices into the full plan)."""
    spec: PlanSpec
    columns: List[int]
    host_inputs: List[int]
    device_inputs: List[int]


def id_numpy_dtype(id_source: int):
    return _ID_NP_DTYPE[id_source]
