"""Plan builder: rewritten GraphDef → column plan.

The non-codegen half of the reference's ``CudaEmitter``: the same walk over every
feature-column subgraph, but each column becomes one :class:`ColumnSpec` record the
pre-compiled gfx950 kernels interpret, instead of a ``struct FCi`` of CUDA text.

Input: the GraphDef the reference's emitter sees, i.e. after ``PreLookupOptimizer``
/ ``LookupOptimizer`` brought every ``embedding_column`` to one of the canonical
forms (``lookup_optimizer.cc:157-440``):

* form 1 ``GatherV2(table, ids, 0)``                                  (``RewriteDenseInput`` ``:270-322``)
* form 2 ``SparseSegment{Sum,Mean}WithNumSegments(table, ids, seg, n)`` (``RewriteSeedWithNumSegments`` ``:157-268``)
* form 3 ``ScatterNd(rows, GatherV2(table, ids, 0), shape)``            (``RewriteGatherScatter`` ``:324-440``)
* ``Sum(x, axis=1)`` (BatchColReduction, ``cuda_emitter.cc:1146-1149``)

all converging on ``ConcatV2`` nodes (``cuda_emitter.cc:2543-2581``).

What follows the reference, and where:

* a table is a ``VariableV2`` / ``Const`` whose consumers (through ``Identity``) are
  only ``*Gather*`` / ``SparseSegment*`` ops — ``graph_info.cc:209-259``;
* the value node of a concat input is found through trailing ``Reshape`` /
  ``ExpandDims`` / ``Squeeze`` nodes — ``FindFCOutputs`` ``cuda_emitter.cc:1060-1069``;
* dispatch on the value node's op — ``EmitSubgraphCode`` ``:1096-1152``;
* index inputs are followed through ``Reshape``-likes, ``Cast``, ``Bucketize``, the
  ``[:, 0:1]`` ``StridedSlice`` of a 2-D index matrix and a ``SparseReshape`` that provably
  keeps ``[rows, k]`` (``:1874-1916`` restricted to that case), and end at the first other node,
  whose tensor becomes a ConcatInputs input — ``EmitInputInline`` ``:1769-1949``;
* tables become FeatureColumnProcess device inputs, deduplicated by tensor name —
  ``:1262-1279``;
* the three ops that replace the subgraphs are wired as ``Rewrite`` ``:2496-2656``
  does (:mod:`recom_amd.graph.rewrite`).

A concat input that is not a lookup (the reference's ``host_inputs`` of ConcatOutputs,
``cuda_emitter.cc:2594-2611``) is handled in one of two ways (``host_concat``):
``"external"`` keeps the reference's wiring — the plan reserves a ``FORM_EXTERNAL`` slot
and the rewritten graph feeds the tensor to ``Addons>ConcatOutputs`` as a host input,
exactly what the reference's own ``Rewrite`` emits; ``"passthrough"`` (default, one H2D
copy fewer) routes it through ConcatInputs as a ``FORM_PASSTHROUGH`` column so that the
fused kernel writes the whole concat matrix.  Other differences by design:
row counts that the reference derives with SymEngine are *symbols* evaluated by
ordinary TF ops in the rewritten graph (``symbols`` input of
``FeatureColumnProcessWithSymbols``, ``cuda_emitter.cc:2446-2458``); there is no
256 MiB table gate (``check_table_size`` ``:1080-1094``) — every table fits 288 GB.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from ..plan import (XFORM_FILTER, XFORM_NONE, XFORM_SELECT,
                    COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_BATCH_COL_REDUCTION, FORM_EXTERNAL, FORM_GATHER,
                    FORM_GATHER_SCATTER, FORM_PASSTHROUGH, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I32, IDS_I64,
                    ROWS_FROM_GROUP, ROWS_FROM_IDS, ROWS_FROM_INPUT_DIM0, ROWS_FROM_SYMBOL, SEG_IDS_I32, SEG_IDS_I64,
                    SEG_NONE, ColumnSpec, PlanSpec)
from . import tf_proto as P
from .view import GraphView, tensor_name

RESHAPE_LIKE = ("Reshape", "ExpandDims", "Squeeze")          # IsReshape, cuda_emitter.cc:62-73
SEGMENT_OPS = {"SparseSegmentSumWithNumSegments": COMBINER_SUM, "SparseSegmentMeanWithNumSegments": COMBINER_MEAN}
_ELEM_SIZE = {P.DT_FLOAT: 4, P.DT_INT32: 4, P.DT_INT64: 8}


class Unsupported(Exception):
    """This subgraph is not one the fused path takes; the caller falls back."""


@dataclass
class IndexSource:
    """Where an index operand really comes from (``EmitInputInline``)."""
    tensor: str                 # graph tensor that ConcatInputs receives
    dtype: int
    rank: int
    stride: int = 1             # element stride into `tensor` (StridedSlice [:, 0:1] of [n, k] → k)
    boundaries: Optional[np.ndarray] = None
    # id transform absorbed on the way (Addons>SelectValue / GatherIndiceValue / GatherValueGenIndice)
    xform_mode: int = XFORM_NONE
    xform_lo: Tuple[int, ...] = ()
    xform_hi: Tuple[int, ...] = ()
    xform_substitute: int = 0
    hash_buckets: int = 0                # StringToHashBucketFast(AsString(int ids)) absorbed: hash on the device
    filter_node: Optional[str] = None    # the Gather* node whose (indices, values) pair this operand belongs to
    generated_rows: bool = False         # indices output of GatherValueGenIndice: row i of the values tensor
    # a SparseReshape folded into the index expression (cuda_emitter.cc:1874-1916): `reshaped` is set while the operand
    # is the op's whole output_indices matrix, (input rank, mul, div, symbol source, symbol slot); the [:, 0:1] slice
    # turns it into the segment-id map below (ColumnSpec.seg_mul / seg_div / seg_sym / seg_sym_slot)
    reshaped: Optional[tuple] = None
    seg_mul: Tuple[int, ...] = ()
    seg_div: int = 1
    seg_sym: Optional[Tuple[str, int]] = None      # (tensor, flat index) whose value multiplies one factor
    seg_sym_slot: int = 0


@dataclass
class SymbolDef:
    """symbol value = ``reshape(tensor, [-1])[index]`` at request time — or, ``last_stride`` > 0, the row count of a plain
    ``SparseSegmentSum`` / ``SparseSegmentMean`` (no ``num_segments``): last segment id + 1 over the sorted segment ids
    ``reshape(tensor, [-1])[::last_stride]``, 0 when there are none (TensorFlow's output shape for those ops)."""
    tensor: str
    index: int
    last_stride: int = 0


@dataclass
class ColumnInfo:
    value_tensor: str           # output tensor of the value node (before trailing reshapes)
    concat_input: str           # the tensor name as written in the ConcatV2 input list
    concat_index: int


@dataclass
class GroupInfo:
    concat_node: str
    dtype: int
    n_inputs: int
    columns: List[int] = field(default_factory=list)     # plan column index per concat input position


@dataclass
class BuiltPlan:
    spec: PlanSpec
    host_inputs: List[Tuple[str, int, int]]               # (tensor, dtype, rank)  → ConcatInputs `inputs`, `T`, `ranks`
    device_inputs: List[Tuple[str, int, int]]             # (tensor, dtype, rank)  → FeatureColumnProcess `inputs`
    symbols: List[SymbolDef]
    groups: List[GroupInfo]
    columns: List[ColumnInfo]
    skipped: List[Tuple[str, str]]                        # (node, reason) — left to TensorFlow

    def describe(self) -> str:
        names = {1: "gather", 2: "segment-reduce", 3: "gather-scatter", 4: "passthrough", 5: "batch-col-reduction",
                 6: "external (ConcatOutputs host input)"}
        lines = [f"{len(self.groups)} concat group(s), {self.spec.n_columns} column(s), "
                 f"{len(self.host_inputs)} host input(s), {len(self.device_inputs)} table(s), "
                 f"{len(self.symbols)} symbol(s)"]
        for g, gi in enumerate(self.groups):
            forms: Dict[str, int] = {}
            for k in gi.columns:
                n = names[self.spec.columns[k].form]
                forms[n] = forms.get(n, 0) + 1
            lines.append(f"  group {g}: {gi.concat_node}  width {self.spec.group_width(g)}  " +
                         ", ".join(f"{v}x {k}" for k, v in sorted(forms.items())))
        for node, why in self.skipped:
            lines.append(f"  skipped {node}: {why}")
        return "\n".join(lines)


class PlanBuilder:
    def __init__(self, graph_def, host_concat: str = "passthrough") -> None:
        if host_concat not in ("passthrough", "external"):
            raise ValueError("host_concat must be 'passthrough' or 'external'")
        self.host_concat = host_concat
        self.g = GraphView(graph_def)
        self.tables = self._find_tables()
        self._host: Dict[str, int] = {}
        self._host_list: List[Tuple[str, int, int]] = []
        self._dev: Dict[str, int] = {}
        self._dev_list: List[Tuple[str, int, int]] = []
        self._sym: Dict[Tuple[str, int], int] = {}
        self._sym_list: List[SymbolDef] = []

    # ---- tables (graph_info.cc:209-259) ----------------------------------------
    def _find_tables(self) -> Dict[str, Tuple[int, int]]:
        """A table is a VariableV2 / Const — or, in TF2 SavedModels, a VarHandleOp (resource variable; the
        reference registers shape functions for it, symbolic_shape/op_infer_fn/) — whose consumers, through
        Identity / ReadVariableOp, are only lookups."""
        out: Dict[str, Tuple[int, int]] = {}
        for n in self.g.gd.node:
            if n.op not in ("VariableV2", "Const", "VarHandleOp"):
                continue
            if n.op == "VarHandleOp":
                if n.attr["dtype"].type != P.DT_FLOAT or n.attr["shape"].shape.unknown_rank:
                    continue
                shape = [int(d.size) if d.size >= 0 else None for d in n.attr["shape"].shape.dim]
            else:
                if self.g.out_dtype(n) != P.DT_FLOAT:
                    continue
                shape = self.g.static_shape(n)
            if shape is None or len(shape) != 2 or None in shape or min(shape) <= 0:
                continue
            stack, lookups, ok = [n.name], 0, True
            while stack and ok:
                for c, i in self.g.data_consumers(stack.pop()):
                    if c.op in ("Identity", "ReadVariableOp"):
                        stack.append(c.name)
                    elif c.op in ("Assign", "SaveV2", "AssignVariableOp", "VarIsInitializedOp"):
                        pass
                    elif ("Gather" in c.op or "SparseSegment" in c.op) and i == 0:
                        lookups += 1
                    else:
                        ok = False
                        break
            if ok and lookups:
                out[n.name] = (int(shape[0]), int(shape[1]))
        return out

    def _table_of(self, node, port: int) -> Tuple[str, int, int]:
        while node.op in ("Identity", "ReadVariableOp") and port == 0:
            node, port = self.g.input(node, 0)
        if node.name not in self.tables or port != 0:
            raise Unsupported(f"{node.name} ({node.op}) is not an embedding table")
        return (node.name,) + self.tables[node.name]

    # ---- operand bookkeeping ------------------------------------------------------
    def _host_input(self, tensor: str, dtype: int, rank: int) -> int:
        if tensor not in self._host:
            self._host[tensor] = len(self._host_list)
            self._host_list.append((tensor, dtype, rank))
        return self._host[tensor]

    def _device_input(self, tensor: str) -> int:
        """Tables become FeatureColumnProcess `inputs`.  A resource variable's handle is not its data: its
        device input is the output of a ReadVariableOp (an existing one, or `<var>/fcp_read`, which
        rewrite_graph adds)."""
        if tensor not in self._dev:
            name = tensor
            node = self.g.nodes.get(tensor)
            if node is not None and node.op == "VarHandleOp":
                reads = [c.name for c, i in self.g.data_consumers(tensor) if c.op == "ReadVariableOp" and i == 0]
                name = reads[0] if reads else tensor + "/fcp_read"
            self._dev[tensor] = len(self._dev_list)
            self._dev_list.append((name, P.DT_FLOAT, 2))
        return self._dev[tensor]

    def _symbol(self, tensor: str, index: int, last_stride: int = 0) -> int:
        key = (tensor, index, last_stride)
        if key not in self._sym:
            self._sym[key] = len(self._sym_list)
            self._sym_list.append(SymbolDef(tensor, index, last_stride))
        return self._sym[key]

    # ---- EmitInputInline (cuda_emitter.cc:1769-1949) ---------------------------------
    def _terminal(self, node, port: int) -> IndexSource:
        dtype = self.g.out_dtype(node, port)
        shape = self.g.static_shape(node, port)
        if shape is None:
            raise Unsupported(f"rank of {node.name} unknown")
        return IndexSource(tensor_name(node.name, port), dtype, len(shape))

    def trace_index(self, node, port: int) -> IndexSource:
        """Follow an index operand to the tensor the host must ship.  Ops that can be
        evaluated on the fly are absorbed; the walk ends at the first other node."""
        try:
            return self._trace_inline(node, port)
        except Unsupported:
            return self._terminal(node, port)

    def _trace_inline(self, node, port: int) -> IndexSource:
        g = self.g
        if node.op.startswith("Addons>"):
            return self._trace_id_filter(node, port)
        if port != 0:
            raise Unsupported("not an inlinable op")
        if node.op == "StringToHashBucketFast":
            # categorical_column_with_hash_bucket over an INTEGER feature: AsString(ids) -> StringToHashBucketFast.
            # The pair is evaluated on the device (Fingerprint64 of the decimal string); a string feature, or an
            # AsString with formatting attrs, ends the walk here and stays on the CPU.
            a_node, a_port = g.input(node, 0)
            if a_node.op != "AsString" or a_port != 0:
                raise Unsupported("StringToHashBucketFast over a string tensor")
            at = a_node.attr
            if at["T"].type not in (P.DT_INT32, P.DT_INT64) or ("width" in at and at["width"].i not in (-1, 0)) or \
                    ("fill" in at and at["fill"].s not in (b"",)) or ("scientific" in at and at["scientific"].b) or \
                    ("shortest" in at and at["shortest"].b):
                raise Unsupported("AsString with formatting")
            src = self.trace_index(*g.input(a_node, 0))
            if src.boundaries is not None or src.xform_mode != XFORM_NONE or src.hash_buckets or src.stride != 1 or \
                    src.filter_node is not None or src.dtype not in (P.DT_INT32, P.DT_INT64):
                raise Unsupported("hash of a transformed id stream")
            src.hash_buckets = int(node.attr["num_buckets"].i)
            return src
        if node.op == "SparseReshape":
            # output_indices of a reshape whose row coordinate is an expression of the input coordinates with
            # provable factors (:1874-1916): the column reads the ORIGINAL index matrix and the pre-pass evaluates the
            # expression.  Anything else (unprovable shapes, several run-time factors) ends the walk: TensorFlow
            # computes the op and its output tensor is shipped, as for any op the reference does not inline (:1924-1933).
            m = self._sparse_reshape_map(node)
            src = self.trace_index(*g.input(node, 0))
            if src.reshaped is not None or src.stride != 1 or src.boundaries is not None or src.rank != 2:
                raise Unsupported("SparseReshape over a transformed index matrix")
            src.reshaped = m
            return src
        if node.op in RESHAPE_LIKE or node.op == "Identity":
            return self.trace_index(*g.input(node, 0))          # flat element index unchanged
        if node.op == "Cast":
            src = self.trace_index(*g.input(node, 0))
            dst = node.attr["DstT"].type
            ints = (P.DT_INT32, P.DT_INT64)
            # integer width changes and the int32→int64 cast TF puts after Bucketize keep the
            # value; a float→int cast has no id source and stays on the host
            if dst in ints and (src.dtype in ints or src.boundaries is not None):
                return src
            raise Unsupported("cast changes the value")
        if node.op == "Bucketize":
            src = self.trace_index(*g.input(node, 0))
            if src.dtype != P.DT_FLOAT or src.boundaries is not None or src.stride != 1:
                raise Unsupported("Bucketize over a non-float32 operand")
            b = np.asarray(list(node.attr["boundaries"].list.f), np.float32)
            if b.size == 0:
                raise Unsupported("Bucketize without boundaries")
            src.boundaries = b
            return src
        if node.op == "StridedSlice":
            # the one case the reference inlines: [n, k] → column 0 (:1864-1873), with or
            # without the shrink the lookup optimizer adds for segment ids (lookup_optimizer.cc:240-242)
            spec = g.strided_slice_spec(node)
            in_node, in_port = g.input(node, 0)
            in_shape = g.static_shape(in_node, in_port)
            if spec is None or in_shape is None or len(in_shape) != 2 or in_shape[1] is None:
                raise Unsupported("StridedSlice operand shape unknown")
            if spec["ellipsis_mask"] or spec["new_axis_mask"] or spec["shrink_axis_mask"] not in (0, 2):
                raise Unsupported("StridedSlice masks")
            if len(spec["begin"]) != 2 or spec["strides"] != [1, 1]:
                raise Unsupported("StridedSlice is not a 2-D unit-stride slice")
            if not ((spec["begin_mask"] & 1 or spec["begin"][0] == 0) and spec["end_mask"] & 1):
                raise Unsupported("StridedSlice does not keep all rows")
            if spec["begin_mask"] & 2 or spec["end_mask"] & 2 or spec["begin"][1] != 0 or spec["end"][1] != 1:
                raise Unsupported("StridedSlice does not select column 0")
            src = self.trace_index(in_node, in_port)
            if src.boundaries is not None:
                raise Unsupported("slice of bucketized values")
            if src.reshaped is not None:       # column 0 of a SparseReshape's output: the row coordinate of the map
                rank, mul, div, sym, slot = src.reshaped
                src.reshaped = None
                src.stride *= rank
                if not (mul == (1,) and div == 1 and sym is None):      # (the identity on idx0 needs no map)
                    src.seg_mul, src.seg_div, src.seg_sym, src.seg_sym_slot = mul, div, sym, slot
                return src
            src.stride *= int(in_shape[1])
            return src
        raise Unsupported("not an inlinable op")

    def _trace_id_filter(self, node, port: int) -> IndexSource:
        """the CPU id ops PreLookupOptimizer leaves in front of a lookup
        (``pre_lookup_optimizer.cc:596-654``) become the column's id transform, evaluated on the device next
        to Bucketize; the walk continues at the op's input.  One transform per column."""
        g = self.g
        lo = tuple(int(v) for v in node.attr["left_boundaries"].list.i)
        hi = tuple(int(v) for v in node.attr["right_boundaries"].list.i)
        if len(lo) != len(hi) or any(a > b for a, b in zip(lo, hi)):
            raise Unsupported("malformed interval attrs")

        def absorb(k: int, mode: int, sub: int = 0) -> IndexSource:
            """the values operand (input k) with this op's transform on top; an operand that already carries a
            transform (two id ops in a row) is not walked further: the inner op stays in TensorFlow"""
            in_node, in_port = g.input(node, k)
            src = self.trace_index(in_node, in_port)
            if src.xform_mode != XFORM_NONE or src.stride != 1 or src.generated_rows or src.filter_node is not None:
                src = self._terminal(in_node, in_port)
            src.xform_mode, src.xform_lo, src.xform_hi, src.xform_substitute = mode, lo, hi, sub
            return src

        if node.op == "Addons>SelectValue" and port == 0:
            return absorb(0, XFORM_SELECT, int(node.attr["substitute"].i))
        if node.op == "Addons>GatherIndiceValue":
            if port == 1:                                    # the surviving values: filter the original ones
                src = absorb(1, XFORM_FILTER)
            elif port == 0:                                  # their indices: the original indices, same filter
                src = self.trace_index(*g.input(node, 0))
                if src.filter_node is not None or src.xform_mode != XFORM_NONE:
                    src = self._terminal(*g.input(node, 0))
            else:
                raise Unsupported("not an inlinable op")
            src.filter_node = node.name
            return src
        if node.op == "Addons>GatherValueGenIndice":
            if port == 1:
                src = absorb(0, XFORM_FILTER)
            elif port == 0:                                  # index [i] of every surviving value i
                src = self._terminal(*g.input(node, 0))
                src.generated_rows = True
            else:
                raise Unsupported("not an inlinable op")
            src.filter_node = node.name
            return src
        raise Unsupported("not an inlinable op")

    def _sparse_reshape_map(self, node) -> tuple:
        """``SparseReshape(indices [nnz, r], shape [r], new_shape [q])``: the row coordinate of the reshaped element is
        ``(sum_k idx_k * prod(shape[k+1:])) // prod(new_shape[1:])`` (the reference's flat-index algebra,
        ``cuda_emitter.cc:1874-1916``, offset 0; it asks SymEngine for the shapes' contents, here every entry is traced
        to a constant or to a plain copy of one element of another tensor, ``GraphView.elem_source``).  A trailing
        coordinate drops out when its dimension divides the denominator (``idx < dim``), which is what makes the usual
        ``[B, L] -> [B, L]`` and ``[B, T, L] -> [B*T, L]`` reshapes independent of the run-time ``L``.  What remains may
        contain ONE run-time factor (it becomes a symbol).  -> (r, mul, div, symbol source or None, symbol slot)."""
        g = self.g
        shape, new = g.input(node, 1), g.input(node, 2)
        rs, qs = g.static_shape(*shape), g.static_shape(*new)
        if rs is None or qs is None or len(rs) != 1 or len(qs) != 1 or rs[0] is None or qs[0] is None:
            raise Unsupported("SparseReshape: ranks unknown")
        r, q = int(rs[0]), int(qs[0])
        if not 1 <= r <= 4 or q < 1:
            raise Unsupported("SparseReshape: more than 4 input coordinates")
        ins = [g.elem_source(*shape, k) for k in range(1, r)]
        outs = [g.elem_source(*new, k) for k in range(1, q)]
        if any(e is None or (e[0] == "const" and e[1] <= 0) for e in ins + outs):
            raise Unsupported("SparseReshape: a dimension is computed (or inferred at run time)")

        def product(entries):
            c, syms = 1, []
            for e in entries:
                if e[0] == "const":
                    c *= int(e[1])
                else:
                    syms.append((e[1], int(e[2])))
            return [c, syms]

        mul = [product(ins[k:]) for k in range(r)]          # multiplier of coordinate k: prod(shape[k+1:])
        div = product(outs)
        n = r
        while n >= 2:                                       # drop trailing coordinates whose dimension divides `div`
            last = ins[n - 2]                               # shape[n-1]
            if last[0] == "const":
                if div[0] % last[1]:
                    break
                div[0] //= last[1]
                for k in range(n - 1):
                    mul[k][0] //= last[1]
            else:
                key = (last[1], int(last[2]))
                if key not in div[1]:
                    break
                div[1].remove(key)
                for k in range(n - 1):
                    mul[k][1].remove(key)
            n -= 1
        mul = mul[:n]
        runtime = [(slot, f[1]) for slot, f in list(enumerate(mul)) + [(4, div)] if f[1]]
        if len(runtime) > 1 or (runtime and len(runtime[0][1]) != 1):
            raise Unsupported("SparseReshape: more than one run-time factor")
        sym, slot = (runtime[0][1][0], runtime[0][0]) if runtime else (None, 0)
        return r, tuple(int(f[0]) for f in mul), int(div[0]), sym, slot

    def _ids_operand(self, node, port: int):
        """-> (host input, id_source, boundaries, transform kwargs for ColumnSpec, filter node)"""
        src = self.trace_index(node, port)
        if src.stride != 1 or src.generated_rows or src.reshaped is not None:  # no strided id source in the column record
            src = self._terminal(node, port)
        if src.boundaries is not None:
            id_source = IDS_F32_BUCKETIZE
        elif src.dtype == P.DT_INT32:
            id_source = IDS_I32
        elif src.dtype == P.DT_INT64:
            id_source = IDS_I64
        else:
            raise Unsupported(f"ids tensor {src.tensor} has dtype {src.dtype}")
        xf = dict(xform_mode=src.xform_mode, xform_lo=src.xform_lo, xform_hi=src.xform_hi,
                  xform_substitute=src.xform_substitute, hash_buckets=src.hash_buckets)
        return self._host_input(src.tensor, src.dtype, src.rank), id_source, src.boundaries, xf, src.filter_node

    def _seg_operand(self, node, port: int, filter_node: Optional[str] = None, allow_map: bool = False) -> tuple:
        """``filter_node``: the Gather* op the column's ids went through; its indices output may be read
        through (the kernel drops the same pairs), any other filtered index stream may not."""
        src = self.trace_index(node, port)
        if src.boundaries is not None or src.xform_mode != XFORM_NONE or src.generated_rows or \
                src.filter_node != filter_node or src.reshaped is not None or (src.seg_mul and not allow_map):
            if filter_node is not None:
                raise Unsupported("segment ids do not come from the id filter's indices output")
            src = self._terminal(node, port)
        kind = {P.DT_INT32: SEG_IDS_I32, P.DT_INT64: SEG_IDS_I64}.get(src.dtype)
        if kind is None:
            raise Unsupported(f"segment ids {src.tensor} have dtype {src.dtype}")
        seg_map = {}
        if src.seg_mul:                  # a SparseReshape folded in: the row coordinate as an expression of the original ones
            seg_map = dict(seg_mul=src.seg_mul, seg_div=src.seg_div, seg_sym_slot=src.seg_sym_slot,
                           seg_sym=-1 if src.seg_sym is None else self._symbol(*src.seg_sym))
        return self._host_input(src.tensor, src.dtype, src.rank), kind, src.stride, seg_map

    # ---- EmitSubgraphCode dispatch (cuda_emitter.cc:1096-1152) --------------------------
    def _match_gather(self, node):
        axis = self.g.const_array(*self.g.input(node, 2))
        if axis is None or int(axis.reshape(-1)[0]) != 0:
            raise Unsupported("GatherV2 axis is not 0")
        if "batch_dims" in node.attr and node.attr["batch_dims"].i != 0:
            raise Unsupported("GatherV2 batch_dims")
        table, vocab, dim = self._table_of(*self.g.input(node, 0))
        ids_in, id_source, boundaries, xf, fnode = self._ids_operand(*self.g.input(node, 1))
        return table, vocab, dim, ids_in, id_source, boundaries, xf, fnode

    def match_column(self, node, port: int, group: int, slot: int) -> ColumnSpec:
        g = self.g
        if port != 0:
            raise Unsupported("value is not output 0")
        if node.op == "ResourceGather":                                             # GatherV2 over a resource variable (TF2)
            if "batch_dims" in node.attr and node.attr["batch_dims"].i != 0:
                raise Unsupported("ResourceGather batch_dims")
            table, vocab, dim = self._table_of(*g.input(node, 0))
            ids_in, id_source, bnd, xf, fnode = self._ids_operand(*g.input(node, 1))
            if fnode is not None:
                raise Unsupported("ResourceGather over filtered values")
            return ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, id_source, self._device_input(table), ids_in,
                              -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, bnd, group, slot, **xf)
        if node.op == "GatherV2":                                                   # EmitGatherRows :1246-1330
            table, vocab, dim, ids_in, id_source, bnd, xf, fnode = self._match_gather(node)
            if fnode is not None:                         # compacted values without their indices: rows are lost
                raise Unsupported("GatherV2 over filtered values")
            return ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, id_source, self._device_input(table), ids_in,
                              -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, bnd, group, slot, **xf)
        if node.op in SEGMENT_OPS:                                                  # EmitSparseSegmentReduce* :1444-1760
            table, vocab, dim = self._table_of(*g.input(node, 0))
            ids_in, id_source, bnd, xf, fnode = self._ids_operand(*g.input(node, 1))
            seg_in, seg_kind, stride, seg_map = self._seg_operand(*g.input(node, 2), fnode, allow_map=True)
            n_node, n_port = g.input(node, 3)
            while n_node.op in RESHAPE_LIKE and n_port == 0:                        # Squeeze(num_segments) lookup_optimizer.cc:248-254
                n_node, n_port = g.input(n_node, 0)
            sym = self._symbol(tensor_name(n_node.name, n_port), 0)
            return ColumnSpec(FORM_SEGMENT_REDUCE, dim, vocab, SEGMENT_OPS[node.op], id_source,
                              self._device_input(table), ids_in, seg_in, seg_kind, stride, ROWS_FROM_SYMBOL, sym, bnd,
                              group, slot, **xf, **seg_map)
        if node.op in ("SparseSegmentSum", "SparseSegmentMean"):
            # no num_segments (the emitter takes these too, cuda_emitter.cc:1096-1113; its row count is a SymEngine symbol,
            # :1444-1622): rows = last segment id + 1, which the rewritten graph computes from the sorted segment ids the host
            # ships anyway (a host tensor: a symbol, not a device -> host round trip)
            table, vocab, dim = self._table_of(*g.input(node, 0))
            ids_in, id_source, bnd, xf, fnode = self._ids_operand(*g.input(node, 1))
            if fnode is not None:
                raise Unsupported("row count of a plain SparseSegment op over filtered ids depends on what the filter keeps")
            seg_in, seg_kind, stride, _ = self._seg_operand(*g.input(node, 2), None)
            sym = self._symbol(self._host_list[seg_in][0], 0, last_stride=stride)
            return ColumnSpec(FORM_SEGMENT_REDUCE, dim, vocab, COMBINER_SUM if node.op == "SparseSegmentSum" else COMBINER_MEAN,
                              id_source, self._device_input(table), ids_in, seg_in, seg_kind, stride, ROWS_FROM_SYMBOL, sym, bnd,
                              group, slot, **xf)
        if node.op == "ScatterNd":                                                  # EmitGatherScatterRows :1332-1442
            upd, upd_port = g.input(node, 1)
            if upd.op != "GatherV2" or upd_port != 0:
                raise Unsupported("ScatterNd updates are not a GatherV2")
            table, vocab, dim, ids_in, id_source, bnd, xf, fnode = self._match_gather(upd)
            rows = self.trace_index(*g.input(node, 0))
            if fnode is not None and rows.generated_rows and rows.filter_node == fnode:
                # ScatterNd(GatherValueGenIndice:0, GatherV2(table, GatherValueGenIndice:1)): surviving value i
                # goes to row i — a one-hot gather over the ORIGINAL values whose dropped ids leave zero rows
                return ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, id_source, self._device_input(table), ids_in,
                                  -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, bnd, group, slot, **xf)
            seg_in, seg_kind, stride, _ = self._seg_operand(*g.input(node, 0), fnode)
            shp, shp_port = g.input(node, 2)
            sym = self._symbol(tensor_name(shp.name, shp_port), 0)
            return ColumnSpec(FORM_GATHER_SCATTER, dim, vocab, COMBINER_NONE, id_source, self._device_input(table),
                              ids_in, seg_in, seg_kind, stride, ROWS_FROM_SYMBOL, sym, bnd, group, slot, **xf)
        if node.op == "Sum":                                                        # EmitBatchColReduction :1180-1244
            axis = g.const_array(*g.input(node, 1))
            if axis is None or axis.size != 1 or int(axis.reshape(-1)[0]) != 1:
                raise Unsupported("Sum is not over axis 1")
            if "keep_dims" in node.attr and node.attr["keep_dims"].b:
                raise Unsupported("Sum keep_dims")
            x, x_port = g.input(node, 0)
            shape = g.static_shape(x, x_port)
            if g.out_dtype(x, x_port) != P.DT_FLOAT or shape is None or len(shape) != 3 or shape[2] is None:
                raise Unsupported("Sum operand is not a float32 [b, r, c] tensor with static c")
            i = self._host_input(tensor_name(x.name, x_port), P.DT_FLOAT, 3)
            return ColumnSpec(FORM_BATCH_COL_REDUCTION, int(shape[2]), 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE,
                              1, ROWS_FROM_INPUT_DIM0, i, None, group, slot)
        raise Unsupported(f"op {node.op} is not a lookup")

    def _passthrough(self, tensor: str, group: int, slot: int) -> ColumnSpec:
        name, port = tensor.partition(":")[0], int(tensor.partition(":")[2] or 0)
        node = self.g.nodes[name]
        shape = self.g.static_shape(node, port)
        if self.g.out_dtype(node, port) != P.DT_FLOAT or shape is None or len(shape) != 2 or shape[1] is None:
            raise Unsupported(f"{tensor}: not a float32 [rows, dim] tensor with static dim")
        i = self._host_input(tensor, P.DT_FLOAT, 2)
        return ColumnSpec(FORM_PASSTHROUGH, int(shape[1]), 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1,
                          ROWS_FROM_INPUT_DIM0, i, None, group, slot)

    def _external(self, tensor: str, group: int, slot: int) -> ColumnSpec:
        """The reference's wiring of a non-FC concat input: it stays a TensorFlow tensor and reaches
        ``Addons>ConcatOutputs`` as a host input; the plan only reserves its concat slot."""
        name, port = tensor.partition(":")[0], int(tensor.partition(":")[2] or 0)
        node = self.g.nodes[name]
        shape = self.g.static_shape(node, port)
        if self.g.out_dtype(node, port) != P.DT_FLOAT or shape is None or len(shape) != 2 or shape[1] is None:
            raise Unsupported(f"{tensor}: not a float32 [rows, dim] tensor with static dim")
        return ColumnSpec(FORM_EXTERNAL, int(shape[1]), 0, COMBINER_NONE, IDS_I32, -1, -1, -1, SEG_NONE, 1,
                          ROWS_FROM_GROUP, 0, None, group, slot)

    # ---- the walk ------------------------------------------------------------------------
    def build(self) -> BuiltPlan:
        g = self.g
        columns: List[ColumnSpec] = []
        infos: List[ColumnInfo] = []
        groups: List[GroupInfo] = []
        skipped: List[Tuple[str, str]] = []
        for concat in g.gd.node:
            if concat.op != "ConcatV2":
                continue
            ins = g.data_inputs(concat)
            n = int(concat.attr["N"].i) if "N" in concat.attr else len(ins) - 1
            axis = g.const_array(*g.input(concat, n))
            dtype = g.out_dtype(concat)
            if axis is None or int(axis.reshape(-1)[0]) not in (1, -1) or dtype != P.DT_FLOAT:
                continue
            # snapshot: a group that turns out unusable must not leave operands behind
            snap = (dict(self._host), list(self._host_list), dict(self._dev), list(self._dev_list), dict(self._sym),
                    list(self._sym_list))
            group = len(groups)
            cols: List[ColumnSpec] = []
            cinfo: List[ColumnInfo] = []
            lookups = 0
            try:
                for i in range(n):
                    node, port = g.input(concat, i)
                    while node.op in RESHAPE_LIKE and port == 0:                    # FindFCOutputs :1060-1066
                        node, port = g.input(node, 0)
                    try:
                        col = self.match_column(node, port, group, i)
                        lookups += 1
                        value = tensor_name(node.name, port)
                    except Unsupported as why:
                        col = (self._external if self.host_concat == "external" else self._passthrough)(ins[i], group, i)
                        value = ins[i]
                        if node.name in self.tables or any(t in self.tables for t in self._upstream_tables(node)):
                            skipped.append((node.name, str(why)))
                    cols.append(col)
                    cinfo.append(ColumnInfo(value, ins[i], i))
                if lookups == 0:
                    raise Unsupported("no lookup column converges here")
            except Unsupported as why:
                (self._host, self._host_list, self._dev, self._dev_list, self._sym, self._sym_list) = snap
                if lookups:
                    skipped.append((concat.name, str(why)))
                continue
            gi = GroupInfo(concat.name, dtype, n, list(range(len(columns), len(columns) + n)))
            columns.extend(cols)
            infos.extend(cinfo)
            groups.append(gi)
        if not groups:
            raise Unsupported("no ConcatV2 with embedding lookups found")
        spec = PlanSpec(columns, [r for _, _, r in self._host_list],
                        [_ELEM_SIZE[d] for _, d, _ in self._host_list], len(self._dev_list), n_groups=len(groups),
                        n_symbols=len(self._sym_list))
        spec.validate()
        return BuiltPlan(spec, list(self._host_list), list(self._dev_list), list(self._sym_list), groups, infos,
                         skipped)

    def _upstream_tables(self, node, limit: int = 256) -> List[str]:
        seen, stack, found = set(), [node], []
        while stack and len(seen) < limit:
            n = stack.pop()
            if n.name in seen:
                continue
            seen.add(n.name)
            if n.name in self.tables:
                found.append(n.name)
            for k in range(len(self.g.data_inputs(n))):
                stack.append(self.g.input(n, k)[0])
        return found


def build_plan(graph_def, host_concat: str = "passthrough") -> BuiltPlan:
    return PlanBuilder(graph_def, host_concat).build()
