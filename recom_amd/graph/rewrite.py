"""Graph rewrite: replace the matched feature-column subgraphs with the three ops.

Node for node what ``CudaEmitter::Rewrite`` emits (``cuda_emitter.cc:2496-2656``):

* ``ConcatInputs`` (``Addons>ConcatInputs``): inputs = the plan's host tensors, attrs
  ``T`` / ``ranks`` (``:2518-2524``);
* ``FeatureColumnProcess`` (``Addons>FeatureColumnProcess``, ``…WithSymbols`` when the
  plan has symbols, ``:2650-2653``): inputs ``ConcatInputs:0..2`` + the tables (+
  symbols), attrs ``input_types`` / ``input_ranks`` / ``output_types`` /
  ``output_ranks`` / ``dlpath`` (``:2501-2512, 2526-2539``) — ``dlpath`` names the plan
  file (``recom_amd.plan_io``) instead of a JIT ``.so``;
* one ``Addons>ConcatOutputs[NoHost]`` per concat group that takes over the ConcatV2's
  name, the original being renamed ``<name>_removed`` (``:2645-2646``), with attrs ``T``,
  ``BLOCK_THREADS``, ``prefix_begin`` / ``prefix_end``, ``output_dir``,
  ``device_concat_indices`` / ``device_input_indices``, ``N`` / ``host_concat_indices`` /
  the ``host_inputs`` themselves for the plan's ``FORM_EXTERNAL`` slots (``:2594-2611``;
  ``NoHost`` when there are none), ``embedd_dims`` over ALL concat positions,
  ``buffer_types`` and the lifetime-extending ``tensor_buffers`` inputs blob, tables,
  arena (``:2547-2643``).

Symbols (row counts the reference derives with SymEngine and ships through a
``ShapeConstruct`` node, ``:2446-2458``) are computed by plain TF ops here:
``Cast(GatherV2(Reshape(t, [-1]), index))`` packed into one int32 vector.

Nodes that only fed the removed concats are pruned (the reference leaves that to its
``UselessNodesPruner`` / TensorFlow's own pruning).
"""
from __future__ import annotations

from typing import List

import numpy as np

from . import tf_proto as P
from .plan_builder import BuiltPlan
from .view import GraphView, numpy_to_tensor, split_tensor

BLOCK_THREADS = 64  # CudaEmitter(graph_info, 1 << 28, 64), fc_optimize_pass.cc:71


def _const(gd, name: str, value: np.ndarray):
    n = gd.node.add(name=name, op="Const")
    n.attr["dtype"].type = {np.dtype(np.int32): P.DT_INT32, np.dtype(np.int64): P.DT_INT64}[value.dtype]
    numpy_to_tensor(value, n.attr["value"].tensor)
    return n


def rewrite_graph(graph_def, built: BuiltPlan, plan_path: str, prune: bool = True, stage=None):
    """Returns a new GraphDef; ``graph_def`` is left untouched.

    ``stage`` (``PlanSpec.staged_for_concat_inputs()``): the graph is rewritten for the STAGED plan — the ConcatInputs
    node carries ``_fcp_plan`` = the plan file (an underscore attr: node-private, not part of the op's registered
    signature, which stays the reference's ``T`` / ``ranks``) and, when row ids are converted to offsets, receives the
    symbols vector as its last input, because the row counts are not among the tensors it packs."""
    gd = P.GraphDef()
    gd.CopyFrom(graph_def)
    view = GraphView(gd)
    for reserved in ("ConcatInputs", "FeatureColumnProcess"):
        if reserved in view.nodes:
            raise ValueError(f"graph already has a node named {reserved}")

    concat_in = gd.node.add(name="ConcatInputs", op="Addons>ConcatInputs")
    for tensor, dtype, rank in built.host_inputs:
        concat_in.input.append(tensor)
        concat_in.attr["T"].list.type.append(dtype)
        concat_in.attr["ranks"].list.i.append(rank)

    if stage is not None:
        concat_in.attr["_fcp_plan"].s = plan_path.encode()
        if stage.symbols_input >= 0:
            if stage.symbols_input != len(built.host_inputs) or not built.symbols:
                raise ValueError("the stage section expects the symbols vector as the last ConcatInputs input")
            concat_in.input.append("FeatureColumnProcess/symbols")
            concat_in.attr["T"].list.type.append(P.DT_INT32)
            concat_in.attr["ranks"].list.i.append(1)

    fuse = gd.node.add(name="FeatureColumnProcess", op="Addons>FeatureColumnProcess")
    fuse.attr["dlpath"].s = plan_path.encode()
    fuse.input.extend(["ConcatInputs", "ConcatInputs:1", "ConcatInputs:2"])
    fuse.attr["input_types"].list.SetInParent()
    fuse.attr["input_ranks"].list.SetInParent()
    for tensor, dtype, rank in built.device_inputs:
        if tensor.endswith("/fcp_read") and tensor not in view.nodes:   # a resource variable nobody reads as a tensor yet
            rd = gd.node.add(name=tensor, op="ReadVariableOp")
            rd.input.append(tensor[:-len("/fcp_read")])
            rd.attr["dtype"].type = dtype
        fuse.input.append(tensor)
        fuse.attr["input_types"].list.type.append(dtype)
        fuse.attr["input_ranks"].list.i.append(rank)
    out_index = {col: i for i, col in enumerate(built.spec.output_columns())}
    for _ in out_index:                                # every column output is [prefix, dim]
        fuse.attr["output_types"].list.type.append(P.DT_FLOAT)
        fuse.attr["output_ranks"].list.i.append(2)

    if built.symbols:
        _const(gd, "FeatureColumnProcess/symbols/flat_shape", np.asarray([-1], np.int32))
        _const(gd, "FeatureColumnProcess/symbols/axis", np.asarray(0, np.int32))
        pack = gd.node.add(name="FeatureColumnProcess/symbols", op="Pack")
        pack.attr["N"].i = len(built.symbols)
        pack.attr["T"].type = P.DT_INT32
        pack.attr["axis"].i = 0
        for k, sym in enumerate(built.symbols):
            src, port = split_tensor(sym.tensor)
            dtype = view.out_dtype(view.nodes[src], port)
            base = f"FeatureColumnProcess/symbols/s{k}"
            flat = gd.node.add(name=base + "/flat", op="Reshape")
            flat.input.extend([sym.tensor, "FeatureColumnProcess/symbols/flat_shape"])
            flat.attr["T"].type = dtype
            flat.attr["Tshape"].type = P.DT_INT32
            if sym.last_stride > 0:
                # rows of a plain SparseSegmentSum / Mean = max(sorted segment ids, -1) + 1 (0 rows without ids)
                last = flat.name
                if sym.last_stride > 1:                    # column 0 of an [nnz, k] index matrix: flat[::k]
                    _const(gd, base + "/zero", np.asarray([0], np.int32))
                    _const(gd, base + "/stride", np.asarray([sym.last_stride], np.int32))
                    col = gd.node.add(name=base + "/col", op="StridedSlice")
                    col.input.extend([flat.name, base + "/zero", base + "/zero", base + "/stride"])
                    col.attr["T"].type = dtype
                    col.attr["Index"].type = P.DT_INT32
                    col.attr["begin_mask"].i = 1
                    col.attr["end_mask"].i = 1
                    col.attr["ellipsis_mask"].i = 0
                    col.attr["new_axis_mask"].i = 0
                    col.attr["shrink_axis_mask"].i = 0
                    last = col.name
                c32 = gd.node.add(name=base + "/cast", op="Cast")
                c32.input.append(last)
                c32.attr["SrcT"].type = dtype
                c32.attr["DstT"].type = P.DT_INT32
                _const(gd, base + "/none", np.asarray([-1], np.int32))
                cat = gd.node.add(name=base + "/cat", op="ConcatV2")
                cat.input.extend([c32.name, base + "/none", "FeatureColumnProcess/symbols/axis"])
                cat.attr["N"].i = 2
                cat.attr["T"].type = P.DT_INT32
                cat.attr["Tidx"].type = P.DT_INT32
                mx = gd.node.add(name=base + "/max", op="Max")
                mx.input.extend([cat.name, "FeatureColumnProcess/symbols/axis"])
                mx.attr["T"].type = P.DT_INT32
                mx.attr["Tidx"].type = P.DT_INT32
                mx.attr["keep_dims"].b = False
                _const(gd, base + "/one", np.asarray(1, np.int32))
                add = gd.node.add(name=base, op="AddV2")
                add.input.extend([mx.name, base + "/one"])
                add.attr["T"].type = P.DT_INT32
                pack.input.append(add.name)
                continue
            _const(gd, base + "/index", np.asarray(sym.index, np.int32))
            pick = gd.node.add(name=base + "/pick", op="GatherV2")
            pick.input.extend([flat.name, base + "/index", "FeatureColumnProcess/symbols/axis"])
            pick.attr["Tparams"].type = dtype
            pick.attr["Tindices"].type = P.DT_INT32
            pick.attr["Taxis"].type = P.DT_INT32
            pick.attr["batch_dims"].i = 0
            cast = gd.node.add(name=base, op="Cast")
            cast.input.append(pick.name)
            cast.attr["SrcT"].type = dtype
            cast.attr["DstT"].type = P.DT_INT32
            pack.input.append(cast.name)
        fuse.op = "Addons>FeatureColumnProcessWithSymbols"
        fuse.input.append(pack.name)

    removed: List[str] = []
    nodes = {n.name: n for n in gd.node}
    for gi in built.groups:
        orig = nodes[gi.concat_node]
        host_pos = [pos for pos, col in enumerate(gi.columns) if col not in out_index]
        new = gd.node.add(op="Addons>ConcatOutputs" if host_pos else "Addons>ConcatOutputsNoHost")
        a = new.attr
        a["T"].type = gi.dtype
        a["BLOCK_THREADS"].i = BLOCK_THREADS
        first = out_index[next(col for col in gi.columns if col in out_index)]
        a["prefix_begin"].i = 2 * first                 # index into output_shapes (rank 2 per output)
        a["prefix_end"].i = 2 * first + 1
        a["output_dir"].s = b""
        a["N"].i = len(host_pos)
        a["host_concat_indices"].list.SetInParent()
        new.input.extend(["FeatureColumnProcess", "FeatureColumnProcess:1"])
        for pos, col in enumerate(gi.columns):
            if col in out_index:
                a["device_concat_indices"].list.i.append(pos)
                a["device_input_indices"].list.i.append(out_index[col])
            else:                                       # the original concat input, untouched (:2597-2602)
                a["host_concat_indices"].list.i.append(pos)
                new.input.append(built.columns[col].concat_input)
            a["embedd_dims"].list.i.append(built.spec.columns[col].dim)
        # tensor_buffers: keep blob, tables and arena alive until the concat output is consumed
        new.input.append("ConcatInputs")
        a["buffer_types"].list.type.append(P.DT_INT8)
        for tensor, dtype, _ in built.device_inputs:
            new.input.append(tensor)
            a["buffer_types"].list.type.append(dtype)
        new.input.append("FeatureColumnProcess:2")
        a["buffer_types"].list.type.append(P.DT_INT8)
        new.name = orig.name
        orig.name = orig.name + "_removed"
        removed.append(orig.name)

    if prune:
        _prune(gd, removed)
    return gd


def _prune(gd, roots: List[str]) -> None:
    """Drop the removed concats and every node that only fed them."""
    nodes = {n.name: n for n in gd.node}
    uses = {n.name: 0 for n in gd.node}
    for n in gd.node:
        for t in n.input:
            uses[split_tensor(t)[0]] += 1
    dead, stack = set(), list(roots)
    while stack:
        name = stack.pop()
        if name in dead or uses[name] != 0:
            continue
        if nodes[name].op in ("Placeholder", "VariableV2", "VarHandleOp"):
            continue                                     # graph interface stays
        dead.add(name)
        for t in nodes[name].input:
            src = split_tensor(t)[0]
            uses[src] -= 1
            stack.append(src)
    keep = [n for n in gd.node if n.name not in dead]
    del gd.node[:]
    gd.node.extend(keep)
