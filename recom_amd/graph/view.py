"""Read-only view of a GraphDef: name → node, consumers, tensor names, constant
folding of ``Const`` nodes, output dtypes and the little static-shape knowledge the
plan builder needs (table shapes, host-input ranks, the inner dimension of a
``[nnz, k]`` index matrix).

The reference gets the same facts from ``GraphInfo`` (``graph_info.cc:153-207``:
``node_mapping`` / ``out_mapping``) and from its SymEngine shape inference
(``symbolic_shape/``); only rank and a few static dimensions are needed here because
row counts are resolved per request from the shapes ConcatInputs ships.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import tf_proto as P

NP_OF_DT = {P.DT_FLOAT: np.float32, P.DT_DOUBLE: np.float64, P.DT_INT32: np.int32, P.DT_INT64: np.int64,
            P.DT_BOOL: np.bool_, P.DT_INT8: np.int8, 4: np.uint8, 5: np.int16}
DT_OF_NP = {np.dtype(v): k for k, v in NP_OF_DT.items()}
Shape = Optional[List[Optional[int]]]


def split_tensor(t: str) -> Tuple[str, int]:
    """``"node:2"`` → ``("node", 2)``; ``"node"`` → ``("node", 0)``
    (``GetNodeNameByTensor`` / ``GetOutputIdxByTensor`` of the reference's utils)."""
    if t.startswith("^"):
        return t[1:], -1
    name, _, port = t.partition(":")
    return name, int(port) if port else 0


def tensor_name(node: str, port: int = 0) -> str:
    return node if port == 0 else f"{node}:{port}"


def tensor_to_numpy(t) -> np.ndarray:
    """``TensorProto`` → ndarray (``tensor_content`` or the typed ``*_val`` lists, a
    short list repeating its last value — TensorFlow's MakeNdarray rule)."""
    dt = NP_OF_DT.get(t.dtype)
    if dt is None:
        raise ValueError(f"unsupported tensor dtype {t.dtype}")
    shape = [int(d.size) for d in t.tensor_shape.dim]
    n = int(np.prod(shape)) if shape else 1
    if t.tensor_content:
        return np.frombuffer(t.tensor_content, dtype=dt).reshape(shape).copy()
    vals = {P.DT_FLOAT: t.float_val, P.DT_DOUBLE: t.double_val, P.DT_INT32: t.int_val, P.DT_INT64: t.int64_val,
            P.DT_BOOL: t.bool_val, P.DT_INT8: t.int_val, 4: t.int_val, 5: t.int_val}[t.dtype]
    a = np.asarray(list(vals), dtype=dt)
    if a.size == 0:
        a = np.zeros(1, dt)
    if a.size < n:
        a = np.concatenate([a, np.full(n - a.size, a[-1], dt)])
    return a[:n].reshape(shape)


def numpy_to_tensor(a: np.ndarray, t=None):
    a = np.asarray(a)
    t = t if t is not None else P.TensorProto()
    t.dtype = DT_OF_NP[a.dtype]
    for d in a.shape:
        t.tensor_shape.dim.add(size=int(d))
    t.tensor_content = np.ascontiguousarray(a).tobytes()
    return t


class GraphView:
    def __init__(self, gd) -> None:
        self.gd = gd
        self.nodes: Dict[str, object] = {}
        self.order: Dict[str, int] = {}
        self.consumers: Dict[str, List[Tuple[object, int]]] = defaultdict(list)
        for k, n in enumerate(gd.node):
            if n.name in self.nodes:
                raise ValueError(f"duplicate node name {n.name}")
            self.nodes[n.name] = n
            self.order[n.name] = k
        for n in gd.node:
            for i, t in enumerate(n.input):
                src, port = split_tensor(t)
                if src not in self.nodes:
                    raise ValueError(f"node {n.name}: input {t} not in graph")
                self.consumers[src].append((n, i if port >= 0 else -1))

    # ---- topology -------------------------------------------------------------
    def data_inputs(self, node) -> List[str]:
        return [t for t in node.input if not t.startswith("^")]

    def input(self, node, i: int):
        """(producer node, producer port) of data input ``i``."""
        name, port = split_tensor(self.data_inputs(node)[i])
        return self.nodes[name], port

    def data_consumers(self, name: str) -> List[Tuple[object, int]]:
        return [(n, i) for n, i in self.consumers.get(name, []) if i >= 0]

    # ---- constants ------------------------------------------------------------
    MAX_CONST_ELEMENTS = 1 << 20   # the walk reads axes, slice bounds, reshape targets; larger Consts are not looked at (as fcp_graph.cc)

    def const_array(self, node, port: int = 0) -> Optional[np.ndarray]:
        while node.op in ("Identity", "StopGradient") and port == 0:
            node, port = self.input(node, 0)
        if node.op != "Const" or port != 0:
            return None
        t = node.attr["value"].tensor
        n = 1
        for d in t.tensor_shape.dim:
            if d.size < 0 or (d.size and n > self.MAX_CONST_ELEMENTS // int(d.size)):
                return None
            n *= int(d.size)
        return tensor_to_numpy(t)

    # ---- dtypes ---------------------------------------------------------------
    def out_dtype(self, node, port: int = 0) -> int:
        a = node.attr
        op = node.op
        key = {"Placeholder": "dtype", "Const": "dtype", "VariableV2": "dtype", "GatherV2": "Tparams",
               "ResourceGather": "dtype", "ReadVariableOp": "dtype", "Cast": "DstT", "Shape": "out_type",
               "Size": "out_type"}.get(op)
        if key:
            return a[key].type
        if op == "Bucketize":
            return P.DT_INT32
        if op in ("SparseReshape", "StringToHashBucketFast"):
            return P.DT_INT64
        if op == "AsString":
            return P.DT_STRING
        if op in ("Unique",):
            return a["T"].type if port == 0 else a["out_idx"].type
        if "T" in a:
            return a["T"].type
        if "dtype" in a:
            return a["dtype"].type
        raise ValueError(f"cannot tell output dtype of {node.name} ({op})")

    # ---- static shapes (partial) ----------------------------------------------
    def static_shape(self, node, port: int = 0, _depth: int = 0) -> Shape:
        """List of dims (None = dynamic), or None when even the rank is unknown."""
        if "_output_shapes" in node.attr and len(node.attr["_output_shapes"].list.shape) > port:
            s = node.attr["_output_shapes"].list.shape[port]
            if not s.unknown_rank:
                return [int(d.size) if d.size >= 0 else None for d in s.dim]
        if _depth > 64:
            return None
        op = node.op
        if op in ("Placeholder", "VariableV2") and "shape" in node.attr:
            s = node.attr["shape"].shape
            return None if s.unknown_rank else [int(d.size) if d.size >= 0 else None for d in s.dim]
        if op == "Const":
            return [int(d.size) for d in node.attr["value"].tensor.tensor_shape.dim]
        if op in ("Identity", "Cast", "Bucketize", "StopGradient", "ZerosLike", "AsString", "StringToHashBucketFast"):
            return self.static_shape(*self.input(node, 0), _depth + 1)
        if op == "Reshape":
            tgt = self.const_array(*self.input(node, 1))
            if tgt is None:
                return None
            dims: List[Optional[int]] = [int(d) if d >= 0 else None for d in tgt.reshape(-1)]
            src = self.static_shape(*self.input(node, 0), _depth + 1)
            if dims.count(None) == 1 and src is not None and None not in src:
                known = int(np.prod([d for d in dims if d is not None])) if len(dims) > 1 else 1
                if known:
                    dims[dims.index(None)] = int(np.prod(src)) // known
            return dims
        if op == "ExpandDims":
            src = self.static_shape(*self.input(node, 0), _depth + 1)
            ax = self.const_array(*self.input(node, 1))
            if src is None or ax is None:
                return None
            ax = int(ax.reshape(-1)[0])
            ax = ax + len(src) + 1 if ax < 0 else ax
            return src[:ax] + [1] + src[ax:]
        if op == "Squeeze":
            src = self.static_shape(*self.input(node, 0), _depth + 1)
            if src is None:
                return None
            dims = [int(d) for d in node.attr["squeeze_dims"].list.i]
            dims = [d + len(src) if d < 0 else d for d in dims]
            if not dims:
                if None in src:
                    return None
                return [d for d in src if d != 1]
            return [d for k, d in enumerate(src) if k not in dims]
        if op == "StridedSlice":
            src = self.static_shape(*self.input(node, 0), _depth + 1)
            spec = self.strided_slice_spec(node)
            if src is None or spec is None or len(spec["begin"]) > len(src):
                return None
            out: List[Optional[int]] = []
            for k, d in enumerate(src):
                if k >= len(spec["begin"]):
                    out.append(d)
                    continue
                if spec["shrink_axis_mask"] >> k & 1:
                    continue
                b, e, s = spec["begin"][k], spec["end"][k], spec["strides"][k]
                bm, em = spec["begin_mask"] >> k & 1, spec["end_mask"] >> k & 1
                if s != 1:
                    return None
                if bm and em:
                    out.append(d)
                elif d is None and (bm or em or b < 0 or e < 0):
                    out.append(None)
                else:
                    lo = 0 if bm else (b + d if b < 0 else b)
                    hi = d if em else (e + d if e < 0 else e)
                    out.append(max(0, min(hi, d if d is not None else hi) - lo))
            return out
        if op in ("VarHandleOp", "ReadVariableOp"):                 # the variable's shape (attr of the handle op)
            h = node if op == "VarHandleOp" else self.input(node, 0)[0]
            if h.op != "VarHandleOp" or h.attr["shape"].shape.unknown_rank:
                return None
            return [int(d.size) if d.size >= 0 else None for d in h.attr["shape"].shape.dim]
        if op in ("GatherV2", "ResourceGather"):
            p = self.static_shape(*self.input(node, 0), _depth + 1)
            i = self.static_shape(*self.input(node, 1), _depth + 1)
            return None if p is None or i is None else i + p[1:]
        if op == "Addons>SelectValue":
            return self.static_shape(*self.input(node, 0), _depth + 1)
        if op == "Addons>GatherIndiceValue":                     # (surviving indices [n, k], surviving values [n])
            idx = self.static_shape(*self.input(node, 0), _depth + 1)
            return [None, idx[1] if idx and len(idx) == 2 else None] if port == 0 else [None]
        if op == "Addons>GatherValueGenIndice":                  # (indices [n, 1], surviving values [n])
            return [None, 1] if port == 0 else [None]
        if op == "SparseReshape":
            # (output_indices [nnz, rank(new_shape)], output_shape [rank(new_shape)])
            idx = self.static_shape(*self.input(node, 0), _depth + 1)
            new = self.static_shape(*self.input(node, 2), _depth + 1)
            if new is None or len(new) != 1 or new[0] is None:
                return None
            return [idx[0] if idx else None, new[0]] if port == 0 else [new[0]]
        if op == "Pack":
            n = len(self.data_inputs(node))
            first = self.static_shape(*self.input(node, 0), _depth + 1)
            axis = int(node.attr["axis"].i) if "axis" in node.attr else 0
            return None if first is None or axis != 0 else [n] + first
        if op in ("Prod", "Sum", "Max", "Min"):
            src = self.static_shape(*self.input(node, 0), _depth + 1)
            ax = self.const_array(*self.input(node, 1))
            if src is None or ax is None or ("keep_dims" in node.attr and node.attr["keep_dims"].b):
                return None
            axes = {int(a) + len(src) if int(a) < 0 else int(a) for a in ax.reshape(-1)}
            return [d for k, d in enumerate(src) if k not in axes]
        return None

    # ---- where does one element of a small integer tensor come from? ----------------
    def elem_source(self, node, port: int, k: int, _depth: int = 0):
        """Element ``k`` (flat index) of a shape-like tensor, traced to ``("const", value)`` or
        ``("elem", tensor, index)`` — a plain copy of one element of another tensor — or None when
        it is computed.  Lets the plan builder prove two shape entries equal without a symbolic
        engine (the reference asks SymEngine, ``symbolic_shape/``)."""
        if _depth > 32:
            return None
        op = node.op
        if op == "Const" and port == 0:
            v = tensor_to_numpy(node.attr["value"].tensor).reshape(-1)
            return ("const", int(v[k])) if k < v.size else None
        if op in ("Identity", "StopGradient", "Reshape", "Squeeze", "ExpandDims") and port == 0:
            return self.elem_source(*self.input(node, 0), k, _depth + 1)       # flat order unchanged
        if op == "Cast" and port == 0:
            ints = (P.DT_INT32, P.DT_INT64)
            if node.attr["SrcT"].type in ints and node.attr["DstT"].type in ints:
                return self.elem_source(*self.input(node, 0), k, _depth + 1)
            return None
        if op == "Pack" and port == 0:
            if ("axis" in node.attr and node.attr["axis"].i != 0) or k >= len(self.data_inputs(node)):
                return None
            src, sp = self.input(node, k)
            return self.elem_source(src, sp, 0, _depth + 1) if self.static_shape(src, sp) == [] else None
        if op == "ConcatV2" and port == 0:
            ins = self.data_inputs(node)
            for i in range(len(ins) - 1):
                src, sp = self.input(node, i)
                shape = self.static_shape(src, sp)
                if shape is None or len(shape) != 1 or shape[0] is None:
                    return None
                if k < shape[0]:
                    return self.elem_source(src, sp, k, _depth + 1)
                k -= shape[0]
            return None
        if op == "StridedSlice" and port == 0:
            spec = self.strided_slice_spec(node)
            src, sp = self.input(node, 0)
            shape = self.static_shape(src, sp)
            if spec is None or shape is None or len(shape) != 1 or len(spec["begin"]) != 1 or spec["strides"] != [1]:
                return None
            if spec["ellipsis_mask"] or spec["new_axis_mask"]:
                return None
            b = 0 if spec["begin_mask"] & 1 else spec["begin"][0]
            if b < 0:
                if shape[0] is None:
                    return None
                b += shape[0]
            return self.elem_source(src, sp, b + k, _depth + 1)
        if op == "GatherV2" and port == 0:
            idx = self.const_array(*self.input(node, 1))
            ax = self.const_array(*self.input(node, 2))
            src, sp = self.input(node, 0)
            shape = self.static_shape(src, sp)
            if idx is None or ax is None or int(ax.reshape(-1)[0]) != 0 or shape is None or len(shape) != 1:
                return None
            i = int(idx.reshape(-1)[k]) if k < idx.size else None
            if i is None or (i < 0 and shape[0] is None):
                return None
            return self.elem_source(src, sp, i + shape[0] if i < 0 else i, _depth + 1)
        if op in ("Prod", "Sum", "Max", "Min") and port == 0:
            src, sp = self.input(node, 0)
            if self.static_shape(src, sp) == [1] and k == 0:                   # a reduction of one element is that element
                return self.elem_source(src, sp, 0, _depth + 1)
            return None
        if op == "SparseReshape" and port == 1:                                # output_shape = new_shape (no -1 handled)
            return self.elem_source(*self.input(node, 2), k, _depth + 1)
        if op in ("Placeholder", "PlaceholderWithDefault") or port != 0 or not self.data_inputs(node):
            return ("elem", tensor_name(node.name, port), k)
        return ("elem", tensor_name(node.name, port), k) if op in ("Shape",) else None

    def strided_slice_spec(self, node) -> Optional[dict]:
        vals = [self.const_array(*self.input(node, k)) for k in (1, 2, 3)]
        if any(v is None for v in vals):
            return None

        def mask(key: str) -> int:
            return int(node.attr[key].i) if key in node.attr else 0

        return {"begin": [int(x) for x in vals[0].reshape(-1)], "end": [int(x) for x in vals[1].reshape(-1)],
                "strides": [int(x) for x in vals[2].reshape(-1)], "begin_mask": mask("begin_mask"),
                "end_mask": mask("end_mask"), "ellipsis_mask": mask("ellipsis_mask"),
                "new_axis_mask": mask("new_axis_mask"), "shrink_axis_mask": mask("shrink_axis_mask")}
