"""NumPy evaluator for the TensorFlow ops a feature-column graph is made of.

TEST INFRASTRUCTURE (like everything under ``oracle/``): it gives the plan builder's
tests the answer TF-CPU would give on the *original* graph, so that

    original GraphDef  --this evaluator-->  ConcatV2 output
    original GraphDef  --plan builder + rewrite-->  Addons> ops --HIP path / C oracle-->  same output

can be compared without TensorFlow (absent from this image; the reference pins TF
2.6.2).  Each op restates the TF op's documented semantics; the reference sites that
rely on them are cited.  fp32 pooling adds in id order, like TF-CPU's
SparseSegment kernels and like ``fcp_oracle.c`` — so results can be compared bit for
bit with the oracle and, by the product's design, with the HIP path.

Parity unpinned (no TF run, no reference-owned vectors); see DESIGN.md §7.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

_NP = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_, 6: np.int8, 4: np.uint8, 5: np.int16}


def _split(t: str) -> Tuple[str, int]:
    name, _, port = t.partition(":")
    return name, int(port) if port else 0


def tensor_value(t) -> np.ndarray:
    dt = _NP[t.dtype]
    shape = [int(d.size) for d in t.tensor_shape.dim]
    n = int(np.prod(shape)) if shape else 1
    if t.tensor_content:
        return np.frombuffer(t.tensor_content, dt).reshape(shape).copy()
    vals = {1: t.float_val, 2: t.double_val, 3: t.int_val, 9: t.int64_val, 10: t.bool_val}.get(t.dtype, t.int_val)
    a = np.asarray(list(vals), dt)
    if a.size == 0:
        a = np.zeros(1, dt)
    if a.size < n:
        a = np.concatenate([a, np.full(n - a.size, a[-1], dt)])
    return a[:n].reshape(shape)


def _strided_slice(x, begin, end, strides, a):
    """tf.strided_slice without ellipsis / new axis (the masks the reference's graphs
    use: lookup_optimizer.cc:237-242, 407-411)."""
    def m(k):
        return int(a[k].i) if k in a else 0
    if m("ellipsis_mask") or m("new_axis_mask"):
        raise NotImplementedError("StridedSlice ellipsis/new_axis")
    idx = []
    for k in range(len(begin)):
        if m("shrink_axis_mask") >> k & 1:
            idx.append(int(begin[k]))
            continue
        b = None if m("begin_mask") >> k & 1 else int(begin[k])
        e = None if m("end_mask") >> k & 1 else int(end[k])
        idx.append(slice(b, e, int(strides[k])))
    return x[tuple(idx)]


def _sparse_segment(table, ids, seg, num_segments: Optional[int], mean: bool) -> np.ndarray:
    """tf.sparse.segment_{sum,mean}[_with_num_segments]: rows of `table` selected by
    `ids`, added per sorted segment id in id order (fp32), missing segments = 0;
    mean divides by the segment's id count (reference: cuda_emitter.cc:402-661)."""
    ids = np.asarray(ids).astype(np.int64).ravel()
    seg = np.asarray(seg).astype(np.int64).ravel()
    if np.any(np.diff(seg) < 0):
        raise ValueError("segment ids must be sorted")
    n = int(num_segments) if num_segments is not None else (int(seg[-1]) + 1 if seg.size else 0)
    out = np.zeros((n, table.shape[1]), np.float32)
    cnt = np.zeros(n, np.int64)
    for i in range(ids.size):
        out[seg[i]] = out[seg[i]] + table[ids[i]]
        cnt[seg[i]] += 1
    if mean:
        nz = cnt > 0
        out[nz] = out[nz] / cnt[nz, None].astype(np.float32)
    return out


class GraphEvaluator:
    """``custom_ops[op](node, inputs) -> list of outputs`` supplies the ``Addons>`` ops."""

    def __init__(self, graph_def, variables: Optional[Dict[str, np.ndarray]] = None,
                 custom_ops: Optional[Dict[str, Callable]] = None) -> None:
        self.nodes = {n.name: n for n in graph_def.node}
        self.variables = variables or {}
        self.custom = custom_ops or {}

    def run(self, fetches: List[str], feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        cache: Dict[str, List[np.ndarray]] = {}
        for k, v in feeds.items():
            name, port = _split(k)
            cache.setdefault(name, [None] * (port + 1))[port] = np.asarray(v)
        return [self._tensor(f, cache) for f in fetches]

    def _tensor(self, t: str, cache) -> np.ndarray:
        name, port = _split(t)
        if name not in cache:
            # iterative post-order evaluation (graphs with thousands of columns are deep only by width)
            stack = [name]
            while stack:
                cur = stack[-1]
                if cur in cache:
                    stack.pop()
                    continue
                node = self.nodes[cur]
                missing = [_split(i)[0] for i in node.input if not i.startswith("^") and _split(i)[0] not in cache]
                if missing:
                    stack.extend(missing)
                    continue
                ins = [cache[_split(i)[0]][_split(i)[1]] for i in node.input if not i.startswith("^")]
                cache[cur] = self._eval(node, ins)
                stack.pop()
        return cache[name][port]

    def _eval(self, node, x: List[np.ndarray]) -> List[np.ndarray]:
        op, a = node.op, node.attr
        if op in self.custom:
            return list(self.custom[op](node, x))
        if op == "Placeholder":
            raise KeyError(f"placeholder {node.name} was not fed")
        if op == "Const":
            return [tensor_value(a["value"].tensor)]
        if op == "VariableV2":
            return [np.asarray(self.variables[node.name])]
        if op == "VarHandleOp":                        # a resource handle: stands for the variable's name
            return [node.name]
        if op == "ReadVariableOp":
            return [np.asarray(self.variables[x[0]])]
        if op == "ResourceGather":
            return [np.take(np.asarray(self.variables[x[0]]), x[1].astype(np.int64), axis=0)]
        if op in ("Identity", "StopGradient"):
            return [x[0]]
        if op == "GatherV2":
            return [np.take(x[0], x[1].astype(np.int64), axis=int(x[2]))]
        if op == "Bucketize":
            b = np.asarray(list(a["boundaries"].list.f), np.float32)
            return [np.searchsorted(b, x[0].astype(np.float32), side="right").astype(np.int32)]
        if op == "Cast":
            return [x[0].astype(_NP[a["DstT"].type])]
        if op == "Reshape":
            return [x[0].reshape([int(d) for d in x[1].ravel()])]
        if op == "Squeeze":
            dims = tuple(int(d) for d in a["squeeze_dims"].list.i)
            return [np.squeeze(x[0], axis=dims) if dims else np.squeeze(x[0])]
        if op == "ExpandDims":
            return [np.expand_dims(x[0], int(x[1]))]
        if op == "StridedSlice":
            return [_strided_slice(x[0], x[1].ravel(), x[2].ravel(), x[3].ravel(), a)]
        if op == "ConcatV2":
            return [np.concatenate(x[:-1], axis=int(x[-1]))]
        if op == "Sum":
            # fp32, in row order — BatchColReduction's order (cuda_emitter.cc:1217-1239)
            axis = int(np.asarray(x[1]).ravel()[0])
            moved = np.moveaxis(x[0].astype(np.float32), axis, 0)
            acc = np.zeros(moved.shape[1:], np.float32)
            for r in range(moved.shape[0]):
                acc = acc + moved[r]
            return [np.expand_dims(acc, axis) if ("keep_dims" in a and a["keep_dims"].b) else acc]
        if op == "Pack":
            return [np.stack(x, axis=int(a["axis"].i) if "axis" in a else 0)]
        if op == "Shape":
            return [np.asarray(x[0].shape, _NP[a["out_type"].type] if "out_type" in a else np.int32)]
        if op == "ScatterNd":
            # tf.scatter_nd with [n, 1] indices: rows of `updates` land in a zero tensor
            # (reference: GatherScatterRows, cuda_emitter.cc:296-345; ≤ 1 update per row)
            shape = [int(d) for d in x[2].ravel()]
            out = np.zeros(shape, x[1].dtype)
            np.add.at(out, x[0].astype(np.int64).reshape(-1), x[1])
            return [out]
        if op in ("Addons>SelectValue", "Addons>GatherIndiceValue", "Addons>GatherValueGenIndice"):
            # the reference's CPU id ops (custom_ops/select_value, gather_indice_value, gather_value_gen_indice) with the
            # INTENDED interval test lo <= x && x <= hi (their `x >= l || x <= r` accepts everything)
            lo = [int(v) for v in a["left_boundaries"].list.i]
            hi = [int(v) for v in a["right_boundaries"].list.i]
            values = x[-1]
            inside = np.zeros(values.shape, bool)
            for l, h in zip(lo, hi):
                inside |= (values >= l) & (values <= h)
            if op == "Addons>SelectValue":
                return [np.where(inside, values, values.dtype.type(int(a["substitute"].i)))]
            keep = inside.ravel()
            if op == "Addons>GatherIndiceValue":
                return [x[0].reshape(keep.size, -1)[keep], values.ravel()[keep]]
            return [np.nonzero(keep)[0].astype(values.dtype).reshape(-1, 1), values.ravel()[keep]]
        if op == "AsString":
            # integers only (what categorical_column_with_hash_bucket(dtype=int64) feeds): plain decimal
            assert x[0].dtype.kind in "iu", "AsString: integer tensors only"
            out = np.empty(x[0].shape, object)
            out.ravel()[:] = [str(int(v)).encode() for v in x[0].ravel()]
            return [out]
        if op == "StringToHashBucketFast":
            # Fingerprint64(s) % num_buckets (TF core/kernels/string_to_hash_bucket_fast_op.h)
            from fcp_oracle import np_fingerprint64
            nb = int(a["num_buckets"].i)
            out = np.empty(x[0].shape, np.int64)
            out.ravel()[:] = [np_fingerprint64(bytes(v)) % nb for v in x[0].ravel()]
            return [out]
        if op == "Max":
            axis = tuple(int(v) for v in np.asarray(x[1]).ravel())
            return [np.max(x[0], axis=axis, keepdims=bool("keep_dims" in a and a["keep_dims"].b)).astype(x[0].dtype)]
        if op in ("AddV2", "Add"):
            return [(x[0] + x[1]).astype(x[0].dtype)]
        if op == "Prod":
            axis = tuple(int(v) for v in np.asarray(x[1]).ravel())
            return [np.prod(x[0], axis=axis, keepdims=bool("keep_dims" in a and a["keep_dims"].b)).astype(x[0].dtype)]
        if op == "SparseReshape":
            # tf.sparse.reshape: row-major flat position of every index, re-expressed in the new shape
            shape = [int(d) for d in x[1].ravel()]
            new = [int(d) for d in x[2].ravel()]
            if -1 in new:
                new[new.index(-1)] = int(np.prod(shape)) // max(1, -int(np.prod(new)))
            assert int(np.prod(new)) == int(np.prod(shape)), "SparseReshape: element counts differ"
            idx = x[0].astype(np.int64).reshape(-1, len(shape))
            flat = np.ravel_multi_index(tuple(idx.T), shape) if idx.size else np.zeros(0, np.int64)
            out = np.stack(np.unravel_index(flat, new), axis=1).astype(np.int64) if idx.size else np.zeros((0, len(new)), np.int64)
            return [out, np.asarray(new, np.int64)]
        if op == "SparseSegmentSqrtN":                      # (a combiner the fused path leaves to TensorFlow)
            out = _sparse_segment(x[0], x[1], x[2], None, False)
            cnt = np.bincount(np.asarray(x[2]).astype(np.int64).ravel(), minlength=out.shape[0]).astype(np.float32)
            nz = cnt > 0
            out[nz] = out[nz] / np.sqrt(cnt[nz])[:, None]
            return [out]
        if op.startswith("SparseSegmentSum") or op.startswith("SparseSegmentMean"):
            n = int(np.asarray(x[3]).ravel()[0]) if op.endswith("WithNumSegments") else None
            return [_sparse_segment(x[0], x[1], x[2], n, "Mean" in op)]
        raise NotImplementedError(f"op {op} ({node.name})")
