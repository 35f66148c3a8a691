"""Synthetic GraphDefs in the canonical post-LookupOptimizer form the reference's
emitter consumes (``lookup_optimizer.cc:157-440``), built with the run-time GraphDef
messages of ``recom_amd.graph.tf_proto`` (TensorFlow is absent here, so no real
SavedModel can be exported; node patterns follow what the reference's rewrites emit)."""
import numpy as np

from recom_amd.graph import tf_proto as P
from recom_amd.graph.view import numpy_to_tensor

DT = {np.dtype(np.float32): P.DT_FLOAT, np.dtype(np.int32): P.DT_INT32, np.dtype(np.int64): P.DT_INT64}


class GB:
    def __init__(self):
        self.gd = P.GraphDef()
        self.gd.versions.producer = 808  # TF 2.6 graph version
        self._consts = {}

    def node(self, name, op, inputs=(), **attrs):
        n = self.gd.node.add(name=name, op=op)
        n.input.extend(inputs)
        for k, v in attrs.items():
            a = n.attr[k]
            if isinstance(v, bool):
                a.b = v
            elif isinstance(v, int):
                a.i = v
            elif isinstance(v, tuple) and v[0] == "type":
                a.type = v[1]
            elif isinstance(v, tuple) and v[0] == "shape":
                for d in v[1]:
                    a.shape.dim.add(size=d)
            elif isinstance(v, tuple) and v[0] == "floats":
                a.list.f.extend(v[1])
            elif isinstance(v, tuple) and v[0] == "ints":
                a.list.i.extend(v[1])
            elif isinstance(v, tuple) and v[0] == "shapes":
                for s in v[1]:
                    sh = a.list.shape.add()
                    for d in s:
                        sh.dim.add(size=d)
            else:
                raise TypeError((k, v))
        return name

    def placeholder(self, name, dtype, shape):
        return self.node(name, "Placeholder", dtype=("type", DT[np.dtype(dtype)]), shape=("shape", shape))

    def const(self, name, value):
        value = np.asarray(value)
        n = self.gd.node.add(name=name, op="Const")
        n.attr["dtype"].type = DT[value.dtype]
        numpy_to_tensor(value, n.attr["value"].tensor)
        return name

    def variable(self, name, vocab, dim):
        return self.node(name, "VariableV2", dtype=("type", P.DT_FLOAT), shape=("shape", [vocab, dim]))

    def gather(self, name, table, ids, ids_dtype):
        zero = self.const(name + "/zero", np.asarray(0, np.int32))
        return self.node(name, "GatherV2", [table, ids, zero], Tparams=("type", P.DT_FLOAT),
                         Tindices=("type", DT[np.dtype(ids_dtype)]), Taxis=("type", P.DT_INT32), batch_dims=0)

    def slice_col0(self, name, indices, shrink):
        """indices[:, 0] (shrink, lookup_optimizer.cc:229-243) or indices[:, 0:1] (:399-412)."""
        b = self.const(name + "/begin_node", np.asarray([0, 0], np.int64))
        e = self.const(name + "/end_node", np.asarray([0, 1], np.int64))
        s = self.const(name + "/stride_node", np.asarray([1, 1], np.int64))
        extra = {"shrink_axis_mask": 2} if shrink else {}
        return self.node(name, "StridedSlice", [indices, b, e, s], T=("type", P.DT_INT64), Index=("type", P.DT_INT64),
                         begin_mask=1, end_mask=1, **extra)


MICRO_BOUNDARIES = [float(x) for x in range(0, 500, 5)]  # microbenchmark.py:46


def canonical_model(B=19, seed=0, unsupported=False):
    """Returns (graph_def, feeds, variables, fetches).  Two concat groups; every column
    form; shared table and shared ids; trailing reshapes; a second prefix size."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables = {}, {}

    def table(name, vocab, dim):
        g.variable(name, vocab, dim)
        variables[name] = rng.standard_normal((vocab, dim)).astype(np.float32)
        return name

    def sparse(prefix, rows, vocab, max_len, min_len=0):
        lens = rng.integers(min_len, max_len + 1, size=rows)
        nnz = int(lens.sum())
        idx = np.stack([np.repeat(np.arange(rows), lens),
                        np.concatenate([np.arange(l) for l in lens]) if nnz else np.zeros(0, np.int64)], 1)
        g.placeholder(prefix + "/values", np.int64, [-1])
        g.placeholder(prefix + "/indices", np.int64, [-1, 2])
        g.placeholder(prefix + "/dense_shape", np.int64, [2])
        feeds[prefix + "/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[prefix + "/indices"] = idx.astype(np.int64).reshape(nnz, 2)
        feeds[prefix + "/dense_shape"] = np.asarray([rows, max(1, int(lens.max(initial=0)))], np.int64)
        # num_segments = dense_shape[0] (the reference builds it with ShapeConstruct; any scalar works)
        b = g.const(prefix + "/ns/b", np.asarray([0], np.int32))
        e = g.const(prefix + "/ns/e", np.asarray([1], np.int32))
        s = g.const(prefix + "/ns/s", np.asarray([1], np.int32))
        g.node(prefix + "/num_segments", "StridedSlice", [prefix + "/dense_shape", b, e, s], T=("type", P.DT_INT64),
               Index=("type", P.DT_INT32))
        g.node(prefix + "/num_segments_squeeze", "Squeeze", [prefix + "/num_segments"], T=("type", P.DT_INT64),
               squeeze_dims=("ints", [0]))
        return prefix + "/values", prefix + "/indices", prefix + "/num_segments_squeeze"

    concat0 = []
    # 1. form 1, int64 ids [B]
    t_a = table("input_layer/a_embedding/embedding_weights", 97, 8)
    g.placeholder("a_ids", np.int64, [-1])
    feeds["a_ids"] = rng.integers(0, 97, size=B).astype(np.int64)
    concat0.append(g.gather("input_layer/a_embedding/GatherDense", t_a, "a_ids", np.int64))
    # 2. form 1 through Bucketize → Cast → Reshape (reference microbenchmark column)
    t_b = table("input_layer/b_embedding/embedding_weights", 101, 8)
    g.placeholder("b_value", np.float32, [-1, 1])
    feeds["b_value"] = rng.uniform(-10, 510, size=(B, 1)).astype(np.float32)
    kat = np.float32([0.0, 495.0, -1.0, 5.0])[:B]      # exact boundary hits and below the first one
    feeds["b_value"][:kat.size, 0] = kat
    g.node("b/Bucketize", "Bucketize", ["b_value"], T=("type", P.DT_FLOAT), boundaries=("floats", MICRO_BOUNDARIES))
    g.node("b/Cast", "Cast", ["b/Bucketize"], SrcT=("type", P.DT_INT32), DstT=("type", P.DT_INT64))
    g.const("b/flat", np.asarray([-1], np.int32))
    g.node("b/Reshape", "Reshape", ["b/Cast", "b/flat"], T=("type", P.DT_INT64), Tshape=("type", P.DT_INT32))
    concat0.append(g.gather("input_layer/b_embedding/GatherDense", t_b, "b/Reshape", np.int64))
    # 3. form 1, int32 ids [B, 1] → [B, 1, 12] → trailing Reshape [-1, 12]; shares table with column 9
    t_c = table("input_layer/c_embedding/embedding_weights", 53, 12)
    g.placeholder("c_ids", np.int32, [-1, 1])
    feeds["c_ids"] = rng.integers(0, 53, size=(B, 1)).astype(np.int32)
    g.gather("input_layer/c_embedding/GatherDense", t_c, "c_ids", np.int32)
    g.const("c/out_shape", np.asarray([-1, 12], np.int32))
    concat0.append(g.node("c/Reshape", "Reshape", ["input_layer/c_embedding/GatherDense", "c/out_shape"],
                          T=("type", P.DT_FLOAT), Tshape=("type", P.DT_INT32)))
    # 4. form 2 mean, segment ids = indices[:, 0]
    t_d = table("input_layer/d_embedding/embedding_weights", 211, 16)
    v, i, n = sparse("d", B, 211, 6)
    seg = g.slice_col0("d/added_strided_slice", i, shrink=True)
    concat0.append(g.node("d/SparseSegmentMean_with_num_segments", "SparseSegmentMeanWithNumSegments",
                          [t_d, v, seg, n], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                          Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
    # 5. form 2 sum, dim 32 (> 20: the reference's "experiment" template), segment ids cast to int32
    t_e = table("input_layer/e_embedding/embedding_weights", 300, 32)
    v, i, n = sparse("e", B, 300, 9)
    seg = g.slice_col0("e/added_strided_slice", i, shrink=True)
    g.node("e/Cast", "Cast", [seg], SrcT=("type", P.DT_INT64), DstT=("type", P.DT_INT32))
    concat0.append(g.node("e/SparseSegmentSum_with_num_segments", "SparseSegmentSumWithNumSegments",
                          [t_e, v, "e/Cast", n], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                          Tsegmentids=("type", P.DT_INT32), Tnumsegments=("type", P.DT_INT64)))
    # 6. form 3: ScatterNd(indices[:, 0:1], GatherV2, [B, dim])
    t_f = table("input_layer/f_embedding/embedding_weights", 64, 4)
    v, i, n = sparse("f", B, 64, 1)
    rows = g.slice_col0("f/added_strided_slice", i, shrink=False)
    g.gather("input_layer/f_embedding/GatherScatter/Gather", t_f, v, np.int64)
    g.const("f/dim", np.asarray(4, np.int64))
    g.node("f/Scatter_shape", "Pack", [n, "f/dim"], N=2, T=("type", P.DT_INT64), axis=0)
    concat0.append(g.node("input_layer/f_embedding/GatherScatter/Scatter", "ScatterNd",
                          [rows, "input_layer/f_embedding/GatherScatter/Gather", "f/Scatter_shape"],
                          T=("type", P.DT_FLOAT), Tindices=("type", P.DT_INT64)))
    # 7. dense features straight into the concat (ConcatOutputs host input in the reference)
    g.placeholder("dense_features", np.float32, [-1, 13])
    feeds["dense_features"] = rng.standard_normal((B, 13)).astype(np.float32)
    concat0.append("dense_features")
    # 8. Sum(x, axis=1)
    g.placeholder("seq_features", np.float32, [-1, 3, 8])
    feeds["seq_features"] = rng.standard_normal((B, 3, 8)).astype(np.float32)
    g.const("seq/axis", np.asarray(1, np.int32))
    concat0.append(g.node("seq/Sum", "Sum", ["seq_features", "seq/axis"], T=("type", P.DT_FLOAT),
                          Tidx=("type", P.DT_INT32), keep_dims=False))
    # 9. second lookup into table c with the ids of column 1 (shared table, shared host input)
    g.node("c2/Cast", "Cast", ["a_ids"], SrcT=("type", P.DT_INT64), DstT=("type", P.DT_INT32))
    feeds["a_ids"] = feeds["a_ids"] % 53
    concat0.append(g.gather("input_layer/c2_embedding/GatherDense", t_c, "c2/Cast", np.int32))
    if unsupported:
        # 10. a pooling op the fused path does not take (sqrt-n combiner): stays in TF
        t_u = table("input_layer/u_embedding/embedding_weights", 40, 4)
        v, i, n = sparse("u", B, 40, 3, min_len=1)
        seg = g.slice_col0("u/added_strided_slice", i, shrink=True)
        concat0.append(g.node("u/SparseSegmentSqrtN", "SparseSegmentSqrtN", [t_u, v, seg], T=("type", P.DT_FLOAT),
                              Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64),
                              _output_shapes=("shapes", [[-1, 4]])))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", concat0 + ["concat/axis"], N=len(concat0), T=("type", P.DT_FLOAT),
           Tidx=("type", P.DT_INT32))

    # second concat with its own prefix size
    B2 = B + 5
    t_g = table("seq_layer/g_embedding/embedding_weights", 77, 8)
    v, i, n = sparse("g", B2, 77, 4)
    seg = g.slice_col0("g/added_strided_slice", i, shrink=True)
    g.node("g/SparseSegmentSum_with_num_segments", "SparseSegmentSumWithNumSegments", [t_g, v, seg, n],
           T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64),
           Tnumsegments=("type", P.DT_INT64))
    t_h = table("seq_layer/h_embedding/embedding_weights", 31, 20)
    g.placeholder("h_ids", np.int64, [-1])
    feeds["h_ids"] = rng.integers(0, 31, size=B2).astype(np.int64)
    g.gather("seq_layer/h_embedding/GatherDense", t_h, "h_ids", np.int64)
    g.const("concat2/axis", np.asarray(-1, np.int32))
    g.node("seq_layer/concat", "ConcatV2", ["g/SparseSegmentSum_with_num_segments",
                                            "seq_layer/h_embedding/GatherDense", "concat2/axis"],
           N=2, T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))

    # consumers of the concats + something unrelated that must survive the rewrite
    g.node("output_0", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    g.node("output_1", "Identity", ["seq_layer/concat"], T=("type", P.DT_FLOAT))
    g.node("dense_copy", "Identity", ["dense_features"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output_0", "output_1"]


def microbenchmark_model(columns=6, B=32, seed=0):
    """The reference micro-benchmark's model (microbenchmark.py:40-66) in rewritten form:
    N × {f32 value → Bucketize(0,5,…,495) → 101-row, dim-8 table → GatherV2}, one concat."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables, ins = {}, {}, []
    for c in range(columns):
        t = g.variable(f"input_layer/col{c}_embedding/embedding_weights", 101, 8)
        variables[t] = rng.standard_normal((101, 8)).astype(np.float32)
        g.placeholder(f"col{c}", np.float32, [-1, 1])
        feeds[f"col{c}"] = rng.integers(-1, 10000, size=(B, 1)).astype(np.float32)  # microbenchmark.py:66
        g.node(f"col{c}/Bucketize", "Bucketize", [f"col{c}"], T=("type", P.DT_FLOAT),
               boundaries=("floats", MICRO_BOUNDARIES))
        g.const(f"col{c}/flat", np.asarray([-1], np.int32))
        g.node(f"col{c}/Reshape", "Reshape", [f"col{c}/Bucketize", f"col{c}/flat"], T=("type", P.DT_INT32),
               Tshape=("type", P.DT_INT32))
        ins.append(g.gather(f"input_layer/col{c}_embedding/GatherDense", t, f"col{c}/Reshape", np.int32))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=columns, T=("type", P.DT_FLOAT),
           Tidx=("type", P.DT_INT32))
    g.node("output", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output"]


def random_model(seed):
    """Random rewritten graph: 1-3 concat groups (each with its own row count), 2-14 columns per
    group of random kind / dim / vocabulary / id dtype, occasionally sharing a table with the
    previous lookup of the same dim.  Returns (graph_def, feeds, variables, fetches, kinds)."""
    rng = np.random.default_rng(1000 + seed)
    g = GB()
    feeds, variables, fetches, kinds = {}, {}, [], []
    uid = [0]

    def name(s):
        uid[0] += 1
        return f"n{uid[0]}_{s}"

    last_table = {}

    def table(dim):
        if dim in last_table and rng.random() < 0.25:
            return last_table[dim]
        vocab = int(rng.integers(3, 400))
        t = g.variable(name("emb") + "/embedding_weights", vocab, dim)
        variables[t] = rng.standard_normal((vocab, dim)).astype(np.float32)
        last_table[dim] = (t, vocab)
        return t, vocab

    def sparse(rows, vocab, max_len):
        p = name("sp")
        lens = rng.integers(0, max_len + 1, size=rows)
        nnz = int(lens.sum())
        k = int(rng.integers(2, 4))                      # rank of the SparseTensor: indices [nnz, 2] or [nnz, 3]
        cols = [np.repeat(np.arange(rows), lens)] + [np.zeros(nnz, np.int64)] * (k - 1)
        g.placeholder(p + "/values", np.int64, [-1])
        g.placeholder(p + "/indices", np.int64, [-1, k])
        g.placeholder(p + "/rows", np.int64, [1])
        feeds[p + "/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[p + "/indices"] = np.stack(cols, 1).astype(np.int64).reshape(nnz, k)
        feeds[p + "/rows"] = np.asarray([rows], np.int64)
        g.node(p + "/n", "Squeeze", [p + "/rows"], T=("type", P.DT_INT64), squeeze_dims=("ints", [0]))
        return p, p + "/values", p + "/indices", p + "/n"

    for grp in range(int(rng.integers(1, 4))):
        B = int(rng.integers(1, 40))
        ins = []
        for _ in range(int(rng.integers(2, 15))):
            kind = str(rng.choice(["dense64", "dense32", "bucket", "mean", "sum", "scatter", "pass", "sum3d"]))
            dim = int(rng.choice([4, 8, 12, 16, 20, 32, 64]))
            kinds.append(kind)
            if kind in ("dense64", "dense32"):
                (t, vocab), dt = table(dim), (np.int64 if kind == "dense64" else np.int32)
                two_d = rng.random() < 0.5
                ph = name("ids")
                g.placeholder(ph, dt, [-1, 1] if two_d else [-1])
                feeds[ph] = rng.integers(0, vocab, size=(B, 1) if two_d else (B,)).astype(dt)
                out = g.gather(name("GatherDense"), t, ph, dt)
                if two_d:
                    shp = g.const(name("shape"), np.asarray([-1, dim], np.int32))
                    out = g.node(name("Reshape"), "Reshape", [out, shp], T=("type", P.DT_FLOAT),
                                 Tshape=("type", P.DT_INT32))
                ins.append(out)
            elif kind == "bucket":
                nb = int(rng.integers(1, 60))
                bnd = np.sort(rng.uniform(-50, 50, nb)).astype(np.float32)
                t = g.variable(name("emb") + "/embedding_weights", nb + 1, dim)
                variables[t] = rng.standard_normal((nb + 1, dim)).astype(np.float32)
                ph = name("val")
                g.placeholder(ph, np.float32, [-1])
                v = rng.uniform(-60, 60, B).astype(np.float32)
                v[: min(B, nb)] = bnd[: min(B, nb)]          # exact boundary hits
                feeds[ph] = v
                bk = g.node(name("Bucketize"), "Bucketize", [ph], T=("type", P.DT_FLOAT),
                            boundaries=("floats", [float(x) for x in bnd]))
                ins.append(g.gather(name("GatherDense"), t, bk, np.int32))
            elif kind in ("mean", "sum"):
                t, vocab = table(dim)
                p, v, i, n = sparse(B, vocab, int(rng.integers(1, 12)))
                seg = g.slice_col0(p + "/added_strided_slice", i, shrink=True)
                op = "SparseSegmentMeanWithNumSegments" if kind == "mean" else "SparseSegmentSumWithNumSegments"
                ins.append(g.node(name(op), op, [t, v, seg, n], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                                  Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
            elif kind == "scatter":
                t, vocab = table(dim)
                p, v, i, n = sparse(B, vocab, 1)
                rows = g.slice_col0(p + "/added_strided_slice", i, shrink=False)
                gat = g.gather(name("Gather"), t, v, np.int64)
                d = g.const(name("dim"), np.asarray(dim, np.int64))
                shp = g.node(name("Scatter_shape"), "Pack", [n, d], N=2, T=("type", P.DT_INT64), axis=0)
                ins.append(g.node(name("Scatter"), "ScatterNd", [rows, gat, shp], T=("type", P.DT_FLOAT),
                                  Tindices=("type", P.DT_INT64)))
            elif kind == "pass":
                ph = name("dense")
                g.placeholder(ph, np.float32, [-1, dim])
                feeds[ph] = rng.standard_normal((B, dim)).astype(np.float32)
                ins.append(ph)
            else:
                r = int(rng.integers(1, 6))
                ph = name("seq")
                g.placeholder(ph, np.float32, [-1, r, dim])
                feeds[ph] = rng.standard_normal((B, r, dim)).astype(np.float32)
                ax = g.const(name("axis"), np.asarray(1, np.int32))
                ins.append(g.node(name("Sum"), "Sum", [ph, ax], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32),
                                  keep_dims=False))
        ax = g.const(name("concat_axis"), np.asarray(1, np.int32))
        c = g.node(f"group{grp}/concat", "ConcatV2", ins + [ax], N=len(ins), T=("type", P.DT_FLOAT),
                   Tidx=("type", P.DT_INT32))
        fetches.append(g.node(f"output_{grp}", "Identity", [c], T=("type", P.DT_FLOAT)))
    return g.gd, feeds, variables, fetches, kinds


def plain_segment_model(B=21, seed=0):
    """Plain ``SparseSegmentSum`` / ``SparseSegmentMean`` — no ``num_segments`` — next to a one-hot column (the emitter
    takes these ops too, cuda_emitter.cc:1096-1113): TensorFlow gives them ``last segment id + 1`` rows, so the fixture keeps
    the last row non-empty (the ConcatV2 needs B rows) while rows in the middle may be.  Column ``s``: SparseTensor
    indices [nnz, 2] int64 behind ``[:, 0]``; column ``m``: int32 row ids [nnz] as they are.
    Returns (graph_def, feeds, variables, fetches)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables = {}, {}
    ins = []
    t = g.variable("input_layer/d_embedding/embedding_weights", 61, 8)
    variables[t] = rng.standard_normal((61, 8)).astype(np.float32)
    g.placeholder("d_ids", np.int64, [-1])
    feeds["d_ids"] = rng.integers(0, 61, size=B).astype(np.int64)
    ins.append(g.gather("input_layer/d_embedding/GatherDense", t, "d_ids", np.int64))
    for name, vocab, dim, op, seg_dtype in (("s", 97, 16, "SparseSegmentSum", np.int64), ("m", 53, 4, "SparseSegmentMean", np.int32)):
        lens = rng.integers(0, 6, size=B)
        lens[-1] = max(1, lens[-1])
        nnz = int(lens.sum())
        rows = np.repeat(np.arange(B), lens)
        tab = g.variable(f"input_layer/{name}_embedding/embedding_weights", vocab, dim)
        variables[tab] = rng.standard_normal((vocab, dim)).astype(np.float32)
        g.placeholder(f"{name}/values", np.int64, [-1])
        feeds[f"{name}/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        if seg_dtype == np.int64:
            g.placeholder(f"{name}/indices", np.int64, [-1, 2])
            feeds[f"{name}/indices"] = np.stack([rows, np.concatenate([np.arange(l) for l in lens])], 1).astype(np.int64)
            seg = g.slice_col0(f"{name}/added_strided_slice", f"{name}/indices", shrink=True)
        else:
            g.placeholder(f"{name}/row_ids", np.int32, [-1])
            feeds[f"{name}/row_ids"] = rows.astype(np.int32)
            seg = f"{name}/row_ids"
        ins.append(g.node(f"{name}/{op}", op, [tab, f"{name}/values", seg], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                          Tsegmentids=("type", P.DT_INT64 if seg_dtype == np.int64 else P.DT_INT32),
                          _output_shapes=("shapes", [[-1, dim]])))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=len(ins), T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))
    return g.gd, feeds, variables, ["input_layer/concat"]


def sparse_reshape_model(B=23, seed=0):
    """Two pooled columns whose SparseTensor indices pass through ``SparseReshape`` before ``[:, 0]``, the
    way ``safe_embedding_lookup_sparse`` flattens its ids (``tf.sparse.reshape(ids, [prod(shape[:-1]),
    shape[-1]])``): column ``p`` reshapes [B, L] -> [B, L] (provably the identity: the plan builder reads the
    input indices in place), column ``q`` reshapes [B, 2L] -> [2B, L] (row = (idx0 * 2L + idx1) // L with 2L known per
    request only: a segment-id map with one run-time factor); ``r``, ``s``, ``u`` below; each feeds its own ConcatV2
    next to a one-hot column.
    Returns (graph_def, feeds, variables, fetches)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables = {}, {}
    ins = []
    for name, vocab, dim, identity in (("p", 131, 16, True), ("q", 89, 8, False)):
        rows, L = B, 6
        lens = rng.integers(0, L + 1, size=rows)
        nnz = int(lens.sum())
        width = L if identity else 2 * L
        cols = np.concatenate([np.sort(rng.choice(width, size=l, replace=False)) for l in lens]) if nnz else np.zeros(0, np.int64)
        idx = np.stack([np.repeat(np.arange(rows), lens), cols], 1).astype(np.int64).reshape(nnz, 2)
        t = g.variable(f"input_layer/{name}_embedding/embedding_weights", vocab, dim)
        variables[t] = rng.standard_normal((vocab, dim)).astype(np.float32)
        g.placeholder(f"{name}/values", np.int64, [-1])
        g.placeholder(f"{name}/indices", np.int64, [-1, 2])
        g.placeholder(f"{name}/dense_shape", np.int64, [2])
        feeds[f"{name}/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[f"{name}/indices"] = idx
        feeds[f"{name}/dense_shape"] = np.asarray([rows, width], np.int64)
        one = lambda s, v: g.const(f"{name}/{s}", np.asarray([v], np.int32))
        # new_shape = [prod(shape[:-1]) (x2 for q), shape[-1] (/2 for q)]
        g.node(f"{name}/lead", "StridedSlice", [f"{name}/dense_shape", one("lb", 0), one("le", 1), one("ls", 1)],
               T=("type", P.DT_INT64), Index=("type", P.DT_INT32))
        g.const(f"{name}/axis0", np.asarray([0], np.int32))
        g.node(f"{name}/rows", "Prod", [f"{name}/lead", f"{name}/axis0"], T=("type", P.DT_INT64), Tidx=("type", P.DT_INT32))
        g.node(f"{name}/last", "StridedSlice", [f"{name}/dense_shape", one("tb", 1), one("te", 2), one("ts", 1)],
               T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1)
        if identity:
            g.node(f"{name}/new_shape", "Pack", [f"{name}/rows", f"{name}/last"], N=2, T=("type", P.DT_INT64), axis=0)
        else:
            g.const(f"{name}/new_shape", np.asarray([2 * rows, L], np.int64))
        g.node(f"{name}/SparseReshape", "SparseReshape", [f"{name}/indices", f"{name}/dense_shape", f"{name}/new_shape"])
        seg = g.slice_col0(f"{name}/added_strided_slice", f"{name}/SparseReshape", shrink=True)
        # num_segments = output_shape[0] of the reshape
        g.node(f"{name}/num_segments", "StridedSlice", [f"{name}/SparseReshape:1", one("nb", 0), one("ne", 1), one("ns", 1)],
               T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1)
        pooled = g.node(f"{name}/SparseSegmentSum_with_num_segments", "SparseSegmentSumWithNumSegments",
                        [t, f"{name}/values", seg, f"{name}/num_segments"], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                        Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64))
        # a dense companion per concat (a ConcatV2 of one input is not a converging point)
        d = g.variable(f"input_layer/{name}_dense_embedding/embedding_weights", 40, 4)
        variables[d] = rng.standard_normal((40, 4)).astype(np.float32)
        n_rows = rows if identity else 2 * rows             # q's reshape doubles the rows
        g.placeholder(f"{name}/dense_ids", np.int64, [-1])
        feeds[f"{name}/dense_ids"] = rng.integers(0, 40, size=n_rows).astype(np.int64)
        dense = g.gather(f"input_layer/{name}_dense_embedding/GatherDense", d, f"{name}/dense_ids", np.int64)
        g.const(f"{name}/concat/axis", np.asarray(1, np.int32))
        g.node(f"{name}_layer/concat", "ConcatV2", [pooled, dense, f"{name}/concat/axis"], N=2, T=("type", P.DT_FLOAT),
               Tidx=("type", P.DT_INT32))
        ins.append(g.node(f"output_{name}", "Identity", [f"{name}_layer/concat"], T=("type", P.DT_FLOAT)))

    # three more: `r` is safe_embedding_lookup_sparse over a RANK-3 SparseTensor, [B, T, L] -> [prod(shape[:-1]), shape[-1]]
    # with T and L known per request only (row = idx0 * T + idx1: one run-time factor); `s` has constant shapes,
    # [2B, 4, 6] -> [4B, 12] (row = (idx0 * 4 + idx1) // 2 after cancelling the 6); `u` takes its new shape from a
    # placeholder: nothing is provable, the op stays in TensorFlow
    for name, vocab, dim in (("r", 77, 12), ("s", 65, 20), ("u", 59, 4)):
        T_, L = 3, 5
        ishape = {"r": (B, T_, L), "s": (2 * B, 4, 6), "u": (B, 2 * L)}[name]
        oshape = {"r": (B * T_, L), "s": (4 * B, 12), "u": (2 * B, L)}[name]
        total = int(np.prod(ishape))
        nnz = int(rng.integers(0, min(total, 6 * B) + 1))
        flat = np.sort(rng.choice(total, size=nnz, replace=False))
        idx = np.stack(np.unravel_index(flat, ishape), 1).astype(np.int64).reshape(nnz, len(ishape))
        t = g.variable(f"input_layer/{name}_embedding/embedding_weights", vocab, dim)
        variables[t] = rng.standard_normal((vocab, dim)).astype(np.float32)
        g.placeholder(f"{name}/values", np.int64, [-1])
        g.placeholder(f"{name}/indices", np.int64, [-1, len(ishape)])
        feeds[f"{name}/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[f"{name}/indices"] = idx
        one = lambda s_, v: g.const(f"{name}/{s_}", np.asarray([v], np.int32))
        if name == "s":
            g.const(f"{name}/dense_shape", np.asarray(ishape, np.int64))
            g.const(f"{name}/new_shape", np.asarray(oshape, np.int64))
        else:
            g.placeholder(f"{name}/dense_shape", np.int64, [len(ishape)])
            feeds[f"{name}/dense_shape"] = np.asarray(ishape, np.int64)
        if name == "r":
            g.node(f"{name}/lead", "StridedSlice", [f"{name}/dense_shape", one("lb", 0), one("le", 2), one("ls", 1)],
                   T=("type", P.DT_INT64), Index=("type", P.DT_INT32))
            g.const(f"{name}/axis0", np.asarray([0], np.int32))
            g.node(f"{name}/rows", "Prod", [f"{name}/lead", f"{name}/axis0"], T=("type", P.DT_INT64), Tidx=("type", P.DT_INT32))
            g.node(f"{name}/last", "StridedSlice", [f"{name}/dense_shape", one("tb", 2), one("te", 3), one("ts", 1)],
                   T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1)
            g.node(f"{name}/new_shape", "Pack", [f"{name}/rows", f"{name}/last"], N=2, T=("type", P.DT_INT64), axis=0)
        if name == "u":
            g.placeholder(f"{name}/new_shape", np.int64, [2])
            feeds[f"{name}/new_shape"] = np.asarray(oshape, np.int64)
        g.node(f"{name}/SparseReshape", "SparseReshape", [f"{name}/indices", f"{name}/dense_shape", f"{name}/new_shape"])
        seg = g.slice_col0(f"{name}/added_strided_slice", f"{name}/SparseReshape", shrink=True)
        g.node(f"{name}/num_segments", "StridedSlice", [f"{name}/SparseReshape:1", one("nb", 0), one("ne", 1), one("ns", 1)],
               T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1)
        op = "SparseSegmentMeanWithNumSegments" if name == "r" else "SparseSegmentSumWithNumSegments"
        pooled = g.node(f"{name}/{op}", op, [t, f"{name}/values", seg, f"{name}/num_segments"], T=("type", P.DT_FLOAT),
                        Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64))
        d = g.variable(f"input_layer/{name}_dense_embedding/embedding_weights", 40, 4)
        variables[d] = rng.standard_normal((40, 4)).astype(np.float32)
        g.placeholder(f"{name}/dense_ids", np.int64, [-1])
        feeds[f"{name}/dense_ids"] = rng.integers(0, 40, size=oshape[0]).astype(np.int64)
        dense = g.gather(f"input_layer/{name}_dense_embedding/GatherDense", d, f"{name}/dense_ids", np.int64)
        g.const(f"{name}/concat/axis", np.asarray(1, np.int32))
        g.node(f"{name}_layer/concat", "ConcatV2", [pooled, dense, f"{name}/concat/axis"], N=2, T=("type", P.DT_FLOAT),
               Tidx=("type", P.DT_INT32))
        ins.append(g.node(f"output_{name}", "Identity", [f"{name}_layer/concat"], T=("type", P.DT_FLOAT)))
    return g.gd, feeds, variables, ins


def id_filter_model(B=29, seed=0):
    """Columns whose ids pass through the CPU id ops PreLookupOptimizer leaves in front of a lookup
    (pre_lookup_optimizer.cc:596-654): SelectValue on a one-hot column; GatherIndiceValue (indices + values)
    in front of a pooled mean and a pooled sum (the sum after Bucketize + Cast); GatherValueGenIndice +
    ScatterNd (the dense-input form); and a column with TWO transforms in a row (only the outer one can be
    fused: the inner op stays in TensorFlow).  Returns (graph_def, feeds, variables, fetches)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables, ins = {}, {}, []

    def table(name, vocab, dim):
        t = g.variable(f"input_layer/{name}_embedding/embedding_weights", vocab, dim)
        variables[t] = rng.standard_normal((vocab, dim)).astype(np.float32)
        return t

    def ivals(lo, hi):
        return dict(left_boundaries=("ints", lo), right_boundaries=("ints", hi), T=("type", P.DT_INT64))

    def sparse(prefix, vocab, max_len, dtype=np.int64):
        lens = rng.integers(0, max_len + 1, size=B)
        nnz = int(lens.sum())
        idx = np.stack([np.repeat(np.arange(B), lens),
                        np.concatenate([np.arange(l) for l in lens]) if nnz else np.zeros(0, np.int64)], 1)
        g.placeholder(prefix + "/values", dtype, [-1])
        g.placeholder(prefix + "/indices", np.int64, [-1, 2])
        g.placeholder(prefix + "/rows", np.int64, [])
        feeds[prefix + "/values"] = (rng.integers(0, vocab, size=nnz).astype(dtype) if dtype != np.float32
                                     else rng.uniform(-10, 510, size=nnz).astype(np.float32))
        feeds[prefix + "/indices"] = idx.astype(np.int64).reshape(nnz, 2)
        feeds[prefix + "/rows"] = np.asarray(B, np.int64)

    # a: one-hot, SelectValue (ids outside [10, 60] u [80, 90] become 3)
    t = table("a", 97, 8)
    g.placeholder("a_ids", np.int64, [-1])
    feeds["a_ids"] = rng.integers(0, 97, size=B).astype(np.int64)
    g.node("a/SelectValue", "Addons>SelectValue", ["a_ids"], substitute=3, **ivals([10, 80], [60, 90]))
    ins.append(g.gather("input_layer/a_embedding/GatherDense", t, "a/SelectValue", np.int64))
    # b: pooled mean, GatherIndiceValue keeps ids in [20, 150]
    t = table("b", 211, 16)
    sparse("b", 211, 6)
    g.node("b/GatherIndiceValue", "Addons>GatherIndiceValue", ["b/indices", "b/values"], **ivals([20], [150]))
    seg = g.slice_col0("b/added_strided_slice", "b/GatherIndiceValue", shrink=True)
    ins.append(g.node("b/SparseSegmentMean_with_num_segments", "SparseSegmentMeanWithNumSegments",
                      [t, "b/GatherIndiceValue:1", seg, "b/rows"], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                      Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
    # c: pooled sum over Bucketize(float) -> Cast -> GatherIndiceValue (buckets 5..95 survive)
    t = table("c", 101, 8)
    sparse("c", 101, 5, np.float32)
    g.node("c/Bucketize", "Bucketize", ["c/values"], T=("type", P.DT_FLOAT), boundaries=("floats", MICRO_BOUNDARIES))
    g.node("c/Cast", "Cast", ["c/Bucketize"], SrcT=("type", P.DT_INT32), DstT=("type", P.DT_INT64))
    g.node("c/GatherIndiceValue", "Addons>GatherIndiceValue", ["c/indices", "c/Cast"], **ivals([5], [95]))
    seg = g.slice_col0("c/added_strided_slice", "c/GatherIndiceValue", shrink=True)
    ins.append(g.node("c/SparseSegmentSum_with_num_segments", "SparseSegmentSumWithNumSegments",
                      [t, "c/GatherIndiceValue:1", seg, "c/rows"], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                      Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
    # d: dense input, GatherValueGenIndice + ScatterNd (a value outside [1, 40] leaves a zero row)
    t = table("d", 64, 4)
    g.placeholder("d_ids", np.int64, [-1])
    feeds["d_ids"] = rng.integers(0, 64, size=B).astype(np.int64)
    g.node("d/GatherValueGenIndice", "Addons>GatherValueGenIndice", ["d_ids"], **ivals([1], [40]))
    g.gather("input_layer/d_embedding/GatherScatter/Gather", t, "d/GatherValueGenIndice:1", np.int64)
    g.placeholder("d/rows", np.int64, [])
    feeds["d/rows"] = np.asarray(B, np.int64)
    g.const("d/dim", np.asarray(4, np.int64))
    g.node("d/Scatter_shape", "Pack", ["d/rows", "d/dim"], N=2, T=("type", P.DT_INT64), axis=0)
    ins.append(g.node("input_layer/d_embedding/GatherScatter/Scatter", "ScatterNd",
                      ["d/GatherValueGenIndice", "input_layer/d_embedding/GatherScatter/Gather", "d/Scatter_shape"],
                      T=("type", P.DT_FLOAT), Tindices=("type", P.DT_INT64)))
    # e: two transforms in a row: SelectValue(SelectValue(ids)) — the outer one is fused, the inner one stays
    t = table("e", 53, 12)
    g.placeholder("e_ids", np.int64, [-1])
    feeds["e_ids"] = rng.integers(0, 53, size=B).astype(np.int64)
    g.node("e/inner", "Addons>SelectValue", ["e_ids"], substitute=1, **ivals([0], [30]))
    g.node("e/outer", "Addons>SelectValue", ["e/inner"], substitute=2, **ivals([5], [52]))
    ins.append(g.gather("input_layer/e_embedding/GatherDense", t, "e/outer", np.int64))
    # f: categorical_column_with_hash_bucket over an int64 feature (dlrm.py "hash-int" columns):
    #    AsString -> StringToHashBucketFast(100) -> lookup, pooled mean over SparseTensor indices
    t = table("f", 100, 8)
    sparse("f", 10**12, 4)
    feeds["f/values"][::3] *= -1                            # negative ids hash too ("-123")
    g.node("f/AsString", "AsString", ["f/values"], T=("type", P.DT_INT64))
    g.node("f/hash", "StringToHashBucketFast", ["f/AsString"], num_buckets=100)
    seg = g.slice_col0("f/added_strided_slice", "f/indices", shrink=True)
    ins.append(g.node("f/SparseSegmentMean_with_num_segments", "SparseSegmentMeanWithNumSegments",
                      [t, "f/hash", seg, "f/rows"], T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64),
                      Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
    # h: the same over a one-hot int32 feature, followed by SelectValue (hash first, then the interval test)
    t = table("h", 1000, 4)
    g.placeholder("h_ids", np.int32, [-1])
    feeds["h_ids"] = rng.integers(0, 2**31 - 1, size=B).astype(np.int32)
    g.node("h/AsString", "AsString", ["h_ids"], T=("type", P.DT_INT32))
    g.node("h/hash", "StringToHashBucketFast", ["h/AsString"], num_buckets=1000)
    g.node("h/SelectValue", "Addons>SelectValue", ["h/hash"], substitute=0, **ivals([100], [899]))
    ins.append(g.gather("input_layer/h_embedding/GatherDense", t, "h/SelectValue", np.int64))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=len(ins), T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))
    g.node("output", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output"]


def resource_variable_model(B=21, seed=0):
    """A TF2-SavedModel-shaped graph: tables are resource variables (VarHandleOp), read through
    ResourceGather (one-hot column), ReadVariableOp -> SparseSegmentSum (pooled column, the variable already
    has a reader) and a table shared by a ResourceGather and a second pooled column through its own
    ReadVariableOp.  Returns (graph_def, feeds, variables, fetches)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables, ins = {}, {}, []

    def var(name, vocab, dim):
        n = g.node(f"input_layer/{name}_embedding/embedding_weights", "VarHandleOp", dtype=("type", P.DT_FLOAT),
                   shape=("shape", [vocab, dim]))
        variables[n] = rng.standard_normal((vocab, dim)).astype(np.float32)
        return n

    def pooled(prefix, table_value, vocab, op):
        lens = rng.integers(0, 5, size=B)
        nnz = int(lens.sum())
        g.placeholder(prefix + "/values", np.int64, [-1])
        g.placeholder(prefix + "/indices", np.int64, [-1, 2])
        g.placeholder(prefix + "/rows", np.int64, [])
        feeds[prefix + "/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[prefix + "/indices"] = np.stack([np.repeat(np.arange(B), lens), np.zeros(nnz, np.int64)], 1).astype(np.int64)
        feeds[prefix + "/rows"] = np.asarray(B, np.int64)
        seg = g.slice_col0(prefix + "/added_strided_slice", prefix + "/indices", shrink=True)
        return g.node(prefix + "/" + op, op, [table_value, prefix + "/values", seg, prefix + "/rows"], T=("type", P.DT_FLOAT),
                      Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64))

    ta = var("a", 77, 8)
    g.placeholder("a_ids", np.int64, [-1])
    feeds["a_ids"] = rng.integers(0, 77, size=B).astype(np.int64)
    ins.append(g.node("input_layer/a_embedding/ResourceGather", "ResourceGather", [ta, "a_ids"], dtype=("type", P.DT_FLOAT),
                      Tindices=("type", P.DT_INT64), batch_dims=0))
    tb = var("b", 140, 16)
    g.node("b/Read", "ReadVariableOp", [tb], dtype=("type", P.DT_FLOAT))
    ins.append(pooled("b", "b/Read", 140, "SparseSegmentSumWithNumSegments"))
    g.placeholder("c_ids", np.int32, [-1])
    feeds["c_ids"] = rng.integers(0, 77, size=B).astype(np.int32)
    ins.append(g.node("input_layer/c_embedding/ResourceGather", "ResourceGather", [ta, "c_ids"], dtype=("type", P.DT_FLOAT),
                      Tindices=("type", P.DT_INT32), batch_dims=0))                 # shares table a
    g.node("d/Read", "ReadVariableOp", [ta], dtype=("type", P.DT_FLOAT))
    ins.append(pooled("d", "d/Read", 77, "SparseSegmentMeanWithNumSegments"))       # table a again, through a reader
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=len(ins), T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))
    g.node("output", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output"]


def random_sparse_reshape_model(seed):
    """One to four pooled columns whose SparseTensor passes through a SparseReshape with RANDOM shapes: input rank 1-4,
    output rank 1-3 (any factorisation of the element count), the input shape a constant or fed, every output dimension
    a constant, a copy of an input dimension (StridedSlice of the fed shape), a Prod over leading dimensions, or -1, or
    the whole new_shape fed.  Whatever the plan builders make of each — identity, segment-id map with or without a
    run-time factor, or "stays in TensorFlow" — the rewritten graph must equal the original.
    Returns (graph_def, feeds, variables, fetches)."""
    rng = np.random.default_rng(7000 + seed)
    g = GB()
    feeds, variables, outs = {}, {}, []
    for ci in range(int(rng.integers(1, 5))):
        name = f"c{ci}"
        r = int(rng.integers(1, 5))
        ishape = [int(rng.integers(1, 7)) for _ in range(r)]
        ishape[0] = int(rng.integers(1, 12))
        total = int(np.prod(ishape))
        # a random factorisation of `total` into q output dims
        q = int(rng.integers(1, 4))
        oshape, rest = [], total
        for _ in range(q - 1):
            divs = [d for d in range(1, rest + 1) if rest % d == 0]
            d = int(rng.choice(divs))
            oshape.append(d)
            rest //= d
        oshape.append(rest)
        rng.shuffle(oshape)
        oshape = [int(d) for d in oshape]
        nnz = int(rng.integers(0, min(total, 40) + 1))
        flat = np.sort(rng.choice(total, size=nnz, replace=False))
        idx = np.stack(np.unravel_index(flat, ishape), 1).astype(np.int64).reshape(nnz, r)
        vocab, dim = int(rng.integers(5, 60)), int(rng.choice([4, 8, 12]))
        t = g.variable(f"input_layer/{name}_embedding/embedding_weights", vocab, dim)
        variables[t] = rng.standard_normal((vocab, dim)).astype(np.float32)
        g.placeholder(f"{name}/values", np.int64, [-1])
        g.placeholder(f"{name}/indices", np.int64, [-1, r])
        feeds[f"{name}/values"] = rng.integers(0, vocab, size=nnz).astype(np.int64)
        feeds[f"{name}/indices"] = idx
        one = lambda s_, v: g.const(f"{name}/{s_}", np.asarray([v], np.int32))
        fed_shape = rng.random() < 0.6
        if fed_shape:
            g.placeholder(f"{name}/dense_shape", np.int64, [r])
            feeds[f"{name}/dense_shape"] = np.asarray(ishape, np.int64)
        else:
            g.const(f"{name}/dense_shape", np.asarray(ishape, np.int64))
        kind = str(rng.choice(["const", "fed", "pack", "pack", "pack"]))
        if kind == "const":
            ns = list(oshape)
            if rng.random() < 0.3:
                ns[int(rng.integers(0, q))] = -1                      # inferred at run time
            g.const(f"{name}/new_shape", np.asarray(ns, np.int64))
        elif kind == "fed":
            g.placeholder(f"{name}/new_shape", np.int64, [q])
            feeds[f"{name}/new_shape"] = np.asarray(oshape, np.int64)
        else:
            parts = []
            for k, d in enumerate(oshape):
                same = [j for j in range(r) if ishape[j] == d]
                pick = rng.random()
                if same and pick < 0.5:                               # a copy of an input dimension
                    j = int(rng.choice(same))
                    parts.append(g.node(f"{name}/o{k}", "StridedSlice", [f"{name}/dense_shape", one(f"o{k}b", j), one(f"o{k}e", j + 1),
                                                                          one(f"o{k}s", 1)],
                                        T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1))
                elif k == 0 and pick < 0.8 and r >= 2 and int(np.prod(ishape[:r - 1])) == d:   # Prod over the leading dims
                    g.node(f"{name}/lead", "StridedSlice", [f"{name}/dense_shape", one("lb", 0), one("le", r - 1), one("ls", 1)],
                           T=("type", P.DT_INT64), Index=("type", P.DT_INT32))
                    g.const(f"{name}/axis0", np.asarray([0], np.int32))
                    g.node(f"{name}/prod", "Prod", [f"{name}/lead", f"{name}/axis0"], T=("type", P.DT_INT64), Tidx=("type", P.DT_INT32))
                    parts.append(g.node(f"{name}/o{k}", "Squeeze", [f"{name}/prod"], T=("type", P.DT_INT64)))
                else:
                    parts.append(g.const(f"{name}/o{k}", np.asarray(d, np.int64)))
            g.node(f"{name}/new_shape", "Pack", parts, N=q, T=("type", P.DT_INT64), axis=0)
        g.node(f"{name}/SparseReshape", "SparseReshape", [f"{name}/indices", f"{name}/dense_shape", f"{name}/new_shape"])
        if q >= 2:
            seg = g.slice_col0(f"{name}/added_strided_slice", f"{name}/SparseReshape", shrink=bool(rng.integers(0, 2)))
        else:                                                         # [nnz, 1]: the lookup optimizer's Squeeze / Reshape form
            g.const(f"{name}/flat", np.asarray([-1], np.int32))
            seg = g.node(f"{name}/seg", "Reshape", [f"{name}/SparseReshape", f"{name}/flat"], T=("type", P.DT_INT64), Tshape=("type", P.DT_INT32))
        g.node(f"{name}/num_segments", "StridedSlice", [f"{name}/SparseReshape:1", one("nb", 0), one("ne", 1), one("ns", 1)],
               T=("type", P.DT_INT64), Index=("type", P.DT_INT32), shrink_axis_mask=1)
        op = "SparseSegmentMeanWithNumSegments" if rng.random() < 0.5 else "SparseSegmentSumWithNumSegments"
        pooled = g.node(f"{name}/{op}", op, [t, f"{name}/values", seg, f"{name}/num_segments"], T=("type", P.DT_FLOAT),
                        Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64))
        d_ = g.variable(f"input_layer/{name}_dense_embedding/embedding_weights", 20, 4)
        variables[d_] = rng.standard_normal((20, 4)).astype(np.float32)
        g.placeholder(f"{name}/dense_ids", np.int64, [-1])
        feeds[f"{name}/dense_ids"] = rng.integers(0, 20, size=oshape[0]).astype(np.int64)
        dense = g.gather(f"input_layer/{name}_dense_embedding/GatherDense", d_, f"{name}/dense_ids", np.int64)
        g.const(f"{name}/concat/axis", np.asarray(1, np.int32))
        g.node(f"{name}_layer/concat", "ConcatV2", [pooled, dense, f"{name}/concat/axis"], N=2, T=("type", P.DT_FLOAT),
               Tidx=("type", P.DT_INT32))
        outs.append(g.node(f"output_{name}", "Identity", [f"{name}_layer/concat"], T=("type", P.DT_FLOAT)))
    return g.gd, feeds, variables, outs


def s1_model(columns=100, dim=16, vocab=10_000, B=128, seed=0):
    """BASELINE.json configs[0] "S1" as a GraphDef in rewritten form (recom_amd.synth.model_s1: 100 columns, dim 16,
    vocab 10 k, batch 128, one id per row): even columns arrive as form 1 (dense GatherV2, int64 ids), odd columns as
    form 2 (SparseSegmentMeanWithNumSegments over exactly one id per row, segment ids = indices[:, 0]) — both rewrites
    the reference's lookup optimizer produces for a one-id embedding_column (lookup_optimizer.cc:157-322)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables, ins = {}, {}, []
    for c in range(columns):
        t = g.variable(f"input_layer/s1_{c}_embedding/embedding_weights", vocab, dim)
        variables[t] = (rng.standard_normal((vocab, dim)) * dim ** -0.5).astype(np.float32)
        if c % 2 == 0:
            g.placeholder(f"s1_{c}/ids", np.int64, [-1])
            feeds[f"s1_{c}/ids"] = rng.integers(0, vocab, size=B).astype(np.int64)
            ins.append(g.gather(f"input_layer/s1_{c}_embedding/GatherDense", t, f"s1_{c}/ids", np.int64))
        else:
            p = f"s1_{c}"
            g.placeholder(p + "/values", np.int64, [-1])
            g.placeholder(p + "/indices", np.int64, [-1, 2])
            g.placeholder(p + "/dense_shape", np.int64, [2])
            feeds[p + "/values"] = rng.integers(0, vocab, size=B).astype(np.int64)
            feeds[p + "/indices"] = np.stack([np.arange(B), np.zeros(B, np.int64)], 1).astype(np.int64)
            feeds[p + "/dense_shape"] = np.asarray([B, 1], np.int64)
            b_ = g.const(p + "/ns/b", np.asarray([0], np.int32))
            e_ = g.const(p + "/ns/e", np.asarray([1], np.int32))
            s_ = g.const(p + "/ns/s", np.asarray([1], np.int32))
            g.node(p + "/num_segments", "StridedSlice", [p + "/dense_shape", b_, e_, s_], T=("type", P.DT_INT64),
                   Index=("type", P.DT_INT32))
            g.node(p + "/num_segments_squeeze", "Squeeze", [p + "/num_segments"], T=("type", P.DT_INT64),
                   squeeze_dims=("ints", [0]))
            seg = g.slice_col0(p + "/added_strided_slice", p + "/indices", shrink=True)
            ins.append(g.node(p + "/SparseSegmentMean_with_num_segments", "SparseSegmentMeanWithNumSegments",
                              [t, p + "/values", seg, p + "/num_segments_squeeze"], T=("type", P.DT_FLOAT),
                              Tidx=("type", P.DT_INT64), Tsegmentids=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT64)))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=columns, T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))
    g.node("output", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output"]


def s2_model(columns=1000, vocab=1_000_000, B=512, seed=0, materialize=True):
    """BASELINE.json configs[1] "S2" as a GraphDef in rewritten form (recom_amd.synth.model_s2: dims cycling 8 / 16 / 32 /
    64, one id per row, every 10th column a float feature bucketized with the micro-benchmark's 100 boundaries, the
    others int64 ids; all form 1).  `materialize=False`: `variables` maps each table to its (vocab, dim) instead of
    values (120 GB at full size: the caller fills them where they live)."""
    rng = np.random.default_rng(seed)
    g = GB()
    feeds, variables, ins = {}, {}, []
    dims = (8, 16, 32, 64)
    for c in range(columns):
        d = dims[c % 4]
        t = g.variable(f"input_layer/s2_{c}_embedding/embedding_weights", vocab, d)
        variables[t] = (rng.standard_normal((vocab, d)) * d ** -0.5).astype(np.float32) if materialize else (vocab, d)
        if c % 10 == 0:
            g.placeholder(f"s2_{c}/value", np.float32, [-1, 1])
            feeds[f"s2_{c}/value"] = rng.uniform(-10.0, 510.0, size=(B, 1)).astype(np.float32)
            g.node(f"s2_{c}/Bucketize", "Bucketize", [f"s2_{c}/value"], T=("type", P.DT_FLOAT), boundaries=("floats", MICRO_BOUNDARIES))
            g.const(f"s2_{c}/flat", np.asarray([-1], np.int32))
            g.node(f"s2_{c}/Reshape", "Reshape", [f"s2_{c}/Bucketize", f"s2_{c}/flat"], T=("type", P.DT_INT32), Tshape=("type", P.DT_INT32))
            ins.append(g.gather(f"input_layer/s2_{c}_embedding/GatherDense", t, f"s2_{c}/Reshape", np.int32))
        else:
            g.placeholder(f"s2_{c}/ids", np.int64, [-1])
            feeds[f"s2_{c}/ids"] = rng.integers(0, vocab, size=B).astype(np.int64)
            ins.append(g.gather(f"input_layer/s2_{c}_embedding/GatherDense", t, f"s2_{c}/ids", np.int64))
    g.const("concat/axis", np.asarray(1, np.int32))
    g.node("input_layer/concat", "ConcatV2", ins + ["concat/axis"], N=columns, T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT32))
    g.node("output", "Identity", ["input_layer/concat"], T=("type", P.DT_FLOAT))
    return g.gd, feeds, variables, ["output"]
