"""Plan builder from a rewritten GraphDef, CPU side.

original GraphDef --NumPy evaluator (oracle/tf_graph_eval.py)--> expected concat outputs
original GraphDef --build_plan + rewrite_graph--> Addons> ops, evaluated with the C oracle
                                                  behind them --> must be identical
(the GPU twin, with the HIP path behind the ops, is in tests/test_gpu_parity.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from graph_fixtures import (MICRO_BOUNDARIES, canonical_model, id_filter_model, microbenchmark_model, random_model,
                            resource_variable_model, sparse_reshape_model)
from recom_amd import plan as PL
from recom_amd.graph import Unsupported, build_plan, parse_graphdef, rewrite_graph
from recom_amd.graph import tf_proto as P
from recom_amd.graph.view import tensor_to_numpy
from recom_amd.plan_io import load_plan, save_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_ops(oracle, built, variables):
    """The three Addons> ops with the C oracle behind them (CPU stand-in for the shim)."""
    def concat_inputs(node, x):
        assert [int(r) for r in node.attr["ranks"].list.i] == [a.ndim for a in x]
        if "_fcp_plan" in node.attr:                             # graph rewritten for a staged plan: the op packs as the
            from recom_amd.ops import ConcatInputs               # plan file's stage section says (host-only product code)
            return list(ConcatInputs(list(node.attr["ranks"].list.i), node.attr["_fcp_plan"].s.decode())(x))
        blob, offsets, shapes = oracle.concat_inputs(x)
        return [blob, offsets, shapes]

    def process(node, x):
        spec = load_plan(node.attr["dlpath"].s.decode())
        n_tab = len(node.attr["input_types"].list.type)
        tables = x[3:3 + n_tab]
        symbols = x[3 + n_tab] if node.op.endswith("WithSymbols") else None
        groups, bad = oracle.process_feature_columns(spec.to_dict(), x[0], x[1], x[2], tables, symbols)
        assert bad == 0
        # output_shapes: [prefix, dim] per column OUTPUT (feature_column_process_op_gpu.cu.cc:113-118);
        # external slots (ConcatOutputs host inputs) are not outputs of this op
        outs = [spec.columns[k] for k in spec.output_columns()]
        assert len(outs) == len(node.attr["output_types"].list.type)
        shapes = np.asarray([v for c in outs for v in (groups[c.concat_group].shape[0], c.dim)], np.int32)
        return [np.zeros(len(outs), np.int64), shapes, groups]

    def concat_outputs(node, x):
        out_cols = built.spec.output_columns()
        col = out_cols[int(node.attr["device_input_indices"].list.i[0])]
        group = built.spec.columns[col].concat_group
        out = x[-1][group]                                    # FeatureColumnProcess:2 is wired last
        shapes = x[1]
        dims = [int(d) for d in node.attr["embedd_dims"].list.i]
        assert out.shape[0] == shapes[int(node.attr["prefix_begin"].i)]
        assert out.shape[1] == sum(dims)
        n = int(node.attr["N"].i)
        assert n == len(node.attr["host_concat_indices"].list.i) and (n == 0) == node.op.endswith("NoHost")
        scan = np.concatenate([[0], np.cumsum(dims)])
        for a, pos in zip(x[2:2 + n], node.attr["host_concat_indices"].list.i):   # host_inputs (concat_outputs_op_gpu.cu.cc:186-216)
            out[:, scan[pos]:scan[pos + 1]] = a
        return [out]

    return {"Addons>ConcatInputs": concat_inputs, "Addons>FeatureColumnProcess": process,
            "Addons>FeatureColumnProcessWithSymbols": process, "Addons>ConcatOutputsNoHost": concat_outputs,
            "Addons>ConcatOutputs": concat_outputs}


def test_graphdef_roundtrip_binary_and_text():
    gd, *_ = canonical_model()
    data = gd.SerializeToString()
    assert parse_graphdef(data) == gd
    from google.protobuf import text_format
    assert parse_graphdef(text_format.MessageToString(gd).encode()) == gd
    # wire compatibility with tensorflow.GraphDef: NodeDef{name=1, op=2, input=3, attr=5}, GraphDef{node=1}
    g = P.GraphDef()
    n = g.node.add(name="x", op="Placeholder")
    n.attr["dtype"].type = P.DT_FLOAT
    assert g.SerializeToString() == b"\n\x1d\n\x01x\x12\x0bPlaceholder*\x0b\n\x05dtype\x12\x020\x01"


def test_build_plan_canonical_forms():
    gd, feeds, variables, _ = canonical_model()
    built = build_plan(gd)
    spec = built.spec
    assert [g.concat_node for g in built.groups] == ["input_layer/concat", "seq_layer/concat"]
    forms = [c.form for c in spec.columns]
    assert forms == [1, 1, 1, 2, 2, 3, 4, 5, 1, 2, 1]
    c = spec.columns
    # column 2: Bucketize + Cast + Reshape absorbed, the float placeholder is shipped
    assert c[1].id_source == PL.IDS_F32_BUCKETIZE and np.array_equal(c[1].boundaries, np.float32(MICRO_BOUNDARIES))
    assert built.host_inputs[c[1].ids_input] == ("b_value", P.DT_FLOAT, 2)
    assert c[0].id_source == PL.IDS_I64 and c[2].id_source == PL.IDS_I32 and c[2].dim == 12
    # segment ids: indices[:, 0] of the [nnz, 2] matrix, read in place with stride 2 (cast absorbed)
    for k in (3, 4, 9):
        assert c[k].seg_kind == PL.SEG_IDS_I64 and c[k].seg_stride == 2 and c[k].rows_source == PL.ROWS_FROM_SYMBOL
        assert built.host_inputs[c[k].seg_input][0].endswith("/indices")
    assert c[3].combiner == PL.COMBINER_MEAN and c[4].combiner == PL.COMBINER_SUM and c[4].dim == 32
    assert c[5].seg_stride == 2 and built.symbols[c[5].rows_arg].tensor == "f/Scatter_shape"
    assert built.symbols[c[3].rows_arg].tensor == "d/num_segments" and built.symbols[c[3].rows_arg].index == 0
    assert c[6].dim == 13 and built.host_inputs[c[6].ids_input][0] == "dense_features"
    assert c[7].dim == 8 and built.host_inputs[c[7].ids_input] == ("seq_features", P.DT_FLOAT, 3)
    # shared table and shared placeholder are bound once; the int cast of the ids is absorbed
    assert c[8].table_input == c[2].table_input and c[8].ids_input == c[0].ids_input and c[8].id_source == PL.IDS_I64
    assert len(built.device_inputs) == 8 and len({t for t, _, _ in built.device_inputs}) == 8
    assert [c[k].concat_group for k in (9, 10)] == [1, 1] and [c[k].concat_slot for k in (9, 10)] == [0, 1]
    assert spec.group_width(0) == 8 + 8 + 12 + 16 + 32 + 4 + 13 + 8 + 12 and spec.group_width(1) == 28
    assert not built.skipped
    assert "2 concat group(s), 11 column(s)" in built.describe()


def test_plan_file_roundtrip(tmp_path):
    built = build_plan(canonical_model()[0])
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    again = load_plan(path)
    assert again.to_dict().keys() == built.spec.to_dict().keys()
    for a, b in zip(again.columns, built.spec.columns):
        da, db = dict(a.__dict__), dict(b.__dict__)
        ba, bb = da.pop("boundaries"), db.pop("boundaries")
        assert da == db and (ba is None) == (bb is None) and (ba is None or np.array_equal(ba, bb))


def test_rewrite_matches_reference_wiring(tmp_path):
    gd, feeds, variables, _ = canonical_model()
    built = build_plan(gd)
    out = rewrite_graph(gd, built, "/models/m.fcp")
    nodes = {n.name: n for n in out.node}
    ci, fc = nodes["ConcatInputs"], nodes["FeatureColumnProcess"]
    assert ci.op == "Addons>ConcatInputs" and list(ci.input) == [t for t, _, _ in built.host_inputs]
    assert list(ci.attr["T"].list.type) == [d for _, d, _ in built.host_inputs]
    assert list(ci.attr["ranks"].list.i) == [r for _, _, r in built.host_inputs]
    assert fc.op == "Addons>FeatureColumnProcessWithSymbols" and fc.attr["dlpath"].s == b"/models/m.fcp"
    assert list(fc.input[:3]) == ["ConcatInputs", "ConcatInputs:1", "ConcatInputs:2"]
    assert list(fc.input[3:-1]) == [t for t, _, _ in built.device_inputs]
    assert fc.input[-1] == "FeatureColumnProcess/symbols" and nodes[fc.input[-1]].op == "Pack"
    assert list(fc.attr["output_ranks"].list.i) == [2] * 11 and list(fc.attr["input_ranks"].list.i) == [2] * 8
    for g, gi in enumerate(built.groups):
        co = nodes[gi.concat_node]
        assert co.op == "Addons>ConcatOutputsNoHost" and co.attr["N"].i == 0
        assert gi.concat_node + "_removed" not in nodes                     # pruned
        assert list(co.input[:2]) == ["FeatureColumnProcess", "FeatureColumnProcess:1"]
        assert co.input[2] == "ConcatInputs" and co.input[-1] == "FeatureColumnProcess:2"
        assert list(co.attr["device_input_indices"].list.i) == gi.columns
        assert list(co.attr["device_concat_indices"].list.i) == list(range(gi.n_inputs))
        assert list(co.attr["embedd_dims"].list.i) == [built.spec.columns[k].dim for k in gi.columns]
        assert co.attr["prefix_begin"].i == 2 * gi.columns[0] and co.attr["prefix_end"].i == 2 * gi.columns[0] + 1
        assert list(co.attr["buffer_types"].list.type) == [P.DT_INT8] + [P.DT_FLOAT] * 8 + [P.DT_INT8]
        assert co.attr["BLOCK_THREADS"].i == 64
    # consumers keep their input names; the lookups are gone; unrelated nodes and the interface stay
    assert list(nodes["output_0"].input) == ["input_layer/concat"]
    assert not any(n.op in ("GatherV2", "ScatterNd", "Bucketize") or n.op.startswith("SparseSegment")
                   for n in out.node if not n.name.startswith("FeatureColumnProcess/symbols"))
    assert "dense_copy" in nodes and "a_ids" in nodes and "d/num_segments" in nodes
    # the input graph is untouched
    assert "ConcatInputs" not in {n.name for n in gd.node}
    with pytest.raises(ValueError):
        rewrite_graph(out, built, "x")


def test_rewrite_external_host_inputs_is_the_reference_wiring(tmp_path):
    """``host_concat="external"``: what the reference's unchanged Rewrite emits for a ConcatV2 with
    non-FC inputs (cuda_emitter.cc:2594-2611) — ``Addons>ConcatOutputs`` with N host inputs at
    ``host_concat_indices``; the plan reserves FORM_EXTERNAL slots and FeatureColumnProcess has one
    output per remaining column."""
    gd, feeds, variables, _ = canonical_model()
    built = build_plan(gd, host_concat="external")
    spec = built.spec
    assert [c.form for c in spec.columns] == [1, 1, 1, 2, 2, 3, 6, 5, 1, 2, 1]
    ext = spec.columns[6]
    assert ext.rows_source == PL.ROWS_FROM_GROUP and ext.ids_input == -1 and ext.dim == 13
    assert "dense_features" not in [t for t, _, _ in built.host_inputs]     # no longer shipped through ConcatInputs
    assert spec.output_columns() == [0, 1, 2, 3, 4, 5, 7, 8, 9, 10] and spec.group_width(0) == 113
    path = str(tmp_path / "m.fcp")
    save_plan(spec, path)
    assert [c.form for c in load_plan(path).columns] == [c.form for c in spec.columns]
    out = rewrite_graph(gd, built, path)
    nodes = {n.name: n for n in out.node}
    assert list(nodes["FeatureColumnProcess"].attr["output_ranks"].list.i) == [2] * 10
    co = nodes["input_layer/concat"]
    assert co.op == "Addons>ConcatOutputs" and co.attr["N"].i == 1
    assert list(co.attr["host_concat_indices"].list.i) == [6]
    assert list(co.attr["device_concat_indices"].list.i) == [0, 1, 2, 3, 4, 5, 7, 8]
    assert list(co.attr["device_input_indices"].list.i) == [0, 1, 2, 3, 4, 5, 6, 7]
    assert list(co.attr["embedd_dims"].list.i) == [c.dim for c in spec.columns[:9]]
    assert list(co.input[:3]) == ["FeatureColumnProcess", "FeatureColumnProcess:1", "dense_features"]
    assert co.input[3] == "ConcatInputs" and co.input[-1] == "FeatureColumnProcess:2"
    co2 = nodes["seq_layer/concat"]
    assert co2.op == "Addons>ConcatOutputsNoHost" and list(co2.attr["device_input_indices"].list.i) == [8, 9]
    assert co2.attr["prefix_begin"].i == 16


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
@pytest.mark.parametrize("B,seed", [(19, 0), (1, 1), (64, 2)])
def test_rewritten_graph_equals_original(oracle, tmp_path, B, seed, host_concat):
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = canonical_model(B=B, seed=seed)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    assert expected[0].shape == (B, 113) and expected[1].shape == (B + 5, 28)
    built = build_plan(gd, host_concat)
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert o.dtype == np.float32 and np.array_equal(e, o)              # same fp32 add order: bit-exact


def test_unsupported_lookup_stays_in_tensorflow(oracle, tmp_path):
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = canonical_model(unsupported=True)
    built = build_plan(gd)
    assert ("u/SparseSegmentSqrtN", "op SparseSegmentSqrtN is not a lookup") in built.skipped
    k = built.groups[0].columns[-1]
    assert built.spec.columns[k].form == PL.FORM_PASSTHROUGH
    assert built.host_inputs[built.spec.columns[k].ids_input][0] == "u/SparseSegmentSqrtN"
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    out = rewrite_graph(gd, built, path)
    assert any(n.name == "u/SparseSegmentSqrtN" for n in out.node)          # still computed by TF
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert all(np.array_equal(e, o) for e, o in zip(expected, got))


@pytest.mark.parametrize("staged", [False, True])
def test_plain_sparse_segment_ops_take_their_rows_from_the_last_segment_id(oracle, tmp_path, staged):
    """SparseSegmentSum / SparseSegmentMean WITHOUT num_segments (cuda_emitter.cc:1096-1113 takes them; VERDICT r03
    missing 2): TensorFlow's row count, last segment id + 1, is a host-side fact of the sorted segment ids the request
    ships — a symbol the rewritten graph computes (max(ids, -1) + 1: zero rows without ids) — so the columns are fused
    like their WithNumSegments twins; both builders, byte-identical; rewritten graph == original graph."""
    from tf_graph_eval import GraphEvaluator
    from recom_amd.graph import native_build
    from graph_fixtures import plain_segment_model
    gd, feeds, variables, fetches = plain_segment_model()
    built = build_plan(gd)
    assert not built.skipped
    forms = [(c.form, c.combiner, c.rows_source, c.seg_stride) for c in built.spec.columns]
    assert forms == [(PL.FORM_GATHER, PL.COMBINER_NONE, PL.ROWS_FROM_IDS, 1), (PL.FORM_SEGMENT_REDUCE, PL.COMBINER_SUM, PL.ROWS_FROM_SYMBOL, 2),
                     (PL.FORM_SEGMENT_REDUCE, PL.COMBINER_MEAN, PL.ROWS_FROM_SYMBOL, 1)]
    assert [(s.tensor, s.last_stride) for s in built.symbols] == [("s/indices", 2), ("m/row_ids", 1)]
    path = str(tmp_path / "m.fcp")
    stage = None
    if staged:
        spec, stage = built.spec.staged_for_concat_inputs()
        save_plan(spec, path, stage)
    else:
        save_plan(built.spec, path)
    out = rewrite_graph(gd, built, path, stage=stage)
    assert not any(n.op.startswith("SparseSegment") for n in out.node)        # fused away
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert expected[0].shape[0] == 21 and np.array_equal(expected[0], got[0])
    cpath = str(tmp_path / "c.fcp")
    c_graph, _ = native_build(gd.SerializeToString(), cpath, staged=staged)
    assert open(cpath).read() == open(path).read()
    assert parse_graphdef(c_graph) == rewrite_graph(gd, built, cpath, stage=stage)
    # no ids at all: zero rows, as TensorFlow says (the symbol graph alone; a ConcatV2 next to a [B] column could not hold it)
    sym = GraphEvaluator(out, variables).run(["FeatureColumnProcess/symbols"],
                                             {**feeds, "s/indices": np.zeros((0, 2), np.int64), "m/row_ids": np.zeros(0, np.int32)})[0]
    assert sym.tolist() == [0, 0]


def test_microbenchmark_graph(oracle, tmp_path):
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = microbenchmark_model(columns=6, B=32)
    built = build_plan(gd)
    assert all(c.form == PL.FORM_GATHER and c.id_source == PL.IDS_F32_BUCKETIZE and c.vocab == 101 and c.dim == 8
               for c in built.spec.columns)
    assert built.spec.n_symbols == 0
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    out = rewrite_graph(gd, built, path)
    assert {n.name: n for n in out.node}["FeatureColumnProcess"].op == "Addons>FeatureColumnProcess"
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert np.array_equal(expected[0], got[0])


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
@pytest.mark.parametrize("seed", range(12))
def test_random_graphs(oracle, tmp_path, seed, host_concat):
    """Random rewritten graphs (kinds, dims, vocabularies, id dtypes, SparseTensor ranks, shared
    tables, 1-3 concat groups): the plan builder never mislabels a column — the rewritten graph
    reproduces the original bit for bit — and every lookup is taken by the fused path."""
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches, kinds = random_model(seed)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd, host_concat)
    want_form = {"dense64": 1, "dense32": 1, "bucket": 1, "mean": 2, "sum": 2, "scatter": 3,
                 "pass": 6 if host_concat == "external" else 4, "sum3d": 5}
    groups_with_lookup = {c.concat_group for c in built.spec.columns if c.form in (1, 2, 3)}
    assert [c.form for c in built.spec.columns] == [want_form[k] for k in kinds] or len(groups_with_lookup) < len(fetches)
    assert not built.skipped
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert e.shape == o.shape and np.array_equal(e, o)


@pytest.mark.parametrize("B,seed", [(23, 0), (1, 1), (70, 2)])
def test_sparse_reshape_is_folded_into_the_segment_ids(oracle, tmp_path, B, seed):
    """a12, SparseReshape (cuda_emitter.cc:1874-1916): the row coordinate of the reshaped element is an expression of the
    ORIGINAL coordinates — (sum idx_k * mul_k) // div — whenever the shapes' entries can be traced to constants or copies
    of tensor elements: the column then reads the original indices (stride = their rank) and the op leaves the data
    path; a reshape with an unprovable shape stays in TensorFlow and its output tensor is what ConcatInputs ships.
    Either way the rewritten graph equals the original bit for bit."""
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = sparse_reshape_model(B=B, seed=seed)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    assert [e.shape for e in expected] == [(B, 20), (2 * B, 12), (3 * B, 16), (4 * B, 24), (2 * B, 8)]
    built = build_plan(gd)
    c = built.spec.columns
    assert [x.form for x in c] == [2, 1] * 5 and not built.skipped
    sym = [s.tensor for s in built.symbols]
    # p: [B, L] -> [B, L], the identity: plain segment ids from the original indices
    assert built.host_inputs[c[0].seg_input] == ("p/indices", P.DT_INT64, 2) and c[0].seg_stride == 2 and not c[0].seg_mul
    # q: [B, 2L] -> [2B, L=6]: row = (idx0 * W + idx1) // 6, W = dense_shape[1] per request
    assert built.host_inputs[c[2].seg_input] == ("q/indices", P.DT_INT64, 2) and c[2].seg_stride == 2
    assert (tuple(c[2].seg_mul), c[2].seg_div, c[2].seg_sym_slot) == ((1, 1), 6, 0)
    assert (built.symbols[c[2].seg_sym].tensor, built.symbols[c[2].seg_sym].index) == ("q/dense_shape", 1)
    # r: [B, T, L] -> [B*T, L]: L cancels, row = idx0 * T + idx1, T = dense_shape[1] per request
    assert built.host_inputs[c[4].seg_input] == ("r/indices", P.DT_INT64, 2) and c[4].seg_stride == 3
    assert (tuple(c[4].seg_mul), c[4].seg_div, c[4].seg_sym_slot) == ((1, 1), 1, 0)
    assert (built.symbols[c[4].seg_sym].tensor, built.symbols[c[4].seg_sym].index) == ("r/dense_shape", 1)
    # s: constants [2B, 4, 6] -> [4B, 12]: row = (idx0 * 4 + idx1) // 2
    assert built.host_inputs[c[6].seg_input] == ("s/indices", P.DT_INT64, 2) and c[6].seg_stride == 3
    assert (tuple(c[6].seg_mul), c[6].seg_div, c[6].seg_sym) == ((4, 1), 2, -1)
    # u: new_shape is fed: computed by TensorFlow
    assert built.host_inputs[c[8].seg_input] == ("u/SparseReshape", P.DT_INT64, 2) and c[8].seg_stride == 2 and not c[8].seg_mul
    assert sym == ["p/num_segments", "q/dense_shape", "q/num_segments", "r/dense_shape", "r/num_segments", "s/num_segments",
                   "u/num_segments"]
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    assert open(path).read().startswith("fcp_plan 4\n")
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    names = {n.name for n in out.node}
    assert "u/SparseReshape" in names and "u/added_strided_slice" not in names     # its indices are shipped whole
    for k in "pqrs":
        assert f"{k}/SparseReshape" in names                # kept for num_segments (output 1) only ...
        assert f"{k}/added_strided_slice" not in names      # ... the indices output is no longer read
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert np.array_equal(e, o)


@pytest.mark.parametrize("B,seed", [(29, 0), (1, 1), (90, 2)])
def test_id_filter_ops_become_column_transforms(oracle, tmp_path, B, seed):
    """Addons>SelectValue / GatherIndiceValue / GatherValueGenIndice in front of a lookup are
    absorbed into the column plan (evaluated on the device next to Bucketize) and disappear from the
    rewritten graph; the rewritten graph equals the original bit for bit."""
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = id_filter_model(B=B, seed=seed)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    assert expected[0].shape == (B, 8 + 16 + 8 + 4 + 12 + 8 + 4)
    built = build_plan(gd)
    c = built.spec.columns
    assert [x.form for x in c] == [1, 2, 2, 1, 1, 2, 1] and not built.skipped
    # AsString -> StringToHashBucketFast over integer ids: hashed on the device, the raw ids are shipped
    assert c[5].hash_buckets == 100 and built.host_inputs[c[5].ids_input] == ("f/values", P.DT_INT64, 1) and c[5].xform_mode == 0
    assert c[6].hash_buckets == 1000 and built.host_inputs[c[6].ids_input] == ("h_ids", P.DT_INT32, 1)
    assert (c[6].xform_mode, c[6].xform_lo, c[6].xform_hi, c[6].id_source) == (PL.XFORM_SELECT, (100,), (899,), PL.IDS_I32)
    assert (c[0].xform_mode, c[0].xform_lo, c[0].xform_hi, c[0].xform_substitute) == (PL.XFORM_SELECT, (10, 80), (60, 90), 3)
    assert built.host_inputs[c[0].ids_input][0] == "a_ids"
    assert (c[1].xform_mode, c[1].xform_lo, c[1].xform_hi) == (PL.XFORM_FILTER, (20,), (150,))
    assert built.host_inputs[c[1].ids_input][0] == "b/values" and built.host_inputs[c[1].seg_input][0] == "b/indices"
    assert c[1].seg_stride == 2 and c[1].combiner == PL.COMBINER_MEAN
    assert c[2].id_source == PL.IDS_F32_BUCKETIZE and c[2].xform_mode == PL.XFORM_FILTER and c[2].xform_lo == (5,)
    assert built.host_inputs[c[2].ids_input][0] == "c/values" and built.host_inputs[c[2].seg_input][0] == "c/indices"
    # GatherValueGenIndice + ScatterNd: a one-hot gather over the original values with a filter
    assert c[3].form == PL.FORM_GATHER and c[3].xform_mode == PL.XFORM_FILTER and built.host_inputs[c[3].ids_input][0] == "d_ids"
    # two transforms: the outer one is fused, the inner op keeps running in TensorFlow
    assert c[4].xform_mode == PL.XFORM_SELECT and c[4].xform_substitute == 2 and built.host_inputs[c[4].ids_input][0] == "e/inner"
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    again = load_plan(path)
    assert [(x.xform_mode, x.xform_lo, x.xform_hi, x.xform_substitute, x.hash_buckets) for x in again.columns] == \
           [(x.xform_mode, x.xform_lo, x.xform_hi, x.xform_substitute, x.hash_buckets) for x in c]
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    ops = [n.op for n in out.node]
    assert ops.count("Addons>SelectValue") == 1 and "Addons>GatherIndiceValue" not in ops and "Addons>GatherValueGenIndice" not in ops
    assert "AsString" not in ops and "StringToHashBucketFast" not in ops
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert np.array_equal(expected[0], got[0])


def test_resource_variable_tables(oracle, tmp_path):
    """TF2 SavedModels keep embedding tables in resource variables: VarHandleOp + ResourceGather /
    ReadVariableOp.  The plan builder takes them like VariableV2 tables; FeatureColumnProcess receives the
    table VALUES (a ReadVariableOp output: an existing reader, or one the rewrite adds)."""
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = resource_variable_model()
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd)
    c = built.spec.columns
    assert [x.form for x in c] == [1, 2, 1, 2] and not built.skipped
    assert c[0].table_input == c[2].table_input == c[3].table_input != c[1].table_input     # table a bound once
    assert [t for t, _, _ in built.device_inputs] == ["d/Read", "b/Read"]                  # values, not handles
    path = str(tmp_path / "m.fcp")
    save_plan(built.spec, path)
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    nodes = {n.name: n for n in out.node}
    assert list(nodes["FeatureColumnProcess"].input[3:5]) == ["d/Read", "b/Read"]
    assert not any(n.op in ("ResourceGather",) or n.op.startswith("SparseSegment") for n in out.node)
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert np.array_equal(expected[0], got[0])
    # a variable nobody reads as a tensor yet: the rewrite adds the reader
    for n in gd.node:
        if n.name == "d/SparseSegmentMeanWithNumSegments":
            n.input[0] = "b/Read"                              # (column d now pools table b; table a is gather-only)
    for i, n in enumerate(gd.node):
        if n.name == "d/Read":
            del gd.node[i]
            break
    feeds["d/values"] = feeds["d/values"] % 140
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd)
    assert [t for t, _, _ in built.device_inputs][0] == "input_layer/a_embedding/embedding_weights/fcp_read"
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    rd = {n.name: n for n in out.node}["input_layer/a_embedding/embedding_weights/fcp_read"]
    assert rd.op == "ReadVariableOp" and list(rd.input) == ["input_layer/a_embedding/embedding_weights"]
    save_plan(built.spec, path)
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert np.array_equal(expected[0], got[0])


def test_elem_source_proofs():
    """GraphView.elem_source: which shape entries are provably the same number."""
    from recom_amd.graph.view import GraphView
    gd, *_ = sparse_reshape_model()
    g = GraphView(gd)
    n = g.nodes
    assert g.elem_source(n["p/new_shape"], 0, 1) == ("elem", "p/dense_shape", 1) == g.elem_source(n["p/dense_shape"], 0, 1)
    assert g.elem_source(n["p/new_shape"], 0, 0) == ("elem", "p/dense_shape", 0)        # Prod over one element
    assert g.elem_source(n["q/new_shape"], 0, 1) == ("const", 6)
    assert g.elem_source(n["p/SparseReshape"], 1, 0) == ("elem", "p/dense_shape", 0)
    assert g.static_shape(n["p/SparseReshape"], 0) == [None, 2] and g.static_shape(n["p/SparseReshape"], 1) == [2]


def test_nothing_to_fuse():
    g = P.GraphDef()
    n = g.node.add(name="x", op="Placeholder")
    n.attr["dtype"].type = P.DT_FLOAT
    with pytest.raises(Unsupported):
        build_plan(g)


def test_cli(tmp_path):
    gd, *_ = canonical_model()
    src = tmp_path / "model.pb"
    src.write_bytes(gd.SerializeToString())
    r = subprocess.run([sys.executable, "-m", "recom_amd.graph", str(src), "--plan", str(tmp_path / "m.fcp"),
                        "--out", str(tmp_path / "out.pbtxt")], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "2 concat group(s), 11 column(s)" in r.stdout
    assert load_plan(str(tmp_path / "m.fcp")).n_columns == 11
    out = parse_graphdef((tmp_path / "out.pbtxt").read_bytes())
    assert any(n.op == "Addons>FeatureColumnProcessWithSymbols" for n in out.node)


def test_graph_evaluator_reproduces_the_tensorflow_documented_examples():
    """oracle/tf_graph_eval.py (the stand-in for TF-CPU that the rewritten graphs are diffed against) evaluated on
    one-op GraphDefs of the TensorFlow API documentation's worked examples (tests/golden/tf_doc_examples.py):
    vectors it did not produce itself."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import tf_doc_examples as T
    from graph_fixtures import GB
    from recom_amd.graph import tf_proto as P
    from tf_graph_eval import GraphEvaluator

    def run(build, fetch="out"):
        g = GB()
        build(g)
        return GraphEvaluator(g.gd, {}).run([fetch], {})[0]

    b = T.BUCKETIZE
    got = run(lambda g: g.node("out", "Bucketize", [g.const("x", b["values"])], T=("type", P.DT_FLOAT),
                               boundaries=("floats", [float(v) for v in b["boundaries"]])))
    assert np.array_equal(got, b["expected"])
    ga = T.GATHER
    got = run(lambda g: g.gather("out", g.const("p", ga["params"]), g.const("i", ga["indices"]), np.int64))
    assert np.array_equal(got, ga["expected"])
    for case in T.SPARSE_SEGMENT_SUM + [dict(T.SPARSE_SEGMENT_MEAN, mean=True)]:
        op = "SparseSegmentMeanWithNumSegments" if case.get("mean") else "SparseSegmentSumWithNumSegments"
        got = run(lambda g: g.node("out", op, [g.const("d", case["data"]), g.const("i", np.asarray(case["indices"], np.int64)),
                                               g.const("s", np.asarray(case["segment_ids"], np.int64)),
                                               g.const("n", np.asarray(case["num_segments"], np.int32))],
                                   T=("type", P.DT_FLOAT), Tidx=("type", P.DT_INT64), Tnumsegments=("type", P.DT_INT32),
                                   Tsegmentids=("type", P.DT_INT64)))
        assert np.array_equal(got, np.asarray(case["expected"], np.float32)), case
    sc = T.SCATTER_ND
    got = run(lambda g: g.node("out", "ScatterNd", [g.const("i", np.asarray(sc["indices"], np.int64).reshape(-1, 1)),
                                                    g.const("u", np.asarray(sc["updates"], np.float32)),
                                                    g.const("s", np.asarray([sc["size"]], np.int64))],
                               T=("type", P.DT_FLOAT), Tindices=("type", P.DT_INT64)))
    assert np.array_equal(got, np.asarray(sc["expected"], np.float32))
    c = T.CONCAT
    got = run(lambda g: g.node("out", "ConcatV2", [g.const("a", c["inputs"][0]), g.const("b", c["inputs"][1]),
                                                   g.const("axis", np.asarray(1, np.int32))], T=("type", P.DT_FLOAT), N=2))
    assert np.array_equal(got, c["expected"])


@pytest.mark.parametrize("which", ["canonical", "random1", "random5", "id_filter", "sparse_reshape"])
def test_staged_rewrite_matches_the_original_graph(oracle, tmp_path, which):
    """`python -m recom_amd.graph --staged`: the staged plan + its stage section + a rewritten graph whose ConcatInputs
    node carries `_fcp_plan` and receives the symbols vector; evaluated with the staged ConcatInputs and the C oracle on
    the staged plan it equals the original graph bit for bit, and the blob is smaller than the reference's byte copy."""
    from tf_graph_eval import GraphEvaluator
    from recom_amd.graph.__main__ import main
    from recom_amd.graph import load_graphdef, save_graphdef
    from recom_amd.plan_io import load_stage
    if which == "canonical":
        gd, feeds, variables, fetches = canonical_model(B=41, seed=2)
    elif which.startswith("random"):
        gd, feeds, variables, fetches, _ = random_model(int(which[6:]))
    elif which == "id_filter":
        gd, feeds, variables, fetches = id_filter_model(B=37, seed=1)
    else:
        gd, feeds, variables, fetches = sparse_reshape_model(B=23, seed=0)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    src, plan, out = str(tmp_path / "m.pb"), str(tmp_path / "m.fcp"), str(tmp_path / "m_fcp.pb")
    save_graphdef(gd, src)
    assert main([src, "--plan", plan, "--out", out, "--staged"]) == 0
    stage = load_stage(plan)
    built = build_plan(gd)
    assert stage is not None and len(stage.modes) == len(built.host_inputs) + (stage.symbols_input >= 0)
    out_gd = load_graphdef(out)
    ci = {n.name: n for n in out_gd.node}["ConcatInputs"]
    assert ci.op == "Addons>ConcatInputs" and ci.attr["_fcp_plan"].s.decode() == plan
    assert len(ci.input) == len(stage.modes) and (stage.symbols_input < 0 or ci.input[-1] == "FeatureColumnProcess/symbols")
    got = GraphEvaluator(out_gd, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert e.shape == o.shape and np.array_equal(e, o)


def _fixture_graphs():
    yield "canonical", canonical_model(B=41, seed=2)[0]
    yield "microbenchmark", microbenchmark_model(columns=12, B=32, seed=1)[0]
    yield "id_filter", id_filter_model(B=37, seed=1)[0]
    yield "sparse_reshape", sparse_reshape_model(B=23, seed=0)[0]
    yield "resource_variable", resource_variable_model(B=16, seed=3)[0]
    yield "unsupported-in-part", canonical_model(B=9, seed=0, unsupported=True)[0]
    for seed in range(24):
        yield f"random{seed}", random_model(seed)[0]


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
@pytest.mark.parametrize("staged", [False, True])
def test_native_plan_builder_matches_the_python_one(tmp_path, host_concat, staged):
    """the plan builder behind the C ABI (fcp_graph_build, recom_amd/csrc/fcp_graph.cc: own GraphDef
    wire reader / writer, no protobuf library, no Python) — what the retained Grappler pass calls in place of
    CudaEmitter::Optimize (cuda_emitter.cc:80-116).  On every fixture graph both builders write the SAME plan file, byte
    for byte, and rewritten graphs that are equal as GraphDef messages."""
    from recom_amd.graph import native_build
    n = 0
    for name, gd in _fixture_graphs():
        data = gd.SerializeToString()
        py_plan, c_plan = str(tmp_path / f"{name}.py.fcp"), str(tmp_path / f"{name}.c.fcp")
        built = build_plan(gd, host_concat)
        stage = None
        if staged:
            spec, stage = built.spec.staged_for_concat_inputs()
            save_plan(spec, py_plan, stage)
        else:
            save_plan(built.spec, py_plan)
        c_graph, desc = native_build(data, c_plan, host_concat, staged)
        assert open(c_plan).read() == open(py_plan).read(), name
        # the rewritten graphs name their plan file: give both the same name before comparing
        want = rewrite_graph(gd, built, c_plan, stage=stage)
        got = parse_graphdef(c_graph)
        assert len(got.node) == len(want.node), name
        for a, b in zip(got.node, want.node):
            assert a == b, (name, a.name, b.name)
        assert got == want, name
        assert desc.splitlines()[0] == built.describe().splitlines()[0], name
        n += 1
    assert n >= 30


def test_native_plan_builder_errors(tmp_path):
    from recom_amd.graph import native_build
    from recom_amd.lib import FcpError
    g = P.GraphDef()
    g.node.add(name="x", op="Placeholder").attr["dtype"].type = P.DT_FLOAT
    with pytest.raises(Unsupported):                               # nothing to fuse: the pass leaves the graph alone
        native_build(g.SerializeToString(), str(tmp_path / "x.fcp"))
    with pytest.raises(FcpError):                                  # not a GraphDef
        native_build(b"\xff\xff\xff\xff\x01", str(tmp_path / "y.fcp"))
    gd = canonical_model()[0]
    with pytest.raises(FcpError):                                  # the plan file cannot be written
        native_build(gd.SerializeToString(), str(tmp_path / "no" / "such" / "dir" / "z.fcp"))
    graph, _ = native_build(gd.SerializeToString(), str(tmp_path / "k.fcp"), prune=False)
    assert len(parse_graphdef(graph).node) > len(gd.node)          # nothing pruned: the original nodes + the new ones


@pytest.mark.parametrize("seed", range(60))
def test_random_sparse_reshapes_through_both_builders(oracle, tmp_path, seed):
    """The segment-id maps of random SparseReshapes (cuda_emitter.cc:1874-1916): whatever each reshape becomes — the
    identity, a map with constant factors, one with a run-time factor, or an op left to TensorFlow — the rewritten graph
    (C oracle behind the ops) equals the original graph evaluated op by op (NumPy ravel / unravel), and the C++ builder
    writes the same plan file and graph as the Python one."""
    from graph_fixtures import random_sparse_reshape_model
    from recom_amd.graph import native_build
    from tf_graph_eval import GraphEvaluator
    gd, feeds, variables, fetches = random_sparse_reshape_model(seed)
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd)
    assert not built.skipped
    path, cpath = str(tmp_path / "m.fcp"), str(tmp_path / "c.fcp")
    save_plan(built.spec, path)
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert e.shape == o.shape and np.array_equal(e, o)
    c_graph, _ = native_build(gd.SerializeToString(), cpath)
    assert open(cpath).read() == open(path).read()
    assert parse_graphdef(c_graph) == rewrite_graph(gd, built, cpath)


def test_native_builder_and_plan_parser_survive_mutated_inputs(tmp_path):
    """The GraphDef wire reader of fcp_graph.cc and the plan-file parser of the library are hand-written: byte-level
    mutations of valid inputs (flips, truncations, spliced garbage, huge varints) must end in a status code or a valid
    result, never in a crash or a hang — `scripts/asan_host.sh` runs this under ASan + UBSan."""
    import ctypes as C
    from recom_amd import lib as _lib
    from recom_amd.graph import native_build
    from recom_amd.lib import FcpError
    rng = np.random.default_rng(0)
    L = _lib.load()
    seeds = [canonical_model(B=7, seed=1)[0], sparse_reshape_model(B=5, seed=2)[0], id_filter_model(B=6, seed=3)[0]]
    ok = bad = 0
    for gd in seeds:
        data = bytearray(gd.SerializeToString())
        for it in range(150):
            m = bytearray(data)
            kind = it % 5
            if kind == 0:
                for _ in range(int(rng.integers(1, 4))):
                    m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                m = m[:int(rng.integers(0, len(m)))]
            elif kind == 2:
                p = int(rng.integers(0, len(m)))
                m[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 12)), dtype=np.uint8))
            elif kind == 3:
                p = int(rng.integers(0, len(m)))
                m[p:p + 1] = b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\x01"     # a 10-byte varint where a byte was
            else:
                a, b = sorted(int(v) for v in rng.integers(0, len(m), 2))
                del m[a:b]
            try:
                native_build(bytes(m), str(tmp_path / "m.fcp"), staged=bool(it & 1))
                ok += 1
            except (FcpError, Unsupported):
                bad += 1
    assert ok + bad == 450 and bad > 50
    # crafted, well-formed input: a Const whose declared shape is astronomically large over a one-value splat
    # list — the GatherV2 axis, a reshape target — must neither overflow the element count nor make the reader allocate what
    # the shape claims; both builders treat such a Const as "not a constant the walk looks at" and agree on the outcome
    from recom_amd.graph import build_plan as py_build
    for dims in ([2 ** 61 + 1], [2 ** 34], [2 ** 32, 2 ** 32], [3, 2 ** 62], [1 << 21]):
        for gd0 in seeds[:2]:
            gd = parse_graphdef(gd0.SerializeToString())
            hit = 0
            for node in gd.node:
                if node.op == "Const" and node.attr["value"].tensor.dtype in (3, 9) and len(node.attr["value"].tensor.tensor_shape.dim) <= 1:
                    t = node.attr["value"].tensor
                    first = int(tensor_to_numpy(t).reshape(-1)[0]) if tensor_to_numpy(t).size else 0
                    t.ClearField("tensor_content")
                    del t.int_val[:]
                    del t.int64_val[:]
                    (t.int_val if t.dtype == 3 else t.int64_val).append(first)
                    del t.tensor_shape.dim[:]
                    for d in dims:
                        t.tensor_shape.dim.add().size = d
                    hit += 1
            assert hit
            outcomes = []
            for build in (lambda: native_build(gd.SerializeToString(), str(tmp_path / "g.fcp")), lambda: py_build(gd)):
                try:
                    build()
                    outcomes.append("ok")
                except (FcpError, Unsupported, ValueError) as e:
                    outcomes.append("refused")
            assert outcomes[0] == outcomes[1], (dims, outcomes)
    # plan files: the same treatment for fcp_plan_create_from_file / fcp_plan_file_stage_info (host-only: no device)
    built = build_plan(seeds[1])
    spec, stage = built.spec.staged_for_concat_inputs()
    path = str(tmp_path / "p.fcp")
    save_plan(spec, path, stage)
    text = open(path).read()
    toks = text.split()
    for it in range(300):
        t = list(toks)
        kind = it % 4
        if kind == 0:
            t[int(rng.integers(0, len(t)))] = str(int(rng.choice([-1, 0, 1 << 31, 1 << 40, -(1 << 40), 99999999])))
        elif kind == 1:
            t = t[:int(rng.integers(0, len(t)))]
        elif kind == 2:
            t.insert(int(rng.integers(0, len(t))), str(rng.choice(["stage", "segmaps", "x", "1e400", "nan", "-"])))
        else:
            del t[int(rng.integers(0, len(t)))]
        with open(path, "w") as f:
            f.write(" ".join(t) + "\n")
        h = C.c_void_p()
        rc = L.fcp_plan_create_from_file(path.encode(), 0, _lib.FLAG_HOST_ONLY, C.byref(h))
        if rc == 0:
            L.fcp_plan_destroy(h)
        n = C.c_int32()
        L.fcp_plan_file_stage_info(path.encode(), C.byref(n), None, None, 0, None)
