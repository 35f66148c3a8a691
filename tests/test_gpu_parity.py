"""GPU parity: the HIP path, called through the C ABI (libfcp_hip.so), against
the CPU oracle on the same seeded inputs, against the committed golden
fixtures, and — at BASELINE.json's full sizes — against the closed-form table
definition.  Bit-exact for index / copy work; pooled vectors are also compared
bit-exactly with the oracle (same sequential fp32 order) and within the north
star's 1e-5 max-abs-diff of the float64 expectation.
"""
import copy
import dataclasses

import sys

import numpy as np
import pytest

from conftest import GOLDEN_NAMES, check_against_expected

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from recom_amd import lib
    lib.load()  # fail loudly if the HIP extension is missing
    return torch


def run_gpu(torch, spec, inputs, tables_np, symbols, op=None, tables_dev=None):
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    dev = torch.device("cuda", 0)
    blob, offsets, shapes = concat_inputs(inputs)
    if tables_dev is None:
        tables_dev = [torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in tables_np]
    if op is None:
        op = FeatureColumnProcess(spec, 0)
    d_blob = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(0, dtype=torch.int8, device=dev)
    out = op(d_blob, offsets, shapes, tables_dev, symbols)
    torch.cuda.synchronize()
    return out, (blob, offsets, shapes), op


def assert_equal_oracle(oracle, spec, packed, tables_np, symbols, out, exact=True):
    blob, offsets, shapes = packed
    want, bad = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tables_np, symbols)
    for g, w in zip(out.groups, want):
        got = g.cpu().numpy()
        assert got.shape == w.shape
        if exact:
            assert np.array_equal(got, w), float(np.abs(got - w).max())
        else:
            assert np.abs(got - w).max(initial=0) < 1e-5
    return want, bad


def test_tensorflow_documented_examples_through_the_hip_path(torch_cuda):
    """The worked examples of the TensorFlow API documentation (tests/golden/tf_doc_examples.py; vectors neither
    this repository nor the reference produced) as one-column plans through the C ABI: Bucketize, GatherV2,
    SparseSegmentSum / Mean with and without missing segments, ScatterNd(GatherV2), ConcatV2 — the documented
    outputs, exactly.  No oracle involved."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import tf_doc_examples as T
    from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_GATHER, FORM_GATHER_SCATTER, FORM_PASSTHROUGH,
                                FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I64, ROWS_FROM_IDS, ROWS_FROM_INPUT_DIM0,
                                ROWS_FROM_SYMBOL, SEG_IDS_I64, SEG_NONE, ColumnSpec, PlanSpec)
    torch = torch_cuda

    def run(cols, ranks, esz, inputs, tables, symbols=None, n_symbols=0):
        spec = PlanSpec(cols, ranks, esz, len(tables), n_groups=1, n_symbols=n_symbols)
        spec.validate()
        out, _, _ = run_gpu(torch, spec, inputs, tables, None if symbols is None else np.asarray(symbols, np.int32))
        return out.groups[0].cpu().numpy()

    b = T.BUCKETIZE                                            # bucket index read back through a table whose row r is [r]
    ident = np.arange(4, dtype=np.float32).reshape(4, 1)
    got = run([ColumnSpec(FORM_GATHER, 1, 4, COMBINER_NONE, IDS_F32_BUCKETIZE, 0, 0, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0,
                          b["boundaries"], 0, 0)], [1], [4], [b["values"].ravel()], [ident])
    assert np.array_equal(got.ravel().astype(np.int32), b["expected"].ravel())
    g = T.GATHER
    got = run([ColumnSpec(FORM_GATHER, 3, 4, COMBINER_NONE, IDS_I64, 0, 0, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0, 0)],
              [1], [8], [g["indices"]], [g["params"]])
    assert np.array_equal(got, g["expected"])
    for case in T.SPARSE_SEGMENT_SUM + [dict(T.SPARSE_SEGMENT_MEAN, mean=True)]:
        comb = COMBINER_MEAN if case.get("mean") else COMBINER_SUM
        got = run([ColumnSpec(FORM_SEGMENT_REDUCE, 4, 3, comb, IDS_I64, 0, 0, 1, SEG_IDS_I64, 1, ROWS_FROM_SYMBOL, 0, None, 0, 0)],
                  [1, 1], [8, 8], [np.asarray(case["indices"], np.int64), np.asarray(case["segment_ids"], np.int64)],
                  [case["data"]], [case["num_segments"]], 1)
        assert np.array_equal(got, np.asarray(case["expected"], np.float32)), case
    sc = T.SCATTER_ND                                          # indices as documented: [4, 3, 1, 7], not sorted
    got = run([ColumnSpec(FORM_GATHER_SCATTER, 1, 4, COMBINER_NONE, IDS_I64, 0, 0, 1, SEG_IDS_I64, 1, ROWS_FROM_SYMBOL, 0, None, 0, 0)],
              [1, 1], [8, 8], [np.arange(len(sc["indices"]), dtype=np.int64), np.asarray(sc["indices"], np.int64)],
              [np.asarray(sc["updates"], np.float32).reshape(-1, 1)], [sc["size"]], 1)
    assert np.array_equal(got.ravel(), np.asarray(sc["expected"], np.float32))
    c = T.CONCAT
    got = run([ColumnSpec(FORM_PASSTHROUGH, 3, 0, COMBINER_NONE, 0, -1, k, -1, SEG_NONE, 1, ROWS_FROM_INPUT_DIM0, k, None, 0, k)
               for k in range(2)], [2, 2], [4, 4], c["inputs"], [])
    assert np.array_equal(got, c["expected"])


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_golden_through_c_abi(torch_cuda, oracle, golden, name):
    case = golden[0][name]
    spec = case.spec()
    out, packed, _ = run_gpu(torch_cuda, spec, case.inputs, case.tables, case.symbols)
    # the packer is bit-exact with the stored ConcatInputs outputs
    assert np.array_equal(packed[0], case.blob) and np.array_equal(packed[1], case.offsets)
    assert np.array_equal(packed[2], case.shapes)
    check_against_expected(case, [g.cpu().numpy() for g in out.groups])
    assert_equal_oracle(oracle, spec, packed, case.tables, case.symbols, out)


@pytest.mark.parametrize("batch", [1, 3, 64, 100, 257])
def test_mixed_model_batches(torch_cuda, oracle, batch):
    from recom_amd import synth
    m = synth.model_mixed(batch=batch, vocab=997)
    tabs = m.numpy_tables()
    op = None
    for seed in range(3):
        req = m.make_request(seed)
        out, packed, op = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols, op)
        assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)
        # output_shapes / output_ptrs contract of FeatureColumnProcess
        for k, c in enumerate(m.spec.columns):
            assert out.output_shapes[2 * k + 1] == c.dim
            col = out.column(k).cpu().numpy()
            off = m.spec.column_offsets()[k]
            assert np.array_equal(col, out.groups[c.concat_group].cpu().numpy()[:, off:off + c.dim])


def test_s1_configuration(torch_cuda, oracle):
    """BASELINE.json configs[0]: S1 100 columns, dim 16, vocab 10k, batch 128."""
    from recom_amd import synth
    m = synth.model_s1()
    tabs = m.numpy_tables()
    req = m.make_request(0)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert out.groups[0].shape == (128, 1600)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.parametrize("seg,prepass", [("csr", False), ("indices", False), ("rowids32", False),
                                         ("indices", True), ("rowids32", True)])
def test_ragged_dynamic_shapes(torch_cuda, oracle, monkeypatch, seg, prepass):
    """RAGGED (reduced to 64 columns): nnz re-drawn per request; more distinct
    shapes than descriptor slots, then a repeat (cache hit).  Segment-id encodings run both
    ways: row ranges searched inside the blocks (64 columns x 256 rows is under the
    threshold) and the ComputeSegmentOffsets pre-pass (forced, as for plans with many such
    columns)."""
    from recom_amd import synth
    if prepass:
        monkeypatch.setenv("FCP_SEG_PREPASS", "1")
    m = synth.model_ragged(columns=64, vocab=5000, batch=256, seg=seg)
    tabs = m.numpy_tables()
    dev_tabs = [torch_cuda.from_numpy(t).cuda() for t in tabs]
    op = None
    for seed in list(range(11)) + [3, 3, 0]:
        req = m.make_request(seed)
        out, packed, op = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols, op, dev_tabs)
        assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.parametrize("layout", ["by_position", "packed", "blob_grouped", "blob_grouped_staged", "mixed_plan", "mixed_encodings"])
def test_regular_csr_front_of_the_ragged_kernel(torch_cuda, oracle, monkeypatch, layout):
    """(r6) The ragged body asks for its row ranges TOGETHER with the column records when the CSR arrays are regular
    (FcpLaunch::csr_reg): the arena scratch laid out by column position and filled by the pre-pass (`by_position`; `packed` =
    the round-5 layout through FCP_DIAG=csr_by_pos=0: the ranges come through the records), or CSR inputs that lie one
    stride apart in the blob because the caller grouped them (`blob_grouped[_staged]`, recognised per request).  `mixed_plan`:
    pooled columns are fewer than half of the plan, so the scratch stays packed and a passthrough column sits between them.
    Empty rows, rows at the batch's end, more shapes than descriptor slots, a row count that is not a multiple of the block's
    four rows; all bit-exact against the oracle."""
    from recom_amd import synth
    monkeypatch.setenv("FCP_SEG_PREPASS", "1")                           # (64 columns x 255 rows would be searched in the blocks)
    if layout == "packed":
        monkeypatch.setenv("FCP_DIAG", "csr_by_pos=0")
    if layout == "mixed_encodings":
        # two thirds of the pooled columns bring SparseTensor indices (pre-pass, scratch), one third CSR offsets in the blob:
        # the shortcut serves a launch from ONE matrix, so such a plan keeps the packed scratch and every column reads its own
        from recom_amd.plan import COMBINER_MEAN, COMBINER_SUM
        b = synth._Builder()
        for c in range(33):
            synth._add_ragged(b, 3000, (8, 16, 32, 64)[c % 4], slot=c, combiner=COMBINER_SUM if c % 2 else COMBINER_MEAN,
                              seg="indices" if c % 3 else "csr")
        m = synth._finish("MIXENC", b, 255, n_symbols=1)
    elif layout == "mixed_plan":
        m = synth.model_mixed(batch=255, vocab=997, n_groups=1)
    elif layout.startswith("blob_grouped"):
        base = synth.model_ragged(columns=64, vocab=5000, batch=255, seg="indices" if layout.endswith("staged") else "csr")
        m = synth.grouped_csr_model(synth.staged_model(base) if layout.endswith("staged") else base)
    else:
        m = synth.model_ragged(columns=64, vocab=5000, batch=255, seg="indices")
    tabs = m.numpy_tables()
    dev_tabs = [torch_cuda.from_numpy(t).cuda() for t in tabs]
    op = None
    for seed in list(range(6)) + [2, 2]:
        req = m.make_request(seed)
        out, packed, op = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols, op, dev_tabs)
        assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)
    if layout.startswith("blob_grouped"):                                # the CSR arrays really are one stride apart in the blob
        offs = packed[1]
        seg = [c.seg_input for c in sorted(m.spec.columns, key=lambda c: c.concat_slot)]
        assert len(set(np.diff([int(offs[i]) for i in seg]))) == 1


@pytest.mark.parametrize("columns", [16, 900])
def test_zipf_and_long_bags(torch_cuda, oracle, columns):
    """Bags of up to 300 ids: what the wave's 384-entry tile cannot take in is staged bag by bag in further
    chunks (same order of adds: bit-exact); 900 columns x 40 rows exceeds the in-block search threshold, so that
    plan takes the pre-pass by itself."""
    from recom_amd import synth
    m = synth.model_ragged(columns=columns, vocab=3000, batch=40, seg="indices", max_len=300, dist="zipf")
    tabs = m.numpy_tables()
    req = m.make_request(5)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)
    # fp32 roundoff of 300-long sums of |x|<1 values stays well inside 1e-4
    import fcp_oracle as O
    truth = O.np_process_feature_columns(m.spec.to_dict(), *packed, tabs, req.symbols)
    assert np.abs(out.groups[0].cpu().numpy() - truth[0]).max() < 1e-4


@pytest.mark.parametrize("dims", [(6, 10, 2, 14), (3, 5, 1, 7), (4, 6, 8, 10)])
def test_narrow_vector_widths(torch_cuda, oracle, dims):
    """dims that are not multiples of 4 select the 8-byte / 4-byte slot kernels."""
    from recom_amd import synth
    m = synth.model_ragged(columns=8, vocab=500, batch=37, seg="csr", dims=dims)
    tabs = m.numpy_tables()
    req = m.make_request(1)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)
    d = synth.model_s2(columns=12, vocab=300, batch=70, dims=dims)
    tabs = d.numpy_tables()
    req = d.make_request(2)
    out, packed, _ = run_gpu(torch_cuda, d.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, d.spec, packed, tabs, req.symbols, out)


def test_per_column_layout_and_concat_outputs(torch_cuda, oracle, ref_alignmem):
    """FCP_LAYOUT_PER_COLUMN reproduces the reference arena (128-byte aligned
    per-column buffers, cuda_emitter.cc:967-969, :2151-2179: offsets step by the
    REFERENCE's own `alignmem`, compiled from its source); ConcatOutputs
    (concat_outputs_op_gpu.cu.cc:85-131) then yields the fused result."""
    from recom_amd import synth
    from recom_amd.ops import concat_outputs
    from recom_amd.plan import LAYOUT_PER_COLUMN
    m = synth.model_mixed(batch=50, vocab=997, n_groups=1)
    tabs = m.numpy_tables()
    req = m.make_request(4)
    fused, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    spec_pc = m.spec.with_layout(LAYOUT_PER_COLUMN)
    pc, _, _ = run_gpu(torch_cuda, spec_pc, req.inputs, tabs, req.symbols)
    # arena layout: prefix sums of alignmem(rows*dim*4)
    base = pc.buffer.data_ptr()
    cursor = 0
    for k, c in enumerate(spec_pc.columns):
        assert int(pc.output_ptrs[k]) - base == cursor
        assert pc.output_row_strides[k] == c.dim
        cursor += ref_alignmem(50 * c.dim * 4)
    order = sorted(range(spec_pc.n_columns), key=lambda k: spec_pc.columns[k].concat_slot)
    cat = concat_outputs([pc.column(k) for k in order])
    torch_cuda.cuda.synchronize()
    assert np.array_equal(cat.cpu().numpy(), fused.groups[0].cpu().numpy())
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, fused)


def test_concat_outputs_many_inputs(torch_cuda, oracle):
    from recom_amd.ops import concat_outputs
    rng = np.random.default_rng(0)
    xs = [rng.standard_normal((19, int(d))).astype(np.float32) for d in rng.integers(1, 40, 450)]
    got = concat_outputs([torch_cuda.from_numpy(x).cuda() for x in xs])
    torch_cuda.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), oracle.concat_outputs(xs))


def test_bad_ids_read_as_zero_and_are_counted(torch_cuda, oracle):
    from recom_amd import synth
    from recom_amd.plan import FLAG_COUNT_BAD_IDS
    m = synth.model_mixed(batch=64, vocab=997, n_groups=1)
    spec = dataclasses.replace(m.spec, flags=FLAG_COUNT_BAD_IDS)
    tabs = m.numpy_tables()
    req = m.make_request(7)
    inputs = [a.copy() for a in req.inputs]
    c0 = spec.columns[0]       # dense int64 gather
    inputs[c0.ids_input][[0, 5]] = [-1, 997]
    c3 = spec.columns[3]       # ragged csr sum
    inputs[c3.ids_input][:3] = [10 ** 12, -7, 1 << 40]
    out, packed, op = run_gpu(torch_cuda, spec, inputs, tabs, req.symbols)
    want, bad = assert_equal_oracle(oracle, spec, packed, tabs, req.symbols, out)
    assert bad == 5
    assert op.plan.read_bad_ids() == 5
    assert not out.column(0).cpu().numpy()[[0, 5]].any()


def test_error_codes(torch_cuda):
    from recom_amd import lib, synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    m = synth.model_mixed(batch=16, vocab=97, n_groups=1)
    tabs = [torch_cuda.from_numpy(t).cuda() for t in m.numpy_tables()]
    op = FeatureColumnProcess(m.spec, 0)
    req = m.make_request(0)
    blob, offsets, shapes = concat_inputs(req.inputs)
    d_blob = torch_cuda.from_numpy(blob).cuda()
    with pytest.raises(lib.FcpError) as e:  # missing symbols
        op(d_blob, offsets, shapes, tabs, None)
    assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT
    with pytest.raises(lib.FcpError) as e:  # groups disagree on the row count
        op(d_blob, offsets, shapes, tabs, np.asarray([17], np.int32))
    assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH
    with pytest.raises(lib.FcpError) as e:  # blob shorter than the shapes say
        op(d_blob[:100], offsets, shapes, tabs, req.symbols)
    assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH
    with pytest.raises(lib.FcpError) as e:  # wrong table shape
        op(d_blob, offsets, shapes, [t[:-1] for t in tabs], req.symbols)
    assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH
    out = op(d_blob, offsets, shapes, tabs, req.symbols)  # still usable afterwards
    torch_cuda.cuda.synchronize()
    assert out.groups[0].shape[0] == 16


def test_streams_share_a_plan(torch_cuda, oracle):
    """Re-entrancy: one immutable plan used from several streams."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    m = synth.model_ragged(columns=32, vocab=2000, batch=128, seg="indices")
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    streams = [torch.cuda.Stream() for _ in range(3)]
    reqs = [m.make_request(s) for s in range(6)]
    packed = [concat_inputs(r.inputs) for r in reqs]
    blobs = [torch.from_numpy(p[0]).cuda() for p in packed]
    torch.cuda.synchronize()
    outs = []
    for i, (r, p, b) in enumerate(zip(reqs, packed, blobs)):
        s = streams[i % 3]
        with torch.cuda.stream(s):
            outs.append(op(b, p[1], p[2], tabs, r.symbols, stream=s.cuda_stream))
    torch.cuda.synchronize()
    for r, p, o in zip(reqs, packed, outs):
        assert_equal_oracle(oracle, m.spec, p, tabs_np, r.symbols, o)


def test_row_sharded_partials_and_finalize(torch_cuda, oracle):
    """SHARD semantics on one GPU: world=4 plans over row shards produce partial
    sums; fcp_shard_finalize adds the slices in rank order and applies the mean.
    Dense columns are exact; pooled ones within 1e-5 of the unsharded result."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=48, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    req = m.make_request(3)
    full, packed, _ = run_gpu(torch, m.spec, req.inputs, tabs_np, req.symbols)
    world = 4
    d_blob = torch.from_numpy(packed[0]).cuda()
    parts, ops, shard_tabs = [], [], []
    for rank in range(world):
        spec = m.spec.with_shard(rank, world)
        tabs = [torch.from_numpy(np.ascontiguousarray(t[rank::world])).cuda() for t in tabs_np]
        op = FeatureColumnProcess(spec, 0)
        out = op(d_blob, packed[1], packed[2], tabs, req.symbols)
        torch.cuda.synchronize()
        # per-rank partials equal the sharded oracle bit for bit
        want, _ = oracle.process_feature_columns(spec.to_dict(), *packed, [t[rank::world] for t in tabs_np],
                                                 req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
        parts.append(out.groups[0].clone())
        ops.append(op)
        shard_tabs.append(tabs)
    stacked = torch.stack(parts)  # [world, rows, width]
    rows = stacked.shape[1]
    # every rank finalizes its batch slice; here rank 1's slice of a 4-way split
    lo, cnt = rows // 4, rows // 4
    sl = stacked[:, lo:lo + cnt, :].contiguous()
    fin = ops[1].shard_finalize(d_blob, packed[1], packed[2], shard_tabs[1], req.symbols, 0, sl, world, lo, cnt)
    torch.cuda.synchronize()
    got = fin.cpu().numpy()
    ref = full.groups[0].cpu().numpy()[lo:lo + cnt]
    offs = m.spec.column_offsets()
    for k, c in enumerate(m.spec.columns):
        a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
        if c.form in (1, 3, 4, 5):  # exactly one owner (table-free columns: rank 0): exact
            assert np.array_equal(a, b)
        else:
            assert np.abs(a - b).max() < 1e-5


def test_column_sharded_blocks_concat(torch_cuda, oracle):
    """Column (table-wise) sharding on one GPU: world=3 sub-plans over whole columns
    produce column blocks; the batch slices of the blocks, put side by side by
    fcp_concat_outputs, are bit-identical to the unsharded result (pooled columns too)."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs, concat_outputs
    from recom_amd.shard import assign_columns, batch_slices
    torch = torch_cuda
    m = synth.model_mixed(batch=50, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    req = m.make_request(5)
    full, _, _ = run_gpu(torch, m.spec, req.inputs, tabs_np, req.symbols)
    world = 3
    assignment = assign_columns(m.spec, world)
    blocks = []
    for rank in range(world):
        sub = m.spec.column_subset(assignment[rank])
        packed = concat_inputs([req.inputs[i] for i in sub.host_inputs])
        tabs = [torch.from_numpy(tabs_np[i]).cuda() for i in sub.device_inputs]
        op = FeatureColumnProcess(sub.spec, 0)
        out = op(torch.from_numpy(packed[0]).cuda(), packed[1], packed[2], tabs, req.symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(sub.spec.to_dict(), *packed, [tabs_np[i] for i in sub.device_inputs],
                                                 req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
        blocks.append(out.groups[0].clone())
    ref = full.groups[0].cpu().numpy()
    for begin, count in batch_slices(ref.shape[0], world):
        got = concat_outputs([b[begin:begin + count].contiguous() for b in blocks])
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), ref[begin:begin + count])


def _graph_through_hip(torch, gd, feeds, variables, fetches, tmp_path, host_concat="passthrough", staged=False, private_streams=0):
    """original graph in NumPy vs rewritten graph with the HIP path behind the Addons> ops.  ``staged``: the graph is
    rewritten for the staged plan (`python -m recom_amd.graph --staged`): ConcatInputs packs as the plan file's stage
    section says (what the shim does with the node's `_fcp_plan` attr)."""
    from tf_graph_eval import GraphEvaluator
    from recom_amd.graph import build_plan, parse_graphdef, rewrite_graph
    from recom_amd.ops import ConcatInputs, ConcatOutputs, FeatureColumnProcess
    from recom_amd.plan_io import load_plan, save_plan
    expected = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd, host_concat)
    path = str(tmp_path / "model.fcp")
    stage = None
    if staged:
        spec, stage = built.spec.staged_for_concat_inputs()
        save_plan(spec, path, stage)
    else:
        save_plan(built.spec, path)
    out_gd = parse_graphdef(rewrite_graph(gd, built, path, stage=stage).SerializeToString())
    ops = {}

    def concat_inputs_node(node, x):
        plan = node.attr["_fcp_plan"].s.decode() if "_fcp_plan" in node.attr else None
        assert (plan is not None) == staged
        return list(ConcatInputs(list(node.attr["ranks"].list.i), plan)(x))

    out_cols = built.spec.output_columns()

    def process(node, x):
        if node.name not in ops:                # what the shim does with the `dlpath` attr
            ops[node.name] = FeatureColumnProcess.from_plan_file(node.attr["dlpath"].s.decode(), 0)
        op = ops[node.name]
        n_tab = len(node.attr["input_types"].list.type)
        tables = [torch.from_numpy(np.ascontiguousarray(t)).cuda() for t in x[3:3 + n_tab]]
        symbols = x[3 + n_tab] if node.op.endswith("WithSymbols") else None
        assert symbols is None or symbols.dtype == np.int32
        if private_streams:
            # the shim with FCP_PRIVATE_STREAMS: the lookup runs on a plan-owned stream, nothing here waits for it — the
            # ConcatOutputs node below does (fcp_result_wait); the op's stream is kept busy so that the reader would race
            if not getattr(op.plan, "private_streams", 0):
                op.plan.set_private_streams(private_streams, always=True)
            blob = torch.from_numpy(x[0]).cuda()
            torch.cuda.synchronize()
            if hasattr(torch.cuda, "_sleep"):
                torch.cuda._sleep(2_000_000)
            res = op(blob, x[1], x[2], tables, symbols, defer_wait=True)
        else:
            res = op(torch.from_numpy(x[0]).cuda(), x[1], x[2], tables, symbols)
            torch.cuda.synchronize()
        # the op's outputs: one pointer / shape pair per OUTPUT column (external slots are not outputs)
        assert len(out_cols) == len(node.attr["output_types"].list.type)
        ptrs = res.output_ptrs[out_cols]
        shapes = res.output_shapes.reshape(-1, 2)[out_cols].reshape(-1)
        return [ptrs, shapes, res.buffer]

    def concat_outputs(node, x):
        a = node.attr
        n = int(a["N"].i)
        op = ConcatOutputs(n, list(a["embedd_dims"].list.i), list(a["device_input_indices"].list.i),
                           list(a["device_concat_indices"].list.i), list(a["host_concat_indices"].list.i),
                           int(a["prefix_begin"].i), int(a["prefix_end"].i))
        out = op(x[0], x[1], x[2:2 + n], x[-1])          # tensor_buffers[-1] = FeatureColumnProcess:2, the arena
        torch.cuda.synchronize()
        return [out.cpu().numpy()]

    custom = {"Addons>ConcatInputs": concat_inputs_node,
              "Addons>FeatureColumnProcess": process, "Addons>FeatureColumnProcessWithSymbols": process,
              "Addons>ConcatOutputsNoHost": concat_outputs, "Addons>ConcatOutputs": concat_outputs}
    got = GraphEvaluator(out_gd, variables, custom).run(fetches, feeds)
    for e, o in zip(expected, got):
        assert e.shape == o.shape and np.array_equal(e, o)
    return built


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
@pytest.mark.parametrize("B,seed", [(19, 0), (300, 3)])
def test_graphdef_to_hip_path(torch_cuda, tmp_path, B, seed, host_concat):
    """SURVEY §8f-1 end to end: a GraphDef in the reference's canonical rewritten form →
    plan builder → plan file → rewritten graph whose three Addons> ops run the HIP path;
    the concat outputs equal the original graph evaluated op by op in NumPy (TF-CPU
    semantics, fp32 adds in id order) bit for bit."""
    from graph_fixtures import canonical_model
    gd, feeds, variables, fetches = canonical_model(B=B, seed=seed)
    built = _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path, host_concat)
    assert built.spec.n_columns == 11 and built.spec.n_groups == 2


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
def test_graphdef_to_hip_path_with_private_streams(torch_cuda, tmp_path, host_concat):
    """The rewritten graph as the shim runs it with FCP_PRIVATE_STREAMS=3: Addons>FeatureColumnProcess enqueues on a
    plan-owned stream and returns, Addons>ConcatOutputs[NoHost] makes ITS stream wait (fcp_result_wait; with host inputs,
    fcp_concat_outputs_host writes the external slots behind the same wait) — equal to the original graph bit for bit."""
    from graph_fixtures import canonical_model
    for B, seed in ((300, 3), (19, 0)):
        gd, feeds, variables, fetches = canonical_model(B=B, seed=seed)
        _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path, host_concat, private_streams=3)


@pytest.mark.parametrize("host_concat", ["passthrough", "external"])
@pytest.mark.parametrize("seed", [0, 3, 7, 10])
def test_random_graphdefs_to_hip_path(torch_cuda, tmp_path, seed, host_concat):
    from graph_fixtures import random_model
    gd, feeds, variables, fetches, _ = random_model(seed)
    _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path, host_concat)


@pytest.mark.parametrize("which", ["canonical", "random3", "random7", "id_filter", "sparse_reshape", "microbenchmark"])
def test_graphdef_to_hip_path_through_the_staged_concat_inputs(torch_cuda, tmp_path, which):
    """The RAGGED config's real path: the reference delivers SparseTensor indices on the HOST (ConcatInputs is a CPU op,
    concat_inputs_ops.cc:42-77), so the staged Addons>ConcatInputs turns them into row offsets (and int64 ids into int32)
    while it packs, and the device runs neither the pre-pass nor a search.  GraphDef -> staged plan + stage section ->
    rewritten graph -> HIP path; equal to the original graph (NumPy, TF-CPU semantics) bit for bit."""
    import graph_fixtures as F
    if which == "canonical":
        gd, feeds, variables, fetches = F.canonical_model(B=300, seed=3)
    elif which.startswith("random"):
        gd, feeds, variables, fetches, _ = F.random_model(int(which[6:]))
    elif which == "id_filter":
        gd, feeds, variables, fetches = F.id_filter_model(B=120, seed=1)
    elif which == "sparse_reshape":
        gd, feeds, variables, fetches = F.sparse_reshape_model(B=45, seed=4)
    else:
        gd, feeds, variables, fetches = F.microbenchmark_model(columns=30, B=64, seed=2)
    _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path, staged=True)


@pytest.mark.parametrize("staged", [False, True])
def test_plain_sparse_segment_graph_to_hip_path(torch_cuda, tmp_path, staged):
    """SparseSegmentSum / SparseSegmentMean without num_segments (cuda_emitter.cc:1096-1113): rows = last segment id + 1 is
    a symbol the rewritten graph computes on the host; GraphDef -> plan -> HIP equals the original graph bit for bit."""
    from graph_fixtures import plain_segment_model
    for B, seed in ((21, 0), (1, 1), (300, 2)):
        gd, feeds, variables, fetches = plain_segment_model(B=B, seed=seed)
        built = _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path, staged=staged)
        assert [c.form for c in built.spec.columns] == [1, 2, 2] and not built.skipped


def test_sparse_reshape_graph_to_hip_path(torch_cuda, tmp_path):
    """a12 on the GPU (cuda_emitter.cc:1874-1916): GraphDef -> plan -> HIP with segment ids read through SparseReshapes
    that are the identity (plain ids from the original indices), folded into a segment-id map (run-time factor, rank-3
    safe_embedding_lookup_sparse, constant shapes) or unprovable (computed by TensorFlow, shipped); five concat groups
    with different row counts; the rewritten graph equals the original bit for bit."""
    from graph_fixtures import sparse_reshape_model
    for B, seed in ((45, 4), (1, 0), (260, 7)):
        gd, feeds, variables, fetches = sparse_reshape_model(B=B, seed=seed)
        built = _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path)
        assert built.spec.n_groups == 5 and [c.form for c in built.spec.columns] == [2, 1] * 5
        assert [len(c.seg_mul) for c in built.spec.columns[::2]] == [0, 2, 2, 2, 0]


def test_id_filter_graph_to_hip_path(torch_cuda, tmp_path):
    """SURVEY 8f-3 end to end: the reference's CPU id ops in the graph -> column transforms -> evaluated by
    the kernels; equal to the original graph (NumPy) bit for bit."""
    from graph_fixtures import id_filter_model
    for B, seed in ((29, 0), (300, 3)):
        gd, feeds, variables, fetches = id_filter_model(B=B, seed=seed)
        built = _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path)
        assert [c.xform_mode for c in built.spec.columns] == [1, 2, 2, 2, 1, 0, 1]
        assert [c.hash_buckets for c in built.spec.columns] == [0, 0, 0, 0, 0, 100, 1000]


def test_resource_variable_graph_to_hip_path(torch_cuda, tmp_path):
    """TF2-style graph (VarHandleOp / ReadVariableOp / ResourceGather tables) end to end on the GPU."""
    from graph_fixtures import resource_variable_model
    gd, feeds, variables, fetches = resource_variable_model(B=64, seed=2)
    built = _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path)
    assert [c.form for c in built.spec.columns] == [1, 2, 1, 2]


def test_concat_outputs_host_inputs_into_external_slots(torch_cuda, oracle):
    """Addons>ConcatOutputs with N > 0 (concat_outputs_op_gpu.cu.cc:186-216): the plan reserves
    FORM_EXTERNAL slots, the fused kernels (dense AND ragged spans) leave them untouched, and
    fcp_concat_outputs_host copies the host tensors there (one pinned staging buffer, one H2D copy, one
    scatter).  Checked against the oracle's group matrix with the host tensors at their concat offsets."""
    import dataclasses
    from recom_amd import synth
    from recom_amd.ops import ConcatOutputs, FeatureColumnProcess, concat_inputs
    from recom_amd.plan import FORM_EXTERNAL, FORM_PASSTHROUGH, ROWS_FROM_GROUP
    torch = torch_cuda
    for batch in (5, 67, 300):
        m = synth.model_mixed(batch=batch, vocab=997, n_groups=1)
        # every passthrough column becomes an external slot; two more are added at the ends of the row
        cols = [dataclasses.replace(c, form=FORM_EXTERNAL, ids_input=-1, rows_source=ROWS_FROM_GROUP, rows_arg=0)
                if c.form == FORM_PASSTHROUGH else c for c in m.spec.columns]
        top = max(c.concat_slot for c in cols) + 1
        cols = [dataclasses.replace(c, concat_slot=c.concat_slot + 1) for c in cols]
        cols.append(dataclasses.replace(cols[0], form=FORM_EXTERNAL, dim=4, vocab=0, table_input=-1, ids_input=-1,
                                        id_source=0, boundaries=None, rows_source=ROWS_FROM_GROUP, rows_arg=0, concat_slot=0))
        cols.append(dataclasses.replace(cols[-1], dim=36, concat_slot=top + 1))
        spec = dataclasses.replace(m.spec, columns=cols)
        spec.validate()
        assert any(c.form == FORM_EXTERNAL for c in spec.columns[:-2])
        req = m.make_request(1)
        tabs_np = m.numpy_tables()
        out, packed, op = run_gpu(torch, spec, req.inputs, tabs_np, req.symbols)
        want, _ = oracle.process_feature_columns(spec.to_dict(), *packed, tabs_np, req.symbols)
        want = want[0].copy()
        offs = spec.column_offsets()
        order = sorted(range(spec.n_columns), key=lambda k: spec.columns[k].concat_slot)
        rng = np.random.default_rng(batch)
        host, host_pos = [], []
        for pos, k in enumerate(order):
            c = spec.columns[k]
            if c.form == FORM_EXTERNAL:
                a = rng.standard_normal((batch, c.dim)).astype(np.float32)
                host.append(a)
                host_pos.append(pos)
                want[:, offs[k]:offs[k] + c.dim] = a
        out_cols = spec.output_columns()
        dev_pos = [pos for pos, k in enumerate(order) if spec.columns[k].form != FORM_EXTERNAL]
        dev_in = [out_cols.index(order[pos]) for pos in dev_pos]
        co = ConcatOutputs(len(host), [spec.columns[k].dim for k in order], dev_in, dev_pos, host_pos, 0, 1)
        ptrs = out.output_ptrs[out_cols]
        shapes = out.output_shapes.reshape(-1, 2)[out_cols].reshape(-1)
        got = co(ptrs, shapes, host, out.buffer)
        torch.cuda.synchronize()
        assert got.data_ptr() == out.groups[0].data_ptr()          # a view of the arena: no second pass
        assert np.array_equal(got.cpu().numpy(), want)
        # a plan whose layout does not match the op's embedd_dims is refused, not mis-copied
        bad = ConcatOutputs(len(host), [spec.columns[k].dim for k in order][::-1], dev_in, dev_pos, host_pos, 0, 1)
        with pytest.raises(ValueError):
            bad(ptrs, shapes, host, out.buffer)
    # an all-one-hot plan (dense kernel only) with external slots inside and between its spans
    m = synth.model_s2(columns=60, vocab=1000, batch=70)
    cols = [dataclasses.replace(c, concat_slot=2 * c.concat_slot) for c in m.spec.columns]
    ext = dataclasses.replace(cols[1], form=FORM_EXTERNAL, vocab=0, table_input=-1, ids_input=-1, id_source=0,
                              boundaries=None, rows_source=ROWS_FROM_GROUP, rows_arg=0)
    for slot, dim in ((1, 8), (33, 300), (77, 4), (119, 20)):
        cols.append(dataclasses.replace(ext, dim=dim, concat_slot=slot))
    spec = dataclasses.replace(m.spec, columns=cols)
    req = m.make_request(4)
    tabs_np = m.numpy_tables()
    out, packed, op = run_gpu(torch, spec, req.inputs, tabs_np, req.symbols)
    want = oracle.process_feature_columns(spec.to_dict(), *packed, tabs_np, req.symbols)[0][0].copy()
    offs = spec.column_offsets()
    host, host_offs = [], []
    for k, c in enumerate(spec.columns):
        if c.form == FORM_EXTERNAL:
            a = np.full((70, c.dim), float(k), np.float32)
            host.append(a)
            host_offs.append(offs[k])
            want[:, offs[k]:offs[k] + c.dim] = a
    from recom_amd.ops import concat_outputs_host
    concat_outputs_host(host, host_offs, out.groups[0])
    torch.cuda.synchronize()
    assert np.array_equal(out.groups[0].cpu().numpy(), want)


def test_dlrm_scaled_vs_oracle(torch_cuda, oracle):
    from recom_amd import synth
    cards = [min(c, 40000) for c in synth.CRITEO_KAGGLE_CARDINALITIES]
    m = synth.model_dlrm(batch=2048, cardinalities=cards)
    tabs = m.numpy_tables()
    req = m.make_request(0)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert out.groups[0].shape == (2048, 26 * 16 + 13)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


def _closed_form_check(torch, model, req, out):
    """Full-size property: a gather is a pure copy, so every output row must be
    exactly the closed-form table row (synth.hash_rows) — no table on the host."""
    from recom_amd import synth
    import fcp_oracle as O
    got = out.groups[0].cpu().numpy()
    offs = model.spec.column_offsets()
    for k, c in enumerate(model.spec.columns):
        sl = got[:, offs[k]:offs[k] + c.dim]
        raw = req.inputs[c.ids_input]
        if c.form == 4:
            assert np.array_equal(sl, raw.reshape(sl.shape))
            continue
        ids = O.np_bucketize(c.boundaries, raw) if c.id_source == 2 else raw
        t = model.tables[c.table_input]
        assert np.array_equal(sl, synth.hash_rows(t.seed, ids, c.dim)), f"column {k}"


def test_dlrm_full_size_closed_form(torch_cuda):
    """BASELINE.json configs[2]: 26 categorical (Criteo cardinalities) + 13 dense, batch 2048."""
    from recom_amd import synth
    torch = torch_cuda
    m = synth.model_dlrm()
    tabs = m.torch_tables(torch.device("cuda", 0))
    req = m.make_request(1)
    out, _, _ = run_gpu(torch, m.spec, req.inputs, None, req.symbols, tables_dev=tabs)
    _closed_form_check(torch, m, req, out)


def test_s2_full_size_closed_form(torch_cuda):
    """BASELINE.json configs[1] at full size: 1000 columns, dims 8-64, vocab 1M
    (120 GB of tables in HBM), batch 512.  Skipped if the device is too small."""
    from recom_amd import synth
    torch = torch_cuda
    m = synth.model_s2()
    free, _total = torch.cuda.mem_get_info()
    if free < m.table_bytes() + (8 << 30):
        pytest.skip(f"needs {m.table_bytes() / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} GiB free")
    tabs = m.torch_tables(torch.device("cuda", 0))
    op = None
    for seed in (0, 1):
        req = m.make_request(seed)
        out, _, op = run_gpu(torch, m.spec, req.inputs, None, req.symbols, op, tables_dev=tabs)
        assert out.groups[0].shape == (512, 30000)
        _closed_form_check(torch, m, req, out)
    del tabs
    torch.cuda.empty_cache()


def test_output_arena_beyond_4_gib(torch_cuda):
    """64-bit offsets end to end (the reference's generated code indexes with 32-bit ints): S2 at batch 40 000
    writes a 4.8 GB concat matrix; rows that lie wholly beyond byte 2^32 of the arena, the row that straddles it and the
    first rows must all equal the closed-form table rows."""
    from recom_amd import synth
    import fcp_oracle as O
    torch = torch_cuda
    B = 40_000
    m = synth.model_s2(batch=B)
    free, _total = torch.cuda.mem_get_info()
    if free < m.table_bytes() + (16 << 30):
        pytest.skip(f"needs {(m.table_bytes() + (16 << 30)) / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} GiB free")
    tabs = m.torch_tables(torch.device("cuda", 0))
    req = m.make_request(11)
    out, _, _ = run_gpu(torch, m.spec, req.inputs, None, req.symbols, tables_dev=tabs)
    g = out.groups[0]
    assert g.shape == (B, 30000) and g.numel() * 4 > (1 << 32)
    straddle = (1 << 32) // (30000 * 4)                         # the row that contains byte 2^32
    offs = m.spec.column_offsets()
    for lo, hi in ((0, 64), (straddle - 8, straddle + 8), (B - 512, B)):
        got = g[lo:hi].cpu().numpy()
        for k, c in enumerate(m.spec.columns):
            raw = req.inputs[c.ids_input][lo:hi]
            ids = O.np_bucketize(c.boundaries, raw) if c.id_source == 2 else raw
            want = synth.hash_rows(m.tables[c.table_input].seed, ids, c.dim)
            assert np.array_equal(got[:, offs[k]:offs[k] + c.dim], want), (lo, k)
    del tabs, out, g
    torch.cuda.empty_cache()


@pytest.mark.parametrize("how", ["indices", "indices-staged", "csr"])
def test_ragged_full_size_closed_form(torch_cuda, how):
    """BASELINE.json configs[3] at full size: 512 multi-hot columns, vocab 100k (6 GB of
    tables), batch 256, 0..10 ids per row, sum / mean alternating.  Every pooled vector is rebuilt from the
    closed-form table rows with fp32 adds in id order — bit-exact.  `indices`: the config's own encoding, SparseTensor
    indices [nnz, 2] int64 as delivered (segment-offset pre-pass on the device); `indices-staged`: the same request
    through the staged Addons>ConcatInputs (int32 ids, row offsets made on the host: fcp_concat_inputs_ex with the
    modes of the plan's stage section); `csr`: row offsets given."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    from recom_amd.plan import COMBINER_MEAN
    torch = torch_cuda
    m = synth.model_ragged(seg="csr" if how == "csr" else "indices")
    tabs = m.torch_tables(torch.device("cuda", 0))
    op = None
    staged_spec, stage = m.spec.staged_for_concat_inputs() if how == "indices-staged" else (None, None)
    for seed in (0, 5):                                     # different nnz: new descriptors
        req = m.make_request(seed)
        if stage is None:
            out, _, op = run_gpu(torch, m.spec, req.inputs, None, req.symbols, op, tables_dev=tabs)
        else:
            assert stage.symbols_input == m.spec.n_host_inputs and stage.modes.count(2) == 512 and stage.modes.count(1) == 512
            blob, offsets, shapes = concat_inputs(list(req.inputs) + [req.symbols], stage)
            assert blob.nbytes < 0.25 * sum(a.nbytes for a in req.inputs)   # 15.7 MB of tensors -> about 3.1 MB on the wire
            op = op or FeatureColumnProcess(staged_spec, 0)
            out = op(torch.from_numpy(blob).cuda(), offsets, shapes, tabs, req.symbols)
            torch.cuda.synchronize()
        got = out.groups[0].cpu().numpy()
        assert got.shape == (256, 15360)
        offs = m.spec.column_offsets()
        for k, c in enumerate(m.spec.columns):
            ids = req.inputs[c.ids_input]
            if how == "csr":
                csr = req.inputs[c.seg_input].astype(np.int64)
            else:
                csr = np.searchsorted(req.inputs[c.seg_input][:, 0], np.arange(257), side="left")
            rows = synth.hash_rows(m.tables[c.table_input].seed, ids, c.dim)
            want = np.zeros((256, c.dim), np.float32)
            seg = np.repeat(np.arange(256), np.diff(csr))
            np.add.at(want, seg, rows)                       # unbuffered: adds in id order, fp32
            if c.combiner == COMBINER_MEAN:
                cnt = np.diff(csr).astype(np.float32)
                want[cnt > 0] = want[cnt > 0] / cnt[cnt > 0, None]
            assert np.array_equal(got[:, offs[k]:offs[k] + c.dim], want), f"column {k}"
    del tabs
    torch.cuda.empty_cache()


class _RawBlob:
    """A device address handed to the op wrapper in place of a torch tensor (the stager's blob)."""

    def __init__(self, ptr, nbytes):
        self._p, self._n = ptr, nbytes

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def element_size(self):
        return 1


@pytest.mark.parametrize("zero_copy", [False, True])
def test_request_stager_matches_concat_inputs_and_feeds_the_kernel(torch_cuda, oracle, zero_copy):
    """ConcatInputs + H2D as one step.  The staged device blob is
    byte-identical to ConcatInputs' output; offsets / shapes are the same arrays; the
    kernel result through it equals the oracle — with the H2D copy and with
    FCP_STAGER_ZERO_COPY (the kernel reads the pinned ring through its device mapping)."""
    import ctypes as C
    from recom_amd import lib, synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=70, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    st = RequestStager(1 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=3, n_threads=4, zero_copy=zero_copy)
    for seed in range(7):  # more requests than ring slots
        req = m.make_request(seed)
        blob, offsets, shapes = concat_inputs(req.inputs)
        d_ptr, nbytes, off2, shp2 = st.stage(req.inputs)
        assert nbytes == blob.nbytes and np.array_equal(off2, offsets) and np.array_equal(shp2, shapes)
        torch.cuda.synchronize()
        tmp = torch.empty(nbytes, dtype=torch.int8, device="cuda")
        hip = C.CDLL("libamdhip64.so")  # already loaded by torch: device-to-device copy of the staged blob
        assert hip.hipMemcpy(C.c_void_p(tmp.data_ptr()), C.c_void_p(d_ptr), C.c_size_t(nbytes), 3) == 0
        assert np.array_equal(tmp.cpu().numpy(), blob)
        out = op(tmp, off2, shp2, tabs, req.symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
        out2 = op(_RawBlob(d_ptr, nbytes), off2, shp2, tabs, req.symbols)       # the kernel on the staged blob itself
        torch.cuda.synchronize()
        assert np.array_equal(out2.groups[0].cpu().numpy(), want[0])
    st.close()


@pytest.mark.parametrize("copy", ["kernel", "sdma"])
@pytest.mark.parametrize("groups", ["1", "4", "16"])
def test_request_stager_ships_a_request_in_groups_with_either_copy_engine(torch_cuda, oracle, monkeypatch, copy, groups):
    """Round 5: a request is packed in groups of inputs and every group is shipped as soon as it is packed (FCP_STAGER_GROUPS,
    here for every request: FCP_DIAG=stager_groups_always), by a copy KERNEL on the stager's stream (the default: no SDMA engine in
    the path) or by hipMemcpyAsync (FCP_STAGER_COPY_SDMA).  A request large enough for several pack chunks (64 KB each), odd
    tensor sizes (group boundaries at any 4-byte offset): the device blob is byte-identical to ConcatInputs' output, the
    kernel's result through it equals the oracle, the stager counts one copy call per group."""
    import ctypes as C
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    torch = torch_cuda
    monkeypatch.setenv("FCP_STAGER_GROUPS", groups)
    monkeypatch.setenv("FCP_DIAG", "stager_groups_always")
    m = synth.model_mixed(batch=2051, vocab=4999, n_groups=1)          # ~1 MB of host tensors per request
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    st = RequestStager(8 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=3, n_threads=6, copy=copy)
    hip = C.CDLL("libamdhip64.so")
    calls = 0
    for seed in range(7):                                              # more requests than ring slots
        req = m.make_request(seed, B=2051 - 3 * seed)
        blob, offsets, shapes = concat_inputs(req.inputs)
        d_ptr, nbytes, off2, shp2 = st.stage(req.inputs)
        assert nbytes == blob.nbytes and np.array_equal(off2, offsets) and np.array_equal(shp2, shapes)
        out = op(_RawBlob(d_ptr, nbytes), off2, shp2, tabs, req.symbols)       # the kernel on the staged blob itself
        torch.cuda.synchronize()
        tmp = torch.empty(nbytes, dtype=torch.int8, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(tmp.data_ptr()), C.c_void_p(d_ptr), C.c_size_t(nbytes), 3) == 0
        assert np.array_equal(tmp.cpu().numpy(), blob)
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
        stats = st.stats()
        assert stats["calls"] == seed + 1 and 1 <= stats["copy_calls"] - calls <= int(groups)
        if groups != "1" and blob.nbytes > (256 << 10):
            assert stats["copy_calls"] - calls > 1                     # several chunks -> several groups
        calls = stats["copy_calls"]
    assert st.stats()["copy_calls_over_1ms"] == 0 or copy == "sdma"   # (the engine's submission may stall: that is why it is not the default)
    st.close()


@pytest.mark.parametrize("copy", ["kernel", "sdma"])
def test_request_stager_groups_with_one_and_two_byte_tensors(torch_cuda, monkeypatch, copy):
    """ConcatInputs packs tensors of ANY dtype (concat_inputs_ops.cc:42-77): int8 / uint16 tensors of odd lengths put the
    boundaries of the stager's groups at arbitrary byte offsets; the copy kernel moves 16-byte words and widens every group's
    range — the device blob must still be byte-identical to ConcatInputs' output, for every request of a ring that is reused."""
    import ctypes as C
    from recom_amd.ops import RequestStager, concat_inputs
    torch = torch_cuda
    monkeypatch.setenv("FCP_STAGER_GROUPS", "7")
    monkeypatch.setenv("FCP_DIAG", "stager_groups_always")
    rng = np.random.default_rng(5)
    st = RequestStager(16 << 20, 64, 128, depth=2, n_threads=5, copy=copy)
    hip = C.CDLL("libamdhip64.so")
    for req in range(6):
        inputs = []
        for i in range(40 + req):
            dt = [np.int8, np.uint16, np.float32, np.int64, np.uint8][int(rng.integers(5))]
            n = int(rng.integers(1, 40_000)) | 1                         # odd element counts
            inputs.append(rng.integers(0, 100, size=n).astype(dt))
        blob, offsets, shapes = concat_inputs(inputs)
        d_ptr, nbytes, off2, shp2 = st.stage(inputs)
        assert nbytes == blob.nbytes and np.array_equal(off2, offsets) and np.array_equal(shp2, shapes)
        torch.cuda.synchronize()
        tmp = torch.empty(nbytes, dtype=torch.int8, device="cuda")
        assert hip.hipMemcpy(C.c_void_p(tmp.data_ptr()), C.c_void_p(d_ptr), C.c_size_t(nbytes), 3) == 0
        assert np.array_equal(tmp.cpu().numpy(), blob), req
    assert st.stats()["copy_calls"] > 6
    st.close()


@pytest.mark.parametrize("zero_copy", [False, True])
def test_request_stager_turns_sparse_indices_into_row_offsets(torch_cuda, oracle, zero_copy):
    """fcp_stager_stage_ex / FCP_STAGE_SEG_TO_CSR: the sorted row ids of multi-hot features (SparseTensor indices [nnz, 2],
    int32 / int64 row ids) become int32 CSR offsets while the host packs them, ids are narrowed; the staged plan
    (PlanSpec.staged(): those columns read CSR, no pre-pass, no in-block search) gives the same bits as the original
    request through the oracle.  The staged blob is the converted tensors in the staged layout (conftest.assert_staged_blob)."""
    import fcp_oracle as O
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    from recom_amd.plan import STAGE_NARROW_I64, STAGE_SEG_TO_CSR
    torch = torch_cuda
    for m in (synth.model_mixed(batch=70, vocab=997, n_groups=1), synth.model_ragged(columns=40, vocab=3000, batch=130, seg="indices"),
              synth.model_ragged(columns=12, vocab=500, batch=33, seg="rowids32")):
        sspec, modes, rows_col = m.spec.staged()
        assert STAGE_SEG_TO_CSR in modes
        tabs_np = m.numpy_tables()
        tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
        op = FeatureColumnProcess(sspec, 0)
        st = RequestStager(4 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=3, n_threads=4, zero_copy=zero_copy)
        for seed in range(5):
            req = m.make_request(seed)
            rows = [int(req.symbols[m.spec.columns[k].rows_arg]) if k >= 0 else 0 for k in rows_col]
            d_ptr, nbytes, offs, shps = st.stage_ex(req.inputs, modes, rows)
            conv = []
            for i, a in enumerate(req.inputs):                  # what the stager must have packed
                if modes[i] == STAGE_SEG_TO_CSR:
                    conv.append(O.np_segment_offsets(np.asarray(a).reshape(a.shape[0], -1)[:, 0], rows[i]).astype(np.int32))
                elif modes[i] == STAGE_NARROW_I64:
                    conv.append(np.where((a >= 0) & (a <= 0x7fffffff), a, -1).astype(np.int32))
                else:
                    conv.append(a)
            out = op(_RawBlob(d_ptr, nbytes), offs, shps, tabs, req.symbols)
            torch.cuda.synchronize()
            staged = np.empty(nbytes, np.int8)
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(staged.ctypes.data), C.c_void_p(d_ptr), C.c_size_t(nbytes), 2) == 0
            from conftest import assert_staged_blob
            assert assert_staged_blob(staged, offs, shps, conv, modes) == nbytes   # (row offsets: one matrix behind the other inputs)
            want, _ = oracle.process_feature_columns(m.spec.to_dict(), *concat_inputs(req.inputs), tabs_np, req.symbols)
            for g, w in enumerate(want):
                assert np.array_equal(out.groups[g].cpu().numpy(), w), (m.name, seed, g)
        st.close()


def test_request_stager_narrows_int64_ids(torch_cuda, oracle):
    """fcp_stager_stage_narrow: int64 ids / SparseTensor indices cross PCIe as int32; the
    narrowed plan gives the same bits as the int64 request through the oracle —
    including ids that do not fit (they are invalid in both: zeros)."""
    import ctypes as C
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=70, vocab=997, n_groups=1)
    nspec, flags = m.spec.narrowed()
    assert any(flags) and nspec.validate() is None
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(nspec, 0)
    st = RequestStager(1 << 20, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=2, n_threads=3)
    hip = C.CDLL("libamdhip64.so")
    for seed in range(5):
        req = m.make_request(seed)
        ids0 = req.inputs[m.spec.columns[0].ids_input]
        ids0[:3] = [(1 << 33) + 5, -3, 2 ** 31]          # none is a valid id; none fits int32 as itself
        blob, offsets, shapes = concat_inputs(req.inputs)
        d_ptr, nbytes, off2, shp2 = st.stage(req.inputs, narrow=flags)
        assert nbytes == blob.nbytes - 4 * sum(req.inputs[i].size for i, f in enumerate(flags) if f)
        assert np.array_equal(shp2, shapes)
        tmp = torch.empty(nbytes, dtype=torch.int8, device="cuda")
        torch.cuda.synchronize()
        assert hip.hipMemcpy(C.c_void_p(tmp.data_ptr()), C.c_void_p(d_ptr), C.c_size_t(nbytes), 3) == 0
        out = op(tmp, off2, shp2, tabs, req.symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob, offsets, shapes, tabs_np, req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    st.close()


def test_process_call_is_hip_graph_capturable(torch_cuda, oracle):
    """With its descriptors cached, fcp_process_feature_columns only enqueues kernels (here the
    segment-offset pre-pass + the ragged kernel), so a serving loop can capture it into a HIP
    graph; a replay reads the current contents of the blob."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=64, vocab=997, n_groups=1)
    tabs_np = m.numpy_tables()
    tabs = [torch.from_numpy(t).cuda() for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    req = m.make_request(1)
    blob, offsets, shapes = concat_inputs(req.inputs)
    d_blob = torch.from_numpy(blob).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        op(d_blob, offsets, shapes, tabs, req.symbols)        # descriptors become resident
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = op(d_blob, offsets, shapes, tabs, req.symbols)
    ids0 = req.inputs[m.spec.columns[0].ids_input]
    for bump in (1, 7):                                        # same shapes, new ids
        ids0[:] = (ids0 + bump) % 997
        blob2, o2, s2 = concat_inputs(req.inputs)
        assert np.array_equal(o2, offsets) and np.array_equal(s2, shapes)
        d_blob.copy_(torch.from_numpy(blob2))
        g.replay()
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(m.spec.to_dict(), blob2, offsets, shapes, tabs_np, req.symbols)
        assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    # The captured launch reads its descriptor slot at every replay: 36 requests with other shapes (more than
    # the 32 slots) on the same stream must not evict it ...
    import ctypes as C
    from recom_amd import lib
    others = [m.make_request(100 + k, B=40 + k) for k in range(36)]
    with torch.cuda.stream(s):
        for r in others:
            b, o, sh = concat_inputs(r.inputs)
            res = op(torch.from_numpy(b).cuda(), o, sh, tabs, r.symbols)
            torch.cuda.synchronize()
            w, _ = oracle.process_feature_columns(m.spec.to_dict(), b, o, sh, tabs_np, r.symbols)
            assert np.array_equal(res.groups[0].cpu().numpy(), w[0])
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    # ... and a request whose shapes are NOT resident cannot be captured (its descriptors would have to be
    # installed inside the capture): refused, loudly
    fresh = m.make_request(999, B=77)
    fb, fo, fs = concat_inputs(fresh.inputs)
    d_fresh = torch.from_numpy(fb).cuda()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(lib.FcpError) as e:
        with torch.cuda.graph(g2, stream=s):
            op(d_fresh, fo, fs, tabs, fresh.symbols)
    assert e.value.status == lib.FCP_ERR_UNSUPPORTED and "capture" in str(e.value)
    torch.cuda.synchronize()
    # ... nor one whose tables have MOVED (re-binding copies records and synchronises the device: ADVICE r02); the plan
    # keeps its old binding, so the next ordinary request with the old tables needs no re-bind
    moved = [t.clone() for t in tabs]
    g3 = torch.cuda.CUDAGraph()
    with pytest.raises(lib.FcpError) as e:
        with torch.cuda.graph(g3, stream=s):
            op(d_blob, offsets, shapes, moved, req.symbols)
    assert e.value.status == lib.FCP_ERR_UNSUPPORTED and "tables" in str(e.value)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    lib.check(lib.load().fcp_plan_release_captures(op.plan.handle), "fcp_plan_release_captures")
    with torch.cuda.stream(s):                                 # the plan serves on after the release
        res = op(d_fresh, fo, fs, tabs, fresh.symbols)
    torch.cuda.synchronize()
    w, _ = oracle.process_feature_columns(m.spec.to_dict(), fb, fo, fs, tabs_np, fresh.symbols)
    assert np.array_equal(res.groups[0].cpu().numpy(), w[0])


@pytest.mark.parametrize("seg", ["indices", "rowids32"])
def test_scatter_rows_arrive_in_any_order(torch_cuda, oracle, seg):
    """Form 3 = ScatterNd(rows, GatherV2(table, ids)): the reference scatters whatever order the row ids come in
    (GatherScatterRows, cuda_emitter.cc:296-345), and so does the HIP path (inverse map built by the pre-pass, last
    write wins).  Shuffled rows, rows hit twice, rows outside [0, B): the RESULT equals the oracle's sequential scatter."""
    import dataclasses
    from recom_amd import synth
    from recom_amd.plan import (COMBINER_NONE, FLAG_COUNT_BAD_IDS, FORM_GATHER_SCATTER, IDS_I64, ROWS_FROM_SYMBOL,
                                SEG_IDS_I32, SEG_IDS_I64, ColumnSpec, PlanSpec)
    rng = np.random.default_rng(11)
    B, vocab = 300, 991
    cols, ranks, esz, tables, inputs = [], [], [], [], []
    for c, dim in enumerate((8, 16, 4, 64, 32)):
        n = [B, B // 2, 0, 2 * B, 1][c]                            # all rows / half of them / none / every row hit about twice / one
        rows = rng.integers(0, B, n).astype(np.int64) if n > B else rng.permutation(B)[:n].astype(np.int64)
        if c == 1:
            rows[:5] = [-1, B, B + 7, -(2 ** 40), 2 ** 40]          # ScatterNd on a GPU drops rows outside the output
        ids = rng.integers(0, vocab, n).astype(np.int64)
        if c == 0:
            ids[:3] = [-1, vocab, 2 ** 35]                         # out-of-vocabulary ids scatter zeros
        tables.append(synth.hash_table_numpy(50 + c, vocab, dim))
        inputs.append(ids)
        if seg == "indices":
            inputs.append(np.stack([rows, np.zeros(n, np.int64)], axis=1).reshape(n, 2))
            kind, stride, r, e = SEG_IDS_I64, 2, 2, 8
        else:
            inputs.append(np.where(np.abs(rows) < 2 ** 31, rows, -1).astype(np.int32))
            kind, stride, r, e = SEG_IDS_I32, 1, 1, 4
        ranks += [1, r]
        esz += [8, e]
        cols.append(ColumnSpec(FORM_GATHER_SCATTER, dim, vocab, COMBINER_NONE, IDS_I64, c, 2 * c, 2 * c + 1, kind, stride,
                               ROWS_FROM_SYMBOL, 0, None, 0, c))
    spec = PlanSpec(cols, ranks, esz, len(tables), n_groups=1, n_symbols=1, flags=FLAG_COUNT_BAD_IDS)
    sym = np.array([B], np.int32)
    out, packed, op = run_gpu(torch_cuda, spec, inputs, tables, sym)
    assert_equal_oracle(oracle, spec, packed, tables, sym, out)
    assert op.plan.read_bad_ids() >= 3 + 5                         # the bad ids and the dropped rows are counted


def test_scatter_rows_with_several_ids_and_a_filter(torch_cuda, oracle):
    """A ScatterNd column behind an id filter (plan_builder: ScatterNd(GatherIndiceValue:0, GatherV2(table,
    GatherIndiceValue:1))): of a row's ids the last one the filter KEEPS wins — the oracle compacts first, TF scatters the
    survivors — in all three row encodings (sorted CSR: walked backwards; row ids: filtered before the inverse map)."""
    from recom_amd import synth
    from recom_amd.plan import (COMBINER_NONE, FORM_GATHER_SCATTER, IDS_I64, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_IDS_I32,
                                SEG_IDS_I64, XFORM_FILTER, ColumnSpec, PlanSpec)
    rng = np.random.default_rng(12)
    B, vocab = 70, 500
    cols, ranks, esz, tables, inputs = [], [], [], [], []
    for c, (seg, dim) in enumerate((("csr", 8), ("indices", 16), ("rowids32", 32), ("csr", 4))):
        lens = rng.integers(0, 5, B)                               # up to 4 ids per row
        if c == 3:
            lens[7] = 450                                          # one row longer than the wave's tile
        nnz = int(lens.sum())
        rows = np.repeat(np.arange(B, dtype=np.int64), lens)
        ids = rng.integers(0, vocab, nnz).astype(np.int64)
        tables.append(synth.hash_table_numpy(70 + c, vocab, dim))
        inputs.append(ids)
        if seg == "csr":
            inputs.append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32))
            kind, stride, r, e = SEG_CSR_I32, 1, 1, 4
        elif seg == "indices":
            inputs.append(np.stack([rows, np.zeros(nnz, np.int64)], axis=1).reshape(nnz, 2))
            kind, stride, r, e = SEG_IDS_I64, 2, 2, 8
        else:
            perm = rng.permutation(nnz)                            # any order: the LAST kept id of a row in this order wins
            inputs[-1] = ids[perm]
            inputs.append(rows[perm].astype(np.int32))
            kind, stride, r, e = SEG_IDS_I32, 1, 1, 4
        ranks += [1, r]
        esz += [8, e]
        cols.append(ColumnSpec(FORM_GATHER_SCATTER, dim, vocab, COMBINER_NONE, IDS_I64, c, 2 * c, 2 * c + 1, kind, stride,
                               ROWS_FROM_SYMBOL, 0, None, 0, c, xform_mode=XFORM_FILTER,
                               xform_lo=[0, 300], xform_hi=[99, 420]))
    spec = PlanSpec(cols, ranks, esz, len(tables), n_groups=1, n_symbols=1)
    sym = np.array([B], np.int32)
    out, packed, _ = run_gpu(torch_cuda, spec, inputs, tables, sym)
    assert_equal_oracle(oracle, spec, packed, tables, sym, out)


def test_empty_batch(torch_cuda, oracle):
    """A request with zero rows: nothing is launched, shapes are still reported."""
    from recom_amd import synth
    m = synth.model_mixed(batch=0, vocab=97, n_groups=1)
    tabs = m.numpy_tables()
    req = m.make_request(0)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert out.groups[0].shape == (0, m.spec.group_width(0))
    assert all(int(x) == 0 for x in out.output_shapes[0::2])
    d = synth.model_s2(columns=5, vocab=50, batch=0)
    req = d.make_request(0)
    out, _, _ = run_gpu(torch_cuda, d.spec, req.inputs, d.numpy_tables(), req.symbols)
    assert out.groups[0].shape[0] == 0


def test_wide_columns_and_large_boundary_lists(torch_cuda, oracle):
    """A 1024-wide column spans four 1-KiB spans; 3000 boundaries exceed the 1024-float
    LDS staging area (the search then runs from global memory); dim-4 columns put 64
    columns into one span."""
    from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, FORM_GATHER, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE,
                                IDS_I64, ROWS_FROM_IDS, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_NONE, ColumnSpec, PlanSpec)
    rng = np.random.default_rng(5)
    B = 37
    bnd = np.sort(rng.uniform(-100, 100, 3000)).astype(np.float32)
    cols, ranks, esz, tables, inputs = [], [], [], [], []

    def add_input(a):
        inputs.append(a)
        ranks.append(a.ndim)
        esz.append(a.dtype.itemsize)
        return len(inputs) - 1

    slot = 0
    tables.append(rng.standard_normal((50, 1024)).astype(np.float32))  # wide gather column
    cols.append(ColumnSpec(FORM_GATHER, 1024, 50, COMBINER_NONE, IDS_I64, 0, add_input(rng.integers(0, 50, B)), -1,
                           SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0, slot))
    slot += 1
    tables.append(rng.standard_normal((3001, 8)).astype(np.float32))   # big boundary list
    cols.append(ColumnSpec(FORM_GATHER, 8, 3001, COMBINER_NONE, IDS_F32_BUCKETIZE, 1,
                           add_input(rng.uniform(-110, 110, B).astype(np.float32)), -1, SEG_NONE, 1, ROWS_FROM_IDS, 0,
                           bnd, 0, slot))
    slot += 1
    for k in range(70):                                               # 70 dim-4 columns: >64 per span
        tables.append(rng.standard_normal((20, 4)).astype(np.float32))
        cols.append(ColumnSpec(FORM_GATHER, 4, 20, COMBINER_NONE, IDS_I64, 2 + k, add_input(rng.integers(0, 20, B)), -1,
                               SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0, slot))
        slot += 1
    dense = PlanSpec(cols, ranks, esz, len(tables))
    out, packed, _ = run_gpu(torch_cuda, dense, inputs, tables, None)
    assert_equal_oracle(oracle, dense, packed, tables, None, out)
    # the same columns behind the ragged kernel (one pooled column makes the plan non-dense)
    lens = rng.integers(0, 5, B)
    ids = rng.integers(0, 50, int(lens.sum()))
    csr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cols2 = list(cols) + [ColumnSpec(FORM_SEGMENT_REDUCE, 1024, 50, COMBINER_MEAN, IDS_I64, 0, len(inputs),
                                     len(inputs) + 1, SEG_CSR_I32, 1, ROWS_FROM_SYMBOL, 0, None, 0, slot)]
    mixed = PlanSpec(cols2, ranks + [1, 1], esz + [8, 4], len(tables), n_symbols=1)
    sym = np.asarray([B], np.int32)
    out, packed, _ = run_gpu(torch_cuda, mixed, inputs + [ids, csr], tables, sym)
    assert_equal_oracle(oracle, mixed, packed, tables, sym, out)


def test_pooled_vectors_vs_the_reference_kernels_own_orders(torch_cuda, oracle):
    """North star: pooled vectors within 1e-5 of the reference.  The HIP path adds in id order (TF-CPU's order for
    bags of up to 9 ids, see the next test); the reference's GPU kernels add in block-scan order for dim <= 20 (cuda_emitter.cc:348-661) and in 8
    strided partials + an LDS tree for dim > 20 (:820-962).  Both orders are restated in the oracle
    (orc_sparse_segment_reduce_refscan / _ref8x8): the HIP result must sit within 1e-5 of either."""
    from recom_amd import synth
    torch = torch_cuda
    m = synth.model_ragged(columns=48, vocab=3000, batch=120, dims=(4, 8, 12, 16, 20, 32, 64), max_len=10)
    tabs_np = m.numpy_tables()
    req = m.make_request(9)
    out, packed, _ = run_gpu(torch, m.spec, req.inputs, tabs_np, req.symbols)
    got = out.groups[0].cpu().numpy()
    offs = m.spec.column_offsets()
    small = large = 0
    for k, c in enumerate(m.spec.columns):
        ids, csr = req.inputs[c.ids_input], req.inputs[c.seg_input]
        mean = c.combiner == 2
        if c.dim <= 20:
            rows = np.repeat(np.arange(120), np.diff(csr))
            ref = oracle.sparse_segment_reduce_refscan(tabs_np[c.table_input], ids, rows, 120, mean)
            small += 1
        else:
            ref, _ = oracle.sparse_segment_reduce(tabs_np[c.table_input], ids, csr, mean, ref_order=True)
            empty = np.diff(csr) == 0
            ref[empty] = 0.0                   # the reference's dim > 20 template divides 0 / 0 on empty rows; TF gives zeros
            large += 1
        assert np.abs(got[:, offs[k]:offs[k] + c.dim] - ref).max() < 1e-5, f"column {k} (dim {c.dim})"
    assert small and large


def test_pooled_vectors_vs_tensorflow_cpus_addition_order(torch_cuda, oracle):
    """North star: pooled vectors within 1e-5 of TF-CPU.  TF 2.6.2's CPU kernel adds the first num & 7 rows of a bag
    left to right and every further 8 rows among themselves first (orc_sparse_segment_reduce_tfcpu: third-party
    arithmetic restated from its published source, unpinned).  RAGGED-shaped bags (BASELINE configs[3]: U{0..10} ids per
    row, every dim, sum and mean) with 9-, 10- and 17-id bags forced in: the HIP result is BIT-IDENTICAL to that order
    for bags of up to 9 ids (the same additions) and within 1e-5 beyond."""
    from recom_amd import synth
    torch = torch_cuda
    B = 160
    m = synth.model_ragged(columns=28, vocab=5000, batch=B, dims=(4, 8, 12, 16, 20, 32, 64), max_len=10)
    tabs_np = m.numpy_tables()
    req = m.make_request(21)
    rng = np.random.default_rng(5)
    for c in m.spec.columns:                                     # bags of exactly 9, 10, 17 (and 25) ids in every column
        csr = np.asarray(req.inputs[c.seg_input]).astype(np.int64)
        lens = np.diff(csr)
        lens[[3, 40, 77, 120]] = [9, 10, 17, 25]
        req.inputs[c.seg_input] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        req.inputs[c.ids_input] = rng.integers(0, c.vocab, int(lens.sum())).astype(np.asarray(req.inputs[c.ids_input]).dtype)
    out, packed, _ = run_gpu(torch, m.spec, req.inputs, tabs_np, req.symbols)
    got = out.groups[0].cpu().numpy()
    offs = m.spec.column_offsets()
    seen_long = 0
    for k, c in enumerate(m.spec.columns):
        ids, csr = req.inputs[c.ids_input], req.inputs[c.seg_input]
        tf = oracle.sparse_segment_reduce_tfcpu(tabs_np[c.table_input], ids, csr, c.combiner == 2)
        mine = got[:, offs[k]:offs[k] + c.dim]
        lens = np.diff(csr)
        assert np.array_equal(mine[lens <= 9], tf[lens <= 9]), f"column {k}"
        assert np.abs(mine - tf).max() < 1e-5, f"column {k} (dim {c.dim})"
        seen_long += int((lens >= 10).sum())
    assert seen_long >= 3 * len(m.spec.columns)


@pytest.mark.parametrize("batch,seed", [(33, 0), (1, 1), (257, 2)])
def test_id_transforms_on_device(torch_cuda, oracle, batch, seed):
    """SelectValue / GatherIndiceValue / GatherValueGenIndice fused into the id stage of
    the kernels (dense one-hot, bucketized, pooled sum / mean over CSR / SparseTensor indices / row ids,
    long bags walked from global memory, scatter).  Integer work: bit-exact with the oracle, which is
    itself pinned to "CPU op, then lookup" in tests/test_oracle.py."""
    from test_oracle import XFORMS, _xform_spec
    from recom_amd import synth
    from recom_amd.plan import FLAG_COUNT_BAD_IDS
    import dataclasses
    torch = torch_cuda
    m = synth.model_mixed(batch=batch, vocab=997, n_groups=1)
    spec = dataclasses.replace(_xform_spec(m, XFORMS), flags=FLAG_COUNT_BAD_IDS)
    tabs = m.numpy_tables()
    req = m.make_request(seed)
    out, packed, op = run_gpu(torch, spec, req.inputs, tabs, req.symbols)
    _, bad = assert_equal_oracle(oracle, spec, packed, tabs, req.symbols, out)
    assert op.plan.read_bad_ids() == bad
    # an all-one-hot plan (dense kernel) with transforms, several intervals (the extras live in the const buffer)
    d = synth.model_s2(columns=24, vocab=2000, batch=max(batch, 5))
    # (substitute 5000 is outside the vocabulary: those ids read as zeros and are counted ONCE each, although
    # the columns straddle the 1-KiB spans of the 720-float row and are staged by two blocks)
    tr = {k: ((2, [(0, 400), (900, 1100), (1990, 1999)], 0) if k % 3 == 0 else (1, [(50, 60)], 5000) if k % 3 == 1 else (0, [], 0))
          for k in range(24)}
    dspec = dataclasses.replace(_xform_spec(d, {k: v for k, v in tr.items() if v[0]}), flags=FLAG_COUNT_BAD_IDS)
    req = d.make_request(seed)
    dt = d.numpy_tables()
    out, packed, dop = run_gpu(torch, dspec, req.inputs, dt, req.symbols)
    _, dbad = assert_equal_oracle(oracle, dspec, packed, dt, req.symbols, out)
    assert dbad > 0 and dop.plan.read_bad_ids() == dbad


def test_bucketize_tiers_against_the_reference_itself(torch_cuda, ref_bucketize):
    """The HIP kernels' three ways to find a bucket (computed boundaries, guess + verify, binary search; dense and ragged
    bodies) against the REFERENCE's own `Bucketize`, compiled from its source (oracle/ref_extract.py, cuda_emitter.cc:233-247)
    — no restatement in between.  Each column reads an identity table (row r = [r, r, r, r]), so the output IS the bucket."""
    from recom_amd.plan import (COMBINER_NONE, COMBINER_SUM, FORM_GATHER, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, ROWS_FROM_IDS,
                                ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_NONE, ColumnSpec, PlanSpec)
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle import _bucketize_cases
    rng = np.random.default_rng(5)
    cases = list(_bucketize_cases(rng))
    n = max(len(x) for _, _, x in cases)
    cols, ranks, esz, tables, inputs = [], [], [], [], []
    for slot, (name, b, x) in enumerate(cases):
        inputs.append(np.concatenate([x, np.full(n - len(x), x[0], np.float32)]))
        ranks.append(1)
        esz.append(4)
        tables.append(np.repeat(np.arange(len(b) + 1, dtype=np.float32)[:, None], 4, axis=1))
        cols.append(ColumnSpec(FORM_GATHER, 4, len(b) + 1, COMBINER_NONE, IDS_F32_BUCKETIZE, slot, slot, -1, SEG_NONE, 1,
                               ROWS_FROM_IDS, 0, b, 0, slot))
    out, _, _ = run_gpu(torch_cuda, PlanSpec(cols, ranks, esz, len(tables)), inputs, tables, None)
    got = out.groups[0].cpu().numpy()
    for k, (name, b, x) in enumerate(cases):
        assert np.array_equal(got[:, 4 * k].astype(np.int32), ref_bucketize(b, inputs[k])), ("dense", name)
    # the same columns pooled one value per row (sum of one row = the row), which routes them through the ragged body
    offs = np.arange(n + 1, dtype=np.int32)
    rcols, rranks, resz, rinputs = [], [], [], []
    for slot, (name, b, x) in enumerate(cases):
        rinputs += [inputs[slot], offs]
        rranks += [1, 1]
        resz += [4, 4]
        rcols.append(ColumnSpec(FORM_SEGMENT_REDUCE, 4, len(b) + 1, COMBINER_SUM, IDS_F32_BUCKETIZE, slot, 2 * slot, 2 * slot + 1,
                                SEG_CSR_I32, 1, ROWS_FROM_SYMBOL, 0, b, 0, slot))
    out, _, _ = run_gpu(torch_cuda, PlanSpec(rcols, rranks, resz, len(tables), n_symbols=1), rinputs, tables, np.asarray([n], np.int32))
    got = out.groups[0].cpu().numpy()
    for k, (name, b, x) in enumerate(cases):
        assert np.array_equal(got[:, 4 * k].astype(np.int32), ref_bucketize(b, inputs[k])), ("ragged", name)


def test_bucketize_tiers_are_exact(torch_cuda, oracle):
    """Bucketize (cuda_emitter.cc:233-247, integer result: bit-exact) through the three ways the kernels
    find a bucket: boundaries reproducible as fma(i, step, b0) are never read (bucketize_arith); evenly
    spaced but not reproducible ones are guessed and verified (bucketize_fast); anything else is searched.
    Values: every boundary, its float neighbours, the midpoints, NaN, +-inf, +-0 — in the dense kernel and
    (same columns next to a pooled one) in the ragged kernel."""
    from recom_amd.plan import (COMBINER_NONE, COMBINER_SUM, FORM_GATHER, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I64,
                                ROWS_FROM_IDS, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_NONE, ColumnSpec, PlanSpec)
    rng = np.random.default_rng(11)
    arrays = {
        "reference 0,5,...,495": np.arange(0, 500, 5, dtype=np.float32),
        "dyadic grid": (np.arange(37, dtype=np.float32) * np.float32(0.375) - np.float32(3.0)),
        "two boundaries": np.asarray([-1.0, 2.0], np.float32),
        "tenths (evenly spaced, not reproducible)": (np.arange(200) * 0.1).astype(np.float32),
        "thirds": (np.arange(1, 90) / 3.0).astype(np.float32),
        "log spaced": np.logspace(-3, 4, 150).astype(np.float32),
        "random": np.unique(rng.uniform(-50, 50, 300).astype(np.float32)),
        "single": np.asarray([7.5], np.float32),
        "huge step": np.asarray([-3e38, 0.0, 3e38], np.float32),
    }
    cols, ranks, esz, tables, inputs = [], [], [], [], []
    for slot, (name, b) in enumerate(arrays.items()):
        assert np.all(np.diff(b) > 0), name
        x = np.concatenate([b, np.nextafter(b, -np.inf), np.nextafter(b, np.inf), (b[:-1] + b[1:]) / 2,
                            [np.nan, np.inf, -np.inf, 0.0, -0.0, b[0] - 1, b[-1] + 1, 3.4e38, -3.4e38],
                            rng.uniform(b[0] - 3, b[-1] + 3, 200)]).astype(np.float32)
        inputs.append(x)
        ranks.append(1)
        esz.append(4)
        tables.append(rng.standard_normal((len(b) + 1, 8)).astype(np.float32))
        cols.append(ColumnSpec(FORM_GATHER, 8, len(b) + 1, COMBINER_NONE, IDS_F32_BUCKETIZE, slot, slot, -1, SEG_NONE, 1,
                               ROWS_FROM_IDS, 0, b, 0, slot))
    n = max(len(x) for x in inputs)
    inputs = [np.concatenate([x, np.full(n - len(x), x[0], np.float32)]) for x in inputs]
    dense = PlanSpec(cols, ranks, esz, len(tables))
    out, packed, _ = run_gpu(torch_cuda, dense, inputs, tables, None)
    assert_equal_oracle(oracle, dense, packed, tables, None, out)
    # every column: the ids the oracle's plain binary search finds (a wrong bucket would read another row)
    import fcp_oracle as O
    got = out.groups[0].cpu().numpy()
    for k, (name, b) in enumerate(arrays.items()):
        assert np.array_equal(got[:, 8 * k:8 * k + 8], tables[k][O.np_bucketize(b, inputs[k])]), name
    # behind the ragged kernel (a pooled column makes every span ragged); ids of the pooled column use the
    # evenly spaced boundaries too
    lens = rng.integers(0, 4, n)
    vals = rng.uniform(-5, 505, int(lens.sum())).astype(np.float32)
    csr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tables.append(rng.standard_normal((101, 16)).astype(np.float32))
    pooled = ColumnSpec(FORM_SEGMENT_REDUCE, 16, 101, COMBINER_SUM, IDS_F32_BUCKETIZE, len(tables) - 1, len(inputs),
                        len(inputs) + 1, SEG_CSR_I32, 1, ROWS_FROM_SYMBOL, 0, arrays["reference 0,5,...,495"], 0, len(cols))
    mixed = PlanSpec(cols + [pooled], ranks + [1, 1], esz + [4, 4], len(tables), n_symbols=1)
    sym = np.asarray([n], np.int32)
    out, packed, _ = run_gpu(torch_cuda, mixed, inputs + [vals, csr], tables, sym)
    assert_equal_oracle(oracle, mixed, packed, tables, sym, out)


@pytest.mark.parametrize("columns,batch,vocab", [(40, 100_000, 5000), (5000, 8, 300), (1, 1, 7), (3, 70_001, 50)])
def test_extreme_shapes_dense(torch_cuda, oracle, columns, batch, vocab):
    """Very tall, very wide and degenerate one-hot requests (grid / span / slot-map limits)."""
    from recom_amd import synth
    m = synth.model_s2(columns=columns, vocab=vocab, batch=batch)
    tabs = m.numpy_tables()
    req = m.make_request(2)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.parametrize("seg", ["csr", "indices"])
def test_extreme_shapes_ragged(torch_cuda, oracle, seg):
    """One row holding 60 000 ids among empty rows (bag walked from global memory; the segment
    search / pre-pass see one huge run), and 20 000 rows of short bags."""
    from recom_amd import synth
    from recom_amd.synth import Request
    m = synth.model_ragged(columns=6, vocab=4000, batch=9, seg=seg)
    req = m.make_request(0)
    rng = np.random.default_rng(3)
    inputs = []
    for c in m.spec.columns:                                  # rebuild every column: row 4 gets 60 000 ids
        lens = np.zeros(9, np.int64)
        lens[4], lens[7] = 60_000, 3
        ids = rng.integers(0, 4000, size=int(lens.sum())).astype(np.int64)
        rows = np.repeat(np.arange(9, dtype=np.int64), lens)
        if seg == "csr":
            inputs += [ids, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)]
        else:
            inputs += [ids, np.stack([rows, np.zeros_like(rows)], 1).astype(np.int64)]
    tabs = m.numpy_tables()
    out, packed, _ = run_gpu(torch_cuda, m.spec, inputs, tabs, req.symbols)
    want, _ = oracle.process_feature_columns(m.spec.to_dict(), *packed, tabs, req.symbols)
    got = out.groups[0].cpu().numpy()
    assert np.array_equal(got, want[0])                       # same fp32 add order even over 60 000 terms
    tall = synth.model_ragged(columns=4, vocab=4000, batch=20_000, seg=seg, max_len=3)
    r2 = tall.make_request(1)
    out, packed, _ = run_gpu(torch_cuda, tall.spec, r2.inputs, tall.numpy_tables(), r2.symbols)
    assert_equal_oracle(oracle, tall.spec, packed, tall.numpy_tables(), r2.symbols, out)


def test_plan_and_stager_lifecycle_does_not_leak(torch_cuda):
    """200 create / run / destroy cycles of a plan and a stager leave the device memory where it was."""
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    torch = torch_cuda
    m = synth.model_mixed(batch=16, vocab=97, n_groups=1)
    tabs = [torch.from_numpy(t).cuda() for t in m.numpy_tables()]
    req = m.make_request(0)
    blob, offsets, shapes = concat_inputs(req.inputs)
    d_blob = torch.from_numpy(blob).cuda()

    def cycle():
        op = FeatureColumnProcess(m.spec, 0)
        op(d_blob, offsets, shapes, tabs, req.symbols)
        st = RequestStager(1 << 16, m.spec.n_host_inputs, sum(m.spec.host_input_ranks), depth=2, n_threads=2)
        st.stage(req.inputs)
        torch.cuda.synchronize()
        st.close()
        op.plan.close()

    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(200):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), f"device memory shrank by {(free0 - free1) >> 20} MiB"


@pytest.mark.parametrize("seg", ["csr", "indices", "rowids32"])
@pytest.mark.parametrize("prepass", [False, True])
def test_all_bags_empty_and_all_ids_in_one_row(torch_cuda, oracle, monkeypatch, seg, prepass):
    """nnz = 0 for every column (zero-length id / segment tensors), then every id in the last row,
    then every id in row 0 — through the in-block segment search and through the pre-pass."""
    from recom_amd import synth
    if prepass:
        monkeypatch.setenv("FCP_SEG_PREPASS", "1")
    B = 37
    m = synth.model_ragged(columns=10, vocab=500, batch=B, seg=seg, max_len=0)
    tabs = m.numpy_tables()
    req = m.make_request(0)
    assert all(a.size == 0 for a in req.inputs[0::2])
    out, packed, op = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)
    assert not out.groups[0].cpu().numpy().any()
    rng = np.random.default_rng(1)
    for row in (B - 1, 0):
        inputs = []
        for _ in m.spec.columns:
            ids = rng.integers(0, 500, size=23).astype(np.int64)
            lens = np.zeros(B, np.int64)
            lens[row] = 23
            if seg == "csr":
                inputs += [ids, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)]
            elif seg == "indices":
                inputs += [ids, np.stack([np.full(23, row, np.int64), np.arange(23, dtype=np.int64)], 1)]
            else:
                inputs += [ids, np.full(23, row, np.int32)]
        out, packed, op = run_gpu(torch_cuda, m.spec, inputs, tabs, req.symbols, op)
        assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.parametrize("prepass", [False, True])
def test_out_of_range_segment_ids_and_nonfinite_bucketize_values(torch_cuda, oracle, monkeypatch, prepass):
    """Sorted segment ids that start below 0 and end beyond the row count (those ids belong to no
    row), and NaN / +-inf / -0.0 fed to Bucketize: the HIP path and the oracle agree bit for bit
    (upper_bound semantics: NaN and +inf land in the last bucket)."""
    from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_GATHER, FORM_SEGMENT_REDUCE,
                                IDS_F32_BUCKETIZE, IDS_I64, ROWS_FROM_IDS, ROWS_FROM_SYMBOL, SEG_IDS_I32, SEG_IDS_I64,
                                SEG_NONE, ColumnSpec, PlanSpec)
    if prepass:
        monkeypatch.setenv("FCP_SEG_PREPASS", "1")
    rng = np.random.default_rng(11)
    B, vocab = 12, 60
    bnd = np.asarray([-3.0, 0.0, 1.5, 7.0, 100.0], np.float32)
    cols = [
        ColumnSpec(FORM_SEGMENT_REDUCE, 8, vocab, COMBINER_SUM, IDS_I64, 0, 0, 1, SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, 0, None, 0, 0),
        ColumnSpec(FORM_SEGMENT_REDUCE, 4, vocab, COMBINER_MEAN, IDS_I64, 1, 2, 3, SEG_IDS_I32, 1, ROWS_FROM_SYMBOL, 0, None, 0, 1),
        ColumnSpec(FORM_GATHER, 12, len(bnd) + 1, COMBINER_NONE, IDS_F32_BUCKETIZE, 2, 4, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, bnd, 0, 2),
    ]
    spec = PlanSpec(cols, [1, 2, 1, 1, 1], [8, 8, 8, 4, 4], 3, n_groups=1, n_symbols=1)
    seg = np.sort(np.concatenate([[-5, -1, -1], rng.integers(0, B, 40), [B, B + 3, 10 ** 6]])).astype(np.int64)
    idx = np.stack([seg, np.zeros_like(seg)], 1)
    vals = np.float32([np.nan, np.inf, -np.inf, -0.0, 0.0, -3.0, 100.0, 99.99, 1e30, -1e30, 1.5, 7.0])
    inputs = [rng.integers(0, vocab, seg.size).astype(np.int64), idx,
              rng.integers(0, vocab, seg.size).astype(np.int64), seg.astype(np.int32), vals]
    tabs = [rng.standard_normal((vocab, 8)).astype(np.float32), rng.standard_normal((vocab, 4)).astype(np.float32),
            rng.standard_normal((len(bnd) + 1, 12)).astype(np.float32)]
    out, packed, _ = run_gpu(torch_cuda, spec, inputs, tabs, np.asarray([B], np.int32))
    want, _ = assert_equal_oracle(oracle, spec, packed, tabs, np.asarray([B], np.int32), out)
    got = out.groups[0].cpu().numpy()
    assert np.array_equal(got[0, 12:], tabs[2][len(bnd)])      # NaN: every comparison false -> last bucket
    assert np.array_equal(got[2, 12:], tabs[2][0])             # -inf: below the first boundary
    assert np.array_equal(got[3, 12:], tabs[2][2]) and np.array_equal(got[4, 12:], tabs[2][2])  # -0.0 == 0.0 >= boundary 0.0


def test_reference_ae_model_e_reduced(torch_cuda, oracle):
    """The reference's own model E recipe (examples/python/dlrm.py:140-203), with the
    2^23-row tables reduced so the oracle can hold them."""
    from recom_amd import synth
    m = synth.model_ae("E", batch=64, large_rows=50_000)
    assert m.spec.n_columns == 1000 and m.spec.group_width(0) == 995 * 8 + 5 * 32
    tabs = m.numpy_tables()
    req = m.make_request(0)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.gpu
@pytest.mark.parametrize("seg64", [True, False])
def test_segment_ids_through_a_folded_sparse_reshape(torch_cuda, oracle, tmp_path, seg64):
    """The general SparseReshape the reference folds into its generated index expression (EmitInputInline,
    cuda_emitter.cc:1874-1916; fcp_column_ext_t::seg_map_*): the pre-pass evaluates seg = (sum idx_k * mul_k) / div on the
    ORIGINAL index matrix.  Bit-exact against (a) the oracle on the same mapped plan, (b) the HIP path itself on the
    plain plan whose segment ids were reshaped with NumPy beforehand; static and per-request factors; also through a
    version-4 plan file, and row-sharded (the finalize pass re-reads the mapped ids for the mean's count)."""
    from recom_amd.ops import FeatureColumnProcess
    from recom_amd.plan_io import load_plan, save_plan
    from segmap_cases import build
    torch = torch_cuda
    for seed in range(4):
        spec_m, spec_p, ins_m, ins_p, tables, symbols = build(seed, batch=9 + 7 * seed, max_nnz=3000, seg64=seg64)
        out_m, packed_m, _ = run_gpu(torch, spec_m, ins_m, tables, symbols)
        assert_equal_oracle(oracle, spec_m, packed_m, tables, symbols, out_m)
        out_p, _, _ = run_gpu(torch, spec_p, ins_p, tables, symbols)
        for a, b in zip(out_m.groups, out_p.groups):
            assert torch.equal(a, b)
    # the plan file carries the maps
    path = str(tmp_path / "mapped.fcp")
    save_plan(spec_m, path)
    assert open(path).read().startswith("fcp_plan 4\n") and load_plan(path).to_dict() == spec_m.to_dict()
    op = FeatureColumnProcess.from_plan_file(path, 0)
    out_f, _, _ = run_gpu(torch, spec_m, ins_m, tables, symbols, op=op)
    for a, b in zip(out_f.groups, out_m.groups):
        assert torch.equal(a, b)
    # row-sharded partials equal the sharded oracle
    world = 3
    for rank in range(world):
        spec = spec_m.with_shard(rank, world)
        shard_tabs = [np.ascontiguousarray(t[rank::world]) for t in tables]
        out, packed, _ = run_gpu(torch, spec, ins_m, shard_tabs, symbols)
        assert_equal_oracle(oracle, spec, packed, shard_tabs, symbols, out)


@pytest.mark.gpu
def test_a_table_beyond_64_gb(torch_cuda, oracle):
    """one table of 80 GB (1.25 G rows x 16 floats = 5 G slots of 16 bytes > 2^32).  The kernels park a
    ROW per id and form the byte offset in 64 bits (round 2 parked 32-bit slot offsets: such a plan was refused).  A one-hot
    column, a pooled one and a ScatterNd one over the same table — once as an all-one-hot plan (dense kernel, wide path) and
    once mixed (hybrid launch) — against the oracle run on a COMPACT table that holds only the touched rows (closed-form
    contents, synth.hash_rows): ids around slot 2^32, the last row, the first, random ones."""
    import dataclasses
    from recom_amd import synth
    from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, FORM_GATHER, FORM_GATHER_SCATTER, FORM_SEGMENT_REDUCE, IDS_I64,
                                ROWS_FROM_IDS, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_IDS_I64, SEG_NONE, ColumnSpec, PlanSpec)
    torch = torch_cuda
    vocab, dim, B = 1_250_000_000, 16, 96
    free, _ = torch.cuda.mem_get_info()
    if free < vocab * dim * 4 + (8 << 30):
        pytest.skip(f"needs {vocab * dim * 4 / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} GiB free")
    dev = torch.device("cuda", 0)
    table = synth.hash_table_torch(5, vocab, dim, dev)
    rng = np.random.default_rng(3)
    edge = np.asarray([0, 1, (1 << 30) - 1, 1 << 30, (1 << 30) + 1, (1 << 28), vocab - 1, vocab - 2, vocab, -1], np.int64)
    def draw(n):
        ids = rng.integers(0, vocab, n).astype(np.int64)
        k = min(n, edge.size)
        ids[rng.choice(n, k, replace=False)] = edge[:k]
        return ids
    lens = rng.integers(0, 9, B)
    nnz = int(lens.sum())
    rows = np.repeat(np.arange(B, dtype=np.int64), lens)
    inputs = [draw(B), draw(nnz), np.concatenate([[0], np.cumsum(lens)]).astype(np.int32),
              draw(B // 2), np.stack([rng.permutation(B)[:B // 2], np.zeros(B // 2, np.int64)], 1).astype(np.int64)]
    cols = [ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, IDS_I64, 0, 0, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0, 0),
            ColumnSpec(FORM_SEGMENT_REDUCE, dim, vocab, COMBINER_MEAN, IDS_I64, 0, 1, 2, SEG_CSR_I32, 1, ROWS_FROM_SYMBOL, 0, None, 0, 1),
            ColumnSpec(FORM_GATHER_SCATTER, dim, vocab, COMBINER_NONE, IDS_I64, 0, 3, 4, SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, 0, None, 0, 2)]
    spec = PlanSpec(cols, [1, 1, 1, 1, 2], [8, 8, 4, 8, 8], n_device_inputs=1, n_groups=1, n_symbols=1)
    symbols = np.asarray([B], np.int32)
    # the compact twin: touched rows only, ids renumbered (ids outside the vocabulary stay outside)
    touched = np.unique(np.concatenate([inputs[0], inputs[1], inputs[3]]))
    touched = touched[(touched >= 0) & (touched < vocab)]
    compact = synth.hash_rows(5, touched, dim)
    remap = lambda ids: np.where((ids >= 0) & (ids < vocab), np.searchsorted(touched, np.clip(ids, 0, vocab - 1)), -1).astype(np.int64)
    c_inputs = [remap(inputs[0]), remap(inputs[1]), inputs[2], remap(inputs[3]), inputs[4]]
    c_spec = dataclasses.replace(spec, columns=[dataclasses.replace(c, vocab=int(touched.size)) for c in cols])
    blob, offsets, shapes = oracle.concat_inputs(c_inputs)
    want, _ = oracle.process_feature_columns(c_spec.to_dict(), blob, offsets, shapes, [compact], symbols)
    out, _, _ = run_gpu(torch, spec, inputs, None, symbols, tables_dev=[table])                    # hybrid launch
    assert np.array_equal(out.groups[0].cpu().numpy(), want[0])
    one_hot = PlanSpec([cols[0]], [1], [8], n_device_inputs=1, n_groups=1, n_symbols=0)           # dense kernel alone
    out1, _, _ = run_gpu(torch, one_hot, inputs[:1], None, None, tables_dev=[table])
    assert np.array_equal(out1.groups[0].cpu().numpy(), want[0][:, :dim])
    del table, out, out1
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_wide_row_path_of_the_dense_kernel_on_ordinary_plans(torch_cuda, oracle, golden, monkeypatch):
    """The dense body's 64-bit row path (taken when some table has 2^32 - 3 slots or more) forced onto ordinary plans
    (FCP_DIAG=wide_rows, read at plan creation): golden cases and the mixed model stay bit-exact."""
    from recom_amd import synth
    monkeypatch.setenv("FCP_DIAG", "wide_rows")
    for name in ("mixed_s0", "bucketize_kat", "scatter"):
        case = golden[0][name]
        out, _, _ = run_gpu(torch_cuda, case.spec(), case.inputs, case.tables, case.symbols)
        check_against_expected(case, [g.cpu().numpy() for g in out.groups])
    m = synth.model_s1() if hasattr(synth, "model_s1") else synth.model_mixed(batch=64)
    tabs = m.numpy_tables()
    req = m.make_request(2)
    out, packed, _ = run_gpu(torch_cuda, m.spec, req.inputs, tabs, req.symbols)
    assert_equal_oracle(oracle, m.spec, packed, tabs, req.symbols, out)


@pytest.mark.gpu
def test_concat_outputs_host_from_many_threads_and_without_an_allocator(torch_cuda):
    """fcp_concat_outputs_host holds its per-device ring lock for the slot bookkeeping only — eight
    threads on eight streams scatter their own payloads concurrently, more calls in flight than the ring has slots, small
    (read through the pinned mapping) and large (copied) payloads mixed; and a payload of at most 1 MiB needs no
    malloc_temp callback at all."""
    import ctypes as C
    import threading
    from recom_amd import lib as _lib
    from recom_amd.ops import concat_outputs_host
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    errors = []

    def worker(t):
        try:
            r = np.random.default_rng(100 + t)
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                for it in range(25):
                    prefix = int(r.choice([1, 64, 513, 6000 if (t + it) % 5 == 0 else 200]))
                    dims = [int(d) for d in r.choice([4, 8, 12, 32, 64], size=3)]
                    width = sum(dims) + 8
                    offs = [4, 4 + dims[0], 8 + dims[0] + dims[1]]
                    host = [r.standard_normal((prefix, d)).astype(np.float32) for d in dims]
                    out = torch.full((prefix, width), -7.0, device=dev)
                    concat_outputs_host(host, offs, out, stream.cuda_stream)
                    stream.synchronize()
                    got = out.cpu().numpy()
                    want = np.full((prefix, width), -7.0, np.float32)
                    for h, o, d in zip(host, offs, dims):
                        want[:, o:o + d] = h
                    assert np.array_equal(got, want), (t, it)
        except Exception as e:                                  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[0]
    # no allocator: fine for a small payload, refused (not crashed) for one that must be copied
    L = _lib.load()
    small = rng.standard_normal((16, 8)).astype(np.float32)
    big = rng.standard_normal((70000, 8)).astype(np.float32)                 # 2.2 MB > the 1 MiB direct limit
    for payload, ok in ((small, True), (big, False)):
        out = torch.zeros((payload.shape[0], 8), device=dev)
        ptrs = (C.c_void_p * 1)(payload.ctypes.data)
        dims, offs = np.asarray([8], np.int32), np.asarray([0], np.int32)
        rc = L.fcp_concat_outputs_host(ptrs, dims.ctypes.data, offs.ctypes.data, 1, payload.shape[0], 8, out.data_ptr(),
                                       _lib.ALLOC_FN(), None, 0, torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        assert (rc == 0) == ok
        if ok:
            assert np.array_equal(out.cpu().numpy(), payload)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_random_sparse_reshape_graphs_to_hip_path(torch_cuda, tmp_path, seed):
    """GraphDef -> plan -> HIP for random SparseReshapes (identity, constant and run-time segment-id maps, ops left to
    TensorFlow; tests/graph_fixtures.py::random_sparse_reshape_model): the rewritten graph with the HIP path behind its
    ops equals the original graph bit for bit."""
    from graph_fixtures import random_sparse_reshape_model
    gd, feeds, variables, fetches = random_sparse_reshape_model(seed)
    _graph_through_hip(torch_cuda, gd, feeds, variables, fetches, tmp_path)


@pytest.mark.gpu
def test_bench_refuses_to_time_wrong_results(torch_cuda):
    """bench.py serves every resident request once and compares it with the closed-form tables before the warm-up
    (ServingHarness.verify_resident, no oracle): correct kernels pass on one-hot and pooled columns alike, and one
    flipped table value is caught."""
    from recom_amd import synth
    from recom_amd.harness import ServingHarness
    torch = torch_cuda
    for model in (synth.model_s1(columns=40, dim=12, vocab=500, batch=64),
                  synth.staged_model(synth.model_ragged(columns=24, vocab=400, batch=48, seg="indices"))):
        h = ServingHarness(model, n_requests=4)
        got = h.verify_resident()
        assert got["checked"] > 0
        h.close()
        tables = model.torch_tables(torch.device("cuda", 0))
        for t in tables:
            t += 0.25                                       # every row differs from the closed form now
        h = ServingHarness(model, n_requests=4, tables=tables)
        with pytest.raises(RuntimeError, match="closed-form"):
            h.verify_resident()
        h.close()


def test_sla_bounded_throughput_search_follows_the_reference_protocol(torch_cuda):
    """The reference's throughput benchmark for this path (benchmark_throughput, recom_examples.patch:264-465): the batch grows
    from 16 — doubling below half of the SLA, by ever smaller fractions towards it — until the average latency of a request
    reaches the SLA; the result is the largest batch that stayed under it and its throughput."""
    from recom_amd import synth
    from recom_amd.harness import sla_throughput_search
    res = sla_throughput_search(lambda b: synth.model_s2(columns=48, vocab=5000, batch=b), sla_ms=0.05, serve_workers=1,
                                num_iterations=40, arena_budget_bytes=2 << 30)
    s = res["search"]
    assert s[0]["batch"] == 16 and s[1]["batch"] == 32                     # far below the SLA: the batch doubles
    assert all(a["batch"] < b["batch"] for a, b in zip(s, s[1:]))
    assert res["ended_by"] in ("sla", "arena memory") and res["max_batch_size"] >= 32 and res["max_throughput"] > 0
    under = [x for x in s if x["avg_latency_ms"] < 0.05]
    assert res["max_batch_size"] == under[-1]["batch"] and res["max_throughput"] == under[-1]["throughput"]
    if res["ended_by"] == "sla":
        assert s[-1]["avg_latency_ms"] >= 0.05 and s[-1]["batch"] > 512
