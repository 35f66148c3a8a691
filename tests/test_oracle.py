"""Pins the CPU oracle (oracle/fcp_oracle.c) — CPU only.

The reference has no tests or golden vectors for this path, so the
C oracle is pinned against (1) the committed fixtures in tests/golden/ (NumPy
float64 + PyTorch-CPU expectations, see make_golden.py), (2) PyTorch-CPU
``embedding_bag`` / ``bucketize`` / ``index_select`` evaluated live, (3) its own
NumPy restatement, (4) the reference's dim>20 summation order restated in
``orc_sparse_segment_reduce_ref8x8``, and (5) tests/golden/ref_device_goldens.npz:
outputs of the reference's OWN device templates, compiled unmodified for gfx950
and run on an MI355X (make_ref_device_goldens.py) — bit for bit, see the last test.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fcp_oracle as O
from conftest import GOLDEN_NAMES, check_against_expected


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_oracle_matches_golden(oracle, golden, name):
    case = golden[0][name]
    outs, bad = oracle.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables,
                                               case.symbols)
    assert bad == 0
    check_against_expected(case, outs)


def test_oracle_threads_equal_serial(oracle, golden):
    case = golden[0]["mixed_s0"]
    a, _ = oracle.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables, case.symbols, 1)
    b, _ = oracle.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables, case.symbols, 4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_bucketize_kat(oracle, golden):
    z = golden[1]
    case = golden[0]["bucketize_kat"]
    got = oracle.bucketize(case.plan["columns"][0]["boundaries"], case.inputs[0])
    assert np.array_equal(got, z["bucketize_kat/expected_buckets"])


def test_bucketize_vs_torch_and_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 100, 255):
        b = np.sort(rng.uniform(-50, 50, n)).astype(np.float32)
        v = np.concatenate([rng.uniform(-60, 60, 1000).astype(np.float32), b, b - 1e-3, b + 1e-3])
        got = oracle.bucketize(b, v)
        assert np.array_equal(got, O.np_bucketize(b, v))
        t = torch.bucketize(torch.from_numpy(v), torch.from_numpy(b), right=True).numpy()
        assert np.array_equal(got, t)
    # NaN never compares below a boundary: last bucket (cuda_emitter.cc:240-244)
    assert oracle.bucketize(np.arange(5, dtype=np.float32), np.asarray([np.nan], np.float32))[0] == 5


def test_gather_rows_vs_torch(oracle):
    rng = np.random.default_rng(1)
    for dim in (1, 4, 8, 20, 64):
        W = rng.standard_normal((300, dim)).astype(np.float32)
        ids = rng.integers(0, 300, 257)
        out, bad = oracle.gather_rows(W, ids)
        assert bad == 0
        assert np.array_equal(out, torch.from_numpy(W).index_select(0, torch.from_numpy(ids)).numpy())
    out, bad = oracle.gather_rows(W, np.asarray([0, -1, 300, 299]))
    assert bad == 2 and not out[1].any() and not out[2].any() and np.array_equal(out[3], W[299])


def test_segment_offsets(oracle):
    rng = np.random.default_rng(2)
    for B in (1, 5, 64, 200):
        for _ in range(5):
            lens = rng.integers(0, 4, B)
            seg = np.repeat(np.arange(B), lens)
            off = oracle.segment_offsets(seg, B)
            assert np.array_equal(off, np.concatenate([[0], np.cumsum(lens)]))
            assert np.array_equal(off, O.np_segment_offsets(seg, B))
    assert np.array_equal(oracle.segment_offsets(np.zeros(0, np.int64), 3), [0, 0, 0, 0])
    # ids beyond num_segments fall off the end
    assert np.array_equal(oracle.segment_offsets(np.asarray([0, 1, 7, 9]), 3), [0, 1, 2, 2])


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("dim", [4, 8, 20, 32, 64])
def test_segment_reduce_vs_embedding_bag(oracle, mean, dim):
    rng = np.random.default_rng(dim + mean)
    W = rng.standard_normal((500, dim)).astype(np.float32)
    lens = rng.integers(0, 40, 77)
    lens[:3] = [0, 1, 130]
    ids = rng.integers(0, 500, int(lens.sum()))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    out, bad = oracle.sparse_segment_reduce(W, ids, off, mean)
    assert bad == 0
    truth = O.np_sparse_segment_reduce(W, ids, off, mean)
    # bags of up to 130 N(0,1) rows: |sum| reaches ~30, one fp32 ulp there is 2e-6
    tol = 1e-5 if mean else 5e-5
    assert np.abs(out - truth).max() < tol
    bag = F.embedding_bag(torch.from_numpy(ids), torch.from_numpy(W), torch.from_numpy(off.astype(np.int64)),
                          mode="mean" if mean else "sum", include_last_offset=True).numpy()
    assert np.abs(out - bag).max() < tol
    assert not out[0].any()  # empty segment -> zeros
    # the reference's own dim>20 summation order differs only by fp32 reassociation
    ref, _ = oracle.sparse_segment_reduce(W, ids, off, mean, ref_order=True)
    assert np.abs(ref - truth).max() < tol
    # a one-id segment is a pure copy in sum mode, in either order
    if not mean:
        assert np.array_equal(out[1], W[ids[0]]) and np.array_equal(ref[1], W[ids[0]])


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("dim", [4, 8, 12, 16, 20])
def test_reference_block_scan_order_dim_le_20(oracle, mean, dim):
    """The reference's dim <= 20 template adds in CUB block-scan order (64-id tiles, Kogge-Stone inside each
    32-lane warp, warp aggregate, carry across tiles: cuda_emitter.cc:348-661, :1542-1618), restated in
    orc_sparse_segment_reduce_refscan.  Against the sequential id order of the oracle (= the HIP path; TF-CPU's order
    for bags of up to 9 ids, test_tensorflow_cpu_addition_order) it differs by fp32 reassociation only; segments of one or two ids are identical in any order;
    the restatement's own tile logic (segments crossing tiles, empty rows, leading empty rows, nnz a
    multiple of 64) is checked against float64."""
    rng = np.random.default_rng(100 * dim + mean)
    W = rng.standard_normal((400, dim)).astype(np.float32)
    for lens in (rng.integers(0, 12, 90), np.asarray([0, 0, 1, 2, 63, 64, 65, 130, 0, 1, 0]), np.full(32, 2), np.asarray([0, 128]),
                 np.asarray([3]), np.zeros(5, np.int64)):
        ids = rng.integers(0, 400, int(lens.sum()))
        rows = np.repeat(np.arange(len(lens)), lens)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        seq, _ = oracle.sparse_segment_reduce(W, ids, off, mean)
        ref = oracle.sparse_segment_reduce_refscan(W, ids, rows, len(lens), mean)
        truth = O.np_sparse_segment_reduce(W, ids, off, mean)
        tol = 1e-5 if mean else 5e-5                       # |sum| of 130 N(0,1) rows reaches ~30: one ulp there is 2e-6
        assert np.abs(ref - truth).max(initial=0) < tol and np.abs(ref - seq).max(initial=0) < tol
        short = lens <= 2
        assert np.array_equal(ref[short], seq[short])      # a + b in either order; empty rows are zeros
        assert not ref[lens == 0].any()
    # the order really is the tree, not the sequence: a long bag of badly conditioned values differs in the last bits
    W2 = (rng.standard_normal((400, dim)) * 10.0 ** rng.integers(-3, 4, (400, 1))).astype(np.float32)
    ids = rng.integers(0, 400, 64)
    seq, _ = oracle.sparse_segment_reduce(W2, ids, np.asarray([0, 64], np.int32), False)
    ref = oracle.sparse_segment_reduce_refscan(W2, ids, np.zeros(64, np.int64), 1, False)
    assert not np.array_equal(seq, ref) and np.allclose(seq, ref, rtol=1e-4, atol=1e-2)


def test_block_scan_order_selector_changes_the_scan_order_only(oracle):
    """orc_sparse_segment_reduce_refscan_assoc: ORC_SCAN_ROCPRIM64 (the order of hipCUB's 64-thread BlockScan on one wavefront;
    tests/test_gpu_reference_kernels.py holds the reference's template compiled against hipCUB to it bit for bit) and
    ORC_SCAN_CUB18 share everything but the order inside a 64-item tile: identical on integer-valued rows (any order is exact),
    on bags of up to two ids and on empty rows; equal to a NumPy statement of each tree for one full tile of one bag."""
    rng = np.random.default_rng(77)
    Wi = rng.integers(-50, 50, (300, 12)).astype(np.float32)
    lens = rng.integers(0, 30, 120)
    ids = rng.integers(0, 300, int(lens.sum()))
    rows = np.repeat(np.arange(len(lens)), lens)
    for mean in (False, True):
        a = oracle.sparse_segment_reduce_refscan(Wi, ids, rows, len(lens), mean)
        b = oracle.sparse_segment_reduce_refscan(Wi, ids, rows, len(lens), mean, rocprim=True)
        assert np.array_equal(a, b) and not b[lens == 0].any()
    W = (rng.standard_normal((300, 8)) * 10.0 ** rng.integers(-3, 4, (300, 1))).astype(np.float32)
    ids = rng.integers(0, 300, 64)
    x = W[ids]

    def kogge_stone(v):                       # inclusive scan of v[n, dim], float32, d = 1, 2, 4 ...
        v = v.copy()
        d = 1
        while d < len(v):
            v[d:] = v[:-d] + v[d:]
            d *= 2
        return v
    cub = np.concatenate([kogge_stone(x[:32]), kogge_stone(x[32:])])
    cub[32:] = cub[31] + cub[32:]
    roc = np.concatenate([kogge_stone(x[r:r + 16]) for r in range(0, 64, 16)])
    roc[16:32] = roc[15] + roc[16:32]
    roc[48:64] = roc[47] + roc[48:64]
    roc[32:64] = roc[31] + roc[32:64]
    assert np.array_equal(cub[63], roc[63])   # the last lane's tree is the same in both; lane 20's is not
    for last in (63, 20, 40, 27):             # a bag of ids 0..last, then a second bag with the rest of the tile
        rows = (np.arange(64) > last).astype(np.int64)
        assert np.array_equal(oracle.sparse_segment_reduce_refscan(W, ids, rows, 2, False)[0], cub[last])
        assert np.array_equal(oracle.sparse_segment_reduce_refscan(W, ids, rows, 2, False, rocprim=True)[0], roc[last])
    assert not np.array_equal(cub[20], roc[20]) and not np.array_equal(cub[27], roc[27])


def _np_tfcpu_segment(rows, mean):
    """Independent NumPy-float32 statement of TensorFlow 2.6.2's SparseSegmentReductionOpBase::Reduce for ONE segment
    (rows: [num, dim] float32): first num & 7 rows (8 for 0, 9 for 1) left to right, / num at once when mean and num < 10,
    every further 8 rows summed among themselves and added, / num at the end when mean and num >= 10."""
    num = rows.shape[0]
    if num == 0:
        return np.zeros(rows.shape[1], np.float32)
    if num == 1:
        return rows[0].copy()
    r = num & 7
    r = {0: 8, 1: 9}.get(r, r)
    acc = rows[0].copy()
    for k in range(1, r):
        acc = acc + rows[k]
    if mean and num < 10:
        acc = acc / np.float32(num)
    for c in range(r, num, 8):
        t = rows[c].copy()
        for k in range(1, 8):
            t = t + rows[c + k]
        acc = acc + t
    if mean and num >= 10:
        acc = acc / np.float32(num)
    return acc


@pytest.mark.parametrize("mean", [False, True])
@pytest.mark.parametrize("dim", [4, 8, 32])
def test_tensorflow_cpu_addition_order(oracle, mean, dim):
    """The north star's tolerance is "vs TF-CPU".  TF 2.6.2's CPU kernel (segment_reduction_ops_impl.h,
    SparseSegmentReductionOpBase::Reduce; third party, absent here, restated from its published source in
    orc_sparse_segment_reduce_tfcpu — unpinned) adds the first num & 7 rows left to right and from then on every 8 rows
    among themselves first.  So: for bags of UP TO 9 ids it performs exactly the additions of the sequential order (the
    oracle's default and the HIP path's): bit-identical.  From 10 ids on it differs by fp32 reassociation, bounded here
    at the north star's 1e-5 for BASELINE's bag lengths (<= 10) and against float64 for long bags."""
    rng = np.random.default_rng(1000 * dim + mean)
    W = (rng.standard_normal((500, dim)) * dim ** -0.5).astype(np.float32)
    lens = np.concatenate([np.arange(0, 27), rng.integers(0, 11, 200), [63, 64, 65, 300]])
    ids = rng.integers(0, 500, int(lens.sum()))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tf = oracle.sparse_segment_reduce_tfcpu(W, ids, off, mean)
    seq, _ = oracle.sparse_segment_reduce(W, ids, off, mean)
    truth = O.np_sparse_segment_reduce(W, ids, off, mean)
    for s, n in enumerate(lens):                                # the C restatement against the independent NumPy one
        assert np.array_equal(tf[s], _np_tfcpu_segment(W[ids[off[s]:off[s + 1]]], mean)), (s, n)
    short = lens <= 9
    assert np.array_equal(tf[short], seq[short])                # the same additions: bit-identical
    assert not tf[lens == 0].any()
    base = lens <= 10                                           # BASELINE configs[3] draws U{0..10} ids per row
    assert np.abs(tf[base] - seq[base]).max() < 1e-5 and np.abs(tf[base] - truth[base]).max() < 1e-5
    assert np.abs(tf - truth).max() < (1e-5 if mean else 1e-4) and np.abs(tf - seq).max() < (1e-5 if mean else 1e-4)
    # and the order really is different from 10 ids on: badly conditioned rows expose it in the last bits
    W2 = (rng.standard_normal((500, dim)) * 10.0 ** rng.integers(-3, 4, (500, 1))).astype(np.float32)
    ids2 = rng.integers(0, 500, 10 * 64)
    off2 = (np.arange(65) * 10).astype(np.int32)
    assert not np.array_equal(oracle.sparse_segment_reduce_tfcpu(W2, ids2, off2, mean), oracle.sparse_segment_reduce(W2, ids2, off2, mean)[0])


def _xform_spec(m, transforms):
    import dataclasses
    cols = list(m.spec.columns)
    for k, t in transforms.items():
        mode, ivals, sub = t[:3]
        cols[k] = dataclasses.replace(cols[k], xform_mode=mode, xform_lo=tuple(a for a, _ in ivals),
                                      xform_hi=tuple(b for _, b in ivals), xform_substitute=sub,
                                      hash_buckets=t[3] if len(t) > 3 else 0)
    spec = dataclasses.replace(m.spec, columns=cols)
    spec.validate()
    return spec


# column -> (xform_mode, closed intervals, substitute[, hash_buckets]); columns 0, 4 and 7 hash their ids first
# (Fingerprint64 of the decimal string, TensorFlow's AsString -> StringToHashBucketFast)
XFORMS = {0: (2, [(100, 500)], 0, 997), 1: (1, [(10, 60), (80, 90)], 3), 2: (1, [], 7), 3: (2, [(0, 300), (600, 996)], 0),
          4: (2, [(200, 800)], 0, 900), 5: (2, [(0, 498)], 0), 6: (1, [(0, 500)], -5), 7: (0, [], 0, 640)}


@pytest.mark.parametrize("batch,seed", [(33, 0), (1, 1), (120, 2)])
def test_id_transforms_equal_the_cpu_ops_followed_by_the_lookup(oracle, batch, seed):
    """a column with a fused id transform == the reference's CPU op applied to the request
    (SelectValue: elementwise; GatherIndiceValue / GatherValueGenIndice: compaction of the (index, value)
    pairs; select_value_ops.cc:33-56, gather_indice_value_ops.cc:33-78, gather_value_gen_indice_ops.cc:33-67,
    with the intended `lo <= x && x <= hi`) followed by the untransformed column.  Also pinned against the
    float64 NumPy restatement."""
    from recom_amd import synth
    from recom_amd.ops import concat_inputs
    from recom_amd.plan import FORM_GATHER, SEG_CSR_I32
    m = synth.model_mixed(batch=batch, vocab=997, n_groups=1)
    spec = _xform_spec(m, XFORMS)
    tabs = m.numpy_tables()
    req = m.make_request(seed)
    packed = concat_inputs(req.inputs)
    got, bad = oracle.process_feature_columns(spec.to_dict(), *packed, tabs, req.symbols)
    # the same request after the CPU ops, through the plain plan
    inputs = [np.array(a) for a in req.inputs]
    zero_rows = {}
    for k, t in XFORMS.items():
        mode, ivals, sub = t[:3]
        c = m.spec.columns[k]
        raw = inputs[c.ids_input]
        ids = O.np_bucketize(c.boundaries, raw).astype(np.int64) if c.id_source == 2 else raw.astype(np.int64)
        if len(t) > 3:                                        # AsString -> StringToHashBucketFast in front of everything
            ids = np.asarray([O.np_fingerprint64(str(int(v)).encode()) % t[3] for v in ids], np.int64)
            raw = ids.astype(raw.dtype)
            inputs[c.ids_input] = raw
        inside = np.zeros(ids.size, bool)
        for lo, hi in ivals:
            inside |= (ids >= lo) & (ids <= hi)
        if c.id_source == 2:
            continue                                          # checked through the NumPy restatement below
        if mode == 0:
            continue
        if mode == 1:                                         # Addons>SelectValue
            inputs[c.ids_input] = np.where(inside, ids, sub).astype(raw.dtype)
        elif c.form == FORM_GATHER:                           # Addons>GatherValueGenIndice: dropped values leave zero rows
            zero_rows[k] = ~inside
        else:                                                 # Addons>GatherIndiceValue
            inputs[c.ids_input] = raw[inside]
            seg = inputs[c.seg_input]
            if c.seg_kind == SEG_CSR_I32:
                rows = np.repeat(np.arange(seg.size - 1), np.diff(seg))[inside]
                inputs[c.seg_input] = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=seg.size - 1))]).astype(np.int32)
            else:
                inputs[c.seg_input] = seg[inside]                  # rows of the [nnz, k] (or [nnz]) index tensor
    plain_cols = {k: v for k, v in XFORMS.items() if m.spec.columns[k].id_source == 2}
    want, bad2 = oracle.process_feature_columns(_xform_spec(m, plain_cols).to_dict(), *concat_inputs(inputs), tabs, req.symbols)
    want = want[0].copy()
    offs = m.spec.column_offsets()
    for k, z in zero_rows.items():
        want[z, offs[k]:offs[k] + m.spec.columns[k].dim] = 0.0
    assert np.array_equal(got[0], want)
    truth = O.np_process_feature_columns(spec.to_dict(), *packed, tabs, req.symbols)
    assert np.abs(got[0] - truth[0]).max() < 1e-5
    # substituted id -5 is out of range (counted), dropped ids are not lookups at all
    c6 = m.spec.columns[6]
    n_sub = int((~((req.inputs[c6.ids_input] >= 0) & (req.inputs[c6.ids_input] <= 500))).sum())
    assert bad == n_sub


def test_gather_scatter(oracle):
    W = np.arange(40, dtype=np.float32).reshape(10, 4)
    out, bad = oracle.gather_scatter_rows(W, [1, 2, 3, 9], [0, 2, 2, 5], 7)
    assert bad == 0
    assert np.array_equal(out[0], W[1]) and np.array_equal(out[2], W[3]) and np.array_equal(out[5], W[9])
    assert not out[[1, 3, 4, 6]].any()


def test_concat_outputs_and_batch_col_reduction(oracle):
    rng = np.random.default_rng(3)
    xs = [rng.standard_normal((9, d)).astype(np.float32) for d in (3, 8, 1, 16)]
    assert np.array_equal(oracle.concat_outputs(xs), np.concatenate(xs, axis=1))
    x = rng.standard_normal((6, 5, 7)).astype(np.float32)
    out = oracle.batch_col_reduction(x)
    assert np.abs(out - x.astype(np.float64).sum(axis=1)).max() < 1e-5
    seq = np.zeros((6, 7), np.float32)
    for r in range(5):
        seq = seq + x[:, r, :]  # r ascending, fp32 (cuda_emitter.cc:1231-1236)
    assert np.array_equal(out, seq)


def test_concat_inputs(oracle):
    rng = np.random.default_rng(4)
    ts = [rng.integers(0, 9, (5,)).astype(np.int64), rng.standard_normal((3, 2)).astype(np.float32),
          np.zeros((0, 2), np.int64), np.asarray(7, np.int32)]
    blob, off, shp = oracle.concat_inputs(ts)
    assert np.array_equal(off, [0, 40, 64, 64])
    assert np.array_equal(shp, [5, 3, 2, 0, 2])
    assert blob.nbytes == 68
    assert blob.tobytes() == b"".join(t.tobytes() for t in ts)


def test_sharded_partials_sum_to_unsharded(oracle, golden):
    """Row sharding: the sum over ranks of the per-rank partial
    sums, with the mean division applied afterwards, equals the 1-GPU result."""
    import copy
    case = golden[0]["mixed_s0"]
    full, _ = oracle.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables,
                                             case.symbols)
    world = 4
    acc = [np.zeros_like(f, dtype=np.float64) for f in full]
    for rank in range(world):
        plan = copy.deepcopy(case.plan)
        plan["shard_rank"], plan["shard_world"] = rank, world
        tabs = [t[rank::world] for t in case.tables]
        part, _ = oracle.process_feature_columns(plan, case.blob, case.offsets, case.shapes, tabs, case.symbols)
        for a, p in zip(acc, part):
            a += p
    # apply the mean division to mean columns (table-free columns are written by rank 0 only)
    from recom_amd.plan import PlanSpec
    spec = case.spec()
    offs = spec.column_offsets()
    so = spec.shape_offsets()
    for k, c in enumerate(spec.columns):
        sl = acc[c.concat_group][:, offs[k]:offs[k] + c.dim]
        if c.form == 2 and c.combiner == 2:
            rows = sl.shape[0]
            if c.seg_kind == 3:
                o = case.inputs[c.seg_input]
            else:
                seg = case.inputs[c.seg_input].reshape(-1)[::c.seg_stride]
                o = O.np_segment_offsets(seg, rows)
            cnt = np.diff(o).astype(np.float64)
            sl /= np.where(cnt > 0, cnt, 1.0)[:, None]
    for a, f in zip(acc, full):
        assert np.abs(a - f).max() < 1e-5


def test_tensorflow_documented_examples(oracle):
    """The worked examples of the TensorFlow API documentation for the ops the path fuses
    (tests/golden/tf_doc_examples.py): vectors that neither this repository nor the reference produced.  The C
    oracle and the NumPy restatement must reproduce every documented output exactly."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import tf_doc_examples as T
    b = T.BUCKETIZE
    assert np.array_equal(oracle.bucketize(b["boundaries"], b["values"]), b["expected"])
    assert np.array_equal(O.np_bucketize(b["boundaries"], b["values"].ravel()).reshape(b["expected"].shape), b["expected"])
    g = T.GATHER
    out, bad = oracle.gather_rows(g["params"], g["indices"])
    assert bad == 0 and np.array_equal(out, g["expected"])
    for case in T.SPARSE_SEGMENT_SUM + [dict(T.SPARSE_SEGMENT_MEAN, mean=True)]:
        offs = oracle.segment_offsets(case["segment_ids"], case["num_segments"])
        assert np.array_equal(offs, O.np_segment_offsets(np.asarray(case["segment_ids"]), case["num_segments"]))
        mean = case.get("mean", False)
        out, bad = oracle.sparse_segment_reduce(case["data"], case["indices"], offs, mean)
        want = np.asarray(case["expected"], np.float32)
        assert bad == 0 and np.array_equal(out, want), case
        assert np.array_equal(O.np_sparse_segment_reduce(case["data"], np.asarray(case["indices"]), offs, mean), want)
        # the reference GPU kernels' own addition orders give the same on these exact values
        assert np.array_equal(oracle.sparse_segment_reduce(case["data"], case["indices"], offs, mean, ref_order=True)[0], want)
        assert np.array_equal(oracle.sparse_segment_reduce_refscan(case["data"], case["indices"], case["segment_ids"],
                                                                  case["num_segments"], mean), want)
    sc = T.SCATTER_ND
    table = np.asarray(sc["updates"], np.float32).reshape(-1, 1)  # indices as documented ([4, 3, 1, 7]: any order)
    out, bad = oracle.gather_scatter_rows(table, np.arange(len(sc["indices"])), np.asarray(sc["indices"]), sc["size"])
    assert bad == 0 and np.array_equal(out.ravel(), np.asarray(sc["expected"], np.float32))
    assert np.array_equal(oracle.concat_outputs(T.CONCAT["inputs"]), T.CONCAT["expected"])
    h = T.TO_HASH_BUCKET_FAST
    assert [int(oracle.lib.orc_fingerprint64(s, len(s))) % h["num_buckets"] for s in h["strings"]] == h["expected"]


def test_fingerprint64_known_answers(oracle):
    """TensorFlow's Fingerprint64 (FarmHash farmhashna::Hash64; TF 2.6.2 pins farmhash 816a4ae6, un-vendored in
    the reference) restated in C (orc_fingerprint64) and independently in Python integers (np_fingerprint64).
    Published known answers: the empty string hashes to k2 (HashLen0to16's `return k2`); "abc" ->
    0x24a5b3a074e7f369 (the CityHash64 v1.1 / FarmHash short-string value); the TensorFlow API docs'
    tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) == [0, 2, 2] — one string per branch of
    HashLen0to16 (len >= 8, >= 4, > 0), the code every decimal id of up to 16 digits goes through."""
    L = oracle.lib
    fp = lambda s: int(L.orc_fingerprint64(s, len(s)))
    assert fp(b"") == 0x9ae16a3b2f90404f == O.np_fingerprint64(b"")
    assert fp(b"abc") == 0x24a5b3a074e7f369 == O.np_fingerprint64(b"abc")
    assert [fp(s) % 3 for s in (b"Hello", b"TensorFlow", b"2.x")] == [0, 2, 2]
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.integers(-2**63, 2**63 - 1, 400, dtype=np.int64, endpoint=True),
                           rng.integers(-1000, 1000, 200), [0, -1, 9, 10, 99999999, 10**15, 10**16, -10**16, 2**63 - 1, -2**63],
                           10 ** np.arange(0, 19, dtype=np.int64), 10 ** np.arange(1, 19, dtype=np.int64) - 1])
    for v in vals:                                             # C and Python restatements agree on every length 1..20
        s = str(int(v)).encode()
        assert fp(s) == O.np_fingerprint64(s), s
        for buckets in (1, 7, 100, 10_000, 2**31 - 1):
            assert L.orc_hash_bucket_int64(int(v), buckets) == O.np_fingerprint64(s) % buckets


def _bucketize_cases(rng):
    arrays = {
        "reference 0,5,...,495": np.arange(0, 500, 5, dtype=np.float32),
        "dyadic grid": (np.arange(37, dtype=np.float32) * np.float32(0.375) - np.float32(3.0)),
        "two boundaries": np.asarray([-1.0, 2.0], np.float32),
        "tenths": (np.arange(200) * 0.1).astype(np.float32),
        "thirds": (np.arange(1, 90) / 3.0).astype(np.float32),
        "log spaced": np.logspace(-3, 4, 150).astype(np.float32),
        "random": np.unique(rng.uniform(-50, 50, 300).astype(np.float32)),
        "single": np.asarray([7.5], np.float32),
        "huge step": np.asarray([-3e38, 0.0, 3e38], np.float32),
        "1024 random": np.unique(rng.standard_normal(4000).astype(np.float32))[:1024],
        "with duplicates": np.sort(np.repeat(rng.uniform(0, 10, 40).astype(np.float32), 2)),
    }
    for name, b in arrays.items():
        x = np.concatenate([b, np.nextafter(b, np.float32(-np.inf)), np.nextafter(b, np.float32(np.inf)), (b[:-1] + b[1:]) / 2,
                            [np.nan, np.inf, -np.inf, 0.0, -0.0, b[0] - 1, b[-1] + 1, 3.4e38, -3.4e38],
                            rng.uniform(b[0] - 3, b[-1] + 3, 300)]).astype(np.float32)
        yield name, b, x


def test_bucketize_is_pinned_by_the_reference_itself(oracle, ref_bucketize):
    """a5 is the one function of the path whose reference source is plain C++ inside its string literal: compiled from
    /root/reference by oracle/ref_extract.py (no copy in the repository), it pins the oracle's restatement and the NumPy
    twin on boundary-exact values, their float neighbours, NaN, infinities and signed zeros, for evenly spaced, uneven,
    duplicated and single-element boundary lists."""
    import fcp_oracle as O
    rng = np.random.default_rng(5)
    for name, b, x in _bucketize_cases(rng):
        want = ref_bucketize(b, x)
        assert np.array_equal(oracle.bucketize(b, x), want), name
        ok = ~np.isnan(x)                                   # searchsorted sorts NaN last: the same bucket (n) as the reference
        assert np.array_equal(O.np_bucketize(b, x)[ok], want[ok]) and np.all(want[~ok] == len(b)), name


@pytest.mark.parametrize("seg64", [True, False])
def test_segment_ids_through_a_folded_sparse_reshape(oracle, seg64):
    """Missing item 4 of VERDICT r02: the general SparseReshape the reference folds into its index expression
    (EmitInputInline, cuda_emitter.cc:1874-1916).  The oracle evaluating seg = (sum idx_k * mul_k) / div on the ORIGINAL
    index matrix must give what it gives on the row coordinate of the reshaped tensor, computed independently with
    NumPy's ravel / unravel (= the definition of SparseReshape) — C oracle and NumPy twin, static and per-request factors."""
    from segmap_cases import build
    for seed in range(6):
        spec_m, spec_p, ins_m, ins_p, tables, symbols = build(seed, seg64=seg64)
        blob_m, off_m, shp_m = oracle.concat_inputs(ins_m)
        blob_p, off_p, shp_p = oracle.concat_inputs(ins_p)
        want, bad_w = oracle.process_feature_columns(spec_p.to_dict(), blob_p, off_p, shp_p, tables, symbols)
        got, bad_g = oracle.process_feature_columns(spec_m.to_dict(), blob_m, off_m, shp_m, tables, symbols)
        twin = O.np_process_feature_columns(spec_m.to_dict(), blob_m, off_m, shp_m, tables, symbols)
        assert bad_w == bad_g
        for w, g, t in zip(want, got, twin):
            assert np.array_equal(w, g)
            assert np.abs(np.asarray(t, np.float64) - w).max(initial=0.0) < 1e-5


def test_oracle_equals_the_vectors_the_references_kernels_produced(oracle):
    """tests/golden/ref_device_goldens.npz: outputs of the REFERENCE's own device templates — GatherRowsToGlbMem,
    GatherScatterRows, experiment::ComputeSegmentOffsets / SparseSegmentReduce (cuda_emitter.cc:250-345, 664-962) and the
    dim <= 20 SparseSegmentSum / Mean (:348-661, against hipCUB) — compiled unmodified for gfx950 and run on an MI355X by
    tests/golden/make_ref_device_goldens.py (committed with the file).  The inputs are re-drawn from the seeds stored with
    every case; the C oracle's restatements must equal the stored outputs BIT FOR BIT, here on the CPU, without a GPU:
    orc_gather_rows, orc_gather_scatter_rows, orc_segment_offsets, orc_sparse_segment_reduce_ref8x8 (non-empty segments; an
    empty MEAN segment is NaN in the reference, zero in TF and the oracle) and orc_sparse_segment_reduce_refscan_assoc with
    the scan order of the library the vectors were produced against (ORC_SCAN_ROCPRIM64)."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_ref_device_goldens", os.path.join(here, "make_ref_device_goldens.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    G = np.load(os.path.join(here, "ref_device_goldens.npz"))
    n = int(G["n_cases"])
    assert n == len(gen.CASES) >= 50
    kinds = set()
    for k in range(n):
        rec = [str(v) for v in G[f"case_{k}"]]
        case = dict(kind=rec[0], seed=int(rec[1]), dim=int(rec[2]), vocab=int(rec[3]), B=int(rec[4]), max_len=int(rec[5]), mean=int(rec[6]))
        assert case == gen.CASES[k]                                       # the file belongs to this generator
        x, want, B, mean = gen.inputs_of(case), G[f"out_{k}"], case["B"], bool(case["mean"])
        kinds.add(case["kind"])
        if case["kind"] == "gather_rows":
            got, bad = oracle.gather_rows(x["table"], x["ids"])
            assert bad == 0 and np.array_equal(got, want)
        elif case["kind"] == "gather_scatter_rows":
            got, bad = oracle.gather_scatter_rows(x["table"], x["ids"], x["rows"], B)
            assert bad == 0 and np.array_equal(got, want)
        elif case["kind"] == "segment_offsets":
            assert np.array_equal(oracle.segment_offsets(x["seg"], B), want)
        elif case["kind"] == "segment_reduce_8x8":
            offs = G[f"offs_{k}"]
            assert np.array_equal(oracle.segment_offsets(x["seg"], B), offs)
            got, _ = oracle.sparse_segment_reduce(x["table"], x["ids"], offs, mean, ref_order=True)
            empty = x["lens"] == 0
            assert np.array_equal(got[~empty], want[~empty])
            assert (np.isnan(want[empty]).all() if mean else not want[empty].any()) and not got[empty].any()
        else:
            got = oracle.sparse_segment_reduce_refscan(x["table"], x["ids"], x["seg"], B, mean, rocprim=True)
            assert np.array_equal(got, want)
            cub = oracle.sparse_segment_reduce_refscan(x["table"], x["ids"], x["seg"], B, mean)
            assert np.abs(cub - want).max(initial=0) < (1e-5 if case["max_len"] <= 10 else 1e-4)   # the other scan order: reassociation only
    assert kinds == {"gather_rows", "gather_scatter_rows", "segment_offsets", "segment_reduce_8x8", "segment_reduce_scan"}


def test_tf_cpu_dataflow_of_the_cpu_baseline_equals_the_fused_layout(oracle):
    """bench.py's cpu_baseline serves through TensorFlow-CPU's dataflow for the unrewritten graph (one [rows, dim] tensor per
    column op, then ConcatV2: orc_process_feature_columns_unfused); the values are those of the checker's direct form bit
    for bit, bad-id count included, on the model with every form / id source / segment encoding — and the serving loop runs
    in both dataflows."""
    from recom_amd import synth
    from recom_amd.ops import concat_inputs
    for model in (synth.model_mixed(batch=33, vocab=97), synth.model_s2(columns=24, vocab=500, batch=20)):
        req = model.make_request(3)
        blob, offsets, shapes = concat_inputs(req.inputs)
        tables = model.numpy_tables()
        plan = model.spec.to_dict()
        a, bad_a = oracle.process_feature_columns(plan, blob, offsets, shapes, tables, req.symbols)
        b, bad_b = oracle.process_feature_columns(plan, blob, offsets, shapes, tables, req.symbols, unfused=True)
        assert bad_a == bad_b and len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    model = synth.model_s2(columns=24, vocab=500, batch=20)
    packed = [concat_inputs(model.make_request(i).inputs) for i in range(3)]
    for dataflow in (0, 1):
        done, sec = oracle.serve_for(model.spec.to_dict(), packed, model.numpy_tables(), None, 2, 0.05, dataflow)
        assert done >= 2 and sec > 0
