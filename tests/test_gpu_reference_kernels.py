"""The reference's OWN device templates as the parity pin.

oracle/_ref/libref_device.so is the unmodified text of the reference's CUB-free literals —
GatherRowsToGlbMem (cuda_emitter.cc:250-293), GatherScatterRows (:296-345), AlignedVector (:664-765),
experiment::ComputeSegmentOffsets / experiment::SparseSegmentReduce (:768-962) — compiled for gfx950 by
oracle/ref_extract.py behind `__global__` wrappers that restate the generated driver loops (oracle/ref_device_wrap.hip,
citing :1305-1327, :1415-1439, :1734-1753) and run HERE on the GPU, one 64-thread block per column as FusedKnl does.
Held to it, bit for bit:
  * the oracle's restatements orc_gather_rows, orc_gather_scatter_rows, orc_segment_offsets and
    orc_sparse_segment_reduce_ref8x8 (the dim > 20 template in ITS OWN summation order),
  * the HIP product path through the C ABI for the copy forms (a6, a9) and for segment offsets (a8, via the pooled result).
The pooled HIP result (sequential id order) is bounded against it at the north star's 1e-5.
Function-level pinning: the templates are the reference's, the dozen driver lines around them are restated.  The dim <= 20
templates (SparseSegmentSum / Mean, :348-661) need cub::BlockScan: CUB 1.8 is absent, so they run against hipCUB's BlockScan
(oracle/_ref/libref_device_scan.so, last test of this file) and the oracle's restatement equals that bit for bit once the
order inside its 64-item scan is switched to rocPRIM's; the CUB 1.8 order itself (a dozen lines) stays restated only.
Documented divergences, asserted below: an empty MEAN segment is 0/0 = NaN in the reference's dim > 20 template and 0 in
TensorFlow, the oracle and the HIP path; ids / rows outside their range are read / written out of bounds by the reference
(never fed to it here) and are zeros / dropped in the oracle and the HIP path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from recom_amd import lib
    lib.load()  # fail loudly if the HIP extension is missing (and: one HIP runtime in the process, torch's)
    return torch


@pytest.fixture(scope="module")
def ref(torch_cuda):
    """ctypes face of oracle/_ref/libref_device.so (host pointers in, host pointers out; it copies and launches itself)."""
    import ref_extract
    path = ref_extract.device_lib_path()
    if path is None:
        pytest.skip("oracle/_ref/libref_device.so is absent and /root/reference is not here to build it from")
    L = C.CDLL(path)
    assert L.ref_dev_block_threads() == 64

    class Ref:
        @staticmethod
        def gather_rows(table, ids):
            t, i = np.ascontiguousarray(table, np.float32), np.ascontiguousarray(ids, np.int64).ravel()
            out = np.full((i.size, t.shape[1]), np.float32(-7e7))
            rc = L.ref_dev_gather_rows(C.c_void_p(t.ctypes.data), C.c_int64(t.shape[0]), t.shape[1], C.c_void_p(i.ctypes.data), i.size,
                                       C.c_void_p(out.ctypes.data))
            assert rc == 0, rc
            return out

        @staticmethod
        def gather_scatter_rows(table, ids, rows, num_rows, row_stride=1):
            t, i = np.ascontiguousarray(table, np.float32), np.ascontiguousarray(ids, np.int64).ravel()
            r = np.ascontiguousarray(rows, np.int64).ravel()
            assert r.size == i.size * row_stride
            out = np.full((num_rows, t.shape[1]), np.float32(-7e7))
            rc = L.ref_dev_gather_scatter_rows(C.c_void_p(t.ctypes.data), C.c_int64(t.shape[0]), t.shape[1], C.c_void_p(i.ctypes.data),
                                               C.c_void_p(r.ctypes.data), row_stride, i.size, num_rows, C.c_void_p(out.ctypes.data))
            assert rc == 0, rc
            return out

        @staticmethod
        def sparse_segment_reduce(table, ids, seg, num_segments, mean, seg_stride=1):
            t, i = np.ascontiguousarray(table, np.float32), np.ascontiguousarray(ids, np.int64).ravel()
            s = np.ascontiguousarray(seg, np.int64).ravel()
            assert s.size == i.size * seg_stride
            out = np.full((num_segments, t.shape[1]), np.float32(-7e7))
            offs = np.full(num_segments + 1, -12345, np.int32)
            rc = L.ref_dev_sparse_segment_reduce(C.c_void_p(t.ctypes.data), C.c_int64(t.shape[0]), t.shape[1], C.c_void_p(i.ctypes.data),
                                                 C.c_void_p(s.ctypes.data), seg_stride, i.size, num_segments, int(mean),
                                                 C.c_void_p(out.ctypes.data), C.c_void_p(offs.ctypes.data))
            assert rc == 0, rc
            return out, offs

        @staticmethod
        def segment_offsets(seg, num_segments, seg_stride=1):
            s = np.ascontiguousarray(seg, np.int64).ravel()
            offs = np.full(num_segments + 1, -12345, np.int32)
            rc = L.ref_dev_segment_offsets(C.c_void_p(s.ctypes.data), seg_stride, s.size // seg_stride, num_segments,
                                           C.c_void_p(offs.ctypes.data))
            assert rc == 0, rc
            return offs
    return Ref


def _one_column_hip(torch, form, dim, vocab, table, inputs, ranks, esz, rows, combiner=0, seg_kind=0, seg_stride=1):
    """One-column plan through the C ABI (the product path): returns the [rows, dim] output."""
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    from recom_amd.plan import IDS_I64, ROWS_FROM_IDS, ROWS_FROM_SYMBOL, ColumnSpec, PlanSpec
    from recom_amd.plan import FORM_GATHER
    seg_input = 1 if len(inputs) > 1 else -1
    col = ColumnSpec(form, dim, vocab, combiner, IDS_I64, 0, 0, seg_input, seg_kind, seg_stride,
                     ROWS_FROM_IDS if form == FORM_GATHER else ROWS_FROM_SYMBOL, 0, None, 0, 0)
    spec = PlanSpec([col], ranks, esz, 1, n_groups=1, n_symbols=0 if form == FORM_GATHER else 1)
    spec.validate()
    blob, offsets, shapes = concat_inputs(inputs)
    op = FeatureColumnProcess(spec, 0)
    d_blob = torch.from_numpy(blob).cuda() if blob.size else torch.empty(0, dtype=torch.int8, device="cuda")
    out = op(d_blob, offsets, shapes, [torch.from_numpy(np.ascontiguousarray(table)).cuda()],
             None if form == FORM_GATHER else np.asarray([rows], np.int32))
    torch.cuda.synchronize()
    return out.groups[0].cpu().numpy()


@pytest.mark.parametrize("dim", [1, 2, 3, 4, 8, 12, 16, 20, 24, 32, 48, 64, 128])
def test_gather_rows_oracle_and_hip_equal_the_references_kernel(torch_cuda, oracle, ref, dim):
    """a6: GatherRowsToGlbMem behind the generated loop (tiles of 64 ids, ragged last tile) == orc_gather_rows == HIP."""
    from recom_amd.plan import FORM_GATHER
    rng = np.random.default_rng(100 + dim)
    vocab = 777
    table = rng.standard_normal((vocab, dim)).astype(np.float32)
    for n in (1, 63, 64, 65, 128, 500):
        ids = rng.integers(0, vocab, n).astype(np.int64)
        ids[0], ids[-1] = vocab - 1, 0
        want = ref.gather_rows(table, ids)
        got, bad = oracle.gather_rows(table, ids)
        assert bad == 0 and np.array_equal(got, want)
        assert np.array_equal(want, table[ids])                           # and both are the plain definition
        hip = _one_column_hip(torch_cuda, FORM_GATHER, dim, vocab, table, [ids], [1], [8], n)
        assert np.array_equal(hip, want)
    assert ref.gather_rows(table, np.zeros(0, np.int64)).shape == (0, dim)


@pytest.mark.parametrize("dim", [1, 4, 8, 16, 20, 32, 64])
def test_gather_scatter_rows_oracle_and_hip_equal_the_references_kernel(torch_cuda, oracle, ref, dim):
    """a9: zero fill + GatherScatterRows behind the generated loop == orc_gather_scatter_rows == HIP (inverse-map path).
    Rows in ANY order; rows hit twice only in DIFFERENT 64-id tiles, where the reference's order is defined (tiles run one
    after the other, the later write stays — inside one tile its lanes race); SparseTensor-style row ids with stride 2."""
    from recom_amd.plan import FORM_GATHER_SCATTER, SEG_IDS_I64
    rng = np.random.default_rng(200 + dim)
    vocab, B = 1009, 300
    table = rng.standard_normal((vocab, dim)).astype(np.float32)
    for n, stride in ((0, 1), (1, 1), (64, 1), (150, 1), (300, 1), (200, 2)):
        rows = rng.permutation(B)[:n].astype(np.int64)
        ids = rng.integers(0, vocab, n).astype(np.int64)
        if n >= 150:                                                      # duplicates 64 or more positions apart: later tile wins
            rows[70], rows[140] = rows[3], rows[3]
        rows_in = rows if stride == 1 else np.stack([rows, rng.integers(0, 5, n)], axis=1).astype(np.int64)
        want = ref.gather_scatter_rows(table, ids, rows_in, B, row_stride=stride)
        got, bad = oracle.gather_scatter_rows(table, ids, rows, B)
        assert bad == 0 and np.array_equal(got, want)
        hip = _one_column_hip(torch_cuda, FORM_GATHER_SCATTER, dim, vocab, table, [ids, rows_in], [1, 1 if stride == 1 else 2], [8, 8], B,
                              seg_kind=SEG_IDS_I64, seg_stride=stride)
        assert np.array_equal(hip, want)
        untouched = np.setdiff1d(np.arange(B), rows)
        assert not want[untouched].any()                                  # the reference's zero fill


def _bags(rng, B, max_len, empty_tail=0):
    lens = rng.integers(0, max_len + 1, B)
    if empty_tail:
        lens[-empty_tail:] = 0
    lens[0] = 0
    return np.repeat(np.arange(B), lens).astype(np.int64), lens


def test_segment_offsets_oracle_equals_the_references_kernel(oracle, ref):
    """a8, integer work: experiment::ComputeSegmentOffsets (tiles of 64 segment ids, neighbour id per thread, every entry of
    offsets[0..num_segments] written exactly as the serial definition says) == orc_segment_offsets, bit for bit."""
    rng = np.random.default_rng(7)
    for B, max_len, tail in ((1, 3, 0), (33, 10, 0), (256, 10, 5), (100, 0, 0), (64, 1, 0), (500, 40, 17), (7, 300, 2)):
        seg, lens = _bags(rng, B, max_len, tail)
        want = ref.segment_offsets(seg, B)
        assert np.array_equal(oracle.segment_offsets(seg, B), want)
        assert np.array_equal(want, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32))
        idx = np.stack([seg, rng.integers(0, 9, seg.size)], axis=1).astype(np.int64)   # indices[:, 0] of an [nnz, 2] matrix
        assert np.array_equal(ref.segment_offsets(idx, B, seg_stride=2), want)


@pytest.mark.parametrize("dim", [4, 8, 16, 24, 32, 64, 128])
@pytest.mark.parametrize("mean", [False, True])
def test_segment_reduce_in_the_references_order_equals_the_references_kernel(torch_cuda, oracle, ref, dim, mean):
    """a8, floating point: experiment::SparseSegmentReduce (8 `ty` lanes stride a segment's rows, LDS tree 4-2-1,
    mean = sum / float(end - begin)) == orc_sparse_segment_reduce_ref8x8 BIT FOR BIT on non-empty segments; the
    reference's offsets == the oracle's.  Empty MEAN segments: NaN there (0/0), zeros in the oracle / TF / HIP —
    asserted.  The HIP path (sequential id order = the oracle's default order) stays within the north star's 1e-5."""
    from recom_amd.plan import COMBINER_MEAN, COMBINER_SUM, FORM_SEGMENT_REDUCE, SEG_IDS_I64
    rng = np.random.default_rng(300 + dim + int(mean))
    vocab = 2003
    table = (rng.standard_normal((vocab, dim)) * dim ** -0.5).astype(np.float32)
    for B, max_len in ((1, 5), (50, 10), (256, 10), (40, 70), (9, 300)):
        seg, lens = _bags(rng, B, max_len, empty_tail=2 if B > 4 else 0)
        ids = rng.integers(0, vocab, seg.size).astype(np.int64)
        want, offs = ref.sparse_segment_reduce(table, ids, seg, B, mean)
        assert np.array_equal(offs, oracle.segment_offsets(seg, B))
        got, _ = oracle.sparse_segment_reduce(table, ids, offs, mean, ref_order=True)
        empty = lens == 0
        assert np.array_equal(got[~empty], want[~empty])                  # bit-exact, the reference's own order
        if mean:
            assert np.isnan(want[empty]).all() and not got[empty].any()   # documented divergence: 0/0 vs TF's zeros
        else:
            assert not want[empty].any() and not got[empty].any()
        hip = _one_column_hip(torch_cuda, FORM_SEGMENT_REDUCE, dim, vocab, table, [ids, seg], [1, 1], [8, 8], B,
                              combiner=COMBINER_MEAN if mean else COMBINER_SUM, seg_kind=SEG_IDS_I64)
        seq, _ = oracle.sparse_segment_reduce(table, ids, offs, mean)
        assert np.array_equal(hip, seq)                                   # HIP == the oracle's sequential order, exactly
        if max_len <= 10:                                                 # BASELINE's bag lengths: the north star's tolerance
            assert np.abs(hip[~empty] - want[~empty]).max(initial=0) < 1e-5
        else:                                                             # longer bags: fp32 reassociation grows with the bag
            l1 = np.zeros(B)
            np.add.at(l1, seg, np.abs(table[ids]).sum(axis=1) / dim)
            assert (np.abs(hip - np.where(empty[:, None], 0, want)).max(axis=1) <= 1e-6 * np.maximum(l1, 1.0)).all()


@pytest.fixture(scope="module")
def ref_scan(torch_cuda):
    """oracle/_ref/libref_device_scan.so: the reference's dim <= 20 SparseSegmentSum / SparseSegmentMean templates
    (cuda_emitter.cc:348-661) compiled against hipCUB — the image's port of the CUB interface, not CUB 1.8 (absent)."""
    import ref_extract
    path = ref_extract.device_lib_path(2)
    if path is None:
        pytest.skip("oracle/_ref/libref_device_scan.so is absent and /root/reference is not here to build it from")
    L = C.CDLL(path)

    def run(table, ids, seg, num_segments, mean, seg_stride=1):
        t, i = np.ascontiguousarray(table, np.float32), np.ascontiguousarray(ids, np.int64).ravel()
        s = np.ascontiguousarray(seg, np.int64).ravel()
        assert s.size == i.size * seg_stride
        out = np.full((num_segments, t.shape[1]), np.float32(-7e7))
        rc = L.ref_dev_scan_segment_reduce(C.c_void_p(t.ctypes.data), C.c_int64(t.shape[0]), t.shape[1], C.c_void_p(i.ctypes.data),
                                           C.c_void_p(s.ctypes.data), seg_stride, i.size, num_segments, int(mean), C.c_void_p(out.ctypes.data))
        assert rc == 0, rc
        return out
    return run


@pytest.mark.parametrize("dim", [1, 2, 3, 4, 8, 12, 16, 20])
@pytest.mark.parametrize("mean", [False, True])
def test_dim_le_20_templates_of_the_reference_run_against_hipcub(torch_cuda, oracle, ref_scan, dim, mean):
    """a7: the reference's OWN dim <= 20 templates — head / tail flags, segmented scan operator, carry across 64-id tiles,
    which rows are written, mean = sum / integer counter, 8-float slabs + the LEFT_DIM tail — executed on the GPU with
    hipCUB's BlockScan standing where CUB 1.8's was (same interface; the order inside the 64-item scan is rocPRIM's).  Held to it:
      * the oracle's restatement orc_sparse_segment_reduce_refscan_assoc with ONLY the order inside the 64-item scan swapped
        for rocPRIM's one-wavefront order (ORC_SCAN_ROCPRIM64) — BIT FOR BIT, every row, every dim: the restatement's flags,
        operator, carry, write-out and mean are thereby pinned by the reference's own text; what stays restated only is the
        dozen lines of CUB 1.8's order (32-lane Kogge-Stone + first warp's aggregate), which this image cannot run;
      * bags of one or two ids and empty rows — no association involved — bit for bit: the oracle (both orders) and the HIP path;
      * every other bag within the north star's 1e-5 (BASELINE's bag lengths) / a reassociation bound (long bags);
      * tiles: nnz around multiples of 64, a bag crossing three tiles, leading / trailing empty rows, SparseTensor stride 2."""
    from recom_amd.plan import COMBINER_MEAN, COMBINER_SUM, FORM_SEGMENT_REDUCE, SEG_IDS_I64
    rng = np.random.default_rng(700 + 2 * dim + int(mean))
    vocab = 1511
    table = (rng.standard_normal((vocab, dim)) * max(dim, 1) ** -0.5).astype(np.float32)
    cases = [rng.integers(0, 11, 200), np.asarray([0, 0, 1, 2, 63, 64, 65, 130, 0, 1, 0]), np.full(32, 2), np.asarray([0, 128]),
             np.asarray([3]), np.asarray([1] * 64), np.asarray([0, 0, 5, 0, 0])]
    for lens in cases:
        B = len(lens)
        seg = np.repeat(np.arange(B), lens).astype(np.int64)
        ids = rng.integers(0, vocab, seg.size).astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        want = ref_scan(table, ids, seg, B, mean)
        seq, _ = oracle.sparse_segment_reduce(table, ids, offs, mean)
        scan = oracle.sparse_segment_reduce_refscan(table, ids, seg, B, mean)
        assert np.array_equal(want, oracle.sparse_segment_reduce_refscan(table, ids, seg, B, mean, rocprim=True))
        short = np.asarray(lens) <= 2
        assert np.array_equal(want[short], seq[short]) and np.array_equal(want[short], scan[short])   # copies, a + b, zeros: exact
        assert not want[np.asarray(lens) == 0].any()
        l1 = np.zeros(B)
        np.add.at(l1, seg, np.abs(table[ids]).max(axis=1))
        tol = np.where(np.asarray(lens) <= 10, 1e-5, 2e-6 * np.maximum(l1, 1.0))
        assert (np.abs(want - seq).max(axis=1) <= tol).all() and (np.abs(want - scan).max(axis=1) <= tol).all()
        if seg.size:                                                        # SparseTensor indices [nnz, 2]: stride 2
            idx = np.stack([seg, rng.integers(0, 4, seg.size)], axis=1).astype(np.int64)
            assert np.array_equal(ref_scan(table, ids, idx, B, mean, seg_stride=2), want)
        if dim % 4 == 0 or dim in (1, 2, 3):
            hip = _one_column_hip(torch_cuda, FORM_SEGMENT_REDUCE, dim, vocab, table, [ids, seg], [1, 1], [8, 8], B,
                                  combiner=COMBINER_MEAN if mean else COMBINER_SUM, seg_kind=SEG_IDS_I64)
            assert np.array_equal(hip, seq)
            assert np.array_equal(hip[short], want[short]) and (np.abs(hip - want).max(axis=1) <= tol).all()
