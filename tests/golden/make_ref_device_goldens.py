#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE's own device templates (oracle/_ref/libref_device.so and
libref_device_scan.so: the unmodified text of the reference's string literals, compiled for gfx950 by oracle/ref_extract.py,
run on an MI355X).  Inputs are drawn from the seeds stored with every case; outputs are what the reference's kernels wrote.

  GPU box:  python tests/golden/make_ref_device_goldens.py gpurun_out/ref_device_goldens.npz
  then copy the file to tests/golden/ref_device_goldens.npz.

tests/test_oracle.py::test_oracle_equals_the_vectors_the_references_kernels_produced holds the C oracle to them on the CPU.
The scan cases (dim <= 20) were produced against hipCUB: they pin the restatement with ORC_SCAN_ROCPRIM64 (see
oracle/ref_device_scan_wrap.hip for what that does and does not say about CUB 1.8's order)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def inputs_of(case):
    """The inputs of one case from its parameters alone (shared with the CPU test)."""
    kind, seed, dim, vocab, B, max_len = (case[k] for k in ("kind", "seed", "dim", "vocab", "B", "max_len"))
    rng = np.random.default_rng(int(seed))
    table = (rng.standard_normal((int(vocab), int(dim))) * max(int(dim), 1) ** -0.5).astype(np.float32)
    if kind == "gather_rows":
        ids = rng.integers(0, vocab, B).astype(np.int64)
        return dict(table=table, ids=ids)
    if kind == "gather_scatter_rows":
        n = int(max_len)
        rows = rng.permutation(int(B))[:n].astype(np.int64)
        ids = rng.integers(0, vocab, n).astype(np.int64)
        if n >= 150:
            rows[70], rows[140] = rows[3], rows[3]      # duplicates in different 64-id tiles: the later tile wins
        return dict(table=table, ids=ids, rows=rows)
    lens = rng.integers(0, int(max_len) + 1, int(B))
    lens[0] = 0
    if B > 4:
        lens[-2:] = 0
    seg = np.repeat(np.arange(int(B)), lens).astype(np.int64)
    ids = rng.integers(0, vocab, seg.size).astype(np.int64)
    return dict(table=table, ids=ids, seg=seg, lens=lens)


CASES = ([dict(kind="gather_rows", seed=11 + d, dim=d, vocab=777, B=b, max_len=0, mean=0) for d in (1, 4, 20, 64) for b in (1, 65, 200)] +
         [dict(kind="gather_scatter_rows", seed=21 + d, dim=d, vocab=1009, B=300, max_len=n, mean=0) for d in (4, 32) for n in (0, 64, 200)] +
         [dict(kind="segment_offsets", seed=31 + b, dim=1, vocab=5, B=b, max_len=m, mean=0) for b, m in ((1, 3), (33, 10), (256, 10), (64, 1), (500, 40))] +
         [dict(kind="segment_reduce_8x8", seed=41 + d + mean, dim=d, vocab=2003, B=b, max_len=m, mean=mean)
          for d in (24, 32, 64) for mean in (0, 1) for b, m in ((50, 10), (40, 70))] +
         [dict(kind="segment_reduce_scan", seed=51 + d + mean, dim=d, vocab=1511, B=b, max_len=m, mean=mean)
          for d in (1, 4, 8, 12, 20) for mean in (0, 1) for b, m in ((60, 10), (12, 90))])


def main(out_path):
    import ref_extract
    L = C.CDLL(ref_extract.device_lib_path(1))
    S = C.CDLL(ref_extract.device_lib_path(2))
    P = C.c_void_p
    store = {"n_cases": np.asarray(len(CASES))}
    for k, case in enumerate(CASES):
        x = inputs_of(case)
        t = np.ascontiguousarray(x["table"])
        vocab, dim, B = t.shape[0], t.shape[1], int(case["B"])
        if case["kind"] == "gather_rows":
            out = np.full((x["ids"].size, dim), np.float32(-7e7))
            assert L.ref_dev_gather_rows(P(t.ctypes.data), C.c_int64(vocab), dim, P(x["ids"].ctypes.data), x["ids"].size, P(out.ctypes.data)) == 0
        elif case["kind"] == "gather_scatter_rows":
            out = np.full((B, dim), np.float32(-7e7))
            assert L.ref_dev_gather_scatter_rows(P(t.ctypes.data), C.c_int64(vocab), dim, P(x["ids"].ctypes.data), P(x["rows"].ctypes.data), 1,
                                                 x["ids"].size, B, P(out.ctypes.data)) == 0
        elif case["kind"] == "segment_offsets":
            out = np.full(B + 1, -12345, np.int32)
            assert L.ref_dev_segment_offsets(P(x["seg"].ctypes.data), 1, x["seg"].size, B, P(out.ctypes.data)) == 0
        elif case["kind"] == "segment_reduce_8x8":
            out = np.full((B, dim), np.float32(-7e7))
            offs = np.full(B + 1, -12345, np.int32)
            assert L.ref_dev_sparse_segment_reduce(P(t.ctypes.data), C.c_int64(vocab), dim, P(x["ids"].ctypes.data), P(x["seg"].ctypes.data), 1,
                                                   x["ids"].size, B, int(case["mean"]), P(out.ctypes.data), P(offs.ctypes.data)) == 0
            store[f"offs_{k}"] = offs
        else:
            out = np.full((B, dim), np.float32(-7e7))
            assert S.ref_dev_scan_segment_reduce(P(t.ctypes.data), C.c_int64(vocab), dim, P(x["ids"].ctypes.data), P(x["seg"].ctypes.data), 1,
                                                 x["ids"].size, B, int(case["mean"]), P(out.ctypes.data)) == 0
        store[f"out_{k}"] = out
        store[f"case_{k}"] = np.asarray([case["kind"]] + [str(case[f]) for f in ("seed", "dim", "vocab", "B", "max_len", "mean")])
    np.savez_compressed(out_path, **store)
    print(f"{len(CASES)} cases -> {out_path} ({os.path.getsize(out_path)} bytes)")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref_device_goldens.npz"))
