"""How often does the reference's dim <= 20 template compiled against hipCUB agree BIT FOR BIT with the oracle's restatement of
CUB 1.8's association (orc_sparse_segment_reduce_refscan) and with the sequential order?  (probe; prints counts)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fcp_oracle, ref_extract
oracle = fcp_oracle.COracle()
L = C.CDLL(ref_extract.device_lib_path(2))
rng = np.random.default_rng(5)
for dim in (1, 4, 8, 12, 16, 20):
    for mean in (0, 1):
        tot = eq_scan = eq_seq = eq_roc = 0
        for trial in range(20):
            lens = rng.integers(0, 40, 300) if trial % 2 else rng.integers(0, 11, 300)
            B = len(lens); seg = np.repeat(np.arange(B), lens).astype(np.int64)
            table = rng.standard_normal((997, dim)).astype(np.float32)
            ids = rng.integers(0, 997, seg.size).astype(np.int64)
            out = np.zeros((B, dim), np.float32)
            assert L.ref_dev_scan_segment_reduce(C.c_void_p(table.ctypes.data), C.c_int64(997), dim, C.c_void_p(ids.ctypes.data),
                                                 C.c_void_p(seg.ctypes.data), 1, ids.size, B, mean, C.c_void_p(out.ctypes.data)) == 0
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            seq, _ = oracle.sparse_segment_reduce(table, ids, offs, bool(mean))
            scan = oracle.sparse_segment_reduce_refscan(table, ids, seg, B, bool(mean))
            roc = oracle.sparse_segment_reduce_refscan(table, ids, seg, B, bool(mean), rocprim=True)
            eq_roc += int((out == roc).all(axis=1).sum())
            tot += B; eq_scan += int((out == scan).all(axis=1).sum()); eq_seq += int((out == seq).all(axis=1).sum())
        print(f"dim {dim:2d} mean {mean}: rows {tot}  == refscan restatement {eq_scan}  == sequential {eq_seq}  == restatement with rocPRIM scan order {eq_roc}", flush=True)
