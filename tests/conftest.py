import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure)."""
    import fcp_oracle
    return fcp_oracle.COracle()


@pytest.fixture(scope="session")
def ref_bucketize():
    """The REFERENCE's own `Bucketize` (cuda_emitter.cc:233-247), compiled from its source where it lies by the recipe
    oracle/ref_extract.py into oracle/_ref/libref_bucketize.so (built in the container that has /root/reference; the built
    library travels to the GPU box).  Returns f(boundaries, values) -> int32 bucket ids."""
    import ctypes as C
    import ref_extract
    if not ref_extract.build():
        pytest.skip("oracle/_ref/libref_bucketize.so is absent and /root/reference is not here to build it from")
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bucketize.so"))
    lib.ref_bucketize_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ref_bucketize_max_boundaries.restype = C.c_int

    def run(boundaries, values):
        b = np.ascontiguousarray(boundaries, np.float32)
        v = np.ascontiguousarray(values, np.float32).ravel()
        assert 1 <= b.size <= lib.ref_bucketize_max_boundaries()
        out = np.empty(v.size, np.int32)
        lib.ref_bucketize_many(b.ctypes.data, int(b.size), v.ctypes.data, int(v.size), out.ctypes.data)
        return out
    return run


@pytest.fixture(scope="session")
def ref_alignmem():
    """The REFERENCE's own `alignmem` (the arena alignment of its generated host code, cuda_emitter.cc:967-969, used by
    :2151-2179), compiled from its source by oracle/ref_extract.py next to `Bucketize`.  Returns f(bytes) -> bytes."""
    import ctypes as C
    import ref_extract
    if not ref_extract.build():
        pytest.skip("oracle/_ref/libref_bucketize.so is absent and /root/reference is not here to build it from")
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_bucketize.so"))
    if not hasattr(lib, "ref_alignmem"):
        pytest.skip("oracle/_ref was built before ref_alignmem existed")
    lib.ref_alignmem.argtypes = [C.c_int]
    lib.ref_alignmem.restype = C.c_int
    return lambda x: int(lib.ref_alignmem(int(x)))


class GoldenCase:
    def __init__(self, z, name):
        self.name = name
        self.plan = json.loads(bytes(z[f"{name}/plan"]).decode())
        for c in self.plan["columns"]:
            if c["boundaries"] is not None:
                c["boundaries"] = np.asarray(c["boundaries"], np.float32)
        self.inputs = [z[f"{name}/in{i}"] for i in range(int(z[f"{name}/n_inputs"]))]
        self.blob = z[f"{name}/blob"]
        self.offsets = z[f"{name}/offsets"]
        self.shapes = z[f"{name}/shapes"]
        sym = z[f"{name}/symbols"]
        self.symbols = sym if sym.size else None
        tk = bytes(z[f"{name}/tables_key"]).decode()
        self.tables = [z[f"{tk}/table{i}"] for i in range(int(z[f"{tk}/n_tables"]))]
        self.expected = [z[f"{name}/expected{g}"] for g in range(self.plan["n_groups"])]
        self.copy_cols = z[f"{name}/copy_cols"]

    def spec(self):
        """PlanSpec rebuilt from the stored plan description."""
        from recom_amd.plan import ColumnSpec, PlanSpec
        d = dict(self.plan)
        cols = [ColumnSpec(**c) for c in d.pop("columns")]
        return PlanSpec(columns=cols, **d)


GOLDEN_NAMES = ["mixed_s0", "mixed_s1", "mixed_empty", "bucketize_kat", "ragged_edges", "scatter"]


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "fcp_golden.npz"))
    return {n: GoldenCase(z, n) for n in GOLDEN_NAMES}, z


def check_against_expected(case, outs, pooled_atol=1e-5):
    """Bit-exact on copy-only columns, max-abs-diff < pooled_atol (the north
    star's fp32 tolerance) on pooled ones."""
    for g, (o, e) in enumerate(zip(outs, case.expected)):
        o = np.asarray(o)
        assert o.shape == e.shape, (case.name, g, o.shape, e.shape)
        assert np.abs(o.astype(np.float64) - e).max(initial=0.0) < pooled_atol, (case.name, g)
    for g, off, dim in case.copy_cols:
        got = np.asarray(outs[g])[:, off:off + dim]
        want = case.expected[g][:, off:off + dim].astype(np.float32)
        assert np.array_equal(got, want), (case.name, "copy column not bit-exact", g, off)


def assert_staged_blob(blob, offsets, shapes, conv, modes):
    """The staged blob of Addons>ConcatInputs (fcp_concat_inputs_ex / fcp_stager_stage_ex; layout: stage_layout in
    recom_amd/csrc/fcp_stager.hip) against the converted tensors `conv` (NumPy): copied and narrowed inputs back to back in
    input order from offset 0 — the reference op's layout, concat_inputs_ops.cc:52-66 — and the inputs converted to row
    offsets (mode 2) behind ALL of them, in input order, from a 4-byte boundary: one [columns, rows + 1] matrix.  Shapes
    stay in input order.  Returns the blob size."""
    from recom_amd.ops import concat_inputs
    blob = np.asarray(blob).view(np.int8).reshape(-1)
    _, _, want_shapes = concat_inputs(conv)
    assert np.array_equal(np.asarray(shapes), want_shapes)
    size = 0
    order = [i for i, m in enumerate(modes) if m != 2] + [i for i, m in enumerate(modes) if m == 2]
    n_first = sum(1 for m in modes if m != 2)
    for pos, i in enumerate(order):
        if pos == n_first:
            size = (size + 3) & ~3
        raw = np.ascontiguousarray(conv[i]).view(np.int8).reshape(-1)
        assert int(offsets[i]) == size, (i, int(offsets[i]), size)
        assert np.array_equal(blob[size:size + raw.size], raw), f"input {i} differs in the staged blob"
        size += raw.size
    assert blob.size >= size
    return size
