"""CPU-only tests: the C-ABI library loads and exports every symbol the header
declares, the host-side pieces (ConcatInputs packer, plan validation, layout
arithmetic, error codes) behave like the reference's ops, and the synthetic
model generator is self-consistent.  No compute call needs a GPU here."""
import ctypes as C
import dataclasses
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from recom_amd import lib
    return lib.load()


def test_library_exports_every_declared_symbol(L):
    from recom_amd import lib
    hdr = open(os.path.join(ROOT, "include", "fcp_hip.h")).read()
    declared = set(re.findall(r"\b(fcp_[a-z_0-9]+)\s*\(", hdr)) - {"fcp_alloc_fn"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.fcp_abi_version() == lib.FCP_ABI_VERSION
    assert L.fcp_status_string(0) == b"ok"
    assert b"shape" in L.fcp_status_string(lib.FCP_ERR_SHAPE_MISMATCH)


def test_no_torch_or_tf_types_in_the_abi():
    hdr = open(os.path.join(ROOT, "include", "fcp_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for banned in ("std::", "torch", "at::", "tensorflow", "hipStream_t", "#include <hip"):
        assert banned not in code, banned


def test_concat_inputs_matches_reference_semantics(oracle):
    """Addons>ConcatInputs (concat_inputs_ops.cc:42-77): byte concatenation, int32
    byte offsets, all dims in order — bit-exact with the oracle restatement."""
    from recom_amd.ops import concat_inputs
    rng = np.random.default_rng(0)
    ts = [rng.integers(0, 9, (5,)).astype(np.int64), rng.standard_normal((3, 2)).astype(np.float32),
          np.zeros((0, 2), np.int64), np.asarray(7, np.int32), rng.integers(0, 100, (4, 2)).astype(np.int64),
          np.frombuffer(b"abc", np.uint8)]
    blob, off, shp = concat_inputs(ts)
    b2, o2, s2 = oracle.concat_inputs(ts)
    assert blob.dtype == np.int8 and off.dtype == np.int32 and shp.dtype == np.int32
    assert np.array_equal(blob, b2) and np.array_equal(off, o2) and np.array_equal(shp, s2)
    assert np.array_equal(off, [0, 40, 64, 64, 68, 132])
    assert np.array_equal(shp, [5, 3, 2, 0, 2, 4, 2, 3])
    assert blob.tobytes() == b"".join(t.tobytes() for t in ts)
    e_blob, e_off, e_shp = concat_inputs([])
    assert e_blob.size == 0 and e_off.size == 0 and e_shp.size == 0


def test_concat_inputs_error_codes(L):
    from recom_amd import lib
    t = lib.HostTensor(None, 4, 1, (C.c_int64 * 1)(-3))
    n = C.c_int64()
    assert L.fcp_concat_inputs_sizes(C.byref(t), 1, C.byref(n), None) == lib.FCP_ERR_INVALID_ARGUMENT
    dims = (C.c_int64 * 1)(4)
    data = np.arange(4, dtype=np.float32)
    t = lib.HostTensor(data.ctypes.data, 4, 1, dims)
    off, shp = np.zeros(1, np.int32), np.zeros(1, np.int32)
    small = np.zeros(8, np.int8)
    assert L.fcp_concat_inputs(C.byref(t), 1, small.ctypes.data, 8, off.ctypes.data, shp.ctypes.data) == \
        lib.FCP_ERR_INVALID_ARGUMENT  # blob too small
    assert b"blob" in L.fcp_last_error()


def _host_plan(spec):
    from recom_amd.ops import Plan
    return Plan(spec, 0, host_only=True)


def test_plan_layout_queries_host_only():
    from recom_amd import synth
    from recom_amd.plan import LAYOUT_PER_COLUMN
    m = synth.model_mixed(batch=33, vocab=211)
    p = _host_plan(m.spec)
    offs = m.spec.column_offsets()
    for k in range(m.spec.n_columns):
        assert p.column_offset(k) == offs[k]
    for g in range(m.spec.n_groups):
        assert p.group_width(g) == m.spec.group_width(g)
    # arena = sum over groups of alignmem(rows*width*4) + CSR scratch for seg-id columns
    req = m.make_request(0)
    from recom_amd.ops import concat_inputs
    _, _, shapes = concat_inputs(req.inputs)
    al = lambda x: (x + 127) // 128 * 128  # alignmem, cuda_emitter.cc:967-969
    rows = 33
    out_bytes = sum(al(rows * m.spec.group_width(g) * 4) for g in range(m.spec.n_groups))
    n_seg = sum(1 for c in m.spec.columns if c.form in (2, 3) and c.seg_kind in (1, 2))
    csr = n_seg * ((rows + 1 + 31) // 32 * 32) * 4
    assert p.arena_bytes(shapes, req.symbols) == out_bytes + csr
    # the reference arena: one 128-byte aligned buffer per column (cuda_emitter.cc:2151-2179)
    pc = _host_plan(m.spec.with_layout(LAYOUT_PER_COLUMN))
    assert pc.arena_bytes(shapes, req.symbols) == sum(al(rows * c.dim * 4) for c in m.spec.columns) + csr
    p.close()
    pc.close()


def test_host_only_plan_cannot_run_and_real_plan_needs_a_gpu():
    import torch
    from recom_amd import lib, synth
    from recom_amd.ops import Plan
    m = synth.model_s1(columns=4, batch=8)
    p = _host_plan(m.spec)
    a = lib.ProcessArgs()
    assert p._L.fcp_process_feature_columns(p.handle, C.byref(a), None) == lib.FCP_ERR_NO_DEVICE
    # the serving-mode entry points refuse a plan without a device the same way, and check their arguments first
    assert p._L.fcp_plan_set_private_streams(p.handle, 3, 0) == lib.FCP_ERR_NO_DEVICE
    assert p._L.fcp_plan_set_private_streams(p.handle, 17, 0) == lib.FCP_ERR_INVALID_ARGUMENT
    assert p._L.fcp_plan_set_private_streams(p.handle, 3, 1 << 3) == lib.FCP_ERR_INVALID_ARGUMENT     # unknown flag
    sa, sb = C.c_double(), C.c_double()
    assert p._L.fcp_plan_probe_private_streams(p.handle, None, 24, 20, 1, C.byref(sa), C.byref(sb)) == lib.FCP_ERR_NO_DEVICE
    assert p._L.fcp_plan_probe_private_streams(p.handle, None, 0, 20, 1, C.byref(sa), C.byref(sb)) == lib.FCP_ERR_INVALID_ARGUMENT
    assert p._L.fcp_plan_probe_private_streams(None, None, 24, 20, 1, C.byref(sa), C.byref(sb)) == lib.FCP_ERR_INVALID_ARGUMENT
    assert p._L.fcp_result_wait(None, None) == lib.FCP_ERR_INVALID_ARGUMENT
    v = C.c_int32(5)
    assert p._L.fcp_plan_private_streams_verdict(p.handle, None, C.byref(v)) == lib.FCP_OK and v.value == -1   # mode off
    assert p._L.fcp_plan_private_streams_verdict(p.handle, None, None) == lib.FCP_ERR_INVALID_ARGUMENT
    v = C.c_int32(5)
    assert p._L.fcp_plan_verify_private_streams(p.handle, None, 50, C.byref(v)) == lib.FCP_ERR_NO_DEVICE and v.value == -1
    assert p._L.fcp_plan_verify_private_streams(None, None, 50, None) == lib.FCP_ERR_INVALID_ARGUMENT
    st = lib.PrivateStreamsStats()
    assert p._L.fcp_plan_private_streams_stats(p.handle, C.byref(st)) == lib.FCP_OK and st.lane_requests == 0 and st.demoted == 0 and st.requests == 0
    assert p._L.fcp_plan_private_streams_stats(p.handle, None) == lib.FCP_ERR_INVALID_ARGUMENT
    p.close()
    if not torch.cuda.is_available():
        with pytest.raises(lib.FcpError) as e:
            Plan(m.spec, 0)  # no silent CPU fallback: creation fails loudly
        assert e.value.status in (lib.FCP_ERR_NO_DEVICE, lib.FCP_ERR_HIP)


@pytest.mark.parametrize("mutate,needle", [
    (lambda s: dataclasses.replace(s, n_groups=0), "n_groups"),
    (lambda s: dataclasses.replace(s, n_groups=17), "n_groups"),
    (lambda s: dataclasses.replace(s, shard_rank=2, shard_world=2), "shard"),
    (lambda s: dataclasses.replace(s, layout=7), "layout"),
])
def test_plan_descriptor_validation(mutate, needle):
    """Bad attrs are refused like OP_REQUIRES in the op constructors
    (feature_column_process_op_gpu.cu.cc:39-44), with FCP_ERR_INVALID_ARGUMENT."""
    from recom_amd import lib, synth
    from recom_amd.ops import Plan
    m = synth.model_s1(columns=4, batch=8)
    spec = mutate(m.spec)
    spec.validate = lambda: None  # bypass the Python-side check: the C side must catch it
    with pytest.raises(lib.FcpError) as e:
        Plan(spec, 0, host_only=True)
    assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT and needle in str(e.value)


def test_column_descriptor_validation():
    from recom_amd import lib, synth
    from recom_amd.ops import Plan
    base = synth.model_mixed(batch=8, vocab=50, n_groups=1).spec

    def broken(k, **kw):
        cols = list(base.columns)
        cols[k] = dataclasses.replace(cols[k], **kw)
        s = dataclasses.replace(base, columns=cols)
        s.validate = lambda: None
        for c in cols:
            c.validate = lambda: None
        return s

    cases = [broken(0, dim=0), broken(0, vocab=0), broken(0, table_input=99), broken(0, ids_input=-1),
             broken(1, boundaries=None), broken(3, seg_kind=0), broken(3, rows_source=0), broken(3, combiner=0),
             broken(0, concat_slot=1), broken(0, id_source=0), broken(4, seg_stride=0)]
    for s in cases:
        with pytest.raises(lib.FcpError) as e:
            Plan(s, 0, host_only=True)
        assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT
    Plan(base, 0, host_only=True).close()


def test_runtime_shape_checks_host_only():
    """Run-time shape errors surface as FCP_ERR_SHAPE_MISMATCH from the layout
    query (the same code path the request entry uses)."""
    from recom_amd import lib, synth
    from recom_amd.ops import concat_inputs
    m = synth.model_mixed(batch=16, vocab=97, n_groups=1)
    p = _host_plan(m.spec)
    req = m.make_request(0)
    _, _, shapes = concat_inputs(req.inputs)
    assert p.arena_bytes(shapes, req.symbols) > 0
    with pytest.raises(lib.FcpError) as e:
        p.arena_bytes(shapes, np.asarray([17], np.int32))
    assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH
    with pytest.raises(lib.FcpError) as e:
        p.arena_bytes(shapes, None)
    assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT
    bad = shapes.copy()
    bad[0] = -1
    with pytest.raises(lib.FcpError) as e:
        p.arena_bytes(bad, req.symbols)
    assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH
    p.close()


def test_s2_algorithmic_bytes_match_the_survey():
    """rows 61.44 MB + ids + out 61.44 MB for S2 (ids are 4 bytes on
    the 100 bucketize-sourced columns, 8 bytes elsewhere)."""
    from recom_amd import synth
    from recom_amd.ops import concat_inputs
    m = synth.model_s2()
    assert m.spec.n_columns == 1000 and m.spec.group_width(0) == 30000
    assert m.table_bytes() == 120_000_000_000
    req = m.make_request(0)
    _, _, shapes = concat_inputs(req.inputs)
    b = m.spec.algorithmic_bytes(shapes, req.symbols)
    assert b["rows"] == 512 * 30000 * 4 == 61_440_000
    assert b["out"] == 61_440_000
    assert b["ids"] == 512 * (900 * 8 + 100 * 4)
    assert b["boundaries"] == 100 * 100 * 4
    assert b["total"] == b["rows"] + b["ids"] + b["boundaries"] + b["out"]


def test_baseline_configs_are_well_formed():
    from recom_amd import synth
    s1 = synth.model_s1()
    assert (s1.spec.n_columns, s1.batch, s1.spec.group_width(0)) == (100, 128, 1600)
    d = synth.model_dlrm()
    assert d.spec.n_columns == 27 and d.spec.group_width(0) == 26 * 16 + 13 and d.batch == 2048
    r = synth.model_ragged()
    assert r.spec.n_columns == 512 and r.batch == 256
    req = r.make_request(0)
    lens = np.diff(req.inputs[1])
    assert lens.min() == 0 and lens.max() == 10  # ids/row ~ U{0..10}
    assert any(not np.array_equal(r.make_request(1).inputs[1], req.inputs[1]) for _ in range(1))  # nnz re-drawn
    sh = synth.model_shard()
    assert sh.spec.n_columns == 4000 and sh.table_bytes() == 480_000_000_000
    for m in (s1, d, r):
        m.spec.validate()


def test_closed_form_tables_agree_between_numpy_and_torch():
    import torch
    from recom_amd import synth
    a = synth.hash_table_numpy(1003, 257, 12)
    b = synth.hash_table_torch(1003, 257, 12, "cpu").numpy()
    assert np.array_equal(a, b)
    assert np.array_equal(synth.hash_rows(1003, [0, 5, 256], 12), a[[0, 5, 256]])
    # one shard of a row-sharded table
    c = synth.hash_table_torch(1003, 257, 12, "cpu", 2, 4).numpy()
    assert np.array_equal(c, a[2::4])
    assert a.min() >= -1.0 and a.max() < 1.0 and abs(float(a.mean())) < 0.05


def test_plan_file_round_trip(tmp_path):
    """The `dlpath` plan file read by the TF shim (tf_shim/fcp_tf_ops.cc) and by Python."""
    from recom_amd import synth
    from recom_amd.plan_io import load_plan, save_plan
    m = synth.model_mixed(batch=9, vocab=50)
    path = str(tmp_path / "plan.fcp")
    save_plan(m.spec, path)
    back = load_plan(path)
    a, b = m.spec.to_dict(), back.to_dict()
    for ca, cb in zip(a.pop("columns"), b.pop("columns")):
        ba, bb = ca.pop("boundaries"), cb.pop("boundaries")
        assert ca == cb
        assert (ba is None and bb is None) or np.array_equal(ba, bb)
    assert a == b


def test_narrowed_plan_and_column_subset():
    """Host-side plan transforms: `narrowed()` (int64 ids shipped as int32 by the stager)
    and `column_subset()` (column-sharded serving)."""
    from recom_amd import synth
    from recom_amd import plan as PL
    m = synth.model_mixed(batch=9, vocab=97, n_groups=2)
    n, flags = m.spec.narrowed()
    n.validate()
    for c0, c1 in zip(m.spec.columns, n.columns):
        if c0.form in (1, 2, 3) and c0.id_source == PL.IDS_I64:
            assert c1.id_source == PL.IDS_I32 and flags[c0.ids_input] and n.host_input_elem_sizes[c0.ids_input] == 4
        if c0.seg_kind == PL.SEG_IDS_I64:
            assert c1.seg_kind == PL.SEG_IDS_I32 and c1.seg_stride == c0.seg_stride and flags[c0.seg_input]
        if c0.id_source == PL.IDS_F32_BUCKETIZE:
            assert c1.id_source == c0.id_source
        if c0.seg_kind == PL.SEG_CSR_I32:
            assert c1.seg_kind == c0.seg_kind
    assert not any(f and e != 8 for f, e in zip(flags, m.spec.host_input_elem_sizes))
    # a vocabulary beyond int32 keeps 8-byte ids
    big = synth.model_s2(columns=4, vocab=1 << 33)
    nb, fb = big.spec.narrowed()
    assert not any(fb) and nb.host_input_elem_sizes == big.spec.host_input_elem_sizes
    # ids that are hashed or interval-transformed on the device keep their 8 bytes (raw ids of any magnitude)
    import dataclasses
    k64 = [k for k, c in enumerate(m.spec.columns) if c.form in (1, 2, 3) and c.id_source == PL.IDS_I64][:2]
    cols = list(m.spec.columns)
    cols[k64[0]] = dataclasses.replace(cols[k64[0]], hash_buckets=97)
    cols[k64[1]] = dataclasses.replace(cols[k64[1]], xform_mode=PL.XFORM_SELECT, xform_lo=(-5,), xform_hi=(1 << 40,), xform_substitute=0)
    hx, fx = dataclasses.replace(m.spec, columns=cols).narrowed()
    for k in k64:
        assert hx.columns[k].id_source == PL.IDS_I64 and not fx[cols[k].ids_input] and hx.host_input_elem_sizes[cols[k].ids_input] == 8
    # staged(): narrowing + sorted row ids -> CSR offsets on the host (fcp_stager_stage_ex)
    sp, modes, rows_col = m.spec.staged()
    for c0, c1 in zip(m.spec.columns, sp.columns):
        if c0.form == 3 and c0.seg_kind in (PL.SEG_IDS_I32, PL.SEG_IDS_I64):
            # ScatterNd row ids come in any order: narrowed at most, never turned into offsets
            assert modes[c0.seg_input] != PL.STAGE_SEG_TO_CSR and c1.seg_kind in (PL.SEG_IDS_I32, PL.SEG_IDS_I64)
        if c0.form == 2 and c0.seg_kind in (PL.SEG_IDS_I32, PL.SEG_IDS_I64) and c0.rows_source == PL.ROWS_FROM_SYMBOL:
            assert modes[c0.seg_input] == PL.STAGE_SEG_TO_CSR and c1.seg_kind == PL.SEG_CSR_I32 and c1.seg_stride == 1
            assert sp.host_input_ranks[c0.seg_input] == 1 and sp.host_input_elem_sizes[c0.seg_input] == 4
            assert m.spec.columns[rows_col[c0.seg_input]].rows_arg == c0.rows_arg
        if c0.seg_kind == PL.SEG_CSR_I32:
            assert modes[c0.seg_input] == PL.STAGE_COPY and c1.seg_kind == PL.SEG_CSR_I32
    assert PL.STAGE_SEG_TO_CSR in modes and all((r >= 0) == (mo == PL.STAGE_SEG_TO_CSR) for r, mo in zip(rows_col, modes))
    # column subset: renumbered operands, same concat slots
    keep = [k for k, c in enumerate(m.spec.columns) if c.concat_group == 0][2:6]
    sub = m.spec.column_subset(keep)
    sub.spec.validate()
    assert sub.columns == keep and sub.spec.n_columns == 4
    for k, c in zip(keep, sub.spec.columns):
        o = m.spec.columns[k]
        assert c.dim == o.dim and c.concat_slot == o.concat_slot
        assert sub.host_inputs[c.ids_input] == o.ids_input
        assert o.table_input < 0 or sub.device_inputs[c.table_input] == o.table_input


def test_pack_pool_under_thread_sanitizer(tmp_path):
    """The stager's worker pool (recom_amd/csrc/pack_pool.h) hammered with short
    back-to-back jobs under ThreadSanitizer (CPU build; GPU sanitizers are unavailable)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "pack_pool_stress")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread",
                            "-I", os.path.join(ROOT, "recom_amd", "csrc"),
                            os.path.join(ROOT, "tests", "native", "pack_pool_stress.cc"), "-o", exe, "-lpthread"],
                           capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower():
        pytest.skip("libtsan not installed")
    assert build.returncode == 0, build.stderr
    for threads, jobs in ((4, 6000), (7, 3000)):
        run = subprocess.run([exe, str(threads), str(jobs)], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr, run.stdout + run.stderr[-3000:]


def test_plan_file_loaded_by_the_library(tmp_path):
    """fcp_plan_create_from_file (what the TF shim calls with `dlpath`) parses what
    plan_io.save_plan / the graph front end write: same layout facts as a plan created
    from the descriptor; malformed files are rejected with a status, not a crash."""
    from recom_amd import lib, synth
    from recom_amd.ops import Plan, concat_inputs
    from recom_amd.plan_io import save_plan
    m = synth.model_mixed(batch=21, vocab=97, n_groups=2)
    path = str(tmp_path / "m.fcp")
    save_plan(m.spec, path)
    a, b = Plan(m.spec, host_only=True), Plan.from_file(path, host_only=True)
    assert b.counts() == {"columns": m.spec.n_columns, "groups": 2, "host_inputs": m.spec.n_host_inputs,
                          "device_inputs": m.spec.n_device_inputs, "symbols": m.spec.n_symbols}
    assert [a.group_width(g) for g in range(2)] == [b.group_width(g) for g in range(2)]
    assert [a.column_offset(k) for k in range(m.spec.n_columns)] == [b.column_offset(k) for k in range(m.spec.n_columns)]
    req = m.make_request(0)
    _, _, shapes = concat_inputs(req.inputs)
    assert a.arena_bytes(shapes, req.symbols) == b.arena_bytes(shapes, req.symbols)
    text = open(path).read()
    for bad in ("", "fcp_plan 2\n", text[: len(text) // 2], text.replace("columns", "colums")):
        open(path, "w").write(bad)
        with pytest.raises(lib.FcpError) as e:
            Plan.from_file(path, host_only=True)
        assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT
    with pytest.raises(lib.FcpError):
        Plan.from_file(str(tmp_path / "missing.fcp"), host_only=True)


def test_header_is_plain_c_and_a_c_client_drives_the_boundary(tmp_path):
    """include/fcp_hip.h compiled as strict C99 (-pedantic -Werror) into a client that drives the boundary the
    way a cgo / JNI / dlsym binding would: host-only plan, ConcatInputs packing, layout / arena / table-byte
    queries, the placement gate, and the loud failure of a compute call without a device
    (tests/native/abi_c_client.c)."""
    import shutil
    import subprocess
    from recom_amd import lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib.load()                                        # builds libfcp_hip.so when it is missing
    exe = str(tmp_path / "abi_c_client")
    libdir = os.path.join(ROOT, "recom_amd")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "native", "abi_c_client.c"), "-L", libdir, "-lfcp_hip",
                            "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and "abi_c_client ok" in run.stdout, run.stdout + run.stderr


def test_tf_shim_parses_against_a_mock_of_the_tf_api():
    """recom_amd/tf_shim/fcp_tf_ops.cc cannot be built here (no TensorFlow); at least it must
    parse and type-check against include/fcp_hip.h and a minimal mock of the TF op-kernel API."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "native", "tf_mock"),
                        os.path.join(ROOT, "recom_amd", "tf_shim", "fcp_tf_ops.cc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_fast_and_general_shape_evaluation_agree(monkeypatch):
    """compute_dyn_fast (one pass, concat-layout plans) and the general routine give the same
    arena size for every synthetic model, and the same error for an inconsistent request."""
    from recom_amd import lib, synth
    from recom_amd.ops import Plan, concat_inputs
    models = [synth.model_mixed(batch=21, vocab=97, n_groups=2), synth.model_s1(), synth.model_ragged(columns=40, batch=17),
              synth.model_ragged(columns=9, batch=5, seg="indices"), synth.model_dlrm(batch=33), synth.model_ae("E", batch=19)]
    for m in models:
        p = Plan(m.spec, host_only=True)
        for seed in range(3):
            req = m.make_request(seed)
            _, _, shapes = concat_inputs(req.inputs)
            monkeypatch.delenv("FCP_DIAG", raising=False)
            fast = p.arena_bytes(shapes, req.symbols)
            monkeypatch.setenv("FCP_DIAG", "dyn_general")
            assert p.arena_bytes(shapes, req.symbols) == fast > 0
        monkeypatch.delenv("FCP_DIAG", raising=False)
        gather = [c for c in m.spec.columns if c.form == 1]
        if gather:                                    # a one-hot column one element longer: row counts disagree
            bad = np.array(shapes, np.int32)
            bad[m.spec.shape_offsets()[gather[0].ids_input]] += 1
            with pytest.raises(lib.FcpError) as e:
                p.arena_bytes(bad, req.symbols)
            assert e.value.status == lib.FCP_ERR_SHAPE_MISMATCH


def test_external_slots_host_side(monkeypatch):
    """FCP_FORM_EXTERNAL (Addons>ConcatOutputs host inputs): reserved in the concat layout, not an output
    of FeatureColumnProcess, rows from the group; both shape-evaluation routines agree; bad plans refused."""
    import ctypes as C
    import dataclasses
    from recom_amd import lib, synth
    from recom_amd.ops import Plan, concat_inputs
    from recom_amd.plan import FORM_EXTERNAL, LAYOUT_PER_COLUMN, ROWS_FROM_GROUP, ROWS_FROM_IDS
    m = synth.model_mixed(batch=21, vocab=97, n_groups=2)
    ext = dataclasses.replace(m.spec.columns[0], form=FORM_EXTERNAL, dim=20, vocab=0, table_input=-1, ids_input=-1,
                              id_source=0, rows_source=ROWS_FROM_GROUP, rows_arg=0, concat_group=1, concat_slot=99)
    spec = dataclasses.replace(m.spec, columns=m.spec.columns + [ext])
    spec.validate()
    p = Plan(spec, host_only=True)
    L = lib.load()
    n = C.c_int32()
    idx = (C.c_int32 * spec.n_columns)()
    lib.check(L.fcp_plan_output_columns(p.handle, C.byref(n), idx, spec.n_columns), "fcp_plan_output_columns")
    assert list(idx[:n.value]) == spec.output_columns() == list(range(spec.n_columns - 1))
    assert p.group_width(1) == m.spec.group_width(1) + 20 and p.column_offset(spec.n_columns - 1) == m.spec.group_width(1)
    req = m.make_request(0)
    _, _, shapes = concat_inputs(req.inputs)
    base = Plan(m.spec, host_only=True).arena_bytes(shapes, req.symbols)
    monkeypatch.delenv("FCP_DIAG", raising=False)
    fast = p.arena_bytes(shapes, req.symbols)
    monkeypatch.setenv("FCP_DIAG", "dyn_general")
    assert p.arena_bytes(shapes, req.symbols) == fast > base
    monkeypatch.delenv("FCP_DIAG", raising=False)
    for bad in (dataclasses.replace(ext, rows_source=ROWS_FROM_IDS),):
        with pytest.raises(ValueError):
            dataclasses.replace(m.spec, columns=m.spec.columns + [bad]).validate()
    with pytest.raises(lib.FcpError) as e:                     # the per-column layout has no concat matrix to leave a hole in
        Plan(dataclasses.replace(spec, layout=LAYOUT_PER_COLUMN), host_only=True)
    assert e.value.status == lib.FCP_ERR_INVALID_ARGUMENT
    only_ext = dataclasses.replace(m.spec, columns=[dataclasses.replace(ext, concat_group=0)], n_groups=1)
    with pytest.raises(lib.FcpError):                          # a group of external slots only has no row count
        Plan(only_ext, host_only=True).arena_bytes(shapes, req.symbols)


def test_placement_gate():
    """a13: replicas while the tables fit one GPU's 288 GB, sharding only beyond (north star); the
    reference's counterpart is the 256 MiB per-table gate, cuda_emitter.cc:1080-1094."""
    from recom_amd import lib, synth
    from recom_amd.placement import COLUMN_SHARD, REPLICATE, ROW_SHARD, decide_placement, table_bytes
    s2 = synth.model_s2().spec
    assert int(table_bytes(s2).sum()) == 120_000_000_000
    for world in (1, 2, 8):
        p = decide_placement(s2, world)
        assert p.mode == REPLICATE and p.bytes_per_gpu == 120_000_000_000 and p.min_world == 1
    shard = synth.model_shard().spec                        # BASELINE config 5: 480 GB
    for world, per_gpu in ((2, 240_000_000_000), (4, 120_000_000_000), (8, 60_000_000_000)):
        p = decide_placement(shard, world)
        assert p.mode == ROW_SHARD and p.bytes_per_gpu == per_gpu and p.min_world == 2
        q = decide_placement(shard, world, prefer="column")
        assert q.mode == COLUMN_SHARD and q.bytes_per_gpu >= per_gpu
    with pytest.raises(lib.FcpError) as e:                  # 480 GB on one GPU: refused, says how many it needs
        decide_placement(shard, 1)
    assert e.value.status == lib.FCP_ERR_UNSUPPORTED and "at least 2" in str(e.value)
    # one table larger than a GPU: only row sharding can hold it
    big = [400 * 10**9, 10**9]
    assert decide_placement(big, 2, prefer="column").mode == ROW_SHARD
    with pytest.raises(lib.FcpError):
        decide_placement(big, 1)
    # column sharding infeasible although the total would fit by rows: 3 tables of 200 GB on 2 GPUs
    three = [200 * 10**9] * 3
    with pytest.raises(lib.FcpError):
        decide_placement(three, 2, prefer="column", reserve_bytes=0, hbm_bytes=250 * 10**9)
    assert decide_placement(three, 3, prefer="column", reserve_bytes=0, hbm_bytes=250 * 10**9).mode == COLUMN_SHARD
    # a shared table counts once
    m = synth.model_mixed()
    assert len(table_bytes(m.spec)) == m.spec.n_device_inputs
    # the mixed preference: whole tables wherever a table fits one GPU, rows only for those that do not
    from recom_amd.placement import MIXED
    p = decide_placement(shard, 8, prefer="mixed")                 # configs[4]: every table fits -> whole columns, no rows spread
    assert p.mode == COLUMN_SHARD and all(0 <= o < 8 for o in p.owners) and len(set(p.owners)) == 8
    loads = np.bincount(p.owners, weights=table_bytes(shard), minlength=8)
    assert loads.max() == p.bytes_per_gpu and loads.max() - loads.min() <= 256_000_000   # longest-first packing: balanced
    p = decide_placement(big + [3 * 10**9] * 5, 2, prefer="mixed")  # one table exceeds a GPU: only that one is spread
    assert p.mode == MIXED and p.owners[0] == -1 and all(o in (0, 1) for o in p.owners[1:])
    assert p.bytes_per_gpu == 200 * 10**9 + max(np.bincount(p.owners[1:], weights=[10**9] + [3 * 10**9] * 5, minlength=2))
    assert decide_placement(s2, 8, prefer="mixed").mode == REPLICATE and decide_placement(s2, 8, prefer="mixed").owners == [0] * 1000
    # whole tables cannot be packed (3 x 200 GB on 2 GPUs of 250 GB) but rows can: falls back to rows for everything
    p = decide_placement([200 * 10**9] * 3 + [10], 2, prefer="mixed", reserve_bytes=0, hbm_bytes=320 * 10**9)
    assert p.mode == ROW_SHARD and set(p.owners) == {-1}
    p = decide_placement(shard, 8, prefer="row")
    assert set(p.owners) == {-1}
    # fewer whole tables than ranks: the whole-column step could not give every rank a block -> everything by rows
    p = decide_placement([400 * 10**9, 10**9, 2 * 10**9], 4, prefer="mixed")
    assert p.mode == ROW_SHARD and set(p.owners) == {-1}
    p = decide_placement([400 * 10**9] + [10**9] * 4, 4, prefer="mixed")          # four whole tables for four ranks: mixed
    assert p.mode == MIXED and p.owners[0] == -1 and sorted(p.owners[1:]) == [0, 1, 2, 3]


def test_row_ids_to_row_offsets_loop_both_code_paths():
    """fcp_pack_seg_to_csr (the host loop behind FCP_STAGE_SEG_TO_CSR): sorted row ids -> offsets[rows + 1], offsets[r] = the
    number of ids below row r — ComputeSegmentOffsets (cuda_emitter.cc:768-818) on the host.  The AVX-512 boundary form
    (int64 stride 2 = SparseTensor indices, int64 / int32 stride 1) and the portable run-length form (any stride; forced
    with FCP_DIAG=pack_no_avx512 in a child process) against NumPy on adversarial inputs: empty, one id, nnz around multiples of
    16, empty rows at both ends, ids of rows < 0 and >= rows (dropped), values beyond int32, unsorted ids (refused)."""
    import ctypes as C
    import subprocess
    import sys
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
from recom_amd import lib
L = lib.load()
L.fcp_pack_seg_to_csr.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
rng = np.random.default_rng(5)
def check(rows_of, rows, dtype, stride):
    n = rows_of.size
    mat = np.zeros((n, stride), dtype)
    mat[:, 0] = rows_of.astype(dtype) if n else 0
    if stride > 1:
        mat[:, 1:] = rng.integers(0, 7, (n, stride - 1))
    out = np.full(rows + 1, -777, np.int32)
    rc = L.fcp_pack_seg_to_csr(mat.ctypes.data, mat.dtype.itemsize, stride, n, rows, out.ctypes.data)
    srt = bool(np.all(np.diff(rows_of) >= 0))
    assert rc == (0 if srt else 1), (rc, srt)
    if srt:
        clamped = np.clip(rows_of.astype(np.int64), -1, rows)
        want = np.searchsorted(clamped, np.arange(rows + 1), side="left").astype(np.int32)
        assert np.array_equal(out, want), (dtype, stride, n, rows, out[:8], want[:8])
cases = 0
for dtype, strides in ((np.int64, (1, 2, 3)), (np.int32, (1, 2))):
    for rows in (1, 2, 7, 64, 257, 1000):
        for n in (0, 1, 2, 15, 16, 17, 31, 32, 33, 100, 1000, 4099):
            for flavour in range(5):
                if flavour == 0:
                    r = np.sort(rng.integers(0, rows, n))
                elif flavour == 1:                                   # empty rows at both ends, long runs
                    r = np.sort(rng.integers(rows // 3, max(rows // 3 + 1, rows // 2), n))
                elif flavour == 2:                                   # strays: below 0 and at / beyond `rows`
                    r = np.sort(rng.integers(-5, rows + 5, n))
                elif flavour == 3:                                   # values that do not fit int32 (int64 inputs only)
                    r = np.sort(rng.integers(0, rows, n))
                    if n and dtype == np.int64:
                        r[-1:] = 2 ** 40
                        r[:1] = -(2 ** 40)
                else:                                                # not sorted: refused
                    r = rng.integers(0, rows, n)
                for stride in strides:
                    check(np.asarray(r, np.int64), rows, dtype, stride)
                    cases += 1
print("ok", cases)
""" % ROOT
    for env_extra in ({}, {"FCP_DIAG": "pack_no_avx512"}):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, **env_extra}, timeout=600)
        assert res.returncode == 0 and res.stdout.startswith("ok"), res.stderr[-3000:] + res.stdout[-500:]


def test_hot_kernels_keep_full_occupancy(tmp_path):
    """The fused kernels must stay at <= 64 VGPRs (8 waves per SIMD, MI355X_MICROARCH.md register table) and
    use no scratch: round 2 lost 2 us of 29 on S2 when a rarely-taken branch (the id transform) pushed the
    dense kernel to 76 VGPRs.  Read from the code object metadata hipcc emits for gfx950 (no GPU needed)."""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    asm = tmp_path / "k.s"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-O3", "--offload-device-only", "-S",
                        os.path.join(ROOT, "recom_amd", "csrc", "fcp_kernels.hip"), "-o", str(asm)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = asm.read_text()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        if not any(k in name for k in ("fcp_dense_kernel", "fcp_ragged_kernel", "fcp_hybrid_kernel")):
            continue
        seen += 1
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body).group(1))
        # the instantiations every full-size workload runs (4 rows per wave, or the ragged kernel) keep 8 waves per
        # SIMD; the small-batch ones (< 64 rows: 1 or 2 rows per wave, the chip is not full anyway) may take 72
        headline = "ELi4ELb" in name or "fcp_ragged_kernel" in name
        assert vgpr <= (64 if headline else 72), f"{name}: {vgpr} VGPRs (> 64 costs a wave per SIMD)"
        assert scratch == 0, f"{name}: uses scratch"
        assert 8 * lds <= 160 * 1024, f"{name}: {lds} bytes of LDS leave fewer than 8 blocks per CU"
    assert seen >= 12
    # (r6) the three output-store policies are three INSTRUCTIONS: written as C++ (`if (plain) *p = t; else
    # __builtin_nontemporal_store(t, p)`) the compiler merged the two stores and dropped the hint — every "nt" request stored
    # with the default policy; both hinted forms are inline asm since
    label = re.search(r"^_ZN\S*fcp_dense_kernelILi4ELi4ELb0E\S*:", text, re.M)
    assert label, "dense kernel not found in the assembly"
    end = text.find(".end_amdhsa_kernel", label.end())            # (the kernel's descriptor follows its code)
    stores = re.findall(r"global_store_dwordx4 [^\n]*", text[label.end():end])
    kinds = {"sc1 nt": 0, "nt": 0, "plain": 0}
    for st in stores:
        tail = st.split("off", 1)[1].strip()
        kinds["sc1 nt" if tail.startswith("sc1 nt") else "nt" if tail.startswith("nt") else "plain"] += 1
    assert kinds["sc1 nt"] >= 4 and kinds["nt"] >= 4 and kinds["plain"] >= 4, kinds


def test_shard_exchange_entry_points_without_a_gpu():
    """The exchange under the C ABI (fcp_shard.hip): the batch split equals recom_amd.shard.batch_slices; bad
    arguments are refused before RCCL or a device is touched."""
    import ctypes as C
    from recom_amd import lib
    from recom_amd.shard import batch_slices
    L = lib.load()
    for rows, world in ((33, 2), (512, 8), (3, 4), (0, 3), (100, 7)):
        for rank, want in enumerate(batch_slices(rows, world)):
            b, c = C.c_int64(), C.c_int64()
            lib.check(L.fcp_shard_batch_slice(rows, world, rank, C.byref(b), C.byref(c)), "fcp_shard_batch_slice")
            assert (b.value, c.value) == want
    b, c = C.c_int64(), C.c_int64()
    assert L.fcp_shard_batch_slice(10, 2, 2, C.byref(b), C.byref(c)) == lib.FCP_ERR_INVALID_ARGUMENT
    assert L.fcp_comm_create(None, 0, 1, 0, C.byref(C.c_void_p())) == lib.FCP_ERR_INVALID_ARGUMENT
    assert L.fcp_shard_exchange(None, None, 4, 4, None, None, None, None) == lib.FCP_ERR_INVALID_ARGUMENT
    assert L.fcp_shard_step_run(None, None, None, None, None) == lib.FCP_ERR_INVALID_ARGUMENT
    assert L.fcp_comm_destroy(None) == lib.FCP_OK and L.fcp_shard_step_destroy(None) == lib.FCP_OK


def test_staged_concat_inputs_and_the_stage_section(L, tmp_path):
    """Addons>ConcatInputs in its staged form (fcp_concat_inputs_ex, host only): with the modes of the plan file's stage
    section the blob carries int32 ids and int32 row offsets — the converted tensors (NumPy conversion) in the staged layout:
    copied / narrowed inputs in the reference op's order, the row offsets behind them as one matrix — and the stage section survives the file (fcp_plan_file_stage_info, what the shim's ConcatInputsOp reads
    through the node's `_fcp_plan` attr)."""
    import fcp_oracle as O  # noqa: F401  (np_segment_offsets: the checker)
    from recom_amd import synth
    from recom_amd import plan as PL
    from recom_amd.ops import ConcatInputs, concat_inputs
    from recom_amd.plan_io import load_plan, load_stage, save_plan
    for m in (synth.model_mixed(batch=70, vocab=997), synth.model_ragged(columns=24, vocab=3000, batch=130, seg="indices"),
              synth.model_ragged(columns=6, vocab=500, batch=33, seg="rowids32"), synth.model_s2(columns=20, vocab=1000, batch=32)):
        spec, stage = m.spec.staged_for_concat_inputs()
        converted = PL.STAGE_SEG_TO_CSR in stage.modes
        assert (stage.symbols_input == m.spec.n_host_inputs) == converted and spec.n_host_inputs == m.spec.n_host_inputs + converted
        for k, (c0, c1) in enumerate(zip(m.spec.columns, spec.columns)):
            if c0.form == 3:                                       # ScatterNd rows come in any order: never offsets
                assert c1.seg_kind != PL.SEG_CSR_I32 or c0.seg_kind == PL.SEG_CSR_I32
        path = str(tmp_path / f"{m.name}.fcp")
        save_plan(spec, path, stage)
        again = load_stage(path)
        assert again.modes == list(stage.modes) and again.rows_symbol == list(stage.rows_symbol) and again.symbols_input == stage.symbols_input
        assert load_plan(path).n_host_inputs == spec.n_host_inputs
        from recom_amd.ops import Plan
        Plan.from_file(path, host_only=True).close()              # the library accepts the version-3 file
        op = ConcatInputs([a.ndim for a in m.make_request(0).inputs] + ([1] if converted else []), path)
        for seed in range(3):
            req = m.make_request(seed)
            raw = list(req.inputs) + ([req.symbols] if converted else [])
            blob, offsets, shapes = op(raw)
            conv = []
            for i, a in enumerate(raw):
                if stage.modes[i] == PL.STAGE_SEG_TO_CSR:
                    rows = int(req.symbols[stage.rows_symbol[i]])
                    conv.append(O.np_segment_offsets(np.asarray(a).reshape(a.shape[0], -1)[:, 0], rows).astype(np.int32))
                elif stage.modes[i] == PL.STAGE_NARROW_I64:
                    conv.append(np.where((a >= 0) & (a <= 0x7fffffff), a, -1).astype(np.int32))
                else:
                    conv.append(a)
            from conftest import assert_staged_blob
            from recom_amd.ops import pack_as_staged
            assert assert_staged_blob(blob, offsets, shapes, conv, stage.modes) == blob.size
            b3, o3, s3 = pack_as_staged(conv, stage.modes)          # the harnesses' NumPy statement of the same layout
            assert np.array_equal(o3, offsets) and np.array_equal(s3, shapes) and b3.size == blob.size
            if not converted:                                      # nothing converted: the reference op's layout, byte for byte
                b2, o2, s2 = concat_inputs(conv)
                assert np.array_equal(blob, b2) and np.array_equal(offsets, o2) and np.array_equal(shapes, s2)
            else:                                                  # the row offsets form ONE matrix behind everything else
                csr = [i for i, md in enumerate(stage.modes) if md == PL.STAGE_SEG_TO_CSR]
                by_rows = {}
                for i in csr:
                    by_rows.setdefault(conv[i].size, []).append(int(offsets[i]))
                for size_, offs_ in by_rows.items():
                    if len(by_rows) == 1:
                        assert np.array_equal(np.diff(offs_), np.full(len(offs_) - 1, 4 * size_)), "CSR arrays are not one stride apart"
                assert min(int(offsets[i]) for i in csr) >= max(int(offsets[i]) for i in range(len(conv)) if i not in csr)
            assert spec.host_input_ranks == [np.asarray(a).ndim for a in conv]
            assert spec.host_input_elem_sizes == [np.asarray(a).dtype.itemsize for a in conv]
    # a plain plan file has no stage section; a wrong one is refused
    plain = str(tmp_path / "plain.fcp")
    save_plan(m.spec, plain)
    assert load_stage(plain) is None
    text = open(path).read().replace("stage ", "stage 9")
    bad = str(tmp_path / "bad.fcp")
    open(bad, "w").write(text)
    with pytest.raises(Exception):
        load_stage(bad)
    with pytest.raises(ValueError):
        ConcatInputs([1, 2], path)                                 # input count differs from the stage section


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) starts N ranks itself through
    torch.distributed.run on 127.0.0.1, before anything touches the GPU; the dry run prints the command."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FCP_BENCH_DRY_LAUNCH"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--workload", "shard"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stderr
    cmd = json.loads(res.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--workload", "shard"]
    # N = 1 never launches anything
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0 and "--gpus" in res.stdout


def test_segment_id_maps_in_the_descriptor_and_the_plan_file(tmp_path):
    """fcp_column_ext_t (segment ids through a folded SparseReshape, cuda_emitter.cc:1874-1916): accepted by
    fcp_plan_create_ex, carried by version-4 plan files (both parsers agree), refused when malformed — no GPU needed."""
    import dataclasses
    from recom_amd.lib import FcpError
    from recom_amd.ops import Plan
    from recom_amd.plan_io import load_plan, save_plan
    from segmap_cases import build
    spec, plain, *_ = build(1)
    p = Plan(spec, host_only=True)
    assert p.counts()["columns"] == spec.n_columns
    path = str(tmp_path / "m.fcp")
    save_plan(spec, path)
    text = open(path).read()
    assert text.startswith("fcp_plan 4\n") and f"segmaps {spec.n_columns}\n" in text
    assert load_plan(path).to_dict() == spec.to_dict()
    Plan.from_file(path, host_only=True)                       # the library's own parser
    save_plan(plain, path)                                     # plans without maps keep their version
    assert open(path).read().startswith("fcp_plan 2\n")
    # a staged plan leaves mapped columns alone (the pre-pass evaluates the map on the device) ...
    staged, stage = spec.staged_for_concat_inputs()
    assert all(len(c.seg_mul) and c.seg_kind in (1, 2) for c in staged.columns)
    save_plan(staged, path, stage)
    assert open(path).read().startswith("fcp_plan 4\n") and "\nstage " in open(path).read()
    Plan.from_file(path, host_only=True)
    assert load_plan(path).to_dict() == staged.to_dict()
    # ... and refusals
    c0 = spec.columns[0]
    for bad in (dict(seg_div=0), dict(seg_mul=(1, 1, 1, 1, 1)), dict(seg_stride=1), dict(seg_sym=99), dict(seg_sym=0, seg_sym_slot=3),
                dict(seg_kind=3), dict(seg_mul=(-1, 1))):
        cols = [dataclasses.replace(c0, **bad)] + list(spec.columns[1:])
        broken = dataclasses.replace(spec, columns=cols)
        with pytest.raises((ValueError, FcpError)):
            Plan(broken, host_only=True)
        broken.validate = lambda: None                         # past the Python check: the library refuses too
        for c in cols:
            c.validate = lambda: None
        with pytest.raises(FcpError):
            Plan(broken, host_only=True)
    with open(path, "w") as f:                                 # a map that names a column twice
        f.write(text.replace(f"segmaps {spec.n_columns}\n", f"segmaps {spec.n_columns}\n0 1 -1 0 1 0 0 0 1\n", 1))
    with pytest.raises(FcpError):
        Plan.from_file(path, host_only=True)


def test_table_size_limits_are_rows_not_bytes():
    """A table (or the shard of one) may hold up to 2^32 - 3 ROWS of any width — the kernels park a row per id and form
    the byte offset in 64 bits; round 2 stopped at 2^32 16-byte slots = 64 GB.  Host-only plans: no GPU needed."""
    from recom_amd import lib
    from recom_amd.lib import FcpError
    from recom_amd.ops import Plan
    from recom_amd.plan import COMBINER_NONE, FORM_GATHER, IDS_I64, ROWS_FROM_IDS, SEG_NONE, ColumnSpec, PlanSpec
    def plan(vocab, dim, world=1):
        c = ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, IDS_I64, 0, 0, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0, 0)
        return Plan(PlanSpec([c], [1], [8], n_device_inputs=1, shard_world=world), host_only=True)
    import ctypes as C
    p = plan(4_000_000_000, 64)                     # 1 TB: 64 G slots
    total, largest = C.c_int64(), C.c_int64()
    lib.check(lib.load().fcp_plan_table_bytes(p.handle, C.byref(total), C.byref(largest)), "fcp_plan_table_bytes")
    assert total.value == largest.value == 4_000_000_000 * 64 * 4
    with pytest.raises(FcpError, match="rows"):
        plan((1 << 32) - 3, 4)
    plan((1 << 32) - 4, 4)
    plan(1 << 33, 8, world=4)                       # row-sharded: the limit applies to the shard


def test_concat_inputs_on_a_pack_pool_equals_the_single_thread_pack():
    """fcp_concat_inputs_ex_pool: one call's inputs split over a worker pool (ranges of about equal INPUT bytes) — the
    blob, offsets and shapes are those of the single-thread call, staged and unstaged, ragged and empty inputs; two
    threads sharing one pool (an op instance shared by serve workers) never block each other and agree as well."""
    import threading
    from recom_amd import lib, synth
    from recom_amd.ops import ConcatInputs, PackPool, concat_inputs
    m = synth.model_ragged(columns=96, vocab=5000, batch=64, seg="indices")
    spec, stage = m.spec.staged_for_concat_inputs()
    pool = PackPool(6)
    reqs = [m.make_request(s) for s in range(6)]
    for r in reqs:
        ins = list(r.inputs) + [r.symbols]
        for st in (None, stage):
            x = ins if st is not None else ins[:-1]
            a, b = concat_inputs(x, st), concat_inputs(x, st, pool)
            assert all(np.array_equal(p, q) for p, q in zip(a, b))
    # tiny and empty requests
    for x in ([np.zeros(0, np.int64)], [np.arange(3, dtype=np.int32), np.zeros((0, 2), np.int64), np.float32(1.5)], []):
        a, b = concat_inputs(x), concat_inputs(x, None, pool)
        assert all(np.array_equal(p, q) for p, q in zip(a, b))
    # a malformed request is refused by the pool path as by the plain one (unsorted row ids)
    bad = list(reqs[0].inputs) + [reqs[0].symbols]
    k = next(i for i, mode in enumerate(stage.modes) if mode == 2 and bad[i].shape[0] > 2)
    bad[k] = bad[k][::-1].copy()
    for p in (None, pool):
        with pytest.raises(lib.FcpError):
            concat_inputs(bad, stage, p)
    errors = []

    def worker(t):
        try:
            for it in range(40):
                r = reqs[(t + it) % len(reqs)]
                ins = list(r.inputs) + [r.symbols]
                got, want = concat_inputs(ins, stage, pool), concat_inputs(ins, stage)
                assert all(np.array_equal(p, q) for p, q in zip(got, want))
        except Exception as e:                                    # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[0]
    op = ConcatInputs(spec.host_input_ranks[:-1] + [1], threads=4)          # the op object with a pool of its own
    assert op.pool is not None
    pool.close()


def test_per_column_arena_steps_by_the_references_own_alignmem(ref_alignmem):
    """FCP_LAYOUT_PER_COLUMN is the reference's arena (cuda_emitter.cc:2151-2179): buffer_size_sum = sum of
    alignmem(elements * sizeof) over the columns' buffers.  `alignmem` here is the REFERENCE's function, compiled from
    its own source (oracle/_ref): the library's arena size for per-column plans must be that sum — no GPU needed."""
    import ctypes as C
    from recom_amd import lib, synth
    from recom_amd.ops import Plan
    from recom_amd.plan import LAYOUT_PER_COLUMN
    assert [ref_alignmem(x) for x in (0, 1, 127, 128, 129)] == [0, 128, 128, 128, 256]
    for batch in (1, 7, 50, 333):
        m = synth.model_s1(columns=23, dim=12, batch=batch)              # one-hot and multi-hot columns
        spec = m.spec.with_layout(LAYOUT_PER_COLUMN)
        p = Plan(spec, host_only=True)
        req = m.make_request(0)
        shapes = np.concatenate([np.asarray(np.asarray(a).shape, np.int32) for a in req.inputs]).astype(np.int32)
        got = C.c_int64()
        sym = None if req.symbols is None else np.ascontiguousarray(req.symbols, np.int32)
        lib.check(lib.load().fcp_plan_arena_bytes(p.handle, shapes.ctypes.data, None if sym is None else sym.ctypes.data,
                                                  C.byref(got)), "fcp_plan_arena_bytes")
        # outputs, then the row-offset buffers of the columns that bring sorted segment ids (the reference's
        # segment_offsets buffer, cuda_emitter.cc:1734-1737: num_segments + 1 ints, one more of its fc_meta buffers)
        want = sum(ref_alignmem(batch * c.dim * 4) for c in spec.columns) + \
            sum(ref_alignmem((batch + 1) * 4) for c in spec.columns if c.seg_kind in (1, 2))
        assert got.value == want


def test_cpu_baseline_record_full_size_sampled_and_quota(monkeypatch):
    """bench.py's cpu_baseline (r6): the whole workload when its tables fit host memory (`sampled: false`), the first k columns
    scaled up when they do not; `value` is the entry at min(32, CPUs the box grants) — the cgroup CPU quota counts, not the
    affinity mask — with the whole sweep and TensorFlow-CPU's dataflow beside it."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from recom_amd import synth
    m = synth.model_s2(columns=32, vocab=4000, batch=64)
    monkeypatch.delenv("FCP_BENCH_CPU_RAM_BYTES", raising=False)
    full = bench.cpu_baseline(m, budget_s=1.5)
    assert full["sampled"] is False and full["kind"] == "port" and full["value"] > 0
    assert full["cores"] <= min(32, full["usable_cores"]) and str(full["cores"]) in full["serve_workers_sweep"]
    assert full["value"] == full["serve_workers_sweep"][str(full["cores"])]
    assert full["tf_cpu_dataflow_serve_workers_sweep"] and "all 32 columns" in full["sample"]
    # a host that holds only a quarter of the tables (+ the fixed 8 GB margin the sizing keeps): a sample, scaled
    tables = sum(t.vocab * t.dim * 4 for t in m.tables)
    monkeypatch.setenv("FCP_BENCH_CPU_RAM_BYTES", str(int(((8 << 30) + tables / 3) / 0.8)))
    part = bench.cpu_baseline(m, budget_s=1.5)
    assert part["sampled"] is True and "first 8 of 32 columns" in part["sample"] or "first 16 of 32 columns" in part["sample"]
    # the quota parser
    assert bench.host_cpu_quota() is None or bench.host_cpu_quota() > 0


def test_live_traffic_falls_back_cleanly_without_a_gpu():
    """bench.py measures `roofline.traffic` in the run itself (two rocprofv3 --pmc passes over fcp_bench); where that cannot
    work — no GPU here — it says why and the caller keeps the committed record."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    got, why = bench.live_traffic(1)
    assert isinstance(why, str) and why
    assert got is None or got > 100_000_000          # (on a GPU box the passes really run: S2 moves ~148 MB per launch)
