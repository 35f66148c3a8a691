"""BASELINE.json configs[4] (SHARD: 4000 S2-shaped columns, 480 GB of tables over 8 GPUs) on
the HIP path: ONE rank's share of the workload at full size on one GPU.

* row sharding (what the config names): the rank holds rows ``id % 8 == rank`` of every one of
  the 4000 tables (60 GB), computes the partial sums of the whole batch — checked against the
  closed-form table definition (owned ids -> the table row, all others -> zeros) — and finalizes
  its batch slice from 8 peer slices (its own from the GPU, the 7 others synthesised from the
  closed form, which is what the peers' kernels produce) with ``fcp_shard_finalize``; the
  result must equal the unsharded closed form bit for bit.
* column sharding (SURVEY.md §8e "cheaper alternative"): the rank holds 500 whole columns
  (60 GB), produces its column block; its batch slice of all 8 blocks, put side by side by
  ``fcp_concat_outputs``, must equal the unsharded closed form.

The exchange itself (RCCL all-to-all over xGMI) needs 8 GPUs and is the driver's to run; its
slicing / ordering is covered by tests/test_shard_gloo.py and tests/test_0_gpu_shard_ranks.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORLD = 8


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from recom_amd import lib
    lib.load()
    return torch


def _ids_of(c, raw):
    import fcp_oracle as O
    return O.np_bucketize(c.boundaries, raw) if c.id_source == 2 else raw


def _closed_form_rows(model, c, ids, owner=None):
    """Table rows of column `c` for `ids`; with `owner` = (rank, world) the rows of ids owned by
    another rank are zeros (what that rank's row-sharded plan contributes)."""
    from recom_amd import synth
    rows = synth.hash_rows(model.tables[c.table_input].seed, ids, c.dim)
    if owner is not None:
        rows[(np.asarray(ids) % owner[1]) != owner[0]] = 0.0
    return rows


def _shard_model(torch, columns):
    from recom_amd import synth
    m = synth.model_shard(columns=columns)
    free, _total = torch.cuda.mem_get_info()
    need = m.table_bytes() // WORLD + (8 << 30)
    if free < need:
        pytest.skip(f"one rank's share needs {need / 2**30:.0f} GiB of HBM, {free / 2**30:.0f} GiB free")
    return m


@pytest.mark.parametrize("rank", [3])
def test_shard_config_row_sharded_rank_share(torch_cuda, rank):
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    from recom_amd.shard import batch_slices
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    m = _shard_model(torch, 4000)
    assert m.table_bytes() == 480_000_000_000 and m.table_bytes() > 288 * 10**9     # the config's premise
    spec = m.spec.with_shard(rank, WORLD)
    tabs = m.torch_tables(dev, rank, WORLD)                                          # 60 GB: rows rank, rank+8, ...
    assert sum(t.numel() * 4 for t in tabs) == 60_000_000_000
    op = FeatureColumnProcess(spec, 0)
    offs = m.spec.column_offsets()
    width = m.spec.group_width(0)
    assert width == 120_000
    for seed in (0, 1):
        req = m.make_request(seed)
        blob, offsets, shapes = concat_inputs(req.inputs)
        d_blob = torch.from_numpy(blob).to(dev)
        part = op.groups_only(d_blob, offsets, shapes, tabs, req.symbols)[0]
        torch.cuda.synchronize()
        assert part.shape == (m.batch, width)
        got = part.cpu().numpy()
        ids_all = [_ids_of(c, req.inputs[c.ids_input]) for c in m.spec.columns]
        for k, c in enumerate(m.spec.columns):                                       # partial sums: closed form
            want = _closed_form_rows(m, c, ids_all[k], (rank, WORLD))
            assert np.array_equal(got[:, offs[k]:offs[k] + c.dim], want), f"partial, column {k}"
        # finalize this rank's batch slice from the 8 peers' slices of their partials
        begin, count = batch_slices(m.batch, WORLD)[rank]
        slices = np.zeros((WORLD, count, width), np.float32)
        for k, c in enumerate(m.spec.columns):
            ids = ids_all[k][begin:begin + count]
            rows = _closed_form_rows(m, c, ids)
            owner = np.asarray(ids) % WORLD
            for g in range(WORLD):
                if g != rank:
                    slices[g, :, offs[k]:offs[k] + c.dim] = np.where((owner == g)[:, None], rows, 0.0)
        d_slices = torch.from_numpy(slices).to(dev)
        d_slices[rank] = part[begin:begin + count]                                   # this rank's own slice: the GPU's
        fin = op.shard_finalize(d_blob, offsets, shapes, tabs, req.symbols, 0, d_slices, WORLD, begin, count)
        torch.cuda.synchronize()
        fin = fin.cpu().numpy()
        for k, c in enumerate(m.spec.columns):                                       # == the unsharded result
            want = _closed_form_rows(m, c, ids_all[k][begin:begin + count])
            assert np.array_equal(fin[:, offs[k]:offs[k] + c.dim], want), f"finalize, column {k}"
    del tabs, op, part, d_slices
    torch.cuda.empty_cache()


@pytest.mark.parametrize("rank", [5])
def test_shard_config_column_sharded_rank_share(torch_cuda, rank):
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs, concat_outputs
    from recom_amd.shard import assign_columns, batch_slices
    torch = torch_cuda
    dev = torch.device("cuda", 0)
    m = _shard_model(torch, 4000)
    assignment = assign_columns(m.spec, WORLD)
    assert all(len(cols) == 500 for cols in assignment)
    sub = m.spec.column_subset(assignment[rank])
    tabs = [synth.hash_table_torch(m.tables[i].seed, m.tables[i].vocab, m.tables[i].dim, dev) for i in sub.device_inputs]
    assert sum(t.numel() * 4 for t in tabs) == 60_000_000_000
    op = FeatureColumnProcess(sub.spec, 0)
    req = m.make_request(2)
    blob, offsets, shapes = concat_inputs([req.inputs[i] for i in sub.host_inputs])
    block = op.groups_only(torch.from_numpy(blob).to(dev), offsets, shapes, tabs, req.symbols)[0]
    torch.cuda.synchronize()
    widths = [sum(m.spec.columns[k].dim for k in cols) for cols in assignment]
    assert block.shape == (m.batch, widths[rank]) and sum(widths) == 120_000
    begin, count = batch_slices(m.batch, WORLD)[rank]

    def closed_block(cols, lo, n):
        out = []
        for k in cols:
            c = m.spec.columns[k]
            out.append(_closed_form_rows(m, c, _ids_of(c, req.inputs[c.ids_input])[lo:lo + n]))
        return np.concatenate(out, axis=1)

    assert np.array_equal(block.cpu().numpy(), closed_block(assignment[rank], 0, m.batch))
    # what rank `rank` holds after the all-to-all: rows [begin, begin+count) of every rank's block
    parts = [block[begin:begin + count].contiguous() if g == rank
             else torch.from_numpy(closed_block(assignment[g], begin, count)).to(dev) for g in range(WORLD)]
    full = concat_outputs(parts)
    torch.cuda.synchronize()
    want = closed_block(list(range(m.spec.n_columns)), begin, count)
    assert np.array_equal(full.cpu().numpy(), want)
    del tabs, op, block, parts
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["row", "col"])
def test_native_exchange_and_step_world_1(torch_cuda, mode):
    """fcp_comm_* / fcp_shard_exchange* / fcp_shard_step_* through RCCL on the one GPU a test box has: a
    communicator of world 1 (send / receive to self inside one group), so what is checked is the binding,
    the buffer arithmetic and the stream ordering of partial kernel -> exchange -> finalize / concat — the
    8-GPU exchange itself is the driver's to run.  The step's result must equal the plain op's."""
    import ctypes as C
    import fcp_oracle as O
    from recom_amd import lib, synth
    from recom_amd.ops import concat_inputs
    from recom_amd.shard import Communicator, NativeShardedStep
    torch = torch_cuda
    L = lib.load()
    comm = Communicator(0, 1, 0)
    r, w = C.c_int32(-1), C.c_int32(-1)
    lib.check(L.fcp_comm_rank(comm.handle, C.byref(r), C.byref(w)), "fcp_comm_rank")
    assert (r.value, w.value) == (0, 1)
    # raw exchanges: to self
    x = torch.arange(37 * 20, dtype=torch.float32, device="cuda").view(37, 20)
    y = torch.zeros_like(x)
    b, c = C.c_int64(), C.c_int64()
    s = torch.cuda.current_stream().cuda_stream
    lib.check(L.fcp_shard_exchange(comm.handle, x.data_ptr(), 37, 20, y.data_ptr(), C.byref(b), C.byref(c), s), "exchange")
    torch.cuda.synchronize()
    assert (b.value, c.value) == (0, 37) and torch.equal(x, y)
    y.zero_()
    widths = np.asarray([20], np.int32)
    lib.check(L.fcp_shard_exchange_columns(comm.handle, x.data_ptr(), 37, widths.ctypes.data, y.data_ptr(), C.byref(b),
                                           C.byref(c), s), "exchange_columns")
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    # the whole step, every column form (mean columns: a world-1 "shard" is already divided by its kernel)
    orc = O.COracle()
    for m in (synth.model_mixed(batch=50, vocab=997, n_groups=1), synth.model_s2(columns=64, vocab=5000, batch=96)):
        step = NativeShardedStep(m, comm, mode)
        tabs_np = m.numpy_tables()
        for seed in range(5):                         # more requests than the step's ring of buffers
            req = m.make_request(seed)
            want, _ = orc.process_feature_columns(m.spec.to_dict(), *concat_inputs(req.inputs), tabs_np, req.symbols)
            ptr, begin, count = step.run(step.prepare(req.inputs, req.symbols))
            got = step.result(ptr, count).cpu().numpy()
            assert (begin, count) == (0, want[0].shape[0]) and np.array_equal(got, want[0]), (m.name, seed)
        step.close()
    comm.close()


def test_sharded_side_record_of_the_default_bench_line_over_rccl_world_1(torch_cuda, monkeypatch):
    """bench.py attaches BASELINE configs[4]'s row-sharded step to its default multi-GPU line (`sharded`,
    recom_amd.shard.bench_row_sharded_record).  On the one GPU a test box has: world 1, no process group — the NATIVE branch
    (RCCL communicator, fcp_shard_step_run, fcp_shard_exchange timed alone), shrunk to 200 columns; the record's arithmetic
    and that RCCL counted the rank."""
    import types
    from recom_amd.shard import bench_row_sharded_record
    monkeypatch.setenv("FCP_BENCH_SHARD_COLUMNS", "200")
    monkeypatch.setenv("FCP_BENCH_SHARD_VOCAB", "4000")
    rec = bench_row_sharded_record(types.SimpleNamespace(batch=96), 0, 1, 0, None, 8 << 30, budget_s=1.0)
    width = sum((8, 16, 32, 64)[c % 4] for c in range(200))
    assert rec["mode"] == "row" and rec["backend"] == "none" and rec["ranks_seen_by_rccl"] == 1
    assert (rec["columns"], rec["batch"], rec["vocab"]) == (200, 96, 4000)
    assert rec["exchange_bytes_sent_per_rank_per_request"] == 0          # world 1: nothing leaves the GPU
    assert rec["step_us"] > 0 and rec["exchange_us"] > 0 and rec["steps"] >= 5
    assert abs(rec["table_GB_per_rank"] - 4000 * width * 4 / 1e9) < 1e-9
