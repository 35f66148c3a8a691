"""Two ranks, ONE GPU, gloo: the sharded serving step with the PRODUCT on both sides of the exchange.

``tests/test_shard_gloo.py`` (CPU) checks the exchange's slicing / ordering with the oracle standing in
for the kernels.  Here every rank runs the real thing on ``cuda:0`` — a row-sharded (or column-sharded)
plan through ``fcp_process_feature_columns``, the all-to-all (gloo, staged through the host because both
ranks share one GPU; on an 8-GPU node it is RCCL over xGMI), then ``fcp_shard_finalize`` /
``fcp_concat_outputs`` — and compares its batch slice with the unsharded oracle.

The file name sorts first on purpose: the ranks are forked from a parent that has not touched the GPU
yet (a process that has initialised HIP must neither fork nor exec).
"""
import multiprocessing
import os
import socket
import sys
import traceback

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, mode, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import fcp_oracle as O
        from recom_amd import synth
        from recom_amd.ops import concat_inputs
        from recom_amd.shard import ColumnShardedFeatureColumns, ShardedFeatureColumns, batch_slices
        torch.cuda.set_device(0)
        orc = O.COracle()
        if mode == "mixed":
            _mixed_rank(rank, world, orc)
            dist.barrier()
            dist.destroy_process_group()
            q.put((rank, "ok"))
            return
        for m in (synth.model_mixed(batch=50, vocab=997, n_groups=1), synth.model_s2(columns=64, vocab=5000, batch=96)):
            tabs_np = m.numpy_tables()
            sfc = (ShardedFeatureColumns if mode == "row" else ColumnShardedFeatureColumns)(m, rank, world, 0)
            for seed in (0, 1, 2):
                req = m.make_request(seed)                       # identical on every rank (ids replicated)
                full = concat_inputs(req.inputs)
                want, _ = orc.process_feature_columns(m.spec.to_dict(), *full, tabs_np, req.symbols)
                if mode == "row":
                    packed = full
                else:
                    packed = concat_inputs(sfc.request_inputs(req.inputs))
                d_blob = torch.from_numpy(packed[0]).cuda()
                mine, begin, count = sfc(d_blob, packed[1], packed[2], req.symbols)
                torch.cuda.synchronize()
                assert (begin, count) == batch_slices(want[0].shape[0], world)[rank]
                got, ref = mine.cpu().numpy(), want[0][begin:begin + count]
                assert got.shape == ref.shape
                if mode == "col":                                # whole columns per rank: bit-identical
                    assert np.array_equal(got, ref)
                else:
                    offs = m.spec.column_offsets()
                    for k, c in enumerate(m.spec.columns):
                        a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
                        if c.form in (1, 3, 4, 5):               # one owner per row: exact
                            assert np.array_equal(a, b), (m.name, seed, k)
                        else:                                    # partial sums added in rank order
                            assert np.abs(a - b).max(initial=0) < 1e-5, (m.name, seed, k)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException:                                        # noqa: BLE001 - report to the parent
        q.put((rank, traceback.format_exc()))


def _mixed_model(slots=tuple(range(10))):
    """Ten columns of every lookup form; one table (1.28 MB, pooled) exceeds a 1.1 MB "GPU", the rest fit.  `slots`: the
    concat slot of every column — a permutation makes the plan's column order differ from the concat order."""
    from recom_amd import synth
    from recom_amd.plan import COMBINER_MEAN, COMBINER_SUM, FORM_GATHER_SCATTER
    b = synth._Builder()
    synth._add_dense(b, 300, 8, slot=slots[0])
    synth._add_ragged(b, 20_000, 16, slot=slots[1], combiner=COMBINER_MEAN, seg="indices")          # 1.28 MB: spread by rows
    synth._add_dense(b, 101, 8, slot=slots[2], id_source=synth.IDS_F32_BUCKETIZE, boundaries=synth.MICROBENCH_BOUNDARIES)
    synth._add_dense(b, 6_250, 8, slot=slots[3])
    synth._add_ragged(b, 700, 32, slot=slots[4], combiner=COMBINER_SUM, seg="csr")
    synth._add_ragged(b, 900, 8, slot=slots[5], combiner=COMBINER_SUM, seg="indices", form=FORM_GATHER_SCATTER)
    synth._add_dense(b, 2_000, 16, slot=slots[6])
    synth._add_ragged(b, 1_500, 12, slot=slots[7], combiner=COMBINER_MEAN, seg="rowids32")
    synth._add_dense(b, 4_000, 4, slot=slots[8])
    synth._add_dense(b, 1_000, 20, slot=slots[9])
    return synth._finish("MIXED-PLACEMENT", b, batch=57, n_symbols=1, description="one table larger than one GPU")


def _mixed_rank(rank, world, orc):
    """FCP_PLACE_MIXED on two ranks: the gate spreads only the tables that exceed "one GPU" by rows, deals the others out
    whole; the step = row-sharded part + whole-column part + strided concat; every column of the result against the
    unsharded oracle (whole columns and one-owner-per-row columns bit-exact, pooled row-sharded ones to fp32 reassociation)."""
    import torch
    from recom_amd import synth
    from recom_amd.ops import concat_inputs
    from recom_amd.placement import MIXED, decide_placement, table_bytes
    from recom_amd.shard import _GlooShardedStep, batch_slices, mixed_assignment
    m = _mixed_model()
    p = decide_placement(m.spec, world, hbm_bytes=1_100_000, reserve_bytes=0, prefer="mixed")
    assert p.mode == MIXED and p.owners.count(-1) == 1 and p.owners[1] == -1 and all(o in (-1, 0, 1) for o in p.owners)
    row_cols, per_rank = mixed_assignment(m.spec, p.owners, world)
    assert row_cols and all(per_rank)
    step = _GlooShardedStep(m, rank, world, 0, "mixed", p.owners)
    tabs_np = m.numpy_tables()
    for seed in (0, 1, 2):
        req = m.make_request(seed)
        want, _ = orc.process_feature_columns(m.spec.to_dict(), *concat_inputs(req.inputs), tabs_np, req.symbols)
        mine, begin, count = step.run(step.prepare(req.inputs, req.symbols))
        torch.cuda.synchronize()
        assert (begin, count) == batch_slices(want[0].shape[0], world)[rank]
        got, ref = mine.cpu().numpy(), want[0][begin:begin + count]
        offs = m.spec.column_offsets()
        for k, c in enumerate(m.spec.columns):
            a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
            if k not in row_cols or c.form in (1, 3, 4, 5):
                assert np.array_equal(a, b), (seed, k)
            else:
                assert np.abs(a - b).max(initial=0) < 1e-5, (seed, k)


@pytest.mark.parametrize("mode", ["row", "col", "mixed"])
def test_two_ranks_one_gpu_product_path(mode):
    import torch
    assert torch.cuda.device_count() >= 1, "needs a GPU"         # device_count() does not initialise HIP
    assert not torch.cuda.is_initialized(), "this test must run before anything initialises the GPU in this process"
    ctx = multiprocessing.get_context("fork")
    q = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, msg = q.get(timeout=600)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert results == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {m}" for r, m in results.items())


@pytest.mark.parametrize("workload", ["s2", "shard", "shard-row", "shard-col"])
def test_bench_gpus_2_starts_its_ranks_and_prints_one_line(workload):
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (torch.distributed.run on
    127.0.0.1), rank 0 prints ONE JSON line with n_gpus 2 — replicas for a model that fits (S2 reduced), the row- and the
    column-sharded step for one that does not (the gate is told the GPU is small).  gloo, both ranks on cuda:0: the
    control flow of the N > 1 run on a 1-GPU box (on a node with N GPUs the backend is RCCL)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FCP_BENCH_DEVICE"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-pcie", "--no-overlap", "--columns", "48", "--vocab", "5000", "--batch", "64"]
    if workload != "s2":
        cmd += ["--workload", workload]
        env["FCP_BENCH_HBM_BYTES"] = str(24 << 20)              # 48 tables of 160 KB .. 1.3 MB: 29 MB do not fit "one GPU"
    else:
        env["FCP_BENCH_SHARD_COLUMNS"], env["FCP_BENCH_SHARD_VOCAB"] = "400", "3000"   # the `sharded` side record, shrunk
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["value"] > 0 and rec["unit"] == "inferences/s"
    if workload == "s2":
        assert rec["scaling"] == "weak" and "replica" in rec["config"]["parallelism"]
        # (r6) the default multi-GPU line carries BASELINE configs[4]'s row-sharded step on the same ranks as a side record:
        # here 400 S2-shaped columns at the run's batch over gloo (the 1-GPU box: no RCCL, the record says so)
        sh = rec["sharded"]
        assert "error" not in sh, sh
        width = sum((8, 16, 32, 64)[c % 4] for c in range(400))
        assert sh["mode"] == "row" and sh["backend"] == "gloo" and sh["columns"] == 400 and sh["batch"] == 64 and sh["vocab"] == 3000
        assert sh["exchange_bytes_sent_per_rank_per_request"] == 64 * width * 4 // 2      # (world - 1) / world of the partial
        assert sh["step_us"] > 0 and sh["exchange_us"] > 0 and sh["ranks_seen_by_rccl"] == 0
    else:
        # `shard` IS BASELINE configs[4]: its own line is the row-sharded step; what the gate prefers (every table of this
        # model fits "one GPU" -> whole columns) rides beside it
        assert ("column-sharded" if workload == "shard-col" else "row-sharded") in rec["config"]["parallelism"]
        assert rec["config"]["exchange_bytes_sent_per_rank_per_request"] > 0
        assert "sharded" not in rec
        if workload == "shard":
            g = rec["gate_choice"]
            assert "column-sharded" in g["parallelism"] and g["value"] > 0
            assert g["exchange_bytes_sent_per_rank_per_request"] < rec["config"]["exchange_bytes_sent_per_rank_per_request"]
        else:
            assert "gate_choice" not in rec


def test_bench_gpus_2_headline_survives_a_sharded_leg_that_never_returns():
    """The `sharded` side record runs AFTER the replicated headline is measured; if its exchange never comes back (RCCL with
    N > 1 ranks is unmeasured on this pool) the watchdog prints the finished headline with the reason and every rank exits 0.
    Here the limit is 50 ms, far less than the leg needs, so the leg never completes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FCP_BENCH_DEVICE"] = "0"
    env["FCP_BENCH_SHARD_COLUMNS"], env["FCP_BENCH_SHARD_VOCAB"] = "400", "3000"
    env["FCP_BENCH_SHARDED_WATCHDOG_S"] = "0.05"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-pcie", "--no-overlap", "--columns", "48", "--vocab", "5000", "--batch", "64"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["value"] > 0 and rec["roofline"]["frac"] > 0
    # rank 0's watchdog fires first ("abandoned: ..."); were a peer to leave before it, its own wait would end in a
    # "connection closed" error instead — either way the headline is whole and the reason is on record
    assert rec["sharded"].get("error"), rec["sharded"]


def build_fake_rccl():
    """tests/native/libfake_rccl.so: the test double for the RCCL entry points (tests/native/fake_rccl.cc)."""
    import subprocess
    src, lib = os.path.join(ROOT, "tests", "native", "fake_rccl.cc"), os.path.join(ROOT, "tests", "native", "libfake_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-std=c++17", "-O1", "-fPIC", "-shared", src, "-o",
                               lib + ".tmp", "-lrt", "-lpthread"])
        os.replace(lib + ".tmp", lib)
    return lib


def _native_mixed(rank, world, orc, comm, slots):
    """FCP_PLACE_MIXED through the NATIVE steps (MixedShardedStep: fcp_shard_step_run for the row part and for the whole-
    column part + fcp_concat_outputs_scatter_strided), every column against the unsharded oracle.  `slots`: concat slots
    that differ from the plan's column order (the pieces must follow the sub-plans' own concat order)."""
    import torch
    from recom_amd.ops import concat_inputs
    from recom_amd.placement import MIXED, decide_placement
    from recom_amd.shard import MixedShardedStep, batch_slices, mixed_assignment
    m = _mixed_model(slots)
    p = decide_placement(m.spec, world, hbm_bytes=1_100_000, reserve_bytes=0, prefer="mixed")
    assert p.mode == MIXED and p.owners.count(-1) == 1
    row_cols, per_rank = mixed_assignment(m.spec, p.owners, world)
    assert row_cols and all(per_rank)
    step = MixedShardedStep(m, comm, p.owners)
    tabs_np = m.numpy_tables()
    for seed, B in ((0, 57), (1, 57), (2, 40), (3, 57), (4, 111), (5, 57)):   # ring reuse; a batch larger than the first one
        req = m.make_request(seed, B)
        want, _ = orc.process_feature_columns(m.spec.to_dict(), *concat_inputs(req.inputs), tabs_np, req.symbols)
        ptr, begin, count = step.run(step.prepare(req.inputs, req.symbols))
        torch.cuda.synchronize()
        assert (begin, count) == batch_slices(want[0].shape[0], world)[rank]
        got, ref = step.result(ptr, count).cpu().numpy(), want[0][begin:begin + count]
        offs = m.spec.column_offsets()
        for k, c in enumerate(m.spec.columns):
            a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
            if k not in row_cols or c.form in (1, 3, 4, 5):
                assert np.array_equal(a, b), (slots, seed, k)
            else:
                assert np.abs(a - b).max(initial=0) < 1e-5, (slots, seed, k)
    step.close()


def _rccl_rank_main(rank, world, port, q, fake_lib=None):
    """One rank per GPU over RCCL — or, with `fake_lib`, `world` ranks as processes on cuda:0 over the test double
    tests/native/fake_rccl.cc bound through FCP_RCCL_PATH: the NATIVE sharded step (fcp_shard_step_run: grouped ncclSend /
    ncclRecv, batch slices, ring reuse, finalize / concat) for both modes and the mixed step against the unsharded oracle."""
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if fake_lib:
            os.environ["FCP_RCCL_PATH"] = fake_lib
        import torch
        import torch.distributed as dist
        device = 0 if fake_lib else rank
        torch.cuda.set_device(device)
        if fake_lib:
            dist.init_process_group("gloo", rank=rank, world_size=world)     # only carries the communicator id
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        import fcp_oracle as O
        from recom_amd import synth
        from recom_amd.ops import concat_inputs
        from recom_amd.shard import Communicator, NativeShardedStep, batch_slices
        orc = O.COracle()
        comm = Communicator(rank, world, device, dist)
        _native_mixed(rank, world, orc, comm, tuple(range(10)))
        _native_mixed(rank, world, orc, comm, (3, 0, 7, 1, 9, 2, 5, 8, 4, 6))
        for m in (synth.model_mixed(batch=50, vocab=997, n_groups=1), synth.model_s2(columns=64, vocab=5000, batch=96)):
            tabs_np = m.numpy_tables()
            for mode in ("row", "col"):
                step = NativeShardedStep(m, comm, mode)
                for seed in (0, 1, 2, 3, 4):                      # more requests than ring entries: the ring is reused
                    req = m.make_request(seed)
                    want, _ = orc.process_feature_columns(m.spec.to_dict(), *concat_inputs(req.inputs), tabs_np, req.symbols)
                    ptr, begin, count = step.run(step.prepare(req.inputs, req.symbols))
                    torch.cuda.synchronize()
                    assert (begin, count) == batch_slices(want[0].shape[0], world)[rank]
                    got, ref = step.result(ptr, count).cpu().numpy(), want[0][begin:begin + count]
                    if mode == "col":
                        assert np.array_equal(got, ref), (m.name, mode, seed)
                    else:                                          # partial sums meet in rank order: fp32 reassociation only
                        assert np.abs(got - ref).max(initial=0) < 1e-5, (m.name, mode, seed)
                step.close()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_native_sharded_step_over_the_rccl_test_double(world):
    """fcp_shard.hip with MORE THAN ONE RANK on a 1-GPU box.  RCCL is bound at run time
    (FCP_RCCL_PATH), so a test double (tests/native/fake_rccl.cc: ncclSend / ncclRecv / ncclGroup* between processes through
    shared memory, sizes checked on both sides) stands in for it and `world` processes share cuda:0: the native step's
    send / recv schedule, batch-slice order against fcp_shard_finalize, column blocks against the concat, ring reuse over
    more requests than ring entries, and the mixed step (row part + whole-column part + strided concat, with concat slots
    that differ from the plan's column order and a batch larger than the first one) — all against the unsharded oracle."""
    import torch
    assert torch.cuda.device_count() >= 1, "needs a GPU"
    lib = build_fake_rccl()
    ctx = multiprocessing.get_context("fork")                     # the parent has not touched the GPU (see the file's docstring)
    assert not torch.cuda.is_initialized()
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_rank_main, args=(r, world, port, q, lib)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, msg = q.get(timeout=900)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert results == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {m}" for r, m in results.items())


# --- BASELINE configs[4] at its own shape: 4000 columns over 8 ranks -------------------------------------------------------
_SHARD_SHAPE = {}        # filled by the parent BEFORE it forks: the ranks inherit the oracle's results copy-on-write


def _shard_shape_models():
    """configs[4]'s shape with a vocabulary that fits a test: 4000 S2-shaped columns (dims 8/16/32/64, sum 120 000 floats per
    row, every 10th column bucketized), batch 512, vocab 2000 (0.96 GB of tables instead of 480 GB).  The second model grows
    two dim-64 tables beyond "one GPU" (200 MB) so that the gate answers MIXED: two columns by rows, 3998 whole."""
    from recom_amd import synth
    plain = synth.model_shard(columns=4000, vocab=2000, batch=512)
    grown = synth.model_shard(columns=4000, vocab=2000, batch=512, vocab_of={1003: 800_000, 2507: 800_000})
    return plain, grown


def _shard_shape_rank_main(rank, world, port, q, fake_lib):
    try:
        for p in (ROOT, os.path.join(ROOT, "oracle")):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["FCP_RCCL_PATH"] = fake_lib
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)         # only carries the communicator id
        from recom_amd.placement import MIXED, decide_placement
        from recom_amd.shard import Communicator, MixedShardedStep, NativeShardedStep, batch_slices, mixed_assignment
        plain, grown = _SHARD_SHAPE["models"]
        comm = Communicator(rank, world, 0, dist)
        B, width = plain.batch, plain.spec.group_width(0)
        assert (plain.spec.n_columns, B, width) == (4000, 512, 120_000)
        for mode in ("row", "col", "mixed"):
            model = grown if mode == "mixed" else plain
            if mode == "mixed":
                p = decide_placement(model.spec, world, hbm_bytes=200_000_000, reserve_bytes=0, prefer="mixed")
                assert p.mode == MIXED and sorted(k for k, o in enumerate(p.owners) if o < 0) == [1003, 2507]
                row_cols, per_rank = mixed_assignment(model.spec, p.owners, world)
                assert row_cols == [1003, 2507] and all(per_rank) and sum(map(len, per_rank)) == 3998
                step = MixedShardedStep(model, comm, p.owners)
            else:
                step = NativeShardedStep(model, comm, mode)
            for seed, want in enumerate(_SHARD_SHAPE[mode]):                 # 5 requests over a ring of 3 buffers
                req = model.make_request(seed)                               # ids replicated on every rank
                ptr, begin, count = step.run(step.prepare(req.inputs, req.symbols))
                torch.cuda.synchronize()
                assert (begin, count) == batch_slices(B, world)[rank] and count == B // world
                got = step.result(ptr, count).cpu().numpy()
                # one id per row: every output row has exactly one owning rank, so the partial sums (row mode: seven slices
                # of zeros + the owner's) come out bit-identical to the unsharded oracle in all three modes
                assert got.shape == (count, width) and np.array_equal(got, want[begin:begin + count]), (mode, seed)
            if mode == "row":                                                # what BASELINE configs[4] puts on the wire per rank
                assert (world - 1) * (B // world) * width * 4 == 7 * 64 * 120_000 * 4
            step.close()
            dist.barrier()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except BaseException:                                                    # noqa: BLE001 - report to the parent
        q.put((rank, traceback.format_exc()))


def test_native_sharded_step_at_configs4_shape_eight_ranks():
    """BASELINE configs[4] AT ITS OWN SHAPE through the native step — 8 ranks x 4000 columns x batch 512
    (30.7 MB per peer and request in row mode, the send / recv schedule of world 8), row-sharded, column-sharded and mixed,
    5 requests each (ring reuse), every rank's batch slice bit for bit against the UNSHARDED oracle.  Eight processes share
    cuda:0 over the RCCL test double; the oracle's results are computed once here, before the fork."""
    import torch
    assert torch.cuda.device_count() >= 1, "needs a GPU"
    assert not torch.cuda.is_initialized()
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fcp_oracle as O
    from recom_amd.ops import concat_inputs
    lib = build_fake_rccl()
    orc = O.COracle()
    plain, grown = _shard_shape_models()
    _SHARD_SHAPE.clear()
    _SHARD_SHAPE["models"] = (plain, grown)
    for mode, model in (("row", plain), ("mixed", grown)):
        tabs = model.numpy_tables()
        wants = []
        for seed in range(5):
            req = model.make_request(seed)
            want, bad = orc.process_feature_columns(model.spec.to_dict(), *concat_inputs(req.inputs), tabs, req.symbols)
            assert bad == 0 and want[0].shape == (512, 120_000)
            wants.append(want[0])
        _SHARD_SHAPE[mode] = wants
        del tabs
    _SHARD_SHAPE["col"] = _SHARD_SHAPE["row"]                                # same model, same requests
    world = 8
    ctx = multiprocessing.get_context("fork")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_shape_rank_main, args=(r, world, port, q, lib)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, msg = q.get(timeout=1500)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        _SHARD_SHAPE.clear()
    assert results == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()))


def test_bench_gpus_8_row_sharded_at_configs4_shape():
    """`python bench.py --gpus 8 --workload shard-row` at configs[4]'s column count and batch (vocab 2000; the gate is told the
    GPU holds 200 MB): the N = 8 control flow on the 1-GPU box (gloo, all ranks on cuda:0), ONE JSON line, and the bytes one
    rank puts on the wire per request = 7 peers x B/8 rows x 120 000 floats x 4 bytes."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FCP_BENCH_DEVICE"] = "0"
    env["FCP_BENCH_HBM_BYTES"] = str(200_000_000)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--steps", "4", "--warmup", "1",
           "--workload", "shard-row", "--vocab", "2000", "--no-cpu-baseline", "--no-pcie", "--no-overlap"]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["steps"] == 4 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["config"]["columns"] == 4000 and rec["config"]["batch"] == 512
    assert "row-sharded x8" in rec["config"]["parallelism"]
    assert rec["config"]["exchange_bytes_sent_per_rank_per_request"] == 7 * (512 // 8) * 120_000 * 4
