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


@pytest.mark.parametrize("mode", ["row", "col"])
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
