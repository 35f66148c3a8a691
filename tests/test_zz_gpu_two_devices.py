"""The one test that needs TWO GPUs: the native sharded step over the real RCCL, one rank per device.  Every box of this
pool has one GPU, so it has never run on hardware (it is skipped there); it lives in the LAST file of the `-x` order so that on
the first box with two devices nothing else hides behind it.  The same step with two, three and eight ranks on one GPU over the
RCCL test double: tests/test_0_gpu_shard_ranks.py."""
import multiprocessing
import queue

import pytest

pytestmark = pytest.mark.gpu


def test_native_sharded_step_over_rccl_two_gpus():
    """the native RCCL step (grouped send / recv layout, batch-slice order against fcp_shard_finalize, ring
    reuse) with world = 2 on two GPUs, both modes, against the unsharded oracle.  Needs 2 GPUs (skipped on the 1-GPU
    boxes of this pool; `torch.cuda.device_count()` does not initialise the GPU in the parent)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    from test_0_gpu_shard_ranks import _free_port, _rccl_rank_main
    ctx = multiprocessing.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = []
    try:
        for _ in procs:
            results.append(q.get(timeout=600))
    except queue.Empty:
        pass
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert len(results) == 2, f"a rank did not report within 600 s (reported: {results})"
    for rank, status in results:
        assert status == "ok", f"rank {rank}:\n{status}"
