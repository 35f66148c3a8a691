"""Multi-process CPU test of the row-sharded path: world_size 2 and
3 over gloo on 127.0.0.1.  The exchange / slicing / ordering logic under test is the
product's (recom_amd/shard.py); the per-rank partial sums and the finalize step are
supplied by the CPU oracle here (on GPUs they are FeatureColumnProcess and
fcp_shard_finalize over RCCL — covered by tests/test_gpu_parity.py on one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import copy
    import fcp_oracle as O
    from conftest import GoldenCase
    from recom_amd.shard import RowShardedPath, batch_slices

    z = np.load(os.path.join(ROOT, "tests", "golden", "fcp_golden.npz"))
    case = GoldenCase(z, "mixed_s0")
    orc = O.COracle()
    spec = case.spec()
    plan = copy.deepcopy(case.plan)
    plan["shard_rank"], plan["shard_world"] = rank, world
    tabs = [t[rank::world] for t in case.tables]
    offs = spec.column_offsets()

    def partial():
        part, _ = orc.process_feature_columns(plan, case.blob, case.offsets, case.shapes, tabs, case.symbols)
        return torch.from_numpy(part[0])

    def finalize(slices, begin, count):
        acc = np.zeros((count, slices.shape[2]), np.float32)
        for w in range(world):  # rank order, fp32 — what fcp_shard_finalize does
            acc = acc + slices[w].numpy()
        for k, c in enumerate(spec.columns):
            if c.concat_group == 0 and c.form == 2 and c.combiner == 2:
                rows = int(case.symbols[c.rows_arg])
                if c.seg_kind == 3:
                    o = case.inputs[c.seg_input]
                else:
                    o = O.np_segment_offsets(case.inputs[c.seg_input].reshape(-1)[::c.seg_stride], rows)
                cnt = np.diff(o)[begin:begin + count].astype(np.float32)
                sl = acc[:, offs[k]:offs[k] + c.dim]
                sl[cnt > 0] = sl[cnt > 0] / cnt[cnt > 0, None]
        return torch.from_numpy(acc)

    path = RowShardedPath(rank, world)
    mine, begin, count = path.run(partial, finalize)
    rows = case.expected[0].shape[0]
    assert (begin, count) == batch_slices(rows, world)[rank]
    full, _ = orc.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables, case.symbols)
    ref = full[0][begin:begin + count]
    got = mine.numpy()
    for k, c in enumerate(spec.columns):
        if c.concat_group != 0:
            continue
        a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
        if c.form in (1, 3, 4, 5):
            assert np.array_equal(a, b), f"rank {rank} column {k}: single-owner column must be exact"
        else:
            assert np.abs(a - b).max(initial=0) < 1e-5, f"rank {rank} column {k}"
    gathered = path.all_gather_batch(mine, rows).numpy()
    assert gathered.shape == full[0].shape
    assert np.abs(gathered - full[0]).max() < 1e-5
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])          # 8: the world BASELINE configs[4] names (one node of 8 GPUs)
def test_row_sharded_exchange_gloo(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_batch_slices():
    from recom_amd.shard import batch_slices
    assert batch_slices(33, 2) == [(0, 17), (17, 16)]
    assert batch_slices(512, 8) == [(64 * r, 64) for r in range(8)]
    assert batch_slices(3, 4) == [(0, 1), (1, 1), (2, 1), (3, 0)]


def _col_worker(rank, world, port, out_dir, case_name):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fcp_oracle as O
    from conftest import GoldenCase
    from recom_amd.shard import ColumnShardedPath, assign_columns, batch_slices

    z = np.load(os.path.join(ROOT, "tests", "golden", "fcp_golden.npz"))
    case = GoldenCase(z, case_name)
    orc = O.COracle()
    spec = case.spec()
    assignment = assign_columns(spec, world)
    assert sorted(k for cols in assignment for k in cols) == list(range(spec.n_columns))
    sub = spec.column_subset(assignment[rank])
    sub.spec.validate()
    blob, offsets, shapes = orc.concat_inputs([case.inputs[i] for i in sub.host_inputs])
    tabs = [case.tables[i] for i in sub.device_inputs]
    blocks, _ = orc.process_feature_columns(sub.spec.to_dict(), blob, offsets, shapes, tabs, case.symbols)
    full, _ = orc.process_feature_columns(case.plan, case.blob, case.offsets, case.shapes, case.tables, case.symbols)
    path = ColumnShardedPath(rank, world)
    for g in range(spec.n_groups):
        widths = [sum(spec.columns[k].dim for k in cols if spec.columns[k].concat_group == g) for cols in assignment]
        assert sum(widths) == spec.group_width(g)
        mine, begin, count = path.run(lambda: torch.from_numpy(blocks[g]), widths, lambda parts: torch.cat(parts, 1))
        assert (begin, count) == batch_slices(full[g].shape[0], world)[rank]
        # whole columns live on one rank: every column (pooled too) is bit-identical
        assert np.array_equal(mine.numpy(), full[g][begin:begin + count]), f"rank {rank} group {g}"
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "mixed_s0"), (2, "mixed_s1"), (3, "ragged_edges")])
def test_column_sharded_exchange_gloo(world, case, tmp_path):
    mp.spawn(_col_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_assign_columns_balanced():
    from recom_amd import synth
    from recom_amd.shard import assign_columns
    m = synth.model_s2(columns=1000)
    a = assign_columns(m.spec, 8)
    w = [sum(m.spec.columns[k].dim for k in cols) for cols in a]
    assert sum(len(c) for c in a) == 1000 and max(w) - min(w) <= 64
    offs = m.spec.column_offsets()
    for cols in a:  # contiguous column block of the concat matrix
        o = [offs[k] for k in cols]
        assert o == sorted(o) and o[-1] + m.spec.columns[cols[-1]].dim - o[0] == sum(m.spec.columns[k].dim for k in cols)
    with pytest.raises(ValueError):
        assign_columns(synth.model_s1(columns=3).spec, 4)
