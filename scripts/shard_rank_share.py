#!/usr/bin/env python3
"""BASELINE.json configs[4] (SHARD: 4000 S2-shaped columns, 480 GB of tables over 8 GPUs): the compute ONE rank does per
request, at full size on one GPU — the part of the sharded step a 1-GPU box can time.  Row sharding: partial kernel over
the whole batch on the rank's 60 GB of rows + fcp_shard_finalize of its batch slice over 8 peer slices.  Column sharding:
the rank's 500 whole columns + the concat of 8 column blocks.  The exchange between the two halves (RCCL send/recv over
xGMI, 7/8 of a [512, 120000] fp32 matrix out and in per rank) is NOT part of these figures.
GPU box:  python scripts/shard_rank_share.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402
from recom_amd.ops import FeatureColumnProcess, concat_inputs, concat_outputs  # noqa: E402
from recom_amd.shard import assign_columns, batch_slices  # noqa: E402

WORLD, RANK = 8, 3
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
m = synth.model_shard(columns=4000)
width = m.spec.group_width(0)
begin, count = batch_slices(m.batch, WORLD)[RANK]
print(f"{m.name}: {m.spec.n_columns} columns, {m.table_bytes() / 1e9:.0f} GB of tables, batch {m.batch}, row width {width} floats; "
      f"rank {RANK} of {WORLD} finalizes rows [{begin}, {begin + count})")


def timed(fn, n):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


reqs = [m.make_request(s) for s in range(4)]

# ---- row sharding ------------------------------------------------------------------------------------------
spec = m.spec.with_shard(RANK, WORLD)
tabs = m.torch_tables(dev, RANK, WORLD)
op = FeatureColumnProcess(spec, 0)
packed = [concat_inputs(r.inputs) for r in reqs]
blobs = [torch.from_numpy(p[0]).to(dev) for p in packed]
slices = torch.randn((WORLD, count, width), device=dev)
keep = [None]


def row_partial(i):
    k = i % len(reqs)
    keep[0] = op.groups_only(blobs[k], packed[k][1], packed[k][2], tabs, reqs[k].symbols)[0]


import ctypes as C  # noqa: E402

from recom_amd import lib as _lib  # noqa: E402

L = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
fin_out = torch.empty((count, width), dtype=torch.float32, device=dev)
fin_args = [op._args(blobs[k], packed[k][1], packed[k][2], tabs, reqs[k].symbols, stream) for k in range(len(reqs))]


def row_finalize(i):  # the C entry point with arguments marshalled once (the Python op wrapper costs >100 us at 4000 tables)
    a = fin_args[i % len(reqs)][0]
    _lib.check(L.fcp_shard_finalize(op.plan.handle, C.byref(a), 0, slices.data_ptr(), WORLD, begin, count, fin_out.data_ptr(), stream),
               "fcp_shard_finalize")


def row_both(i):
    row_partial(i)
    row_finalize(i)


t_fin = timed(row_finalize, steps)
h = ServingHarness(m, n_requests=4, arena_ring=3, tables=tabs, spec=spec)      # the partial kernel from the native loop
h.run(20)
_, dev_ms, _ = h.run(steps)
t_part = dev_ms * 1e3 / steps
t_both = t_part + t_fin
h.close()
alg = m.spec.algorithmic_bytes(packed[0][2], reqs[0].symbols)
part_bytes = alg["rows"] / WORLD + alg["ids"] + alg["boundaries"] + m.batch * width * 4
fin_bytes = (WORLD + 1) * count * width * 4
print(f"row-sharded  : partial kernel {t_part:7.1f} us ({part_bytes / 1e6:.1f} MB algorithmic: 1/8 of the rows, all ids, the whole "
      f"[512, {width}] partial matrix written = {part_bytes / t_part / 1e6:.2f} TB/s), finalize {t_fin:6.1f} us "
      f"({fin_bytes / 1e6:.1f} MB = {fin_bytes / t_fin / 1e6:.2f} TB/s), sum {t_both:7.1f} us per request "
      f"-> {m.batch / t_both:.2f} M inferences/s per 8-GPU node if the exchange hides; exchanged per rank and request: "
      f"{7 / 8 * m.batch * width * 4 / 1e6:.0f} MB out + the same in")
del tabs, op, slices
keep[0] = None
torch.cuda.empty_cache()

# ---- column sharding ---------------------------------------------------------------------------------------
assignment = assign_columns(m.spec, WORLD)
sub = m.spec.column_subset(assignment[RANK])
tabs = [synth.hash_table_torch(m.tables[i].seed, m.tables[i].vocab, m.tables[i].dim, dev) for i in sub.device_inputs]
op = FeatureColumnProcess(sub.spec, 0)
packed = [concat_inputs([r.inputs[i] for i in sub.host_inputs]) for r in reqs]
blobs = [torch.from_numpy(p[0]).to(dev) for p in packed]
widths = [sum(m.spec.columns[k].dim for k in cols) for cols in assignment]
parts = [torch.randn((count, w), device=dev) for w in widths]


def col_block(i):
    k = i % len(reqs)
    keep[0] = op.groups_only(blobs[k], packed[k][1], packed[k][2], tabs, reqs[k].symbols)[0]


cat_out = torch.empty((count, width), dtype=torch.float32, device=dev)
cat_ptrs = (C.c_void_p * WORLD)(*[t.data_ptr() for t in parts])
cat_dims = np.asarray(widths, np.int32)


def col_concat(i):
    _lib.check(L.fcp_concat_outputs(cat_ptrs, cat_dims.ctypes.data, WORLD, count, cat_out.data_ptr(), stream), "fcp_concat_outputs")


def col_both(i):
    col_block(i)
    col_concat(i)


t_cat = timed(col_concat, steps)
h = ServingHarness(synth.submodel(m, assignment[RANK]), n_requests=4, arena_ring=3, tables=tabs)
h.run(20)
_, dev_ms, _ = h.run(steps)
t_blk = dev_ms * 1e3 / steps
t_both = t_blk + t_cat
h.close()
salg = sub.spec.algorithmic_bytes(packed[0][2], reqs[0].symbols)
cat_bytes = 2 * count * width * 4
print(f"column-sharded: block kernel {t_blk:7.1f} us ({salg['total'] / 1e6:.1f} MB algorithmic: {len(assignment[RANK])} whole columns "
      f"= {salg['total'] / t_blk / 1e6:.2f} TB/s), concat of 8 blocks {t_cat:6.1f} us ({cat_bytes / 1e6:.1f} MB = "
      f"{cat_bytes / t_cat / 1e6:.2f} TB/s), sum {t_both:7.1f} us per request -> {m.batch / t_both:.2f} M inferences/s "
      f"per 8-GPU node if the exchange hides; exchanged per rank and request: {7 / 8 * m.batch * widths[RANK] * 4 / 1e6:.0f} MB out "
      f"+ {7 / 8 * count * width * 4 / 1e6:.0f} MB in")

# ---- what the gate chooses (bench.py --workload shard) -----------------------------------------------------------
from recom_amd.placement import MODE_NAMES, decide_placement  # noqa: E402
from recom_amd.shard import mixed_assignment  # noqa: E402

p = decide_placement(m.spec, WORLD, prefer="mixed")
row_cols, per_rank = mixed_assignment(m.spec, p.owners, WORLD)
w_row = sum(m.spec.columns[k].dim for k in row_cols)
w_mine = sum(m.spec.columns[k].dim for k in per_rank[RANK])
sent = 4 * m.batch * (WORLD - 1) / WORLD * (w_row + w_mine)
print(f"gate (mixed preference, the default of `bench.py --workload shard`): {MODE_NAMES[p.mode]}; {len(row_cols)} column(s) spread by rows, "
      f"{sum(len(c) for c in per_rank)} whole ({len(per_rank[RANK])} on rank {RANK}, {p.bytes_per_gpu / 1e9:.1f} GB of tables on the fullest GPU); "
      f"exchanged per rank and request: {sent / 1e6:.0f} MB out (row sharding: {7 / 8 * m.batch * width * 4 / 1e6:.0f} MB)")
