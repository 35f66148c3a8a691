"""Row-sharded tables across the GPUs of one node (BASELINE.json configs[4]).

The reference has no multi-GPU code at all; this is the
MI355X-native addition for models whose tables exceed one GPU's 288 GB HBM.

Partitioning: rank ``g`` of ``world`` owns the table rows ``{r : r % world == g}``
(stored densely, local row ``r // world``).  Every rank receives the whole request
(ids are tiny next to table rows), gathers only the ids it owns and produces a
*partial* pooled sum ``P_g[rows, width]`` (``fcp`` plan with ``shard_rank`` /
``shard_world``).  One exchange step follows: an all-to-all of the partials
partitioned along the batch — rank ``h`` receives slice ``P_g[rows_h, :]`` from every
``g`` — realised with ``all_to_all_single`` (RCCL over xGMI: point-to-point, so all
7 links of a GPU carry traffic concurrently, unlike a ring all-reduce).  Rank ``h``
then adds its ``world`` slices in rank order (deterministic) and applies the mean
division (``fcp_shard_finalize``).  Columns with exactly one owner per row (dense
one-hot, scatter, passthrough) come out bit-identical to the single-GPU result.

When the tables fit one GPU no collective is used at all: requests are sharded
across replicas (``bench.py`` default).
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional, Tuple

import numpy as np


def batch_slices(rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous split of the batch: (begin, count) per rank; the first
    ``rows % world`` ranks get one extra row."""
    base, extra = divmod(rows, world)
    out, begin = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((begin, cnt))
        begin += cnt
    return out


class RowShardedPath:
    """Exchange + finalize around a per-rank partial computation.

    ``partial_fn() -> tensor [rows, width]`` (this rank's partial sums, on the
    device the process group communicates from) and ``finalize_fn(slices[world,
    count, width], begin, count) -> tensor [count, width]`` are injected, so the same
    orchestration runs on GPUs (``FeatureColumnProcess`` + ``shard_finalize`` over
    RCCL) and in the CPU tests (oracle + NumPy over gloo).
    """

    def __init__(self, rank: int, world: int, group=None) -> None:
        self.rank, self.world, self.group = rank, world, group

    def exchange(self, partial):
        import torch
        import torch.distributed as dist
        rows, width = partial.shape
        sl = batch_slices(rows, self.world)
        begin, count = sl[self.rank]
        if self.world == 1:
            return partial.reshape(1, rows, width), begin, count
        recv = torch.empty((self.world * count, width), dtype=partial.dtype, device=partial.device)
        dist.all_to_all_single(recv, partial.contiguous(), output_split_sizes=[count] * self.world,
                               input_split_sizes=[c for _, c in sl], group=self.group)
        return recv.view(self.world, count, width), begin, count

    def run(self, partial_fn: Callable, finalize_fn: Callable):
        partial = partial_fn()
        slices, begin, count = self.exchange(partial)
        return finalize_fn(slices, begin, count), begin, count

    def all_gather_batch(self, mine, rows: int):
        """Optional: give every rank the full [rows, width] result (slices are padded
        to the largest count so one all_gather suffices)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return mine
        sl = batch_slices(rows, self.world)
        cmax = max(c for _, c in sl)
        padded = torch.zeros((cmax, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        padded[:mine.shape[0]] = mine
        parts = [torch.empty_like(padded) for _ in range(self.world)]
        dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:c] for p, (_, c) in zip(parts, sl)], dim=0)


class ShardedFeatureColumns:
    """GPU implementation: one row-sharded plan per rank."""

    def __init__(self, model, rank: int, world: int, device: int, group=None) -> None:
        import torch
        from .ops import FeatureColumnProcess
        self.torch = torch
        self.model = model
        self.spec = model.spec.with_shard(rank, world)
        self.dev = torch.device("cuda", device)
        self.op = FeatureColumnProcess(self.spec, device)
        self.tables = model.torch_tables(self.dev, rank, world)
        self.path = RowShardedPath(rank, world, group)

    def __call__(self, d_blob, offsets, shapes, symbols, group: int = 0):
        op, tables = self.op, self.tables

        def partial():
            return op.groups_only(d_blob, offsets, shapes, tables, symbols)[group]

        def finalize(slices, begin, count):
            return op.shard_finalize(d_blob, offsets, shapes, tables, symbols, group, slices.contiguous(),
                                     self.path.world, begin, count)

        return self.path.run(partial, finalize)


def bench_sharded(args, rank: int, world: int, local_rank: int, dist) -> dict:
    """`bench.py --workload shard`: S2-shaped model with 500 columns per GPU
    (60 GB of table rows per GPU; 4000 columns / 480 GB at 8 GPUs), tables
    row-sharded, one all-to-all of partial sums per request."""
    import torch
    from . import synth
    from .ops import concat_inputs

    columns = args.columns or 500 * world
    model = synth.model_s2(columns=columns)
    model.name = "SHARD"
    sfc = ShardedFeatureColumns(model, rank, world, local_rank)
    reqs = [model.make_request(s) for s in range(8)]  # identical on every rank (ids replicated)
    packed = [concat_inputs(r.inputs) for r in reqs]
    blobs = [torch.from_numpy(p[0]).to(sfc.dev) for p in packed]

    def step(i):
        k = i % len(reqs)
        return sfc(blobs[k], packed[k][1], packed[k][2], reqs[k].symbols)

    for i in range(max(args.warmup, 1)):
        step(i)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(args.steps):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    batch = model.batch
    width = model.spec.group_width(0)
    bytes_alg = model.spec.algorithmic_bytes(packed[0][2], reqs[0].symbols)
    # per-GPU algorithmic bytes: 1/world of the table rows, all ids, the partial
    # [rows, width] written once, its slices sent / received, the final slice written
    per_gpu = bytes_alg["rows"] / world + bytes_alg["ids"] + bytes_alg["boundaries"] + batch * width * 4 * (
        1 + 2.0 * (world - 1) / world + 1.0 / world)
    dev_s = e0.elapsed_time(e1) * 1e-3 / args.steps
    return {
        "metric": "inference QPS, row-sharded tables (SHARD config)", "value": batch * args.steps / elapsed,
        "unit": "inferences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"SHARD: {model.description}; {columns} columns, tables row-sharded over {world} GPU(s)",
                   "batch": batch, "columns": columns, "table_bytes": model.table_bytes(),
                   "parallelism": f"row-sharded x{world}, all_to_all_single of partial sums (RCCL)"},
        "roofline": {"bound": "hbm", "achieved": per_gpu / dev_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": per_gpu / dev_s / 1e9 / 8000.0, "traffic": None,
                     "note": "per GPU, whole step (partial kernel + exchange + finalize), torch events on the compute stream"},
    }
