"""Synthetic models and requests for the BASELINE.json configurations.

The reference builds its synthetic SavedModels with TensorFlow
(``examples/python/microbenchmark.py:52-69``, ``examples/python/dlrm.py:56-203``)
and feeds random inputs from its C++ harness (``examples/cc/recom_examples.patch
:3363-3452``).  TensorFlow is not available here, so this module generates the
*post-rewrite* form of those models directly — the column plan the retained
matchers would hand to the kernel (SURVEY.md §8a forms 1–3) — plus seeded
requests.  Shapes and distributions follow SURVEY.md §8d.

Tables are defined by a closed-form hash ``table[t][r][e] = h(seed_t, r, e)`` so
that (a) a 120 GB model can be filled on the GPU without ever existing on the
host and (b) the expected value of any gathered row can be recomputed on the
CPU at full size (``hash_rows``).  No GPU is needed to import this module.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .plan import (COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_BATCH_COL_REDUCTION, FORM_GATHER,
                   FORM_GATHER_SCATTER, FORM_PASSTHROUGH, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I32,
                   IDS_I64, ROWS_FROM_IDS, ROWS_FROM_INPUT_DIM0, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_IDS_I32,
                   SEG_IDS_I64, SEG_NONE, ColumnSpec, PlanSpec)

# Criteo-Kaggle categorical cardinalities (public DLRM configuration).
CRITEO_KAGGLE_CARDINALITIES = [
    1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
    5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]

# the reference's dominant column type: numeric -> bucketized by 100 boundaries
# 0,5,...,495 (microbenchmark.py:46)
MICROBENCH_BOUNDARIES = np.arange(0, 500, 5, dtype=np.float32)


# ---------------------------------------------------------------------------
# closed-form tables
# ---------------------------------------------------------------------------
def _mix(u: np.ndarray) -> np.ndarray:
    u = u & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(15)
    u = (u * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(12)
    u = (u * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    u ^= u >> np.uint64(15)
    return u


def hash_rows(seed: int, rows: np.ndarray, dim: int) -> np.ndarray:
    """float32 [len(rows), dim] = the table rows ``rows`` of table ``seed``."""
    r = np.asarray(rows, np.int64).astype(np.uint64)[:, None]
    e = np.arange(dim, dtype=np.uint64)[None, :]
    u = r * np.uint64(2654435761) + e * np.uint64(40503) + np.uint64((seed * 7919 + 12345) & 0xFFFFFFFF)
    u = _mix(u)
    return ((u >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).astype(np.float32)


def hash_table_numpy(seed: int, vocab: int, dim: int) -> np.ndarray:
    return hash_rows(seed, np.arange(vocab, dtype=np.int64), dim)


def hash_table_torch(seed: int, vocab: int, dim: int, device, row_begin: int = 0, row_step: int = 1):
    """The same table on a torch device (int64 arithmetic, exact).  ``row_begin`` /
    ``row_step`` select the rows of one shard (global row = begin + i*step)."""
    import torch
    n = (vocab - row_begin + row_step - 1) // row_step if vocab > row_begin else 0
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    chunk = max(1, (1 << 24) // max(dim, 1))
    e = torch.arange(dim, device=device, dtype=torch.int64)[None, :]
    M = 0xFFFFFFFF
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        r = (torch.arange(s, s + m, device=device, dtype=torch.int64) * row_step + row_begin)[:, None]
        u = (r * 2654435761 + e * 40503 + ((seed * 7919 + 12345) & M)) & M
        u = u ^ (u >> 15)
        u = (u * 0x2C1B3C6D) & M
        u = u ^ (u >> 12)
        u = (u * 0x297A2D39) & M
        u = u ^ (u >> 15)
        out[s:s + m] = (u >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    return out


# ---------------------------------------------------------------------------
@dataclass
class TableSpec:
    vocab: int
    dim: int
    seed: int


@dataclass
class Request:
    """One request: the host tensors ConcatInputs packs, and the symbols."""
    inputs: List[np.ndarray]
    symbols: Optional[np.ndarray]


@dataclass
class SynthModel:
    name: str
    spec: PlanSpec
    tables: List[TableSpec]
    batch: int
    make_request: Callable[[int], Request]
    description: str = ""

    def numpy_tables(self) -> List[np.ndarray]:
        return [hash_table_numpy(t.seed, t.vocab, t.dim) for t in self.tables]

    def torch_tables(self, device, shard_rank: int = 0, shard_world: int = 1):
        return [hash_table_torch(t.seed, t.vocab, t.dim, device, shard_rank, shard_world) for t in self.tables]

    def table_bytes(self) -> int:
        return sum(t.vocab * t.dim * 4 for t in self.tables)


def submodel(model: SynthModel, keep: Sequence[int], name: Optional[str] = None) -> SynthModel:
    """The model restricted to the columns ``keep`` (same tables, same request stream:
    request ``seed`` of the sub-model carries exactly the kept columns' tensors of request
    ``seed`` of the full model).  Used for column-sharded serving and for timing parts of
    a model in isolation."""
    sub = model.spec.column_subset(keep)

    def make_request(seed: int, B: int = model.batch) -> Request:
        r = model.make_request(seed, B)
        return Request([r.inputs[i] for i in sub.host_inputs], r.symbols)

    return SynthModel(name or f"{model.name}[{len(sub.columns)} cols]", sub.spec,
                      [model.tables[i] for i in sub.device_inputs], model.batch, make_request,
                      f"{len(sub.columns)} of {model.spec.n_columns} columns of: {model.description}")


def staged_model(model: SynthModel) -> SynthModel:
    """The model as its requests look AFTER the staging step (``fcp_stager_stage_ex`` with the modes of
    ``PlanSpec.staged()``): int64 ids narrowed to int32, sorted row ids / SparseTensor indices turned into int32 CSR
    offsets.  The conversion here is NumPy's (the stager's is tested against it bit for bit); used to time the kernels
    on inputs resident in HBM in the form the staging step leaves there."""
    from .plan import STAGE_NARROW_I64, STAGE_SEG_TO_CSR
    spec, modes, rows_col = model.spec.staged()

    def make_request(seed: int, B: int = model.batch) -> Request:
        r = model.make_request(seed, B)
        out = []
        for i, a in enumerate(r.inputs):
            if modes[i] == STAGE_SEG_TO_CSR:
                rows = int(r.symbols[model.spec.columns[rows_col[i]].rows_arg])
                seg = np.asarray(a).reshape(a.shape[0], -1)[:, 0]
                out.append(np.searchsorted(seg, np.arange(rows + 1), side="left").astype(np.int32))
            elif modes[i] == STAGE_NARROW_I64:
                out.append(np.where((a >= 0) & (a <= 0x7FFFFFFF), a, -1).astype(np.int32))
            else:
                out.append(a)
        return Request(out, r.symbols)

    out = SynthModel(model.name, spec, model.tables, model.batch, make_request,
                     model.description + "; requests as staged (ids int32, row ids -> CSR offsets on the host)")
    out.stage_modes = list(modes)   # how the staged op lays these requests out in its blob (ops.pack_as_staged)
    return out


def grouped_csr_model(model: SynthModel) -> SynthModel:
    """The same model with its host inputs re-ordered so that the row-offset (CSR) inputs of the pooled columns come LAST,
    in the columns' concat order: ``Addons>ConcatInputs`` packs inputs back to back in input order
    (concat_inputs_ops.cc:52-60), so the CSR arrays then lie one (rows + 1) x 4 bytes apart in the blob — the regular layout
    the ragged kernel's front recognises per request (FcpLaunch::csr_reg, mode 2: ranges requested together with the column
    records).  The order of a ConcatInputs node's inputs is the plan builder's to choose (``recom_amd.graph`` --staged)."""
    import dataclasses
    from .plan import SEG_CSR_I32
    spec = model.spec
    order = sorted(range(spec.n_columns), key=lambda k: (spec.columns[k].concat_group, spec.columns[k].concat_slot))
    seg_inputs = []
    for k in order:
        c = spec.columns[k]
        if c.seg_kind == SEG_CSR_I32 and c.seg_input >= 0 and c.seg_input not in seg_inputs:
            seg_inputs.append(c.seg_input)
    tail = set(seg_inputs)
    perm = [i for i in range(spec.n_host_inputs) if i not in tail] + seg_inputs      # new position -> old index
    new_of = {old: new for new, old in enumerate(perm)}
    from .plan import ROWS_FROM_INPUT_DIM0
    cols = [dataclasses.replace(c, ids_input=new_of.get(c.ids_input, -1), seg_input=new_of.get(c.seg_input, -1),
                                rows_arg=new_of[c.rows_arg] if c.rows_source == ROWS_FROM_INPUT_DIM0 else c.rows_arg)
            for c in spec.columns]
    spec2 = dataclasses.replace(spec, columns=cols, host_input_ranks=[spec.host_input_ranks[i] for i in perm],
                                host_input_elem_sizes=[spec.host_input_elem_sizes[i] for i in perm])
    spec2.validate()

    def make_request(seed: int, B: int = model.batch) -> Request:
        r = model.make_request(seed, B)
        return Request([r.inputs[i] for i in perm], r.symbols)

    return SynthModel(model.name, spec2, model.tables, model.batch, make_request,
                      model.description + "; CSR inputs last, in column order (regular in the blob)")


class _Builder:
    """Assigns host-input / table slots while columns are added."""

    def __init__(self) -> None:
        self.columns: List[ColumnSpec] = []
        self.ranks: List[int] = []
        self.esizes: List[int] = []
        self.tables: List[TableSpec] = []
        self.gens: List[Callable[[np.random.Generator, int], List[np.ndarray]]] = []

    def host_input(self, rank: int, esize: int) -> int:
        self.ranks.append(rank)
        self.esizes.append(esize)
        return len(self.ranks) - 1

    def table(self, vocab: int, dim: int) -> int:
        self.tables.append(TableSpec(vocab, dim, seed=1000 + len(self.tables)))
        return len(self.tables) - 1

    def spec(self, n_groups: int = 1, n_symbols: int = 0) -> PlanSpec:
        return PlanSpec(self.columns, self.ranks, self.esizes, len(self.tables), n_groups=n_groups,
                        n_symbols=n_symbols)


def _draw_ids(rng: np.random.Generator, n: int, vocab: int, dist: str) -> np.ndarray:
    if dist == "zipf":
        z = rng.zipf(1.05, size=n).astype(np.int64) - 1
        return (z % vocab).astype(np.int64)
    return rng.integers(0, vocab, size=n, dtype=np.int64)


def _add_dense(b: _Builder, vocab: int, dim: int, slot: int, group: int = 0, id_source: int = IDS_I64,
               dist: str = "uniform", boundaries: Optional[np.ndarray] = None) -> None:
    """Form 1: GatherV2(table, ids) (RewriteDenseInput, lookup_optimizer.cc:270-322)."""
    t = b.table(vocab, dim)
    if id_source == IDS_F32_BUCKETIZE:
        i = b.host_input(1, 4)
        lo, hi = float(boundaries[0]) - 5.0, float(boundaries[-1]) + 5.0
        b.gens.append(lambda rng, B: [rng.uniform(lo, hi, size=B).astype(np.float32)])
    elif id_source == IDS_I32:
        i = b.host_input(1, 4)
        b.gens.append(lambda rng, B: [_draw_ids(rng, B, vocab, dist).astype(np.int32)])
    else:
        i = b.host_input(1, 8)
        b.gens.append(lambda rng, B: [_draw_ids(rng, B, vocab, dist)])
    b.columns.append(ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, id_source, t, i, -1, SEG_NONE, 1,
                                ROWS_FROM_IDS, 0, boundaries, group, slot))


def _ragged_lengths(rng: np.random.Generator, B: int, max_len: int, min_len: int = 0) -> np.ndarray:
    return rng.integers(min_len, max_len + 1, size=B, dtype=np.int64)


def _add_ragged(b: _Builder, vocab: int, dim: int, slot: int, combiner: int, seg: str, max_len: int = 10,
                group: int = 0, symbol: int = 0, dist: str = "uniform", form: int = FORM_SEGMENT_REDUCE) -> None:
    """Form 2 (RewriteSeedWithNumSegments, lookup_optimizer.cc:157-268) or form 3
    (RewriteGatherScatter, :324-440).  ``seg``: 'csr' (int32 offsets[B+1]),
    'indices' (SparseTensor indices int64[nnz,2], stride 2 — what the reference
    graph delivers), 'rowids32' (int32[nnz])."""
    t = b.table(vocab, dim)
    ids_in = b.host_input(1, 8)
    if seg == "csr":
        seg_in, kind, stride = b.host_input(1, 4), SEG_CSR_I32, 1
    elif seg == "indices":
        seg_in, kind, stride = b.host_input(2, 8), SEG_IDS_I64, 2
    else:
        seg_in, kind, stride = b.host_input(1, 4), SEG_IDS_I32, 1
    mlen = 1 if form == FORM_GATHER_SCATTER else max_len

    def gen(rng: np.random.Generator, B: int) -> List[np.ndarray]:
        lens = _ragged_lengths(rng, B, mlen)
        nnz = int(lens.sum())
        ids = _draw_ids(rng, nnz, vocab, dist)
        rows = np.repeat(np.arange(B, dtype=np.int64), lens)
        if seg == "csr":
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            return [ids, offs]
        if seg == "indices":
            pos = np.concatenate([np.arange(l, dtype=np.int64) for l in lens]) if nnz else np.zeros(0, np.int64)
            return [ids, np.stack([rows, pos], axis=1).astype(np.int64).reshape(nnz, 2)]
        return [ids, rows.astype(np.int32)]

    b.gens.append(gen)
    b.columns.append(ColumnSpec(form, dim, vocab, combiner if form == FORM_SEGMENT_REDUCE else COMBINER_NONE,
                                IDS_I64, t, ids_in, seg_in, kind, stride, ROWS_FROM_SYMBOL, symbol, None,
                                group, slot))


def _finish(name: str, b: _Builder, batch: int, n_groups: int = 1, n_symbols: int = 0, description: str = "",
            symbol_values: Optional[Callable[[int], np.ndarray]] = None) -> SynthModel:
    spec = b.spec(n_groups, n_symbols)
    spec.validate()
    gens = list(b.gens)

    def make_request(seed: int, B: int = batch) -> Request:
        rng = np.random.Generator(np.random.PCG64(seed))
        inputs: List[np.ndarray] = []
        for g in gens:
            inputs.extend(g(rng, B))
        sym = None
        if n_symbols:
            sym = symbol_values(B) if symbol_values else np.full(n_symbols, B, np.int32)
        return Request(inputs, sym)

    return SynthModel(name, spec, b.tables, batch, make_request, description)


# ---------------------------------------------------------------------------
# BASELINE.json configurations
# ---------------------------------------------------------------------------
def model_s1(columns: int = 100, dim: int = 16, vocab: int = 10_000, batch: int = 128) -> SynthModel:
    """S1: 100 columns, dim 16, vocab 10k, batch 128, one id per row.  Even
    columns arrive as form 1 (dense GatherV2), odd columns as form 2 with
    SparseTensor indices (mean over exactly one id) — both rewrites the
    reference produces for a 1-id-per-row embedding_column."""
    b = _Builder()
    for c in range(columns):
        if c % 2 == 0:
            _add_dense(b, vocab, dim, slot=c)
        else:
            t = b.table(vocab, dim)
            ids_in = b.host_input(1, 8)
            seg_in = b.host_input(2, 8)

            def gen(rng, B, vocab=vocab):
                ids = rng.integers(0, vocab, size=B, dtype=np.int64)
                idx = np.stack([np.arange(B, dtype=np.int64), np.zeros(B, np.int64)], axis=1)
                return [ids, idx]

            b.gens.append(gen)
            b.columns.append(ColumnSpec(FORM_SEGMENT_REDUCE, dim, vocab, COMBINER_MEAN, IDS_I64, t, ids_in,
                                        seg_in, SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, 0, None, 0, c))
    return _finish("S1", b, batch, n_symbols=1,
                   description=f"{columns} cols, dim {dim}, vocab {vocab}, B {batch}, 1 id/row")


def model_s2(columns: int = 1000, vocab: int = 1_000_000, batch: int = 512, dist: str = "uniform",
             dims: Sequence[int] = (8, 16, 32, 64), vocab_of: Optional[dict] = None) -> SynthModel:
    """S2 (headline): 1000 columns, dims cycling 8/16/32/64, vocab 1M (120 GB of
    tables), batch 512, one id per row (form 1); every 10th column is sourced by
    a float feature bucketized with 100 boundaries (the reference's dominant
    column type), the others by int64 ids.  `vocab_of`: {column: vocab} for the
    few columns whose table is of another size (placement tests)."""
    b = _Builder()
    for c in range(columns):
        d = dims[c % len(dims)]
        v = (vocab_of or {}).get(c, vocab)
        if c % 10 == 0:
            _add_dense(b, v, d, slot=c, id_source=IDS_F32_BUCKETIZE, boundaries=MICROBENCH_BOUNDARIES)
        else:
            _add_dense(b, v, d, slot=c, dist=dist)
    return _finish("S2", b, batch,
                   description=f"{columns} cols, dims {'/'.join(map(str, dims))}, vocab {vocab}, B {batch}, "
                               f"1 id/row, 10% bucketize-f32, ids {dist}")


def model_dlrm(batch: int = 2048, dim: int = 16, cardinalities: Sequence[int] = tuple(CRITEO_KAGGLE_CARDINALITIES),
               n_dense: int = 13) -> SynthModel:
    """DLRM-style: 26 categorical (Criteo-Kaggle cardinalities, dim 16, one id
    per row) + 13 dense features passed through into one concat slot."""
    b = _Builder()
    for c, v in enumerate(cardinalities):
        _add_dense(b, int(v), dim, slot=c)
    i = b.host_input(2, 4)
    b.gens.append(lambda rng, B: [rng.uniform(0, 100, size=(B, n_dense)).astype(np.float32)])
    b.columns.append(ColumnSpec(FORM_PASSTHROUGH, n_dense, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1,
                                ROWS_FROM_INPUT_DIM0, i, None, 0, len(cardinalities)))
    return _finish("DLRM", b, batch, description=f"26 categorical dim {dim} + {n_dense} dense, B {batch}")


def model_ragged(columns: int = 512, vocab: int = 100_000, batch: int = 256, seg: str = "csr",
                 max_len: int = 10, dims: Sequence[int] = (8, 16, 32, 64), dist: str = "uniform") -> SynthModel:
    """RAGGED: 512 multi-hot columns, ids/row ~ U{0..10}, sum/mean alternating,
    nnz re-drawn per request (dynamic shapes)."""
    b = _Builder()
    for c in range(columns):
        comb = COMBINER_SUM if c % 2 == 0 else COMBINER_MEAN
        _add_ragged(b, vocab, dims[c % len(dims)], slot=c, combiner=comb, seg=seg, max_len=max_len, dist=dist)
    return _finish("RAGGED", b, batch, n_symbols=1,
                   description=f"{columns} cols multi-hot U{{0..{max_len}}}, vocab {vocab}, B {batch}, seg={seg}")


def model_shard(columns: int = 4000, vocab: int = 1_000_000, batch: int = 512, vocab_of: Optional[dict] = None) -> SynthModel:
    """SHARD: as S2 with 4000 columns (480 GB) — row-sharded over 8 GPUs."""
    m = model_s2(columns, vocab, batch, vocab_of=vocab_of)
    m.name = "SHARD"
    return m


def model_mixed(batch: int = 33, vocab: int = 997, seed_dims: Sequence[int] = (4, 8, 12, 16, 32, 64, 20),
                n_groups: int = 2) -> SynthModel:
    """Small model with every form, id source and segment encoding — the
    parity / golden-fixture workhorse."""
    b = _Builder()
    slot = [0, 0]

    def nxt(g: int) -> int:
        s = slot[g]
        slot[g] += 1
        return s

    _add_dense(b, vocab, seed_dims[0], nxt(0))
    _add_dense(b, 101, seed_dims[1], nxt(0), id_source=IDS_F32_BUCKETIZE, boundaries=MICROBENCH_BOUNDARIES)
    _add_dense(b, vocab, seed_dims[2], nxt(0), id_source=IDS_I32)
    _add_ragged(b, vocab, seed_dims[3], nxt(0), COMBINER_SUM, "csr")
    _add_ragged(b, vocab, seed_dims[4], nxt(0), COMBINER_MEAN, "indices")
    _add_ragged(b, vocab, seed_dims[5], nxt(0), COMBINER_MEAN, "rowids32", max_len=70)
    _add_ragged(b, vocab, seed_dims[6], nxt(0), COMBINER_MEAN, "csr", max_len=3)
    _add_ragged(b, vocab, seed_dims[1], nxt(0), COMBINER_NONE, "indices", form=FORM_GATHER_SCATTER)
    # passthrough dense features
    i = b.host_input(2, 4)
    b.gens.append(lambda rng, B: [rng.standard_normal((B, 12)).astype(np.float32)])
    b.columns.append(ColumnSpec(FORM_PASSTHROUGH, 12, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1,
                                ROWS_FROM_INPUT_DIM0, i, None, 0, nxt(0)))
    # Sum(x, axis=1) on a rank-3 tensor
    j = b.host_input(3, 4)
    b.gens.append(lambda rng, B: [rng.standard_normal((B, 5, 8)).astype(np.float32)])
    b.columns.append(ColumnSpec(FORM_BATCH_COL_REDUCTION, 8, 0, COMBINER_NONE, IDS_I32, -1, j, -1, SEG_NONE, 1,
                                ROWS_FROM_INPUT_DIM0, j, None, 0, nxt(0)))
    if n_groups > 1:
        # a second ConcatV2 with its own (different) prefix size: symbol 1
        _add_ragged(b, vocab, 16, nxt(1), COMBINER_SUM, "csr", group=1, symbol=1)
        _add_ragged(b, vocab, 8, nxt(1), COMBINER_MEAN, "indices", group=1, symbol=1)
    m = _finish("MIXED", b, batch, n_groups=n_groups, n_symbols=2 if n_groups > 1 else 1,
                description="every form / id source / segment encoding")
    if n_groups > 1:
        # group-1 columns draw for B rows too (same generators); symbols = [B, B]
        pass
    return m


def model_ae(which: str = "E", batch: int = 512, large_rows: int = 1 << 23) -> SynthModel:
    """The reference's own AE models E / F (``examples/python/dlrm.py:140-203``) in
    post-rewrite form: E = (880, 50, 50, 15, 5), F = (1000, 90, 100, 7, 3) columns of
    {bucketize(100 boundaries 0,5,..,495 -> 101 rows, dim 8, mean, 1 value/row),
     hash-int (100 rows, dim 8, mean), hash-str (10 000 rows, dim 8, mean),
     sparse hash-str (10 000 rows, dim 8, sum, 1-10 ids/row),
     large sparse hash-str (2^23 rows, dim 32, sum, 1-10 ids/row)}.
    String hashing / splitting is CPU id preprocessing upstream of the path: those ids arrive
    already hashed; the integer hash columns bring their raw integers and are hashed on the device.  On MI355X the 1 GiB tables the
    reference leaves on the CPU (256 MiB gate, ``fc_optimize_pass.cc:71``) stay on the GPU."""
    counts = {"E": (880, 50, 50, 15, 5), "F": (1000, 90, 100, 7, 3)}[which.upper()]
    b = _Builder()
    slot = 0
    for _ in range(counts[0]):
        t = b.table(101, 8)
        i = b.host_input(1, 4)
        b.gens.append(lambda rng, B: [rng.integers(0, 100, size=B).astype(np.float32)])  # make_num_input
        b.columns.append(ColumnSpec(FORM_GATHER, 8, 101, COMBINER_NONE, IDS_F32_BUCKETIZE, t, i, -1, SEG_NONE, 1,
                                    ROWS_FROM_IDS, 0, MICROBENCH_BOUNDARIES, 0, slot))
        slot += 1
    for _ in range(counts[1]):
        # categorical_column_with_hash_bucket over an int32 feature (make_categ_hashbucket_int, dlrm.py:55-69): the raw
        # integers arrive, AsString -> StringToHashBucketFast runs on the device (hash_buckets)
        t = b.table(100, 8)
        i = b.host_input(1, 4)
        b.gens.append(lambda rng, B: [rng.integers(0, 100, size=B).astype(np.int32)])       # make_num_input
        b.columns.append(ColumnSpec(FORM_GATHER, 8, 100, COMBINER_NONE, IDS_I32, t, i, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, None, 0,
                                    slot, hash_buckets=100))
        slot += 1
    for _ in range(counts[2]):
        _add_dense(b, 10_000, 8, slot)                          # string features: hashed on the CPU, bucket ids arrive
        slot += 1
    for rows, dim, n in ((10_000, 8, counts[3]), (large_rows, 32, counts[4])):
        for _ in range(n):
            t = b.table(rows, dim)
            ids_in, seg_in = b.host_input(1, 8), b.host_input(2, 8)

            def gen(rng, B, rows=rows):
                lens = rng.integers(1, 11, size=B)  # random.randint(1, input_cols)
                nnz = int(lens.sum())
                r = np.repeat(np.arange(B, dtype=np.int64), lens)
                pos = np.concatenate([np.arange(l, dtype=np.int64) for l in lens])
                return [rng.integers(0, rows, size=nnz, dtype=np.int64), np.stack([r, pos], axis=1)]

            b.gens.append(gen)
            b.columns.append(ColumnSpec(FORM_SEGMENT_REDUCE, dim, rows, COMBINER_SUM, IDS_I64, t, ids_in, seg_in,
                                        SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, 0, None, 0, slot))
            slot += 1
    return _finish(f"AE-{which.upper()}", b, batch, n_symbols=1,
                   description=f"reference model {which.upper()}: {counts} columns of bucketize/hash-int/hash-str/sparse/"
                               f"large-sparse, B {batch}")


MODELS = {
    "s1": model_s1, "s2": model_s2, "dlrm": model_dlrm, "ragged": model_ragged, "shard": model_shard,
    "mixed": model_mixed, "e": lambda **kw: model_ae("E", **kw), "f": lambda **kw: model_ae("F", **kw),
}
