"""Randomised parity: random column plans (forms, dims, id sources, segment
encodings, concat groups, batch sizes, bag lengths; bucketize boundary arrays of every tier —
arithmetic, nearly uniform, duplicated, random — with boundary-exact / NaN / inf values; id transforms
and integer hashing; ids outside the vocabulary) through the C ABI against the oracle — bit-exact,
including pooled columns (same sequential fp32 order); the same for row-sharded plans (per-rank
partials).  Also host threads sharing one plan on their own streams with ever-changing shapes."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_BATCH_COL_REDUCTION, FORM_GATHER,  # noqa: E402
                            FORM_GATHER_SCATTER, FORM_PASSTHROUGH, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I32,
                            IDS_I64, ROWS_FROM_IDS, ROWS_FROM_INPUT_DIM0, ROWS_FROM_SYMBOL, SEG_CSR_I32, SEG_IDS_I32,
                            SEG_IDS_I64, SEG_NONE, XFORM_FILTER, XFORM_NONE, XFORM_SELECT, ColumnSpec, PlanSpec)


def random_boundaries(rng, nb):
    """One boundary array per bucketize tier of the kernels (computed / guess + verify / binary search)."""
    kind = str(rng.choice(["random", "arith", "arith", "linspace", "jitter", "dups"]))
    if kind == "arith":                               # exactly fma(i, step, b0) in fp32 for most (b0, step)
        b0, step = np.float32(rng.choice([-10.0, -3.5, 0.0, 0.25])), np.float32(rng.choice([0.5, 1.0, 0.37, 2.0, 0.001]))
        b = b0 + np.arange(nb, dtype=np.float32) * step
    elif kind == "linspace":
        b = np.linspace(-10, 10, nb, dtype=np.float32)
    elif kind == "jitter":
        b = np.linspace(-10, 10, nb) + rng.uniform(-0.4, 0.4, nb) * (20.0 / max(nb, 1))
    elif kind == "dups":
        b = np.round(rng.uniform(-10, 10, nb) * 2) / 2
    else:
        b = rng.uniform(-10, 10, nb)
    return np.sort(np.asarray(b, np.float32))


def random_values(rng, n, bnd):
    """float32 features for a bucketize column: mostly uniform, some exactly on a boundary, a few NaN / inf / huge."""
    v = rng.uniform(-12, 12, n).astype(np.float32)
    if n:
        on = rng.random(n) < 0.15
        v[on] = bnd[rng.integers(0, len(bnd), int(on.sum()))]
        odd = rng.random(n) < 0.03
        v[odd] = rng.choice(np.asarray([np.nan, np.inf, -np.inf, 3e38, -3e38, 0.0, -0.0], np.float32), int(odd.sum()))
    return v


def random_model(rng, dense_only=False, extended=True, n_groups_fixed=None, forms_fixed=None, segs_fixed=None):
    """Returns (spec, tables, make_inputs(batch per group) -> (inputs, symbols)).  The `_fixed` arguments narrow the draw
    (the regular-row-offsets fuzz below); the default draw is unchanged by them."""
    vec = int(rng.choice([4, 4, 4, 2, 1]))
    n_groups = int(rng.integers(1, 4))
    if n_groups_fixed:
        n_groups = n_groups_fixed
    n_cols = int(rng.integers(1, 40)) if rng.random() > 0.12 else int(rng.integers(100, 400))   # some plans span dozens of 64-slot spans
    cols, ranks, esz, tables, gens = [], [], [], [], []
    extra_syms = []                      # symbols beyond the groups' row counts: per-request factors of segment-id maps
    slots = [0] * n_groups

    def host(rank, e):
        ranks.append(rank)
        esz.append(e)
        return len(ranks) - 1

    for _ in range(n_cols):
        g = int(rng.integers(0, n_groups))
        dim = vec * int(rng.integers(1, 17 if vec < 4 else 33))
        vocab = int(rng.integers(1, 400))
        forms = [FORM_GATHER, FORM_PASSTHROUGH] if dense_only else \
            [FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER, FORM_PASSTHROUGH,
             FORM_BATCH_COL_REDUCTION]
        if forms_fixed:
            forms = list(forms_fixed)
        form = int(rng.choice(forms))
        slot = slots[g]
        slots[g] += 1
        if form == FORM_PASSTHROUGH:
            i = host(2, 4)
            gens.append((g, lambda r, B, dim=dim: [r.standard_normal((B, dim)).astype(np.float32)]))
            cols.append(ColumnSpec(form, dim, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1, ROWS_FROM_INPUT_DIM0, i,
                                   None, g, slot))
            continue
        if form == FORM_BATCH_COL_REDUCTION:
            i = host(3, 4)
            inner = int(rng.integers(0, 6))
            gens.append((g, lambda r, B, dim=dim, inner=inner: [r.standard_normal((B, inner, dim)).astype(np.float32)]))
            cols.append(ColumnSpec(form, dim, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1, ROWS_FROM_INPUT_DIM0, i,
                                   None, g, slot))
            continue
        tables.append(rng.standard_normal((vocab, dim)).astype(np.float32))
        t = len(tables) - 1
        src = int(rng.choice([IDS_I32, IDS_I64, IDS_I64, IDS_F32_BUCKETIZE]))
        bnd = None
        if src == IDS_F32_BUCKETIZE:
            nb = vocab - 1 if vocab > 1 else 1        # vocab 1: bucket 1 overflows the 1-row table: bad ids are part of the test
            bnd = random_boundaries(rng, nb)
        id_esz = 8 if src == IDS_I64 else 4
        # integer hashing in front (raw ids of any magnitude), interval select / filter, ids outside the vocabulary
        hashb = int(rng.integers(1, vocab + 1)) if extended and src != IDS_F32_BUCKETIZE and rng.random() < 0.25 else 0
        xf = dict(xform_mode=XFORM_NONE)
        if extended and rng.random() < 0.3:
            n_iv = int(rng.integers(0, 4))
            los = rng.integers(-2, vocab + 2, n_iv)
            his = los + rng.integers(0, max(vocab // 2, 1) + 1, n_iv)
            xf = dict(xform_mode=int(rng.choice([XFORM_SELECT, XFORM_FILTER])), xform_lo=tuple(int(v) for v in los),
                      xform_hi=tuple(int(v) for v in his), xform_substitute=int(rng.integers(-1, vocab + 1)))
        oob = extended and not hashb and rng.random() < 0.25
        xf["hash_buckets"] = hashb

        def draw_ids(r, n, src=src, vocab=vocab, bnd=bnd, hashb=hashb, oob=oob):
            if src == IDS_F32_BUCKETIZE:
                return random_values(r, n, bnd) if extended else r.uniform(-12, 12, n).astype(np.float32)
            dt = np.int64 if src == IDS_I64 else np.int32
            if hashb:
                info = np.iinfo(dt)
                return r.integers(info.min, info.max, n, dtype=dt, endpoint=True)
            ids = r.integers(0, vocab, n).astype(dt)
            if oob and n:
                bad = r.random(n) < 0.05
                ids[bad] = r.choice(np.asarray([-1, -7, vocab, vocab + 3, np.iinfo(dt).max, np.iinfo(dt).min], dt), int(bad.sum()))
            return ids

        if form == FORM_GATHER:
            i = host(1, id_esz)
            gens.append((g, lambda r, B, d=draw_ids: [d(r, B)]))
            cols.append(ColumnSpec(form, dim, vocab, COMBINER_NONE, src, t, i, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, bnd, g, slot, **xf))
            continue
        seg = str(rng.choice(list(segs_fixed) if segs_fixed else ["csr", "indices", "rowids32"]))
        # ScatterNd columns: mostly at most one id per row, sometimes rows hit several times (the last write wins); with
        # row ids (not offsets) the pairs arrive in any order and some rows lie outside the output
        max_len = int(rng.choice([1, 1, 3])) if form == FORM_GATHER_SCATTER else int(rng.choice([0, 1, 3, 10, 10, 70, 200, 500]))
        shuffle = form == FORM_GATHER_SCATTER and seg != "csr" and rng.random() < 0.7
        i = host(1, id_esz)
        # a SparseReshape folded into the segment ids (fcp_column_ext_t::seg_map_*, cuda_emitter.cc:1874-1916): the row is
        # idx0 // A ("div": [B*A, L] -> [B, A*L]) or idx0 * T + idx1 ("mul": [B/T, T, L] -> [B, L]) of the index matrix,
        # with A / T a constant or a per-request symbol
        segmap, seg_kw = None, {}
        if extended and seg == "indices" and form == FORM_SEGMENT_REDUCE and rng.random() < 0.35:
            segmap = (str(rng.choice(["div", "mul"])), int(rng.integers(1, 6)))
            as_symbol = rng.random() < 0.5
            if as_symbol:
                extra_syms.append(segmap[1])
            sym = n_groups + len(extra_syms) - 1 if as_symbol else -1
            f = 1 if as_symbol else segmap[1]
            seg_kw = dict(seg_mul=(1,), seg_div=f, seg_sym=sym, seg_sym_slot=4) if segmap[0] == "div" else \
                dict(seg_mul=(f, 1), seg_div=1, seg_sym=sym, seg_sym_slot=0)
        if seg == "csr":
            si, kind, stride = host(1, 4), SEG_CSR_I32, 1
        elif seg == "indices":
            si, kind, stride = host(2, 8), SEG_IDS_I64, (3 if segmap and segmap[0] == "mul" else 2)
        else:
            si, kind, stride = host(1, 4), SEG_IDS_I32, 1

        def gen(r, B, d=draw_ids, seg=seg, max_len=max_len, shuffle=shuffle, segmap=segmap):
            lens = r.integers(0, max_len + 1, B)
            nnz = int(lens.sum())
            rows = np.repeat(np.arange(B, dtype=np.int64), lens)
            if shuffle and nnz:
                rows = rows[r.permutation(nnz)]
                stray = r.random(nnz) < 0.03
                rows[stray] = r.choice(np.asarray([-1, B, B + 5, -B - 1], np.int64), int(stray.sum()))
            if seg == "csr":
                s = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            elif seg == "indices" and segmap:
                kind_, f = segmap
                pos = np.arange(nnz, dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)     # position inside the bag
                if kind_ == "div":     # idx0 = row * A + a, a non-decreasing inside the bag: idx0 // A = row
                    a = np.minimum(pos * f // np.maximum(np.repeat(lens, lens), 1), f - 1)
                    s = np.stack([rows * f + a, pos], axis=1).astype(np.int64).reshape(nnz, 2)
                else:                  # (idx0, idx1) = divmod(row, T): idx0 * T + idx1 = row
                    s = np.stack([rows // f, rows % f, pos], axis=1).astype(np.int64).reshape(nnz, 3)
            elif seg == "indices":
                s = np.stack([rows, np.zeros_like(rows)], axis=1).astype(np.int64).reshape(nnz, 2)
            else:
                s = rows.astype(np.int32)
            return [d(r, nnz), s]

        gens.append((g, gen))
        comb = int(rng.choice([COMBINER_SUM, COMBINER_MEAN])) if form == FORM_SEGMENT_REDUCE else COMBINER_NONE
        cols.append(ColumnSpec(form, dim, vocab, comb, src, t, i, si, kind, stride, ROWS_FROM_SYMBOL, g, bnd, g, slot, **xf,
                               **seg_kw))
    # every group needs a column
    for g in range(n_groups):
        if slots[g] == 0:
            i = host(2, 4)
            gens.append((g, lambda r, B: [r.standard_normal((B, vec)).astype(np.float32)]))
            cols.append(ColumnSpec(FORM_PASSTHROUGH, vec, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1,
                                   ROWS_FROM_INPUT_DIM0, i, None, g, 0))
            slots[g] = 1
    spec = PlanSpec(cols, ranks, esz, len(tables), n_groups=n_groups, n_symbols=n_groups + len(extra_syms))
    spec.validate()

    def make(r, batches):
        inputs = []
        for g, gen in gens:
            inputs.extend(gen(r, batches[g]))
        return inputs, np.asarray(list(batches) + extra_syms, np.int32)      # row counts, then the maps' factors

    return spec, tables, make


# FCP_FUZZ_SEEDS=<n>: longer soak runs (the suite keeps 24)
@pytest.mark.parametrize("seed", range(int(os.environ.get("FCP_FUZZ_SEED0", "0")),
                                        int(os.environ.get("FCP_FUZZ_SEED0", "0")) + int(os.environ.get("FCP_FUZZ_SEEDS", "24"))))
def test_random_plans_match_oracle(oracle, seed):
    import dataclasses
    import torch
    from recom_amd.plan import FLAG_COUNT_BAD_IDS
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    rng = np.random.default_rng(1000 + seed)
    spec, tables, make = random_model(rng, dense_only=(seed % 4 == 3))
    spec = dataclasses.replace(spec, flags=FLAG_COUNT_BAD_IDS)
    bad_total = 0
    dev = torch.device("cuda", 0)
    d_tabs = [torch.from_numpy(t).to(dev) for t in tables]
    op = FeatureColumnProcess(spec, 0)
    # (r4) the serving modes ride along: a third of the seeds on the plan's private streams, a third with the inputs-ready
    # request order (any-order launch; the blob is synchronised first, the arena fresh: the promise holds)
    mode = ("stream", "private", "inputs_ready")[seed % 3]
    if mode == "private":
        op.plan.set_private_streams(1 + seed % 4, always=True)
    elif mode == "inputs_ready":
        op.plan.set_inputs_ready(True)
    for trial in range(4):
        batches = [int(rng.choice([1, 2, 5, 33, 64, 130, 257, 700])) for _ in range(spec.n_groups)]
        inputs, symbols = make(rng, batches)
        blob, offsets, shapes = concat_inputs(inputs)
        d_blob = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(0, dtype=torch.int8, device=dev)
        if mode == "inputs_ready":
            torch.cuda.synchronize()
        out = op(d_blob, offsets, shapes, d_tabs, symbols)
        torch.cuda.synchronize()
        want, _bad = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tables, symbols)
        for g, w in enumerate(want):
            got = out.groups[g].cpu().numpy()
            assert got.shape == w.shape, (seed, trial, g)
            assert np.array_equal(got, w), (seed, trial, g, float(np.abs(got - w).max()))
        bad_total += int(_bad)                          # ids outside the vocabulary: zeros in the output, counted once each
        assert op.plan.read_bad_ids() == bad_total, (seed, trial)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FCP_FUZZ_SEED0", "0")),
                                        int(os.environ.get("FCP_FUZZ_SEED0", "0")) + int(os.environ.get("FCP_FUZZ_SHARD_SEEDS", "8"))))
def test_random_plans_row_sharded_match_oracle(oracle, seed):
    """The same random plans as one rank of a row-sharded world (rows id % world == rank of every table): the
    rank's partial sums equal the sharded oracle bit for bit — transforms and hashing run before the ownership
    test, table-free columns belong to rank 0."""
    import torch
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    rng = np.random.default_rng(5000 + seed)
    spec, tables, make = random_model(rng, dense_only=(seed % 4 == 3))
    world = int(rng.choice([2, 3, 8]))
    dev = torch.device("cuda", 0)
    for rank in sorted({0, int(rng.integers(0, world)), world - 1}):
        sspec = spec.with_shard(rank, world)
        shard = [np.ascontiguousarray(t[rank::world]) for t in tables]
        d_tabs = [torch.from_numpy(t).to(dev) for t in shard]
        op = FeatureColumnProcess(sspec, 0)
        for trial in range(2):
            batches = [int(rng.choice([1, 5, 64, 130])) for _ in range(spec.n_groups)]
            inputs, symbols = make(rng, batches)
            blob, offsets, shapes = concat_inputs(inputs)
            d_blob = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(0, dtype=torch.int8, device=dev)
            out = op(d_blob, offsets, shapes, d_tabs, symbols)
            torch.cuda.synchronize()
            want, _bad = oracle.process_feature_columns(sspec.to_dict(), blob, offsets, shapes, shard, symbols)
            for g, w in enumerate(want):
                got = out.groups[g].cpu().numpy()
                assert got.shape == w.shape and np.array_equal(got, w), (seed, rank, world, trial, g)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FCP_FUZZ_SEED0", "0")),
                                        int(os.environ.get("FCP_FUZZ_SEED0", "0")) + int(os.environ.get("FCP_FUZZ_FINALIZE_SEEDS", "6"))))
def test_random_plans_sharded_then_finalized_equal_the_unsharded_result(oracle, seed):
    """All ranks' partial sums of a random plan, a random batch slice of each, fcp_shard_finalize: equal to the
    unsharded oracle — exactly for columns with one owner per row (gathers, scatters, table-free columns),
    within 1e-5 of the bag's absolute sum for pooled ones (the adds are regrouped by owner); a filtered mean
    divides by the kept count, which the finalizing rank recounts from the row's ids."""
    import dataclasses
    import torch
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    rng = np.random.default_rng(9000 + seed)
    spec, tables, make = random_model(rng, dense_only=(seed % 4 == 3))
    world = int(rng.choice([2, 3, 8]))
    dev = torch.device("cuda", 0)
    batches = [int(rng.choice([3, 64, 130])) for _ in range(spec.n_groups)]
    inputs, symbols = make(rng, batches)
    blob, offsets, shapes = concat_inputs(inputs)
    d_blob = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(0, dtype=torch.int8, device=dev)
    full, _ = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tables, symbols)
    # an upper bound of what reassociation can move: the same plan over |tables|, pooled by sum
    abs_spec = dataclasses.replace(spec, columns=[dataclasses.replace(c, combiner=COMBINER_SUM) if c.form == FORM_SEGMENT_REDUCE else c
                                                  for c in spec.columns])
    mag, _ = oracle.process_feature_columns(abs_spec.to_dict(), blob, offsets, shapes, [np.abs(t) for t in tables], symbols)
    ops, tabs, parts = [], [], []
    for rank in range(world):
        sspec = spec.with_shard(rank, world)
        d_tabs = [torch.from_numpy(np.ascontiguousarray(t[rank::world])).to(dev) for t in tables]
        op = FeatureColumnProcess(sspec, 0)
        out = op(d_blob, offsets, shapes, d_tabs, symbols)
        torch.cuda.synchronize()
        parts.append([g.clone() for g in out.groups])
        ops.append(op)
        tabs.append(d_tabs)
    offs = spec.column_offsets()
    for g in range(spec.n_groups):
        rows = batches[g]
        lo = int(rng.integers(0, rows))
        cnt = int(rng.integers(1, rows - lo + 1))
        sl = torch.stack([parts[r][g][lo:lo + cnt] for r in range(world)]).contiguous()
        r = int(rng.integers(0, world))
        fin = ops[r].shard_finalize(d_blob, offsets, shapes, tabs[r], symbols, g, sl, world, lo, cnt)
        torch.cuda.synchronize()
        got, ref, m = fin.cpu().numpy(), full[g][lo:lo + cnt], mag[g][lo:lo + cnt]
        for k, c in enumerate(spec.columns):
            if c.concat_group != g:
                continue
            a, b = got[:, offs[k]:offs[k] + c.dim], ref[:, offs[k]:offs[k] + c.dim]
            if c.form != FORM_SEGMENT_REDUCE:
                assert np.array_equal(a, b), (seed, g, k, c.form)
            else:
                assert np.all(np.abs(a - b) <= 1e-5 * np.maximum(np.abs(m[:, offs[k]:offs[k] + c.dim]), 1.0)), (seed, g, k)


def test_sixteen_concat_groups_with_their_own_batches(oracle):
    """The most concat groups a plan may have (FCP_MAX_GROUPS_ABI = 16), every group with its own batch size and a mix of
    one-hot, pooled and passthrough columns: the block -> group search of the kernels and the per-group geometry."""
    import torch
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    rng = np.random.default_rng(99)
    cols, ranks, esz, tables, gens = [], [], [], [], []

    def host(rank, e):
        ranks.append(rank)
        esz.append(e)
        return len(ranks) - 1

    for g in range(16):
        for slot in range(int(rng.integers(1, 5))):
            dim = 4 * int(rng.integers(1, 20))
            vocab = int(rng.integers(5, 300))
            kind = int(rng.integers(0, 3))
            if kind == 2:
                i = host(2, 4)
                gens.append((g, lambda r, B, dim=dim: [r.standard_normal((B, dim)).astype(np.float32)]))
                cols.append(ColumnSpec(FORM_PASSTHROUGH, dim, 0, COMBINER_NONE, IDS_I32, -1, i, -1, SEG_NONE, 1, ROWS_FROM_INPUT_DIM0, i, None, g, slot))
                continue
            tables.append(rng.standard_normal((vocab, dim)).astype(np.float32))
            t = len(tables) - 1
            if kind == 0:
                i = host(1, 8)
                gens.append((g, lambda r, B, vocab=vocab: [r.integers(0, vocab, B).astype(np.int64)]))
                cols.append(ColumnSpec(FORM_GATHER, dim, vocab, COMBINER_NONE, IDS_I64, t, i, -1, SEG_NONE, 1, ROWS_FROM_IDS, 0, None, g, slot))
            else:
                i, si = host(1, 8), host(2, 8)

                def gen(r, B, vocab=vocab):
                    lens = r.integers(0, 6, B)
                    rows = np.repeat(np.arange(B, dtype=np.int64), lens)
                    return [r.integers(0, vocab, int(lens.sum())).astype(np.int64), np.stack([rows, np.zeros_like(rows)], axis=1)]

                gens.append((g, gen))
                cols.append(ColumnSpec(FORM_SEGMENT_REDUCE, dim, vocab, COMBINER_MEAN, IDS_I64, t, i, si, SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, g, None, g, slot))
    spec = PlanSpec(cols, ranks, esz, len(tables), n_groups=16, n_symbols=16)
    spec.validate()
    dev = torch.device("cuda", 0)
    d_tabs = [torch.from_numpy(t).to(dev) for t in tables]
    op = FeatureColumnProcess(spec, 0)
    for trial in range(3):
        batches = [int(rng.choice([1, 3, 17, 64, 200])) for _ in range(16)]
        inputs = []
        for g, gen in gens:
            inputs.extend(gen(rng, batches[g]))
        symbols = np.asarray(batches, np.int32)
        blob, offsets, shapes = concat_inputs(inputs)
        out = op(torch.from_numpy(blob).to(dev), offsets, shapes, d_tabs, symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(spec.to_dict(), blob, offsets, shapes, tables, symbols)
        assert len(out.groups) == 16
        for g, w in enumerate(want):
            got = out.groups[g].cpu().numpy()
            assert got.shape == w.shape and np.array_equal(got, w), (trial, g)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FCP_FUZZ_SEED0", "0")),
                                        int(os.environ.get("FCP_FUZZ_SEED0", "0")) + int(os.environ.get("FCP_FUZZ_REGULAR_SEEDS", "9"))))
def test_random_one_group_pooled_plans_with_regular_row_offsets(oracle, monkeypatch, seed):
    """(r6) The shapes of plan the ragged kernel's regular-row-offsets front serves (FcpLaunch::csr_reg, `find_regular_csr`),
    drawn at random — the general fuzz above seldom draws them (one concat group AND no pooled column with CSR offsets of its
    own).  seed % 3 == 0: mostly pooled columns over SparseTensor indices / sorted row ids with the pre-pass forced: the arena
    scratch is laid out by column position (mode 1), gathers and table-free columns in between.  == 1: every column pooled or
    scattered over CSR offsets packed the plain way: the arrays lie at irregular distances, which the per-request check must
    refuse.  == 2: every column pooled over indices / row ids and STAGED (fcp_stager_stage_ex: ids narrowed, row ids -> one
    [columns, rows + 1] matrix of offsets behind the copied inputs): mode 2.  Bit-exact against the oracle on the original
    request; batches that are not multiples of the block's rows, empty bags, 1-row batches, new shapes every request."""
    import torch
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    kind = seed % 3
    rng = np.random.default_rng(11000 + seed)
    monkeypatch.setenv("FCP_SEG_PREPASS", "1")
    if kind == 0:
        spec, tables, make = random_model(rng, n_groups_fixed=1, segs_fixed=("indices", "rowids32"),
                                          forms_fixed=(FORM_SEGMENT_REDUCE, FORM_SEGMENT_REDUCE, FORM_SEGMENT_REDUCE, FORM_GATHER,
                                                       FORM_GATHER_SCATTER, FORM_PASSTHROUGH))
    elif kind == 1:
        spec, tables, make = random_model(rng, n_groups_fixed=1, segs_fixed=("csr",),
                                          forms_fixed=(FORM_SEGMENT_REDUCE, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER))
    else:
        spec, tables, make = random_model(rng, extended=False, n_groups_fixed=1, segs_fixed=("indices", "rowids32"),
                                          forms_fixed=(FORM_SEGMENT_REDUCE,))
    dev = torch.device("cuda", 0)
    d_tabs = [torch.from_numpy(t).to(dev) for t in tables]
    st = None
    if kind == 2:
        sspec, modes, rows_col = spec.staged()
        assert all(m == 2 for i, m in enumerate(modes) if any(c.seg_input == i for c in spec.columns))   # every row-id input is converted
        op = FeatureColumnProcess(sspec, 0)
        st = RequestStager(64 << 20, max(spec.n_host_inputs, 1), max(sum(spec.host_input_ranks), 1), depth=3, n_threads=4)
    else:
        op = FeatureColumnProcess(spec, 0)
    for trial in range(5):
        batches = [int(rng.choice([1, 2, 3, 5, 33, 64, 130, 255, 257]))]
        inputs, symbols = make(rng, batches)
        packed = concat_inputs(inputs)
        if kind == 2:
            rows = [int(symbols[spec.columns[k].rows_arg]) if k >= 0 else 0 for k in rows_col]
            d_ptr, nbytes, offs, shps = st.stage_ex(inputs, modes, rows)
            seg_in = [c.seg_input for c in sorted(sspec.columns, key=lambda c: c.concat_slot)]
            assert len({int(offs[b]) - int(offs[a]) for a, b in zip(seg_in, seg_in[1:])}) <= 1      # one stride apart, in column order?
            out = op(_RawBlob(d_ptr, nbytes), offs, shps, d_tabs, symbols)
        else:
            blob, offsets, shapes = packed
            d_blob = torch.from_numpy(blob).to(dev) if blob.size else torch.empty(0, dtype=torch.int8, device=dev)
            out = op(d_blob, offsets, shapes, d_tabs, symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(spec.to_dict(), *packed, tables, symbols)
        got = out.groups[0].cpu().numpy()
        assert got.shape == want[0].shape and np.array_equal(got, want[0]), (seed, kind, trial)
    if st is not None:
        st.close()


class _RawBlob:
    def __init__(self, ptr, nbytes):
        self._p, self._n = ptr, nbytes

    def data_ptr(self):
        return self._p

    def numel(self):
        return self._n

    def element_size(self):
        return 1


@pytest.mark.parametrize("seed", range(int(os.environ.get("FCP_FUZZ_SEED0", "0")),
                                        int(os.environ.get("FCP_FUZZ_SEED0", "0")) + int(os.environ.get("FCP_FUZZ_STAGER_SEEDS", "6"))))
def test_random_plans_through_the_stager(oracle, seed):
    """Random plans, requests staged from host tensors with PlanSpec.staged() / fcp_stager_stage_ex (ids narrowed where the
    plan allows, sorted row ids -> CSR offsets on the host), copy and zero-copy rings: the staged plan on the staged blob
    equals the oracle on the original request."""
    import torch
    from recom_amd.ops import FeatureColumnProcess, RequestStager, concat_inputs
    rng = np.random.default_rng(7000 + seed)
    spec, tables, make = random_model(rng, dense_only=False)
    sspec, modes, rows_col = spec.staged()
    dev = torch.device("cuda", 0)
    d_tabs = [torch.from_numpy(t).to(dev) for t in tables]
    op = FeatureColumnProcess(sspec, 0)
    st = RequestStager(64 << 20, max(spec.n_host_inputs, 1), max(sum(spec.host_input_ranks), 1), depth=3, n_threads=4,
                       zero_copy=bool(seed % 2))
    for trial in range(4):
        batches = [int(rng.choice([1, 5, 64, 130, 257])) for _ in range(spec.n_groups)]
        inputs, symbols = make(rng, batches)
        rows = [int(symbols[spec.columns[k].rows_arg]) if k >= 0 else 0 for k in rows_col]
        d_ptr, nbytes, offs, shps = st.stage_ex(inputs, modes, rows)
        out = op(_RawBlob(d_ptr, nbytes), offs, shps, d_tabs, symbols)
        torch.cuda.synchronize()
        want, _ = oracle.process_feature_columns(spec.to_dict(), *concat_inputs(inputs), tables, symbols)
        for g, w in enumerate(want):
            got = out.groups[g].cpu().numpy()
            assert got.shape == w.shape and np.array_equal(got, w), (seed, trial, g)
    st.close()


def test_host_threads_share_a_plan_with_changing_shapes(oracle):
    """serve_workers: ten host threads (more than descriptor slots), one plan, one stream each; half of
    them bring a new shape on every request, the other half share three requests (slots hit from
    several streams, pinned while their launches run outside the plan mutex, evicted under contention)."""
    import torch
    from recom_amd import synth
    from recom_amd.ops import FeatureColumnProcess, concat_inputs
    m = synth.model_ragged(columns=48, vocab=3000, batch=96, seg="indices")
    tabs_np = m.numpy_tables()
    dev = torch.device("cuda", 0)
    tabs = [torch.from_numpy(t).to(dev) for t in tabs_np]
    op = FeatureColumnProcess(m.spec, 0)
    n_threads, per_thread = 10, 30
    # even threads: a new shape on every request; odd threads: all cycle over the SAME three requests, so
    # that one descriptor slot is used from several streams at once and then evicted by the others
    reqs = [[m.make_request(100 * t + k) if t % 2 == 0 else m.make_request(7000 + k % 3) for k in range(per_thread)]
            for t in range(n_threads)]
    packed = [[concat_inputs(r.inputs) for r in rs] for rs in reqs]
    blobs = [[torch.from_numpy(p[0]).to(dev) for p in ps] for ps in packed]
    torch.cuda.synchronize()
    results = [[None] * per_thread for _ in range(n_threads)]
    errors = []

    def worker(t):
        try:
            s = torch.cuda.Stream(device=dev)
            for k in range(per_thread):
                results[t][k] = op(blobs[t][k], packed[t][k][1], packed[t][k][2], tabs, reqs[t][k].symbols,
                                   stream=s.cuda_stream)  # ctypes drops the GIL: calls really overlap
            s.synchronize()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    torch.cuda.synchronize()
    spec_d = m.spec.to_dict()
    for t in range(n_threads):
        for k in range(per_thread):
            want, _ = oracle.process_feature_columns(spec_d, *packed[t][k], tabs_np, reqs[t][k].symbols)
            assert np.array_equal(results[t][k].groups[0].cpu().numpy(), want[0]), (t, k)
