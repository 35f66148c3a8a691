#!/usr/bin/env python3
"""Generates tests/golden/fcp_golden.npz — committed input/expected-output vectors.

The reference owns no golden vectors for this path and cannot
be built or imported here, so the expectations come from implementations that are
independent of both the C oracle and the HIP kernels and that define the TF op
semantics the reference targets:

  * NumPy (float64 accumulation): gather, searchsorted(side="right") == TF
    Bucketize, CSR segment sums / means;
  * PyTorch-CPU: ``torch.bucketize(right=True)``, ``F.embedding_bag(mode=sum|mean,
    include_last_offset=True)``, ``index_select`` — cross-checked against the NumPy
    result at generation time (this script asserts agreement before writing).

Run from the repo root:  python tests/golden/make_golden.py
Fixture = data only (inputs, tables, plan description, expected outputs).
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from recom_amd import synth  # noqa: E402
from recom_amd.plan import (COMBINER_MEAN, COMBINER_NONE, COMBINER_SUM, FORM_GATHER, FORM_GATHER_SCATTER,  # noqa: E402
                            FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE, IDS_I64, ROWS_FROM_IDS, ROWS_FROM_SYMBOL,
                            SEG_CSR_I32, SEG_IDS_I32, SEG_IDS_I64, SEG_NONE, ColumnSpec, PlanSpec)
import fcp_oracle as O  # noqa: E402  (np_* restatement only; the C oracle is NOT used to make expectations)

VOCAB = 211


def plan_json(spec: PlanSpec) -> str:
    d = spec.to_dict()
    for c in d["columns"]:
        c["boundaries"] = None if c["boundaries"] is None else [float(x) for x in c["boundaries"]]
    return json.dumps(d)


def pack(inputs):
    """ConcatInputs semantics in plain NumPy (byte concat, int32 offsets, dims)."""
    blobs, offsets, shapes, off = [], [], [], 0
    for a in inputs:
        a = np.ascontiguousarray(a)
        offsets.append(off)
        off += a.nbytes
        shapes.extend(a.shape)
        blobs.append(a.view(np.uint8).ravel())
    blob = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
    return blob.view(np.int8), np.asarray(offsets, np.int32), np.asarray(shapes, np.int32)


def torch_check(spec: PlanSpec, inputs, tables, symbols, expected):
    """Independent PyTorch-CPU evaluation of every lookup column."""
    offs = spec.column_offsets()
    for k, c in enumerate(spec.columns):
        if c.form not in (FORM_GATHER, FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
            continue
        W = torch.from_numpy(tables[c.table_input])
        raw = inputs[c.ids_input]
        if c.id_source == IDS_F32_BUCKETIZE:
            ids = torch.bucketize(torch.from_numpy(raw.astype(np.float32)),
                                  torch.from_numpy(np.asarray(c.boundaries, np.float32)), right=True).long()
        else:
            ids = torch.from_numpy(raw.astype(np.int64))
        exp = expected[c.concat_group][:, offs[k]:offs[k] + c.dim]
        if c.form == FORM_GATHER:
            got = W.index_select(0, ids.reshape(-1)).numpy()
            assert np.array_equal(got, exp.astype(np.float32)), f"column {k}: torch gather mismatch"
            continue
        rows = int(symbols[c.rows_arg])
        if c.seg_kind == SEG_CSR_I32:
            offsets = torch.from_numpy(inputs[c.seg_input].astype(np.int64))
        else:
            seg = inputs[c.seg_input].reshape(-1)[::c.seg_stride][:ids.numel()]
            offsets = torch.from_numpy(np.searchsorted(seg, np.arange(rows + 1)).astype(np.int64))
        if c.form == FORM_SEGMENT_REDUCE:
            mode = "mean" if c.combiner == COMBINER_MEAN else "sum"
            got = F.embedding_bag(ids.reshape(-1), W.double(), offsets, mode=mode, include_last_offset=True).numpy()
            assert np.allclose(got, exp, rtol=0, atol=1e-12), f"column {k}: torch embedding_bag mismatch"
        else:
            got = np.zeros((rows, c.dim))
            o = offsets.numpy()
            for r in range(rows):
                if o[r + 1] > o[r]:
                    got[r] = tables[c.table_input][int(ids[o[r + 1] - 1])]
            assert np.array_equal(got, exp), f"column {k}: scatter mismatch"


def make_case(name, spec, inputs, tables, symbols, store, tables_key=None):
    spec.validate()
    blob, offsets, shapes = pack(inputs)
    expected = O.np_process_feature_columns(spec.to_dict(), blob, offsets, shapes, tables, symbols)
    torch_check(spec, inputs, tables, symbols, expected)
    store[f"{name}/plan"] = np.frombuffer(plan_json(spec).encode(), np.uint8)
    store[f"{name}/n_inputs"] = np.asarray(len(inputs))
    for i, a in enumerate(inputs):
        store[f"{name}/in{i}"] = a
    store[f"{name}/blob"] = blob
    store[f"{name}/offsets"] = offsets
    store[f"{name}/shapes"] = shapes
    store[f"{name}/symbols"] = np.zeros(0, np.int32) if symbols is None else np.asarray(symbols, np.int32)
    # tables shared between cases are stored once under `tables_key`
    tk = tables_key or name
    store[f"{name}/tables_key"] = np.frombuffer(tk.encode(), np.uint8)
    store[f"{tk}/n_tables"] = np.asarray(len(tables))
    for i, t in enumerate(tables):
        store[f"{tk}/table{i}"] = t
    for g, e in enumerate(expected):
        store[f"{name}/expected{g}"] = e  # float64 truth
    # which output elements are pure copies (bit-exact contract)
    copy_cols = []
    offs = spec.column_offsets()
    for k, c in enumerate(spec.columns):
        if c.form not in (FORM_SEGMENT_REDUCE, 5):  # 5 = BatchColReduction (an fp32 sum)
            copy_cols.append((c.concat_group, offs[k], c.dim))
    store[f"{name}/copy_cols"] = np.asarray(copy_cols, np.int32).reshape(-1, 3)
    print(f"{name}: {len(spec.columns)} columns, blob {blob.nbytes} B, groups {[e.shape for e in expected]}")


def main():
    store = {}
    # --- mixed model, random requests ---------------------------------------------
    m = synth.model_mixed(batch=33, vocab=VOCAB)
    tables = m.numpy_tables()
    for seed in (0, 1):
        req = m.make_request(seed)
        make_case(f"mixed_s{seed}", m.spec, req.inputs, tables, req.symbols, store, "mixed_tables")
    # --- the same model with every ragged row empty --------------------------------
    req = m.make_request(2)
    inputs = []
    for a, r, e in zip(req.inputs, m.spec.host_input_ranks, m.spec.host_input_elem_sizes):
        inputs.append(a)
    for c in m.spec.columns:
        if c.form in (FORM_SEGMENT_REDUCE, FORM_GATHER_SCATTER):
            inputs[c.ids_input] = np.zeros(0, np.int64)
            if c.seg_kind == SEG_CSR_I32:
                inputs[c.seg_input] = np.zeros(m.batch + 1, np.int32)
            elif c.seg_kind == SEG_IDS_I64:
                inputs[c.seg_input] = np.zeros((0, 2), np.int64)
            else:
                inputs[c.seg_input] = np.zeros(0, np.int32)
    make_case("mixed_empty", m.spec, inputs, tables, req.symbols, store, "mixed_tables")

    # --- bucketize known-answer test -------------------------------------------------
    bnd = synth.MICROBENCH_BOUNDARIES
    vals = np.concatenate([bnd, np.nextafter(bnd, -np.inf), np.nextafter(bnd, np.inf),
                           np.asarray([-1.0, -np.inf, np.inf, 1e9, -1e9, 494.999, 495.0, 495.001, 2.5, 0.0, -0.0],
                                      np.float32)]).astype(np.float32)
    spec = PlanSpec([ColumnSpec(FORM_GATHER, 8, 101, COMBINER_NONE, IDS_F32_BUCKETIZE, 0, 0, -1, SEG_NONE, 1,
                                ROWS_FROM_IDS, 0, bnd, 0, 0)], [1], [4], 1)
    t = [synth.hash_table_numpy(7, 101, 8)]
    make_case("bucketize_kat", spec, [vals], t, None, store)
    store["bucketize_kat/expected_buckets"] = np.searchsorted(bnd, vals, side="right").astype(np.int32)

    # --- ragged edge cases --------------------------------------------------------------
    lens = np.asarray([0, 1, 64, 65, 0, 0, 130, 1, 0], np.int64)
    B, nnz = len(lens), int(lens.sum())
    rng = np.random.Generator(np.random.PCG64(99))
    ids = rng.integers(0, VOCAB, size=nnz, dtype=np.int64)
    ids[0] = 0
    ids[1] = VOCAB - 1
    ids[5:9] = 17  # repeated ids inside one bag
    rows = np.repeat(np.arange(B, dtype=np.int64), lens)
    csr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    pos = np.concatenate([np.arange(l, dtype=np.int64) for l in lens])
    indices = np.stack([rows, pos], axis=1)
    cols = [
        ColumnSpec(FORM_SEGMENT_REDUCE, 8, VOCAB, COMBINER_SUM, IDS_I64, 0, 0, 1, SEG_CSR_I32, 1, ROWS_FROM_SYMBOL, 0, None, 0, 0),
        ColumnSpec(FORM_SEGMENT_REDUCE, 16, VOCAB, COMBINER_MEAN, IDS_I64, 1, 0, 2, SEG_IDS_I64, 2, ROWS_FROM_SYMBOL, 0, None, 0, 1),
        ColumnSpec(FORM_SEGMENT_REDUCE, 64, VOCAB, COMBINER_MEAN, IDS_I64, 2, 0, 3, SEG_IDS_I32, 1, ROWS_FROM_SYMBOL, 0, None, 0, 2),
    ]
    spec = PlanSpec(cols, [1, 1, 2, 1], [8, 4, 8, 4], 3, n_symbols=1)
    t = [synth.hash_table_numpy(11 + i, VOCAB, d) for i, d in enumerate((8, 16, 64))]
    make_case("ragged_edges", spec, [ids, csr, indices, rows.astype(np.int32)], t, np.asarray([B], np.int32), store)

    # --- scatter (form 3): at most one id per row, some rows absent ------------------------
    B = 12
    present = np.asarray([0, 3, 4, 7, 11], np.int64)
    ids = np.asarray([5, 0, VOCAB - 1, 17, 17], np.int64)
    indices = np.stack([present, np.zeros_like(present)], axis=1)
    spec = PlanSpec([ColumnSpec(FORM_GATHER_SCATTER, 12, VOCAB, COMBINER_NONE, IDS_I64, 0, 0, 1, SEG_IDS_I64, 2,
                                ROWS_FROM_SYMBOL, 0, None, 0, 0)], [1, 2], [8, 8], 1, n_symbols=1)
    t = [synth.hash_table_numpy(21, VOCAB, 12)]
    make_case("scatter", spec, [ids, indices], t, np.asarray([B], np.int32), store)

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fcp_golden.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
