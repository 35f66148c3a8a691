"""Plans whose segment ids come through a folded SparseReshape (fcp_column_ext_t::seg_map_*; the reference's
EmitInputInline SparseReshape case, cuda_emitter.cc:1874-1916), next to the equivalent PLAIN plan whose segment ids were
computed here with NumPy (`np.ravel_multi_index` over the input shape, `np.unravel_index` over the output shape — the
definition of SparseReshape).  Test data only."""
import dataclasses

import numpy as np

from recom_amd.plan import (COMBINER_MEAN, COMBINER_SUM, FORM_SEGMENT_REDUCE, IDS_I32, IDS_I64, ROWS_FROM_SYMBOL,
                            SEG_IDS_I32, SEG_IDS_I64, ColumnSpec, PlanSpec)

# (input shape with B first, output shape, seg_mul, seg_div, which factor is a symbol: None / ("mul", k) / ("div",))
#   the symbol's value is the factor itself; the plan then stores 1 in its place
RESHAPES = [
    ((None, 5, 7), (-1, 7), (5, 1), 1, None),            # [B, T, L] -> [B*T, L], T static (safe_embedding_lookup_sparse, rank 3)
    ((None, 5, 7), (-1, 7), (5, 1), 1, ("mul", 0)),      # the same with T known per request only
    ((None, 6), (-1, 18), (1,), 3, None),                # [B*3, 6] -> [B, 18]: seg = idx0 // 3
    ((None, 6), (-1, 18), (1,), 3, ("div",)),
    ((None, 3, 2, 9), (-1, 9), (6, 2, 1), 1, None),      # [B, T, U, L] -> [B*T*U, L]
    ((None, 4, 6), (-1, 12), (24, 6, 1), 12, None),      # [B, 4, 6] -> [B*2, 12], the unreduced expression of the reference
    ((None, 4, 6), (-1, 12), (4, 1), 2, None),           # the same after cancelling L = 6
]


def build(seed=0, batch=11, vocab=300, max_nnz=400, seg64=True, dims=(8, 12, 16, 20, 32, 64, 4)):
    """-> (mapped PlanSpec, plain PlanSpec, mapped inputs, plain inputs, tables, symbols)"""
    rng = np.random.default_rng(seed)
    cols_m, cols_p, ins_m, ins_p, tables, symbols = [], [], [], [], [], []
    ranks_m, ranks_p, esz = [], [], []
    for k, (ishape, oshape, mul, div, symwhere) in enumerate(RESHAPES):
        o_tail, i_tail = int(np.prod(oshape[1:])), int(np.prod(ishape[1:]))
        lead = (batch + k) * (o_tail // np.gcd(o_tail, i_tail))          # so that the element count divides the output tail
        ishape = (int(lead),) + tuple(ishape[1:])
        total = int(np.prod(ishape))
        assert total % o_tail == 0
        oshape = (total // o_tail,) + tuple(oshape[1:])
        rows = oshape[0]
        nnz = int(rng.integers(0, max_nnz))
        flat = np.sort(rng.choice(total, size=min(nnz, total), replace=False))          # sorted = lexicographic order
        coords = np.stack(np.unravel_index(flat, ishape), axis=1).astype(np.int64 if seg64 else np.int32)
        seg = np.unravel_index(flat, oshape)[0].astype(coords.dtype)                   # SparseReshape, row coordinate
        ids = rng.integers(-2, vocab + 2, flat.size).astype(np.int64 if k % 2 else np.int32)
        dim = dims[k % len(dims)]
        tables.append(rng.standard_normal((vocab, dim)).astype(np.float32))
        sym_rows = len(symbols)
        symbols.append(rows)
        seg_sym, slot, mul, div = -1, 0, list(mul), div
        if symwhere is not None:
            seg_sym = len(symbols)
            if symwhere[0] == "mul":
                slot = symwhere[1]
                symbols.append(mul[slot])
                mul[slot] = 1
            else:
                slot = 4
                symbols.append(div)
                div = 1
        base = dict(form=FORM_SEGMENT_REDUCE, dim=dim, vocab=vocab, combiner=COMBINER_MEAN if k % 3 else COMBINER_SUM,
                    id_source=IDS_I64 if k % 2 else IDS_I32, table_input=k, ids_input=2 * k, seg_input=2 * k + 1,
                    seg_kind=SEG_IDS_I64 if seg64 else SEG_IDS_I32, rows_source=ROWS_FROM_SYMBOL, rows_arg=sym_rows,
                    concat_group=k, concat_slot=0)
        cols_m.append(ColumnSpec(seg_stride=len(ishape), seg_mul=tuple(mul), seg_div=div, seg_sym=seg_sym, seg_sym_slot=slot,
                                 **base))
        cols_p.append(ColumnSpec(seg_stride=1, **base))
        ins_m += [ids, coords]
        ins_p += [ids, seg]
        ranks_m += [1, 2]
        ranks_p += [1, 1]
        esz += [ids.dtype.itemsize, coords.dtype.itemsize]
    n = len(RESHAPES)
    mk = lambda cols, ranks: PlanSpec(cols, ranks, list(esz), n_device_inputs=n, n_groups=n, n_symbols=len(symbols))
    return mk(cols_m, ranks_m), mk(cols_p, ranks_p), ins_m, ins_p, tables, np.asarray(symbols, np.int32)
