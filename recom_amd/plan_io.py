"""Column-plan files: what the `dlpath` attr of ``Addons>FeatureColumnProcess``
points at in this build (the reference points it at a JIT-compiled ``.so``,
``feature_column_process_op_gpu.cu.cc:49-55``).  Plain text, parsed by
``tf_shim/fcp_tf_ops.cc::LoadPlanFile`` and by :func:`load_plan`.
"""
from __future__ import annotations

import numpy as np

from .plan import ColumnSpec, PlanSpec, StageInfo


def save_plan(spec: PlanSpec, path: str, stage: "StageInfo | None" = None) -> None:
    """``stage``: the stage section (version 3 files) — how ``Addons>ConcatInputs`` packs each of its inputs for this
    (staged) plan; one entry per host input of ``spec``."""
    spec.validate()
    if stage is not None and len(stage.modes) != spec.n_host_inputs:
        raise ValueError("the stage section lists one entry per host input of the plan")
    with open(path, "w") as f:
        maps = [(k, c) for k, c in enumerate(spec.columns) if len(c.seg_mul)]
        f.write(f"fcp_plan {4 if maps else 3 if stage is not None else 2}\n")
        f.write(f"layout {spec.layout}\n")
        f.write(f"groups {spec.n_groups} symbols {spec.n_symbols} device_inputs {spec.n_device_inputs}\n")
        f.write(f"host_inputs {spec.n_host_inputs}\n")
        for r, e in zip(spec.host_input_ranks, spec.host_input_elem_sizes):
            f.write(f"{r} {e}\n")
        f.write(f"columns {spec.n_columns}\n")
        for c in spec.columns:
            b = [] if c.boundaries is None else [repr(float(np.float32(x))) for x in c.boundaries]
            f.write(" ".join(str(x) for x in (
                c.form, c.combiner, c.dim, c.id_source, c.vocab, c.table_input, c.ids_input, c.seg_input,
                c.seg_kind, c.seg_stride, c.rows_source, c.rows_arg, c.concat_group, c.concat_slot, len(b))))
            f.write(" " + " ".join(b) if b else "")
            # version 2: the id transform — mode, number of intervals, substitute, hash buckets, (lo, hi) pairs
            x = [c.xform_mode, len(c.xform_lo), c.xform_substitute, c.hash_buckets] + \
                [v for p in zip(c.xform_lo, c.xform_hi) for v in p]
            f.write(" " + " ".join(str(int(v)) for v in x) + "\n")
        if maps:     # version 4: segment-id maps — column, coordinates, symbol, symbol slot, mul0..mul3, div
            f.write(f"segmaps {len(maps)}\n")
            for k, c in maps:
                mul = [int(v) for v in c.seg_mul] + [0] * (4 - len(c.seg_mul))
                f.write(" ".join(str(int(v)) for v in [k, len(c.seg_mul), c.seg_sym, c.seg_sym_slot] + mul + [c.seg_div]) + "\n")
        if stage is not None:
            f.write(f"stage {len(stage.modes)} symbols_input {stage.symbols_input}\n")
            for m, k in zip(stage.modes, stage.rows_symbol):
                f.write(f"{int(m)} {int(k)}\n")


def load_plan(path: str) -> PlanSpec:
    tok = open(path).read().split()
    it = iter(tok)

    def nxt():
        try:
            return next(it)
        except StopIteration:
            raise ValueError(f"truncated column plan {path}") from None

    if nxt() != "fcp_plan":
        raise ValueError("bad plan header")
    version = int(nxt())
    if version not in (1, 2, 3, 4):
        raise ValueError("bad plan header")
    assert nxt() == "layout"
    layout = int(nxt())
    assert nxt() == "groups"
    n_groups = int(nxt())
    assert nxt() == "symbols"
    n_symbols = int(nxt())
    assert nxt() == "device_inputs"
    n_dev = int(nxt())
    assert nxt() == "host_inputs"
    n_host = int(nxt())
    ranks, esz = [], []
    for _ in range(n_host):
        ranks.append(int(nxt()))
        esz.append(int(nxt()))
    assert nxt() == "columns"
    cols = []
    for _ in range(int(nxt())):
        v = [int(nxt()) for _ in range(15)]
        b = np.asarray([float(nxt()) for _ in range(v[14])], np.float32) if v[14] else None
        mode, lo, hi, sub, hb = 0, [], [], 0, 0
        if version >= 2:
            mode, n, sub, hb = int(nxt()), int(nxt()), int(nxt()), int(nxt())
            for _ in range(n):
                lo.append(int(nxt()))
                hi.append(int(nxt()))
        cols.append(ColumnSpec(form=v[0], combiner=v[1], dim=v[2], id_source=v[3], vocab=v[4], table_input=v[5],
                               ids_input=v[6], seg_input=v[7], seg_kind=v[8], seg_stride=v[9], rows_source=v[10],
                               rows_arg=v[11], concat_group=v[12], concat_slot=v[13], boundaries=b,
                               xform_mode=mode, xform_lo=tuple(lo), xform_hi=tuple(hi), xform_substitute=sub,
                               hash_buckets=hb))
    rest = list(it)
    if version >= 4 and rest[:1] == ["segmaps"]:
        import dataclasses
        for j in range(int(rest[1])):
            v = [int(x) for x in rest[2 + 9 * j: 11 + 9 * j]]
            cols[v[0]] = dataclasses.replace(cols[v[0]], seg_mul=tuple(v[4:4 + v[1]]), seg_div=v[8], seg_sym=v[2],
                                             seg_sym_slot=v[3])
    spec = PlanSpec(cols, ranks, esz, n_dev, n_groups=n_groups, n_symbols=n_symbols, layout=layout)
    spec.validate()
    return spec


def load_stage(path: str) -> "StageInfo | None":
    """The stage section of a plan file as the LIBRARY parses it (``fcp_plan_file_stage_info`` — what the shim's
    ConcatInputsOp calls with the node's ``_fcp_plan`` attr), or None: a plain plan."""
    import ctypes as C
    from . import lib as _lib
    L = _lib.load()
    n, sym_in = C.c_int32(0), C.c_int32(-1)
    _lib.check(L.fcp_plan_file_stage_info(path.encode(), C.byref(n), None, None, 0, C.byref(sym_in)), "fcp_plan_file_stage_info")
    if n.value == 0:
        return None
    modes = np.zeros(n.value, np.uint8)
    rows = np.zeros(n.value, np.int32)
    _lib.check(L.fcp_plan_file_stage_info(path.encode(), C.byref(n), modes.ctypes.data, rows.ctypes.data, n.value,
                                          C.byref(sym_in)), "fcp_plan_file_stage_info")
    return StageInfo([int(m) for m in modes], [int(r) for r in rows], int(sym_in.value))
