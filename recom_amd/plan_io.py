"""Column-plan files: what the `dlpath` attr of ``Addons>FeatureColumnProcess``
points at in this build (the reference points it at a JIT-compiled ``.so``,
``feature_column_process_op_gpu.cu.cc:49-55``).  Plain text, parsed by
``tf_shim/fcp_tf_ops.cc::LoadPlanFile`` and by :func:`load_plan`.
"""
from __future__ import annotations

import numpy as np

from .plan import ColumnSpec, PlanSpec


def save_plan(spec: PlanSpec, path: str) -> None:
    spec.validate()
    with open(path, "w") as f:
        f.write("fcp_plan 2\n")
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
    if version not in (1, 2):
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
    spec = PlanSpec(cols, ranks, esz, n_dev, n_groups=n_groups, n_symbols=n_symbols, layout=layout)
    spec.validate()
    return spec
