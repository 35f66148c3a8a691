#!/usr/bin/env python3
"""Where do the reference's AE models E / F spend their request?  Times the whole model,
its one-hot (dense-kernel) columns alone and its multi-hot (ragged-kernel) columns alone.
Run on a GPU box:  python scripts/ae_split.py [e|f] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402


def timed(model, steps):
    h = ServingHarness(model, n_requests=8)
    h.run(50)
    wall, dev, _ = h.run(steps)
    b = h.algorithmic_bytes()["total"]
    h.close()
    return dev * 1e3 / steps, wall * 1e3 / steps, b


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "e"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    m = synth.model_ae(which.upper())
    if os.environ.get("AE_STAGED", "1") != "0":          # the form the rewritten graph's ConcatInputs leaves (the bench default)
        m = synth.staged_model(m)
    dense = [k for k, c in enumerate(m.spec.columns) if c.form in (1, 4)]
    ragged = [k for k, c in enumerate(m.spec.columns) if c.form not in (1, 4)]
    small = [k for k in ragged if m.spec.columns[k].vocab < (1 << 20)]
    large = [k for k in ragged if m.spec.columns[k].vocab >= (1 << 20)]
    parts = (("all", None), ("one-hot", dense), ("multi-hot", ragged), ("multi-hot small tables", small),
             ("multi-hot large tables", large))
    if os.environ.get("AE_PARTS"):
        parts = [p for p in parts if p[0] in os.environ["AE_PARTS"].split(",")]
    for name, keep in parts:
        if keep is not None and not keep:
            continue
        mm = m if keep is None else synth.submodel(m, keep)
        dev, wall, b = timed(mm, steps)
        print(f"{which.upper()} {name:24s} cols {mm.spec.n_columns:5d}  dev {dev:6.2f} us  wall {wall:6.2f} us  "
              f"{b / 1e6:6.2f} MB/request")


if __name__ == "__main__":
    main()
