#!/usr/bin/env python3
"""Phase timeline of the ragged kernel from the diagnostic build (build/stamps, -DFCP_STAMPS).
GPU box:  FCP_LIB_DIR=$GRAFT_REPO_ROOT/build/stamps python scripts/ragged_stamps.py [ragged|e-multihot]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recom_amd import lib, synth  # noqa: E402
from recom_amd.harness import ServingHarness  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "ragged"
if which == "ragged":
    m = synth.model_ragged()
elif which == "ragged-staged":
    m = synth.staged_model(synth.model_ragged(seg="indices"))
elif which == "ragged-indices":
    m = synth.model_ragged(seg="indices")
else:
    e = synth.model_ae("E")
    m = synth.submodel(e, [k for k, c in enumerate(e.spec.columns) if c.form not in (1, 4)])
h = ServingHarness(m, n_requests=16)
h.run(50)
_, dev, _ = h.run(200)
L = lib.load()
nb = 65536
st = np.zeros(8 * nb, np.uint64)
L.fcp_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert L.fcp_debug_read_stamps(h.plan.handle, st.ctypes.data, nb) == 0
st = st.reshape(nb, 8).astype(np.int64)
live = st[st[:, 3] > 0]
t0 = live[:, 0].min()
span = (live[:, 3].max() - t0) / 100.0
print(f"{m.name}: {dev * 1e3 / 200:.2f} us/request; last launch: {len(live)} blocks (first row of each), span {span:.2f} us")
names = ["desc", "seg search", "ranges+scan", "ids", "bag walk+store"]
edges = [(0, 1), (1, 4), (4, 5), (5, 2), (2, 3)]
binw = 200 if span > 16 else 50
for b0 in range(0, int(span * 100) + binw, binw):
    sel = live[(live[:, 0] - t0 >= b0) & (live[:, 0] - t0 < b0 + binw)]
    ends = ((live[:, 3] - t0 >= b0) & (live[:, 3] - t0 < b0 + binw)).sum()
    if len(sel) == 0 and ends == 0:
        continue
    line = f"  t={b0 / 100:5.1f} us  begin {len(sel):5d}  end {ends:5d}"
    if len(sel):
        line += "   mean us: " + "  ".join(f"{n} {((sel[:, b] - sel[:, a]) / 100.0).mean():.2f}" for n, (a, b) in zip(names, edges))
    print(line)
