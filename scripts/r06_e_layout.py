#!/usr/bin/env python3
"""Round 6: model E staged — requests resident in input order against the staged op's two-region layout (row offsets behind the
other inputs)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from recom_amd import synth
from recom_amd.harness import ServingHarness
for name in ("e", "f"):
    m = synth.staged_model(synth.model_ae(name))
    modes = m.stage_modes
    a = ServingHarness(m, n_requests=64, arena_ring=1)
    m.stage_modes = None
    b = ServingHarness(m, n_requests=64, arena_ring=1, tables=a.tables)
    m.stage_modes = modes
    for rnd in range(3):
        out = []
        for h in (a, b):
            h.run(200)
            out.append(round(h.run(2000)[1] * 1e3 / 2000, 2))
        print(f"model {name} round {rnd}: staged layout {out[0]} us, input order {out[1]} us")
    # resident shapes only (8 requests <= 32 slots): the GPU side alone
    a8 = ServingHarness(m, n_requests=8, arena_ring=1, tables=a.tables)
    m.stage_modes = None
    b8 = ServingHarness(m, n_requests=8, arena_ring=1, tables=a.tables)
    m.stage_modes = modes
    for h, nm in ((a8, "staged layout"), (b8, "input order")):
        h.run(200)
        print(f"model {name} 8 resident requests, {nm}: {round(h.run(2000)[1] * 1e3 / 2000, 2)} us")
    for h in (a, b, a8, b8):
        h.close()
