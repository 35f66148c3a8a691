#!/usr/bin/env python3
"""Round 6: model E's timed loop in bench.py's MAIN harness reads 12.4-13.3 us per request, the side harnesses created later in the
same process 10.6.  Is it the harness created FIRST, the loop timed first, or the closed-form check (verify_resident)?"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from recom_amd import synth
from recom_amd.harness import ServingHarness
m = synth.staged_model(synth.model_ae("e"))
a = ServingHarness(m, n_requests=64, arena_ring=1)
b = ServingHarness(m, n_requests=64, arena_ring=1, tables=a.tables)
def t(h, n=2000):
    h.run(200)
    return round(h.run(n)[1] * 1e3 / n, 2)
print("A (created first), B:", t(a), t(b), "| again:", t(a), t(b))
print("A after verify_resident:", a.verify_resident()["checked"], t(a), "B:", t(b))
c = ServingHarness(m, n_requests=64, arena_ring=1, tables=a.tables)
print("C (created last):", t(c), "A:", t(a))
print("B with verify first:", b.verify_resident()["checked"], t(b), t(b))
