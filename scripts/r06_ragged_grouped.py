#!/usr/bin/env python3
"""Round 6: RAGGED with its CSR inputs as delivered (ids, offsets alternating in the blob) against the same requests with the
CSR inputs grouped last in column order (regular in the blob -> FcpLaunch::csr_reg mode 2), raw (int64 ids) and staged
(int32 ids); interleaved, HIP-event us per request."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recom_amd import synth
from recom_amd.harness import ServingHarness
base = synth.model_ragged(seg="csr")
staged = synth.staged_model(synth.model_ragged(seg="indices"))
models = {"csr as delivered": base, "csr grouped": synth.grouped_csr_model(base),
          "staged": staged, "staged grouped": synth.grouped_csr_model(staged)}
hs, tables = {}, None
for k, m in models.items():
    hs[k] = ServingHarness(m, n_requests=64, arena_ring=1, tables=tables)
    tables = hs[k].tables
    assert hs[k].verify_resident()["checked"] > 0
    hs[k].run(200)
for rnd in range(3):
    for k, h in hs.items():
        _, dev, _ = h.run(1500)
        print(f"round {rnd} RAGGED {k:18s}: {dev * 1e3 / 1500:6.2f} us per request")
