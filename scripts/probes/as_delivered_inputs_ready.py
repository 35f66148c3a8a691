import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recom_amd import synth
from recom_amd.harness import ServingHarness
for name, model in (("ragged as delivered", synth.model_ragged(seg="indices")), ("model E as delivered", synth.model_ae("e"))):
    for ready in (False, True, False, True):
        h = ServingHarness(model, n_requests=64, arena_ring=6, n_threads=1)
        assert h.verify_resident()["checked"] > 0
        if ready:
            h.plan.set_inputs_ready(True)
        h.run(200)
        print(name, "inputs_ready" if ready else "stream order", round(h.run(1200)[0] * 1e3 / 1200, 2), "us per request")
        h.close()
