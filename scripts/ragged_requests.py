import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recom_amd import synth
from recom_amd.harness import ServingHarness
m = synth.model_ragged()
for n_req in (1, 4, 8, 9, 16):
    h = ServingHarness(m, n_requests=n_req)
    h.run(100)
    wall, dev, _ = h.run(1000)
    print(f"RAGGED distinct requests {n_req:2d} ({'cached descriptors' if n_req <= 8 else 'descriptor miss every request'}): dev {dev:6.2f} us")
    h.close()
