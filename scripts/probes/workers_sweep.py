import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
from recom_amd import synth
from recom_amd.harness import ServingHarness
model = synth.model_s2()
base = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
base.run(100)
print("queues", os.environ["GPU_MAX_HW_QUEUES"], "1 worker", round(base.run(600)[0] * 1e3 / 600, 2))
for n in (2, 3, 4, 5, 6, 8):
    h = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=n, tables=base.tables)
    h.run(100)
    print(n, "workers", round(h.run(600)[0] * 1e3 / (600 * n), 2), "us per request")
    h.close()
