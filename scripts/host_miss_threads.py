import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recom_amd import synth
from recom_amd.harness import ServingHarness
m = synth.model_ae("E", batch=2, large_rows=1 << 12)
threads = int(sys.argv[1])
h = ServingHarness(m, n_requests=16, n_threads=threads)
h.run(100)
wall, dev, _ = h.run(3000)
print(f"E b=2 new shapes, threads {threads}: {wall * 1e3 / (3000 * threads):6.2f} us per request (aggregate)")
h.close()
