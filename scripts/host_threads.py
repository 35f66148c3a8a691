import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from recom_amd import synth
from recom_amd.harness import ServingHarness
m = synth.model_s2(batch=2, vocab=1000)
for threads in (1, 2, 3, 4):
    h = ServingHarness(m, n_requests=4, n_threads=threads)
    h.run(100)
    wall, dev, _ = h.run(3000)
    print(f"S2 b=2 cached shapes, threads {threads}: {wall * 1e3 / (3000 * threads):6.2f} us per request (aggregate)")
    h.close()
