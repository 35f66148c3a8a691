"""One S2 harness under rocprofv3 --kernel-trace: mode = workers3 | lanes3 | lanes4 | inputs_ready | stream.  Prints nothing;
scripts/r04_private_trace_summarize.py reads the kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from recom_amd import synth
from recom_amd.harness import ServingHarness
mode = sys.argv[1]
model = synth.model_s2()
if mode.startswith("workers"):
    n = int(mode[7:])
    h = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=n)
    h.run(60); h.run(200)
else:
    h = ServingHarness(model, n_requests=16, arena_ring=8, n_threads=1)
    if mode.startswith("lanes"):
        n = int(mode[5:])
        h.plan.set_private_streams(n)
        h.run_private(100, n); h.run_private(400, n)
    else:
        if mode == "inputs_ready":
            h.plan.set_inputs_ready(True)
        h.run(100); h.run(400)
h.close()
