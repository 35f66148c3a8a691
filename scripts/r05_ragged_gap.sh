#!/bin/bash
# Round 5 (last part): where does the staged RAGGED kernel's time go?  Product against the timing-only build
# without table reads (-DFCP_ABLATE=4: every table read hits one hot zero line: everything but the gather traffic), interleaved;
# then the block timeline of the stamps build.
cd "$GRAFT_REPO_ROOT"
make -C recom_amd/csrc OUT=../../build/abl4 DEFS="-DFCP_ABLATE=4" >/dev/null 2>&1 || echo "abl4 build failed"
make -C recom_amd/csrc OUT=../../build/stamps DEFS="-DFCP_STAMPS" >/dev/null 2>&1 || echo "stamps build failed"
one() { python - "$@" <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from recom_amd import synth
from recom_amd.harness import ServingHarness
m = synth.staged_model(synth.model_ragged(seg="indices"))
h = ServingHarness(m, n_requests=64)
h.run(100)
best = min(h.run(1000)[1] for _ in range(3))
print(f"{os.environ.get('FCP_LIB_DIR', 'product'):>28s}: {best:.2f} us per request (staged RAGGED, 64 requests cycled)")
PY
}
for r in 1 2 3; do one; FCP_LIB_DIR=$GRAFT_REPO_ROOT/build/abl4 one; done 2>/dev/null
FCP_LIB_DIR=$GRAFT_REPO_ROOT/build/stamps python scripts/ragged_stamps.py ragged-staged 2>/dev/null
