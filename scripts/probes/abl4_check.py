import os, sys, dataclasses
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from recom_amd import synth
from recom_amd.ops import FeatureColumnProcess, concat_inputs
from recom_amd.plan import FLAG_COUNT_BAD_IDS
m = synth.model_s2(columns=1000, vocab=2000)
spec = dataclasses.replace(m.spec, flags=m.spec.flags | FLAG_COUNT_BAD_IDS)
op = FeatureColumnProcess(spec, 0)
tabs = m.torch_tables(torch.device("cuda", 0))
r = m.make_request(0)
blob, offs, shp = concat_inputs(r.inputs)
print("offsets head", offs[:12], "blob", blob.nbytes)
out = op(torch.from_numpy(blob).cuda(), offs, shp, tabs, r.symbols)
torch.cuda.synchronize()
print("bad ids", op.plan.read_bad_ids(), "nonzero frac", float((out.groups[0] != 0).float().mean()))
g = out.groups[0].cpu().numpy()
offs_c = spec.column_offsets()
for k in range(12):
    c = spec.columns[k]
    blk = g[:, offs_c[k]:offs_c[k] + c.dim]
    print(k, "dim", c.dim, "src", c.id_source, "nonzero rows", int((blk != 0).any(axis=1).sum()), "of", blk.shape[0], "nonzero elems frac", float((blk != 0).mean()),
          "rows pattern", "".join("1" if x else "0" for x in (blk != 0).any(axis=1)[:32]))
