"""SURVEY.md section 8 row f-4 (the shim against a real TensorFlow, the reference's --embedding_only A/B): runs wherever a
TF-ROCm wheel exists, skips cleanly everywhere else (this pool has no TensorFlow: profiles/r03_gpu_box_tensorflow_probe.txt).
The CPU half checks what CAN be checked here: the script skips with its documented exit code, and the S1 / S2 GraphDefs it
would export go through the plan builder into the expected column plans and evaluate (NumPy, TF-CPU semantics) to what the
oracle computes from those plans."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "tf_validate.py")


def _has_tensorflow():
    import importlib.util
    return importlib.util.find_spec("tensorflow") is not None


def test_script_skips_cleanly_without_tensorflow():
    if _has_tensorflow():
        pytest.skip("TensorFlow is importable: the GPU test below runs the script for real")
    res = subprocess.run([sys.executable, SCRIPT, "--model", "s1"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 77, res.stderr[-2000:]
    rec = json.loads(res.stdout.strip().splitlines()[-1])
    assert rec["skipped"] is True and "TensorFlow" in rec["reason"]


@pytest.mark.parametrize("which", ["s1", "s2"])
def test_the_graphs_the_script_exports_build_into_the_expected_plans(which, oracle, tmp_path):
    """The generator of scripts/tf_validate.py (tests/graph_fixtures.py s1_model / s2_model) -> plan builder -> rewritten graph
    with the ORACLE behind the three Addons> ops equals the op-by-op NumPy evaluation of the original graph (TF-CPU
    semantics) bit for bit."""
    import graph_fixtures as F
    from test_graph_plan import oracle_ops
    from tf_graph_eval import GraphEvaluator
    from recom_amd.graph import build_plan, parse_graphdef, rewrite_graph
    from recom_amd.plan import FORM_GATHER, FORM_SEGMENT_REDUCE, IDS_F32_BUCKETIZE
    from recom_amd.plan_io import save_plan
    if which == "s1":
        gd, feeds, variables, fetches = F.s1_model(columns=10, vocab=300, B=17)
    else:
        gd, feeds, variables, fetches = F.s2_model(columns=20, vocab=500, B=33)
    want = GraphEvaluator(gd, variables).run(fetches, feeds)
    built = build_plan(gd, "passthrough")
    spec = built.spec
    if which == "s1":
        assert [c.form for c in spec.columns] == [FORM_GATHER, FORM_SEGMENT_REDUCE] * 5 and not built.skipped
    else:
        assert all(c.form == FORM_GATHER for c in spec.columns) and not built.skipped
        assert [c.id_source == IDS_F32_BUCKETIZE for c in spec.columns] == [k % 10 == 0 for k in range(20)]
        assert [c.dim for c in spec.columns[:4]] == [8, 16, 32, 64]
    path = str(tmp_path / "m.fcp")
    save_plan(spec, path)
    out = parse_graphdef(rewrite_graph(gd, built, path).SerializeToString())
    got = GraphEvaluator(out, variables, oracle_ops(oracle, built, variables)).run(fetches, feeds)
    assert len(got) == len(want) == 1 and np.array_equal(got[0], want[0])


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["s1", "microbenchmark", "canonical"])
def test_rewritten_graph_through_the_shim_equals_tensorflow_cpu(model):
    """The A/B itself; needs TensorFlow-ROCm (skips here)."""
    pytest.importorskip("tensorflow")
    res = subprocess.run([sys.executable, SCRIPT, "--model", model, "--seconds", "1"], capture_output=True, text=True, timeout=1800)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    rec = json.loads(res.stdout.strip().splitlines()[-1])
    assert rec["parity"]["ok"] and rec["parity"]["copy_max_abs_diff"] == 0.0 and rec["parity"]["pooled_max_abs_diff"] < 1e-5
    assert rec["cpu_baseline"]["kind"] == "tensorflow" and rec["cpu_baseline"]["value"] > 0
