#!/usr/bin/env python3
"""SURVEY.md section 8 row f-4, made executable: the shim compiled against a REAL TensorFlow, the rewritten graph run through
it, and the reference's `--embedding_only` A/B (examples/cc/recom_examples.patch:161-165, 3506-3519: fetch every ConcatV2 of
the embedding stage once from plain TensorFlow on the CPU and once from the rewritten graph) asserted:

  * copy columns (GatherV2 / ScatterNd / passthrough): bit-exact;
  * pooled columns (SparseSegmentSum / Mean): max-abs-diff < 1e-5 (BASELINE.json north_star).

It also times TensorFlow-CPU on the same graph (`cpu_baseline.kind = "tensorflow"`, what SURVEY 8d prefers over the C port
bench.py times where TensorFlow is absent).

Nothing in this pool has TensorFlow (profiles/r03_gpu_box_tensorflow_probe.txt), so this file has never run end to end:
`python scripts/tf_validate.py` exits 77 ("skipped") without TensorFlow and tests/test_tf_validate.py skips.  Wherever a
TF-ROCm wheel exists it runs unattended:

    python scripts/tf_validate.py [--model s1|microbenchmark|canonical|s2] [--columns N] [--batch B] [--vocab V]
                                  [--seconds S] [--baseline-only] [--threads 1,8,32]

Steps: (1) build recom_amd/tf_shim/librecom_fcp.so with the command at the top of fcp_tf_ops.cc; (2) generate the model's
GraphDef with THIS repository's generator (tests/graph_fixtures.py — not the reference's Python files); (3) TF-CPU: import
the GraphDef, assign the tables, fetch the concat(s); (4) `python -m recom_amd.graph` (build_plan + rewrite_graph): plan
file + rewritten GraphDef; (5) load the shim (`tf.load_op_library`), import the rewritten graph on the GPU, assign the same
tables, fetch the same tensors; (6) compare; (7) time TF-CPU.  Prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SKIPPED = 77


def import_tensorflow():
    try:
        import tensorflow as tf  # noqa: F401
    except Exception as e:  # ImportError, or a wheel that cannot load its libraries
        return None, f"{type(e).__name__}: {e}"
    return tf, None


def build_shim(tf, force=False):
    """The command at the top of recom_amd/tf_shim/fcp_tf_ops.cc.  Returns the path of librecom_fcp.so."""
    import __graft_entry__ as entry
    entry.build()
    shim_dir = os.path.join(ROOT, "recom_amd", "tf_shim")
    src, out = os.path.join(shim_dir, "fcp_tf_ops.cc"), os.path.join(shim_dir, "librecom_fcp.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-std=c++17", "-shared", "-fPIC", src, "-o", out + ".tmp"]
           + tf.sysconfig.get_compile_flags() + tf.sysconfig.get_link_flags()
           + ["-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "recom_amd"), "-lfcp_hip",
              "-Wl,-rpath,$ORIGIN/..", "-DTENSORFLOW_USE_ROCM=1"])
    subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def make_model(args):
    import graph_fixtures as F
    import numpy as np
    if args.model == "s1":
        gd, feeds, variables, fetches = F.s1_model(columns=args.columns or 100, B=args.batch or 128, vocab=args.vocab or 10_000)
    elif args.model == "microbenchmark":
        gd, feeds, variables, fetches = F.microbenchmark_model(columns=args.columns or 100, B=args.batch or 128)
    elif args.model == "canonical":
        gd, feeds, variables, fetches = F.canonical_model(B=args.batch or 300, seed=3)
    else:
        gd, feeds, variables, fetches = F.s2_model(columns=args.columns or 1000, vocab=args.vocab or 1_000_000, B=args.batch or 512,
                                                   materialize=False)
        rng = np.random.default_rng(1)
        variables = {k: rng.standard_normal(v, dtype=np.float32) for k, v in variables.items()}  # (values do not matter to the A/B)
    return gd, feeds, variables, fetches


def session_for(tf, graph_bytes, variables, device, threads=0):
    """Imports a GraphDef, assigns every VariableV2 its table and returns (session, graph).  device: "/cpu:0" keeps the
    whole graph on TF-CPU; None lets the placer put the Addons> GPU ops on the GPU."""
    tf1 = tf.compat.v1
    graph = tf.Graph()
    with graph.as_default():
        gd = tf1.GraphDef()
        gd.ParseFromString(graph_bytes)
        if device:
            with tf.device(device):
                tf.graph_util.import_graph_def(gd, name="")
        else:
            tf.graph_util.import_graph_def(gd, name="")
        assigns = []
        for name, value in variables.items():
            ref = graph.get_tensor_by_name(name + ":0")
            ph = tf1.placeholder(tf.float32, value.shape, name=name.replace("/", "_") + "_init")
            assigns.append((tf1.assign(ref, ph, validate_shape=True).op, ph, value))
    cfg = tf1.ConfigProto(allow_soft_placement=True)
    if device == "/cpu:0":
        cfg.device_count["GPU"] = 0
    if threads:
        cfg.intra_op_parallelism_threads = threads
        cfg.inter_op_parallelism_threads = threads
    sess = tf1.Session(graph=graph, config=cfg)
    for op, ph, value in assigns:                      # one at a time: a 120-GB feed in one run would be held twice
        sess.run(op, {ph: value})
    return sess, graph


def classify_fetch_columns(built):
    """(offset, width, pooled?) of every column inside its concat group, from the plan the builder wrote."""
    from recom_amd.plan import FORM_BATCH_COL_REDUCTION, FORM_SEGMENT_REDUCE
    offs = built.spec.column_offsets()
    return [(c.concat_group, offs[k], c.dim, c.form in (FORM_SEGMENT_REDUCE, FORM_BATCH_COL_REDUCTION))
            for k, c in enumerate(built.spec.columns)]


def time_tf_cpu(tf, sess, fetches, feeds, batch, seconds, thread_counts):
    """Session::Run threads sharing ONE session — the reference harness' serve_workers (recom_examples.patch:193-216)."""
    fd = {k + ":0": v for k, v in feeds.items()}
    ft = [f + ":0" for f in fetches]
    for _ in range(3):
        sess.run(ft, fd)
    sweep = {}
    for t in thread_counts:
        counts = [0] * t
        stop = time.perf_counter() + seconds

        def worker(i):
            while time.perf_counter() < stop:
                sess.run(ft, fd)
                counts[i] += 1
        th = [threading.Thread(target=worker, args=(i,)) for i in range(t)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        sweep[t] = batch * sum(counts) / (time.perf_counter() - t0)
    return sweep


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--model", default="s1", choices=["s1", "microbenchmark", "canonical", "s2"])
    ap.add_argument("--columns", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--vocab", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=4.0, help="TF-CPU timing: seconds per thread count")
    ap.add_argument("--threads", default="1,8,32", help="Session::Run threads of the TF-CPU timing (the reference's budget: 32 cores)")
    ap.add_argument("--baseline-only", action="store_true", help="time TF-CPU only (no shim, no GPU): bench.py's cpu_baseline hook")
    ap.add_argument("--staged", action="store_true", help="rewrite for the staged plan (ConcatInputs converts while it packs)")
    args = ap.parse_args(argv)

    tf, why = import_tensorflow()
    if tf is None:
        print(json.dumps({"skipped": True, "reason": f"TensorFlow is not importable here ({why})"}))
        return SKIPPED
    import numpy as np
    from recom_amd.graph import build_plan, rewrite_graph
    from recom_amd.plan_io import save_plan

    gd, feeds, variables, fetches = make_model(args)
    batch = int(next(iter(feeds.values())).shape[0])
    rec = {"model": args.model, "tensorflow": tf.__version__, "batch": batch, "tables": len(variables)}

    cores = len(os.sched_getaffinity(0))
    cpu_sess, _ = session_for(tf, gd.SerializeToString(), variables, "/cpu:0", threads=min(32, cores))
    want = cpu_sess.run([f + ":0" for f in fetches], {k + ":0": v for k, v in feeds.items()})
    threads = [int(t) for t in args.threads.split(",") if int(t) <= cores] or [1]
    sweep = time_tf_cpu(tf, cpu_sess, fetches, feeds, batch, args.seconds, threads)
    head = 32 if 32 in sweep else max(sweep, key=sweep.get)
    rec["cpu_baseline"] = {"value": sweep[head], "unit": "inferences/s", "cores": min(32, cores), "kind": "tensorflow",
                           "session_run_threads_sweep": {str(k): v for k, v in sweep.items()},
                           "sample": f"{args.model}: {len(variables)} tables, batch {batch}, TensorFlow {tf.__version__} on the CPU "
                                     f"(intra/inter-op threads {min(32, cores)}), {args.seconds} s per thread count, one shared Session"}
    if args.baseline_only:
        print(json.dumps(rec))
        return 0

    shim = build_shim(tf)
    tf.load_op_library(shim)
    with tempfile.TemporaryDirectory() as tmp:
        plan_path = os.path.join(tmp, "model.fcp")
        built = build_plan(gd, "passthrough")
        stage = None
        if args.staged:
            spec, stage = built.spec.staged_for_concat_inputs()
            save_plan(spec, plan_path, stage)
        else:
            save_plan(built.spec, plan_path)
        out_gd = rewrite_graph(gd, built, plan_path, stage=stage)
        ops = sorted({n.op for n in out_gd.node if n.op.startswith("Addons>")})
        gpu_sess, _ = session_for(tf, out_gd.SerializeToString(), variables, None)
        got = gpu_sess.run([f + ":0" for f in fetches], {k + ":0": v for k, v in feeds.items()})
    cols = classify_fetch_columns(built)
    worst_copy, worst_pooled, n_copy, n_pooled = 0.0, 0.0, 0, 0
    # fetches are Identity nodes behind the concat groups, in group order
    for g, (w, h) in enumerate(zip(want, got)):
        w, h = np.asarray(w), np.asarray(h)
        assert w.shape == h.shape, (fetches[g], w.shape, h.shape)
        for grp, off, dim, pooled in cols:
            if grp != g or w.ndim != 2 or off + dim > w.shape[1]:
                continue
            d = float(np.abs(w[:, off:off + dim] - h[:, off:off + dim]).max(initial=0.0))
            if pooled:
                worst_pooled, n_pooled = max(worst_pooled, d), n_pooled + 1
            else:
                worst_copy, n_copy = max(worst_copy, d), n_copy + 1
                if not np.array_equal(w[:, off:off + dim], h[:, off:off + dim]):
                    rec.setdefault("copy_columns_not_bit_exact", []).append([g, off, dim])
    rec["parity"] = {"addons_ops_in_the_rewritten_graph": ops, "copy_columns": n_copy, "pooled_columns": n_pooled,
                     "copy_max_abs_diff": worst_copy, "pooled_max_abs_diff": worst_pooled,
                     "ok": "copy_columns_not_bit_exact" not in rec and worst_pooled < 1e-5,
                     "protocol": "the reference's --embedding_only A/B (recom_examples.patch:161-165, 3506-3519): the embedding stage's "
                                 "concat(s) fetched from TF-CPU and from the rewritten graph on the same feeds"}
    print(json.dumps(rec))
    return 0 if rec["parity"]["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
