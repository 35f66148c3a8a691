"""``python -m recom_amd.graph model.pb --plan model.fcp --out model_fcp.pb``

Offline counterpart of the reference's in-process ``CudaEmitter::Optimize``
(``cuda_emitter.cc:80-116``): instead of generating CUDA, calling nvcc and caching the
``.so`` by md5, it writes a column-plan file and the rewritten GraphDef whose
``FeatureColumnProcess.dlpath`` points at it."""
import argparse
import sys

from ..plan_io import save_plan
from . import Unsupported, build_plan, load_graphdef, rewrite_graph, save_graphdef


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m recom_amd.graph", description=__doc__)
    ap.add_argument("graph", help="GraphDef after the lookup optimizers (.pb or .pbtxt)")
    ap.add_argument("--plan", required=True, help="column-plan file to write")
    ap.add_argument("--out", help="rewritten GraphDef to write (.pb or .pbtxt)")
    ap.add_argument("--no-prune", action="store_true", help="keep the replaced subgraphs in the output graph")
    ap.add_argument("--host-concat", choices=["passthrough", "external"], default="passthrough",
                    help="non-lookup concat inputs: through ConcatInputs as passthrough columns (default), or as "
                         "Addons>ConcatOutputs host inputs into reserved slots — the reference's own wiring")
    ap.add_argument("--staged", action="store_true",
                    help="write the STAGED plan: Addons>ConcatInputs packs int64 ids as int32 and turns the sorted row ids / "
                         "SparseTensor indices of pooled columns into int32 row offsets (the plan file's stage section tells it "
                         "how; the rewritten ConcatInputs node names the plan in its `_fcp_plan` attr and receives the symbols "
                         "vector as one more input) - the device then runs neither the segment-offset pre-pass nor a search")
    ap.add_argument("--native", action="store_true",
                    help="build through the C ABI (fcp_graph_build, recom_amd/csrc/fcp_graph.cc) - the in-process entry of the "
                         "retained Grappler pass; binary .pb graphs only; same plan file, same rewritten graph")
    args = ap.parse_args(argv)
    if args.native:
        from . import native_build
        try:
            graph, text = native_build(open(args.graph, "rb").read(), args.plan, args.host_concat, args.staged,
                                       prune=not args.no_prune, want_graph=bool(args.out))
        except Unsupported as why:
            print(f"{why}", file=sys.stderr)
            return 1
        print(text)
        if args.out:
            with open(args.out, "wb") as f:
                f.write(graph)
        return 0
    gd = load_graphdef(args.graph)
    try:
        built = build_plan(gd, args.host_concat)
    except Unsupported as why:
        print(f"nothing to fuse: {why}", file=sys.stderr)
        return 1
    stage = None
    if args.staged:
        spec, stage = built.spec.staged_for_concat_inputs()
        save_plan(spec, args.plan, stage)
    else:
        save_plan(built.spec, args.plan)
    print(built.describe())
    if stage is not None:
        names = {0: "copied", 1: "int64 -> int32", 2: "row ids -> row offsets"}
        counts = {v: sum(1 for m in stage.modes if names[m] == v) for v in names.values()}
        print("  staged ConcatInputs: " + ", ".join(f"{n}x {k}" for k, n in counts.items() if n))
    if args.out:
        save_graphdef(rewrite_graph(gd, built, args.plan, prune=not args.no_prune, stage=stage), args.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
