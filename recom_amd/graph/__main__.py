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
    args = ap.parse_args(argv)
    gd = load_graphdef(args.graph)
    try:
        built = build_plan(gd, args.host_concat)
    except Unsupported as why:
        print(f"nothing to fuse: {why}", file=sys.stderr)
        return 1
    save_plan(built.spec, args.plan)
    print(built.describe())
    if args.out:
        save_graphdef(rewrite_graph(gd, built, args.plan, prune=not args.no_prune), args.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
