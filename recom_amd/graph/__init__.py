"""GraphDef front end of the fused feature-column path: read a
rewritten TensorFlow GraphDef, build the column plan the HIP kernels interpret, and
emit the graph that calls the three ``Addons>`` ops.  Needs ``google.protobuf`` only —
no TensorFlow, no GPU."""
from .plan_builder import BuiltPlan, PlanBuilder, Unsupported, build_plan  # noqa: F401
from .rewrite import rewrite_graph  # noqa: F401
from .tf_proto import load_graphdef, parse_graphdef, save_graphdef  # noqa: F401


def native_build(graph_def_bytes: bytes, plan_path: str, host_concat: str = "passthrough", staged: bool = False,
                 prune: bool = True, want_graph: bool = True):
    """The same build through the C ABI (``fcp_graph_build``, ``recom_amd/csrc/fcp_graph.cc``) — what the retained
    Grappler pass calls in-process in place of ``CudaEmitter::Optimize`` (``cuda_emitter.cc:80-116``).  Writes the plan
    file, returns ``(rewritten GraphDef bytes or None, description)``; raises :class:`Unsupported` when the graph has
    nothing to fuse."""
    import ctypes as C
    from .. import lib as _lib
    L = _lib.load()
    flags = (1 if host_concat == "external" else 0) | (2 if staged else 0) | (0 if prune else 4)
    if host_concat not in ("passthrough", "external"):
        raise ValueError("host_concat must be 'passthrough' or 'external'")
    out, n, desc = C.c_void_p(), C.c_size_t(0), C.c_char_p()
    rc = L.fcp_graph_build(graph_def_bytes, len(graph_def_bytes), flags, plan_path.encode(),
                           C.byref(out) if want_graph else None, C.byref(n), C.byref(desc))
    try:
        if rc == _lib.FCP_ERR_UNSUPPORTED:
            raise Unsupported(L.fcp_last_error().decode(errors="replace"))
        _lib.check(rc, "fcp_graph_build")
        graph = C.string_at(out.value, n.value) if want_graph and out.value else None
        text = C.cast(desc, C.c_char_p).value.decode(errors="replace") if desc else ""
        return graph, text
    finally:
        if out.value:
            L.fcp_graph_free(out)
        if desc:
            L.fcp_graph_free(C.cast(desc, C.c_void_p))
