"""GraphDef front end of the fused feature-column path: read a
rewritten TensorFlow GraphDef, build the column plan the HIP kernels interpret, and
emit the graph that calls the three ``Addons>`` ops.  Needs ``google.protobuf`` only —
no TensorFlow, no GPU."""
from .plan_builder import BuiltPlan, PlanBuilder, Unsupported, build_plan  # noqa: F401
from .rewrite import rewrite_graph  # noqa: F401
from .tf_proto import load_graphdef, parse_graphdef, save_graphdef  # noqa: F401
