"""TensorFlow GraphDef messages, declared at run time (no TensorFlow needed).

The plan builder reads and writes the same artefact the reference's pass works on —
a serialized ``tensorflow.GraphDef`` (``fc_optimize_pass.cc:26-31`` converts the graph
to a GraphDef and dumps it as ``before_opt`` / ``after_opt``).  TensorFlow is not in
this image, so the handful of messages a GraphDef is made of are declared here with
``google.protobuf``'s dynamic descriptors, using the public field numbers of
``tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,types,
versions}.proto`` (TF 2.6.2, the reference's pinned version).  Binary and text
(``.pbtxt``) formats both work; fields not declared here (function library, debug
info) survive a binary round trip as unknown fields.
"""
from __future__ import annotations

from google.protobuf import descriptor_pb2 as dpb
from google.protobuf import descriptor_pool, message_factory, text_format

_F = dpb.FieldDescriptorProto
_OPT, _REP = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED

DATA_TYPES = {
    "DT_INVALID": 0, "DT_FLOAT": 1, "DT_DOUBLE": 2, "DT_INT32": 3, "DT_UINT8": 4, "DT_INT16": 5, "DT_INT8": 6,
    "DT_STRING": 7, "DT_COMPLEX64": 8, "DT_INT64": 9, "DT_BOOL": 10, "DT_QINT8": 11, "DT_QUINT8": 12,
    "DT_QINT32": 13, "DT_BFLOAT16": 14, "DT_QINT16": 15, "DT_QUINT16": 16, "DT_UINT16": 17, "DT_COMPLEX128": 18,
    "DT_HALF": 19, "DT_RESOURCE": 20, "DT_VARIANT": 21, "DT_UINT32": 22, "DT_UINT64": 23,
}
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 6, 7, 9, 10


def _field(msg, name, number, ftype, label=_OPT, type_name=None, oneof=None, packed=None):
    f = msg.field.add(name=name, number=number, type=ftype, label=label)
    if type_name:
        f.type_name = type_name
    if oneof is not None:
        f.oneof_index = oneof
    if packed is not None:
        f.options.packed = packed
    return f


def _map_field(msg, name, number, value_type_name):
    entry = msg.nested_type.add(name=name.capitalize() + "Entry")
    entry.options.map_entry = True
    _field(entry, "key", 1, _F.TYPE_STRING)
    _field(entry, "value", 2, _F.TYPE_MESSAGE, type_name=value_type_name)
    _field(msg, name, number, _F.TYPE_MESSAGE, _REP, type_name=f".tensorflow.{msg.name}.{entry.name}")


def _build_pool():
    fp = dpb.FileDescriptorProto(name="recom_amd/tf_graph_subset.proto", package="tensorflow", syntax="proto3")
    e = fp.enum_type.add(name="DataType")
    for k, v in DATA_TYPES.items():
        e.value.add(name=k, number=v)
    for k, v in DATA_TYPES.items():
        if v:
            e.value.add(name=k + "_REF", number=v + 100)

    shape = fp.message_type.add(name="TensorShapeProto")
    dim = shape.nested_type.add(name="Dim")
    _field(dim, "size", 1, _F.TYPE_INT64)
    _field(dim, "name", 2, _F.TYPE_STRING)
    _field(shape, "dim", 2, _F.TYPE_MESSAGE, _REP, ".tensorflow.TensorShapeProto.Dim")
    _field(shape, "unknown_rank", 3, _F.TYPE_BOOL)

    t = fp.message_type.add(name="TensorProto")
    _field(t, "dtype", 1, _F.TYPE_ENUM, type_name=".tensorflow.DataType")
    _field(t, "tensor_shape", 2, _F.TYPE_MESSAGE, type_name=".tensorflow.TensorShapeProto")
    _field(t, "version_number", 3, _F.TYPE_INT32)
    _field(t, "tensor_content", 4, _F.TYPE_BYTES)
    _field(t, "float_val", 5, _F.TYPE_FLOAT, _REP, packed=True)
    _field(t, "double_val", 6, _F.TYPE_DOUBLE, _REP, packed=True)
    _field(t, "int_val", 7, _F.TYPE_INT32, _REP, packed=True)
    _field(t, "string_val", 8, _F.TYPE_BYTES, _REP)
    _field(t, "scomplex_val", 9, _F.TYPE_FLOAT, _REP, packed=True)
    _field(t, "int64_val", 10, _F.TYPE_INT64, _REP, packed=True)
    _field(t, "bool_val", 11, _F.TYPE_BOOL, _REP, packed=True)
    _field(t, "dcomplex_val", 12, _F.TYPE_DOUBLE, _REP, packed=True)
    _field(t, "half_val", 13, _F.TYPE_INT32, _REP, packed=True)
    _field(t, "uint32_val", 16, _F.TYPE_UINT32, _REP, packed=True)
    _field(t, "uint64_val", 17, _F.TYPE_UINT64, _REP, packed=True)

    nal = fp.message_type.add(name="NameAttrList")
    _field(nal, "name", 1, _F.TYPE_STRING)
    _map_field(nal, "attr", 2, ".tensorflow.AttrValue")

    av = fp.message_type.add(name="AttrValue")
    lv = av.nested_type.add(name="ListValue")
    _field(lv, "s", 2, _F.TYPE_BYTES, _REP)
    _field(lv, "i", 3, _F.TYPE_INT64, _REP, packed=True)
    _field(lv, "f", 4, _F.TYPE_FLOAT, _REP, packed=True)
    _field(lv, "b", 5, _F.TYPE_BOOL, _REP, packed=True)
    _field(lv, "type", 6, _F.TYPE_ENUM, _REP, ".tensorflow.DataType", packed=True)
    _field(lv, "shape", 7, _F.TYPE_MESSAGE, _REP, ".tensorflow.TensorShapeProto")
    _field(lv, "tensor", 8, _F.TYPE_MESSAGE, _REP, ".tensorflow.TensorProto")
    _field(lv, "func", 9, _F.TYPE_MESSAGE, _REP, ".tensorflow.NameAttrList")
    av.oneof_decl.add(name="value")
    _field(av, "list", 1, _F.TYPE_MESSAGE, type_name=".tensorflow.AttrValue.ListValue", oneof=0)
    _field(av, "s", 2, _F.TYPE_BYTES, oneof=0)
    _field(av, "i", 3, _F.TYPE_INT64, oneof=0)
    _field(av, "f", 4, _F.TYPE_FLOAT, oneof=0)
    _field(av, "b", 5, _F.TYPE_BOOL, oneof=0)
    _field(av, "type", 6, _F.TYPE_ENUM, type_name=".tensorflow.DataType", oneof=0)
    _field(av, "shape", 7, _F.TYPE_MESSAGE, type_name=".tensorflow.TensorShapeProto", oneof=0)
    _field(av, "tensor", 8, _F.TYPE_MESSAGE, type_name=".tensorflow.TensorProto", oneof=0)
    _field(av, "placeholder", 9, _F.TYPE_STRING, oneof=0)
    _field(av, "func", 10, _F.TYPE_MESSAGE, type_name=".tensorflow.NameAttrList", oneof=0)

    nd = fp.message_type.add(name="NodeDef")
    _field(nd, "name", 1, _F.TYPE_STRING)
    _field(nd, "op", 2, _F.TYPE_STRING)
    _field(nd, "input", 3, _F.TYPE_STRING, _REP)
    _field(nd, "device", 4, _F.TYPE_STRING)
    _map_field(nd, "attr", 5, ".tensorflow.AttrValue")

    vd = fp.message_type.add(name="VersionDef")
    _field(vd, "producer", 1, _F.TYPE_INT32)
    _field(vd, "min_consumer", 2, _F.TYPE_INT32)
    _field(vd, "bad_consumers", 3, _F.TYPE_INT32, _REP, packed=True)

    gd = fp.message_type.add(name="GraphDef")
    _field(gd, "node", 1, _F.TYPE_MESSAGE, _REP, ".tensorflow.NodeDef")
    _field(gd, "version", 3, _F.TYPE_INT32)
    _field(gd, "versions", 4, _F.TYPE_MESSAGE, type_name=".tensorflow.VersionDef")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fp)
    return pool


_POOL = _build_pool()


def _cls(name):
    return message_factory.GetMessageClass(_POOL.FindMessageTypeByName("tensorflow." + name))


GraphDef = _cls("GraphDef")
NodeDef = _cls("NodeDef")
AttrValue = _cls("AttrValue")
TensorProto = _cls("TensorProto")
TensorShapeProto = _cls("TensorShapeProto")


def parse_graphdef(data: bytes):
    """Binary GraphDef, or text format when the bytes do not parse as binary."""
    g = GraphDef()
    try:
        g.ParseFromString(data)
        if len(g.node) or not data.strip():
            return g
    except Exception:  # noqa: BLE001 - fall through to the text parser
        pass
    g = GraphDef()
    text_format.Parse(data.decode("utf-8"), g, allow_unknown_field=True)
    return g


def load_graphdef(path: str):
    with open(path, "rb") as f:
        return parse_graphdef(f.read())


def save_graphdef(g, path: str) -> None:
    with open(path, "wb") as f:
        if path.endswith(".pbtxt"):
            f.write(text_format.MessageToString(g).encode("utf-8"))
        else:
            f.write(g.SerializeToString())
