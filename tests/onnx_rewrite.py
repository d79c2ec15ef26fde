"""Test helper: rewrite an ONNX ModelProto the way a graph simplifier (onnxsim, `df/scripts/export.py:39-56,123-125` with `--simplify`)
leaves it — without `onnx` / `onnxsim`, which are not installed here: a ~100-line protobuf wire codec over the handful of message
types involved (onnx.proto3: ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5, .input = 11, .output = 12,
.value_info = 13; NodeProto.input = 1, .output = 2, .name = 3, .op_type = 4, .attribute = 5; AttributeProto.name = 1, .t = 5;
TensorProto.name = 8; ValueInfoProto.name = 1).

What `simplify(model_bytes)` does to a graph, all of it semantics-preserving:
  * every value (graph inputs / outputs excepted: the runtime addresses those by name) and every initializer gets a new, opaque name
    (`onnx::Conv_<n>` style), node names are dropped;
  * `Constant` nodes become initializers (constant folding's first step);
  * `Identity` nodes are removed and `Pad` nodes in front of a Conv are folded away (the consumer is rewired to the producer — onnxsim's
    eliminate_identity / fuse_pad_into_conv; the weight reader never looks at pads);
  * the initializers are stored in a different order (sorted by their new names)."""
from __future__ import annotations

from typing import Dict, List, Tuple


def _rd_varint(b: bytes, i: int) -> Tuple[int, int]:
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def _wr_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        if v:
            out.append(c | 0x80)
        else:
            out.append(c)
            return bytes(out)


def fields(b: bytes) -> List[Tuple[int, int, object]]:
    """[(field number, wire type, value)]: value = int for varint / fixed, bytes for length-delimited."""
    out, i = [], 0
    while i < len(b):
        key, i = _rd_varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _rd_varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            n, i = _rd_varint(b, i)
            v, i = b[i:i + n], i + n
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {w}")
        out.append((f, w, v))
    return out


def encode(fs: List[Tuple[int, int, object]]) -> bytes:
    out = bytearray()
    for f, w, v in fs:
        out += _wr_varint((f << 3) | w)
        if w == 0:
            out += _wr_varint(v)
        elif w == 2:
            out += _wr_varint(len(v)) + v
        else:
            out += v
    return bytes(out)


def _get(fs, f):
    return [v for ff, _, v in fs if ff == f]


def simplify(model: bytes) -> bytes:
    mf = fields(model)
    gi = next(i for i, (f, w, _) in enumerate(mf) if f == 7 and w == 2)
    g = fields(mf[gi][2])
    keep = set()   # names the runtime uses: graph inputs and outputs stay
    for f, w, v in g:
        if f in (11, 12):
            keep.add(_get(fields(v), 1)[0].decode())
    nodes = [fields(v) for f, w, v in g if f == 1]
    inits = [fields(v) for f, w, v in g if f == 5]
    # ---- Constant -> initializer
    rest = []
    for n in nodes:
        op = _get(n, 4)[0].decode()
        if op == "Constant":
            out = _get(n, 2)[0]
            t = None
            for a in _get(n, 5):
                af = fields(a)
                if _get(af, 1)[0] == b"value" and _get(af, 5):
                    t = [x for x in fields(_get(af, 5)[0]) if x[0] != 8]
            if t is None:
                rest.append(n)
                continue
            inits.append(t + [(8, 2, out)])
        else:
            rest.append(n)
    nodes = rest
    # ---- Identity / Pad-before-Conv: rewire the consumers to the producer
    alias: Dict[bytes, bytes] = {}
    consumers: Dict[bytes, List[str]] = {}
    for n in nodes:
        for i in _get(n, 1):
            consumers.setdefault(i, []).append(_get(n, 4)[0].decode())
    rest = []
    for n in nodes:
        op = _get(n, 4)[0].decode()
        ins, outs = _get(n, 1), _get(n, 2)
        drop = op == "Identity" or (op == "Pad" and consumers.get(outs[0], []) in (["Conv"], ["ConvTranspose"]))
        if drop and outs[0].decode() not in keep:
            alias[outs[0]] = ins[0]
        else:
            rest.append(n)
    nodes = rest

    def res(name: bytes) -> bytes:
        while name in alias:
            name = alias[name]
        return name

    # ---- new names
    new: Dict[bytes, bytes] = {}
    count = [0]

    def rn(name: bytes, kind: str = "v") -> bytes:
        name = res(name)
        if name == b"" or name.decode() in keep:
            return name
        if name not in new:
            count[0] += 1
            new[name] = f"onnx::{kind}_{count[0] * 7 + 1000}".encode()
        return new[name]

    init_names = {_get(t, 8)[0] for t in inits}
    out_nodes = []
    for n in nodes:
        op = _get(n, 4)[0].decode()
        fs = []
        for f, w, v in n:
            if f == 1:
                fs.append((1, 2, rn(v, op if v in init_names else "v")))
            elif f == 2:
                fs.append((2, 2, rn(v)))
            elif f == 3:
                continue   # node names dropped
            else:
                fs.append((f, w, v))
        out_nodes.append(encode(fs))
    out_inits = []
    for t in inits:
        name = _get(t, 8)[0]
        out_inits.append((rn(name, "Init"), encode([x for x in t if x[0] != 8] + [(8, 2, rn(name, "Init"))])))
    out_inits.sort(key=lambda kv: kv[0], reverse=True)
    g2 = [(1, 2, b) for b in out_nodes]
    g2 += [(f, w, v) for f, w, v in g if f not in (1, 5, 13)]   # name, doc, inputs, outputs kept; value_info dropped (stale names)
    g2 += [(5, 2, b) for _, b in out_inits]
    mf[gi] = (7, 2, encode(g2))
    return encode(mf)
