"""Test helper: writes the tensors of a weights blob as a minimal ONNX file with LoFTR_teacher's topology as far as
the product's loader looks at it (21 Conv nodes in execution order, 48 MatMul nodes with an initializer operand, the
positional-encoding Constant, LayerNorm parameters under their module names).  Lets the ONNX loading path be exercised
where the reference's model file is not available (the GPU box)."""
import struct

import numpy as np


def _varint(v):
    out = b""
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def _tensor(name, a):
    a = np.ascontiguousarray(a, np.float32)
    t = b"".join(_vi(1, d) for d in a.shape) + _vi(2, 1) + _ld(8, name.encode()) + _ld(9, a.tobytes())
    return t


def _node(op, ins, outs, value=None):
    n = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs) + _ld(4, op.encode())
    if value is not None:
        n += _ld(5, _ld(1, b"value") + _ld(5, _tensor("", value)) + _vi(20, 4))
    return n


def read_blob(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"MSFLTR01"
    n = struct.unpack_from("<I", raw, 8)[0]
    payload = 12 + 60 * n
    t = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from("<32sI4III", raw, 12 + 60 * i)
        dims = [d0, d1, d2, d3][:ndim]
        t[name.rstrip(b"\0").decode()] = np.frombuffer(raw, np.float32, cnt, payload + 4 * off).reshape(dims)
    return t


def write_onnx(path, t):
    init, nodes = [], []
    for i in range(21):
        w = "conv%02d.w" % i if i < 20 else "outconv.w"
        init.append(_tensor("W%d" % i, t[w]))
        ins = ["x%d" % i, "W%d" % i]
        if i < 20:
            init.append(_tensor("B%d" % i, t["conv%02d.b" % i]))
            ins.append("B%d" % i)
        nodes.append(_node("Conv", ins, ["x%d" % (i + 1)]))
    for k in range(2):   # the graph holds the positional encoding once per image
        nodes.append(_node("Constant", [], ["pe%d" % k], value=t["pe"].reshape(1, 32, 30, 40)))
    for b in range(8):
        for j, nm in enumerate(("wq", "wk", "wv", "wmerge", "wmlp0", "wmlp1")):
            wn = "M%d_%d" % (b, j)
            init.append(_tensor(wn, t["blk%d.%s" % (b, nm)]))
            nodes.append(_node("MatMul", ["t%d_%d" % (b, j), wn], ["u%d_%d" % (b, j)]))
            if j == 2:   # products without an initializer operand in between, as in the real graph
                nodes.append(_node("MatMul", ["kT%d" % b, "v%d" % b], ["kv%d" % b]))
    for L in range(4):
        for src, dst in (("norm1.weight", "n1w"), ("norm1.bias", "n1b"), ("norm2.weight", "n2w"), ("norm2.bias", "n2b")):
            init.append(_tensor("loftr_coarse.layers.%d.%s" % (L, src), t["ln%d.%s" % (L, dst)]))
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"synthetic") + b"".join(_ld(5, x) for x in init)
    model = _vi(1, 7) + _ld(2, b"tests/onnx_writer.py") + _ld(7, graph)
    with open(path, "wb") as f:
        f.write(model)
