"""oracle/onnx_oracle.py -- TEST INFRASTRUCTURE ONLY.

Runs the reference's LoFTR graph (/root/reference/model/LoFTR_teacher.onnx, the file
::DNNFeatureMatcher hands to Ort::Session at src/dnnfeaturematcher.cpp:18-21) node by node with torch-CPU,
and restates the rest of DNNFeatureMatcher::MatchFrames (src/dnnfeaturematcher.cpp:44-102): u8 -> f32 * (1/255),
output[0], strict '> threshold', row-major findNonZero, decode to the 16-px cell grid.

Neither `onnx` nor `onnxruntime` exists in this image, so the protobuf wire format is read directly
 and the 23 operator types the graph uses are interpreted with torch ops in f32.
This script only runs where /root/reference exists (the build container).  It is used to
  (1) extract the weights blob the HIP path loads (mono_slam_framework_amd/weights/), and
  (2) emit the golden fixtures under tests/golden/ that pin both the C restatement (oracle/loftr_oracle.c)
      and the HIP path; the GPU box only ever sees those fixtures.
"""
import os
import struct
import sys

import numpy as np

MODEL = "/root/reference/model/LoFTR_teacher.onnx"


# ----------------------------------------------------------------------------- protobuf wire reader
def _varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b):
    i = 0
    n = len(b)
    while i < n:
        key, i = _varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield f, wt, v


def _s64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(wt, v):
    if wt == 0:
        return [_s64(v)]
    out = []
    i = 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_s64(x))
    return out


def _tensor(b):
    dims, dtype, name, raw = [], 0, "", b""
    floats, int64s = [], []
    for f, wt, v in _fields(b):
        if f == 1:
            dims += _packed_ints(wt, v)
        elif f == 2:
            dtype = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            floats += list(struct.unpack("<%df" % (len(v) // 4), v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 7:
            int64s += _packed_ints(wt, v)
    np_t = {1: np.float32, 7: np.int64, 6: np.int32, 9: np.bool_, 11: np.float64}[dtype]
    if raw:
        a = np.frombuffer(raw, dtype=np_t).copy()
    elif floats:
        a = np.array(floats, np_t)
    else:
        a = np.array(int64s, np_t)
    return name, a.reshape(dims)


def _attr(b):
    name, val = "", None
    ints, floats = [], []
    for f, wt, v in _fields(b):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _s64(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(struct.unpack("<%df" % (len(v) // 4), v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_ints(wt, v)
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def _node(b):
    ins, outs, op, attrs, name = [], [], "", {}, ""
    for f, wt, v in _fields(b):
        if f == 1:
            ins.append(v.decode())
        elif f == 2:
            outs.append(v.decode())
        elif f == 3:
            name = v.decode()
        elif f == 4:
            op = v.decode()
        elif f == 5:
            k, a = _attr(v)
            attrs[k] = a
    return {"op": op, "in": ins, "out": outs, "attr": attrs, "name": name}


def _value_info_name(b):
    for f, wt, v in _fields(b):
        if f == 1:
            return v.decode()
    return ""


def load_graph(path=MODEL):
    data = open(path, "rb").read()
    graph = None
    for f, wt, v in _fields(data):
        if f == 7:
            graph = v
    nodes, inits, inputs, outputs = [], {}, [], []
    for f, wt, v in _fields(graph):
        if f == 1:
            nodes.append(_node(v))
        elif f == 5:
            n, a = _tensor(v)
            inits[n] = a
        elif f == 11:
            inputs.append(_value_info_name(v))
        elif f == 12:
            outputs.append(_value_info_name(v))
    inputs = [i for i in inputs if i not in inits]
    return {"nodes": nodes, "init": inits, "inputs": inputs, "outputs": outputs}


# ----------------------------------------------------------------------------- interpreter
def run_graph(g, feeds, want=None, keep_all=False):
    import torch
    import torch.nn.functional as F
    env = {k: torch.from_numpy(v) for k, v in g["init"].items()}
    for k, v in feeds.items():
        env[k] = torch.from_numpy(np.ascontiguousarray(v))
    for nd in g["nodes"]:
        op, a = nd["op"], nd["attr"]
        x = [env[i] if i else None for i in nd["in"]]
        if op == "Constant":
            y = torch.from_numpy(np.array(a["value"]))
        elif op == "Concat":
            y = torch.cat(x, dim=a["axis"])
        elif op == "Conv":
            assert a.get("group", 1) == 1 and all(d == 1 for d in a.get("dilations", [1, 1]))
            p = a["pads"]
            assert p[0] == p[2] and p[1] == p[3]
            y = F.conv2d(x[0], x[1], x[2] if len(x) > 2 else None, stride=a["strides"], padding=(p[0], p[1]))
        elif op == "Relu":
            y = torch.relu(x[0])
        elif op == "Elu":
            y = F.elu(x[0], alpha=a.get("alpha", 1.0))
        elif op == "Add":
            y = x[0] + x[1]
        elif op == "Sub":
            y = x[0] - x[1]
        elif op == "Mul":
            y = x[0] * x[1]
        elif op == "Div":
            y = x[0] / x[1]
        elif op == "Pow":
            y = torch.pow(x[0], x[1])
        elif op == "Sqrt":
            y = torch.sqrt(x[0])
        elif op == "MatMul":
            y = torch.matmul(x[0], x[1])
        elif op == "Split":
            parts = torch.split(x[0], a["split"], dim=a["axis"])
            for o, pz in zip(nd["out"], parts):
                env[o] = pz
            continue
        elif op == "Shape":
            y = torch.tensor(list(x[0].shape), dtype=torch.int64)
        elif op == "Slice":
            st, en, ax = x[1].tolist(), x[2].tolist(), (x[3].tolist() if len(x) > 3 and x[3] is not None else None)
            steps = x[4].tolist() if len(x) > 4 and x[4] is not None else [1] * len(st)
            y = x[0]
            for j in range(len(st)):
                d = ax[j] if ax is not None else j
                sl = [slice(None)] * y.dim()
                sl[d] = slice(st[j], None if en[j] >= (1 << 62) else en[j], steps[j])
                y = y[tuple(sl)]
        elif op == "Reshape":
            shp = x[1].tolist()
            shp = [x[0].shape[i] if s == 0 else s for i, s in enumerate(shp)]
            y = x[0].reshape(shp)
        elif op == "Transpose":
            y = x[0].permute(a["perm"])
        elif op == "Unsqueeze":
            y = x[0]
            for d in sorted(a["axes"]):
                y = y.unsqueeze(d)
        elif op == "Squeeze":
            y = x[0]
            for d in sorted(a["axes"], reverse=True):
                y = y.squeeze(d)
        elif op == "ReduceSum":
            y = x[0].sum(dim=a["axes"], keepdim=bool(a.get("keepdims", 1)))
        elif op == "ReduceMean":
            y = x[0].mean(dim=a["axes"], keepdim=bool(a.get("keepdims", 1)))
        elif op == "Cast":
            y = x[0].to({1: torch.float32, 7: torch.int64, 6: torch.int32, 11: torch.float64}[a["to"]])
        elif op == "Softmax":
            # opset 11 semantics: flatten to 2-D at `axis`, softmax over the trailing block
            ax = a.get("axis", 1)
            shp = x[0].shape
            lead = int(np.prod(shp[:ax])) if ax > 0 else 1
            y = torch.softmax(x[0].reshape(lead, -1), dim=1).reshape(shp)
        else:
            raise NotImplementedError(op)
        env[nd["out"][0]] = y
    if keep_all:
        return env
    names = want or g["outputs"]
    return [env[n].numpy() for n in names]


# ----------------------------------------------------------------------------- DNNFeatureMatcher restated
def convert_image_to_float(img_u8):
    """ConvertImageToFloat (dnnfeaturematcher.cpp:5-9): cv::Mat::convertTo(CV_32F, 1/255.) works in f32."""
    return img_u8.astype(np.float32) * np.float32(1.0 / 255.0)


def decode_matches(conf, threshold, model_width=40, res=16):
    """dnnfeaturematcher.cpp:75-99: strict '>', findNonZero row-major, top-left cell corners."""
    ii, jj = np.nonzero(conf > np.float32(threshold))
    out = np.stack([(ii % model_width) * res, (ii // model_width) * res,
                    (jj % model_width) * res, (jj // model_width) * res], 1).astype(np.int32)
    return out.reshape(-1, 4)


class DNNFeatureMatcherOracle:
    def __init__(self, threshold=0.15, path=MODEL):
        self.g = load_graph(path)
        self.threshold = threshold

    def SetThreshold(self, v):
        self.threshold = v

    def run(self, im1, im2, want=None):
        f0 = convert_image_to_float(im1)[None, None]
        f1 = convert_image_to_float(im2)[None, None]
        return run_graph(self.g, {self.g["inputs"][0]: f0, self.g["inputs"][1]: f1}, want)

    def conf(self, im1, im2):
        return self.run(im1, im2)[0][0]

    def MatchFrames(self, im1, im2):
        return decode_matches(self.conf(im1, im2), self.threshold)


if __name__ == "__main__":
    g = load_graph()
    from collections import Counter
    print(len(g["nodes"]), Counter(n["op"] for n in g["nodes"]).most_common())
    print(g["inputs"], g["outputs"], len(g["init"]), sum(v.size for v in g["init"].values()))
