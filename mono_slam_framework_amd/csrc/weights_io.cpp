// see weights_io.h
#include "weights_io.h"

#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace msf {
namespace {

// ------------------------------------------------------------------ file helpers
std::string read_file(const std::string& path, std::vector<uint8_t>* buf) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "io: cannot open weights file " + path;
  if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return "io: cannot seek in " + path; }
  const long sz = ftell(f);
  if (sz < 12 || sz > (64l << 20)) { fclose(f); return "io: implausible size of weights file " + path; }
  rewind(f);
  buf->resize((size_t)sz);
  const size_t got = fread(buf->data(), 1, (size_t)sz, f);
  fclose(f);
  if (got != (size_t)sz) return "io: short read of " + path;
  return "";
}

// ------------------------------------------------------------------ MSFLTR01 blob
struct BlobRec { char name[32]; uint32_t ndim, dims[4], off, count; };
static_assert(sizeof(BlobRec) == 60, "record layout");

std::string parse_blob(const std::vector<uint8_t>& b, WeightMap* out) {
  uint32_t n = 0;
  memcpy(&n, b.data() + 8, 4);
  if (n == 0 || n > 4096 || 12 + (size_t)n * sizeof(BlobRec) > b.size()) return "io: bad weights header";
  const size_t payload = 12 + (size_t)n * sizeof(BlobRec);
  const size_t avail = (b.size() - payload) / 4;
  for (uint32_t i = 0; i < n; i++) {
    BlobRec r;
    memcpy(&r, b.data() + 12 + (size_t)i * sizeof(BlobRec), sizeof r);
    if (r.ndim < 1 || r.ndim > 4 || (size_t)r.off + r.count > avail) return "io: weights record out of bounds";
    size_t prod = 1;
    for (uint32_t d = 0; d < r.ndim; d++) prod *= r.dims[d];
    if (prod != r.count) return "io: weights record shape does not match its size";
    WeightTensor t;
    t.dims.assign(r.dims, r.dims + r.ndim);
    t.data.resize(r.count);
    memcpy(t.data.data(), b.data() + payload + (size_t)r.off * 4, (size_t)r.count * 4);
    (*out)[std::string(r.name, strnlen(r.name, 32))] = std::move(t);
  }
  return "";
}

// ------------------------------------------------------------------ protobuf wire format (what ONNX needs of it)
struct Span { const uint8_t* p; size_t n; };

bool varint(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
  uint64_t r = 0;
  for (int s = 0; s < 70 && p < end; s += 7) {
    const uint8_t c = *p++;
    r |= (uint64_t)(c & 0x7F) << s;
    if (!(c & 0x80)) { *v = r; return true; }
  }
  return false;
}

// Calls fn(field, wire_type, varint value, bytes) for every field of a message; false on malformed input.
template <class F>
bool fields(Span m, F&& fn) {
  const uint8_t* p = m.p;
  const uint8_t* end = m.p + m.n;
  while (p < end) {
    uint64_t key;
    if (!varint(p, end, &key)) return false;
    const uint32_t f = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
    uint64_t v = 0;
    Span s{nullptr, 0};
    if (wt == 0) {
      if (!varint(p, end, &v)) return false;
    } else if (wt == 1) {
      if (end - p < 8) return false;
      s = {p, 8};
      p += 8;
    } else if (wt == 2) {
      uint64_t len;
      if (!varint(p, end, &len) || len > (uint64_t)(end - p)) return false;
      s = {p, (size_t)len};
      p += len;
    } else if (wt == 5) {
      if (end - p < 4) return false;
      s = {p, 4};
      p += 4;
    } else {
      return false;
    }
    if (!fn(f, wt, v, s)) return false;
  }
  return true;
}

struct OnnxTensor {
  std::string name;
  std::vector<int64_t> dims;
  int dtype = 0;
  std::vector<float> data;
};

// TensorProto {1 dims*, 2 data_type, 4 float_data*, 8 name, 9 raw_data}
bool parse_tensor(Span m, OnnxTensor* t) {
  Span raw{nullptr, 0};
  std::vector<float> fl;
  const bool ok = fields(m, [&](uint32_t f, uint32_t wt, uint64_t v, Span s) {
    if (f == 1) {
      if (wt == 0) {
        t->dims.push_back((int64_t)v);
      } else {
        const uint8_t* p = s.p;
        uint64_t x;
        while (p < s.p + s.n) {
          if (!varint(p, s.p + s.n, &x)) return false;
          t->dims.push_back((int64_t)x);
        }
      }
    } else if (f == 2) {
      t->dtype = (int)v;
    } else if (f == 4) {
      if (wt == 5) { float x; memcpy(&x, s.p, 4); fl.push_back(x); }
      else if (wt == 2) { const size_t n = s.n / 4; const size_t o = fl.size(); fl.resize(o + n); memcpy(fl.data() + o, s.p, n * 4); }
    } else if (f == 8) {
      t->name.assign((const char*)s.p, s.n);
    } else if (f == 9) {
      raw = s;
    }
    return true;
  });
  if (!ok) return false;
  if (t->dtype == 1) {   // FLOAT
    if (raw.n) { t->data.resize(raw.n / 4); memcpy(t->data.data(), raw.p, t->data.size() * 4); }
    else t->data = std::move(fl);
  }
  return true;
}

struct OnnxNode {
  std::string op;
  std::vector<std::string> in, out;
  OnnxTensor value;   // Constant: attribute "value"
  bool has_value = false;
};

// NodeProto {1 input*, 2 output*, 4 op_type, 5 attribute* {1 name, 5 t}}
bool parse_node(Span m, OnnxNode* nd) {
  return fields(m, [&](uint32_t f, uint32_t, uint64_t, Span s) {
    if (f == 1) nd->in.emplace_back((const char*)s.p, s.n);
    else if (f == 2) nd->out.emplace_back((const char*)s.p, s.n);
    else if (f == 4) nd->op.assign((const char*)s.p, s.n);
    else if (f == 5) {
      std::string an;
      Span tp{nullptr, 0};
      if (!fields(s, [&](uint32_t af, uint32_t, uint64_t, Span as) {
            if (af == 1) an.assign((const char*)as.p, as.n);
            else if (af == 5) tp = as;
            return true;
          })) return false;
      if (an == "value" && tp.p && (nd->op == "Constant" || nd->op.empty())) {
        // op_type may come after the attributes on the wire: keep the tensor, the caller checks the op
        if (!parse_tensor(tp, &nd->value)) return false;
        nd->has_value = true;
      }
    }
    return true;
  });
}

std::string parse_onnx(const std::vector<uint8_t>& b, WeightMap* out) {
  Span graph{nullptr, 0};
  if (!fields(Span{b.data(), b.size()}, [&](uint32_t f, uint32_t wt, uint64_t, Span s) {
        if (f == 7 && wt == 2) graph = s;   // ModelProto.graph
        return true;
      }) || !graph.p)
    return "io: not an ONNX model (no graph)";
  std::map<std::string, OnnxTensor> init;
  std::vector<OnnxNode> nodes;
  if (!fields(graph, [&](uint32_t f, uint32_t wt, uint64_t, Span s) {
        if (f == 1 && wt == 2) {
          OnnxNode nd;
          if (!parse_node(s, &nd)) return false;
          if (nd.op == "Conv" || nd.op == "MatMul" || (nd.op == "Constant" && nd.has_value)) nodes.push_back(std::move(nd));
        } else if (f == 5 && wt == 2) {
          OnnxTensor t;
          if (!parse_tensor(s, &t)) return false;
          init[t.name] = std::move(t);
        }
        return true;
      }))
    return "io: malformed ONNX graph";

  auto put = [&](const std::string& name, const OnnxTensor& t, std::vector<uint32_t> dims) -> std::string {
    size_t prod = 1;
    for (uint32_t d : dims) prod *= d;
    if (t.dtype != 1 || t.data.size() != prod) return "io: ONNX tensor for " + name + " is not f32 of the expected size";
    (*out)[name] = WeightTensor{std::move(dims), t.data};
    return "";
  };
  // backbone: the Conv nodes in execution order: weight [out][in][k][k], bias [out];
  // the last one (layer4_outconv, 1x1) has no bias
  static const int kConv[21][3] = {{1, 8, 7},   {8, 8, 3},   {8, 8, 3},   {8, 8, 3},   {8, 8, 3},   {8, 16, 3},  {16, 16, 3},
                                   {8, 16, 1},  {16, 16, 3}, {16, 16, 3}, {16, 32, 3}, {32, 32, 3}, {16, 32, 1}, {32, 32, 3},
                                   {32, 32, 3}, {32, 32, 3}, {32, 32, 3}, {32, 32, 1}, {32, 32, 3}, {32, 32, 3}, {32, 32, 1}};
  int nconv = 0, nmm = 0;
  bool have_pe = false;
  char nm[32];
  for (const OnnxNode& nd : nodes) {
    if (nd.op == "Conv") {
      if (nconv >= 21 || nd.in.size() < 2) return "io: ONNX graph is not LoFTR_teacher (convolutions)";
      auto w = init.find(nd.in[1]);
      if (w == init.end()) return "io: ONNX Conv weight is not an initializer";
      const uint32_t ci = kConv[nconv][0], co = kConv[nconv][1], ks = kConv[nconv][2];
      if (nconv < 20) snprintf(nm, sizeof nm, "conv%02d.w", nconv); else snprintf(nm, sizeof nm, "outconv.w");
      std::string e = put(nm, w->second, {co, ci, ks, ks});
      if (!e.empty()) return e;
      if (nconv < 20) {
        if (nd.in.size() < 3 || init.find(nd.in[2]) == init.end()) return "io: ONNX Conv bias missing";
        snprintf(nm, sizeof nm, "conv%02d.b", nconv);
        e = put(nm, init[nd.in[2]], {co});
        if (!e.empty()) return e;
      }
      nconv++;
    } else if (nd.op == "MatMul") {
      // the 48 weight products of the 8 encoder blocks, in graph order: q, k, v, merge, mlp0, mlp1 (Appendix C.3);
      // the other MatMuls (K^T V, Q KV, the similarity) have no initializer operand
      if (nd.in.size() < 2) continue;
      auto w = init.find(nd.in[1]);
      if (w == init.end()) continue;
      if (nmm >= 48) return "io: ONNX graph is not LoFTR_teacher (linear layers)";
      static const char* kNames[6] = {"wq", "wk", "wv", "wmerge", "wmlp0", "wmlp1"};
      static const uint32_t kIn[6] = {32, 32, 32, 32, 64, 64}, kOut[6] = {32, 32, 32, 32, 64, 32};
      const int j = nmm % 6;
      snprintf(nm, sizeof nm, "blk%d.%s", nmm / 6, kNames[j]);
      const std::string e = put(nm, w->second, {kIn[j], kOut[j]});
      if (!e.empty()) return e;
      nmm++;
    } else if (!have_pe && nd.value.dims.size() == 4 && nd.value.dims[0] == 1 && nd.value.dims[1] == 32 &&
               nd.value.dims[2] == 30 && nd.value.dims[3] == 40) {
      // positional encoding: the first [1,32,30,40] Constant (the graph holds it twice, once per image)
      const std::string e = put("pe", nd.value, {32, 30, 40});
      if (!e.empty()) return e;
      have_pe = true;
    }
  }
  if (nconv != 21 || nmm != 48 || !have_pe) return "io: ONNX graph is not LoFTR_teacher (21 Conv, 48 weight MatMul, PE expected)";
  // LayerNorm parameters keep their module names through the export
  for (int L = 0; L < 4; L++) {
    static const char* kSrc[4] = {"norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"};
    static const char* kDst[4] = {"n1w", "n1b", "n2w", "n2b"};
    for (int k = 0; k < 4; k++) {
      char src[64];
      snprintf(src, sizeof src, "loftr_coarse.layers.%d.%s", L, kSrc[k]);
      auto it = init.find(src);
      if (it == init.end()) return std::string("io: ONNX model lacks ") + src;
      snprintf(nm, sizeof nm, "ln%d.%s", L, kDst[k]);
      const std::string e = put(nm, it->second, {32});
      if (!e.empty()) return e;
    }
  }
  return "";
}

}  // namespace

std::string load_weights(const std::string& path, WeightMap* out) {
  std::vector<uint8_t> buf;
  std::string err = read_file(path, &buf);
  if (!err.empty()) return err;
  out->clear();
  if (memcmp(buf.data(), "MSFLTR01", 8) == 0) err = parse_blob(buf, out);
  else err = parse_onnx(buf, out);
  if (!err.empty()) return err + " (" + path + ")";
  return "";
}

std::string save_blob(const std::string& path, const WeightMap& w) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return "io: cannot create " + path;
  const uint32_t n = (uint32_t)w.size();
  bool ok = fwrite("MSFLTR01", 1, 8, f) == 8 && fwrite(&n, 4, 1, f) == 1;
  uint32_t off = 0;
  for (const auto& kv : w) {
    BlobRec r{};
    strncpy(r.name, kv.first.c_str(), sizeof r.name - 1);
    r.ndim = (uint32_t)kv.second.dims.size();
    for (uint32_t d = 0; d < 4; d++) r.dims[d] = d < r.ndim ? kv.second.dims[d] : 1u;
    r.off = off;
    r.count = (uint32_t)kv.second.data.size();
    off += r.count;
    ok = ok && fwrite(&r, sizeof r, 1, f) == 1;
  }
  for (const auto& kv : w) ok = ok && fwrite(kv.second.data.data(), 4, kv.second.data.size(), f) == kv.second.data.size();
  ok = fclose(f) == 0 && ok;
  return ok ? "" : "io: write error on " + path;
}

uint64_t weights_digest(const WeightMap& w, int64_t* n_floats) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  };
  int64_t total = 0;
  for (const auto& kv : w) {
    mix(kv.first.data(), kv.first.size());
    mix(kv.second.dims.data(), kv.second.dims.size() * 4);
    mix(kv.second.data.data(), kv.second.data.size() * 4);
    total += (int64_t)kv.second.data.size();
  }
  if (n_floats) *n_floats = total;
  return h;
}

}  // namespace msf
