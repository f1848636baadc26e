// LoFTR_teacher weights: the reference's model file or the packed blob made from it.
//
// ::DNNFeatureMatcher opens the ONNX file its caller names (src/dnnfeaturematcher.cpp:11-21,
// src/main.cpp:61-63 "model/LoFTR_teacher.onnx").  load_weights() accepts that file directly -- a schema-less
// reader of the ONNX protobuf wire format (ModelProto.graph -> initializers, Conv / MatMul / Constant nodes; all
// tensors of this model are f32 raw_data) -- and also the MSFLTR01 blob (the same tensors under fixed names), which
// is only a cache: both yield byte-identical tensors (tests/test_weights_io.py).
//
// Tensor names (blob and in-memory): conv00.w .. conv19.w [out][in][kh][kw], conv00.b .. conv19.b, outconv.w,
// pe [32][30][40], blk0..7.{wq,wk,wv,wmerge,wmlp0,wmlp1} [in][out], ln0..3.{n1w,n1b,n2w,n2b}.
#pragma once

#include <stdint.h>

#include <map>
#include <string>
#include <vector>

namespace msf {

struct WeightTensor {
  std::vector<uint32_t> dims;
  std::vector<float> data;
};
using WeightMap = std::map<std::string, WeightTensor>;

// Empty string on success, else "io: ..." (file missing / malformed / not the LoFTR_teacher topology).
std::string load_weights(const std::string& path, WeightMap* out);
std::string save_blob(const std::string& path, const WeightMap& w);
// FNV-1a 64 over names, dims and f32 bytes in name order: equal digests <=> identical tensors
uint64_t weights_digest(const WeightMap& w, int64_t* n_floats);

}  // namespace msf
