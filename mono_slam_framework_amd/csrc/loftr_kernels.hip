// LoFTR_teacher kernels (placeholder until the LoFTR path lands; ORB is built first).
#include "loftr_pipeline.h"

namespace msf {
struct LoftrPipeline::Impl {};
LoftrPipeline::~LoftrPipeline() { destroy(); }
void LoftrPipeline::destroy() { delete p_; p_ = nullptr; }
std::string LoftrPipeline::init(const char*, int, bool) { return "LoFTR path is not built in this revision"; }
hipError_t LoftrPipeline::match(int, const uint8_t*, const uint8_t*, long long, int, float, msf_match*, int, int32_t*,
                                hipStream_t) { return hipErrorNotSupported; }
int LoftrPipeline::debug_get(int, int, int, void*, size_t, size_t*, std::string* err) { *err = "LoFTR path is not built"; return MSF_ERR_UNSUPPORTED; }
int LoftrPipeline::stage_times(const char**, float*, int) { return 0; }
}  // namespace msf
