// Match-list consumer: Initializer::CheckHomography / CheckFundamental
// (slam_pipeline/src/Initializer.cc:322-405 and :407-487) evaluated for all RANSAC hypotheses of FindHomography /
// FindFundamental (:152-199, :201-245) in one launch.  The 8-point solves (cv::SVD) stay on the host.
//
// Bit-exactness: every per-match expression is evaluated in f32 in the reference's operation order (the library is
// built with -ffp-contract=off and correctly rounded division), and the score is accumulated by ONE lane in match
// order, two additions per match, exactly like the reference's `score +=` loop -- f32 addition is not associative.
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdint.h>

#include "msf_abi.h"

namespace msf {

constexpr int kMaxRansacMatches = 8192;   // 2 x f32 per match in LDS

// one workgroup per hypothesis
__global__ __launch_bounds__(256) void k_check_hypotheses(int model, const float* __restrict__ m21,
                                                          const float* __restrict__ m12, int n,
                                                          const msf_match* __restrict__ matches, float sigma,
                                                          float* __restrict__ scores, uint8_t* __restrict__ inliers) {
  extern __shared__ float terms[];   // [2 * n]: what the reference adds to `score` for match i (0 when it skips)
  const int hyp = blockIdx.x, tid = threadIdx.x;
  const float* M = m21 + 9 * hyp;
  const float a11 = M[0], a12 = M[1], a13 = M[2], a21 = M[3], a22 = M[4], a23 = M[5], a31 = M[6], a32 = M[7], a33 = M[8];
  const float inv_var = 1.0f / (sigma * sigma);
  if (model == MSF_MODEL_HOMOGRAPHY) {
    const float* I = m12 + 9 * hyp;
    const float i11 = I[0], i12 = I[1], i13 = I[2], i21 = I[3], i22 = I[4], i23 = I[5], i31 = I[6], i32 = I[7], i33 = I[8];
    const float th = 5.991f;
    for (int i = tid; i < n; i += 256) {
      const msf_match q = matches[i];
      const float u1 = (float)q.x1, v1 = (float)q.y1, u2 = (float)q.x2, v2 = (float)q.y2;
      bool consistent = true;
      // reprojection error in the first image, x2in1 = H12 * x2 (Initializer.cc:368-381)
      const float back_w = 1.0f / (i31 * u2 + i32 * v2 + i33);
      const float back_x = (i11 * u2 + i12 * v2 + i13) * back_w;
      const float back_y = (i21 * u2 + i22 * v2 + i23) * back_w;
      const float d2_first = (u1 - back_x) * (u1 - back_x) + (v1 - back_y) * (v1 - back_y);
      const float chi_first = d2_first * inv_var;
      float t1 = 0.f;
      if (chi_first > th) consistent = false; else t1 = th - chi_first;
      // reprojection error in the second image, x1in2 = H21 * x1 (:386-399)
      const float fwd_w = 1.0f / (a31 * u1 + a32 * v1 + a33);
      const float fwd_x = (a11 * u1 + a12 * v1 + a13) * fwd_w;
      const float fwd_y = (a21 * u1 + a22 * v1 + a23) * fwd_w;
      const float d2_second = (u2 - fwd_x) * (u2 - fwd_x) + (v2 - fwd_y) * (v2 - fwd_y);
      const float chi_second = d2_second * inv_var;
      float t2 = 0.f;
      if (chi_second > th) consistent = false; else t2 = th - chi_second;
      terms[2 * i] = t1;
      terms[2 * i + 1] = t2;
      inliers[(long long)hyp * n + i] = consistent;
    }
  } else {
    const float th = 3.841f, gain_cut = 5.991f;
    for (int i = tid; i < n; i += 256) {
      const msf_match q = matches[i];
      const float u1 = (float)q.x1, v1 = (float)q.y1, u2 = (float)q.x2, v2 = (float)q.y2;
      bool consistent = true;
      // l2 = F21 x1 (Initializer.cc:443-456)
      const float l2a = a11 * u1 + a12 * v1 + a13;
      const float l2b = a21 * u1 + a22 * v1 + a23;
      const float l2c = a31 * u1 + a32 * v1 + a33;
      const float line2_dot = l2a * u2 + l2b * v2 + l2c;
      const float d2_first = line2_dot * line2_dot / (l2a * l2a + l2b * l2b);
      const float chi_first = d2_first * inv_var;
      float t1 = 0.f;
      if (chi_first > th) consistent = false; else t1 = gain_cut - chi_first;
      // l1 = x2' F21 (:461-474)
      const float l1a = a11 * u2 + a21 * v2 + a31;
      const float l1b = a12 * u2 + a22 * v2 + a32;
      const float l1c = a13 * u2 + a23 * v2 + a33;
      const float line1_dot = l1a * u1 + l1b * v1 + l1c;
      const float d2_second = line1_dot * line1_dot / (l1a * l1a + l1b * l1b);
      const float chi_second = d2_second * inv_var;
      float t2 = 0.f;
      if (chi_second > th) consistent = false; else t2 = gain_cut - chi_second;
      terms[2 * i] = t1;
      terms[2 * i + 1] = t2;
      inliers[(long long)hyp * n + i] = consistent;
    }
  }
  __syncthreads();
  if (tid == 0) {
    // the reference's accumulation order; a skipped term was stored as +0.0f, and score + 0.0f == score bit for bit
    // (score is never -0: it starts at +0 and only receives th - chi >= 0 or NaN)
    float score = 0.0f;
    for (int i = 0; i < 2 * n; i++) score += terms[i];
    scores[hyp] = score;
  }
}

hipError_t check_hypotheses(int model, int n_hyp, const float* d_m21, const float* d_m12, int n,
                            const msf_match* d_matches, float sigma, float* d_scores, uint8_t* d_inliers,
                            hipStream_t st) {
  if (n_hyp <= 0) return hipSuccess;
  if (n > kMaxRansacMatches) return hipErrorInvalidValue;
  const size_t lds = (size_t)2 * (n > 0 ? n : 1) * sizeof(float);
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_check_hypotheses), hipFuncAttributeMaxDynamicSharedMemorySize,
                        2 * kMaxRansacMatches * (int)sizeof(float));
  });
  hipLaunchKernelGGL(k_check_hypotheses, dim3(n_hyp), dim3(256), lds, st, model, d_m21, d_m12, n, d_matches, sigma,
                     d_scores, d_inliers);
  return hipGetLastError();
}

}  // namespace msf
