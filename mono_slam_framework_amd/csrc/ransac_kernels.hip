// Match-list consumer: Initializer::CheckHomography / CheckFundamental
// (slam_pipeline/src/Initializer.cc:322-405 and :407-487) evaluated for all RANSAC hypotheses of FindHomography /
// FindFundamental (:152-199, :201-245) in one launch.  The 8-point solves (cv::SVD) stay on the host.
//
// Bit-exactness: every per-match expression is evaluated in f32 in the reference's operation order (the library is
// built with -ffp-contract=off and correctly rounded division), and the score is accumulated by ONE lane in match
// order, two additions per match, exactly like the reference's `score +=` loop -- f32 addition is not associative.
#include <hip/hip_runtime.h>
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
  const float invSigmaSquare = 1.0f / (sigma * sigma);
  if (model == MSF_MODEL_HOMOGRAPHY) {
    const float* I = m12 + 9 * hyp;
    const float i11 = I[0], i12 = I[1], i13 = I[2], i21 = I[3], i22 = I[4], i23 = I[5], i31 = I[6], i32 = I[7], i33 = I[8];
    const float th = 5.991f;
    for (int i = tid; i < n; i += 256) {
      const msf_match q = matches[i];
      const float u1 = (float)q.x1, v1 = (float)q.y1, u2 = (float)q.x2, v2 = (float)q.y2;
      bool bIn = true;
      // reprojection error in the first image, x2in1 = H12 * x2 (Initializer.cc:368-381)
      const float w2in1inv = 1.0f / (i31 * u2 + i32 * v2 + i33);
      const float u2in1 = (i11 * u2 + i12 * v2 + i13) * w2in1inv;
      const float v2in1 = (i21 * u2 + i22 * v2 + i23) * w2in1inv;
      const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
      const float chiSquare1 = squareDist1 * invSigmaSquare;
      float t1 = 0.f;
      if (chiSquare1 > th) bIn = false; else t1 = th - chiSquare1;
      // reprojection error in the second image, x1in2 = H21 * x1 (:386-399)
      const float w1in2inv = 1.0f / (a31 * u1 + a32 * v1 + a33);
      const float u1in2 = (a11 * u1 + a12 * v1 + a13) * w1in2inv;
      const float v1in2 = (a21 * u1 + a22 * v1 + a23) * w1in2inv;
      const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
      const float chiSquare2 = squareDist2 * invSigmaSquare;
      float t2 = 0.f;
      if (chiSquare2 > th) bIn = false; else t2 = th - chiSquare2;
      terms[2 * i] = t1;
      terms[2 * i + 1] = t2;
      inliers[(long long)hyp * n + i] = bIn;
    }
  } else {
    const float th = 3.841f, thScore = 5.991f;
    for (int i = tid; i < n; i += 256) {
      const msf_match q = matches[i];
      const float u1 = (float)q.x1, v1 = (float)q.y1, u2 = (float)q.x2, v2 = (float)q.y2;
      bool bIn = true;
      // l2 = F21 x1 (Initializer.cc:443-456)
      const float a2 = a11 * u1 + a12 * v1 + a13;
      const float b2 = a21 * u1 + a22 * v1 + a23;
      const float c2 = a31 * u1 + a32 * v1 + a33;
      const float num2 = a2 * u2 + b2 * v2 + c2;
      const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
      const float chiSquare1 = squareDist1 * invSigmaSquare;
      float t1 = 0.f;
      if (chiSquare1 > th) bIn = false; else t1 = thScore - chiSquare1;
      // l1 = x2' F21 (:461-474)
      const float a1 = a11 * u2 + a21 * v2 + a31;
      const float b1 = a12 * u2 + a22 * v2 + a32;
      const float c1 = a13 * u2 + a23 * v2 + a33;
      const float num1 = a1 * u1 + b1 * v1 + c1;
      const float squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
      const float chiSquare2 = squareDist2 * invSigmaSquare;
      float t2 = 0.f;
      if (chiSquare2 > th) bIn = false; else t2 = thScore - chiSquare2;
      terms[2 * i] = t1;
      terms[2 * i + 1] = t2;
      inliers[(long long)hyp * n + i] = bIn;
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
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_check_hypotheses), hipFuncAttributeMaxDynamicSharedMemorySize,
                        2 * kMaxRansacMatches * (int)sizeof(float));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_check_hypotheses, dim3(n_hyp), dim3(256), lds, st, model, d_m21, d_m12, n, d_matches, sigma,
                     d_scores, d_inliers);
  return hipGetLastError();
}

}  // namespace msf
