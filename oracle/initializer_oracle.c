/* TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into or called by the product library.
 *
 * Scalar restatement of Initializer::CheckHomography and ::CheckFundamental
 * (slam_pipeline/src/Initializer.cc:322-405 and :407-487) and of the keep-the-best loop around them in
 * FindHomography / FindFundamental (:178-196, :224-242), in f32 with the reference's operation order.
 * Build with -ffp-contract=off (oracle/Makefile) so no multiply-add is fused.
 * Parity: unpinned -- the reference has no tests or golden vectors for the Initializer; the statements below follow
 * the source line by line. */
#include <stdint.h>

/* key points are MatchFramesResult's integer points converted to float (Initializer.cc:79-80); m: [n][4] x1 y1 x2 y2 */
float initializer_check_homography(const float* H21, const float* H12, int n, const int32_t* m, float sigma,
                                   uint8_t* inliers) {
  const float h11 = H21[0], h12 = H21[1], h13 = H21[2], h21 = H21[3], h22 = H21[4], h23 = H21[5], h31 = H21[6],
              h32 = H21[7], h33 = H21[8];
  const float h11inv = H12[0], h12inv = H12[1], h13inv = H12[2], h21inv = H12[3], h22inv = H12[4], h23inv = H12[5],
              h31inv = H12[6], h32inv = H12[7], h33inv = H12[8];
  float score = 0.0f;
  const float th = 5.991f;
  const float invSigmaSquare = 1.0f / (sigma * sigma);
  for (int i = 0; i < n; i++) {
    int bIn = 1;
    const float u1 = (float)m[4 * i], v1 = (float)m[4 * i + 1], u2 = (float)m[4 * i + 2], v2 = (float)m[4 * i + 3];
    const float w2in1inv = 1.0f / (h31inv * u2 + h32inv * v2 + h33inv);
    const float u2in1 = (h11inv * u2 + h12inv * v2 + h13inv) * w2in1inv;
    const float v2in1 = (h21inv * u2 + h22inv * v2 + h23inv) * w2in1inv;
    const float squareDist1 = (u1 - u2in1) * (u1 - u2in1) + (v1 - v2in1) * (v1 - v2in1);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) bIn = 0; else score += th - chiSquare1;
    const float w1in2inv = 1.0f / (h31 * u1 + h32 * v1 + h33);
    const float u1in2 = (h11 * u1 + h12 * v1 + h13) * w1in2inv;
    const float v1in2 = (h21 * u1 + h22 * v1 + h23) * w1in2inv;
    const float squareDist2 = (u2 - u1in2) * (u2 - u1in2) + (v2 - v1in2) * (v2 - v1in2);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) bIn = 0; else score += th - chiSquare2;
    inliers[i] = (uint8_t)bIn;
  }
  return score;
}

float initializer_check_fundamental(const float* F21, int n, const int32_t* m, float sigma, uint8_t* inliers) {
  const float f11 = F21[0], f12 = F21[1], f13 = F21[2], f21 = F21[3], f22 = F21[4], f23 = F21[5], f31 = F21[6],
              f32 = F21[7], f33 = F21[8];
  float score = 0.0f;
  const float th = 3.841f;
  const float thScore = 5.991f;
  const float invSigmaSquare = 1.0f / (sigma * sigma);
  for (int i = 0; i < n; i++) {
    int bIn = 1;
    const float u1 = (float)m[4 * i], v1 = (float)m[4 * i + 1], u2 = (float)m[4 * i + 2], v2 = (float)m[4 * i + 3];
    const float a2 = f11 * u1 + f12 * v1 + f13;
    const float b2 = f21 * u1 + f22 * v1 + f23;
    const float c2 = f31 * u1 + f32 * v1 + f33;
    const float num2 = a2 * u2 + b2 * v2 + c2;
    const float squareDist1 = num2 * num2 / (a2 * a2 + b2 * b2);
    const float chiSquare1 = squareDist1 * invSigmaSquare;
    if (chiSquare1 > th) bIn = 0; else score += thScore - chiSquare1;
    const float a1 = f11 * u2 + f21 * v2 + f31;
    const float b1 = f12 * u2 + f22 * v2 + f32;
    const float c1 = f13 * u2 + f23 * v2 + f33;
    const float num1 = a1 * u1 + b1 * v1 + c1;
    const float squareDist2 = num1 * num1 / (a1 * a1 + b1 * b1);
    const float chiSquare2 = squareDist2 * invSigmaSquare;
    if (chiSquare2 > th) bIn = 0; else score += thScore - chiSquare2;
    inliers[i] = (uint8_t)bIn;
  }
  return score;
}

/* the RANSAC loop's bookkeeping (Initializer.cc:166-196 / :213-242) over precomputed hypotheses:
 * returns the kept hypothesis (-1: none scored above 0), fills scores[n_hyp] and best_inliers[n] (zeros if none) */
int initializer_find_best(int model, int n_hyp, const float* m21, const float* m12, int n, const int32_t* m,
                          float sigma, float* scores, uint8_t* best_inliers, uint8_t* scratch) {
  float score = 0.0f;
  int best = -1;
  for (int i = 0; i < n; i++) best_inliers[i] = 0;
  for (int it = 0; it < n_hyp; it++) {
    const float cur = model == 0 ? initializer_check_homography(m21 + 9 * it, m12 + 9 * it, n, m, sigma, scratch)
                                 : initializer_check_fundamental(m21 + 9 * it, n, m, sigma, scratch);
    scores[it] = cur;
    if (cur > score) {
      for (int i = 0; i < n; i++) best_inliers[i] = scratch[i];
      score = cur;
      best = it;
    }
  }
  return best;
}
