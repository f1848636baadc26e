/*
 * oracle/orb_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the ORB FeatureMatcher path of the reference
 * (/root/reference/src/featurematcher.cpp:3-45) including the arithmetic it
 * delegates to OpenCV (cv::ORB::create() defaults + BruteForce-Hamming
 * knnMatch(k=2)); see orb_oracle.c for the per-function citations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (libmsf.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors and OpenCV is
 * neither vendored nor installed, so this restatement
 * is pinned only by the published OpenCV algorithm as restated here.
 */
#ifndef ORACLE_ORB_ORACLE_H
#define ORACLE_ORB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORB_ORACLE_MAX_LEVELS 8

typedef struct orb_oracle_opts {
  int32_t nfeatures;       /* 500  (cv::ORB::create default) */
  int32_t nlevels;         /* 8 */
  int32_t fast_threshold;  /* 20 */
  int32_t edge_threshold;  /* 31 */
  int32_t blur_tie_even;   /* 1: column pass rounds ties to even (OpenCV SIMD f32 column
                              filter); 0: (sum + 32768) >> 16 (scalar FixedPtCastEx) */
  int32_t level_size_mul_inv; /* 0: level size = cvRound(W / scale_l) (ORB_Impl::detectAndCompute of OpenCV 3.x / 4.x:
                              `Size sz(cvRound(image.cols/scale), cvRound(image.rows/scale))`);
                              1: cvRound(W * (1.f / scale_l)) (the 2.4-era computeImagePyramid).  The two differ for
                              139 widths in [64, 4096] (69 is the smallest), for none of 640/480/1280/720/1920/1080 */
  int32_t blur_kernel_sum256; /* 0: the sepFilter2D integer kernel 18 34 49 55 49 34 18 (sum 257, SURVEY.md A.6 default);
                              1: OpenCV's bit-exact fixed-point Gaussian kernel 18 34 48 56 48 34 18 (sum 256), rounding
                              half up whatever blur_tie_even says (SURVEY.md A.6 "keep behind a switch") */
} orb_oracle_opts;

/* one keypoint, 32 bytes */
typedef struct orb_oracle_kp {
  float x, y;        /* level-0 coordinates: pt * scale_l (f32), as cv::KeyPoint::pt */
  float response;    /* Harris response */
  float angle;       /* degrees, fastAtan2 */
  int32_t octave;    /* pyramid level */
  int32_t lx, ly;    /* integer coordinates inside the level */
  int32_t fast_score;
} orb_oracle_kp;

typedef struct orb_oracle_ctx orb_oracle_ctx;

void orb_oracle_default_opts(orb_oracle_opts* o);
orb_oracle_ctx* orb_oracle_create(int width, int height, const orb_oracle_opts* opts);
void orb_oracle_destroy(orb_oracle_ctx* c);

/* full detectAndCompute; returns number of keypoints (>= 0) or < 0 on error */
int orb_oracle_extract(orb_oracle_ctx* c, const uint8_t* img, ptrdiff_t stride);

int orb_oracle_num_keypoints(const orb_oracle_ctx* c);
const orb_oracle_kp* orb_oracle_keypoints(const orb_oracle_ctx* c);
const uint8_t* orb_oracle_descriptors(const orb_oracle_ctx* c); /* [n][32] */

/* intermediates, for stage-level parity */
int orb_oracle_level_size(const orb_oracle_ctx* c, int level, int* w, int* h);
float orb_oracle_level_scale(const orb_oracle_ctx* c, int level);
int orb_oracle_level_quota(const orb_oracle_ctx* c, int level);
const uint8_t* orb_oracle_level_pixels(const orb_oracle_ctx* c, int level);   /* unblurred, stride = w */
const uint8_t* orb_oracle_level_blurred(const orb_oracle_ctx* c, int level);  /* blurred,   stride = w */
/* FAST candidates after NMS + border reject, row-major; triplets (x, y, score) */
int orb_oracle_fast_candidates(const orb_oracle_ctx* c, int level, const int32_t** xys);
/* uint8 [h][w] FAST score of every pixel of a level before NMS / border reject (0 = not a corner); after extract */
int orb_oracle_fast_score_map(const orb_oracle_ctx* c, int level, uint8_t* out);
/* keypoints after retainBest(2N) by FAST score + Harris, before the Harris cull (all levels) */
int orb_oracle_stage1_keypoints(const orb_oracle_ctx* c, const orb_oracle_kp** kps);

/* BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) + the reference's ratio filter
 * (featurematcher.cpp:29-42).  out = (x1,y1,x2,y2) int32 quadruples, returns
 * the number of matches (may exceed cap; only cap are written). */
int orb_oracle_knn_match(const uint8_t* d1, int n1, const orb_oracle_kp* k1,
                         const uint8_t* d2, int n2, const orb_oracle_kp* k2,
                         float ratio, int32_t* out, int cap);
/* raw 2-NN table: per query (idx0, d0, idx1, d1) */
void orb_oracle_knn2(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int32_t* out);

/* FeatureMatcher::MatchFrames on two images with two contexts of equal size */
int orb_oracle_match_frames(orb_oracle_ctx* ca, orb_oracle_ctx* cb,
                            const uint8_t* a, ptrdiff_t stride_a,
                            const uint8_t* b, ptrdiff_t stride_b,
                            float ratio, int32_t* out, int cap);

/* the shared deterministic sin/cos (double polynomial rounded to f32) */
void orb_oracle_sincosf(float t, float* s, float* c);
float orb_oracle_fast_atan2(float y, float x);

#ifdef __cplusplus
}
#endif
#endif
