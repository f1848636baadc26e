/*
 * oracle/orb_oracle.c -- TEST INFRASTRUCTURE ONLY (see orb_oracle.h).
 *
 * Scalar CPU restatement of what the reference's ORB matcher computes:
 *   /root/reference/src/featurematcher.cpp:3-6    cv::ORB::create() defaults,
 *                                                 "BruteForce-Hamming"
 *   /root/reference/src/featurematcher.cpp:10-45  MatchFrames: all-255 mask,
 *       detectAndCompute x2, knnMatch(k=2), ratio test, int truncation.
 * The arithmetic lives in OpenCV (un-vendored, version unpinned:
 * /root/reference/CMakeLists.txt:44).  Each function below names the upstream
 * OpenCV 4.x routine whose published algorithm it restates (SURVEY.md
 * Appendix A).  PARITY UNPINNED: no OpenCV and no reference golden vectors
 * exist in this environment.
 *
 * Deliberate, documented definitions (DESIGN.md "ORB definitions"):
 *  - keypoint order after retainBest is canonical (level, y, x); OpenCV's is
 *    whatever std::nth_element leaves (implementation-defined), the SET is the same.
 *  - cos/sin of the orientation use orb_oracle_sincosf (double polynomial
 *    rounded to f32) instead of the host libm, so CPU and GPU agree bit for bit.
 *  - train sets with fewer than 2 descriptors give no matches (the reference
 *    indexes matches[i][1] unconditionally there: UB).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (see oracle/Makefile).
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_pattern.h"

#define NL ORB_ORACLE_MAX_LEVELS
#define PATCH_SIZE 31
#define HALF_PATCH 15
#define HARRIS_BLOCK 7
#define HARRIS_K 0.04f

typedef struct {
  int32_t x, y, score;
} cand_t;

struct orb_oracle_ctx {
  orb_oracle_opts o;
  int w0, h0;
  int lw[NL], lh[NL];
  float scale[NL];
  int quota[NL];
  int umax[HALF_PATCH + 2];
  uint8_t* lvl[NL];
  uint8_t* blr[NL];
  /* resize tables */
  int* xofs[NL];
  int* yofs[NL];
  uint16_t* xw[NL]; /* w1 (8.8), w0 = 256 - w1 */
  uint16_t* yw[NL];
  /* per level candidates */
  cand_t* cand[NL];
  int ncand[NL];
  int cand_cap[NL];
  /* stage-1 keypoints (after retainBest(2N) + Harris) */
  orb_oracle_kp* s1;
  int ns1, s1_cap;
  int s1_count[NL];
  /* final */
  orb_oracle_kp* kp;
  int nkp, kp_cap;
  uint8_t* desc;
  int desc_cap;
};

/* ---------------------------------------------------------------- helpers */

/* cvRound(float/double): round half to even (SSE cvtss2si semantics). */
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

void orb_oracle_default_opts(orb_oracle_opts* o) {
  o->nfeatures = 500;
  o->nlevels = 8;
  o->fast_threshold = 20;
  o->edge_threshold = 31;
  o->blur_tie_even = 1;
  o->level_size_mul_inv = 0;
  o->blur_kernel_sum256 = 0;
}

/* Deterministic sin/cos for t in [0, ~2*pi]: Cody-Waite reduction by pi/2 and
 * the fdlibm kernel polynomials, every operation a plain IEEE double add/mul
 * (no FMA), result rounded once to f32.  Stands in for cosf/sinf in
 * computeOrbDescriptors (OpenCV orb.cpp: "float a = (float)cos(angle), b =
 * (float)sin(angle)").  The HIP side spells the same operation sequence. */
static double k_sin(double r) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = r * r;
  double p = S6;
  p = p * z + S5;
  p = p * z + S4;
  p = p * z + S3;
  p = p * z + S2;
  p = p * z + S1;
  return r + (r * z) * p;
}
static double k_cos(double r) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = r * r;
  double p = C6;
  p = p * z + C5;
  p = p * z + C4;
  p = p * z + C3;
  p = p * z + C2;
  p = p * z + C1;
  return (1.0 - 0.5 * z) + (z * z) * p;
}
void orb_oracle_sincosf(float t, float* s, float* c) {
  const double TWO_OVER_PI = 6.36619772367581382433e-01;
  const double PIO2_HI = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
  const double PIO2_LO = 6.07710050650619224932e-11;
  double x = (double)t;
  int k = (int)(x * TWO_OVER_PI + 0.5);
  double kd = (double)k;
  double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
  double sr = k_sin(r), cr = k_cos(r);
  double sv, cv;
  switch (k & 3) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
  }
  *s = (float)sv;
  *c = (float)cv;
}

/* cv::fastAtan2 scalar path (OpenCV core mathfuncs_core.simd.hpp atan_f32). */
float orb_oracle_fast_atan2(float y, float x) {
  const float k = (float)(180.0 / 3.14159265358979323846);
  const float p1 = 0.9997878412794807f * k, p3 = -0.3258083974640975f * k,
              p5 = 0.1555786518463281f * k, p7 = -0.04432655554792128f * k;
  float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

/* interpolationLinear<uchar>::getCoeffs (OpenCV imgproc resize.cpp, the
 * INTER_LINEAR_EXACT path): 8.8 fixed-point weights from IEEE double maths. */
static void make_resize_table(int src, int dst, int* ofs, uint16_t* w1) {
  double inv_scale = (double)dst / (double)src; /* cv::resize: dsize.width / ssize.width */
  double scale = 1.0 / inv_scale;
  for (int d = 0; d < dst; d++) {
    double f = scale * ((double)d + 0.5) - 0.5;
    int i = (int)floor(f);
    if (i >= 0 && src > 1) {
      if (i < src - 1) {
        ofs[d] = i;
        w1[d] = (uint16_t)cv_round_d((f - (double)i) * 256.0);
      } else {
        ofs[d] = src - 1; /* replicate last; second tap never read (weight 0) */
        w1[d] = 0;
      }
    } else {
      ofs[d] = 0; /* replicate first */
      w1[d] = 0;
    }
  }
}

orb_oracle_ctx* orb_oracle_create(int width, int height, const orb_oracle_opts* opts) {
  orb_oracle_ctx* c = (orb_oracle_ctx*)calloc(1, sizeof(*c));
  if (!c) return NULL;
  if (opts) c->o = *opts; else orb_oracle_default_opts(&c->o);
  if (c->o.nlevels < 1 || c->o.nlevels > NL || width < 8 || height < 8) {
    free(c);
    return NULL;
  }
  c->w0 = width;
  c->h0 = height;
  const float scale_factor_f = 1.2f;            /* cv::ORB::create(..., scaleFactor = 1.2f, ...) */
  const double scale_factor = (double)scale_factor_f; /* ORB_Impl keeps it as double */
  int nl = c->o.nlevels;
  /* ORB_Impl::detectAndCompute: getScale() and layer sizes */
  for (int l = 0; l < nl; l++) {
    float s = (float)pow(scale_factor, (double)l);
    float inv = 1.0f / s;
    c->scale[l] = s;
    if (c->o.level_size_mul_inv) {
      c->lw[l] = cv_round_f((float)width * inv);
      c->lh[l] = cv_round_f((float)height * inv);
    } else {                                     /* Size sz(cvRound(image.cols/scale), cvRound(image.rows/scale)) */
      c->lw[l] = cv_round_f((float)width / s);
      c->lh[l] = cv_round_f((float)height / s);
    }
  }
  /* computeKeyPoints: nfeaturesPerLevel */
  {
    float factor = (float)(1.0 / scale_factor);
    float nd = (float)c->o.nfeatures * (1 - factor) /
               (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
      c->quota[l] = cv_round_f(nd);
      sum += c->quota[l];
      nd *= factor;
    }
    c->quota[nl - 1] = c->o.nfeatures - sum > 0 ? c->o.nfeatures - sum : 0;
  }
  /* computeKeyPoints: umax */
  {
    int v, v0;
    int vmax = (int)floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v)
      c->umax[v] = cv_round_d(sqrt((double)HALF_PATCH * HALF_PATCH - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
      while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
      c->umax[v] = v0;
      ++v0;
    }
  }
  for (int l = 0; l < nl; l++) {
    size_t n = (size_t)c->lw[l] * c->lh[l];
    c->lvl[l] = (uint8_t*)malloc(n);
    c->blr[l] = (uint8_t*)malloc(n);
    if (l > 0) {
      /* level 1 is resized from the input image, level l>1 from level l-1 */
      int sw = c->lw[l - 1], sh = c->lh[l - 1];
      c->xofs[l] = (int*)malloc(sizeof(int) * c->lw[l]);
      c->yofs[l] = (int*)malloc(sizeof(int) * c->lh[l]);
      c->xw[l] = (uint16_t*)malloc(sizeof(uint16_t) * c->lw[l]);
      c->yw[l] = (uint16_t*)malloc(sizeof(uint16_t) * c->lh[l]);
      make_resize_table(sw, c->lw[l], c->xofs[l], c->xw[l]);
      make_resize_table(sh, c->lh[l], c->yofs[l], c->yw[l]);
    }
  }
  return c;
}

void orb_oracle_destroy(orb_oracle_ctx* c) {
  if (!c) return;
  for (int l = 0; l < NL; l++) {
    free(c->lvl[l]); free(c->blr[l]); free(c->xofs[l]); free(c->yofs[l]);
    free(c->xw[l]); free(c->yw[l]); free(c->cand[l]);
  }
  free(c->s1); free(c->kp); free(c->desc);
  free(c);
}

/* resize_bitExact<uchar, ufixedpoint16> (hlineResizeCn + vlineResize):
 * 16-bit horizontal sums w0*p0 + w1*p1, 32-bit vertical, (v + 32768) >> 16. */
static void resize_level(const orb_oracle_ctx* c, int l) {
  int sw = c->lw[l - 1], dw = c->lw[l], dh = c->lh[l];
  const uint8_t* src = c->lvl[l - 1];
  uint8_t* dst = c->lvl[l];
  uint16_t* h0 = (uint16_t*)malloc(sizeof(uint16_t) * dw);
  uint16_t* h1 = (uint16_t*)malloc(sizeof(uint16_t) * dw);
  for (int y = 0; y < dh; y++) {
    int sy = c->yofs[l][y];
    uint32_t wy1 = c->yw[l][y], wy0 = 256 - wy1;
    const uint8_t* r0 = src + (size_t)sy * sw;
    const uint8_t* r1 = wy1 ? r0 + sw : r0;
    for (int x = 0; x < dw; x++) {
      int sx = c->xofs[l][x];
      uint32_t wx1 = c->xw[l][x], wx0 = 256 - wx1;
      int sx1 = wx1 ? sx + 1 : sx;
      h0[x] = (uint16_t)(wx0 * r0[sx] + wx1 * r0[sx1]);
      h1[x] = (uint16_t)(wx0 * r1[sx] + wx1 * r1[sx1]);
    }
    for (int x = 0; x < dw; x++) {
      uint32_t v = (uint32_t)h0[x] * wy0 + (uint32_t)h1[x] * wy1;
      uint32_t r = (v + 32768u) >> 16;
      dst[(size_t)y * dw + x] = (uint8_t)(r > 255 ? 255 : r);
    }
  }
  free(h0);
  free(h1);
}

/* FAST-9/16 + cornerScore<16> + 3x3 strict NMS (OpenCV features2d fast.cpp
 * FAST_t<16>, fast_score.cpp cornerScore<16>), then
 * KeyPointsFilter::runByImageBorder(edgeThreshold) (keypoint.cpp). */
static const int ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_score_at(const uint8_t* p, int stride, int t) {
  int d[16];
  int v = p[0];
  for (int k = 0; k < 16; k++) d[k] = v - p[ring_dy[k] * stride + ring_dx[k]];
  int A = INT_MIN, B = INT_MIN;
  for (int s = 0; s < 16; s++) {
    int mn = INT_MAX, mx = INT_MIN;
    for (int i = 0; i < 9; i++) {
      int q = d[(s + i) & 15];
      if (q < mn) mn = q;
      if (q > mx) mx = q;
    }
    if (mn > A) A = mn;      /* darker arc:   all p < v - A' */
    if (-mx > B) B = -mx;    /* brighter arc: all p > v + B' */
  }
  int m = A > B ? A : B;
  if (m <= t) return 0;      /* not a corner (no arc of 9 beyond the threshold) */
  return m - 1;              /* cornerScore: max(t, A, B) - 1, m > t here */
}

static void push_cand(orb_oracle_ctx* c, int l, int x, int y, int s) {
  if (c->ncand[l] == c->cand_cap[l]) {
    c->cand_cap[l] = c->cand_cap[l] ? c->cand_cap[l] * 2 : 1024;
    c->cand[l] = (cand_t*)realloc(c->cand[l], sizeof(cand_t) * c->cand_cap[l]);
  }
  cand_t* q = &c->cand[l][c->ncand[l]++];
  q->x = x; q->y = y; q->score = s;
}

static void fast_level(orb_oracle_ctx* c, int l) {
  int w = c->lw[l], h = c->lh[l], t = c->o.fast_threshold, e = c->o.edge_threshold;
  const uint8_t* img = c->lvl[l];
  c->ncand[l] = 0;
  if (t < 0) t = 0;
  if (t > 255) t = 255;
  if (w < 7 || h < 7) return;
  uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);
  /* FAST_t: scores for rows 3..h-4, cols 3..w-4; everything else scores 0 */
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++)
      sc[(size_t)y * w + x] = (uint8_t)fast_score_at(img + (size_t)y * w + x, w, t);
  /* runByImageBorder: clear all if the image is too small */
  int small = (e > 0) && (h <= e * 2 || w <= e * 2);
  if (!small) {
    for (int y = 3; y < h - 3; y++)
      for (int x = 3; x < w - 3; x++) {
        int s = sc[(size_t)y * w + x];
        if (!s) continue;
        const uint8_t* q = sc + (size_t)y * w + x;
        if (s > q[-1] && s > q[1] && s > q[-w - 1] && s > q[-w] && s > q[-w + 1] &&
            s > q[w - 1] && s > q[w] && s > q[w + 1]) {
          /* RoiPredicate: keep iff e <= x < w - e and e <= y < h - e */
          if (e > 0 && !(x >= e && x < w - e && y >= e && y < h - e)) continue;
          push_cand(c, l, x, y, s);
        }
      }
  }
  free(sc);
}

/* KeyPointsFilter::retainBest threshold: the n-th largest response; callers
 * keep every element >= it (ties may exceed n), order preserved (canonical). */
static int cmp_float_desc(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x < y) - (x > y);
}
static float nth_largest(const float* v, int n, int nth) {
  float* t = (float*)malloc(sizeof(float) * n);
  memcpy(t, v, sizeof(float) * n);
  qsort(t, n, sizeof(float), cmp_float_desc);
  float r = t[nth - 1];
  free(t);
  return r;
}

static void ensure_kp(orb_oracle_kp** p, int* cap, int need) {
  if (need > *cap) {
    int nc = *cap ? *cap : 1024;
    while (nc < need) nc *= 2;
    *p = (orb_oracle_kp*)realloc(*p, sizeof(orb_oracle_kp) * nc);
    *cap = nc;
  }
}

/* HarrisResponses (OpenCV orb.cpp), blockSize 7, on the unblurred level. */
static float harris_at(const uint8_t* img, int step, int x0, int y0) {
  const int r = HARRIS_BLOCK / 2;
  float scale = 1.f / ((1 << 2) * HARRIS_BLOCK * 255.f);
  float scale_sq_sq = scale * scale * scale * scale;
  int a = 0, b = 0, c = 0;
  for (int i = 0; i < HARRIS_BLOCK; i++)
    for (int j = 0; j < HARRIS_BLOCK; j++) {
      const uint8_t* p = img + (size_t)(y0 - r + i) * step + (x0 - r + j);
      int Ix = (p[1] - p[-1]) * 2 + (p[-step + 1] - p[-step - 1]) + (p[step + 1] - p[step - 1]);
      int Iy = (p[step] - p[-step]) * 2 + (p[step - 1] - p[-step - 1]) + (p[step + 1] - p[-step + 1]);
      a += Ix * Ix;
      b += Iy * Iy;
      c += Ix * Iy;
    }
  return ((float)a * (float)b - (float)c * (float)c -
          HARRIS_K * ((float)a + (float)b) * ((float)a + (float)b)) * scale_sq_sq;
}

/* ICAngles (OpenCV orb.cpp): intensity centroid over the 31-px disc. */
static float ic_angle_at(const orb_oracle_ctx* c, const uint8_t* img, int step, int x0, int y0) {
  const uint8_t* center = img + (size_t)y0 * step + x0;
  int m_01 = 0, m_10 = 0;
  for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH; ++v) {
    int v_sum = 0;
    int d = c->umax[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return orb_oracle_fast_atan2((float)m_01, (float)m_10);
}

/* GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101) as reached
 * from ORB_Impl::detectAndCompute: the level is a sub-matrix with a
 * non-isolated border, so OpenCV 4.x takes sepFilter2D with the 8u->8u
 * symmetric-smooth integer engine: kernel round(256*g) = 18 34 49 55 49 34 18,
 * int32 row pass, int32 column pass, 16 fractional bits dropped with rounding.
 * Out-of-level taps are reflect-101 of the unblurred level (the pyramid's own
 * border holds exactly that). */
static const int blur_k257[7] = {18, 34, 49, 55, 49, 34, 18};
/* blur_kernel_sum256: the kernel of OpenCV's bit-exact fixed-point Gaussian (smooth.dispatch.cpp,
 * getGaussianKernelFixedPoint_ED, 8 fraction bits): the softdouble weights 256 * g = 17.9607 33.5552 48.8225 55.3231
 * rounded with error diffusion from the outside in (18, err -0.039; 34, err -0.484; 48, err +0.338) and the centre
 * tap taking what is left of 256: 18 34 48 56 48 34 18.  That path (fixedSmoothInvoker, ufixedpoint16 rows,
 * ufixedpoint32 columns) rounds half up, (sum + 32768) >> 16.  cv::ORB does NOT reach it in OpenCV 4.x (its level is a
 * sub-matrix with a non-isolated border); the switch exists so that a real OpenCV, the day one is
 * reachable, can be compared against both forms. */
static const int blur_k256[7] = {18, 34, 48, 56, 48, 34, 18};
static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    else i = 2 * (n - 1) - i;
  }
  return i;
}
static void blur_level(orb_oracle_ctx* c, int l) {
  int w = c->lw[l], h = c->lh[l];
  const uint8_t* src = c->lvl[l];
  uint8_t* dst = c->blr[l];
  const int* blur_k = c->o.blur_kernel_sum256 ? blur_k256 : blur_k257;
  const int tie_even = c->o.blur_tie_even && !c->o.blur_kernel_sum256;   /* the fixed-point path rounds half up */
  int32_t* rows = (int32_t*)malloc(sizeof(int32_t) * (size_t)w * h);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t s = 0;
      for (int k = -3; k <= 3; k++) s += blur_k[k + 3] * src[(size_t)y * w + reflect101(x + k, w)];
      rows[(size_t)y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t s = 0;
      for (int k = -3; k <= 3; k++) s += blur_k[k + 3] * rows[(size_t)reflect101(y + k, h) * w + x];
      int32_t r;
      if (tie_even) {
        /* OpenCV's SymmColumnVec_32s8u evaluates sum * 2^-16 in f32 (exact below
         * 256) and rounds to nearest even */
        r = s >> 16;
        int32_t frac = s & 0xFFFF;
        if (frac > 0x8000 || (frac == 0x8000 && (r & 1))) r++;
      } else {
        r = (s + 32768) >> 16; /* FixedPtCastEx<int, uchar>(16) */
      }
      dst[(size_t)y * w + x] = (uint8_t)(r > 255 ? 255 : r);
    }
  free(rows);
}

/* computeOrbDescriptors (OpenCV orb.cpp), WTA_K = 2, on the blurred level. */
static void describe(const orb_oracle_ctx* c, const orb_oracle_kp* kp, uint8_t* desc) {
  int l = kp->octave;
  int step = c->lw[l];
  float scale = 1.f / c->scale[l];
  float angle = kp->angle;
  angle *= (float)(3.14159265358979323846 / 180.f);
  float a, b;
  orb_oracle_sincosf(angle, &b, &a); /* a = cos, b = sin */
  int cx = cv_round_f(kp->x * scale), cy = cv_round_f(kp->y * scale);
  const uint8_t* center = c->blr[l] + (size_t)cy * step + cx;
  const signed char* pat = orb_bit_pattern_31;
  for (int i = 0; i < 32; i++) {
    int val = 0;
    for (int j = 0; j < 8; j++, pat += 4) {
      float x0 = (float)pat[0] * a - (float)pat[1] * b;
      float y0 = (float)pat[0] * b + (float)pat[1] * a;
      float x1 = (float)pat[2] * a - (float)pat[3] * b;
      float y1 = (float)pat[2] * b + (float)pat[3] * a;
      int t0 = center[cv_round_f(y0) * step + cv_round_f(x0)];
      int t1 = center[cv_round_f(y1) * step + cv_round_f(x1)];
      val |= (t0 < t1) << j;
    }
    desc[i] = (uint8_t)val;
  }
}

/* ORB_Impl::detectAndCompute + computeKeyPoints (OpenCV orb.cpp), as called at
 * /root/reference/src/featurematcher.cpp:15-17 (all-255 mask == no mask). */
int orb_oracle_extract(orb_oracle_ctx* c, const uint8_t* img, ptrdiff_t stride) {
  if (!c || !img) return -1;
  int nl = c->o.nlevels;
  /* pyramid: level 0 = image; level 1 from the image; level l from level l-1 */
  for (int y = 0; y < c->h0; y++) memcpy(c->lvl[0] + (size_t)y * c->w0, img + (size_t)y * stride, c->w0);
  for (int l = 1; l < nl; l++) resize_level(c, l);

  /* computeKeyPoints, first loop: FAST, border, retainBest(2N) per level */
  c->ns1 = 0;
  for (int l = 0; l < nl; l++) {
    fast_level(c, l);
    int n = c->ncand[l];
    int keep_n = 2 * c->quota[l];
    float thr = -FLT_MAX;
    if (n > keep_n) {
      if (keep_n == 0) { c->s1_count[l] = 0; continue; }
      float* r = (float*)malloc(sizeof(float) * n);
      for (int i = 0; i < n; i++) r[i] = (float)c->cand[l][i].score;
      thr = nth_largest(r, n, keep_n);
      free(r);
    }
    int cnt = 0;
    for (int i = 0; i < n; i++) {
      const cand_t* q = &c->cand[l][i];
      if ((float)q->score >= thr) {
        ensure_kp(&c->s1, &c->s1_cap, c->ns1 + 1);
        orb_oracle_kp* k = &c->s1[c->ns1++];
        k->x = (float)q->x; k->y = (float)q->y; /* level coords until the end */
        k->lx = q->x; k->ly = q->y;
        k->octave = l; k->fast_score = q->score;
        k->response = harris_at(c->lvl[l], c->lw[l], q->x, q->y);
        k->angle = -1.f;
        cnt++;
      }
    }
    c->s1_count[l] = cnt;
  }
  /* Harris cull: retainBest(N_l) per level */
  c->nkp = 0;
  int off = 0;
  for (int l = 0; l < nl; l++) {
    int n = c->s1_count[l], keep_n = c->quota[l];
    float thr = -FLT_MAX;
    int drop_all = 0;
    if (n > keep_n) {
      if (keep_n == 0) drop_all = 1;
      else {
        float* r = (float*)malloc(sizeof(float) * n);
        for (int i = 0; i < n; i++) r[i] = c->s1[off + i].response;
        thr = nth_largest(r, n, keep_n);
        free(r);
      }
    }
    for (int i = 0; i < n && !drop_all; i++)
      if (c->s1[off + i].response >= thr) {
        ensure_kp(&c->kp, &c->kp_cap, c->nkp + 1);
        c->kp[c->nkp++] = c->s1[off + i];
      }
    off += n;
  }
  /* ICAngles, then pt *= scale */
  for (int i = 0; i < c->nkp; i++) {
    orb_oracle_kp* k = &c->kp[i];
    int l = k->octave;
    k->angle = ic_angle_at(c, c->lvl[l], c->lw[l], cv_round_f(k->x), cv_round_f(k->y));
    k->x *= c->scale[l];
    k->y *= c->scale[l];
  }
  /* blur + descriptors */
  for (int l = 0; l < nl; l++) blur_level(c, l);
  if (c->nkp * 32 > c->desc_cap) {
    c->desc_cap = c->nkp * 32 + 4096;
    c->desc = (uint8_t*)realloc(c->desc, c->desc_cap);
  }
  for (int i = 0; i < c->nkp; i++) describe(c, &c->kp[i], c->desc + (size_t)i * 32);
  return c->nkp;
}

int orb_oracle_num_keypoints(const orb_oracle_ctx* c) { return c->nkp; }
const orb_oracle_kp* orb_oracle_keypoints(const orb_oracle_ctx* c) { return c->kp; }
const uint8_t* orb_oracle_descriptors(const orb_oracle_ctx* c) { return c->desc; }
int orb_oracle_level_size(const orb_oracle_ctx* c, int l, int* w, int* h) {
  if (l < 0 || l >= c->o.nlevels) return -1;
  *w = c->lw[l]; *h = c->lh[l];
  return 0;
}
float orb_oracle_level_scale(const orb_oracle_ctx* c, int l) { return c->scale[l]; }
int orb_oracle_level_quota(const orb_oracle_ctx* c, int l) { return c->quota[l]; }
const uint8_t* orb_oracle_level_pixels(const orb_oracle_ctx* c, int l) { return c->lvl[l]; }
const uint8_t* orb_oracle_level_blurred(const orb_oracle_ctx* c, int l) { return c->blr[l]; }
int orb_oracle_fast_candidates(const orb_oracle_ctx* c, int l, const int32_t** xys) {
  *xys = (const int32_t*)c->cand[l];
  return c->ncand[l];
}
/* FAST_t<16> + cornerScore<16> of every pixel of a level BEFORE the 3x3 NMS and the border reject (0 = no corner):
 * the detector's segment decision on its own, for anchors that hold a FAST-9 of their own (tests/test_fast_anchor.py) */
int orb_oracle_fast_score_map(const orb_oracle_ctx* c, int l, uint8_t* out) {
  if (!c || !out || l < 0 || l >= c->o.nlevels) return -1;
  int w = c->lw[l], h = c->lh[l], t = c->o.fast_threshold;
  memset(out, 0, (size_t)w * h);
  if (w < 7 || h < 7) return 0;
  for (int y = 3; y < h - 3; y++)
    for (int x = 3; x < w - 3; x++)
      out[(size_t)y * w + x] = (uint8_t)fast_score_at(c->lvl[l] + (size_t)y * w + x, w, t);
  return 0;
}
int orb_oracle_stage1_keypoints(const orb_oracle_ctx* c, const orb_oracle_kp** kps) {
  *kps = c->s1;
  return c->ns1;
}

/* batchDistance(..., NORM_HAMMING, K = 2) insertion (OpenCV core
 * batch_distance.cpp): strict '<', so ties keep the lower train index. */
static int hamming256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 32; i++) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}
void orb_oracle_knn2(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int32_t* out) {
  for (int q = 0; q < n1; q++) {
    int bi0 = -1, bi1 = -1, bd0 = INT_MAX, bd1 = INT_MAX;
    for (int t = 0; t < n2; t++) {
      int d = hamming256(d1 + (size_t)q * 32, d2 + (size_t)t * 32);
      if (d < bd1) {
        if (d < bd0) { bd1 = bd0; bi1 = bi0; bd0 = d; bi0 = t; }
        else { bd1 = d; bi1 = t; }
      }
    }
    out[q * 4 + 0] = bi0; out[q * 4 + 1] = bd0; out[q * 4 + 2] = bi1; out[q * 4 + 3] = bd1;
  }
}

/* /root/reference/src/featurematcher.cpp:23-42 */
int orb_oracle_knn_match(const uint8_t* d1, int n1, const orb_oracle_kp* k1,
                         const uint8_t* d2, int n2, const orb_oracle_kp* k2,
                         float ratio, int32_t* out, int cap) {
  if (n1 <= 0 || n2 < 2) return 0; /* empty guard (:23); n2 == 1 is UB in the reference */
  int32_t* nn = (int32_t*)malloc(sizeof(int32_t) * 4 * n1);
  orb_oracle_knn2(d1, n1, d2, n2, nn);
  int m = 0;
  for (int q = 0; q < n1; q++) {
    float dist0 = (float)nn[q * 4 + 1], dist1 = (float)nn[q * 4 + 3];
    if (dist0 < ratio * dist1) {
      if (m < cap) {
        int t = nn[q * 4 + 0];
        out[m * 4 + 0] = (int)k1[q].x;
        out[m * 4 + 1] = (int)k1[q].y;
        out[m * 4 + 2] = (int)k2[t].x;
        out[m * 4 + 3] = (int)k2[t].y;
      }
      m++;
    }
  }
  free(nn);
  return m;
}

int orb_oracle_match_frames(orb_oracle_ctx* ca, orb_oracle_ctx* cb,
                            const uint8_t* a, ptrdiff_t stride_a,
                            const uint8_t* b, ptrdiff_t stride_b,
                            float ratio, int32_t* out, int cap) {
  int n1 = orb_oracle_extract(ca, a, stride_a);
  int n2 = orb_oracle_extract(cb, b, stride_b);
  if (n1 < 0 || n2 < 0) return -1;
  return orb_oracle_knn_match(ca->desc, n1, ca->kp, cb->desc, n2, cb->kp, ratio, out, cap);
}
