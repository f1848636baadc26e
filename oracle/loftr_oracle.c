/*
 * oracle/loftr_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * f32 CPU restatement of what ::DNNFeatureMatcher::MatchFrames computes
 * (/root/reference/src/dnnfeaturematcher.cpp:44-102): ConvertImageToFloat (:5-9),
 * the graph of /root/reference/model/LoFTR_teacher.onnx (run by Ort::Session::Run, :62-64)
 * restated layer by layer, then the strict
 * '> threshold', row-major findNonZero and the 16-px cell decode (:75-99).
 *
 * ONNXRuntime is an un-vendored, unpinned dependency that is absent here; the graph
 * file itself is the specification.  This restatement is PINNED: tests/test_loftr_oracle.py
 * checks it against tests/golden/loftr_kat.npz, which oracle/onnx_oracle.py produced by
 * interpreting the reference's .onnx node by node (tools/make_loftr_fixtures.py), and
 * against the known-answer numbers of SURVEY.md section 8c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Weights come from mono_slam_framework_amd/weights/loftr_teacher.bin (same blob the HIP
 * path loads; parsed independently here).
 */
#include "loftr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NTOK 1200
#define DM 32
#define FH 30
#define FW 40

typedef struct {
  char name[33];
  unsigned ndim, dims[4], off, count;
} rec_t;

struct loftr_oracle {
  float* data;
  rec_t* recs;
  unsigned nrec;
};

static const float* get(const loftr_oracle* o, const char* name, unsigned expect) {
  for (unsigned i = 0; i < o->nrec; i++)
    if (strcmp(o->recs[i].name, name) == 0) {
      if (expect && o->recs[i].count != expect) return NULL;
      return o->data + o->recs[i].off;
    }
  return NULL;
}

/* number of OpenMP threads the convolutions / similarity loops use (for an honest cpu_baseline `cores`) */
int loftr_oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

loftr_oracle* loftr_oracle_create(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  char magic[8];
  unsigned n = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "MSFLTR01", 8) != 0 || fread(&n, 4, 1, f) != 1 || n > 4096) {
    fclose(f);
    return NULL;
  }
  loftr_oracle* o = (loftr_oracle*)calloc(1, sizeof(*o));
  o->nrec = n;
  o->recs = (rec_t*)calloc(n, sizeof(rec_t));
  size_t total = 0;
  for (unsigned i = 0; i < n; i++) {
    unsigned char raw[60];
    if (fread(raw, 1, 60, f) != 60) { fclose(f); loftr_oracle_destroy(o); return NULL; }
    memcpy(o->recs[i].name, raw, 32);
    o->recs[i].name[32] = 0;
    memcpy(&o->recs[i].ndim, raw + 32, 4);
    memcpy(o->recs[i].dims, raw + 36, 16);
    memcpy(&o->recs[i].off, raw + 52, 4);
    memcpy(&o->recs[i].count, raw + 56, 4);
    if ((size_t)o->recs[i].off + o->recs[i].count > total) total = (size_t)o->recs[i].off + o->recs[i].count;
  }
  o->data = (float*)malloc(total * sizeof(float));
  if (fread(o->data, sizeof(float), total, f) != total) { fclose(f); loftr_oracle_destroy(o); return NULL; }
  fclose(f);
  return o;
}

void loftr_oracle_destroy(loftr_oracle* o) {
  if (!o) return;
  free(o->data);
  free(o->recs);
  free(o);
}

/* ONNX Conv, NCHW, batch 2, square kernel, symmetric padding; optional bias, residual add and ReLU */
static void conv2d(const float* in, int cin, int hin, int win, const float* w, const float* bias, int cout, int k,
                   int stride, int pad, const float* residual, int relu, float* out) {
  int hout = (hin + 2 * pad - k) / stride + 1, wout = (win + 2 * pad - k) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < 2; n++)
    for (int co = 0; co < cout; co++) {
      const float* inn = in + (size_t)n * cin * hin * win;
      float* o = out + ((size_t)n * cout + co) * hout * wout;
      for (int y = 0; y < hout; y++)
        for (int x = 0; x < wout; x++) {
          float acc = bias ? bias[co] : 0.f;
          for (int ci = 0; ci < cin; ci++) {
            const float* wp = w + ((size_t)co * cin + ci) * k * k;
            const float* ip = inn + (size_t)ci * hin * win;
            for (int ky = 0; ky < k; ky++) {
              int iy = y * stride - pad + ky;
              if (iy < 0 || iy >= hin) continue;
              for (int kx = 0; kx < k; kx++) {
                int ix = x * stride - pad + kx;
                if (ix < 0 || ix >= win) continue;
                acc += ip[(size_t)iy * win + ix] * wp[ky * k + kx];
              }
            }
          }
          if (residual) acc += residual[((size_t)n * cout + co) * hout * wout + (size_t)y * wout + x];
          if (relu && acc < 0.f) acc = 0.f;
          o[(size_t)y * wout + x] = acc;
        }
    }
}

static void matmul(const float* a, const float* b, int m, int k, int n, float* c) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) {
      float s = 0.f;
      for (int p = 0; p < k; p++) s += a[(size_t)i * k + p] * b[(size_t)p * n + j];
      c[(size_t)i * n + j] = s;
    }
}

/* the graph's long-hand LayerNorm: (t - mean) / sqrt(mean((t - mean)^2) + 1e-7) * w + b */
static void layer_norm(float* t, int rows, int d, const float* w, const float* b) {
  const float eps = 1.0000000116860974e-07f;
  for (int r = 0; r < rows; r++) {
    float* p = t + (size_t)r * d;
    float mean = 0.f;
    for (int i = 0; i < d; i++) mean += p[i];
    mean /= (float)d;
    float var = 0.f;
    for (int i = 0; i < d; i++) { float c = p[i] - mean; var += c * c; }
    var /= (float)d;
    float inv = sqrtf(var + eps);
    for (int i = 0; i < d; i++) p[i] = (p[i] - mean) / inv * w[i] + b[i];
  }
}

static float elu1(float x) { return (x > 0.f ? x : expf(x) - 1.f) + 1.f; }

/* one LoFTR encoder layer with linear attention (nodes 70-144 for block 0) */
static int block(const loftr_oracle* o, int b, const float* x, const float* s, float* out) {
  char nm[64];
  const float *wq, *wk, *wv, *wm, *w0, *w1, *n1w, *n1b, *n2w, *n2b;
#define G(dst, fmt, idx, cnt) snprintf(nm, sizeof nm, fmt, idx); dst = get(o, nm, cnt); if (!dst) return -1;
  G(wq, "blk%d.wq", b, 1024) G(wk, "blk%d.wk", b, 1024) G(wv, "blk%d.wv", b, 1024) G(wm, "blk%d.wmerge", b, 1024)
  G(w0, "blk%d.wmlp0", b, 4096) G(w1, "blk%d.wmlp1", b, 2048)
  G(n1w, "ln%d.n1w", b / 2, 32) G(n1b, "ln%d.n1b", b / 2, 32) G(n2w, "ln%d.n2w", b / 2, 32) G(n2b, "ln%d.n2b", b / 2, 32)
#undef G
  float* q = (float*)malloc(sizeof(float) * NTOK * DM);
  float* k = (float*)malloc(sizeof(float) * NTOK * DM);
  float* v = (float*)malloc(sizeof(float) * NTOK * DM);
  float* msg = (float*)malloc(sizeof(float) * NTOK * DM);
  float* cat = (float*)malloc(sizeof(float) * NTOK * 2 * DM);
  float* hid = (float*)malloc(sizeof(float) * NTOK * 2 * DM);
  matmul(x, wq, NTOK, DM, DM, q);
  matmul(s, wk, NTOK, DM, DM, k);
  matmul(s, wv, NTOK, DM, DM, v);
  float KV[DM][DM], Ksum[DM];
  memset(KV, 0, sizeof KV);
  memset(Ksum, 0, sizeof Ksum);
  for (int i = 0; i < NTOK * DM; i++) { q[i] = elu1(q[i]); k[i] = elu1(k[i]); v[i] = v[i] / 1200.0f; }
  for (int t = 0; t < NTOK; t++)
    for (int d = 0; d < DM; d++) {
      float kd = k[t * DM + d];
      Ksum[d] += kd;
      for (int e = 0; e < DM; e++) KV[d][e] += kd * v[t * DM + e];
    }
  for (int t = 0; t < NTOK; t++) {
    float z = 0.f;
    for (int d = 0; d < DM; d++) z += q[t * DM + d] * Ksum[d];
    z = 1.0f / (z + 9.999999974752427e-07f);
    for (int e = 0; e < DM; e++) {
      float a = 0.f;
      for (int d = 0; d < DM; d++) a += q[t * DM + d] * KV[d][e];
      msg[t * DM + e] = a * z * 1200.0f;
    }
  }
  float* merged = q; /* reuse */
  matmul(msg, wm, NTOK, DM, DM, merged);
  layer_norm(merged, NTOK, DM, n1w, n1b);
  for (int t = 0; t < NTOK; t++) {
    memcpy(cat + (size_t)t * 64, x + (size_t)t * DM, sizeof(float) * DM);
    memcpy(cat + (size_t)t * 64 + DM, merged + (size_t)t * DM, sizeof(float) * DM);
  }
  matmul(cat, w0, NTOK, 64, 64, hid);
  for (int i = 0; i < NTOK * 64; i++) if (hid[i] < 0.f) hid[i] = 0.f;
  matmul(hid, w1, NTOK, 64, DM, msg);
  layer_norm(msg, NTOK, DM, n2w, n2b);
  for (int i = 0; i < NTOK * DM; i++) out[i] = x[i] + msg[i];
  free(q); free(k); free(v); free(msg); free(cat); free(hid);
  return 0;
}

int loftr_oracle_run(loftr_oracle* o, const uint8_t* im0, ptrdiff_t s0, const uint8_t* im1, ptrdiff_t s1,
                     float* conf, float* sim_out, float* feat0_out, float* feat1_out, float* tok_out) {
  const int H = 480, W = 640;
  /* ConvertImageToFloat: u8 * (float)(1/255.) */
  float* x = (float*)malloc(sizeof(float) * 2 * H * W);
  const float k255 = (float)(1.0 / 255.0);
  for (int y = 0; y < H; y++)
    for (int xx = 0; xx < W; xx++) {
      x[(size_t)y * W + xx] = (float)im0[(size_t)y * s0 + xx] * k255;
      x[(size_t)H * W + (size_t)y * W + xx] = (float)im1[(size_t)y * s1 + xx] * k255;
    }
  size_t big = (size_t)2 * 8 * 240 * 320;
  float* a = (float*)malloc(sizeof(float) * big);
  float* b = (float*)malloc(sizeof(float) * big);
  float* c = (float*)malloc(sizeof(float) * big);
  float* d = (float*)malloc(sizeof(float) * big);
  char nw[32], nb[32];
#define CW(i) (snprintf(nw, sizeof nw, "conv%02d.w", i), get(o, nw, 0))
#define CB(i) (snprintf(nb, sizeof nb, "conv%02d.b", i), get(o, nb, 0))
  /* stem + layer1 @240x320, 8 ch */
  conv2d(x, 1, 480, 640, CW(0), CB(0), 8, 7, 2, 3, NULL, 1, a);
  conv2d(a, 8, 240, 320, CW(1), CB(1), 8, 3, 1, 1, NULL, 1, b);
  conv2d(b, 8, 240, 320, CW(2), CB(2), 8, 3, 1, 1, a, 1, c);
  conv2d(c, 8, 240, 320, CW(3), CB(3), 8, 3, 1, 1, NULL, 1, b);
  conv2d(b, 8, 240, 320, CW(4), CB(4), 8, 3, 1, 1, c, 1, a);           /* a = 196 */
  /* layer2 @120x160, 16 ch */
  conv2d(a, 8, 240, 320, CW(5), CB(5), 16, 3, 2, 1, NULL, 1, b);
  conv2d(a, 8, 240, 320, CW(7), CB(7), 16, 1, 2, 0, NULL, 0, d);       /* shortcut */
  conv2d(b, 16, 120, 160, CW(6), CB(6), 16, 3, 1, 1, d, 1, c);         /* c = 205 */
  conv2d(c, 16, 120, 160, CW(8), CB(8), 16, 3, 1, 1, NULL, 1, b);
  conv2d(b, 16, 120, 160, CW(9), CB(9), 16, 3, 1, 1, c, 1, a);         /* a = 212 */
  /* layer3 @60x80, 32 ch */
  conv2d(a, 16, 120, 160, CW(10), CB(10), 32, 3, 2, 1, NULL, 1, b);
  conv2d(a, 16, 120, 160, CW(12), CB(12), 32, 1, 2, 0, NULL, 0, d);
  conv2d(b, 32, 60, 80, CW(11), CB(11), 32, 3, 1, 1, d, 1, c);         /* c = 221 */
  conv2d(c, 32, 60, 80, CW(13), CB(13), 32, 3, 1, 1, NULL, 1, b);
  conv2d(b, 32, 60, 80, CW(14), CB(14), 32, 3, 1, 1, c, 1, a);         /* a = 228 */
  /* layer4 @30x40, 32 ch */
  conv2d(a, 32, 60, 80, CW(15), CB(15), 32, 3, 2, 1, NULL, 1, b);
  conv2d(a, 32, 60, 80, CW(17), CB(17), 32, 1, 2, 0, NULL, 0, d);
  conv2d(b, 32, 30, 40, CW(16), CB(16), 32, 3, 1, 1, d, 1, c);         /* c = 237 */
  conv2d(c, 32, 30, 40, CW(18), CB(18), 32, 3, 1, 1, NULL, 1, b);
  conv2d(b, 32, 30, 40, CW(19), CB(19), 32, 3, 1, 1, c, 1, a);         /* a = 244 */
  conv2d(a, 32, 30, 40, get(o, "outconv.w", 1024), NULL, 32, 1, 1, 0, NULL, 0, b); /* b = 245 */
#undef CW
#undef CB
  /* + positional encoding, n c h w -> n (h w) c */
  const float* pe = get(o, "pe", 32 * FH * FW);
  float* f0 = (float*)malloc(sizeof(float) * NTOK * DM);
  float* f1 = (float*)malloc(sizeof(float) * NTOK * DM);
  float* t0 = (float*)malloc(sizeof(float) * NTOK * DM);
  float* t1 = (float*)malloc(sizeof(float) * NTOK * DM);
  if (!pe) return -1;
  for (int ch = 0; ch < DM; ch++)
    for (int t = 0; t < NTOK; t++) {
      f0[t * DM + ch] = b[(size_t)ch * NTOK + t] + pe[(size_t)ch * NTOK + t];
      f1[t * DM + ch] = b[(size_t)(DM + ch) * NTOK + t] + pe[(size_t)ch * NTOK + t];
    }
  if (tok_out) { memcpy(tok_out, f0, sizeof(float) * NTOK * DM); memcpy(tok_out + NTOK * DM, f1, sizeof(float) * NTOK * DM); }
  /* 4 layers: self, cross, self, cross; the cross layer of feat1 sees the already updated feat0 */
  int rc = 0;
  rc |= block(o, 0, f0, f0, t0);
  rc |= block(o, 1, f1, f1, t1);
  rc |= block(o, 2, t0, t1, f0);
  rc |= block(o, 3, t1, f0, f1);
  rc |= block(o, 4, f0, f0, t0);
  rc |= block(o, 5, f1, f1, t1);
  rc |= block(o, 6, t0, t1, f0);
  rc |= block(o, 7, t1, f0, f1);
  if (rc) return -1;
  if (feat0_out) memcpy(feat0_out, f0, sizeof(float) * NTOK * DM);
  if (feat1_out) memcpy(feat1_out, f1, sizeof(float) * NTOK * DM);
  /* matching head: sim = (f0/sqrt(32)) (f1/sqrt(32))^T; conf = softmax_j(sim/0.1) * softmax_i(sim/0.1) */
  const float sq = 5.656854f;
  for (int i = 0; i < NTOK * DM; i++) { f0[i] = f0[i] / sq; f1[i] = f1[i] / sq; }
  float* sim = (float*)malloc(sizeof(float) * NTOK * NTOK);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < NTOK; i++)
    for (int j = 0; j < NTOK; j++) {
      float s = 0.f;
      for (int p = 0; p < DM; p++) s += f0[i * DM + p] * f1[j * DM + p];
      sim[(size_t)i * NTOK + j] = s;
    }
  if (sim_out) memcpy(sim_out, sim, sizeof(float) * NTOK * NTOK);
  float* rmax = (float*)malloc(sizeof(float) * NTOK * 4);
  float *rsum = rmax + NTOK, *cmax = rmax + 2 * NTOK, *csum = rmax + 3 * NTOK;
  for (int i = 0; i < NTOK; i++) { rmax[i] = -INFINITY; cmax[i] = -INFINITY; rsum[i] = 0.f; csum[i] = 0.f; }
  for (int i = 0; i < NTOK; i++)
    for (int j = 0; j < NTOK; j++) {
      float v = sim[(size_t)i * NTOK + j] / 0.1f;
      sim[(size_t)i * NTOK + j] = v;
      if (v > rmax[i]) rmax[i] = v;
      if (v > cmax[j]) cmax[j] = v;
    }
  for (int i = 0; i < NTOK; i++)
    for (int j = 0; j < NTOK; j++) {
      float v = sim[(size_t)i * NTOK + j];
      rsum[i] += expf(v - rmax[i]);
      csum[j] += expf(v - cmax[j]);
    }
  for (int i = 0; i < NTOK; i++)
    for (int j = 0; j < NTOK; j++) {
      float v = sim[(size_t)i * NTOK + j];
      conf[(size_t)i * NTOK + j] = (expf(v - cmax[j]) / csum[j]) * (expf(v - rmax[i]) / rsum[i]);
    }
  free(rmax); free(sim); free(f0); free(f1); free(t0); free(t1);
  free(a); free(b); free(c); free(d); free(x);
  return 0;
}

/* dnnfeaturematcher.cpp:75-99 */
int loftr_oracle_decode(const float* conf, float threshold, int32_t* out, int cap) {
  const int model_width = 640 / 16, res = 16;
  int n = 0;
  for (int i = 0; i < NTOK; i++)
    for (int j = 0; j < NTOK; j++)
      if (conf[(size_t)i * NTOK + j] > threshold) {
        if (n < cap) {
          out[n * 4 + 0] = (i % model_width) * res;
          out[n * 4 + 1] = (i / model_width) * res;
          out[n * 4 + 2] = (j % model_width) * res;
          out[n * 4 + 3] = (j / model_width) * res;
        }
        n++;
      }
  return n;
}

int loftr_oracle_match(loftr_oracle* o, const uint8_t* im0, ptrdiff_t s0, const uint8_t* im1, ptrdiff_t s1,
                       float threshold, int32_t* out, int cap) {
  float* conf = (float*)malloc(sizeof(float) * NTOK * NTOK);
  int rc = loftr_oracle_run(o, im0, s0, im1, s1, conf, NULL, NULL, NULL, NULL);
  int n = rc ? -1 : loftr_oracle_decode(conf, threshold, out, cap);
  free(conf);
  return n;
}
