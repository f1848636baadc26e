// LoFTR_teacher on gfx950: the graph of model/LoFTR_teacher.onnx restated as HIP kernels.
//
// Reference entry point: ::DNNFeatureMatcher::MatchFrames (src/dnnfeaturematcher.cpp:44-102):
//   ConvertImageToFloat (:5-9)  -> fused into the stem convolution's tile load
//   Ort::Session::Run (:62-64)  -> k_conv (21 convolutions as implicit GEMM on v_mfma_f32_16x16x4_f32, exact f32),
//                                  k_tokens (PE add + layout), k_attn_kv / k_attn_update (8 linear-attention
//                                  encoder blocks), k_sim_stats / k_conf_mask (similarity + dual softmax)
//   '> threshold' + findNonZero + decode (:75-99) -> k_conf_mask (bit mask, conf never written to HBM) + k_decode
//
// Activations are NCHW f32 (the graph's own layout) so MFMA results store as float4 runs along W.
#include "loftr_pipeline.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <vector>

namespace msf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NTOK = 1200, DM = 32, FH = 30, FW = 40;
constexpr int MASK_WORDS = 38;  // ceil(1200 / 32)

// ------------------------------------------------------------------ convolution (implicit GEMM, f32 MFMA)
// D[16 px][16 cout] += A[16 px][4 k] * B[4 k][16 cout]; k enumerates (ky, kx, cin) with cin fastest.
// Block = 4 waves; wave w computes output row oy0 + w, OTW pixels wide, all output channels.
template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN>
struct ConvCfg {
  static constexpr int OTH = 4;
  static constexpr int MT = OTW / 16;
  static constexpr int NT = (COUT + 15) / 16;
  static constexpr int NPAD = NT * 16;
  static constexpr int PAD = KS / 2;
  static constexpr int IN_H = (OTH - 1) * S + KS;
  static constexpr int IN_W = (OTW - 1) * S + KS;
  static constexpr int PITCH = IN_W + 1;
  // plane stride: == 16 (mod 32) for stride 1, odd for stride 2, so the 16 px x 2 k lanes of a
  // ds_read_b32 group hit 32 distinct banks
  static constexpr int RAW = IN_H * PITCH;
  static constexpr int PLANE = S == 1 ? ((RAW + 15) / 32) * 32 + 16 : (RAW | 1);
  static constexpr int KTOT = KS * KS * CIN;
  static constexpr int KSTEPS = (KTOT + 3) / 4;
};

template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN>
__global__ __launch_bounds__(256) void k_conv(const void* __restrict__ in_, long long in_img_stride, int in_row_stride,
                                              const float* __restrict__ wB, const float* __restrict__ bias,
                                              const float* __restrict__ res, float* __restrict__ out, int Hin, int Win,
                                              int Hout, int Wout) {
  using C = ConvCfg<CIN, COUT, KS, S, OTW, RELU, RES, U8IN>;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int img = blockIdx.z;
  const int ox0 = blockIdx.x * OTW, oy0 = blockIdx.y * C::OTH;
  const int ix0 = ox0 * S - C::PAD, iy0 = oy0 * S - C::PAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // stage the input tile (zero padding outside the image)
  for (int idx = tid; idx < CIN * C::IN_H * C::IN_W; idx += 256) {
    const int c = idx / (C::IN_H * C::IN_W);
    const int rem = idx - c * (C::IN_H * C::IN_W);
    const int r = rem / C::IN_W, x = rem - r * C::IN_W;
    const int gy = iy0 + r, gx = ix0 + x;
    float v = 0.f;
    if (gy >= 0 && gy < Hin && gx >= 0 && gx < Win) {
      if (U8IN) {
        const uint8_t* p = static_cast<const uint8_t*>(in_) + (long long)img * in_img_stride;
        v = (float)p[(long long)gy * in_row_stride + gx] * (float)(1.0 / 255.0);  // ConvertImageToFloat
      } else {
        const float* p = static_cast<const float*>(in_) + (long long)img * in_img_stride;
        v = p[((long long)c * Hin + gy) * Win + gx];
      }
    }
    tile[c * C::PLANE + r * C::PITCH + x] = v;
  }
  __syncthreads();

  const int i = lane & 15, kq = lane >> 4;
  f32x4 acc[C::MT][C::NT];
#pragma unroll
  for (int m = 0; m < C::MT; m++)
#pragma unroll
    for (int n = 0; n < C::NT; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int step = 0; step < C::KSTEPS; step++) {
    int a_off;
    if (CIN % 4 == 0) {
      const int kk = (step * 4) / CIN;           // (ky, kx) shared by the 4 k of this step
      const int c = (step * 4) % CIN + kq;
      const int ky = kk / KS, kx = kk - ky * KS;
      a_off = c * C::PLANE + (wave * S + ky) * C::PITCH + kx;
    } else {                                      // stem: CIN = 1, k = ky * KS + kx, padded with zero weights
      int k = step * 4 + kq;
      k = k < C::KTOT ? k : C::KTOT - 1;
      const int ky = k / KS, kx = k - ky * KS;
      a_off = (wave * S + ky) * C::PITCH + kx;
    }
    float a[C::MT], b[C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; m++) a[m] = tile[a_off + (m * 16 + i) * S];
#pragma unroll
    for (int n = 0; n < C::NT; n++) b[n] = wB[(step * 4 + kq) * C::NPAD + n * 16 + i];
#pragma unroll
    for (int m = 0; m < C::MT; m++)
#pragma unroll
      for (int n = 0; n < C::NT; n++) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
  }

  // epilogue: D[row = 4*(lane>>4) + r][col = lane & 15] -> out[img][cout][oy][4 consecutive px]
  const int oy = oy0 + wave;
  if (oy >= Hout) return;
#pragma unroll
  for (int n = 0; n < C::NT; n++) {
    const int co = n * 16 + i;
    if (co >= COUT) continue;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < C::MT; m++) {
      const int px = ox0 + m * 16 + kq * 4;
      if (px >= Wout) continue;
      const long long o = (((long long)img * COUT + co) * Hout + oy) * Wout + px;
      f32x4 v = acc[m][n];
      v += f32x4{bv, bv, bv, bv};
      if (RES) v += *reinterpret_cast<const f32x4*>(res + o);
      if (RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      *reinterpret_cast<f32x4*>(out + o) = v;
    }
  }
}

// ------------------------------------------------------------------ tokens: + positional encoding, n c h w -> n (h w) c
__global__ __launch_bounds__(256) void k_tokens(const float* __restrict__ bb, const float* __restrict__ pe,
                                                float* __restrict__ tok, int n_img) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_img * NTOK * DM) return;
  const int img = idx / (NTOK * DM), rem = idx % (NTOK * DM);
  const int t = rem / DM, c = rem % DM;
  tok[idx] = bb[((long long)img * DM + c) * NTOK + t] + pe[c * NTOK + t];
}

// ------------------------------------------------------------------ linear-attention encoder block
struct BlockW {
  const float *wq, *wk, *wv, *wm, *w0, *w1, *n1w, *n1b, *n2w, *n2b;
};

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expf(x) - 1.f) + 1.f; }

// phase A: K = elu(s Wk) + 1, V = (s Wv) / 1200, KV = sum_t K_t^T V_t (32x32), Ksum = sum_t K_t.  One workgroup per
// source sequence.
__global__ __launch_bounds__(256) void k_attn_kv(const float* __restrict__ src, long long seq_stride, BlockW w,
                                                 float* __restrict__ kv /*[n][1056]*/) {
  __shared__ float sWk[DM * DM], sWv[DM * DM];
  __shared__ float sK[64 * (DM + 1)], sV[64 * (DM + 1)];
  const int tid = threadIdx.x;
  const float* s = src + (long long)blockIdx.x * seq_stride;
  for (int i = tid; i < DM * DM; i += 256) { sWk[i] = w.wk[i]; sWv[i] = w.wv[i]; }
  // thread owns KV[d][e0..e0+3]
  const int d = tid >> 3, e0 = (tid & 7) * 4;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f, ksum = 0.f;
  __syncthreads();
  for (int t0 = 0; t0 < NTOK; t0 += 64) {
    const int nt = min(64, NTOK - t0);
    // 64 tokens x 32 outputs x {K, V}: thread computes token (tid & 63), outputs j = (tid >> 6) * 8 .. +8
    {
      const int t = tid & 63, j0 = (tid >> 6) * 8;
      float kk[8], vv[8];
#pragma unroll
      for (int j = 0; j < 8; j++) { kk[j] = 0.f; vv[j] = 0.f; }
      if (t < nt) {
        const float* row = s + (long long)(t0 + t) * DM;
        for (int p = 0; p < DM; p++) {
          const float xv = row[p];
#pragma unroll
          for (int j = 0; j < 8; j++) {
            kk[j] += xv * sWk[p * DM + j0 + j];
            vv[j] += xv * sWv[p * DM + j0 + j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        sK[t * (DM + 1) + j0 + j] = t < nt ? elu1(kk[j]) : 0.f;
        sV[t * (DM + 1) + j0 + j] = t < nt ? vv[j] / 1200.0f : 0.f;
      }
    }
    __syncthreads();
    for (int t = 0; t < nt; t++) {
      const float kd = sK[t * (DM + 1) + d];
      acc0 += kd * sV[t * (DM + 1) + e0];
      acc1 += kd * sV[t * (DM + 1) + e0 + 1];
      acc2 += kd * sV[t * (DM + 1) + e0 + 2];
      acc3 += kd * sV[t * (DM + 1) + e0 + 3];
      if ((tid & 7) == 0) ksum += kd;
    }
    __syncthreads();
  }
  float* o = kv + (long long)blockIdx.x * (DM * DM + DM);
  o[d * DM + e0] = acc0; o[d * DM + e0 + 1] = acc1; o[d * DM + e0 + 2] = acc2; o[d * DM + e0 + 3] = acc3;
  if ((tid & 7) == 0) o[DM * DM + d] = ksum;
}

// phase B: one lane per token; weights are wave-uniform LDS broadcasts.
__device__ __forceinline__ void layer_norm32(float* v, const float* w, const float* b) {
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < DM; i++) mean += v[i];
  mean /= (float)DM;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < DM; i++) { const float c = v[i] - mean; var += c * c; }
  var /= (float)DM;
  const float den = sqrtf(var + 1.0000000116860974e-07f);
#pragma unroll
  for (int i = 0; i < DM; i++) v[i] = (v[i] - mean) / den * w[i] + b[i];
}

__global__ __launch_bounds__(64) void k_attn_update(const float* __restrict__ xsrc, long long x_stride,
                                                    const float* __restrict__ kv, BlockW w, float* __restrict__ dst,
                                                    long long d_stride) {
  __shared__ __attribute__((aligned(16))) float sW[DM * DM * 3 + 64 * 64 + 64 * DM + 4 * DM + DM];
  float* sWq = sW;
  float* sKV = sWq + DM * DM;
  float* sWm = sKV + DM * DM;
  float* sW0 = sWm + DM * DM;
  float* sW1 = sW0 + 64 * 64;
  float* sLN = sW1 + 64 * DM;   // n1w n1b n2w n2b
  float* sKs = sLN + 4 * DM;
  const int seq = blockIdx.y, lane = threadIdx.x;
  const float* kvp = kv + (long long)seq * (DM * DM + DM);
  for (int i = lane; i < DM * DM; i += 64) { sWq[i] = w.wq[i]; sKV[i] = kvp[i]; sWm[i] = w.wm[i]; }
  for (int i = lane; i < 64 * 64; i += 64) sW0[i] = w.w0[i];
  for (int i = lane; i < 64 * DM; i += 64) sW1[i] = w.w1[i];
  if (lane < DM) {
    sLN[lane] = w.n1w[lane]; sLN[DM + lane] = w.n1b[lane]; sLN[2 * DM + lane] = w.n2w[lane]; sLN[3 * DM + lane] = w.n2b[lane];
    sKs[lane] = kvp[DM * DM + lane];
  }
  __syncthreads();
  const int t = blockIdx.x * 64 + lane;
  if (t >= NTOK) return;
  const float* xr = xsrc + (long long)seq * x_stride + (long long)t * DM;
  float x[DM], q[DM];
#pragma unroll
  for (int i = 0; i < DM; i += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + i);
    x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
  }
  // q = x Wq; Q = elu(q) + 1
#pragma unroll
  for (int j = 0; j < DM; j++) q[j] = 0.f;
#pragma unroll
  for (int p = 0; p < DM; p++)
#pragma unroll
    for (int j = 0; j < DM; j++) q[j] += x[p] * sWq[p * DM + j];
  float z = 0.f;
#pragma unroll
  for (int j = 0; j < DM; j++) { q[j] = elu1(q[j]); z += q[j] * sKs[j]; }
  z = 1.0f / (z + 9.999999974752427e-07f);
  // msg = (Q KV) * Z * 1200
  float msg[DM];
#pragma unroll
  for (int e = 0; e < DM; e++) msg[e] = 0.f;
#pragma unroll
  for (int dd = 0; dd < DM; dd++)
#pragma unroll
    for (int e = 0; e < DM; e++) msg[e] += q[dd] * sKV[dd * DM + e];
#pragma unroll
  for (int e = 0; e < DM; e++) msg[e] = msg[e] * z * 1200.0f;
  // merge + LN1
  float mg[DM];
#pragma unroll
  for (int j = 0; j < DM; j++) mg[j] = 0.f;
#pragma unroll
  for (int p = 0; p < DM; p++)
#pragma unroll
    for (int j = 0; j < DM; j++) mg[j] += msg[p] * sWm[p * DM + j];
  layer_norm32(mg, sLN, sLN + DM);
  // MLP on [x | mg]: 64 -> 64 (ReLU) -> 32, LN2, residual
  float h[64];
#pragma unroll
  for (int j = 0; j < 64; j++) h[j] = 0.f;
#pragma unroll
  for (int p = 0; p < DM; p++)
#pragma unroll
    for (int j = 0; j < 64; j++) h[j] += x[p] * sW0[p * 64 + j];
#pragma unroll
  for (int p = 0; p < DM; p++)
#pragma unroll
    for (int j = 0; j < 64; j++) h[j] += mg[p] * sW0[(DM + p) * 64 + j];
  float o[DM];
#pragma unroll
  for (int j = 0; j < DM; j++) o[j] = 0.f;
#pragma unroll
  for (int p = 0; p < 64; p++) {
    const float hv = fmaxf(h[p], 0.f);
#pragma unroll
    for (int j = 0; j < DM; j++) o[j] += hv * sW1[p * DM + j];
  }
  layer_norm32(o, sLN + 2 * DM, sLN + 3 * DM);
  float* dr = dst + (long long)seq * d_stride + (long long)t * DM;
#pragma unroll
  for (int i = 0; i < DM; i += 4)
    *reinterpret_cast<f32x4*>(dr + i) = f32x4{x[i] + o[i], x[i + 1] + o[i + 1], x[i + 2] + o[i + 2], x[i + 3] + o[i + 3]};
}

// ------------------------------------------------------------------ matching head
// s_ij = ((fa_i / sqrt(32)) . (fb_j / sqrt(32))) / 0.1.  Row statistics (max, sum of exp) of S; calling it with the
// operands swapped gives the column statistics with bit-identical s_ij (same products, same order).
constexpr int STRIP = 16;
__global__ __launch_bounds__(256) void k_sim_stats(const float* __restrict__ fa, const float* __restrict__ fb,
                                                   long long pair_stride, float* __restrict__ stats /*[pair][2][1200]*/,
                                                   long long stats_stride) {
  __shared__ float sA[STRIP * DM];
  __shared__ float red[STRIP][256 / 64][2];
  const int pair = blockIdx.y, i0 = blockIdx.x * STRIP, tid = threadIdx.x;
  const float* A = fa + (long long)pair * pair_stride;
  const float* B = fb + (long long)pair * pair_stride;
  for (int i = tid; i < STRIP * DM; i += 256) sA[i] = A[(long long)i0 * DM + i] / 5.656854f;
  __syncthreads();
  float mx[STRIP], sm[STRIP];
#pragma unroll
  for (int r = 0; r < STRIP; r++) { mx[r] = -INFINITY; sm[r] = 0.f; }
  for (int j = tid; j < NTOK; j += 256) {
    float b[DM];
#pragma unroll
    for (int p = 0; p < DM; p += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(B + (long long)j * DM + p);
      b[p] = v.x / 5.656854f; b[p + 1] = v.y / 5.656854f; b[p + 2] = v.z / 5.656854f; b[p + 3] = v.w / 5.656854f;
    }
#pragma unroll
    for (int r = 0; r < STRIP; r++) {
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < DM; p++) s += sA[r * DM + p] * b[p];
      s = s / 0.1f;
      // online softmax statistics
      if (s > mx[r]) { sm[r] = sm[r] * expf(mx[r] - s) + 1.f; mx[r] = s; }
      else sm[r] += expf(s - mx[r]);
    }
  }
  // combine across the workgroup
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < STRIP; r++) {
    float m = mx[r], s = sm[r];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
      const float mm = fmaxf(m, m2);
      s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
      m = mm;
    }
    if (lane == 0) { red[r][wave][0] = m; red[r][wave][1] = s; }
  }
  __syncthreads();
  if (tid < STRIP) {
    float m = -INFINITY, s = 0.f;
    for (int w = 0; w < 4; w++) {
      const float m2 = red[tid][w][0], s2 = red[tid][w][1];
      const float mm = fmaxf(m, m2);
      s = (m == -INFINITY ? 0.f : s * expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * expf(m2 - mm));
      m = mm;
    }
    float* st = stats + (long long)pair * stats_stride;
    st[i0 + tid] = m;
    st[NTOK + i0 + tid] = s;
  }
}

// conf_ij = softmax_i(s)_ij * softmax_j(s)_ij, '> threshold' -> bit mask; the 5.76 MB confidence matrix is never
// written (except for the debug pair).
__global__ __launch_bounds__(256) void k_conf_mask(const float* __restrict__ f0, const float* __restrict__ f1,
                                                   long long pair_stride, const float* __restrict__ rstats,
                                                   const float* __restrict__ cstats, long long stats_stride,
                                                   float threshold, uint32_t* __restrict__ mask, float* conf_dbg,
                                                   int dbg_pair) {
  __shared__ float sA[STRIP * DM];
  __shared__ float sRm[STRIP], sRs[STRIP];
  const int pair = blockIdx.y, i0 = blockIdx.x * STRIP, tid = threadIdx.x, lane = tid & 63;
  const float* A = f0 + (long long)pair * pair_stride;
  const float* B = f1 + (long long)pair * pair_stride;
  const float* rs = rstats + (long long)pair * stats_stride;
  const float* cs = cstats + (long long)pair * stats_stride;
  for (int i = tid; i < STRIP * DM; i += 256) sA[i] = A[(long long)i0 * DM + i] / 5.656854f;
  if (tid < STRIP) { sRm[tid] = rs[i0 + tid]; sRs[tid] = rs[NTOK + i0 + tid]; }
  __syncthreads();
  uint32_t* mk = mask + (long long)pair * NTOK * MASK_WORDS;
  float* dbg = (conf_dbg && pair == dbg_pair) ? conf_dbg : nullptr;
  for (int j0 = 0; j0 < 1280; j0 += 256) {   // 5 x 256 columns cover the 38 mask words (1216 bits) of a row
    const int j = j0 + tid;
    const bool ok = j < NTOK;
    float b[DM];
    float cm = 0.f, csum = 1.f;
    if (ok) {
#pragma unroll
      for (int p = 0; p < DM; p += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(B + (long long)j * DM + p);
        b[p] = v.x / 5.656854f; b[p + 1] = v.y / 5.656854f; b[p + 2] = v.z / 5.656854f; b[p + 3] = v.w / 5.656854f;
      }
      cm = cs[j]; csum = cs[NTOK + j];
    } else {
#pragma unroll
      for (int p = 0; p < DM; p++) b[p] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < STRIP; r++) {
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < DM; p++) s += sA[r * DM + p] * b[p];
      s = s / 0.1f;
      const float conf = (expf(s - cm) / csum) * (expf(s - sRm[r]) / sRs[r]);
      const bool hit = ok && conf > threshold;   // strict '>' (dnnfeaturematcher.cpp:75)
      const unsigned long long bal = __ballot(hit);
      if (j0 + (tid & ~63) < NTOK) {
        const int word = (j0 + (tid & ~63)) >> 5;
        if (lane == 0) mk[(i0 + r) * MASK_WORDS + word] = (uint32_t)bal;
        if (lane == 32 && word + 1 < MASK_WORDS) mk[(i0 + r) * MASK_WORDS + word + 1] = (uint32_t)(bal >> 32);
      }
      if (dbg && ok) dbg[(long long)(i0 + r) * NTOK + j] = conf;
    }
  }
}

// findNonZero row-major + decode (dnnfeaturematcher.cpp:80-99): one workgroup per pair scans the bit mask in order.
__global__ __launch_bounds__(256) void k_decode(const uint32_t* __restrict__ mask, msf_match* __restrict__ out, int cap,
                                                int32_t* __restrict__ n_out) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t running;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* mk = mask + (long long)pair * NTOK * MASK_WORDS;
  msf_match* o = out + (long long)pair * cap;
  if (tid == 0) running = 0;
  __syncthreads();
  const int total = NTOK * MASK_WORDS;
  for (int w0 = 0; w0 < total; w0 += 256) {
    const int w = w0 + tid;
    const uint32_t bits = w < total ? mk[w] : 0u;
    const uint32_t c = __popc(bits);
    // inclusive scan inside the wave
    uint32_t s = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = __shfl_up(s, d);
      if (lane >= d) s += v;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    uint32_t base = running;
    for (int k = 0; k < wave; k++) base += wsum[k];
    uint32_t pos = base + s - c;
    if (bits) {
      const int i = w / MASK_WORDS, jw = (w % MASK_WORDS) * 32;
      uint32_t bb = bits;
      while (bb) {
        const int bit = __ffs(bb) - 1;
        bb &= bb - 1;
        const int j = jw + bit;
        if (pos < (uint32_t)cap) {
          msf_match m;
          m.x1 = (i % FW) * 16; m.y1 = (i / FW) * 16;   // row -> frame 1 cell, top-left corner
          m.x2 = (j % FW) * 16; m.y2 = (j / FW) * 16;   // col -> frame 2 cell
          o[pos] = m;
        }
        pos++;
      }
    }
    __syncthreads();
    if (tid == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) n_out[pair] = (int32_t)running;
}

// ================================================================== host side
struct ConvDesc {
  int cin, cout, ks, stride, hin, win, hout, wout;
  float* d_w = nullptr;   // [KSTEPS*4][NPAD]
  float* d_b = nullptr;   // [cout] or null
};

struct LoftrPipeline::Impl {
  int max_pairs = 0, chunk = 0;
  bool profile = false;
  std::vector<float*> allocs;
  ConvDesc conv[21];
  float* d_pe = nullptr;
  BlockW blk[8];
  // workspace (per chunk of pairs)
  float *bufA = nullptr, *bufB = nullptr, *bufC = nullptr, *bufD = nullptr;
  float *tok[4] = {nullptr, nullptr, nullptr, nullptr};  // f0 f1 t0 t1, each [chunk][1200][32]
  float* kv = nullptr;       // [chunk][1056]
  float* rstats = nullptr;   // [chunk][2][1200]
  float* cstats = nullptr;
  uint32_t* mask = nullptr;  // [chunk][1200][38]
  float* conf_dbg = nullptr; // [1200][1200]
  float* feat_dbg = nullptr; // [2][1200][32]
  int dbg_pair = 0;
  bool have_dbg = false;
  std::vector<hipEvent_t> ev;  // 4 per chunk
  int ev_chunks = 0;
  bool ev_ok = false, ev_rec = false;
};

LoftrPipeline::~LoftrPipeline() { destroy(); }

void LoftrPipeline::destroy() {
  if (!p_) return;
  for (float* a : p_->allocs) hipFree(a);
  for (auto& e : p_->ev) hipEventDestroy(e);
  delete p_;
  p_ = nullptr;
}

namespace {

struct Blob {
  std::map<std::string, std::pair<std::vector<uint32_t>, std::vector<float>>> t;
};

std::string load_blob(const std::string& path, Blob* b) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return "io: cannot open weights file " + path;
  char magic[8];
  uint32_t n = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "MSFLTR01", 8) != 0 || fread(&n, 4, 1, f) != 1 || n > 4096) {
    fclose(f);
    return "io: bad weights header in " + path;
  }
  struct Rec { char name[32]; uint32_t ndim, dims[4], off, count; };
  static_assert(sizeof(Rec) == 60, "record layout");
  std::vector<Rec> recs(n);
  if (fread(recs.data(), sizeof(Rec), n, f) != n) { fclose(f); return "io: truncated weights table"; }
  size_t total = 0;
  for (auto& r : recs) total = std::max(total, (size_t)r.off + r.count);
  std::vector<float> data(total);
  if (fread(data.data(), 4, total, f) != total) { fclose(f); return "io: truncated weights payload"; }
  fclose(f);
  for (auto& r : recs) {
    std::string name(r.name, strnlen(r.name, 32));
    std::vector<uint32_t> dims(r.dims, r.dims + r.ndim);
    b->t[name] = {dims, std::vector<float>(data.begin() + r.off, data.begin() + r.off + r.count)};
  }
  return "";
}

std::string default_weights() {
  Dl_info info;
  std::string dir = ".";
  if (dladdr((void*)&default_weights, &info) && info.dli_fname) {
    std::string p(info.dli_fname);
    size_t s = p.rfind('/');
    if (s != std::string::npos) dir = p.substr(0, s);
  }
  return dir + "/weights/loftr_teacher.bin";
}

}  // namespace

#define LF_TRY(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return std::string(#expr) + ": " + hipGetErrorString(e_);      \
  } while (0)

std::string LoftrPipeline::init(const char* weights_path, int max_pairs, bool profile) {
  destroy();
  p_ = new Impl();
  Impl& P = *p_;
  P.max_pairs = max_pairs;
  P.chunk = max_pairs < 32 ? max_pairs : 32;
  P.profile = profile;
  Blob blob;
  std::string err = load_blob(weights_path && weights_path[0] ? weights_path : default_weights(), &blob);
  if (!err.empty()) return err;
  auto need = [&](const std::string& n, size_t count) -> const std::vector<float>* {
    auto it = blob.t.find(n);
    if (it == blob.t.end() || it->second.second.size() != count) return nullptr;
    return &it->second.second;
  };
  auto upload = [&](const std::vector<float>& h, float** d) -> hipError_t {
    hipError_t e = hipMalloc(d, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    P.allocs.push_back(*d);
    return hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  };
  // backbone, index = execution order used by oracle and fixtures
  const int spec[21][6] = {
      // cin cout ks stride hin win
      {1, 8, 7, 2, 480, 640},   {8, 8, 3, 1, 240, 320},   {8, 8, 3, 1, 240, 320},   {8, 8, 3, 1, 240, 320},
      {8, 8, 3, 1, 240, 320},   {8, 16, 3, 2, 240, 320},  {16, 16, 3, 1, 120, 160}, {8, 16, 1, 2, 240, 320},
      {16, 16, 3, 1, 120, 160}, {16, 16, 3, 1, 120, 160}, {16, 32, 3, 2, 120, 160}, {32, 32, 3, 1, 60, 80},
      {16, 32, 1, 2, 120, 160}, {32, 32, 3, 1, 60, 80},   {32, 32, 3, 1, 60, 80},   {32, 32, 3, 2, 60, 80},
      {32, 32, 3, 1, 30, 40},   {32, 32, 1, 2, 60, 80},   {32, 32, 3, 1, 30, 40},   {32, 32, 3, 1, 30, 40},
      {32, 32, 1, 1, 30, 40}};
  for (int i = 0; i < 21; i++) {
    ConvDesc& c = P.conv[i];
    c.cin = spec[i][0]; c.cout = spec[i][1]; c.ks = spec[i][2]; c.stride = spec[i][3]; c.hin = spec[i][4]; c.win = spec[i][5];
    const int pad = c.ks / 2;
    c.hout = (c.hin + 2 * pad - c.ks) / c.stride + 1;
    c.wout = (c.win + 2 * pad - c.ks) / c.stride + 1;
    char nm[32];
    if (i < 20) snprintf(nm, sizeof nm, "conv%02d.w", i); else snprintf(nm, sizeof nm, "outconv.w");
    const auto* w = need(nm, (size_t)c.cout * c.cin * c.ks * c.ks);
    if (!w) return std::string("io: weights blob lacks ") + nm;
    const int ktot = c.ks * c.ks * c.cin, ksteps = (ktot + 3) / 4, npad = ((c.cout + 15) / 16) * 16;
    std::vector<float> wb((size_t)ksteps * 4 * npad, 0.f);
    for (int co = 0; co < c.cout; co++)
      for (int ci = 0; ci < c.cin; ci++)
        for (int ky = 0; ky < c.ks; ky++)
          for (int kx = 0; kx < c.ks; kx++) {
            const int k = (ky * c.ks + kx) * c.cin + ci;
            wb[(size_t)k * npad + co] = (*w)[(((size_t)co * c.cin + ci) * c.ks + ky) * c.ks + kx];
          }
    LF_TRY(upload(wb, &c.d_w));
    if (i < 20) {
      snprintf(nm, sizeof nm, "conv%02d.b", i);
      const auto* b = need(nm, c.cout);
      if (!b) return std::string("io: weights blob lacks ") + nm;
      LF_TRY(upload(*b, &c.d_b));
    }
  }
  {
    const auto* pe = need("pe", (size_t)DM * NTOK);
    if (!pe) return "io: weights blob lacks pe";
    LF_TRY(upload(*pe, &P.d_pe));
  }
  for (int b = 0; b < 8; b++) {
    char nm[32];
    struct { const char* n; size_t cnt; const float** dst; } items[] = {
        {"wq", 1024, &P.blk[b].wq}, {"wk", 1024, &P.blk[b].wk}, {"wv", 1024, &P.blk[b].wv},
        {"wmerge", 1024, &P.blk[b].wm}, {"wmlp0", 4096, &P.blk[b].w0}, {"wmlp1", 2048, &P.blk[b].w1}};
    for (auto& it : items) {
      snprintf(nm, sizeof nm, "blk%d.%s", b, it.n);
      const auto* w = need(nm, it.cnt);
      if (!w) return std::string("io: weights blob lacks ") + nm;
      float* d = nullptr;
      LF_TRY(upload(*w, &d));
      *it.dst = d;
    }
    struct { const char* n; const float** dst; } lns[] = {
        {"n1w", &P.blk[b].n1w}, {"n1b", &P.blk[b].n1b}, {"n2w", &P.blk[b].n2w}, {"n2b", &P.blk[b].n2b}};
    for (auto& it : lns) {
      snprintf(nm, sizeof nm, "ln%d.%s", b / 2, it.n);
      const auto* w = need(nm, 32);
      if (!w) return std::string("io: weights blob lacks ") + nm;
      float* d = nullptr;
      LF_TRY(upload(*w, &d));
      *it.dst = d;
    }
  }
  // workspace
  auto dalloc = [&](float** d, size_t floats) -> hipError_t {
    hipError_t e = hipMalloc(d, floats * sizeof(float));
    if (e == hipSuccess) P.allocs.push_back(*d);
    return e;
  };
  const size_t big = (size_t)P.chunk * 2 * 8 * 240 * 320;
  LF_TRY(dalloc(&P.bufA, big));
  LF_TRY(dalloc(&P.bufB, big));
  LF_TRY(dalloc(&P.bufC, big));
  LF_TRY(dalloc(&P.bufD, big / 2));
  for (int i = 0; i < 4; i++) LF_TRY(dalloc(&P.tok[i], (size_t)P.chunk * NTOK * DM));
  LF_TRY(dalloc(&P.kv, (size_t)P.chunk * (DM * DM + DM)));
  LF_TRY(dalloc(&P.rstats, (size_t)P.chunk * 2 * NTOK));
  LF_TRY(dalloc(&P.cstats, (size_t)P.chunk * 2 * NTOK));
  {
    float* m = nullptr;
    LF_TRY(dalloc(&m, (size_t)P.chunk * NTOK * MASK_WORDS));
    P.mask = reinterpret_cast<uint32_t*>(m);
  }
  LF_TRY(dalloc(&P.conf_dbg, (size_t)NTOK * NTOK));
  LF_TRY(dalloc(&P.feat_dbg, (size_t)2 * NTOK * DM));
  if (profile) {
    P.ev.resize((size_t)4 * ((max_pairs + P.chunk - 1) / P.chunk));
    for (auto& e : P.ev) LF_TRY(hipEventCreate(&e));
    P.ev_ok = true;
  }
  return "";
}

namespace {

template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN>
void launch_conv(const ConvDesc& c, const void* in, long long in_img_stride, int in_row_stride, const float* res,
                 float* out, int n_img, hipStream_t st) {
  using C = ConvCfg<CIN, COUT, KS, S, OTW, RELU, RES, U8IN>;
  static_assert(S == 2 || (C::PLANE % 32) == 16, "plane stride must be 16 mod 32 for stride-1 convs");
  static_assert(C::PLANE >= C::RAW, "plane too small");
  const size_t lds = (size_t)CIN * C::PLANE * sizeof(float);
  auto kern = k_conv<CIN, COUT, KS, S, OTW, RELU, RES, U8IN>;
  static bool attr_set = false;
  if (!attr_set && lds > 48 * 1024) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  dim3 grid((c.wout + OTW - 1) / OTW, (c.hout + C::OTH - 1) / C::OTH, n_img);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, in, in_img_stride, in_row_stride, c.d_w, c.d_b, res, out, c.hin,
                     c.win, c.hout, c.wout);
}

}  // namespace

hipError_t LoftrPipeline::match(int n_pairs, const uint8_t* d_a, const uint8_t* d_b, long long frame_stride,
                                int row_stride, float threshold, msf_match* d_out, int cap, int32_t* d_n_out,
                                hipStream_t st) {
  if (!p_) return hipErrorNotInitialized;
  Impl& P = *p_;
  for (int p0 = 0; p0 < n_pairs; p0 += P.chunk) {
    const int n = std::min(P.chunk, n_pairs - p0);
    const int ni = 2 * n;  // images: [0, n) = frame 1 of each pair, [n, 2n) = frame 2
    const bool first = p0 == 0;
    hipEvent_t* ev = P.ev_ok ? &P.ev[(size_t)4 * (p0 / P.chunk)] : nullptr;
    if (ev) hipEventRecord(ev[0], st);
    // the stem reads u8 frames from two arrays: launch it per array
    const ConvDesc* c = P.conv;
    float *a = P.bufA, *b = P.bufB, *cc = P.bufC, *d = P.bufD;
    const long long s8 = 8LL * 240 * 320;
    launch_conv<1, 8, 7, 2, 64, true, false, true>(c[0], d_a + (long long)p0 * frame_stride, frame_stride, row_stride, nullptr, a, n, st);
    launch_conv<1, 8, 7, 2, 64, true, false, true>(c[0], d_b + (long long)p0 * frame_stride, frame_stride, row_stride, nullptr, a + (long long)n * s8, n, st);
    // layer1 @240x320, 8 ch
    launch_conv<8, 8, 3, 1, 64, true, false, false>(c[1], a, s8, 0, nullptr, b, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, true, false>(c[2], b, s8, 0, a, cc, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, false, false>(c[3], cc, s8, 0, nullptr, b, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, true, false>(c[4], b, s8, 0, cc, a, ni, st);                 // a = 196
    // layer2 @120x160, 16 ch
    const long long s16 = 16LL * 120 * 160;
    launch_conv<8, 16, 3, 2, 32, true, false, false>(c[5], a, s8, 0, nullptr, b, ni, st);
    launch_conv<8, 16, 1, 2, 32, false, false, false>(c[7], a, s8, 0, nullptr, d, ni, st);          // shortcut
    launch_conv<16, 16, 3, 1, 32, true, true, false>(c[6], b, s16, 0, d, cc, ni, st);              // cc = 205
    launch_conv<16, 16, 3, 1, 32, true, false, false>(c[8], cc, s16, 0, nullptr, b, ni, st);
    launch_conv<16, 16, 3, 1, 32, true, true, false>(c[9], b, s16, 0, cc, a, ni, st);              // a = 212
    // layer3 @60x80, 32 ch
    const long long s32 = 32LL * 60 * 80;
    launch_conv<16, 32, 3, 2, 16, true, false, false>(c[10], a, s16, 0, nullptr, b, ni, st);
    launch_conv<16, 32, 1, 2, 16, false, false, false>(c[12], a, s16, 0, nullptr, d, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[11], b, s32, 0, d, cc, ni, st);             // cc = 221
    launch_conv<32, 32, 3, 1, 16, true, false, false>(c[13], cc, s32, 0, nullptr, b, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[14], b, s32, 0, cc, a, ni, st);             // a = 228
    // layer4 @30x40, 32 ch
    const long long s40 = 32LL * 30 * 40;
    launch_conv<32, 32, 3, 2, 16, true, false, false>(c[15], a, s32, 0, nullptr, b, ni, st);
    launch_conv<32, 32, 1, 2, 16, false, false, false>(c[17], a, s32, 0, nullptr, d, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[16], b, s40, 0, d, cc, ni, st);             // cc = 237
    launch_conv<32, 32, 3, 1, 16, true, false, false>(c[18], cc, s40, 0, nullptr, b, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[19], b, s40, 0, cc, a, ni, st);             // a = 244
    launch_conv<32, 32, 1, 1, 16, false, false, false>(c[20], a, s40, 0, nullptr, b, ni, st);      // b = 245
    if (ev) hipEventRecord(ev[1], st);
    // tokens: images [0,n) -> tok[0] (feat0), [n,2n) -> tok[1] (feat1)
    const long long ts = (long long)NTOK * DM;
    hipLaunchKernelGGL(k_tokens, dim3((n * NTOK * DM + 255) / 256), dim3(256), 0, st, b, P.d_pe, P.tok[0], n);
    hipLaunchKernelGGL(k_tokens, dim3((n * NTOK * DM + 255) / 256), dim3(256), 0, st, b + (long long)n * s40, P.d_pe, P.tok[1], n);
    // 8 encoder blocks: (x, source) -> dst   [self, self, cross, cross(updated feat0)] x 2
    float *f0 = P.tok[0], *f1 = P.tok[1], *t0 = P.tok[2], *t1 = P.tok[3];
    struct { const float* x; const float* s; float* o; } seq[8] = {
        {f0, f0, t0}, {f1, f1, t1}, {t0, t1, f0}, {t1, f0, f1}, {f0, f0, t0}, {f1, f1, t1}, {t0, t1, f0}, {t1, f0, f1}};
    for (int bi = 0; bi < 8; bi++) {
      hipLaunchKernelGGL(k_attn_kv, dim3(n), dim3(256), 0, st, seq[bi].s, ts, P.blk[bi], P.kv);
      hipLaunchKernelGGL(k_attn_update, dim3((NTOK + 63) / 64, n), dim3(64), 0, st, seq[bi].x, ts, P.kv, P.blk[bi],
                         seq[bi].o, ts);
    }
    if (ev) hipEventRecord(ev[2], st);
    // matching head on (f0, f1)
    hipLaunchKernelGGL(k_sim_stats, dim3(NTOK / STRIP, n), dim3(256), 0, st, f0, f1, ts, P.rstats, 2LL * NTOK);
    hipLaunchKernelGGL(k_sim_stats, dim3(NTOK / STRIP, n), dim3(256), 0, st, f1, f0, ts, P.cstats, 2LL * NTOK);
    const bool dbg = first;  // keep pair 0's confidence matrix + features for the parity tests
    hipLaunchKernelGGL(k_conf_mask, dim3(NTOK / STRIP, n), dim3(256), 0, st, f0, f1, ts, P.rstats, P.cstats,
                       2LL * NTOK, threshold, P.mask, dbg ? P.conf_dbg : nullptr, 0);
    hipLaunchKernelGGL(k_decode, dim3(n), dim3(256), 0, st, P.mask, d_out + (long long)p0 * cap, cap, d_n_out + p0);
    if (dbg) {
      hipMemcpyAsync(P.feat_dbg, f0, ts * sizeof(float), hipMemcpyDeviceToDevice, st);
      hipMemcpyAsync(P.feat_dbg + ts, f1, ts * sizeof(float), hipMemcpyDeviceToDevice, st);
      P.have_dbg = true;
    }
    if (ev) { hipEventRecord(ev[3], st); P.ev_rec = true; P.ev_chunks = p0 / P.chunk + 1; }
  }
  return hipGetLastError();
}

int LoftrPipeline::stage_times(const char** names, float* ms, int cap) {
  static const char* kNames[3] = {"backbone_convs", "transformer", "match_head"};
  if (!p_ || !p_->ev_ok || !p_->ev_rec) return 0;
  if (hipEventSynchronize(p_->ev[(size_t)4 * p_->ev_chunks - 1]) != hipSuccess) return 0;
  int n = 0;
  for (int i = 0; i < 3 && n < cap; i++, n++) {
    names[n] = kNames[i];
    ms[n] = 0.f;
    for (int c = 0; c < p_->ev_chunks; c++) {   // sum over the chunks of the last call
      float t = 0.f;
      if (hipEventElapsedTime(&t, p_->ev[(size_t)4 * c + i], p_->ev[(size_t)4 * c + i + 1]) == hipSuccess) ms[n] += t;
    }
  }
  return n;
}

int LoftrPipeline::debug_get(int what, int slot, int level, void* host_out, size_t cap, size_t* n_bytes,
                             std::string* err) {
  if (!p_ || !p_->have_dbg) { *err = "no LoFTR batch has run yet"; return MSF_ERR_INVALID_ARG; }
  if (slot != 0) { *err = "LoFTR debug tensors are kept for pair 0 of the last call only"; return MSF_ERR_INVALID_ARG; }
  if (hipDeviceSynchronize() != hipSuccess) { *err = "hipDeviceSynchronize failed"; return MSF_ERR_HIP; }
  const float* src = nullptr;
  size_t bytes = 0;
  if (what == MSF_DBG_LOFTR_CONF) { src = p_->conf_dbg; bytes = (size_t)NTOK * NTOK * 4; }
  else if (what == MSF_DBG_LOFTR_FEAT) { src = p_->feat_dbg; bytes = (size_t)2 * NTOK * DM * 4; }
  else { *err = "unknown debug item for LoFTR"; return MSF_ERR_INVALID_ARG; }
  *n_bytes = bytes;
  const size_t n = bytes < cap ? bytes : cap;
  if (n && hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost) != hipSuccess) { *err = "hipMemcpy failed"; return MSF_ERR_HIP; }
  return 0;
}

}  // namespace msf
