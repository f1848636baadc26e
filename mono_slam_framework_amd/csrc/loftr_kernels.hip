// LoFTR_teacher on gfx950: the graph of model/LoFTR_teacher.onnx restated as HIP kernels.
//
// Reference entry point: ::DNNFeatureMatcher::MatchFrames (src/dnnfeaturematcher.cpp:44-102):
//   ConvertImageToFloat (:5-9)  -> fused into the stem convolution's tile load
//   Ort::Session::Run (:62-64)  -> the 21 convolutions: default k_stem_strip8x / k_strip8x / k_down16x (streaming
//                                  strips), k_block16x, k_convx on split-bf16 MFMAs (three bf16 products of hi/lo-split
//                                  f32 operands, f32 accumulation) + three f32 k_conv; MSF_FLAG_LOFTR_F32: k_conv /
//                                  k_block8 / k_block16 on v_mfma_f32_16x16x4_f32 (exact f32) throughout;
//                                  k_tokens (PE add + layout), k_attn_kv / k_attn_update (8 linear-attention
//                                  encoder blocks), k_sim_stats / k_conf_mask (similarity + dual softmax)
//   '> threshold' + findNonZero + decode (:75-99) -> k_conf_mask (bit mask, conf never written to HBM) + k_decode
//
// Activations in HBM: NCHW f32 (the graph's own layout: MFMA results store as float4 runs along W) everywhere except
// BETWEEN the six streaming kernels of the default path, which hand each other "split pixels" -- channels-last 16-byte
// pixels in [hi | lo] bf16 planes, the operand format of their LDS rings (see sx_off) -- so that the producer splits once.
#include "loftr_pipeline.h"
#include "weights_io.h"

#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

// Load-queue depth of the streaming kernels: steps a row pair's loads are requested ahead of their use.  Each step of the
// queue is a statically named register set and the step loop is unrolled by the depth, so the depth sets the register
// count: at 4 k_down16x / k_strip32x needed 164 registers and ran ONE workgroup of eight waves per CU (three waves per
// SIMD allow twelve); see the defaults' comments for what each kernel measured.
#define MSF_LOFTR_STRIP8_DEPTH 3   // k_strip8x: 4 / 3 / 2 steps measured 649 / 641 / 657 us per 512 images (120 / 108 / 96 registers)
#define MSF_LOFTR_STEM_DEPTH 4   // k_stem_strip8x: 858 / 867 / 898 us
#define MSF_LOFTR_STRIP16_DEPTH 2   // k_strip16x: 452 / 435 / 419 us
#define MSF_LOFTR_STRIP32_DEPTH 2   // k_strip32x: 484 / 492 us at one workgroup per CU (164 / 145 registers); 2 steps + the 128-register cap below: two workgroups, 368-375 us
#define MSF_LOFTR_DOWN32_DEPTH 2   // k_down32x (r04, one loop for both stages: 4 / 3 / 2 steps = 479 / 484 / 505 us at 232 / 200 / 164 registers, one
                                   // workgroup per CU at any depth; r05: per-stage loops + 2 steps fit two workgroups)
#define MSF_LOFTR_DOWN16_DEPTH 3   // 110 registers, two workgroups per CU, no spills (4: 164 registers, one workgroup): 584 -> 475 us

namespace msf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NTOK = 1200, DM = 32, FW = 40;   // 30 x 40 coarse cells of 16 px
constexpr int MASK_WORDS = 38;  // ceil(1200 / 32)

// ------------------------------------------------------------------ convolution (implicit GEMM, f32 MFMA)
// D[16 px][16 cout] += A[16 px][4 k] * B[4 k][16 cout]; k enumerates (ky, kx, cin) with cin fastest.
// Block = 4 waves; wave w computes output row oy0 + w, OTW pixels wide, all output channels.
// RP = 2 ("row packing", for COUT = 8): the 16 MFMA columns carry 8 output channels of TWO adjacent output rows; k then
// spans KS + S input rows, with zero weights where a row does not contribute (3x3 stride 1: 75 % useful MFMA work
// instead of 50 %; the 7x7 stride-2 stem: 16 k steps per row pair instead of 2 x 13).
constexpr int kConvWaves = 4;   // waves (= output rows or row pairs) per workgroup; 8 measured slower on every layer (conv stack 8.8 -> 9.4 ms)
template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN, int RP = 1>
struct ConvCfg {
  static_assert(RP == 1 || (RP == 2 && COUT * RP <= 16), "row packing: 2 * COUT <= 16");
  static constexpr int OTH = kConvWaves * RP;
  static constexpr int MT = OTW / 16;
  static constexpr int NT = (COUT * RP + 15) / 16;
  static constexpr int NPAD = NT * 16;
  static constexpr int PAD = KS / 2;
  static constexpr int KY = KS + S * (RP - 1);      // input rows spanned by one MFMA column group
  static constexpr int IN_H = (OTH - 1) * S + KS;
  static constexpr int IN_W = (OTW - 1) * S + KS;
  // the tile is staged with 16-byte global loads: its first column is the 4-float-aligned x just left of the
  // window (ox0 * S is a multiple of 16), XO = floats between that column and the window's first column
  static constexpr int XO = PAD ? 4 - PAD : 0;
  static constexpr int W4 = (IN_W + XO + 3) / 4;     // float4 per tile row
  static constexpr int PITCH = 4 * W4;
  // plane stride == 16 (mod 32), so the 16 px x 2 k lanes of a ds_read_b32 group hit 32 distinct banks.  Stride-2
  // layers keep each tile row as two half rows (even columns, then odd columns): a fragment read of 16 pixels two
  // columns apart is then 16 consecutive floats of one half row (stride-2 reads hit every bank twice), and a staged
  // float4 becomes two 8-byte LDS writes instead of four scalar ones.
  static constexpr int RAW = IN_H * PITCH;
  static constexpr int PLANE = ((RAW + 15) / 32) * 32 + 16;
  static constexpr int KTOT = KY * KS * CIN;
  static constexpr int KSTEPS = (KTOT + 3) / 4;
  // offset of tile column `col` inside a row (stride 2: even columns first, then the odd ones)
  static __device__ __host__ constexpr int col_off(int col) { return S == 1 ? col : (col & 1) * (PITCH / 2) + (col >> 1); }
  static constexpr int G = KSTEPS >= 4 ? 4 : KSTEPS;   // k steps per weight-prefetch group
  static constexpr int NG = (KSTEPS + G - 1) / G;
};

// SC: the 1x1 stride-2 shortcut convolution of a down-sampling BasicBlock reads exactly the centre taps of the block's
// 3x3 stride-2 convolution, so it rides along: extra accumulators fed only in the centre-tap k steps, second output
// (bias, no ReLU).  Saves a full pass over the block input and a launch.
template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN, int RP = 1, bool SC = false>
__global__ __launch_bounds__(64 * kConvWaves) void k_conv(const void* __restrict__ in_, long long in_img_stride, int in_row_stride,
                                              const float* __restrict__ wB, const float* __restrict__ bias,
                                              const float* __restrict__ res, float* __restrict__ out, int Hin, int Win,
                                              int Hout, int Wout, const float* __restrict__ wSC,
                                              const float* __restrict__ biasSC, float* __restrict__ outSC,
                                              int n_bands) {
  static_assert(!SC || (KS == 3 && S == 2 && RP == 1 && CIN % 4 == 0), "shortcut fusion: 3x3 stride-2 blocks only");
  using C = ConvCfg<CIN, COUT, KS, S, OTW, RELU, RES, U8IN, RP>;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  // Workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2.  Re-number them so one XCD owns a
  // contiguous run of (image, band) units: the halo rows two neighbouring bands share are then fetched from HBM once.
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * C::OTH;
  const int iy0 = oy0 * S - C::PAD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;

  // A workgroup walks all the x tiles of its row band.  The 16-byte global loads of tile t+1 are issued into
  // registers before the MFMA loop of tile t and committed to LDS after it, so fetch latency, MFMAs and the
  // epilogue stores of neighbouring tiles overlap (co-resident workgroups start together and would otherwise all
  // load, then all compute: measured, the phases simply added up).
  constexpr int TOTAL = CIN * C::IN_H * C::W4;          // float4 groups per tile
  constexpr int NLD = (TOTAL + 64 * kConvWaves - 1) / (64 * kConvWaves);
  f32x4 pre[NLD];
  int loff[NLD], x4v[NLD];
  long long grow[NLD];                                  // element offset of the source row (or -1: outside the image)
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int idx = tid + 64 * kConvWaves * u;
    const int c = idx / (C::IN_H * C::W4);
    const int rem = idx - c * (C::IN_H * C::W4);
    const int r = rem / C::W4;
    x4v[u] = rem - r * C::W4;
    loff[u] = idx < TOTAL ? c * C::PLANE + r * C::PITCH + (S == 1 ? 4 : 2) * x4v[u] : -1;
    const int gy = iy0 + r;
    const bool ok = idx < TOTAL && gy >= 0 && gy < Hin;
    grow[u] = !ok ? -1 : U8IN ? (long long)gy * in_row_stride : ((long long)c * Hin + gy) * Win;
  }
  const uint8_t* in8 = static_cast<const uint8_t*>(in_) + (long long)img * in_img_stride;
  const float* inf = static_cast<const float*>(in_) + (long long)img * in_img_stride;

  // zero padding outside the image: image widths are multiples of 4, so a group is wholly inside or outside the row.
  // (Keeping the decoded offsets in registers measured faster than re-deriving them per tile, despite the occupancy.)
#define MSF_CONV_ISSUE(ox0_)                                                                                       \
  {                                                                                                                \
    const int gx0_ = (ox0_) * S - C::PAD - C::XO;                                                                  \
    _Pragma("unroll") for (int u = 0; u < NLD; u++) {                                                              \
      const int gx = gx0_ + 4 * x4v[u];                                                                            \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                                         \
      if (grow[u] >= 0 && gx >= 0 && gx + 4 <= Win) {                                                              \
        if (U8IN) {                                                                                                \
          const uint32_t q = *reinterpret_cast<const uint32_t*>(in8 + grow[u] + gx);                               \
          const float k255 = (float)(1.0 / 255.0); /* ConvertImageToFloat */                                       \
          v = f32x4{(float)(q & 0xFFu) * k255, (float)((q >> 8) & 0xFFu) * k255, (float)((q >> 16) & 0xFFu) * k255, \
                    (float)(q >> 24) * k255};                                                                      \
        } else {                                                                                                   \
          v = *reinterpret_cast<const f32x4*>(inf + grow[u] + gx);                                                 \
        }                                                                                                          \
      }                                                                                                            \
      pre[u] = v;                                                                                                  \
    }                                                                                                              \
  }

  const int ntx = (Wout + OTW - 1) / OTW;
  MSF_CONV_ISSUE(0)
  // weight fragments of the first k group: the same for every x tile, fetched once
  float bfirst[C::G][C::NT];
#pragma unroll
  for (int j = 0; j < C::G; j++)
#pragma unroll
    for (int n = 0; n < C::NT; n++) bfirst[j][n] = wB[(j * 4 + kq) * C::NPAD + n * 16 + i];
  for (int tx = 0; tx < ntx; tx++) {
    const int ox0 = tx * OTW;
    __syncthreads();                       // every wave is done reading the previous tile
#pragma unroll
    for (int u = 0; u < NLD; u++) {
      if (loff[u] < 0) continue;
      float* t = &tile[loff[u]];
      if (S == 1) {
        *reinterpret_cast<f32x4*>(t) = pre[u];        // PLANE and PITCH are multiples of 4
      } else {                                         // columns 4q .. 4q+3 -> even half [2q, 2q+1], odd half [2q, 2q+1]
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2*>(t) = f32x2{pre[u].x, pre[u].z};
        *reinterpret_cast<f32x2*>(t + C::PITCH / 2) = f32x2{pre[u].y, pre[u].w};
      }
    }
    __syncthreads();
    if (tx + 1 < ntx) MSF_CONV_ISSUE(ox0 + OTW)

    f32x4 acc[C::MT][C::NT], accS[C::MT][C::NT];
#pragma unroll
    for (int m = 0; m < C::MT; m++)
#pragma unroll
      for (int n = 0; n < C::NT; n++) { acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; accS[m][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // residual operand of the epilogue: requested now, consumed after the MFMA loop (the load used to sit exposed
    // between the last MFMA and the store: +63 us per 128 images on the 8-channel layer, i.e. its full HBM time)
    f32x4 rv[C::MT][C::NT];
    if (RES) {
#pragma unroll
      for (int n = 0; n < C::NT; n++) {
        const int col = n * 16 + i;
        const int co = RP == 1 ? col : col % COUT;
        const int oy = oy0 + wave * RP + (RP == 1 ? 0 : col / COUT);
#pragma unroll
        for (int m = 0; m < C::MT; m++) {
          const int px = ox0 + m * 16 + kq * 4;
          rv[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (col < COUT * RP && oy < Hout && px < Wout)
            rv[m][n] = *reinterpret_cast<const f32x4*>(res + (((long long)img * COUT + co) * Hout + oy) * Wout + px);
        }
      }
    }

    // k loop in groups of G steps with the weight fragments of the NEXT group in flight while this group's MFMAs
    // run (the weights come straight from global/L1).  The host pads the packed weights with zero rows up to a
    // whole number of groups.
    constexpr int G = C::G, NG = C::NG;
    float bcur[G][C::NT], bnext[G][C::NT];
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int n = 0; n < C::NT; n++) bcur[j][n] = bfirst[j][n];
    for (int grp = 0; grp < NG; grp++) {
      if (grp + 1 < NG) {
#pragma unroll
        for (int j = 0; j < G; j++)
#pragma unroll
          for (int n = 0; n < C::NT; n++) bnext[j][n] = wB[(((grp + 1) * G + j) * 4 + kq) * C::NPAD + n * 16 + i];
      }
#pragma unroll
      for (int j = 0; j < G; j++) {
        int step = grp * G + j;
        step = step < C::KSTEPS ? step : 0;        // padded steps: any valid tile address, the weights are zero
        int a_off;
        if (CIN % 4 == 0) {
          const int kk = (step * 4) / CIN;           // (ky, kx) shared by the 4 k of this step
          const int c = (step * 4) % CIN + kq;
          const int ky = kk / KS, kx = kk - ky * KS;
          a_off = c * C::PLANE + (wave * RP * S + ky) * C::PITCH + C::col_off(kx + C::XO);
        } else {                                      // stem: CIN = 1, k = ky * KS + kx, padded with zero weights
          int k = step * 4 + kq;
          k = k < C::KTOT ? k : C::KTOT - 1;
          const int ky = k / KS, kx = k - ky * KS;
          a_off = (wave * RP * S + ky) * C::PITCH + C::col_off(kx + C::XO);
        }
        float av[C::MT];
#pragma unroll
        for (int m = 0; m < C::MT; m++) av[m] = tile[a_off + (m * 16 + i)];   // stride 2: same half row, consecutive
#pragma unroll
        for (int m = 0; m < C::MT; m++)
#pragma unroll
          for (int n = 0; n < C::NT; n++)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bcur[j][n], acc[m][n], 0, 0, 0);
        if (SC) {
          constexpr int kCentre0 = (KS * (KS / 2) + KS / 2) * CIN / 4;   // first k step of tap (1, 1)
          const int sstep = grp * G + j - kCentre0;
          if (sstep >= 0 && sstep < CIN / 4) {
#pragma unroll
            for (int n = 0; n < C::NT; n++) {
              const float bs = wSC[(sstep * 4 + kq) * C::NPAD + n * 16 + i];
#pragma unroll
              for (int m = 0; m < C::MT; m++)
                accS[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bs, accS[m][n], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < G; j++)
#pragma unroll
        for (int n = 0; n < C::NT; n++) bcur[j][n] = bnext[j][n];
    }

    // epilogue: D[row = 4*(lane>>4) + r][col = lane & 15] -> out[img][cout][oy][4 consecutive px]
#pragma unroll
    for (int n = 0; n < C::NT; n++) {
      const int col = n * 16 + i;
      const int co = RP == 1 ? col : col % COUT;
      const int oy = oy0 + wave * RP + (RP == 1 ? 0 : col / COUT);
      if (col >= COUT * RP || oy >= Hout) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int m = 0; m < C::MT; m++) {
        const int px = ox0 + m * 16 + kq * 4;
        if (px >= Wout) continue;
        const long long o = (((long long)img * COUT + co) * Hout + oy) * Wout + px;
        f32x4 v = acc[m][n];
        v += f32x4{bv, bv, bv, bv};
        if (RES) v += rv[m][n];
        if (RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4*>(out + o) = v;
        if (SC) {
          const float bs = biasSC[co];
          *reinterpret_cast<f32x4*>(outSC + o) = accS[m][n] + f32x4{bs, bs, bs, bs};
        }
      }
    }
  }
#undef MSF_CONV_ISSUE
}

// ------------------------------------------------------------------ fused BasicBlock, 8 channels, stride 1 (layer1 @ 240 x 320)
// y = relu(conv2(relu(conv1(x) + b1)) + b2 + x) in ONE kernel: the intermediate t never goes to HBM.  Unfused, a block of
// this stage moves five activation-sized transfers (x read, t written, t read, x read as residual, y written) and runs at
// its HBM time; fused it moves x (+ halo rows) and y.  Same MFMA form as k_conv with RP = 2 (two output rows in the 16
// MFMA columns), the same k order in both convolutions: results are bit-identical to the two-kernel path.
//  * a workgroup (4 waves) owns a band of R = 8 output rows and walks its x tiles of 64 columns left to right;
//  * conv1 of tile k produces t rows oy0-1 .. oy0+8 (5 row pairs: wave w takes M tile w of every pair) into LDS, columns
//    64k .. 64k+63, in segment k % 2 of the t rows;
//  * conv2 of tile k-1 runs right after it: it needs t columns 64(k-1)-1 .. 64(k-1)+64, i.e. its own segment, the first
//    column of the segment just written, and the last column of tile k-2, which was copied to a halo column before its
//    segment was overwritten (two halo columns alternate) -- no t column is computed twice, only the two halo ROWS per
//    band are (10 rows for 8);
//  * zero padding of conv2 applies to t: t rows outside the image are stored as 0, columns -1 and W read a zero column.
namespace blk8 {
constexpr int R = 8, TW = 64;
constexpr int XH = R + 4;                          // x rows oy0-2 .. oy0+9
constexpr int XO = 3;                              // staged rows start at the float4-aligned column 64k - 4; window starts at 64k - 1
constexpr int XW4 = (TW + 2 + XO + 3) / 4;         // 18 float4 per staged row
constexpr int XPITCH = 4 * XW4;                    // 72
constexpr int XPLANE = ((XH * XPITCH + 15) / 32) * 32 + 16;   // == 16 (mod 32): conflict-free fragment reads (see ConvCfg)
constexpr int TROWS = R + 2;                       // t rows oy0-1 .. oy0+8
constexpr int TPITCH = 132;                        // two 64-column segments, left-halo column (128), zero column (129), pad
constexpr int TPLANE = ((TROWS * TPITCH + 15) / 32) * 32 + 16;
constexpr int KSTEPS = 24;                         // (3 + 1) rows x 3 columns x 8 channels / 4
constexpr int LDS_FLOATS = 8 * XPLANE + 8 * TPLANE;
static_assert(XPLANE % 32 == 16 && TPLANE % 32 == 16, "plane strides");
}  // namespace blk8

__global__ __launch_bounds__(256) void k_block8(const float* __restrict__ in, const float* __restrict__ w1,
                                                const float* __restrict__ b1, const float* __restrict__ w2,
                                                const float* __restrict__ b2, float* __restrict__ out, int H, int W,
                                                int n_bands) {
  using namespace blk8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;
  float* tT = lds + 8 * XPLANE;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, band) order
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m = wave;                              // this wave's M tile (16 columns) of every row pair
  const int rowsel = i >> 3, co = i & 7;           // MFMA column -> (row of the pair, output channel)
  const float* inf = in + (long long)img * 8 * H * W;
  float* outf = out + (long long)img * 8 * H * W;

  // both convolutions' weight fragments stay in registers for the whole band (row-packed [k][16], 24 k steps each)
  float bw1[KSTEPS], bw2[KSTEPS];
#pragma unroll
  for (int j = 0; j < KSTEPS; j++) {
    bw1[j] = w1[(j * 4 + kq) * 16 + i];
    bw2[j] = w2[(j * 4 + kq) * 16 + i];
  }
  const float bias1 = b1[co], bias2 = b2[co];
  // halo column (left of tile 0 = padding) and the zero column
  for (int idx = tid; idx < 8 * TROWS; idx += 256) {
    const int c = idx / TROWS, r = idx - c * TROWS;
    *reinterpret_cast<f32x4*>(&tT[c * TPLANE + r * TPITCH + 128]) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // x tile staging (as in k_conv): 16-byte global loads of tile k+1 are in flight while tile k is computed
  constexpr int TOTAL = 8 * XH * XW4;
  constexpr int NLD = (TOTAL + 255) / 256;
  f32x4 pre[NLD];
  int loff[NLD], x4v[NLD];
  long long grow[NLD];
#pragma unroll
  for (int u = 0; u < NLD; u++) {
    const int idx = tid + 256 * u;
    const int c = idx / (XH * XW4);
    const int rm = idx - c * (XH * XW4);
    const int r = rm / XW4;
    x4v[u] = rm - r * XW4;
    loff[u] = idx < TOTAL ? c * XPLANE + r * XPITCH + 4 * x4v[u] : -1;
    const int gy = oy0 - 2 + r;
    grow[u] = (idx < TOTAL && gy >= 0 && gy < H) ? ((long long)c * H + gy) * W : -1;
  }
#define MSF_BLK_ISSUE(k_)                                                                       \
  {                                                                                             \
    const int gx0_ = TW * (k_) - 4;                                                             \
    _Pragma("unroll") for (int u = 0; u < NLD; u++) {                                           \
      const int gx = gx0_ + 4 * x4v[u];                                                         \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
      if (grow[u] >= 0 && gx >= 0 && gx + 4 <= W) v = *reinterpret_cast<const f32x4*>(inf + grow[u] + gx); \
      pre[u] = v;                                                                               \
    }                                                                                           \
  }
  const int ntx = W / TW;
  MSF_BLK_ISSUE(0)
#pragma unroll
  for (int u = 0; u < NLD; u++)
    if (loff[u] >= 0) *reinterpret_cast<f32x4*>(&xT[loff[u]]) = pre[u];
  // Iteration k: conv1 of tile k, barrier, then conv2 of tile k-1 together with the hand-over to iteration k+1 (x tile
  // k+1 into LDS, last column of tile k-1 into the halo column conv2 of tile k will read), barrier: two barriers per
  // tile.  The two halo columns (128, 130) alternate so the one conv2(k-1) reads is not the one being written.
  for (int k = 0; k <= ntx; k++) {
    __syncthreads();
    if (k + 1 < ntx) MSF_BLK_ISSUE(k + 1)
    if (k < ntx) {
      // ---- conv1 of tile k: t rows 2u, 2u+1 (u = 0..4), columns 16m .. 16m+15 of the tile
      f32x4 acc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      // software pipeline: the five fragment reads of step s + 1 are issued before the five MFMAs of step s (the
      // scheduler, left alone, hoists dozens of reads)
      const float* xbase = &xT[kq * XPLANE + XO + 16 * m + i];
      float an[5], ac[5];
#pragma unroll
      for (int u = 0; u < 5; u++) ac[u] = xbase[2 * u * XPITCH];
#pragma unroll
      for (int step = 0; step < KSTEPS; step++) {
        if (step + 1 < KSTEPS) {
          const int kk = ((step + 1) * 4) / 8, c = ((step + 1) * 4) % 8;
          const int ky = kk / 3, kx = kk - ky * 3;
          const float* ap = xbase + c * XPLANE + ky * XPITCH + kx;
#pragma unroll
          for (int u = 0; u < 5; u++) an[u] = ap[2 * u * XPITCH];
        }
#pragma unroll
        for (int u = 0; u < 5; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bw1[step], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 5; u++) ac[u] = an[u];
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // 5 LDS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);   // 5 MFMAs
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int tr = 2 * u + rowsel, gy = oy0 - 1 + tr;
        f32x4 v = acc[u] + f32x4{bias1, bias1, bias1, bias1};
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (gy < 0 || gy >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};        // conv2 pads t with zeros
        *reinterpret_cast<f32x4*>(&tT[co * TPLANE + tr * TPITCH + (k & 1) * 64 + 16 * m + 4 * kq]) = v;
      }
    }
    __syncthreads();                       // t of tile k is complete; nobody reads the x tile any more
    if (k + 1 < ntx) {
#pragma unroll
      for (int u = 0; u < NLD; u++)
        if (loff[u] >= 0) *reinterpret_cast<f32x4*>(&xT[loff[u]]) = pre[u];
    }
    if (k >= 1 && k < ntx) {               // last column of tile k-1 -> the halo column conv2 of tile k reads (next iteration)
      for (int idx = tid; idx < 8 * TROWS; idx += 256) {
        const int c = idx / TROWS, r = idx - c * TROWS;
        float* row = &tT[c * TPLANE + r * TPITCH];
        row[128 + 2 * (k & 1)] = row[((k - 1) & 1) * 64 + 63];
      }
    }
    if (k >= 1) {
      // ---- conv2 of tile j = k-1: output rows oy0 + 2u, +1 (u = 0..3), columns 64j + 16m .. +15
      const int j = k - 1;
      f32x4 rv[4];
#pragma unroll
      for (int u = 0; u < 4; u++)     // residual = x, requested before the MFMA loop
        rv[u] = *reinterpret_cast<const f32x4*>(inf + ((long long)co * H + (oy0 + 2 * u + rowsel)) * W + TW * j + 16 * m + 4 * kq);
      int colterm[3];
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int cr = 16 * m + i + kx - 1;                               // column inside the tile, -1 .. 64
        colterm[kx] = cr < 0 ? 128 + 2 * (j & 1) : cr >= TW ? (k < ntx ? (k & 1) * 64 : 129) : (j & 1) * 64 + cr;
      }
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* tbase = &tT[kq * TPLANE];
      float an[4], ac[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ac[u] = tbase[colterm[0] + 2 * u * TPITCH];
#pragma unroll
      for (int step = 0; step < KSTEPS; step++) {
        if (step + 1 < KSTEPS) {
          const int kk = ((step + 1) * 4) / 8, c = ((step + 1) * 4) % 8;
          const int ky = kk / 3, kx = kk - ky * 3;
          const float* ap = tbase + c * TPLANE + ky * TPITCH + colterm[kx];
#pragma unroll
          for (int u = 0; u < 4; u++) an[u] = ap[2 * u * TPITCH];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bw2[step], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) ac[u] = an[u];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int oy = oy0 + 2 * u + rowsel;
        f32x4 v = acc[u] + f32x4{bias2, bias2, bias2, bias2};
        v += rv[u];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f32x4*>(outf + ((long long)co * H + oy) * W + TW * j + 16 * m + 4 * kq) = v;
      }
    }
  }
#undef MSF_BLK_ISSUE
}

// ------------------------------------------------------------------ fused BasicBlock, 8 channels, split-bf16 MFMA
// The same block as k_block8 on v_mfma_f32_16x16x32_bf16 (16 x the f32 MFMA rate) with every f32 operand split into
// hi = bf16(v) and lo = bf16(v - hi): v * w = hi*whi + hi*wlo + lo*whi + O(2^-16 |v w|), three MFMAs accumulating in
// f32.  One MFMA spans K = 32 = 4 input rows x 8 channels of one kx (row packing as in k_block8: the 16 output columns
// are 8 channels of two adjacent rows), so a 3 x 3 x 8 convolution of a row pair is 3 (kx) x 3 (products) MFMAs of
// 16 cycles instead of 24 of 32: the block is no longer MFMA-bound but runs at memory speed (all its MFMAs together
// are 10 % of its time), so the structure minimises traffic and exposed latency instead of matrix work:
//  * activations live in LDS channels-last, as two planes (hi, lo) of 16-byte pixels (8 x bf16): one ds_read_b128 is
//    one operand fragment (lane = pixel i of the M tile and input row kq); row pitches are multiples of 16 pixels so
//    the two input rows a lane group reads fall on disjoint banks;
//  * conv1 produces t for the tile AND its two halo columns (66 columns = 5 M tiles, the fifth nearly empty: cheap at
//    this MFMA rate), so conv2 of the same tile follows directly -- no lag, no halo hand-over, one t segment;
//  * conv1 runs transposed (A = weights, B = pixels): a lane then holds 4 consecutive channels of one t pixel, i.e.
//    one 8-byte store per plane; conv2 runs as in k_block8 (A = pixels, B = weights): 4 consecutive pixels of one
//    output channel per lane, float4 stores to NCHW;
//  * the x tile is fetched as one dword per (pixel, channel) -- lanes along x, coalesced -- two tiles ahead, and split
//    on the way into LDS; the residual operand (x again, as f32) is requested right behind the tile's own loads, while
//    its lines are still in L2 (requested when conv2 needs it, it missed: 185 of 900 us per 512 images).
// Not bit-identical to the f32 path: |error| <= ~2^-15 of the operand products per convolution; the f32 kernels
// remain behind MSF_FLAG_LOFTR_F32 / MSF_LOFTR_F32=1 (tests compare the two and the ONNX golden).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace blk8x {
constexpr int R = 8, TW = 64;
constexpr int XH = R + 4;                          // x rows oy0-2 .. oy0+9
constexpr int XP = 96;                             // x row pitch in pixels: columns 64k-2 .. 64k+65 in slots 0 .. 67; conv1's
                                                   // fifth M tile reads up to slot 81 (values unused); multiple of 16
constexpr int XPLANE = XH * XP;
constexpr int TROWS = R + 2;                       // t rows oy0-1 .. oy0+8
constexpr int TP = 80;                             // t columns 64k-1 .. 64k+64 in slots 0 .. 65 (5 M tiles written)
constexpr int TPLANE = TROWS * TP;
constexpr int LDS_BYTES = 16 * (2 * XPLANE + 2 * TPLANE);   // 62 464: two workgroups per CU
constexpr int WFRAG = 2 * 3 * 64 * 8;              // bf16 elements of one convolution's packed weights [hi|lo][kx][lane][8]
}  // namespace blk8x

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
  hi = (__bf16)v;
  lo = (__bf16)(v - (float)hi);
}

// The streaming kernels are bound by VALU issue, so their epilogues spell the cheapest split sequence out (left to the
// compiler, every value is converted twice).  Inline assembly must never read an MFMA result directly: the hazard
// recogniser does not pad the MFMA -> VALU read latency for it (measured: stale accumulators).  Every value that
// reaches cvt_pk_bf16 below has gone through a compiler-visible f32 add (bias / residual) first -- which also makes
// fmaxf a single v_max_f32, because the result of an add needs no canonicalisation.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {   // (bf16(a), bf16(b)) round-to-nearest-even, a in the low half
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// ---- "split pixels": the HBM format of the activations that travel between the streaming kernels (r05)
// An image of C = 8 CB channels, H x W, is  [y][plane: hi | lo][channel block cb][x]  pixels of 16 bytes (8 x bf16):
// exactly the operand format of the LDS rings, split ONCE by the producer on its accumulators (the split the consumer's
// loader used to redo on f32 NCHW: 24 VALU instructions and four 8-byte LDS writes per pixel block and step).  A loader
// thread fetches its pixel block as two 16-byte loads and commits two 16-byte LDS writes; a producer lane stores its
// four channels as two 8-byte halves, 16 lanes = 256 contiguous bytes per (row, plane, channel block).  Same bytes per
// value (4) and the same image stride (4 C H W) as f32 NCHW, so the ping-pong buffers are unchanged.  Values: hi =
// bf16(v), lo = bf16(v - hi) -- what split4 gives; every consumer then computes exactly what it computed from f32 input.
// sx_off: byte offset of pixel x of (row y, plane, channel block cb) in an image of CB channel blocks and width W.
__device__ __forceinline__ uint32_t sx_off(int y, int plane, int cb, int x, int W, int CB) {
  return 16u * (uint32_t)(((y * 2 + plane) * CB + cb) * W + x);
}
__device__ __forceinline__ u32x4v sx_ld(const void* base, uint32_t off) {      // SGPR base + 32-bit lane offset
#ifdef MSF_ABL_NOLOADS   // timing-only build (results invalid): no global loads in the streaming kernels' loaders
  u32x4v r = u32x4v{off, off, off, off};
  asm volatile("" : "+v"(r));
  return r;
#else
  return *reinterpret_cast<const u32x4v*>(reinterpret_cast<const char*>(base) + off);
#endif
}
// a producer lane's four channels 4 half .. 4 half + 3 of pixel (y, x), channel block cb: hi and lo halves
__device__ __forceinline__ void sx_st4(void* base, int y, int cb, int x, int half, int W, int CB, f32x4 v);
__device__ __forceinline__ void split4(f32x4 v, bf16x4& vh, bf16x4& vl) {   // 10-12 instructions for four values
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  const uint32_t h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
  // the remainders as two-wide subtractions (v_pk_add_f32 with negated operands where the register pairs allow)
  const f32x2v r0 = f32x2v{v.x, v.y} - f32x2v{__uint_as_float(h0 << 16), __uint_as_float(h0 & 0xffff0000u)};
  const f32x2v r1 = f32x2v{v.z, v.w} - f32x2v{__uint_as_float(h1 << 16), __uint_as_float(h1 & 0xffff0000u)};
  vh = __builtin_bit_cast(bf16x4, u32x2v{h0, h1});
  vl = __builtin_bit_cast(bf16x4, u32x2v{cvt_pk_bf16(r0.x, r0.y), cvt_pk_bf16(r1.x, r1.y)});
}
__device__ __forceinline__ void sx_st4(void* base, int y, int cb, int x, int half, int W, int CB, f32x4 v) {
  bf16x4 vh, vl;
  split4(v, vh, vl);
  char* ob = reinterpret_cast<char*>(base);
  const uint32_t oo = sx_off(y, 0, cb, x, W, CB) + 8u * (uint32_t)half;
#ifdef MSF_ABL_NOSTORES  // timing-only build (results invalid): the values are formed but not stored
  asm volatile("" :: "v"(vh), "v"(vl), "v"(ob + oo));
#else
  *reinterpret_cast<bf16x4*>(ob + oo) = vh;
  *reinterpret_cast<bf16x4*>(ob + (oo + 16u * (uint32_t)(CB * W))) = vl;
#endif
}

__global__ __launch_bounds__(256, 2) void k_block8x(const float* __restrict__ in, const uint16_t* __restrict__ wx1,
                                                    const float* __restrict__ b1, const uint16_t* __restrict__ wx2,
                                                    const float* __restrict__ b2, float* __restrict__ out, int H, int W,
                                                    int n_bands) {
  using namespace blk8x;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xh = reinterpret_cast<bf16x8*>(lds);
  bf16x8* xl = xh + XPLANE;
  bf16x8* th = xl + XPLANE;
  bf16x8* tl = th + TPLANE;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, band) order
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m = wave;                              // this wave's M tile (16 pixels) of every row pair
  const int rowsel = i >> 3, co = i & 7;           // conv2: MFMA column -> (row of the pair, output channel)
  const long long HW = (long long)H * W;
  const float* inf = in + (long long)img * 8 * HW;
  float* outf = out + (long long)img * 8 * HW;

  // weight fragments of both convolutions: [hi | lo][kx] (lane = (channel + 8 row, input row kq), 8 input channels)
  bf16x8 w1h[3], w1l[3], w2h[3], w2l[3];
  {
    const bf16x8* p1 = reinterpret_cast<const bf16x8*>(wx1);
    const bf16x8* p2 = reinterpret_cast<const bf16x8*>(wx2);
#pragma unroll
    for (int g = 0; g < 3; g++) {
      w1h[g] = p1[g * 64 + lane]; w1l[g] = p1[(3 + g) * 64 + lane];
      w2h[g] = p2[g * 64 + lane]; w2l[g] = p2[(3 + g) * 64 + lane];
    }
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + 4 * (kq & 1));   // conv1: lane = 4 channels of one t pixel
  const float bias2 = b2[co];                                               // conv2: lane = one channel, 4 pixels

  // x staging: item = one pixel (8 channel dwords).  Items 0..2 of a thread: row (tid >> 6) + 4u, slot (tid & 63) + 2
  // (image column 64k + (tid & 63): always inside); item 3 (threads 0..47): the four halo columns, slots 0, 1, 66, 67.
  constexpr int kNoRow = -(1 << 30);
  float pre[4][8];
  int goff[4];                                     // gy * W + column offset inside the tile window (>= -2), or kNoRow
  const int hc = tid & 3;                          // item 3: halo column 0..3
#pragma unroll
  for (int u = 0; u < 3; u++) {
    const int gy = oy0 - 2 + (tid >> 6) + 4 * u;
    goff[u] = (gy >= 0 && gy < H) ? gy * W + (tid & 63) : kNoRow;
  }
  {
    const int gy = oy0 - 2 + (tid >> 2);
    goff[3] = (tid < 4 * XH && gy >= 0 && gy < H) ? gy * W + (hc < 2 ? hc - 2 : TW + hc - 2) : kNoRow;
  }
  const int ntx = W / TW;
#define MSF_BX_ISSUE(k_)                                                                          \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                               \
      bool ok = goff[u] != kNoRow;                                                                \
      if (u == 3) ok = ok && (hc < 2 ? (k_) > 0 : (k_) + 1 < ntx);                                \
      const float* src = inf + (ok ? goff[u] : 0) + TW * (k_);                                    \
      _Pragma("unroll") for (int c = 0; c < 8; c++) pre[u][c] = ok ? src[c * HW] : 0.f;           \
    }                                                                                             \
  }
#define MSF_BX_COMMIT()                                                                           \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                               \
      if (u == 3 && tid >= 4 * XH) continue;                                                      \
      const int slot = u < 3 ? ((tid >> 6) + 4 * u) * XP + (tid & 63) + 2 : (tid >> 2) * XP + (hc < 2 ? hc : TW + hc); \
      bf16x8 vh, vl;                                                                              \
      _Pragma("unroll") for (int c = 0; c < 8; c++) {                                             \
        __bf16 a, b;                                                                              \
        split_bf16(pre[u][c], a, b);                                                              \
        vh[c] = a; vl[c] = b;                                                                     \
      }                                                                                           \
      xh[slot] = vh; xl[slot] = vl;                                                               \
    }                                                                                             \
  }
  // residual operand of tile k_: 4 consecutive pixels of channel co, rows oy0 + 2u + rowsel
#define MSF_BX_RV(dst_, k_)                                                                       \
  _Pragma("unroll") for (int u = 0; u < 4; u++)                                                   \
    dst_[u] = *reinterpret_cast<const f32x4*>(inf + ((long long)co * H + (oy0 + 2 * u + rowsel)) * W + TW * (k_) + 16 * m + 4 * kq);

  f32x4 rv_cur[4], rv_nxt[4];
  MSF_BX_ISSUE(0)
  MSF_BX_RV(rv_cur, 0)
  MSF_BX_COMMIT()
  if (1 < ntx) {
    MSF_BX_ISSUE(1)
    MSF_BX_RV(rv_nxt, 1)
  }
  bf16x4* th4 = reinterpret_cast<bf16x4*>(th);
  bf16x4* tl4 = reinterpret_cast<bf16x4*>(tl);
  for (int k = 0; k < ntx; k++) {
    __syncthreads();                       // x tile k is in LDS; every wave is done with t of tile k-1
    // ---- conv1 of tile k, transposed: D[channel + 8 row][pixel]; t rows 2u, 2u+1, t columns 16q .. 16q+15.
    // Jobs (q, u): this wave's M tile q = m for u = 0..4, then the halo M tile q = 4 for u = m (wave 0 also u = 4).
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
      const int q = pass == 0 ? m : 4;
      const int nu = pass == 0 ? 5 : (m == 0 ? 2 : 1);
      f32x4 acc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int xb = kq * XP + 16 * q + i;
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (u >= nu) continue;                       // wave-uniform
        const int uu = pass == 0 ? u : (u == 0 ? m : 4);
#pragma unroll
        for (int g = 0; g < 3; g++) {
          const bf16x8 ph = xh[xb + 2 * uu * XP + g], pl = xl[xb + 2 * uu * XP + g];
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1l[g], ph, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[g], pl, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[g], ph, acc[u], 0, 0, 0);
        }
      }
      const int tc = 16 * q + i;                   // t column slot; image column 64k - 1 + tc
      const int gx = TW * k - 1 + tc;
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (u >= nu) continue;
        const int uu = pass == 0 ? u : (u == 0 ? m : 4);
        const int tr = 2 * uu + (kq >> 1), gy = oy0 - 1 + tr;
        f32x4 v = acc[u] + bias1;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (gy < 0 || gy >= H || gx < 0 || gx >= W) v = f32x4{0.f, 0.f, 0.f, 0.f};   // conv2 pads t with zeros
        bf16x4 vh, vl;
        __bf16 a, b;
        split_bf16(v.x, a, b); vh[0] = a; vl[0] = b;
        split_bf16(v.y, a, b); vh[1] = a; vl[1] = b;
        split_bf16(v.z, a, b); vh[2] = a; vl[2] = b;
        split_bf16(v.w, a, b); vh[3] = a; vl[3] = b;
        const int slot = tr * TP + tc;
        th4[2 * slot + (kq & 1)] = vh;
        tl4[2 * slot + (kq & 1)] = vl;
      }
    }
    __syncthreads();                       // t of tile k is complete; nobody reads the x tile any more
    if (k + 1 < ntx) MSF_BX_COMMIT()
    if (k + 2 < ntx) MSF_BX_ISSUE(k + 2)
    {
      // ---- conv2 of tile k: output rows oy0 + 2u, +1 (u = 0..3), pixels 64k + 16m .. +15 (t slots + 1)
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int tb = kq * TP + 16 * m + i;
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int g = 0; g < 3; g++) {
          const bf16x8 ah = th[tb + 2 * u * TP + g], al = tl[tb + 2 * u * TP + g];
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w2l[g], acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w2h[g], acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w2h[g], acc[u], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int oy = oy0 + 2 * u + rowsel;
        f32x4 v = acc[u] + f32x4{bias2, bias2, bias2, bias2};
        v += rv_cur[u];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f32x4*>(outf + ((long long)co * H + oy) * W + TW * k + 16 * m + 4 * kq) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) rv_cur[u] = rv_nxt[u];
    if (k + 2 < ntx) { MSF_BX_RV(rv_nxt, k + 2) }
  }
#undef MSF_BX_ISSUE
#undef MSF_BX_COMMIT
#undef MSF_BX_RV
}

// ------------------------------------------------------------------ streaming 8-channel BasicBlocks (layer1), split-bf16 MFMA
// k_block8x runs at memory speed, and what it moves is 1.5 x its input (row halos of 8-row bands) + the residual
// operand + the output, once per block.  This kernel removes all of that but one read and one write for BOTH blocks of
// layer1: a workgroup owns a 64-column strip of one image and walks it top to bottom two rows (one MFMA row pair) per
// step, and every convolution stage keeps its last five row pairs in an LDS ring ([hi | lo] planes of 16-byte pixels
// as in k_block8x).  The 2 NB convolutions are a software pipeline over the steps with ONE barrier per step: stage c
// (waves 8 / NS * (c - 1) ...) works on pair n - 2c at step n, reading only what stage c-1 wrote in earlier steps:
//   step n:  pair n of x -> ring 0 (fetched three steps ahead into a statically named register queue, one split pixel =
//            16 bytes hi + 16 bytes lo per loader thread: two 16-byte loads, two 16-byte LDS writes)
//            stage 1: pair n-2 of t1 from pairs n-3 .. n-1 of x
//            stage 2: pair n-4 of y1 from pairs n-5 .. n-3 of t1, + residual pair n-4 of x (still in ring 0, hi + lo)
//            stage 3: pair n-6 of t2 from y1,  stage 4: pair n-8 of y2 from t2 + residual y1 -> global memory
// No row is fetched or computed twice, no intermediate leaves LDS; only the strip's halo columns (2 NB each side) are
// recomputed.  All stages run transposed (A = weights, B = pixels): a lane holds 4 consecutive channels of one pixel,
// i.e. one 8-byte LDS access per plane for ring writes and residual reads; the last stage splits its sums and stores the
// two 8-byte halves (16 lanes x 2 halves = 256 contiguous bytes per row and plane).  A wave holds only its own stage's
// weight fragments (24 VGPRs).
namespace strip8 {
constexpr int S = 64;                              // output columns per strip
// ring row pitch in pixels (widest ring: 64 + 2 * 4 columns).  r03 counters for k_strip8x: waves wait 52 % of their
// cycles (barrier / s_waitcnt), issue in 26 %; SQ_LDS_BANK_CONFLICT is 45 % of the LDS cycles.  A pitch of 80 slots (rows a
// multiple of 256 B apart, so that the 16 slots a ds_read_b128 lane group takes from two ring rows meet no bank twice)
// changed nothing: 642 vs 617 us -- the conflicts are not in the fragment reads, and they are not what bounds the step.
constexpr int XP = 72;
// five row pairs per ring: exactly what is live in a step (x: the pair being written, the three pairs stage 1 reads, the
// pair stage 2 takes its residual from; a pair is overwritten in the step after its last read, behind the step's
// barrier).  r04 kept a sixth; without it k_stem_strip8x's LDS is 51 KB: THREE workgroups per CU (below)
constexpr int RROWS = 10;
constexpr int RING = RROWS * XP;                   // pixel slots per plane
constexpr int TAIL = 16;                           // the fifth M tile of a stage reads up to 10 slots past a ring row
constexpr int WAVES = 8;
// k_strip8x: a step of one block has 5 + 4 M-tile jobs (stage 1 writes two halo columns more than stage 2), so with 8
// waves one wave of stage 1 takes two jobs.  Tried (r03): 10 waves, 5 per stage, every wave at most one job -- the kernel
// needs 118 VGPRs (24 of weight fragments, 32 of load queue), two workgroups per CU would need <= 96: forced there it
// spills 60 B per lane and runs 1.47 ms against 0.62 (one workgroup per CU, unforced: not faster either).  8 it stays.
// Tried (r03): strips of 62 columns (MSF_LOFTR_STRIP8_S=62) -- stage 1 then writes 64 columns = 4 M tiles and stage 2
// 62 = 4 tiles, 8 jobs for 8 waves in ONE round per step instead of 5 + 4 jobs in two; six strips instead of five, the last
// overlapping its neighbour.  774 vs 617 us: exactly the 6 / 5 more strips, i.e. a step costs the same with one job less
// on its busiest wave.  The kernel moves 2.46 GB in 0.617 ms = 4.0 TB/s of mixed read + write traffic: it runs at what
// the memory system gives, and the step time is the prefetch distance (4 steps) into the loaded-memory latency.
template <int NB>
constexpr int lds_bytes() { return 16 * (2 * NB * 2 * RING + TAIL); }
__device__ __forceinline__ int ring_row(int r) {   // r mod RROWS for r >= -4 RROWS (only when the cursors are set up: they are carried afterwards)
  return (r + 4 * RROWS) % RROWS;
}
}  // namespace strip8

struct StripW {
  const uint16_t* wx[4];
  const float* b[4];
};

// (r05: capped at 80 registers for three workgroups per CU -- its 46 KB of LDS would allow them -- the kernel spills 32
// registers into its step loops: 637-649 -> 1 634 us.  It needs 117: 24 of weights, 24 of load queue, 24 of fragments.)
template <int NB, int WV, int SB>
__global__ __launch_bounds__(64 * WV) void k_strip8x(const float* __restrict__ in, StripW sw, float* __restrict__ out,
                                                     int H, int W, int n_strips) {
  using namespace strip8;
  constexpr int WAVES = WV;                        // (hides strip8::WAVES, the stem kernel's count)
  constexpr int S = SB;                            // output columns per strip (hides strip8::S, the stem kernel's 64)
  constexpr int NS = 2 * NB;                       // convolution stages
  constexpr int WPS = WAVES / NS;                  // waves per stage
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* ring = reinterpret_cast<bf16x8*>(lds);   // ring c (c = 0: x): hi plane at c * 2 RING, lo plane RING behind it
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, strip) order
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);              // SGPRs: uniform base pointers below
  // strips are S columns apart; the last one is moved left to end at the image's edge (it recomputes columns of its
  // neighbour: the same values, written twice)
  const int X0 = __builtin_amdgcn_readfirstlane(min((unit - img * n_strips) * S, W - S));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int HW = H * W;                            // one image is 8 HW floats: 32-bit offsets from wave-uniform bases
  const float* inf = in + (long long)img * 8 * HW;                  // SGPR bases: loads / stores take a 32-bit lane offset
  float* outf = out + (long long)img * 8 * HW;
  const int npairs = H / 2;
  // this wave's stage (1-based) and its share of the stage's M tiles
  // (the stage's LAST wave takes M tile 0 and the extra fifth tile: the first ones carry the loader)
  const int cst = __builtin_amdgcn_readfirstlane(wave / WPS) + 1, ws = WPS - 1 - __builtin_amdgcn_readfirstlane(wave % WPS);

  bf16x8 wh[3], wl[3];
  {
    const bf16x8* pw = reinterpret_cast<const bf16x8*>(sw.wx[cst - 1]);
#pragma unroll
    for (int g = 0; g < 3; g++) {
      wh[g] = pw[g * 64 + lane];
      wl[g] = pw[(3 + g) * 64 + lane];
    }
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>(sw.b[cst - 1] + 4 * (kq & 1));   // lane = channels 4 (kq & 1) .. +3
  {  // rows above the image (pair -1) and everything a stage has not written yet read as zero
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; j++) z[j] = (__bf16)0.f;
    for (int idx = tid; idx < NS * 2 * RING + TAIL; idx += 64 * WAVES) ring[idx] = z;
  }
  // x loader (the FIRST 2 XW threads of the workgroup): thread (row lr of the pair, ring column lc) fetches one pixel
  // (16 bytes hi + 16 bytes lo) per step.  r05: the loaders sit in stage-1 waves, which never store to global memory.
  // s_waitcnt vmcnt counts loads and stores together, in order, and the compiler counts only the operations every path
  // issues: in a wave that also ran the last stage, the wait for a load requested three steps earlier became "all but
  // the four youngest operations" -- the stores of the step just finished included, i.e. a wait for a store
  // acknowledgement (microseconds under load) in every step.
  constexpr int XW = S + 2 * NS;
  const int ltid = tid < 2 * XW ? tid : -1;
  const bool ld = ltid >= 0;
  const int lr = ld ? ltid / XW : 0, lc = ld ? ltid - lr * XW : 0;
  const int lgx = X0 - NS + lc;
  const bool colok = ld && lgx >= 0 && lgx < W;
  const uint32_t lofs = colok ? (uint32_t)lgx : 0u;
  const bool ldwave = wave <= (2 * XW - 1) / 64;                 // wave-uniform: waves with loader threads
  static_assert((2 * XW - 1) / 64 < WPS, "the loader must fit into the stage-1 waves");
  // The loads of a step are consumed kQDepth steps later.  The compiler's s_waitcnt insertion only keeps that distance if
  // every path between issue and use issues the same loads: no guard around an issue (addresses are clamped to valid
  // ones instead, the value is discarded at commit time), and loader and non-loader waves run separate copies of the
  // step loop (with a per-wave `if` around the issue it waited for vmcnt(0) at every step).
#define MSF_ST_ISSUE(q_, n_)                                                                      \
  {                                                                                               \
    const int gy = 2 * (n_) + lr;                                                                 \
    const bool ok = colok && gy < H;                                                              \
    const uint32_t so = ok ? sx_off(gy, 0, 0, (int)lofs, W, 1) : 0u;   /* byte offset from the SGPR base */ \
    q_[0] = sx_ld(inf, so);                                                                       \
    q_[1] = sx_ld(inf, so + 16u * (uint32_t)W);                                                   \
  }
#define MSF_ST_COMMIT(q_, n_)                                                                     \
  if (ld) {                                                                                       \
    const bool ok = colok && 2 * (n_) + lr < H;   /* outside the image: the (valid-address) load is discarded */ \
    const u32x4v z4 = u32x4v{0u, 0u, 0u, 0u};                                                     \
    u32x4v* dst = reinterpret_cast<u32x4v*>(ring + (crow + lc));                                  \
    dst[0] = ok ? q_[0] : z4; dst[RING] = ok ? q_[1] : z4;                                        \
  }
  // Ring rows advance by two per step: the row cursors (row * XP, modulo the ring) are carried from step to step
  // instead of being re-derived (a multiply-shift modulo per cursor and step showed in the VALU-bound profile).
  constexpr int RWRAP = RROWS * XP;
  const bool edge = X0 == 0 || X0 + S == W;        // wave-uniform: only the outer strips have columns outside the image
  constexpr int kQDepth = MSF_LOFTR_STRIP8_DEPTH;  // load queue: steps ahead, each step's registers named statically, loop unrolled by it
  static_assert(kQDepth >= 2 && kQDepth <= 4, "load queue depth");
  // stage c makes pair n - 2c at step n (the loader writes pair n at step n); its last pair at step npairs - 1 + 2 NS
  const int nsteps = ((npairs + 2 * NS + kQDepth - 1) / kQDepth) * kQDepth;
  // r05: ONE COPY OF THE STEP LOOP PER (loader?, stage), everything a stage decides -- last stage / residual / number of M
  // tiles / ring addresses -- a compile-time constant, and the steps in which every pair index is inside the image (all
  // but the first and last few) without a single range check.  r04's loop was one body for all stages: its uniform
  // `if`s were 12 scalar branches per wave and step in k_strip8x and 24 in k_stem_strip8x (SQ_INSTS_BRANCH), each a
  // fetch bubble for its wave, on waves that issue ~70 vector instructions per step.
  auto run = [&](auto is_loader, auto stage_c) {
    constexpr bool kLd = decltype(is_loader)::value;
    constexpr int CST = decltype(stage_c)::value;
    constexpr bool last = CST == NS, has_res = (CST & 1) == 0;
    constexpr int MT = (S + 2 * (NS - CST) + 15) / 16;   // M tiles: S (+ 2 (NS - c) halo) columns
    static_assert(MT >= WPS && MT <= 2 * WPS, "one or two M tiles per wave");
    constexpr int LAG = 2 * CST;
    const bool two = ws + WPS < MT;                      // wave-uniform: this wave also makes the stage's extra tile
    const bf16x8* inh = ring + (CST - 1) * 2 * RING;
    const bf16x4* resh = reinterpret_cast<const bf16x4*>(ring + (has_res ? CST - 2 : 0) * 2 * RING);
    bf16x4* outh = reinterpret_cast<bf16x4*>(ring + (last ? 0 : CST) * 2 * RING);
    int crow = ring_row(lr) * XP;                        // loader: row 2n + lr
    int rin = ring_row(-2 * LAG - 1 + kq) * XP;          // sums: fragment row 2p - 1 + kq, p = n - LAG
    int ror = ring_row(-2 * LAG + (kq >> 1)) * XP;       // epilogue: row 2p + (kq >> 1) of the output / residual ring
    // epilogue of M tile q of pair pe (sums v) -- or, for pe == npairs (CHK only), the zero rows below the image
    auto epilogue = [&](auto chk, int pe, int q, f32x4 v) {
      constexpr bool CHK = decltype(chk)::value;
      const int j = 16 * q + i;                    // this lane's pixel slot in the stage's output geometry
      if (CHK && pe >= npairs) {                   // the pair below the image: the next stage's zero padding
        if (!last && j < XP) {
          bf16x4 z;
          z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
          bf16x4* op = outh + (2 * (ror + j) + (kq & 1));
          op[0] = z; op[2 * RING] = z;
        }
        return;
      }
      v += bias;
      if (has_res) {                               // second convolution of a block: + block input (ring CST-2, slot j + 2)
        const bf16x4* rp = resh + (2 * (ror + j + 2) + (kq & 1));
        const bf16x4 a = rp[0], b = rp[2 * RING];
        v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
        v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
      }
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (last) {
        if (j < S)                                 // (the last tile of a 62-column strip holds two columns of the next strip)
          sx_st4(outf, 2 * pe + (kq >> 1), 0, X0 + j, kq & 1, W, 1, v);
      } else {
        if (edge) {
          const int gx = X0 - (NS - CST) + j;      // image column of slot j of ring CST
          if (gx < 0 || gx >= W) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (j < XP) {
          bf16x4 vh, vl;
          split4(v, vh, vl);
          bf16x4* op = outh + (2 * (ror + j) + (kq & 1));
          op[0] = vh; op[2 * RING] = vl;
        }
      }
    };
    // M tile q of this wave, pair p: fragment reads, MFMAs, epilogue
    auto tile = [&](auto chk, int q, int p) {
      constexpr bool CHK = decltype(chk)::value;
      const bool sums = !CHK || (p >= 0 && p < npairs);     // wave-uniform
      bf16x8 fh[3], fl[3];
      if (sums) {
        const bf16x8* src = inh + (rin + 16 * q + i);
#pragma unroll
        for (int g = 0; g < 3; g++) { fh[g] = src[g]; fl[g] = src[RING + g]; }
      }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sums) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
#ifndef MSF_ABL_ONEMFMA   // timing-only build (results invalid): one product per fragment instead of three
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[g], fh[g], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g], fl[g], acc, 0, 0, 0);
#endif
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g], fh[g], acc, 0, 0, 0);
        }
      }
      if (!CHK || (p >= 0 && p <= npairs)) epilogue(chk, p, q, acc);
    };
    // one step: barrier (everything written in the previous step is visible; nothing read in it is overwritten before),
    // pair n of x into its ring, the queue slot refilled with pair n + depth, this wave's stage
#define MSF_ST_STEP(chk_, q_, n_)                                                                 \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_ST_COMMIT(q_, n_)                                                                       \
      MSF_ST_ISSUE(q_, (n_) + kQDepth)                                                            \
    }                                                                                             \
    tile(chk_, ws, (n_) - LAG);                                                                   \
    if (two) tile(chk_, ws + WPS, (n_) - LAG);                                                    \
    crow += 2 * XP; crow = crow >= RWRAP ? crow - RWRAP : crow;                                   \
    rin += 2 * XP; rin = rin >= RWRAP ? rin - RWRAP : rin;                                        \
    ror += 2 * XP; ror = ror >= RWRAP ? ror - RWRAP : ror;                                        \
  }
    u32x4v q0[2]; u32x4v q1[2];
    u32x4v q2[2];
    if (kLd) {
      MSF_ST_ISSUE(q0, 0)
      MSF_ST_ISSUE(q1, 1)
      MSF_ST_ISSUE(q2, 2)
    }
    // steps n with 0 <= n - LAG <= npairs - 1 need no range check on the stage's pairs
    const int n_lo = ((LAG + kQDepth - 1) / kQDepth) * kQDepth, n_hi = ((npairs + LAG) / kQDepth) * kQDepth;
    for (int n = 0; n < nsteps; n += kQDepth) {
      if (n >= n_lo && n < n_hi) {
        MSF_ST_STEP(std::false_type{}, q0, n)
        MSF_ST_STEP(std::false_type{}, q1, n + 1)
        MSF_ST_STEP(std::false_type{}, q2, n + 2)
      } else {
        MSF_ST_STEP(std::true_type{}, q0, n)
        MSF_ST_STEP(std::true_type{}, q1, n + 1)
        MSF_ST_STEP(std::true_type{}, q2, n + 2)
      }
    }
#undef MSF_ST_STEP
  };
  static_assert(kQDepth == 3, "the step macro is written out for a three-step queue");
  // dispatch: the loader lives in stage-1 waves only
  if (cst == 1) {
    if (ldwave) run(std::true_type{}, std::integral_constant<int, 1>{});
    else run(std::false_type{}, std::integral_constant<int, 1>{});
  } else if (NS == 2 || cst == 2) {
    run(std::false_type{}, std::integral_constant<int, 2>{});
  } else if (cst == 3) {
    run(std::false_type{}, std::integral_constant<int, NS >= 3 ? 3 : 1>{});
  } else {
    run(std::false_type{}, std::integral_constant<int, NS >= 4 ? 4 : 1>{});
  }
#undef MSF_ST_ISSUE
#undef MSF_ST_COMMIT
}

// ------------------------------------------------------------------ stem + first BasicBlock as one streaming pass
// k_strip8x<1> with the stem (ConvertImageToFloat + 7x7 stride-2 convolution, 1 -> 8 channels, + ReLU) as stage 0: ring 0
// is no longer fetched as f32 activations but computed from the u8 frame, of which 4 rows x 148 bytes arrive per step
// (the stem's output, 2.46 MB per image, is neither written nor read back).  Stage 0 keeps 16 image rows in LDS as bf16
// (0..255 are exact in bf16: no lo plane) and runs the same transposed, row-packed MFMA form: K = 32 of one MFMA = 4
// image rows x 8 consecutive pixels (7 taps of a row + one with zero weight); an output row pair spans 9 image rows,
// i.e. 3 MFMA groups x 2 products (weights hi and lo; the 1 / 255 of ConvertImageToFloat is folded into them).  A
// fragment starts at pixel 2 j of the image ring row, i.e. on a 4-byte boundary: four dword reads.
// Pipeline lags: stage 0 makes pair n - 2 of x at step n (its last image row arrived at step n - 1), stage c >= 1 pair
// n - 2 - 2c.  Waves 0-2 run the stem, 3-5 the first convolution, 6-7 the second; the image loader lives in waves 5-7.
namespace stem8 {
using namespace strip8;
constexpr int NS = 2;                              // one BasicBlock behind the stem
constexpr int IROWS = 16, IP = 160;                // image ring: 16 rows of 160 bf16 pixels; row pitch = 80 dwords == 16 (mod 32)
constexpr int IDW = 37;                            // aligned dwords fetched per image row: tile pixels -1 .. 146
constexpr int NLOAD = 4 * IDW;                     // loader threads (4 image rows per step)
constexpr int LDS_BYTES = 16 * (NS * 2 * RING + TAIL) + 2 * IROWS * IP + 64;
constexpr int WFRAG = 2 * 3 * 64 * 8;              // stem fragments [hi | lo][row group][lane][8]
}  // namespace stem8

// r05: three workgroups per CU -- 51 KB of LDS (five-pair rings) and six waves per SIMD = 80 registers (86 uncapped: five
// registers are spilled, 20 B of scratch).  711-723 -> 683 us per 512 images; the grid (2 560 workgroups) is 3.3 rounds of
// the 768 resident ones, the last one a third full.
#ifndef MSF_STEM_WPE
#define MSF_STEM_WPE 6
#endif
#define MSF_STEM_ATTR __attribute__((amdgpu_waves_per_eu(MSF_STEM_WPE, MSF_STEM_WPE)))
__global__ __launch_bounds__(64 * strip8::WAVES) MSF_STEM_ATTR void k_stem_strip8x(const uint8_t* __restrict__ framesA, int nA,
                                                                     const uint8_t* __restrict__ framesB, long long frame_stride,
                                                                     int row_stride, const uint16_t* __restrict__ wx0,
                                                                     const float* __restrict__ b0, StripW sw,
                                                                     float* __restrict__ out, int H, int W, int n_strips) {
  using namespace stem8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* ring = reinterpret_cast<bf16x8*>(lds);
  __bf16* iring = reinterpret_cast<__bf16*>(ring + NS * 2 * RING + TAIL);
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, strip) order
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);
  const int X0 = __builtin_amdgcn_readfirstlane((unit - img * n_strips) * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int HW = H * W, Hin = 2 * H, Win = 2 * W;
  // images 0 .. nA-1 come from the first frame array, the rest from the second (one launch: whole rounds of workgroups)
  const uint8_t* fr = img < nA ? framesA + (long long)img * frame_stride : framesB + (long long)(img - nA) * frame_stride;
  float* outf = out + (long long)img * 8 * HW;
  const int npairs = H / 2;
  // this wave's stage: 0 = stem (waves 0-2), 1 / 2 = the block's convolutions (waves 3-5 / 6-7)
  const int sid = __builtin_amdgcn_readfirstlane(wave < 3 ? 0 : wave < 6 ? 1 : 2);
  const int ws = __builtin_amdgcn_readfirstlane(wave < 3 ? wave : wave < 6 ? wave - 3 : wave - 6);

  bf16x8 wh[3], wl[3];
  {
    const bf16x8* pw = reinterpret_cast<const bf16x8*>(sid == 0 ? wx0 : sw.wx[sid - 1]);
#pragma unroll
    for (int g = 0; g < 3; g++) {
      wh[g] = pw[g * 64 + lane];
      wl[g] = pw[(3 + g) * 64 + lane];
    }
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>((sid == 0 ? b0 : sw.b[sid - 1]) + 4 * (kq & 1));
  {  // rings and image rows above the frame read as zero
    uint32_t* z = reinterpret_cast<uint32_t*>(lds);
    for (int idx = tid; idx < LDS_BYTES / 4; idx += 64 * WAVES) z[idx] = 0u;
  }
  // image loader: thread (row r4 of the step's four, aligned dword d) fetches 4 pixels.  The
  // tile's pixel 0 is image column CB = 2 (X0 - NS) - 3 == 1 (mod 4): the dword at CB - 1 + 4d holds tile pixels 4d-1 .. 4d+2.
  // (r05: the FIRST NLOAD threads, i.e. the stem waves, which never store to global memory -- see k_strip8x)
  const int ltid = tid < NLOAD ? tid : -1;
  const bool ld = ltid >= 0;
  const int r4 = ld ? ltid / IDW : 0, dd = ld ? ltid - r4 * IDW : 0;
  const int cx = 2 * (X0 - NS) - 4 + 4 * dd;                      // image column of the dword (a multiple of 4)
  const bool colok = ld && cx >= 0 && cx < Win;
  const uint32_t lofs = colok ? (uint32_t)cx : 0u;
#define MSF_SS_ISSUE(q_, n_)                                                                      \
  {                                                                                               \
    const int gy = 4 * (n_) + r4;                                                                 \
    const uint32_t so = (colok && gy < Hin) ? lofs + (uint32_t)(gy * row_stride) : 0u;            \
    q_ = *reinterpret_cast<const uint32_t*>(fr + so);                                             \
  }
#define MSF_SS_COMMIT(q_, n_)                                                                     \
  if (ld) {                                                                                       \
    const int gy = 4 * (n_) + r4;                                                                 \
    const uint32_t v = (colok && gy < Hin) ? q_ : 0u;                                             \
    __bf16* dst = iring + ((gy & (IROWS - 1)) * IP + 4 * dd);                                     \
    const __bf16 p0 = (__bf16)(float)(v & 0xFFu), p1 = (__bf16)(float)((v >> 8) & 0xFFu);         \
    const __bf16 p2 = (__bf16)(float)((v >> 16) & 0xFFu), p3 = (__bf16)(float)(v >> 24);          \
    if (dd > 0) dst[-1] = p0;                                                                     \
    dst[0] = p1; dst[1] = p2; dst[2] = p3;                                                        \
  }
  constexpr int RWRAP = RROWS * XP;
  const bool edge = X0 == 0 || X0 + S == W;
  constexpr int kQDepth = MSF_LOFTR_STEM_DEPTH;    // load queue: steps ahead, statically named registers, loop unrolled by it
  static_assert(kQDepth == 4, "the step macro is written out for a four-step queue");
  // Lags behind the image loader (rows 4n .. 4n + 3 arrive at step n; the stem's pair p needs rows up to 4p + 5, arrived
  // at step p + 1): stage s makes pair n - 2 - 2s at step n
  const int nsteps = ((npairs + 2 * NS + 2 + kQDepth - 1) / kQDepth) * kQDepth;
  // one copy of the step loop per stage (the loader lives in the stem waves), see k_strip8x
  auto run = [&](auto is_loader, auto stage_c) {
    constexpr bool kLd = decltype(is_loader)::value;
    constexpr int SID = decltype(stage_c)::value;
    constexpr bool last = SID == NS, has_res = SID == 2;
    constexpr int MT = last ? 4 : 5, wps = SID == 2 ? 2 : 3;
    constexpr int LAG = 2 + 2 * SID;
    const bool two = ws + wps < MT;                      // wave-uniform
    const bf16x8* inh = ring + (SID > 0 ? SID - 1 : 0) * 2 * RING;
    const bf16x4* resh = reinterpret_cast<const bf16x4*>(ring);
    bf16x4* outh = reinterpret_cast<bf16x4*>(ring + (last ? 0 : SID) * 2 * RING);
    int rin = ring_row(-2 * LAG - 1 + kq) * XP;          // sums: fragment row 2p - 1 + kq of ring SID - 1
    int ror = ring_row(-2 * LAG + (kq >> 1)) * XP;       // epilogue: row 2p + (kq >> 1)
    auto epilogue = [&](auto chk, int pe, int q, f32x4 v) {
      constexpr bool CHK = decltype(chk)::value;
      const int j = 16 * q + i;
      if (CHK && pe >= npairs) {                   // the pair below the image: the next stage's zero padding
        if (!last && j < XP) {
          bf16x4 z;
          z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
          bf16x4* op = outh + (2 * (ror + j) + (kq & 1));
          op[0] = z; op[2 * RING] = z;
        }
        return;
      }
      v += bias;
      if (has_res) {
        const bf16x4* rp = resh + (2 * (ror + j + 2) + (kq & 1));
        const bf16x4 a = rp[0], b = rp[2 * RING];
        v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
        v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
      }
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (last) {
        sx_st4(outf, 2 * pe + (kq >> 1), 0, X0 + j, kq & 1, W, 1, v);
      } else {
        if (edge) {
          const int gx = X0 - (NS - SID) + j;      // image column of slot j of ring SID
          if (gx < 0 || gx >= W) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (j < XP) {
          bf16x4 vh, vl;
          split4(v, vh, vl);
          bf16x4* op = outh + (2 * (ror + j) + (kq & 1));
          op[0] = vh; op[2 * RING] = vl;
        }
      }
    };
    auto tile = [&](auto chk, int q, int p) {
      constexpr bool CHK = decltype(chk)::value;
      const bool sums = !CHK || (p >= 0 && p < npairs);     // wave-uniform
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      bf16x8 fh[3], fl[3];
      if (sums) {
        if (SID == 0) {
          // stem: image rows 4p - 3 + 4g + kq, pixels 2j .. 2j + 7 of the tile
#pragma unroll
          for (int g = 0; g < 3; g++) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(iring + (((4 * p - 3 + 4 * g + kq) & (IROWS - 1)) * IP + 2 * (16 * q + i)));
            fh[g] = __builtin_bit_cast(bf16x8, u32x4{src[0], src[1], src[2], src[3]});
          }
        } else {
          const bf16x8* src = inh + (rin + 16 * q + i);
#pragma unroll
          for (int g = 0; g < 3; g++) { fh[g] = src[g]; fl[g] = src[RING + g]; }
        }
      }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (sums) {
        if (SID == 0) {
#pragma unroll
          for (int g = 0; g < 3; g++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[g], fh[g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g], fh[g], acc, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 3; g++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[g], fh[g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g], fl[g], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[g], fh[g], acc, 0, 0, 0);
          }
        }
      }
      if (!CHK || (p >= 0 && p <= npairs)) epilogue(chk, p, q, acc);
    };
#define MSF_SS_STEP(chk_, q_, n_)                                                                 \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_SS_COMMIT(q_, n_)                                                                       \
      MSF_SS_ISSUE(q_, (n_) + kQDepth)                                                            \
    }                                                                                             \
    tile(chk_, ws, (n_) - LAG);                                                                   \
    if (two) tile(chk_, ws + wps, (n_) - LAG);                                                    \
    rin += 2 * XP; rin = rin >= RWRAP ? rin - RWRAP : rin;                                        \
    ror += 2 * XP; ror = ror >= RWRAP ? ror - RWRAP : ror;                                        \
  }
    uint32_t q0 = 0; uint32_t q1 = 0;
    uint32_t q2 = 0;
    uint32_t q3 = 0;
    if (kLd) {
      MSF_SS_ISSUE(q0, 0)
      MSF_SS_ISSUE(q1, 1)
      MSF_SS_ISSUE(q2, 2)
      MSF_SS_ISSUE(q3, 3)
    }
    const int n_lo = ((LAG + 1 + kQDepth - 1) / kQDepth) * kQDepth, n_hi = ((npairs + LAG) / kQDepth) * kQDepth;
    for (int n = 0; n < nsteps; n += kQDepth) {
      if (n >= n_lo && n < n_hi) {
        MSF_SS_STEP(std::false_type{}, q0, n)
        MSF_SS_STEP(std::false_type{}, q1, n + 1)
        MSF_SS_STEP(std::false_type{}, q2, n + 2)
        MSF_SS_STEP(std::false_type{}, q3, n + 3)
      } else {
        MSF_SS_STEP(std::true_type{}, q0, n)
        MSF_SS_STEP(std::true_type{}, q1, n + 1)
        MSF_SS_STEP(std::true_type{}, q2, n + 2)
        MSF_SS_STEP(std::true_type{}, q3, n + 3)
      }
    }
#undef MSF_SS_STEP
  };
  static_assert((NLOAD - 1) / 64 < 3, "the image loader must fit into the stem waves");
  if (sid == 0) run(std::true_type{}, std::integral_constant<int, 0>{});
  else if (sid == 1) run(std::false_type{}, std::integral_constant<int, 1>{});
  else run(std::false_type{}, std::integral_constant<int, 2>{});
#undef MSF_SS_ISSUE
#undef MSF_SS_COMMIT
}

// ------------------------------------------------------------------ streaming down-sampling BasicBlock 8 -> 16 (layer2, block 1)
// u = relu(conv3x3(relu(conv3x3_s2(x))) + conv1x1_s2(x)) as one pass in the style of k_strip8x: a workgroup owns 32 output
// columns (64 + 5 input columns) of one image and walks down two output rows (four input rows) per step.
//   ring X: 12 input rows, 8 channels, columns de-interleaved (even | odd halves), so the stride-2 fragments of 16
//           neighbouring output pixels are 16 consecutive slots;
//   stage 1 (waves 0-5, one (row, M tile) job each): t = relu(conv3x3 stride 2) and the shortcut sc = conv1x1 stride 2
//           of the same output pixels -- the shortcut's input pixel is the centre tap, i.e. the fragment of kx = 1 in
//           the ky = 1 group, so it costs three more MFMAs and no LDS read.  K = 32 of an MFMA = the 3 kx of one ky x 8
//           channels (+ one zero block).  Both results go to rings (16 channels = 2 channel-block planes, hi | lo);
//   stage 2 (waves 6-7, two jobs each): u = relu(conv3x3(t) + sc) with k_block16x's K grouping (two taps x two channel
//           blocks per MFMA, five groups), stored to global memory as split pixels (sx_st4: 256 contiguous bytes per row and plane).
// Pipeline: input rows 4n .. 4n+3 arrive at step n (fetched four steps ahead), stage 1 makes row pair n - 1, stage 2
// pair n - 3; one barrier per step.  Moves x once and u once instead of x, t, sc twice and u (4.9 -> 1.9 GB per step).
namespace down16 {
constexpr int S = 32;                              // output columns per strip
constexpr int TW = S + 2;                          // t / sc columns (stage 2's halo)
constexpr int IROWS = 12, IPX = 76, IODD = 38;     // input ring: rows, pitch, first slot of the odd-column half
constexpr int INW = 2 * TW + 1;                    // 69 input columns: 2 X0 - 3 .. 2 X0 + 65
constexpr int IPLANE = IROWS * IPX;                // slots per plane (hi, lo)
constexpr int TROWS = 8, TPX = 36;                 // t / sc rings: four row pairs, 34 columns
constexpr int TCB = TROWS * TPX;                   // 288 = 18 x 16 slots per channel-block plane
constexpr int TRING = 4 * TCB;                     // [hi | lo][channel block]
static_assert(TCB % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
constexpr int WAVES = 8;
constexpr int NLOAD = 4 * INW;                     // loader threads: one input pixel (16 bytes hi + 16 bytes lo) each per step
constexpr int LDS_BYTES = 16 * (2 * IPLANE + 2 * TRING + 16);
constexpr int W1FRAG = 2 * 3 * 64 * 8;             // conv3x3 stride 2: [hi | lo][ky][lane][8]
constexpr int WSFRAG = 2 * 64 * 8;                 // shortcut: [hi | lo][lane][8]
}  // namespace down16

struct DownW {
  const uint16_t *w1, *wsc, *w2;
  const float *b1, *bsc, *b2;
};

#define MSF_LOFTR_DOWN16_WPE 0     // > 0: cap k_down16x's registers for this many waves per SIMD (with the 4-step queue the cap 4 spilled
                                   // 10 registers and still gained 7 %: two workgroups per CU; the 3-step queue needs no cap)
__global__ __launch_bounds__(64 * down16::WAVES) void k_down16x(const float* __restrict__ in, DownW dw, float* __restrict__ out,
                                                                int H, int W, int n_strips) {
  using namespace down16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xr = reinterpret_cast<bf16x8*>(lds);     // input ring: hi plane, lo plane IPLANE behind it
  bf16x8* tr = xr + 2 * IPLANE;                    // t ring: [hi | lo][cb]
  bf16x8* sr = tr + TRING;                         // sc ring
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);
  const int X0 = __builtin_amdgcn_readfirstlane((unit - img * n_strips) * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int Hin = 2 * H, Win = 2 * W, HWin = Hin * Win, HW = H * W;
  const float* inf = in + (long long)img * 8 * HWin;
  float* outf = out + (long long)img * 16 * HW;
  const int npairs = H / 2;
  const bool st1 = wave < 6;                       // wave-uniform: stage 1 (one job) or stage 2 (two jobs)
  const int jr = st1 ? wave / 3 : wave - 6;        // row of the pair this wave works on
  const int jq = st1 ? wave - 3 * jr : 0;          // stage 1: M tile

  // weight fragments: stage 1 waves hold the stride-2 convolution and the shortcut, stage 2 waves the 3x3 convolution
  bf16x8 wa[5], wb[5], wsh, wsl;
  f32x4 bias, bias_sc;
  if (st1) {
    const bf16x8* p1 = reinterpret_cast<const bf16x8*>(dw.w1);
    const bf16x8* ps = reinterpret_cast<const bf16x8*>(dw.wsc);
#pragma unroll
    for (int g = 0; g < 3; g++) { wa[g] = p1[g * 64 + lane]; wb[g] = p1[(3 + g) * 64 + lane]; }
    wa[3] = wa[4] = wb[3] = wb[4] = wa[0];
    wsh = ps[lane]; wsl = ps[64 + lane];
    bias = *reinterpret_cast<const f32x4*>(dw.b1 + 4 * kq);
    bias_sc = *reinterpret_cast<const f32x4*>(dw.bsc + 4 * kq);
  } else {
    const bf16x8* p2 = reinterpret_cast<const bf16x8*>(dw.w2);
#pragma unroll
    for (int g = 0; g < 5; g++) { wa[g] = p2[g * 64 + lane]; wb[g] = p2[(5 + g) * 64 + lane]; }
    wsh = wsl = wa[0];
    bias = *reinterpret_cast<const f32x4*>(dw.b2 + 4 * kq);
    bias_sc = bias;
  }
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds);
    for (int idx = tid; idx < LDS_BYTES / 4; idx += 64 * WAVES) z[idx] = 0u;
  }
  // loader (threads 0 .. NLOAD-1): thread (row r4 of the step's four, tile column lc) fetches one split pixel (2 x 16 bytes)
  const bool ld = tid < NLOAD;
  const int r4 = ld ? tid / INW : 0, lc = ld ? tid - r4 * INW : 0;
  const int lgx = 2 * X0 - 3 + lc;
  const bool colok = ld && lgx >= 0 && lgx < Win;
  const uint32_t lofs = colok ? (uint32_t)lgx : 0u;
  const int lslot = (lc & 1) ? IODD + (lc >> 1) : (lc >> 1);
  const bool ldwave = wave <= (NLOAD - 1) / 64;
#define MSF_DN_ISSUE(q_, n_)                                                                      \
  {                                                                                               \
    const int gy = 4 * (n_) + r4;                                                                 \
    const uint32_t so = (colok && gy < Hin) ? sx_off(gy, 0, 0, (int)lofs, Win, 1) : 0u;           \
    q_[0] = sx_ld(inf, so);                                                                       \
    q_[1] = sx_ld(inf, so + 16u * (uint32_t)Win);                                                 \
  }
#define MSF_DN_COMMIT(q_, n_)                                                                     \
  if (ld) {                                                                                       \
    const bool ok = colok && 4 * (n_) + r4 < Hin;                                                 \
    const u32x4v z4 = u32x4v{0u, 0u, 0u, 0u};                                                     \
    u32x4v* dst = reinterpret_cast<u32x4v*>(xr + (irow + r4 * IPX + lslot));                      \
    dst[0] = ok ? q_[0] : z4; dst[IPLANE] = ok ? q_[1] : z4;                                      \
  }
  int irow = 0;                                    // loader cursor: (4n mod 12) * IPX (a step's four rows never wrap inside)
  constexpr int IWRAP = IROWS * IPX;
  const bool edge = X0 == 0 || X0 + S == W;
  // ---- stage 1: out row Y = 2p + jr, t / sc slots 16 jq + i
  auto stage1 = [&](int p) {
    const int jt = 16 * jq + i;
    const int Y = 2 * p + jr;
    bf16x4* th4 = reinterpret_cast<bf16x4*>(tr);
    bf16x4* sh4 = reinterpret_cast<bf16x4*>(sr);
    const int os = 2 * ((kq >> 1) * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);   // channels 4 kq .. +3: block kq >> 1, half kq & 1
    if (p >= npairs) {                             // the row pair below the image: zero padding for stage 2
      if (jt < TW) {
        bf16x4 z;
        z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
        th4[os] = z; th4[os + 4 * TCB] = z;
      }
      return;
    }
    const int ja = jt < TW + 2 ? jt : TW + 1;      // lanes past the strip's columns compute nothing useful: keep their reads inside the row
    const int col = kq == 1 ? IODD + ja : kq == 0 ? ja : ja + 1;   // kx = kq (kq = 3: zero weights)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, asc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 3; g++) {
      // input row 2Y - 1 + g = 4p + 2 jr - 1 + g; ring rows are input rows modulo 12
      int rr = 4 * p + 2 * jr - 1 + g + IROWS;
      rr = rr - IROWS * ((rr * 2731) >> 15);
      const bf16x8* src = xr + (rr * IPX + col);
      const bf16x8 ph = src[0], pl = src[IPLANE];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ph, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], pl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ph, acc, 0, 0, 0);
      if (g == 1) {
        asc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsl, ph, asc, 0, 0, 0);
        asc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsh, pl, asc, 0, 0, 0);
        asc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsh, ph, asc, 0, 0, 0);
      }
    }
    acc += bias;
    asc += bias_sc;
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    if (edge) {
      const int gx = X0 - 1 + jt;
      if (gx < 0 || gx >= W) acc = f32x4{0.f, 0.f, 0.f, 0.f};   // stage 2 pads t with zeros
    }
    if (jt < TW) {
      bf16x4 vh, vl;
      split4(acc, vh, vl);
      th4[os] = vh; th4[os + 4 * TCB] = vl;
      split4(asc, vh, vl);
      sh4[os] = vh; sh4[os + 4 * TCB] = vl;
    }
  };
  // ---- stage 2: out row Y = 2p + jr, pixels 16 q + i (q = 0, 1): tap 2g + (kq >> 1) (the tenth has zero weights), block kq & 1
  auto stage2 = [&](int p) {
    if (p >= npairs) return;
    const int Y = 2 * p + jr;
    f32x4 acc[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 5; g++) {
        int t = 2 * g + (kq >> 1);
        t = t < 9 ? t : 8;
        const int ky = t / 3, kx = t - 3 * ky;
        const bf16x8* src = tr + ((kq & 1) * TCB + ((Y - 1 + ky) & (TROWS - 1)) * TPX + 16 * q + i + kx);
        const bf16x8 ah = src[0], al = src[2 * TCB];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ah, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], al, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ah, acc[q], 0, 0, 0);
      }
    }
    const bf16x4* sh4 = reinterpret_cast<const bf16x4*>(sr);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = 16 * q + i;
      const int rs = 2 * ((kq >> 1) * TCB + (Y & (TROWS - 1)) * TPX + j + 1) + (kq & 1);
      const bf16x4 a = sh4[rs], b = sh4[rs + 4 * TCB];
      f32x4 v = acc[q] + bias;
      v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
      v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      sx_st4(outf, Y, kq >> 1, X0 + j, kq & 1, W, 2, v);        // channels 4 kq .. +3: block kq >> 1, half kq & 1
    }
  };
#define MSF_DN_STEP(q_, n_)                                                                       \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_DN_COMMIT(q_, n_)                                                                       \
      MSF_DN_ISSUE(q_, (n_) + MSF_LOFTR_DOWN16_DEPTH)                                             \
    }                                                                                             \
    if (st1) {                                                                                    \
      const int p_ = (n_) - 1;                                                                    \
      if (p_ >= 0 && p_ <= npairs) stage1(p_);                                                    \
    } else {                                                                                      \
      const int p_ = (n_) - 3;                                                                    \
      if (p_ >= 0) stage2(p_);                                                                    \
    }                                                                                             \
    irow += 4 * IPX; irow = irow >= IWRAP ? irow - IWRAP : irow;                                  \
  }
  // the load queue is MSF_LOFTR_DOWN16_DEPTH steps deep, each step's registers named statically (a rotating queue would
  // make every step wait for the newest load)
  constexpr int kDepth = MSF_LOFTR_DOWN16_DEPTH;
  const int nsteps = ((npairs + 3 + kDepth - 1) / kDepth) * kDepth;
  auto run = [&](auto is_loader) {
    constexpr bool kLd = decltype(is_loader)::value;
    u32x4v q0[2], q1[2], q2[2];
    if (kLd) {
      MSF_DN_ISSUE(q0, 0)
      MSF_DN_ISSUE(q1, 1)
      MSF_DN_ISSUE(q2, 2)
    }
    for (int n = 0; n < nsteps; n += kDepth) {
      MSF_DN_STEP(q0, n)
      MSF_DN_STEP(q1, n + 1)
      MSF_DN_STEP(q2, n + 2)
    }
  };
  if (ldwave) run(std::true_type{});
  else run(std::false_type{});
#undef MSF_DN_ISSUE
#undef MSF_DN_COMMIT
#undef MSF_DN_STEP
}

// ------------------------------------------------------------------ streaming 16-channel BasicBlock (layer2, block 2)
// k_down16x's layout with a stride-1 first stage: a workgroup owns 32 output columns of one image and walks down two rows
// per step; ring X holds 12 rows of the block input (16 channels = 2 channel-block planes, hi | lo), stage A (waves 0-5,
// one (row, M tile) job each) makes t = relu(conv3x3(x)), stage B (waves 6-7, two jobs each) y = relu(conv3x3(t) + x)
// with the residual read from ring X (hi + lo) -- no halo rows, no separately fetched residual (1.95 -> 1.3 GB per step
// against the banded k_block16x).  K = 32 of an MFMA = two taps x two channel blocks (five groups, k_block16x's packing).
// Pipeline: rows 2n, 2n+1 of x arrive at step n, stage A makes pair n - 2, stage B pair n - 4; one barrier per step.
namespace strip16 {
constexpr int S = 32;
constexpr int XROWS = 12, XPX = 40;                // x ring: columns X0-2 .. X0+33 in slots 0 .. 35
constexpr int XW = S + 4;
constexpr int XCB = XROWS * XPX;                   // 480 = 30 x 16
constexpr int XRING = 4 * XCB;                     // [hi | lo][channel block]
constexpr int TROWS = 8, TPX = 36, TW = S + 2;     // t ring: four row pairs, columns X0-1 .. X0+32
constexpr int TCB = TROWS * TPX;
constexpr int TRING = 4 * TCB;
static_assert(XCB % 16 == 0 && TCB % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
constexpr int WAVES = 8;
constexpr int NLOAD = 2 * 2 * XW;                  // loader threads: (channel block, row of the pair, column)
constexpr int LDS_BYTES = 16 * (XRING + TRING + 16);
}  // namespace strip16

__global__ __launch_bounds__(64 * strip16::WAVES) void k_strip16x(const float* __restrict__ in, const uint16_t* __restrict__ wx1,
                                                                  const float* __restrict__ b1, const uint16_t* __restrict__ wx2,
                                                                  const float* __restrict__ b2, float* __restrict__ out, int H,
                                                                  int W, int n_strips) {
  using namespace strip16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xr = reinterpret_cast<bf16x8*>(lds);     // x ring: [hi cb0][hi cb1][lo cb0][lo cb1]
  bf16x8* tr = xr + XRING;                         // t ring, same order
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);
  const int X0 = __builtin_amdgcn_readfirstlane((unit - img * n_strips) * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int HW = H * W;
  const float* inf = in + (long long)img * 16 * HW;
  float* outf = out + (long long)img * 16 * HW;
  const int npairs = H / 2;
  const bool stA = wave < 6;
  const int jr = stA ? wave / 3 : wave - 6;        // row of the pair this wave works on
  const int jq = stA ? wave - 3 * jr : 0;          // stage A: M tile

  bf16x8 wa[5], wb[5];                             // this wave's stage: fragments hi (wa) and lo (wb) of its convolution
  {
    const bf16x8* pw = reinterpret_cast<const bf16x8*>(stA ? wx1 : wx2);
#pragma unroll
    for (int g = 0; g < 5; g++) { wa[g] = pw[g * 64 + lane]; wb[g] = pw[(5 + g) * 64 + lane]; }
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>((stA ? b1 : b2) + 4 * kq);   // lane = channels 4 kq .. +3 of one pixel
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds);
    for (int idx = tid; idx < LDS_BYTES / 4; idx += 64 * WAVES) z[idx] = 0u;
  }
  // loader (threads 0 .. NLOAD-1): thread (channel block lcb, row lr of the pair, ring column lc) fetches one split pixel block (2 x 16 bytes)
  const bool ld = tid < NLOAD;
  const int lcb = ld ? tid / (2 * XW) : 0, lrm = ld ? tid - lcb * 2 * XW : 0;
  const int lr = lrm / XW, lc = lrm - lr * XW;
  const int lgx = X0 - 2 + lc;
  const bool colok = ld && lgx >= 0 && lgx < W;
  const uint32_t lofs = colok ? (uint32_t)lgx : 0u;
  const bool ldwave = wave <= (NLOAD - 1) / 64;
#define MSF_S16_ISSUE(q_, n_)                                                                     \
  {                                                                                               \
    const int gy = 2 * (n_) + lr;                                                                 \
    const uint32_t so = (colok && gy < H) ? sx_off(gy, 0, lcb, (int)lofs, W, 2) : 0u;             \
    q_[0] = sx_ld(inf, so);                                                                       \
    q_[1] = sx_ld(inf, so + 32u * (uint32_t)W);                                                   \
  }
#define MSF_S16_COMMIT(q_, n_)                                                                    \
  if (ld) {                                                                                       \
    const bool ok = colok && 2 * (n_) + lr < H;                                                   \
    const u32x4v z4 = u32x4v{0u, 0u, 0u, 0u};                                                     \
    u32x4v* dst = reinterpret_cast<u32x4v*>(xr + (lcb * XCB + crow + lr * XPX + lc));             \
    dst[0] = ok ? q_[0] : z4; dst[2 * XCB] = ok ? q_[1] : z4;                                     \
  }
  int crow = 0;                                    // loader cursor: (2n mod 12) * XPX
  constexpr int XWRAP = XROWS * XPX;
  const bool edge = X0 == 0 || X0 + S == W;
  // this lane's K block of MFMA group g: tap 2g + (kq >> 1) (the tenth has zero weights), channel block kq & 1
  int tky[5], tkx[5];
#pragma unroll
  for (int g = 0; g < 5; g++) {
    int t = 2 * g + (kq >> 1);
    t = t < 9 ? t : 8;
    tky[g] = t / 3;
    tkx[g] = t - 3 * tky[g];
  }
  // ---- stage A: t row Y = 2p + jr, t slots 16 jq + i (columns X0 - 1 + slot)
  auto stageA = [&](int p) {
    const int jt = 16 * jq + i;
    const int Y = 2 * p + jr;
    bf16x4* th4 = reinterpret_cast<bf16x4*>(tr);
    const int os = 2 * ((kq >> 1) * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);
    if (p >= npairs) {                             // the row pair below the image: zero padding for stage B
      if (jt < TW) {
        bf16x4 z;
        z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
        th4[os] = z; th4[os + 4 * TCB] = z;
      }
      return;
    }
    const int ja = jt < TW + 2 ? jt : TW + 1;      // lanes past the strip's columns: keep their reads inside the row
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 5; g++) {
      int rr = Y - 1 + tky[g] + XROWS;             // x rows are ring rows modulo 12
      rr = rr - XROWS * ((rr * 2731) >> 15);
      const bf16x8* src = xr + ((kq & 1) * XCB + rr * XPX + ja + tkx[g]);
      const bf16x8 ph = src[0], pl = src[2 * XCB];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ph, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], pl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ph, acc, 0, 0, 0);
    }
    acc += bias;
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    if (edge) {
      const int gx = X0 - 1 + jt;
      if (gx < 0 || gx >= W) acc = f32x4{0.f, 0.f, 0.f, 0.f};   // stage B pads t with zeros
    }
    if (jt < TW) {
      bf16x4 vh, vl;
      split4(acc, vh, vl);
      th4[os] = vh; th4[os + 4 * TCB] = vl;
    }
  };
  // ---- stage B: out row Y = 2p + jr, pixels 16 q + i (q = 0, 1)
  auto stageB = [&](int p) {
    if (p >= npairs) return;
    const int Y = 2 * p + jr;
    f32x4 acc[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 5; g++) {
        const bf16x8* src = tr + ((kq & 1) * TCB + ((Y - 1 + tky[g]) & (TROWS - 1)) * TPX + 16 * q + i + tkx[g]);
        const bf16x8 ah = src[0], al = src[2 * TCB];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ah, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], al, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ah, acc[q], 0, 0, 0);
      }
    }
    int ry = Y + XROWS;                            // residual: x row Y, slot j + 2
    ry = ry - XROWS * ((ry * 2731) >> 15);
    const bf16x4* xh4 = reinterpret_cast<const bf16x4*>(xr);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int j = 16 * q + i;
      const int rs = 2 * ((kq >> 1) * XCB + ry * XPX + j + 2) + (kq & 1);
      const bf16x4 a = xh4[rs], b = xh4[rs + 4 * XCB];
      f32x4 v = acc[q] + bias;
      v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
      v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      sx_st4(outf, Y, kq >> 1, X0 + j, kq & 1, W, 2, v);
    }
  };
#define MSF_S16_STEP(q_, n_)                                                                      \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_S16_COMMIT(q_, n_)                                                                      \
      MSF_S16_ISSUE(q_, (n_) + MSF_LOFTR_STRIP16_DEPTH)                                                                 \
    }                                                                                             \
    if (stA) {                                                                                    \
      const int p_ = (n_) - 2;                                                                    \
      if (p_ >= 0 && p_ <= npairs) stageA(p_);                                                    \
    } else {                                                                                      \
      const int p_ = (n_) - 4;                                                                    \
      if (p_ >= 0) stageB(p_);                                                                    \
    }                                                                                             \
    crow += 2 * XPX; crow = crow >= XWRAP ? crow - XWRAP : crow;                                  \
  }
  // the load queue is MSF_LOFTR_STRIP16_DEPTH steps deep (2, 3 or 4), each step's registers named statically; the loop is unrolled by the depth
  constexpr int kQDepth = MSF_LOFTR_STRIP16_DEPTH;
  static_assert(kQDepth >= 2 && kQDepth <= 4, "load queue depth");
  const int nsteps = ((npairs + 4 + kQDepth - 1) / kQDepth) * kQDepth;
  auto run = [&](auto is_loader) {
    constexpr bool kLd = decltype(is_loader)::value;
    u32x4v q0[2]; u32x4v q1[2];
    if (kLd) {
      MSF_S16_ISSUE(q0, 0)
      MSF_S16_ISSUE(q1, 1)
    }
    for (int n = 0; n < nsteps; n += kQDepth) {
      MSF_S16_STEP(q0, n)
      MSF_S16_STEP(q1, n + 1)
    }
  };
  if (ldwave) run(std::true_type{});
  else run(std::false_type{});
#undef MSF_S16_ISSUE
#undef MSF_S16_COMMIT
#undef MSF_S16_STEP
}

// ------------------------------------------------------------------ streaming 32-channel BasicBlock (layer3 / layer4, block 2)
// k_strip16x for 32 channels: a workgroup owns 16 output columns of one image and walks down two rows per step; ring X
// holds 12 rows of the block input (4 channel-block planes, hi | lo), ring T 8 rows of t.  K = 32 of an MFMA = the 4
// channel blocks of one tap (9 groups, k_convx's packing); the 32 output channels are two MFMA tiles, and a wave holds
// the fragments of ONE (convolution, output tile): 72 VGPRs.  Stage A (waves 0-3: (tile, row), two M-tile jobs each)
// makes t = relu(conv3x3(x)), stage B (waves 4-7: (tile, row), one job each) y = relu(conv3x3(t) + x), residual from
// ring X.  The image loader lives in the stage-B waves.  Replaces two k_convx passes (x, t and the residual through
// HBM with 1.4 x halos) by one read and one write.
namespace strip32 {
constexpr int S = 16;
constexpr int XROWS = 12, XPX = 24, XW = S + 4;    // x ring: columns X0-2 .. X0+17 in slots 0 .. 19
constexpr int XCB = XROWS * XPX;                   // 288 = 18 x 16
constexpr int XRING = 8 * XCB;                     // [hi | lo][4 channel blocks]
constexpr int TROWS = 8, TPX = 20, TW = S + 2;     // t ring: columns X0-1 .. X0+16
constexpr int TCB = TROWS * TPX;                   // 160 = 10 x 16
constexpr int TRING = 8 * TCB;
static_assert(XCB % 16 == 0 && TCB % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
constexpr int WAVES = 8;
constexpr int NLOAD = 4 * 2 * XW;                  // loader threads: (channel block, row of the pair, column)
constexpr int LDS_BYTES = 16 * (XRING + TRING + 16);
}  // namespace strip32

#define MSF_LOFTR_STRIP32_WPE 4   // k_strip32x capped at 128 registers (with the 2-step queue 9 are spilled): two workgroups per CU instead of one
#define MSF_STRIP32_ATTR __attribute__((amdgpu_waves_per_eu(MSF_LOFTR_STRIP32_WPE, MSF_LOFTR_STRIP32_WPE)))
__global__ __launch_bounds__(64 * strip32::WAVES) MSF_STRIP32_ATTR void k_strip32x(const float* __restrict__ in, const uint16_t* __restrict__ wx1,
                                                                  const float* __restrict__ b1, const uint16_t* __restrict__ wx2,
                                                                  const float* __restrict__ b2, float* __restrict__ out, int H,
                                                                  int W, int n_strips) {
  using namespace strip32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xr = reinterpret_cast<bf16x8*>(lds);     // x ring: [hi cb0..3][lo cb0..3]
  bf16x8* tr = xr + XRING;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);
  const int X0 = __builtin_amdgcn_readfirstlane((unit - img * n_strips) * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int HW = H * W;
  const float* inf = in + (long long)img * 32 * HW;
  float* outf = out + (long long)img * 32 * HW;
  const int npairs = H / 2;
  const bool stA = wave < 4;
  const int nt = wave & 1, jr = (wave >> 1) & 1;   // output-channel tile, row of the pair

  bf16x8 wa[9], wb[9];                             // fragments hi / lo of (this wave's convolution, tile nt): k_convx's [g][nt][hl][lane]
  {
    const bf16x8* pw = reinterpret_cast<const bf16x8*>(stA ? wx1 : wx2);
#pragma unroll
    for (int g = 0; g < 9; g++) { wa[g] = pw[((g * 2 + nt) * 2 + 0) * 64 + lane]; wb[g] = pw[((g * 2 + nt) * 2 + 1) * 64 + lane]; }
  }
  const f32x4 bias = *reinterpret_cast<const f32x4*>((stA ? b1 : b2) + 16 * nt + 4 * kq);   // lane = channels 16 nt + 4 kq .. +3
  const int cbp = 2 * nt + (kq >> 1);              // channel-block plane of those four channels (half kq & 1)
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds);
    for (int idx = tid; idx < LDS_BYTES / 4; idx += 64 * WAVES) z[idx] = 0u;
  }
  // loader (the last NLOAD threads): thread (channel block lcb, row lr of the pair, ring column lc) fetches one split pixel block (2 x 16 bytes)
#ifndef MSF_S32_LOADERS_FIRST
  const int ltid = tid - (64 * WAVES - NLOAD);     // (r05: in the stage-A waves instead -- no stores there, see k_strip8x -- 362 -> 654 us:
                                                   // those waves carry two M-tile jobs each, the loader on top makes them the step)
#else
  const int ltid = tid < NLOAD ? tid : -1;
#endif
  const bool ld = ltid >= 0;
  const int lcb = ld ? ltid / (2 * XW) : 0, lrm = ld ? ltid - lcb * 2 * XW : 0;
  const int lr = lrm / XW, lc = lrm - lr * XW;
  const int lgx = X0 - 2 + lc;
  const bool colok = ld && lgx >= 0 && lgx < W;
  const uint32_t lofs = colok ? (uint32_t)lgx : 0u;
#ifndef MSF_S32_LOADERS_FIRST
  const bool ldwave = wave >= (64 * WAVES - NLOAD) / 64;
#else
  const bool ldwave = wave <= (NLOAD - 1) / 64;
#endif
#define MSF_S32_ISSUE(q_, n_)                                                                     \
  {                                                                                               \
    const int gy = 2 * (n_) + lr;                                                                 \
    const uint32_t so = (colok && gy < H) ? sx_off(gy, 0, lcb, (int)lofs, W, 4) : 0u;             \
    q_[0] = sx_ld(inf, so);                                                                       \
    q_[1] = sx_ld(inf, so + 64u * (uint32_t)W);                                                   \
  }
#define MSF_S32_COMMIT(q_, n_)                                                                    \
  if (ld) {                                                                                       \
    const bool ok = colok && 2 * (n_) + lr < H;                                                   \
    const u32x4v z4 = u32x4v{0u, 0u, 0u, 0u};                                                     \
    u32x4v* dst = reinterpret_cast<u32x4v*>(xr + (lcb * XCB + crow + lr * XPX + lc));             \
    dst[0] = ok ? q_[0] : z4; dst[4 * XCB] = ok ? q_[1] : z4;                                     \
  }
  int crow = 0;                                    // loader cursor: (2n mod 12) * XPX
  constexpr int XWRAP = XROWS * XPX;
  const bool edge = X0 == 0 || X0 + S >= W;        // the last strip may be partial (W = 40: columns 32 .. 39)
  // ---- stage A: t row Y = 2p + jr, channels of tile nt, t slots 16 q + i (q = 0, 1; columns X0 - 1 + slot, 18 valid)
  auto stageA = [&](int p) {
    const int Y = 2 * p + jr;
    bf16x4* th4 = reinterpret_cast<bf16x4*>(tr);
    if (p >= npairs) {                             // the row pair below the image: zero padding for stage B
      bf16x4 z;
      z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int jt = 16 * q + i;
        if (jt < TW) {
          const int os = 2 * (cbp * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);
          th4[os] = z; th4[os + 8 * TCB] = z;
        }
      }
      return;
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int g = 0; g < 9; g++) {
      const int ky = g / 3, kx = g - 3 * ky;
      int rr = Y - 1 + ky + XROWS;                 // x rows are ring rows modulo 12
      rr = rr - XROWS * ((rr * 2731) >> 15);
      const bf16x8* row = xr + (kq * XCB + rr * XPX + kx);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int jt = 16 * q + i;
        const bf16x8* src = row + (jt < TW + 2 ? jt : TW + 1);   // lanes past the strip's columns stay inside the row
        const bf16x8 ph = src[0], pl = src[4 * XCB];
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ph, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], pl, acc[q], 0, 0, 0);
        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ph, acc[q], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int jt = 16 * q + i;
      f32x4 v = acc[q] + bias;
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (edge) {
        const int gx = X0 - 1 + jt;
        if (gx < 0 || gx >= W) v = f32x4{0.f, 0.f, 0.f, 0.f};    // stage B pads t with zeros
      }
      if (jt < TW) {
        bf16x4 vh, vl;
        split4(v, vh, vl);
        const int os = 2 * (cbp * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);
        th4[os] = vh; th4[os + 8 * TCB] = vl;
      }
    }
  };
  // ---- stage B: out row Y = 2p + jr, channels of tile nt, pixels X0 + i
  auto stageB = [&](int p) {
    if (p >= npairs) return;
    const int Y = 2 * p + jr;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 9; g++) {
      const int ky = g / 3, kx = g - 3 * ky;
      const bf16x8* src = tr + (kq * TCB + ((Y - 1 + ky) & (TROWS - 1)) * TPX + i + kx);
      const bf16x8 ah = src[0], al = src[4 * TCB];
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g], ah, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], al, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g], ah, acc, 0, 0, 0);
    }
    int ry = Y + XROWS;                            // residual: x row Y, slot i + 2
    ry = ry - XROWS * ((ry * 2731) >> 15);
    const bf16x4* xh4 = reinterpret_cast<const bf16x4*>(xr);
    const int rs = 2 * (cbp * XCB + ry * XPX + i + 2) + (kq & 1);
    const bf16x4 a = xh4[rs], b = xh4[rs + 8 * XCB];
    f32x4 v = acc + bias;
    v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
    v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    if (X0 + i < W) {
      const uint32_t oo = 4u * (uint32_t)(((16 * nt + 4 * kq) * H + Y) * W + X0 + i);
      char* ob = reinterpret_cast<char*>(outf);
      *reinterpret_cast<float*>(ob + oo) = v.x;
      *reinterpret_cast<float*>(ob + (oo + 4u * (uint32_t)HW)) = v.y;
      *reinterpret_cast<float*>(ob + (oo + 8u * (uint32_t)HW)) = v.z;
      *reinterpret_cast<float*>(ob + (oo + 12u * (uint32_t)HW)) = v.w;
    }
  };
#define MSF_S32_STEP(q_, n_)                                                                      \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_S32_COMMIT(q_, n_)                                                                      \
      MSF_S32_ISSUE(q_, (n_) + MSF_LOFTR_STRIP32_DEPTH)                                                                 \
    }                                                                                             \
    if (stA) {                                                                                    \
      const int p_ = (n_) - 2;                                                                    \
      if (p_ >= 0 && p_ <= npairs) stageA(p_);                                                    \
    } else {                                                                                      \
      const int p_ = (n_) - 4;                                                                    \
      if (p_ >= 0) stageB(p_);                                                                    \
    }                                                                                             \
    crow += 2 * XPX; crow = crow >= XWRAP ? crow - XWRAP : crow;                                  \
  }
  // the load queue is MSF_LOFTR_STRIP32_DEPTH steps deep (2, 3 or 4), each step's registers named statically; the loop is unrolled by the depth
  constexpr int kQDepth = MSF_LOFTR_STRIP32_DEPTH;
  static_assert(kQDepth >= 2 && kQDepth <= 4, "load queue depth");
  const int nsteps = ((npairs + 4 + kQDepth - 1) / kQDepth) * kQDepth;
  auto run = [&](auto is_loader) {
    constexpr bool kLd = decltype(is_loader)::value;
    u32x4v q0[2]; u32x4v q1[2];
    if (kLd) {
      MSF_S32_ISSUE(q0, 0)
      MSF_S32_ISSUE(q1, 1)
    }
    for (int n = 0; n < nsteps; n += kQDepth) {
      MSF_S32_STEP(q0, n)
      MSF_S32_STEP(q1, n + 1)
    }
  };
  if (ldwave) run(std::true_type{});
  else run(std::false_type{});
#undef MSF_S32_ISSUE
#undef MSF_S32_COMMIT
#undef MSF_S32_STEP
}

// ------------------------------------------------------------------ streaming down-sampling BasicBlock 16 -> 32 (layer3, block 1)
// k_down16x for 16 input / 32 output channels with k_strip32x's wave layout: a workgroup owns 16 output columns (37 input
// columns, de-interleaved even | odd, 2 channel-block planes) and walks down two output rows (four input rows) per
// step.  Stage 1 (waves 0-3: (output tile, row), two M-tile jobs each): t = relu(conv3x3 stride 2) with K = 32 = two
// taps x two channel blocks (5 groups, k_convx2<16>'s packing), and the shortcut sc = conv1x1 stride 2 on the centre
// tap's fragment (group 2, K blocks 0 and 1); stage 2 (waves 4-7): u = relu(conv3x3(t) + sc) with K = 32 = 4 channel
// blocks of one tap (9 groups, k_convx's packing).  Replaces k_convx2<16> + k_convx<32, true>.
namespace down32 {
constexpr int S = 16, TW = S + 2;
constexpr int IROWS = 12, IPX = 40, IODD = 20;     // input ring: even columns in slots 0 .. 18, odd columns in 20 .. 37
constexpr int INW = 2 * TW + 1;                    // 37 input columns: 2 X0 - 3 .. 2 X0 + 33
constexpr int ICB = IROWS * IPX;                   // 480 = 30 x 16
constexpr int IRING = 4 * ICB;                     // [hi | lo][2 channel blocks]
constexpr int TROWS = 8, TPX = 20;
constexpr int TCB = TROWS * TPX;                   // 160
constexpr int TRING = 8 * TCB;                     // [hi | lo][4 channel blocks]
static_assert(ICB % 16 == 0 && TCB % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
constexpr int WAVES = 8;
constexpr int NLOAD = 2 * 4 * INW;                 // loader threads: (channel block, row of the step's four, column)
constexpr int LDS_BYTES = 16 * (IRING + 2 * TRING + 16);
}  // namespace down32

#ifndef MSF_LOFTR_DOWN32_WPE
#define MSF_LOFTR_DOWN32_WPE 4     // registers capped for this many waves per SIMD (4 = 128 registers = two workgroups per CU)
#endif
__global__ __launch_bounds__(64 * down32::WAVES) __attribute__((amdgpu_waves_per_eu(MSF_LOFTR_DOWN32_WPE, MSF_LOFTR_DOWN32_WPE))) void k_down32x(const float* __restrict__ in, DownW dw, float* __restrict__ out,
                                                                int H, int W, int n_strips) {
  using namespace down32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xr = reinterpret_cast<bf16x8*>(lds);     // input ring: [hi cb0][hi cb1][lo cb0][lo cb1]
  bf16x8* tr = xr + IRING;                         // t ring: [hi cb0..3][lo cb0..3]
  bf16x8* sr = tr + TRING;                         // sc ring
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = __builtin_amdgcn_readfirstlane(unit / n_strips);
  const int X0 = __builtin_amdgcn_readfirstlane((unit - img * n_strips) * S);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int Hin = 2 * H, Win = 2 * W, HWin = Hin * Win, HW = H * W;
  const float* inf = in + (long long)img * 16 * HWin;
  float* outf = out + (long long)img * 32 * HW;
  const int npairs = H / 2;
  const bool st1 = wave < 4;
  const int nt = wave & 1, jr = (wave >> 1) & 1;   // output-channel tile, row of the pair
  const int cbp = 2 * nt + (kq >> 1);              // channel-block plane of this lane's channels 16 nt + 4 kq .. +3 (half kq & 1)
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds);
    for (int idx = tid; idx < LDS_BYTES / 4; idx += 64 * WAVES) z[idx] = 0u;
  }
  // loader (threads 0 .. NLOAD-1): thread (channel block lcb, row r4 of the step's four, tile column lc) fetches one split pixel block (2 x 16 bytes)
  const bool ld = tid < NLOAD;
  const int lcb = ld ? tid / (4 * INW) : 0, lrm = ld ? tid - lcb * 4 * INW : 0;
  const int r4 = lrm / INW, lc = lrm - r4 * INW;
  const int lgx = 2 * X0 - 3 + lc;
  const bool colok = ld && lgx >= 0 && lgx < Win;
  const uint32_t lofs = colok ? (uint32_t)lgx : 0u;
  const int lslot = (lc & 1) ? IODD + (lc >> 1) : (lc >> 1);
  const bool ldwave = wave <= (NLOAD - 1) / 64;
  static_assert((NLOAD - 1) / 64 == 4, "waves 0-3 (stage 1) are loader waves, wave 4 (stage 2) partly");
#define MSF_D32_ISSUE(q_, n_)                                                                     \
  {                                                                                               \
    const int gy = 4 * (n_) + r4;                                                                 \
    const uint32_t so = (colok && gy < Hin) ? sx_off(gy, 0, lcb, (int)lofs, Win, 2) : 0u;         \
    q_[0] = sx_ld(inf, so);                                                                       \
    q_[1] = sx_ld(inf, so + 32u * (uint32_t)Win);                                                 \
  }
#define MSF_D32_COMMIT(q_, n_)                                                                    \
  if (ld) {                                                                                       \
    const bool ok = colok && 4 * (n_) + r4 < Hin;                                                 \
    const u32x4v z4 = u32x4v{0u, 0u, 0u, 0u};                                                     \
    u32x4v* dst = reinterpret_cast<u32x4v*>(xr + (lcb * ICB + irow + r4 * IPX + lslot));          \
    dst[0] = ok ? q_[0] : z4; dst[2 * ICB] = ok ? q_[1] : z4;                                     \
  }
  constexpr int IWRAP = IROWS * IPX;
  const bool edge = X0 == 0 || X0 + S >= W;
  constexpr int kQDepth = MSF_LOFTR_DOWN32_DEPTH;
  static_assert(kQDepth == 2, "the step macro is written out for a two-step queue");
  const int nsteps = ((npairs + 3 + kQDepth - 1) / kQDepth) * kQDepth;
  // r05: one copy of the step loop per (loader?, stage) -- a wave then holds only ITS stage's weight fragments (stage 1:
  // five groups + the shortcut = 48 registers, stage 2: nine groups = 72; r04: 80 in every wave) and, with a two-step
  // queue, the kernel fits 128 registers: two workgroups per CU instead of one (226 registers)
  auto run = [&](auto is_loader, auto is_st1) {
    constexpr bool kLd = decltype(is_loader)::value, ST1 = decltype(is_st1)::value;
    constexpr int NG = ST1 ? 5 : 9;
    bf16x8 wa[NG], wb[NG], wsh, wsl;
    {
      const bf16x8* pw = reinterpret_cast<const bf16x8*>(ST1 ? dw.w1 : dw.w2);
#pragma unroll
      for (int g = 0; g < NG; g++) {
        wa[g] = pw[((g * 2 + nt) * 2 + 0) * 64 + lane];
        wb[g] = pw[((g * 2 + nt) * 2 + 1) * 64 + lane];
      }
      const bf16x8* ps = reinterpret_cast<const bf16x8*>(dw.wsc);
      wsh = ps[(nt * 2 + 0) * 64 + lane];            // (stage 1 only: dead in the stage-2 copies)
      wsl = ps[(nt * 2 + 1) * 64 + lane];
    }
    const f32x4 bias = *reinterpret_cast<const f32x4*>((ST1 ? dw.b1 : dw.b2) + 16 * nt + 4 * kq);
    const f32x4 bias_sc = *reinterpret_cast<const f32x4*>(dw.bsc + 16 * nt + 4 * kq);
    int irow = 0;                                    // loader cursor: (4n mod 12) * IPX
    // stage 1: this lane's K block of group g: tap 2g + (kq >> 1) (the tenth has zero weights), channel block kq & 1
    int s1off[5];
#pragma unroll
    for (int g = 0; g < 5; g++) {
      int t = 2 * g + (kq >> 1);
      t = t < 9 ? t : 8;
      const int ky = t / 3, kx = t - 3 * ky;
      s1off[g] = ky * 256 + (kx == 1 ? IODD : kx == 0 ? 0 : 1);     // ky in bits 8.., column offset below
    }
    auto stage1 = [&](int p) {
      const int Y = 2 * p + jr;
      bf16x4* th4 = reinterpret_cast<bf16x4*>(tr);
      bf16x4* sh4 = reinterpret_cast<bf16x4*>(sr);
      if (p >= npairs) {
        bf16x4 z;
        z[0] = z[1] = z[2] = z[3] = (__bf16)0.f;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int jt = 16 * q + i;
          if (jt < TW) {
            const int os = 2 * (cbp * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);
            th4[os] = z; th4[os + 8 * TCB] = z;
          }
        }
        return;
      }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      f32x4 asc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int g = 0; g < 5; g++) {
        const int ky = s1off[g] >> 8, co = s1off[g] & 255;
        int rr = 4 * p + 2 * jr - 1 + ky + IROWS;    // input row 2Y - 1 + ky, ring rows modulo 12
        rr = rr - IROWS * ((rr * 2731) >> 15);
        const bf16x8* row = xr + ((kq & 1) * ICB + rr * IPX + co);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int jt = 16 * q + i;
          const bf16x8* src = row + (jt < TW ? jt : TW);           // lanes past the strip's columns stay inside the row
          const bf16x8 ph = src[0], pl = src[2 * ICB];
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g < NG ? g : 0], ph, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g < NG ? g : 0], pl, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g < NG ? g : 0], ph, acc[q], 0, 0, 0);
          if (g == 2) {                                             // the centre tap: K blocks 0, 1 of this group
            asc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsl, ph, asc[q], 0, 0, 0);
            asc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsh, pl, asc[q], 0, 0, 0);
            asc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wsh, ph, asc[q], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int jt = 16 * q + i;
        f32x4 v = acc[q] + bias, vs = asc[q] + bias_sc;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (edge) {
          const int gx = X0 - 1 + jt;
          if (gx < 0 || gx >= W) v = f32x4{0.f, 0.f, 0.f, 0.f};    // stage 2 pads t with zeros
        }
        if (jt < TW) {
          bf16x4 vh, vl;
          const int os = 2 * (cbp * TCB + (Y & (TROWS - 1)) * TPX + jt) + (kq & 1);
          split4(v, vh, vl);
          th4[os] = vh; th4[os + 8 * TCB] = vl;
          split4(vs, vh, vl);
          sh4[os] = vh; sh4[os + 8 * TCB] = vl;
        }
      }
    };
    auto stage2 = [&](int p) {
      if (p >= npairs) return;
      const int Y = 2 * p + jr;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 9; g++) {
        const int ky = g / 3, kx = g - 3 * ky;
        const bf16x8* src = tr + (kq * TCB + ((Y - 1 + ky) & (TROWS - 1)) * TPX + i + kx);
        const bf16x8 ah = src[0], al = src[4 * TCB];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[g < NG ? g : 0], ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g < NG ? g : 0], al, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[g < NG ? g : 0], ah, acc, 0, 0, 0);
      }
      const bf16x4* sh4 = reinterpret_cast<const bf16x4*>(sr);
      const int rs = 2 * (cbp * TCB + (Y & (TROWS - 1)) * TPX + i + 1) + (kq & 1);
      const bf16x4 a = sh4[rs], b = sh4[rs + 8 * TCB];
      f32x4 v = acc + bias;
      v.x += (float)a[0] + (float)b[0]; v.y += (float)a[1] + (float)b[1];
      v.z += (float)a[2] + (float)b[2]; v.w += (float)a[3] + (float)b[3];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      if (X0 + i < W) sx_st4(outf, Y, cbp, X0 + i, kq & 1, W, 4, v);   // channels 16 nt + 4 kq .. +3: block cbp, half kq & 1
    };
#define MSF_D32_STEP(q_, n_)                                                                      \
  {                                                                                               \
    __syncthreads();                                                                              \
    if (kLd) {                                                                                    \
      MSF_D32_COMMIT(q_, n_)                                                                      \
      MSF_D32_ISSUE(q_, (n_) + kQDepth)                                                           \
    }                                                                                             \
    if (ST1) {                                                                                    \
      const int p_ = (n_) - 1;                                                                    \
      if (p_ >= 0 && p_ <= npairs) stage1(p_);                                                    \
    } else {                                                                                      \
      const int p_ = (n_) - 3;                                                                    \
      if (p_ >= 0) stage2(p_);                                                                    \
    }                                                                                             \
    irow += 4 * IPX; irow = irow >= IWRAP ? irow - IWRAP : irow;                                  \
  }
    u32x4v q0[2]; u32x4v q1[2];
    if (kLd) {
      MSF_D32_ISSUE(q0, 0)
      MSF_D32_ISSUE(q1, 1)
    }
    for (int n = 0; n < nsteps; n += kQDepth) {
      MSF_D32_STEP(q0, n)
      MSF_D32_STEP(q1, n + 1)
    }
#undef MSF_D32_STEP
  };
  if (st1) run(std::true_type{}, std::true_type{});                  // waves 0-3: stage 1, all of them loaders
  else if (ldwave) run(std::true_type{}, std::false_type{});         // wave 4: stage 2 + the loader's last 40 threads
  else run(std::false_type{}, std::false_type{});
#undef MSF_D32_ISSUE
#undef MSF_D32_COMMIT
}

// ------------------------------------------------------------------ fused BasicBlock, 16 channels, split-bf16 MFMA
// k_block16's tiling (bands of 8 rows, x tiles of 32 columns, wave = (M tile, row half), conv2 one tile behind conv1)
// with the arithmetic and LDS layout of k_block8x / k_convx: planes [hi | lo][channel block of 8][row][pixel] x 16 B.
// K = 32 of one MFMA = two taps x two channel blocks (lane group kq: tap 2g + (kq >> 1), channel block kq & 1; five
// groups, the tenth tap has zero weights).  conv1 runs transposed (a lane holds 4 consecutive channels of one t pixel:
// 8-byte stores), conv2 as k_block16.  Both convolutions' fragments (80 VGPRs) stay in registers.
namespace blk16x {
constexpr int R = 8, TW = 32;
constexpr int XH = R + 4;                          // x rows oy0-2 .. oy0+9
constexpr int XP = 36;                             // x row pitch in pixels (columns 32k-1 .. 32k+32 in slots 0 .. 33)
constexpr int XCB = XH * XP;                       // 432 = 27 x 16: channel-block planes on disjoint banks
constexpr int TROWS = R + 2;                       // t rows oy0-1 .. oy0+8
constexpr int TP = 72;                             // two 32-pixel segments, halo pixels 64 / 65, zero pixel 66
constexpr int TCB = TROWS * TP;                    // 720 = 45 x 16
static_assert(XCB % 16 == 0 && TCB % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
constexpr int G = 5;
constexpr int LDS_BYTES = 16 * (4 * XCB + 4 * TCB);   // 73 728: two workgroups per CU
constexpr int WFRAG = 2 * G * 64 * 8;              // bf16 elements of one convolution's fragments [hi | lo][g][lane][8]
}  // namespace blk16x

__global__ __launch_bounds__(256, 2) void k_block16x(const float* __restrict__ in, const uint16_t* __restrict__ wx1,
                                                     const float* __restrict__ b1, const uint16_t* __restrict__ wx2,
                                                     const float* __restrict__ b2, float* __restrict__ out, int H, int W,
                                                     int n_bands) {
  using namespace blk16x;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xh = reinterpret_cast<bf16x8*>(lds);
  bf16x8* xl = xh + 2 * XCB;
  bf16x8* th = xl + 2 * XCB;
  bf16x8* tl = th + 2 * TCB;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, band) order
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int m = wave & 1, hf = wave >> 1;           // M tile (16 pixels of the 32), row half
  const long long HW = (long long)H * W;
  const float* inf = in + (long long)img * 16 * HW;
  float* outf = out + (long long)img * 16 * HW;

  bf16x8 w1h[G], w1l[G], w2h[G], w2l[G];
  {
    const bf16x8* p1 = reinterpret_cast<const bf16x8*>(wx1);
    const bf16x8* p2 = reinterpret_cast<const bf16x8*>(wx2);
#pragma unroll
    for (int g = 0; g < G; g++) {
      w1h[g] = p1[g * 64 + lane]; w1l[g] = p1[(G + g) * 64 + lane];
      w2h[g] = p2[g * 64 + lane]; w2l[g] = p2[(G + g) * 64 + lane];
    }
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b1 + 4 * kq);          // conv1: lane = channels 4 kq .. +3 of one t pixel
  const float bias2 = b2[i];                                                 // conv2: lane = channel i, 4 pixels
  // halo pixels and the zero pixel of every t row, every plane
  if (tid < 4 * TROWS) {
    bf16x8* pl = th + (tid / TROWS) * TCB;          // th[cb 0], th[cb 1], tl[cb 0], tl[cb 1] are contiguous
    const int r = tid % TROWS;
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; j++) z[j] = (__bf16)0.f;
    pl[r * TP + 2 * TW] = z; pl[r * TP + 2 * TW + 1] = z; pl[r * TP + 2 * TW + 2] = z;
  }

  // x staging: item = 8 channel dwords of one pixel.  Items 0..2 of a thread: (channel block, row, column 1 + (tid & 31))
  // with (block, row) = (tid >> 5) + 8u; item 3 (threads 0..47): the halo columns 0 and 33 of (block, row) = tid >> 1.
  constexpr int kNoRow = -(1 << 30);
  float pre[4][8];
  int goff[4], lslot[4];
#pragma unroll
  for (int u = 0; u < 3; u++) {
    const int br = (tid >> 5) + 8 * u, cb = br / XH, r = br - cb * XH;
    const int gy = oy0 - 2 + r;
    goff[u] = (gy >= 0 && gy < H) ? (8 * cb * H + gy) * W + (tid & 31) : kNoRow;
    lslot[u] = cb * XCB + r * XP + (tid & 31) + 1;
  }
  const bool hcol = (tid & 1) != 0;                 // item 3: right halo column
  {
    const int br = tid >> 1, cb = br / XH, r = br - cb * XH;
    const int gy = oy0 - 2 + r;
    goff[3] = (tid < 4 * XH && gy >= 0 && gy < H) ? (8 * cb * H + gy) * W + (hcol ? TW : -1) : kNoRow;
    lslot[3] = tid < 4 * XH ? cb * XCB + r * XP + (hcol ? TW + 1 : 0) : -1;
  }
  const int ntx = W / TW;
#define MSF_BX_ISSUE(k_)                                                                          \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                               \
      bool ok = goff[u] != kNoRow;                                                                \
      if (u == 3) ok = ok && (hcol ? (k_) + 1 < ntx : (k_) > 0);                                  \
      const float* src = inf + (ok ? goff[u] : 0) + TW * (k_);                                    \
      _Pragma("unroll") for (int c = 0; c < 8; c++) pre[u][c] = ok ? src[c * HW] : 0.f;           \
    }                                                                                             \
  }
#define MSF_BX_COMMIT()                                                                           \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                               \
      if (lslot[u] < 0) continue;                                                                 \
      bf16x8 vh, vl;                                                                              \
      _Pragma("unroll") for (int c = 0; c < 8; c++) {                                             \
        __bf16 a, b;                                                                              \
        split_bf16(pre[u][c], a, b);                                                              \
        vh[c] = a; vl[c] = b;                                                                     \
      }                                                                                           \
      xh[lslot[u]] = vh; xl[lslot[u]] = vl;                                                       \
    }                                                                                             \
  }
  // this lane's tap of MFMA group g: 2 g + (kq >> 1) (the tenth tap has zero weights: any valid address), block kq & 1
  int xoff[G], tky[G], tkx[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    int t = 2 * g + (kq >> 1);
    t = t < 9 ? t : 8;
    tky[g] = t / 3;
    tkx[g] = t - 3 * tky[g];
    xoff[g] = (kq & 1) * XCB + tky[g] * XP + tkx[g];
  }
  // the loads of tile k+2 are issued right after tile k+1 went to LDS: they have conv2(k-1) and conv1(k+1) to land
  // (these kernels run at memory speed: one convolution's MFMA time is shorter than a trip to HBM)
  MSF_BX_ISSUE(0)
  MSF_BX_COMMIT()
  if (1 < ntx) MSF_BX_ISSUE(1)
  for (int k = 0; k <= ntx; k++) {
    __syncthreads();
    if (k < ntx) {
      // ---- conv1 of tile k, transposed: D[channel][pixel]; t rows 5 hf + u (u = 0..4), pixels 16m .. 16m+15
      f32x4 acc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int xb = 5 * hf * XP + 16 * m + i;
#pragma unroll
      for (int u = 0; u < 5; u++)
#pragma unroll
        for (int g = 0; g < G; g++) {
          const bf16x8 ph = xh[xb + u * XP + xoff[g]], pl = xl[xb + u * XP + xoff[g]];
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1l[g], ph, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[g], pl, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[g], ph, acc[u], 0, 0, 0);
        }
      bf16x4* th4 = reinterpret_cast<bf16x4*>(th);
      bf16x4* tl4 = reinterpret_cast<bf16x4*>(tl);
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int tr = 5 * hf + u, gy = oy0 - 1 + tr;
        f32x4 v = acc[u] + bias1;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (gy < 0 || gy >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};        // conv2 pads t with zeros
        bf16x4 vh, vl;
        __bf16 a, b;
        split_bf16(v.x, a, b); vh[0] = a; vl[0] = b;
        split_bf16(v.y, a, b); vh[1] = a; vl[1] = b;
        split_bf16(v.z, a, b); vh[2] = a; vl[2] = b;
        split_bf16(v.w, a, b); vh[3] = a; vl[3] = b;
        const int slot = (kq >> 1) * TCB + tr * TP + (k & 1) * TW + 16 * m + i;   // channels 4 kq .. +3: block kq >> 1, half kq & 1
        th4[2 * slot + (kq & 1)] = vh;
        tl4[2 * slot + (kq & 1)] = vl;
      }
    }
    __syncthreads();                       // t of tile k is complete; nobody reads the x tile any more
    if (k + 1 < ntx) MSF_BX_COMMIT()
    if (k + 2 < ntx) MSF_BX_ISSUE(k + 2)
    if (k >= 1 && k < ntx && tid < 4 * TROWS) {   // last pixel of tile k-1 -> the halo pixel conv2 of tile k reads
      bf16x8* pl = th + (tid / TROWS) * TCB;
      const int r = tid % TROWS;
      pl[r * TP + 2 * TW + (k & 1)] = pl[r * TP + ((k - 1) & 1) * TW + TW - 1];
    }
    if (k >= 1) {
      // ---- conv2 of tile j = k-1: output rows oy0 + 4 hf + u (u = 0..3), pixels 32j + 16m .. +15
      const int j = k - 1;
      f32x4 rv[4];
#pragma unroll
      for (int u = 0; u < 4; u++)     // residual = x (f32, from global / L2), requested before the MFMAs
        rv[u] = *reinterpret_cast<const f32x4*>(inf + ((long long)i * H + (oy0 + 4 * hf + u)) * W + TW * j + 16 * m + 4 * kq);
      int toff[G];
#pragma unroll
      for (int g = 0; g < G; g++) {
        const int cr = 16 * m + i + tkx[g] - 1;                          // pixel inside the tile, -1 .. 32
        const int cs = cr < 0 ? 2 * TW + (j & 1) : cr >= TW ? (k < ntx ? (k & 1) * TW : 2 * TW + 2) : (j & 1) * TW + cr;
        toff[g] = (kq & 1) * TCB + (4 * hf + tky[g]) * TP + cs;
      }
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int g = 0; g < G; g++) {
          const bf16x8 ah = th[toff[g] + u * TP], al = tl[toff[g] + u * TP];
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w2l[g], acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, w2h[g], acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, w2h[g], acc[u], 0, 0, 0);
        }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int oy = oy0 + 4 * hf + u;
        f32x4 v = acc[u] + f32x4{bias2, bias2, bias2, bias2};
        v += rv[u];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f32x4*>(outf + ((long long)i * H + oy) * W + TW * j + 16 * m + 4 * kq) = v;
      }
    }
  }
#undef MSF_BX_ISSUE
#undef MSF_BX_COMMIT
}

// ------------------------------------------------------------------ 3x3 stride-1 convolution, C -> C channels, split-bf16 MFMA
// k_conv's tiling (a workgroup walks the 16-pixel x tiles of a band of output rows, the next tile's loads in flight)
// with the arithmetic of k_block8x: operands split into bf16 hi + lo, three v_mfma_f32_16x16x32_bf16 per product.
//  * LDS holds the input tile as planes of 8 channels: [hi | lo][channel block][row][pixel] x 16 bytes, so one
//    ds_read_b128 is the fragment of (tap, channel block) for a lane's pixel; K = 32 of one MFMA = the 4 channel
//    blocks of one tap (C = 32: lane group kq = channel block; 9 MFMA groups = the 9 taps).  A channel-block plane is
//    a multiple of 16 pixels, so the two blocks a lane group reads fall on disjoint banks.
//  * the packed weights ([tap][cout tile][hi | lo][lane], 36.9 KB for C = 32) are copied to LDS once per workgroup;
//  * a wave computes output rows w and w + 4 of the band (8 rows), all C output channels: per tap 4 pixel + 4 weight
//    fragment reads feed 12 MFMAs.
// At 1/5 of the f32 MFMA time these layers run at their memory time.  Not bit-identical to k_conv (see k_block8x).
namespace cvx {
constexpr int OTW = 16, OTH = 8;
constexpr int IN_H = OTH + 2, IN_W = OTW + 2;
constexpr int PITCH = 24;                          // pixel slots per tile row (18 used)
constexpr int CBPLANE = IN_H * PITCH;              // 240 = 15 x 16
static_assert(CBPLANE % 16 == 0, "channel-block planes must be multiples of 16 pixel slots");
template <int C>
struct Cfg {
  static_assert(C == 32, "k_convx: 32 channels");
  static constexpr int NCB = C / 8, NT = C / 16, G = 9;
  static constexpr int HLPLANE = NCB * CBPLANE;
  static constexpr int XSLOTS = 2 * HLPLANE;
  static constexpr int WSLOTS = G * NT * 2 * 64;   // 16-byte fragments: [tap][cout tile][hi | lo][lane]
  static constexpr int LDS_BYTES = 16 * (XSLOTS + WSLOTS);
  static constexpr int NITEMS = NCB * IN_H * IN_W;
  static constexpr int NLD = (NITEMS + 255) / 256;
};
}  // namespace cvx

template <int C, bool RES>
__global__ __launch_bounds__(256, 2) void k_convx(const float* __restrict__ in, const uint16_t* __restrict__ wx,
                                                  const float* __restrict__ bias, const float* __restrict__ res,
                                                  float* __restrict__ out, int H, int W, int n_bands) {
  using namespace cvx;
  using F = Cfg<C>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xs = reinterpret_cast<bf16x8*>(lds);
  bf16x8* ws = xs + F::XSLOTS;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, band) order
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * OTH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const long long HW = (long long)H * W;
  const float* inf = in + (long long)img * C * HW;

  {  // the layer's weight fragments -> LDS
    const bf16x8* src = reinterpret_cast<const bf16x8*>(wx);
    for (int idx = tid; idx < F::WSLOTS; idx += 256) ws[idx] = src[idx];
  }
  // staging items: (channel block, tile row, tile column) -> 8 channel dwords of one pixel
  constexpr int kNoRow = -(1 << 30);
  float pre[F::NLD][8];
  int goff[F::NLD], lslot[F::NLD], col[F::NLD];
#pragma unroll
  for (int u = 0; u < F::NLD; u++) {
    const int idx = tid + 256 * u;
    const int cb = idx / (IN_H * IN_W), rm = idx - cb * (IN_H * IN_W);
    const int r = rm / IN_W, c = rm - r * IN_W;
    const int gy = oy0 - 1 + r;
    col[u] = c - 1;
    lslot[u] = idx < F::NITEMS ? cb * CBPLANE + r * PITCH + c : -1;
    goff[u] = (idx < F::NITEMS && gy >= 0 && gy < H) ? (8 * cb * H + gy) * W + c - 1 : kNoRow;
  }
#define MSF_CX_ISSUE(ox0_)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < F::NLD; u++) {                                          \
      const int gx = (ox0_) + col[u];                                                             \
      const bool ok = goff[u] != kNoRow && gx >= 0 && gx < W;                                     \
      const float* src = inf + (ok ? goff[u] + (ox0_) : 0);                                       \
      _Pragma("unroll") for (int c = 0; c < 8; c++) pre[u][c] = ok ? src[c * HW] : 0.f;           \
    }                                                                                             \
  }
  const int ntx = (W + OTW - 1) / OTW;
  MSF_CX_ISSUE(0)
  for (int tx = 0; tx < ntx; tx++) {
    const int ox0 = tx * OTW;
    __syncthreads();                       // every wave is done reading the previous tile
#pragma unroll
    for (int u = 0; u < F::NLD; u++) {
      if (lslot[u] < 0) continue;
      bf16x8 vh, vl;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        __bf16 a, b;
        split_bf16(pre[u][c], a, b);
        vh[c] = a; vl[c] = b;
      }
      xs[lslot[u]] = vh;
      xs[F::HLPLANE + lslot[u]] = vl;
    }
    __syncthreads();
    if (tx + 1 < ntx) MSF_CX_ISSUE(ox0 + OTW)

    f32x4 acc[2][F::NT], rv[2][F::NT];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int n = 0; n < F::NT; n++) {
        acc[u][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        rv[u][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (RES) {      // residual operand: requested now, consumed after the MFMAs
          const int oy = oy0 + wave + 4 * u, px = ox0 + 4 * kq;
          if (oy < H && px < W)
            rv[u][n] = *reinterpret_cast<const f32x4*>(res + (((long long)img * C + 16 * n + i) * H + oy) * W + px);
        }
      }
    const int ab = kq * CBPLANE + wave * PITCH + i;
#pragma unroll
    for (int g = 0; g < F::G; g++) {
      const int ky = g / 3, kx = g - 3 * ky;
      bf16x8 bh[F::NT], bl[F::NT];
#pragma unroll
      for (int n = 0; n < F::NT; n++) {
        bh[n] = ws[((g * F::NT + n) * 2 + 0) * 64 + lane];
        bl[n] = ws[((g * F::NT + n) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int slot = ab + (4 * u + ky) * PITCH + kx;
        const bf16x8 ah = xs[slot], al = xs[F::HLPLANE + slot];
#pragma unroll
        for (int n = 0; n < F::NT; n++) {
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[u][n], 0, 0, 0);
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[u][n], 0, 0, 0);
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[u][n], 0, 0, 0);
        }
      }
    }
    // epilogue: D[pixel 4 kq + r][cout 16 n + i] -> out[img][cout][oy][4 consecutive px], + bias (+ residual), ReLU
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int n = 0; n < F::NT; n++) {
        const int oy = oy0 + wave + 4 * u, px = ox0 + 4 * kq, co = 16 * n + i;
        if (oy >= H || px >= W) continue;
        const float bv = bias[co];
        f32x4 v = acc[u][n] + f32x4{bv, bv, bv, bv};
        if (RES) v += rv[u][n];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f32x4*>(out + (((long long)img * C + co) * H + oy) * W + px) = v;
      }
  }
#undef MSF_CX_ISSUE
}

// ------------------------------------------------------------------ 3x3 stride-2 convolution CIN -> 32 + 1x1 stride-2 shortcut, split-bf16
// The entry convolutions of layer3 (16 -> 32) and layer4 (32 -> 32) in k_convx's form: a workgroup walks the 16-pixel x
// tiles of a band of output rows, input tile in LDS as [hi | lo][channel block][row][column] planes of 16-byte pixels
// with the columns de-interleaved (even | odd), so the fragment of 16 neighbouring stride-2 pixels is 16 consecutive
// slots.  K = 32 of an MFMA: CIN = 32: the 4 channel blocks of one tap (9 groups); CIN = 16: two taps x 2 channel blocks
// (5 groups, the tenth tap has zero weights).  The shortcut's input pixel is the centre tap: its fragment is already in
// registers, three more MFMAs per output-channel tile into a second accumulator (weights in registers).
// Output rows per wave: 2 (CIN = 16, band of 8) or 1 (CIN = 32, band of 4: the tile has twice the planes).
namespace cvx2 {
constexpr int OTW = 16;
constexpr int IN_W = 2 * OTW + 1;                  // 33 columns: 2 ox0 - 1 .. 2 ox0 + 31
constexpr int PITCH = 36, IODD = 18;               // even columns in slots 0 .. 16, odd columns in slots 18 .. 33
template <int CIN>
struct Cfg {
  static_assert(CIN == 16 || CIN == 32, "k_convx2: 16 or 32 input channels");
  static constexpr int NCB = CIN / 8, NT = 2;
  static constexpr int RPW = CIN == 16 ? 2 : 1;    // output rows per wave
  static constexpr int OTH = 4 * RPW;
  static constexpr int IN_H = 2 * OTH + 1;
  static constexpr int G = CIN == 32 ? 9 : 5;
  static constexpr int GC = CIN == 32 ? 4 : 2;     // the MFMA group that holds the centre tap
  static constexpr int CBPLANE = IN_H * PITCH;
  static constexpr int HLPLANE = NCB * CBPLANE;
  static constexpr int XSLOTS = 2 * HLPLANE;
  static constexpr int WSLOTS = G * NT * 2 * 64;   // [g][cout tile][hi | lo][lane]
  static constexpr int SCSLOTS = NT * 2 * 64;      // shortcut: [cout tile][hi | lo][lane]
  static constexpr int LDS_BYTES = 16 * (XSLOTS + WSLOTS);
  static constexpr int NITEMS = NCB * IN_H * IN_W;
  static constexpr int NLD = (NITEMS + 255) / 256;
};
}  // namespace cvx2

template <int CIN>
__global__ __launch_bounds__(256, 2) void k_convx2(const float* __restrict__ in, const uint16_t* __restrict__ wx,
                                                   const float* __restrict__ bias, const uint16_t* __restrict__ wsc,
                                                   const float* __restrict__ bias_sc, float* __restrict__ out,
                                                   float* __restrict__ out_sc, int Hin, int Win, int H, int W, int n_bands) {
  using namespace cvx2;
  using F = Cfg<CIN>;
  constexpr int COUT = 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bf16x8* xs = reinterpret_cast<bf16x8*>(lds);
  bf16x8* ws = xs + F::XSLOTS;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * F::OTH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const long long HWin = (long long)Hin * Win;
  const float* inf = in + (long long)img * CIN * HWin;
  {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(wx);
    for (int idx = tid; idx < F::WSLOTS; idx += 256) ws[idx] = src[idx];
  }
  bf16x8 sch[F::NT], scl[F::NT];                   // shortcut fragments stay in registers
#pragma unroll
  for (int n = 0; n < F::NT; n++) {
    sch[n] = reinterpret_cast<const bf16x8*>(wsc)[(n * 2 + 0) * 64 + lane];
    scl[n] = reinterpret_cast<const bf16x8*>(wsc)[(n * 2 + 1) * 64 + lane];
  }
  // staging items: (channel block, tile row, tile column) -> 8 channel dwords of one pixel
  constexpr int kNoRow = -(1 << 30);
  float pre[F::NLD][8];
  int goff[F::NLD], lslot[F::NLD], col[F::NLD];
#pragma unroll
  for (int u = 0; u < F::NLD; u++) {
    const int idx = tid + 256 * u;
    const int cb = idx / (F::IN_H * IN_W), rm = idx - cb * (F::IN_H * IN_W);
    const int r = rm / IN_W, c = rm - r * IN_W;
    const int gy = 2 * oy0 - 1 + r;
    col[u] = c - 1;
    lslot[u] = idx < F::NITEMS ? cb * F::CBPLANE + r * PITCH + ((c & 1) ? IODD + (c >> 1) : (c >> 1)) : -1;
    goff[u] = (idx < F::NITEMS && gy >= 0 && gy < Hin) ? (8 * cb * Hin + gy) * Win + c - 1 : kNoRow;
  }
#define MSF_C2_ISSUE(ox0_)                                                                        \
  {                                                                                               \
    _Pragma("unroll") for (int u = 0; u < F::NLD; u++) {                                          \
      const int gx = 2 * (ox0_) + col[u];                                                         \
      const bool ok = goff[u] != kNoRow && gx >= 0 && gx < Win;                                   \
      const float* src = inf + (ok ? goff[u] + 2 * (ox0_) : 0);                                   \
      _Pragma("unroll") for (int c = 0; c < 8; c++) pre[u][c] = ok ? src[c * HWin] : 0.f;         \
    }                                                                                             \
  }
  // this lane's K block of MFMA group g: CIN = 32: tap g, channel block kq; CIN = 16: tap 2g + (kq >> 1), block kq & 1
  int foff[F::G];
#pragma unroll
  for (int g = 0; g < F::G; g++) {
    int t = CIN == 32 ? g : 2 * g + (kq >> 1);
    t = t < 9 ? t : 8;
    const int ky = t / 3, kx = t - 3 * ky;
    const int cb = CIN == 32 ? kq : (kq & 1);
    foff[g] = cb * F::CBPLANE + ky * PITCH + (kx == 1 ? IODD + i : kx == 0 ? i : i + 1);
  }
  const int ntx = (W + OTW - 1) / OTW;
  MSF_C2_ISSUE(0)
  for (int tx = 0; tx < ntx; tx++) {
    const int ox0 = tx * OTW;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < F::NLD; u++) {
      if (lslot[u] < 0) continue;
      bf16x4 h0, l0, h1, l1;
      split4(f32x4{pre[u][0], pre[u][1], pre[u][2], pre[u][3]}, h0, l0);
      split4(f32x4{pre[u][4], pre[u][5], pre[u][6], pre[u][7]}, h1, l1);
      bf16x4* dst = reinterpret_cast<bf16x4*>(xs + lslot[u]);
      dst[0] = h0; dst[1] = h1; dst[2 * F::HLPLANE] = l0; dst[2 * F::HLPLANE + 1] = l1;
    }
    __syncthreads();
    if (tx + 1 < ntx) MSF_C2_ISSUE(ox0 + OTW)

    f32x4 acc[F::RPW][F::NT], asc[F::RPW][F::NT];
#pragma unroll
    for (int u = 0; u < F::RPW; u++)
#pragma unroll
      for (int n = 0; n < F::NT; n++) { acc[u][n] = f32x4{0.f, 0.f, 0.f, 0.f}; asc[u][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int g = 0; g < F::G; g++) {
      bf16x8 bh[F::NT], bl[F::NT];
#pragma unroll
      for (int n = 0; n < F::NT; n++) {
        bh[n] = ws[((g * F::NT + n) * 2 + 0) * 64 + lane];
        bl[n] = ws[((g * F::NT + n) * 2 + 1) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < F::RPW; u++) {
        const int slot = foff[g] + 2 * (wave + 4 * u) * PITCH;     // output row wave + 4u: input rows 2 (wave + 4u) + ky
        const bf16x8 ah = xs[slot], al = xs[F::HLPLANE + slot];
#pragma unroll
        for (int n = 0; n < F::NT; n++) {
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[u][n], 0, 0, 0);
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[u][n], 0, 0, 0);
          acc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[u][n], 0, 0, 0);
          if (g == F::GC) {
            asc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, scl[n], asc[u][n], 0, 0, 0);
            asc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, sch[n], asc[u][n], 0, 0, 0);
            asc[u][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, sch[n], asc[u][n], 0, 0, 0);
          }
        }
      }
    }
    // epilogue: D[pixel 4 kq + r][cout 16 n + i] -> out (bias, ReLU) and out_sc (bias, no ReLU), 4 consecutive px
#pragma unroll
    for (int u = 0; u < F::RPW; u++)
#pragma unroll
      for (int n = 0; n < F::NT; n++) {
        const int oy = oy0 + wave + 4 * u, px = ox0 + 4 * kq, co = 16 * n + i;
        if (oy >= H || px >= W) continue;
        const float bv = bias[co], bs = bias_sc[co];
        f32x4 v = acc[u][n] + f32x4{bv, bv, bv, bv};
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        const long long o = (((long long)img * COUT + co) * H + oy) * W + px;
        *reinterpret_cast<f32x4*>(out + o) = v;
        *reinterpret_cast<f32x4*>(out_sc + o) = asc[u][n] + f32x4{bs, bs, bs, bs};
      }
  }
#undef MSF_C2_ISSUE
}

// ------------------------------------------------------------------ fused BasicBlock, 16 channels, stride 1 (layer2 @ 120 x 160)
// The same scheme as k_block8 without row packing (the 16 MFMA columns are the 16 output channels): a workgroup owns a
// band of R = 8 output rows and walks its x tiles of 32 columns; wave w takes M tile w & 1 (16 columns) of the t rows
// 5 (w >> 1) .. +4 in conv1 and of the output rows 4 (w >> 1) .. +3 in conv2.  Both convolutions' 36 weight fragments
// stay in registers.  75.8 KB of LDS: two workgroups per CU.  Same MFMA chains in the same k order as k_conv<16,16,3,1>:
// bit-identical to the two-kernel path.
namespace blk16 {
constexpr int CH = 16, R = 8, TW = 32;
constexpr int XH = R + 4;                          // x rows oy0-2 .. oy0+9
constexpr int XO = 3;                              // staged rows start at the float4-aligned column 32k - 4; window starts at 32k - 1
constexpr int XW4 = (TW + 2 + XO + 3) / 4;         // 10 float4 per staged row
constexpr int XPITCH = 4 * XW4;                    // 40
constexpr int XPLANE = ((XH * XPITCH + 15) / 32) * 32 + 16;
constexpr int TROWS = R + 2;                       // t rows oy0-1 .. oy0+8
constexpr int TPITCH = 2 * TW + 4;                 // two 32-column segments, halo columns 64 / 66, zero column 65
constexpr int TPLANE = ((TROWS * TPITCH + 15) / 32) * 32 + 16;
constexpr int KSTEPS = 36;                         // 3 x 3 x 16 / 4
constexpr int LDS_FLOATS = CH * XPLANE + CH * TPLANE;
static_assert(XPLANE % 32 == 16 && TPLANE % 32 == 16, "plane strides");
static_assert(XPLANE >= XH * XPITCH && TPLANE >= TROWS * TPITCH, "planes too small");
}  // namespace blk16

__global__ __launch_bounds__(256, 2) void k_block16(const float* __restrict__ in, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, float* __restrict__ out, int H, int W,
                                                 int n_bands) {
  using namespace blk16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;
  float* tT = lds + CH * XPLANE;
  const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7;
  const int unit = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);     // XCD-contiguous (image, band) order
  const int img = unit / n_bands;
  const int oy0 = (unit - img * n_bands) * R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;          // MFMA column = output channel i
  const int m = wave & 1, hf = wave >> 1;           // M tile (16 columns of the 32), row half
  const float* inf = in + (long long)img * CH * H * W;
  float* outf = out + (long long)img * CH * H * W;

  float bw1[KSTEPS], bw2[KSTEPS];
#pragma unroll
  for (int j = 0; j < KSTEPS; j++) {
    bw1[j] = w1[(j * 4 + kq) * 16 + i];
    bw2[j] = w2[(j * 4 + kq) * 16 + i];
  }
  const float bias1 = b1[i], bias2 = b2[i];
  // halo column of tile 0 (= padding), the zero column and the second halo column
  for (int idx = tid; idx < CH * TROWS; idx += 256) {
    const int c = idx / TROWS, r = idx - c * TROWS;
    *reinterpret_cast<f32x4*>(&tT[c * TPLANE + r * TPITCH + 2 * TW]) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  constexpr int TOTAL = CH * XH * XW4;
  constexpr int NLD = (TOTAL + 255) / 256;
  f32x4 pre[NLD];
  // (channel, row, float4 column) of a thread's u-th group is re-derived where it is needed (divisions by constants):
  // keeping the offsets in registers would cost 16 VGPRs and the second wave per SIMD
#define MSF_BLK_DECODE(u_)                                  \
  const int idx_ = tid + 256 * (u_);                        \
  const int c_ = idx_ / (XH * XW4);                         \
  const int rm_ = idx_ - c_ * (XH * XW4);                   \
  const int r_ = rm_ / XW4;                                 \
  const int x4_ = rm_ - r_ * XW4;
#define MSF_BLK_ISSUE(k_)                                                                       \
  {                                                                                             \
    const int gx0_ = TW * (k_) - 4;                                                             \
    _Pragma("unroll") for (int u = 0; u < NLD; u++) {                                           \
      MSF_BLK_DECODE(u)                                                                         \
      const int gy = oy0 - 2 + r_, gx = gx0_ + 4 * x4_;                                         \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
      if (idx_ < TOTAL && gy >= 0 && gy < H && gx >= 0 && gx + 4 <= W)                          \
        v = *reinterpret_cast<const f32x4*>(inf + (c_ * H + gy) * W + gx);                      \
      pre[u] = v;                                                                               \
    }                                                                                           \
  }
#define MSF_BLK_COMMIT()                                                                        \
  {                                                                                             \
    _Pragma("unroll") for (int u = 0; u < NLD; u++) {                                           \
      MSF_BLK_DECODE(u)                                                                         \
      if (idx_ < TOTAL) *reinterpret_cast<f32x4*>(&xT[c_ * XPLANE + r_ * XPITCH + 4 * x4_]) = pre[u]; \
    }                                                                                           \
  }
  const int ntx = W / TW;
  MSF_BLK_ISSUE(0)
  MSF_BLK_COMMIT()
  // iteration k: conv1 of tile k | barrier | x tile k+1 and the halo column into LDS, conv2 of tile k-1 (see k_block8)
  for (int k = 0; k <= ntx; k++) {
    __syncthreads();
    if (k + 1 < ntx) MSF_BLK_ISSUE(k + 1)
    if (k < ntx) {
      // ---- conv1 of tile k: t rows 5 hf + u (u = 0..4), columns 16m .. 16m+15 of the tile
      f32x4 acc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      // software pipeline: the five fragment reads of step s + 1 are issued before the five MFMAs of step s, and the
      // scheduler is told to keep it that way (left alone it hoists dozens of reads and the kernel no longer fits two
      // waves per SIMD)
      const float* xbase = &xT[kq * XPLANE + 5 * hf * XPITCH + XO + 16 * m + i];
      float an[5], ac[5];
#pragma unroll
      for (int u = 0; u < 5; u++) ac[u] = xbase[u * XPITCH];
#pragma unroll
      for (int step = 0; step < KSTEPS; step++) {
        if (step + 1 < KSTEPS) {
          const int kk = ((step + 1) * 4) / CH, c = ((step + 1) * 4) % CH;
          const int ky = kk / 3, kx = kk - ky * 3;
          const float* ap = xbase + c * XPLANE + ky * XPITCH + kx;
#pragma unroll
          for (int u = 0; u < 5; u++) an[u] = ap[u * XPITCH];
        }
#pragma unroll
        for (int u = 0; u < 5; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bw1[step], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 5; u++) ac[u] = an[u];
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);   // 5 LDS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);   // 5 MFMAs
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int tr = 5 * hf + u, gy = oy0 - 1 + tr;
        f32x4 v = acc[u] + f32x4{bias1, bias1, bias1, bias1};
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (gy < 0 || gy >= H) v = f32x4{0.f, 0.f, 0.f, 0.f};        // conv2 pads t with zeros
        *reinterpret_cast<f32x4*>(&tT[i * TPLANE + tr * TPITCH + (k & 1) * TW + 16 * m + 4 * kq]) = v;
      }
    }
    __syncthreads();                       // t of tile k is complete; nobody reads the x tile any more
    if (k + 1 < ntx) MSF_BLK_COMMIT()
    if (k >= 1 && k < ntx) {               // last column of tile k-1 -> the halo column conv2 of tile k reads (next iteration)
      for (int idx = tid; idx < CH * TROWS; idx += 256) {
        const int c = idx / TROWS, r = idx - c * TROWS;
        float* row = &tT[c * TPLANE + r * TPITCH];
        row[2 * TW + 2 * (k & 1)] = row[((k - 1) & 1) * TW + TW - 1];
      }
    }
    if (k >= 1) {
      // ---- conv2 of tile j = k-1: output rows oy0 + 4 hf + u (u = 0..3), columns 32j + 16m .. +15
      const int j = k - 1;
      f32x4 rv[4];
#pragma unroll
      for (int u = 0; u < 4; u++)     // residual = x, requested before the MFMA loop
        rv[u] = *reinterpret_cast<const f32x4*>(inf + ((long long)i * H + (oy0 + 4 * hf + u)) * W + TW * j + 16 * m + 4 * kq);
      int colterm[3];
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int cr = 16 * m + i + kx - 1;                               // column inside the tile, -1 .. 32
        colterm[kx] = cr < 0 ? 2 * TW + 2 * (j & 1) : cr >= TW ? (k < ntx ? (k & 1) * TW : 2 * TW + 1) : (j & 1) * TW + cr;
      }
      f32x4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* tbase = &tT[kq * TPLANE + 4 * hf * TPITCH];
      float an[4], ac[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ac[u] = tbase[colterm[0] + u * TPITCH];
#pragma unroll
      for (int step = 0; step < KSTEPS; step++) {
        if (step + 1 < KSTEPS) {
          const int kk = ((step + 1) * 4) / CH, c = ((step + 1) * 4) % CH;
          const int ky = kk / 3, kx = kk - ky * 3;
          const float* ap = tbase + c * TPLANE + ky * TPITCH + colterm[kx];
#pragma unroll
          for (int u = 0; u < 4; u++) an[u] = ap[u * TPITCH];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bw2[step], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; u++) ac[u] = an[u];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int oy = oy0 + 4 * hf + u;
        f32x4 v = acc[u] + f32x4{bias2, bias2, bias2, bias2};
        v += rv[u];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<f32x4*>(outf + ((long long)i * H + oy) * W + TW * j + 16 * m + 4 * kq) = v;
      }
    }
  }
#undef MSF_BLK_DECODE
#undef MSF_BLK_COMMIT
#undef MSF_BLK_ISSUE
}

// ------------------------------------------------------------------ tokens: + positional encoding, n c h w -> n (h w) c
// PE add + 'n c h w -> n (hw) c' as a tiled transpose: a workgroup takes 32 channels x 64 tokens of one image, reads each
// channel's 64 tokens as one 256-byte segment, and writes the 64 tokens' 32 channels as one contiguous 8-KB run (the
// element-per-thread form read with a 4 800-byte stride between neighbouring lanes: 42 us per 256 images at 1.9 TB/s)
constexpr int kTokTile = 64;
__global__ __launch_bounds__(256) void k_tokens(const float* __restrict__ bb, const float* __restrict__ pe,
                                                float* __restrict__ tok, int n_img) {
  __shared__ float tile[DM][kTokTile + 1];
  const int img = blockIdx.y, t0 = blockIdx.x * kTokTile, tid = threadIdx.x;
  if (img >= n_img) return;
  const float* src = bb + (long long)img * DM * NTOK;
#pragma unroll
  for (int k = 0; k < DM * kTokTile / 256; k++) {
    const int e = k * 256 + tid, c = e / kTokTile, t = e - c * kTokTile;
    if (t0 + t < NTOK) tile[c][t] = src[(long long)c * NTOK + t0 + t] + pe[c * NTOK + t0 + t];
  }
  __syncthreads();
  float* dst = tok + ((long long)img * NTOK + t0) * DM;
#pragma unroll
  for (int k = 0; k < DM * kTokTile / 256; k++) {
    const int e = k * 256 + tid, t = e / DM, c = e - t * DM;
    if (t0 + t < NTOK) dst[e] = tile[c][t];
  }
}

// r05: the backbone's last convolution (1 x 1, 32 -> 32, no ReLU) + positional encoding + 'n c h w -> n (h w) c' in ONE pass:
// k_conv<32, 32, 1, 1> wrote its result as NCHW and k_tokens read it back to transpose it (two launches per side, 0.16 GB
// written and read per 512 images for 20 MFLOP per image).  A workgroup takes 64 tokens of one image -- 32 channels x 256
// contiguous bytes in, as k_tokens did -- and runs the product TRANSPOSED (A = weights, B = tokens): a lane then holds four
// consecutive output channels of one token, i.e. one 16-byte store into the token's 128-byte row.  Same f32 MFMA chain in
// the same k order as k_conv (operands swapped, products and their order unchanged), then + bias, then + PE, as the two
// kernels did: bit-identical (MSF_LOFTR_OUT_FUSED=0 keeps the two-kernel tail; tests compare).
__global__ __launch_bounds__(256) void k_out_tokens(const float* __restrict__ act /*[img][32][NTOK]*/, const float* __restrict__ wB,
                                                    const float* __restrict__ bias, const float* __restrict__ pe,
                                                    float* __restrict__ tok_a, int n_a, float* __restrict__ tok_b, int n_img) {
  __shared__ float xt[DM][kTokTile + 1], pt[DM][kTokTile + 1];
  const int img = blockIdx.y, t0 = blockIdx.x * kTokTile, tid = threadIdx.x;
  if (img >= n_img) return;
  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
  const float* src = act + (long long)img * DM * NTOK;
  float wf[DM / 4][2];                         // weight fragments: k step s, output-channel tile mt
#pragma unroll
  for (int sI = 0; sI < DM / 4; sI++)
#pragma unroll
    for (int mt = 0; mt < 2; mt++) wf[sI][mt] = wB[(sI * 4 + kq) * DM + mt * 16 + i];
#pragma unroll
  for (int k = 0; k < DM * kTokTile / 256; k++) {
    const int e = k * 256 + tid, c = e / kTokTile, t = e - c * kTokTile;
    const bool ok = t0 + t < NTOK;
    xt[c][t] = ok ? src[(long long)c * NTOK + t0 + t] : 0.f;
    pt[c][t] = ok ? pe[c * NTOK + t0 + t] : 0.f;
  }
  __syncthreads();
  const int tl = 16 * wave + i;                // this lane's token inside the tile
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int sI = 0; sI < DM / 4; sI++) {
    const float b = xt[sI * 4 + kq][tl];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[sI][mt], b, acc[mt], 0, 0, 0);
  }
  if (t0 + tl >= NTOK) return;
  float* dst = (img < n_a ? tok_a + (long long)img * NTOK * DM : tok_b + (long long)(img - n_a) * NTOK * DM) + (long long)(t0 + tl) * DM;
#pragma unroll
  for (int mt = 0; mt < 2; mt++) {
    const int co = 16 * mt + 4 * kq;           // D[row = channel 4 kq + r of the tile][col = token i]
    f32x4 v = acc[mt] + (bias ? *reinterpret_cast<const f32x4*>(bias + co) : f32x4{0.f, 0.f, 0.f, 0.f});
    v += f32x4{pt[co][tl], pt[co + 1][tl], pt[co + 2][tl], pt[co + 3][tl]};
    *reinterpret_cast<f32x4*>(dst + co) = v;
  }
}

// ------------------------------------------------------------------ linear-attention encoder block (MFMA)
// All products of a block are 16-wide tiles of v_mfma_f32_16x16x4_f32 (exact f32).  Two k-slot orders let every
// operand be used where it already is, with no transposes and no LDS round trips for activations:
//   P8: operand rows that come from memory: slot (s, kq) <-> feature 8*kq + s, so a lane reads 8 contiguous floats;
//   PD: operands that are MFMA results in registers (lane = column, regs = rows 4*(lane>>4)+r of each 16-row tile):
//       slot (s', g) <-> feature 16*(s'/4) + 4*g + (s'%4), i.e. exactly the register the lane already holds.
// Weights are re-ordered to those slot orders once on the host (pack_weight).
struct BlockW {
  const float *wq_p, *wk_p, *wv_p, *wm_p, *w0_p, *w1_p;   // permuted: [(mtile*KS + slot)*64 + lane]
  const float *n1w, *n1b, *n2w, *n2b;
  // the same matrices as split-bf16 MFMA fragments (k_attn_update_x): [mtile][K group of 8 slots][hi | lo][lane][8]
  const uint16_t *wq_x, *wm_x, *w0_x, *w1_x;
};

__device__ __forceinline__ float elu1(float x) { return (x > 0.f ? x : expf(x) - 1.f) + 1.f; }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float quad_sum(float v) {   // sum over the 4 lane groups that share lane & 15
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// phase A: K = elu(s Wk) + 1, V = (s Wv) / 1200, KV = sum_t K_t^T V_t, Ksum = sum_t K_t.  One workgroup per source
// sequence, 4 waves x 16-token tiles; tokens on MFMA rows so K and V tiles feed the KV product straight from registers.
constexpr int kKvWaves = 8;   // one workgroup per sequence: 8 waves (2 per SIMD) so loads and MFMAs of different waves overlap
__global__ __launch_bounds__(64 * kKvWaves) void k_attn_kv(const float* __restrict__ src, long long seq_stride, BlockW w,
                                                 float* __restrict__ kv /*[n][1056]: KV in PD order | Ksum*/) {
  __shared__ float sWk[DM * DM], sWv[DM * DM];
  __shared__ float red[kKvWaves][DM * DM + DM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = lane & 15, g = lane >> 4;
  const float* s = src + (long long)blockIdx.x * seq_stride;
  for (int i = tid; i < DM * DM; i += 64 * kKvWaves) { sWk[i] = w.wk_p[i]; sWv[i] = w.wv_p[i]; }
  __syncthreads();
  f32x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ksum[2] = {0.f, 0.f};
  for (int tile = wave; tile < NTOK / 16; tile += kKvWaves) {
    const float* xr = s + (long long)(tile * 16 + tl) * DM + 8 * g;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr), x1 = *reinterpret_cast<const f32x4*>(xr + 4);
    const float xa[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    f32x4 K[2], V[2];
#pragma unroll
    for (int n = 0; n < 2; n++) {
      K[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      V[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 8; sI++) {
        K[n] = mfma4(xa[sI], sWk[(n * 8 + sI) * 64 + lane], K[n]);
        V[n] = mfma4(xa[sI], sWv[(n * 8 + sI) * 64 + lane], V[n]);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        K[n][r] = elu1(K[n][r]);
        V[n][r] = V[n][r] / 1200.0f;
        ksum[n] += K[n][r];
      }
    }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[m][n] = mfma4(K[m][r], V[n][r], acc[m][n]);   // k-slot = token 4g + r
  }
  // per-wave partials -> LDS (plain [d][e]) ; Ksum[16n + tl] summed over the lane groups
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[wave][(16 * m + 4 * g + r) * DM + 16 * n + tl] = acc[m][n][r];
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const float t = quad_sum(ksum[n]);
    if (g == 0) red[wave][DM * DM + 16 * n + tl] = t;
  }
  __syncthreads();
  for (int i = tid; i < DM * DM + DM; i += 64 * kKvWaves) {
    float t = red[0][i];
#pragma unroll
    for (int w2 = 1; w2 < kKvWaves; w2++) t += red[w2][i];
    red[0][i] = t;
  }
  __syncthreads();
  float* o = kv + (long long)blockIdx.x * (DM * DM + DM);
  for (int i = tid; i < DM * DM; i += 64 * kKvWaves) {     // KV as the A operand of msg = KV^T Q, PD slot order over d
    const int ln = i & 63, sl = (i >> 6) & 7, me = i >> 9;
    const int d = 16 * (sl >> 2) + 4 * (ln >> 4) + (sl & 3), e = 16 * me + (ln & 15);
    o[i] = red[0][d * DM + e];
  }
  if (tid < DM) o[DM * DM + tid] = red[0][DM * DM + tid];
}

// k_attn_kv on split-bf16 MFMAs.  Projections: the lane's 8 loaded features are the K = 32 fragment (as in
// k_attn_update_x).  KV = sum over tokens of K[tok][d] V[tok][e]: the MFMA's K dimension is the token index, and a lane
// holds 4 tokens of a tile per accumulator register set, so two token tiles are taken together: 8 registers = one
// fragment, 3 MFMAs per (d tile, e tile) and tile pair instead of 8.  Defined after split8 / mfma3x below.
// LayerNorm over the 32 features of a token held as v[2] (rows 16m + 4g + r of column lane & 15)
__device__ __forceinline__ void layer_norm_cols(f32x4* v, const float* w, const float* b, int g) {
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) sum += v[m][r];
  const float mean = quad_sum(sum) / (float)DM;
  float var = 0.f;
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) { const float c = v[m][r] - mean; var += c * c; }
  const float den = sqrtf(quad_sum(var) / (float)DM + 1.0000000116860974e-07f);
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int f = 16 * m + 4 * g + r;
      v[m][r] = (v[m][r] - mean) / den * w[f] + b[f];
    }
}

// phase B: features on MFMA rows, tokens on columns; one wave carries 16 tokens through the whole block.
#define MSF_LOFTR_UPD_TILES 2
constexpr int kUpdTilesPerWave = MSF_LOFTR_UPD_TILES;   // token tiles per wave and workgroup item (1 / 2 / 3 / 5 measured)
__global__ __launch_bounds__(256) void k_attn_update(const float* __restrict__ xsrc, long long x_stride,
                                                     const float* __restrict__ kv, BlockW w, float* __restrict__ dst,
                                                     long long d_stride) {
  __shared__ float sWq[DM * DM], sKV[DM * DM], sWm[DM * DM], sW0[64 * 64], sW1[64 * DM];
  __shared__ float sLN[4 * DM], sKs[DM];
  const int seq = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = lane & 15, g = lane >> 4;
  const float* kvp = kv + (long long)seq * (DM * DM + DM);
  for (int i = tid; i < DM * DM; i += 256) { sWq[i] = w.wq_p[i]; sKV[i] = kvp[i]; sWm[i] = w.wm_p[i]; }
  for (int i = tid; i < 64 * 64; i += 256) sW0[i] = w.w0_p[i];
  for (int i = tid; i < 64 * DM; i += 256) sW1[i] = w.w1_p[i];
  if (tid < DM) {
    sLN[tid] = w.n1w[tid]; sLN[DM + tid] = w.n1b[tid]; sLN[2 * DM + tid] = w.n2w[tid]; sLN[3 * DM + tid] = w.n2b[tid];
    sKs[tid] = kvp[DM * DM + tid];
  }
  __syncthreads();
  for (int it = 0; it < kUpdTilesPerWave; it++) {
    const int tile = (blockIdx.x * kUpdTilesPerWave + it) * 4 + wave;
    if (tile >= NTOK / 16) break;
    const float* xr = xsrc + (long long)seq * x_stride + (long long)(tile * 16 + tl) * DM;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(xr + 8 * g), b1 = *reinterpret_cast<const f32x4*>(xr + 8 * g + 4);
    const float xb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};                 // P8 slots
    const f32x4 xd0 = *reinterpret_cast<const f32x4*>(xr + 4 * g), xd1 = *reinterpret_cast<const f32x4*>(xr + 16 + 4 * g);
    // q = Wq^T x ; Q = elu(q) + 1 ; Z = 1 / (Q . Ksum + eps)
    f32x4 q[2];
    float zp = 0.f;
#pragma unroll
    for (int m = 0; m < 2; m++) {
      q[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 8; sI++) q[m] = mfma4(sWq[(m * 8 + sI) * 64 + lane], xb[sI], q[m]);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        q[m][r] = elu1(q[m][r]);
        zp += q[m][r] * sKs[16 * m + 4 * g + r];
      }
    }
    const float z = 1.0f / (quad_sum(zp) + 9.999999974752427e-07f);
    // msg = (KV^T Q) * Z * 1200
    f32x4 ms[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      ms[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 8; sI++) ms[m] = mfma4(sKV[(m * 8 + sI) * 64 + lane], q[sI >> 2][sI & 3], ms[m]);
#pragma unroll
      for (int r = 0; r < 4; r++) ms[m][r] = ms[m][r] * z * 1200.0f;
    }
    // merge + LN1
    f32x4 mg[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      mg[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 8; sI++) mg[m] = mfma4(sWm[(m * 8 + sI) * 64 + lane], ms[sI >> 2][sI & 3], mg[m]);
    }
    layer_norm_cols(mg, sLN, sLN + DM, g);
    // MLP on [x | mg]: 64 -> 64 (ReLU) -> 32, LN2, residual
    f32x4 h[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      h[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 8; sI++) h[m] = mfma4(sW0[(m * 16 + sI) * 64 + lane], xb[sI], h[m]);
#pragma unroll
      for (int sI = 0; sI < 8; sI++) h[m] = mfma4(sW0[(m * 16 + 8 + sI) * 64 + lane], mg[sI >> 2][sI & 3], h[m]);
#pragma unroll
      for (int r = 0; r < 4; r++) h[m][r] = fmaxf(h[m][r], 0.f);
    }
    f32x4 o[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      o[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 16; sI++) o[m] = mfma4(sW1[(m * 16 + sI) * 64 + lane], h[sI >> 2][sI & 3], o[m]);
    }
    layer_norm_cols(o, sLN + 2 * DM, sLN + 3 * DM, g);
    float* dr = dst + (long long)seq * d_stride + (long long)(tile * 16 + tl) * DM;
    *reinterpret_cast<f32x4*>(dr + 4 * g) = xd0 + o[0];
    *reinterpret_cast<f32x4*>(dr + 16 + 4 * g) = xd1 + o[1];
  }
}

// The same block on split-bf16 MFMAs.  In k_attn_update's formulation (features on MFMA rows, tokens on columns) the
// activation operand of every product is the set of 8 registers a lane already holds -- features 8g .. 8g+7 of its
// token after the load, or the two accumulator tiles of the previous product -- and 4 lane groups x 8 registers are
// exactly the K = 32 of one v_mfma_f32_16x16x32_bf16.  So each run of 8 f32 MFMAs (K = 4 each) becomes 3 bf16 MFMAs
// (hi.hi, hi.lo, lo.hi) on the same registers, split in place (24 VALU instructions per 8 values); the weights are
// pre-split fragments in LDS (one ds_read_b128 per plane instead of 8 ds_read_b32).  144 f32 MFMAs of 32 cycles per
// 16-token tile -> 54 bf16 MFMAs of 16: the kernel goes from MFMA-bound to VALU-bound (LayerNorms, ELU, splits).
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
  bf16x4 h0, l0, h1, l1;
  split4(f32x4{v[0], v[1], v[2], v[3]}, h0, l0);
  split4(f32x4{v[4], v[5], v[6], v[7]}, h1, l1);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma3x(bf16x8 wh, bf16x8 wl, bf16x8 xh, bf16x8 xl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, c, 0, 0, 0);
}

// The split-bf16 encoder kernels are bound by vector issue (0.74 VALU-busy), and a third of their vector instructions were
// IEEE divisions and libm's expf: a LayerNorm divided each of its 8 values per lane by the standard deviation (ten
// instructions per correctly rounded division), V was divided by 1200 value by value, ELU went through expf's range
// reduction.  The fast forms below multiply by ONE correctly rounded reciprocal and take the exponential as v_exp_f32 of
// x log2(e), the normalisers as v_rsq_f32 / v_rcp_f32: each result within 1-2 ulp of the exact form's (the exact-f32 kernels k_attn_kv / k_attn_update keep the
// graph's operations one for one); the conv stack's split products are the larger error by far.
// features / sqrt(32) as a product with the rounded reciprocal (the split path's fused form in k_attn_update_x; the exact-f32
// path's k_scale_feats divides)
constexpr float kInvSqrtDM = 1.0f / 5.656854f;
// quad_sum on v_permlane16_swap / v_permlane32_swap: the same two additions in the same order (bit-identical) without the
// two LDS round trips of the shuffles, which sat in the dependent chain of every LayerNorm and normaliser
__device__ __forceinline__ float quad_sum_pl(float v) {
  const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);               // lanes l and l ^ 16
  const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);            // ... and l ^ 32
}
__device__ __forceinline__ float elu1_fast(float x) {
  return (x > 0.f ? x : __builtin_amdgcn_exp2f(x * 1.44269504089f) - 1.f) + 1.f;
}
__device__ __forceinline__ void layer_norm_cols_fast(f32x4* v, const float* w, const float* b, int g) {
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) sum += v[m][r];
  const float mean = quad_sum_pl(sum) * (1.f / (float)DM);            // (a power of two: the same value as the division)
  float var = 0.f;
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) { v[m][r] -= mean; var += v[m][r] * v[m][r]; }
  // v_rsq_f32 (1 ulp) for the correctly rounded square root and division: twenty instructions fewer per LayerNorm
  const float rden = __builtin_amdgcn_rsqf(quad_sum_pl(var) * (1.f / (float)DM) + 1.0000000116860974e-07f);
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int f = 16 * m + 4 * g + r;
      v[m][r] = v[m][r] * rden * w[f] + b[f];
    }
}

// r05: a launch may carry TWO independent encoder blocks (the two self-attention blocks of a layer pair: feat0 <- feat0 and
// feat1 <- feat1): workgroups [0, n_first) work on (src, w, kv), the rest on (src2, w2, kv2); n_first = gridDim.x: one block.
__global__ __launch_bounds__(64 * kKvWaves) void k_attn_kv_x(const float* __restrict__ src_a, long long seq_stride, BlockW w_a,
                                                             float* __restrict__ kv_a /*[n][1056]: KV in PD order | Ksum*/,
                                                             int n_first, const float* __restrict__ src_b, BlockW w_b,
                                                             float* __restrict__ kv_b) {
  __shared__ bf16x8 sWk[2 * 2 * 64], sWv[2 * 2 * 64];           // fragments [cout tile][hi | lo][lane]
  __shared__ float red[kKvWaves][DM * DM + DM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = lane & 15, g = lane >> 4;
  const bool second = (int)blockIdx.x >= n_first;                // workgroup-uniform
  const int seq_i = second ? (int)blockIdx.x - n_first : (int)blockIdx.x;
  const float* src = second ? src_b : src_a;
  float* kv = second ? kv_b : kv_a;
  const BlockW w = second ? w_b : w_a;
  const float* s = src + (long long)seq_i * seq_stride;
  constexpr int NT16 = NTOK / 16;                                // 75 token tiles: 37 pairs + one single
  // (the first pair goes out together with the weight loads below)
  // the token rows of the next pair of tiles are requested before the current pair is processed (a sequence is walked
  // by 8 waves in 4-5 dependent steps: the kernel is latency-bound for small batches); clamped, so unconditional
  f32x4 nx[2][2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const float* xr = s + (long long)(min(2 * wave + h, NT16 - 1) * 16 + tl) * DM + 8 * g;
    nx[h][0] = *reinterpret_cast<const f32x4*>(xr);
    nx[h][1] = *reinterpret_cast<const f32x4*>(xr + 4);
  }
  if (tid < 256) {                                               // f32 packing [(n * 8 + slot) * 64 + lane] -> fragments
    const bool isv = tid >= 128;
    const int n = (tid >> 6) & 1, ln = tid & 63;
    const float* wp = isv ? w.wv_p : w.wk_p;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = wp[(n * 8 + j) * 64 + ln];
    bf16x8 hi, lo;
    split8(v, hi, lo);
    (isv ? sWv : sWk)[(n * 2 + 0) * 64 + ln] = hi;
    (isv ? sWv : sWk)[(n * 2 + 1) * 64 + ln] = lo;
  }
  __syncthreads();
  const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[2][2] = {{zero, zero}, {zero, zero}};
  float ksum[2] = {0.f, 0.f};
  for (int t0 = 2 * wave; t0 < NT16; t0 += 2 * kKvWaves) {
    float Kv[2][8], Vv[2][8];                                    // [feature tile][tile of the pair * 4 + r]
    f32x4 cx[2][2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      cx[h][0] = nx[h][0];
      cx[h][1] = nx[h][1];
      const float* xr = s + (long long)(min(t0 + 2 * kKvWaves + h, NT16 - 1) * 16 + tl) * DM + 8 * g;
      nx[h][0] = *reinterpret_cast<const f32x4*>(xr);
      nx[h][1] = *reinterpret_cast<const f32x4*>(xr + 4);
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int tile = t0 + h;
      if (tile < NT16) {
        const f32x4 x0 = cx[h][0], x1 = cx[h][1];
        const float xa[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        bf16x8 xh, xl;
        split8(xa, xh, xl);
#pragma unroll
        for (int n = 0; n < 2; n++) {
          // projections as k_attn_kv: tokens on MFMA rows (A = x), features on columns (B = weights)
          const f32x4 K = mfma3x(xh, xl, sWk[(n * 2 + 0) * 64 + lane], sWk[(n * 2 + 1) * 64 + lane], zero);
          const f32x4 V = mfma3x(xh, xl, sWv[(n * 2 + 0) * 64 + lane], sWv[(n * 2 + 1) * 64 + lane], zero);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            Kv[n][4 * h + r] = elu1_fast(K[r]);
            Vv[n][4 * h + r] = V[r] * (1.0f / 1200.0f);
            ksum[n] += Kv[n][4 * h + r];
          }
        }
      } else {
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
          for (int r = 0; r < 4; r++) { Kv[n][4 * h + r] = 0.f; Vv[n][4 * h + r] = 0.f; }
      }
    }
    bf16x8 Kh[2], Kl[2], Vh[2], Vl[2];
#pragma unroll
    for (int n = 0; n < 2; n++) { split8(Kv[n], Kh[n], Kl[n]); split8(Vv[n], Vh[n], Vl[n]); }
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int n = 0; n < 2; n++) acc[m][n] = mfma3x(Kh[m], Kl[m], Vh[n], Vl[n], acc[m][n]);   // k-slots = the pair's 32 tokens
  }
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[wave][(16 * m + 4 * g + r) * DM + 16 * n + tl] = acc[m][n][r];
#pragma unroll
  for (int n = 0; n < 2; n++) {
    const float t = quad_sum_pl(ksum[n]);
    if (g == 0) red[wave][DM * DM + 16 * n + tl] = t;
  }
  __syncthreads();
  for (int i = tid; i < DM * DM + DM; i += 64 * kKvWaves) {
    float t = red[0][i];
#pragma unroll
    for (int w2 = 1; w2 < kKvWaves; w2++) t += red[w2][i];
    red[0][i] = t;
  }
  __syncthreads();
  float* o = kv + (long long)seq_i * (DM * DM + DM);
  for (int i = tid; i < DM * DM; i += 64 * kKvWaves) {     // KV as the A operand of msg = KV^T Q, PD slot order over d
    const int ln = i & 63, sl = (i >> 6) & 7, me = i >> 9;
    const int d = 16 * (sl >> 2) + 4 * (ln >> 4) + (sl & 3), e = 16 * me + (ln & 15);
    o[i] = red[0][d * DM + e];
  }
  if (tid < DM) o[DM * DM + tid] = red[0][DM * DM + tid];
}

// (r05: as k_attn_kv_x, a launch may carry two independent blocks: workgroups [0, wg_first) take the first argument set)
__global__ __launch_bounds__(256) void k_attn_update_x(const float* __restrict__ xsrc_a, long long x_stride,
                                                       const float* __restrict__ kv_a, BlockW w_a, float* __restrict__ dst_a,
                                                       long long d_stride, int n_items, int items_per_wg,
                                                       float* __restrict__ fs_out, __bf16* __restrict__ pl_out, int wg_first,
                                                       const float* __restrict__ xsrc_b, const float* __restrict__ kv_b,
                                                       BlockW w_b, float* __restrict__ dst_b) {
  const bool second = (int)blockIdx.x >= wg_first;               // workgroup-uniform
  const int wg_i = second ? (int)blockIdx.x - wg_first : (int)blockIdx.x;
  const float* xsrc = second ? xsrc_b : xsrc_a;
  const float* kv = second ? kv_b : kv_a;
  float* dst = second ? dst_b : dst_a;
  const BlockW w = second ? w_b : w_a;
  // fragments: [mtile][K group][hi | lo][lane]
  __shared__ bf16x8 sWq[2 * 2 * 64], sKV[2 * 2 * 64], sWm[2 * 2 * 64], sW0[4 * 2 * 2 * 64], sW1[2 * 2 * 2 * 64];
  __shared__ float sLN[4 * DM], sKs[DM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tl = lane & 15, g = lane >> 4;
  // The block's weights (32 KB of fragments) are the same for every sequence: a workgroup stages them once and then walks
  // a contiguous run of (sequence, group of 8 token tiles) items, re-staging only the sequence's KV when it changes.
  // Runs longer than one item did not pay (the host launches one item per workgroup): the other workgroups of a CU cover
  // a prologue better than a serial run of items amortises it.
  for (int i = tid; i < 2 * 2 * 64; i += 256) {
    sWq[i] = reinterpret_cast<const bf16x8*>(w.wq_x)[i];
    sWm[i] = reinterpret_cast<const bf16x8*>(w.wm_x)[i];
  }
  for (int i = tid; i < 4 * 2 * 2 * 64; i += 256) sW0[i] = reinterpret_cast<const bf16x8*>(w.w0_x)[i];
  for (int i = tid; i < 2 * 2 * 2 * 64; i += 256) sW1[i] = reinterpret_cast<const bf16x8*>(w.w1_x)[i];
  if (tid < DM) {
    sLN[tid] = w.n1w[tid]; sLN[DM + tid] = w.n1b[tid]; sLN[2 * DM + tid] = w.n2w[tid]; sLN[3 * DM + tid] = w.n2b[tid];
  }
  constexpr int kUpdBlocks = (NTOK / 16 + 4 * kUpdTilesPerWave - 1) / (4 * kUpdTilesPerWave);
  const int item_lo = wg_i * items_per_wg, item_hi = min(item_lo + items_per_wg, n_items);
  int cur_seq = -1;
  for (int item = item_lo; item < item_hi; item++) {
  const int seq = item / kUpdBlocks, xblk = item - seq * kUpdBlocks;
  if (seq != cur_seq) {                            // uniform over the workgroup
    cur_seq = seq;
    __syncthreads();                               // every wave is done with the previous sequence's KV
    const float* kvp = kv + (long long)seq * (DM * DM + DM);
    if (tid < 128) {                               // this sequence's KV (f32, [(m * 8 + slot) * 64 + lane]) -> fragments
      const int m = tid >> 6, ln = tid & 63;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = kvp[(m * 8 + j) * 64 + ln];
      bf16x8 hi, lo;
      split8(v, hi, lo);
      sKV[(m * 2 + 0) * 64 + ln] = hi;
      sKV[(m * 2 + 1) * 64 + ln] = lo;
    }
    if (tid < DM) sKs[tid] = kvp[DM * DM + tid];
    __syncthreads();
  }
  for (int it = 0; it < kUpdTilesPerWave; it++) {
    const int tile = (xblk * kUpdTilesPerWave + it) * 4 + wave;
    if (tile >= NTOK / 16) break;
    const float* xr = xsrc + (long long)seq * x_stride + (long long)(tile * 16 + tl) * DM;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(xr + 8 * g), b1 = *reinterpret_cast<const f32x4*>(xr + 8 * g + 4);
    const float xb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};                 // P8 slots
    const f32x4 xd0 = *reinterpret_cast<const f32x4*>(xr + 4 * g), xd1 = *reinterpret_cast<const f32x4*>(xr + 16 + 4 * g);
    bf16x8 xh, xl;
    split8(xb, xh, xl);
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    // q = Wq^T x ; Q = elu(q) + 1 ; Z = 1 / (Q . Ksum + eps)
    float qv[8];
    float zp = 0.f;
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const f32x4 q = mfma3x(sWq[(m * 2 + 0) * 64 + lane], sWq[(m * 2 + 1) * 64 + lane], xh, xl, zero);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        qv[4 * m + r] = elu1_fast(q[r]);
        zp += qv[4 * m + r] * sKs[16 * m + 4 * g + r];
      }
    }
    const float z = __builtin_amdgcn_rcpf(quad_sum_pl(zp) + 9.999999974752427e-07f);      // v_rcp_f32 (1 ulp)
    bf16x8 qh, ql;
    split8(qv, qh, ql);
    // msg = (KV^T Q) * Z * 1200
    float msv[8];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      const f32x4 t = mfma3x(sKV[(m * 2 + 0) * 64 + lane], sKV[(m * 2 + 1) * 64 + lane], qh, ql, zero);
#pragma unroll
      for (int r = 0; r < 4; r++) msv[4 * m + r] = t[r] * z * 1200.0f;
    }
    bf16x8 mh, ml;
    split8(msv, mh, ml);
    // merge + LN1
    f32x4 mg[2];
#pragma unroll
    for (int m = 0; m < 2; m++) mg[m] = mfma3x(sWm[(m * 2 + 0) * 64 + lane], sWm[(m * 2 + 1) * 64 + lane], mh, ml, zero);
    layer_norm_cols_fast(mg, sLN, sLN + DM, g);
    const float mgv[8] = {mg[0][0], mg[0][1], mg[0][2], mg[0][3], mg[1][0], mg[1][1], mg[1][2], mg[1][3]};
    bf16x8 gh, gl;
    split8(mgv, gh, gl);
    // MLP on [x | mg]: 64 -> 64 (ReLU) -> 32, LN2, residual
    float hv[16];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      f32x4 t = mfma3x(sW0[((m * 2 + 0) * 2 + 0) * 64 + lane], sW0[((m * 2 + 0) * 2 + 1) * 64 + lane], xh, xl, zero);
      t = mfma3x(sW0[((m * 2 + 1) * 2 + 0) * 64 + lane], sW0[((m * 2 + 1) * 2 + 1) * 64 + lane], gh, gl, t);
#pragma unroll
      for (int r = 0; r < 4; r++)      // ReLU as a signed-integer maximum: a float maximum of an MFMA result costs a second v_max_f32 (NaN quieting)
        hv[4 * m + r] = __int_as_float(max(__float_as_int(t[r]), 0));
    }
    bf16x8 h0h, h0l, h1h, h1l;
    split8(hv, h0h, h0l);
    split8(hv + 8, h1h, h1l);
    f32x4 o[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      o[m] = mfma3x(sW1[((m * 2 + 0) * 2 + 0) * 64 + lane], sW1[((m * 2 + 0) * 2 + 1) * 64 + lane], h0h, h0l, zero);
      o[m] = mfma3x(sW1[((m * 2 + 1) * 2 + 0) * 64 + lane], sW1[((m * 2 + 1) * 2 + 1) * 64 + lane], h1h, h1l, o[m]);
    }
    layer_norm_cols_fast(o, sLN + 2 * DM, sLN + 3 * DM, g);
    float* dr = dst + (long long)seq * d_stride + (long long)(tile * 16 + tl) * DM;
    const f32x4 r0 = xd0 + o[0], r1 = xd1 + o[1];
    *reinterpret_cast<f32x4*>(dr + 4 * g) = r0;
    *reinterpret_cast<f32x4*>(dr + 16 + 4 * g) = r1;
    if (fs_out) {
      // the last block of a side: the matching head's inputs leave here as well (what k_scale_feats makes of dst in the
      // exact-f32 path: the features / sqrt(32) -- here times the rounded reciprocal -- and their three bf16 planes)
      const long long e0 = (long long)(tile * 16 + tl) * DM;
      float* fr = fs_out + (long long)seq * d_stride + e0;
      __bf16* pr = pl_out + (long long)seq * 3 * d_stride + e0;
      const float rv[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
      float sv[8];
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      bf16x4_t ph[2], pm[2], pl[2];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const float v = rv[q] * kInvSqrtDM;
        sv[q] = v;
        const __bf16 h = (__bf16)v;
        const float e1 = v - (float)h;
        const __bf16 m = (__bf16)e1;
        const __bf16 l = (__bf16)(e1 - (float)m);
        ph[q >> 2][q & 3] = h; pm[q >> 2][q & 3] = m; pl[q >> 2][q & 3] = l;
      }
      *reinterpret_cast<f32x4*>(fr + 4 * g) = f32x4{sv[0], sv[1], sv[2], sv[3]};
      *reinterpret_cast<f32x4*>(fr + 16 + 4 * g) = f32x4{sv[4], sv[5], sv[6], sv[7]};
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int c = hf * 16 + 4 * g;
        *reinterpret_cast<bf16x4_t*>(pr + c) = ph[hf];
        *reinterpret_cast<bf16x4_t*>(pr + d_stride + c) = pm[hf];
        *reinterpret_cast<bf16x4_t*>(pr + 2 * d_stride + c) = pl[hf];
      }
    }
  }
  }
}

// ------------------------------------------------------------------ matching head (MFMA)
// s_ij = ((f0_i / sqrt(32)) . (f1_j / sqrt(32))) / 0.1 on 16x16 tiles, P8 slot order on both operands.
// planes (split path): the scaled features also as three bf16 planes per pair, v = h + m + l exactly (24 significand
// bits in 3 x 8): [pair][3][1200 * 32].  k_sim_stats<.., true> then forms s = f0 . f1 from six bf16 MFMAs
// (h.h, h.m, m.h, h.l, l.h, m.m: everything above 2^-24 of the product) with no split arithmetic in its inner loop.
__global__ __launch_bounds__(256) void k_scale_feats(const float* __restrict__ in, float* __restrict__ out, long long n,
                                                     __bf16* __restrict__ planes, long long ts) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = in[i] / 5.656854f;
  out[i] = v;
  if (planes) {
    const long long pair = i / ts, e = i - pair * ts;
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const __bf16 l = (__bf16)(r1 - (float)m);
    __bf16* dst = planes + pair * 3 * ts + e;
    dst[0] = h; dst[ts] = m; dst[2 * ts] = l;
  }
}

// x / 0.1f, correctly rounded, in three instructions: 10.0f is the correctly rounded reciprocal of 0.1f, so one
// Newton step on the exact fma residual gives RN(x / 0.1f) (Markstein); checked bit for bit against IEEE division on
// 1.2e8 values across the magnitudes that occur.  The generic correctly rounded division costs ~10 instructions, four
// times per 16 x 16 tile, in kernels where VALU and MFMA time are comparable.
__device__ __forceinline__ float div_temperature(float x) {
  const float q0 = x * 10.0f;
  const float r = __builtin_fmaf(-q0, 0.1f, x);
  return __builtin_fmaf(r, 10.0f, q0);
}

__device__ __forceinline__ void load8(const float* p, float* v) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// exp is the hardware v_exp_f32 path (__expf, ~2 ulp): confidences move by < 1e-6, far inside the 1e-3 bar.
// Row statistics (max, sum of exp) of S = A B^T / 0.1; called with (f0s, f1s) for the rows and with (f1s, f0s) for
// the columns: the products commute and are summed in the same order, so both calls see bit-identical s_ij.
// EMIT (the column call, when row statistics exist): conf_ij <= softmax_j(s)_ij = exp(s - rm_i) / rsum_i, so only
// entries with s >= lim_i = rm_i + log(threshold * rsum_i) - margin can pass the threshold -- at most 1 / (0.99 thr)
// per row.  They are appended to a per-pair candidate list (i, j, s) and evaluated exactly by k_conf_cand once the
// column statistics are complete; the similarity GEMM is not recomputed a third time.
constexpr int kCandPerRow = 21;                       // >= 1 / (0.99 * kCandMinThreshold)
constexpr float kCandMinThreshold = 0.05f;            // below this the dense k_conf_mask pass is used
constexpr int kCandCap = NTOK * kCandPerRow;          // per pair; cannot overflow for thresholds >= kCandMinThreshold
struct SimCand { uint32_t ij; float s; };

__global__ __launch_bounds__(256) void k_row_limits(const float* __restrict__ rstats, long long stats_stride,
                                                    float threshold, float* __restrict__ lim, int n_pairs,
                                                    const uint32_t* __restrict__ only_if) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_pairs * NTOK) return;
  const int pair = idx / NTOK, i = idx - pair * NTOK;
  if (only_if && !only_if[pair]) return;
  const float* rs = rstats + (long long)pair * stats_stride;
  lim[idx] = rs[i] + __logf(threshold * rs[NTOK + i]) - 1e-2f;
}

template <bool EMIT, bool SPLIT>
__global__ __launch_bounds__(256) void k_sim_stats(const float* __restrict__ fa, const float* __restrict__ fb,
                                                   long long pair_stride, float* __restrict__ stats /*[pair][2][1200]*/,
                                                   long long stats_stride, const float* __restrict__ lim,
                                                   SimCand* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
                                                   const __bf16* __restrict__ pa, const __bf16* __restrict__ pb,
                                                   const uint32_t* __restrict__ only_if) {
  const int pair = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 15, g = lane >> 4;
  const int it = blockIdx.x * 4 + wave;
  if (it >= NTOK / 16) return;
  if (only_if && !only_if[pair]) return;      // the single-pass statistics held for this pair (k_sim_finish)
  const float* A = fa + (long long)pair * pair_stride;
  const float* B = fb + (long long)pair * pair_stride;
  float a[8];
  bf16x8 a3[3];
  const bf16x8* PA = reinterpret_cast<const bf16x8*>(pa + (SPLIT ? (long long)pair * 3 * pair_stride : 0));
  const bf16x8* PB = reinterpret_cast<const bf16x8*>(pb + (SPLIT ? (long long)pair * 3 * pair_stride : 0));
  constexpr int kPlane = NTOK * DM / 8;                  // bf16x8 fragments per plane
  if (SPLIT) {
#pragma unroll
    for (int pl = 0; pl < 3; pl++) a3[pl] = PA[pl * kPlane + (it * 16 + tl) * (DM / 8) + g];
  } else {
    load8(A + (long long)(it * 16 + tl) * DM + 8 * g, a);
  }
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sm[4] = {0.f, 0.f, 0.f, 0.f};
  // three column tiles per step: three independent MFMA chains in flight, and the running (max, sum) of a row is
  // rescaled once per three new entries (4 exponentials per 3 entries, no divergent branch)
  constexpr int TJ = 3;
  static_assert((NTOK / 16) % TJ == 0, "column tiles per step");
  for (int jt0 = 0; jt0 < NTOK / 16; jt0 += TJ) {
    f32x4 d[TJ];
#pragma unroll
    for (int u = 0; u < TJ; u++) {
      d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (SPLIT) {
        bf16x8 b3[3];
#pragma unroll
        for (int pl = 0; pl < 3; pl++) b3[pl] = PB[pl * kPlane + ((jt0 + u) * 16 + tl) * (DM / 8) + g];
        // K = 32 is one MFMA: the six products above 2^-24, smallest first
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[2], b3[0], d[u], 0, 0, 0);
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[0], b3[2], d[u], 0, 0, 0);
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[1], b3[1], d[u], 0, 0, 0);
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[1], b3[0], d[u], 0, 0, 0);
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[0], b3[1], d[u], 0, 0, 0);
        d[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[0], b3[0], d[u], 0, 0, 0);
      } else {
        float b[8];
        load8(B + (long long)((jt0 + u) * 16 + tl) * DM + 8 * g, b);
#pragma unroll
        for (int sI = 0; sI < 8; sI++) d[u] = mfma4(a[sI], b[sI], d[u]);
      }
    }
    float sv[TJ][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float mn = mx[r];
#pragma unroll
      for (int u = 0; u < TJ; u++) {
        sv[u][r] = div_temperature(d[u][r]);
        mn = fmaxf(mn, sv[u][r]);
      }
      float add = 0.f;
#pragma unroll
      for (int u = 0; u < TJ; u++) add += __expf(sv[u][r] - mn);
      sm[r] = sm[r] * __expf(mx[r] - mn) + add;
      mx[r] = mn;
    }
    if (EMIT) {
#pragma unroll
      for (int u = 0; u < TJ; u++) {
        // here the tile is S^T: this lane holds s_ij for i = jt*16 + tl (the row of S) and j = it*16 + 4g + r
        const int i = (jt0 + u) * 16 + tl;
        const float li = lim[(long long)pair * NTOK + i];
        const bool h0 = sv[u][0] >= li, h1 = sv[u][1] >= li, h2 = sv[u][2] >= li, h3 = sv[u][3] >= li;
        if (__any(h0 | h1 | h2 | h3)) {
          const uint32_t nh = (uint32_t)h0 + h1 + h2 + h3;
          if (nh) {
            uint32_t k = atomicAdd(&cand_cnt[pair], nh);
            SimCand* c = cand + (long long)pair * kCandCap;
            const bool hs[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (hs[r]) {
                if (k < (uint32_t)kCandCap) c[k] = SimCand{(uint32_t)i | ((uint32_t)(it * 16 + 4 * g + r) << 16), sv[u][r]};
                k++;
              }
          }
        }
      }
    }
  }
  // combine the 16 lanes (columns) that share each row
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float m = mx[r], s = sm[r];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
      const float mm = fmaxf(m, m2);
      s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
      m = mm;
    }
    if (tl == 0) {
      float* st = stats + (long long)pair * stats_stride;
      st[it * 16 + 4 * g + r] = m;
      st[NTOK + it * 16 + 4 * g + r] = s;
    }
  }
}

// The split path's form of k_sim_stats: a wave carries THREE row tiles (75 = 25 x 3) through the column loop, so every
// column-tile fragment is fetched once per three tiles of S -- the one-tile-per-wave form re-reads f1's planes 75 times
// per pair through L2 (4.3 GB per pass, ~14 TB/s aggregate: the pass was bound there, not by its soft-max VALU work).
constexpr int kSimRT = 3;
constexpr int kSimWaves = 5;                              // 5 waves x 5 workgroups = the 25 row-tile triples of a pair
template <bool EMIT>
__global__ __launch_bounds__(64 * kSimWaves) void k_sim_stats3(const __bf16* __restrict__ pa, const __bf16* __restrict__ pb,
                                                               long long pair_stride, float* __restrict__ stats,
                                                               long long stats_stride, const float* __restrict__ lim,
                                                               SimCand* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
                                                               const uint32_t* __restrict__ only_if) {
  const int pair = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 15, g = lane >> 4;
  const int it0 = (blockIdx.x * kSimWaves + wave) * kSimRT;
  if (it0 >= NTOK / 16) return;
  if (only_if && !only_if[pair]) return;
  static_assert((NTOK / 16) % kSimRT == 0, "row tiles per wave");
  const bf16x8* PA = reinterpret_cast<const bf16x8*>(pa + (long long)pair * 3 * pair_stride);
  const bf16x8* PB = reinterpret_cast<const bf16x8*>(pb + (long long)pair * 3 * pair_stride);
  constexpr int kPlane = NTOK * DM / 8;
  bf16x8 a3[kSimRT][3];
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int pl = 0; pl < 3; pl++) a3[t][pl] = PA[pl * kPlane + ((it0 + t) * 16 + tl) * (DM / 8) + g];
  float mx[kSimRT][4], sm[kSimRT][4];
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) { mx[t][r] = -INFINITY; sm[t][r] = 0.f; }
  constexpr int TJ = 3;
  for (int jt0 = 0; jt0 < NTOK / 16; jt0 += TJ) {
    float sv[kSimRT][TJ][4];
#pragma unroll
    for (int u = 0; u < TJ; u++) {
      bf16x8 b3[3];
#pragma unroll
      for (int pl = 0; pl < 3; pl++) b3[pl] = PB[pl * kPlane + ((jt0 + u) * 16 + tl) * (DM / 8) + g];
#pragma unroll
      for (int t = 0; t < kSimRT; t++) {
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][2], b3[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[2], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[0], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[1], d, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[0], d, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) sv[t][u][r] = div_temperature(d[r]);
      }
    }
#pragma unroll
    for (int t = 0; t < kSimRT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float mn = mx[t][r];
#pragma unroll
        for (int u = 0; u < TJ; u++) mn = fmaxf(mn, sv[t][u][r]);
        float add = 0.f;
#pragma unroll
        for (int u = 0; u < TJ; u++) add += __expf(sv[t][u][r] - mn);
        sm[t][r] = sm[t][r] * __expf(mx[t][r] - mn) + add;
        mx[t][r] = mn;
      }
    if (EMIT) {
#pragma unroll
      for (int u = 0; u < TJ; u++) {
        // the tile is S^T: this lane holds s_ij for i = jt*16 + tl (the row of S) and j = it*16 + 4g + r
        const int i = (jt0 + u) * 16 + tl;
        const float li = lim[(long long)pair * NTOK + i];
#pragma unroll
        for (int t = 0; t < kSimRT; t++) {
          const bool h0 = sv[t][u][0] >= li, h1 = sv[t][u][1] >= li, h2 = sv[t][u][2] >= li, h3 = sv[t][u][3] >= li;
          if (__any(h0 | h1 | h2 | h3)) {
            const uint32_t nh = (uint32_t)h0 + h1 + h2 + h3;
            if (nh) {
              uint32_t k = atomicAdd(&cand_cnt[pair], nh);
              SimCand* c = cand + (long long)pair * kCandCap;
              const bool hs[4] = {h0, h1, h2, h3};
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (hs[r]) {
                  if (k < (uint32_t)kCandCap) c[k] = SimCand{(uint32_t)i | ((uint32_t)((it0 + t) * 16 + 4 * g + r) << 16), sv[t][u][r]};
                  k++;
                }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float m = mx[t][r], sx = sm[t][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(sx, o);
        const float mm = fmaxf(m, m2);
        sx = sx * __expf(m - mm) + s2 * __expf(m2 - mm);
        m = mm;
      }
      if (tl == 0) {
        float* st = stats + (long long)pair * stats_stride;
        st[(it0 + t) * 16 + 4 * g + r] = m;
        st[NTOK + (it0 + t) * 16 + 4 * g + r] = sx;
      }
    }
}

// ---- row AND column statistics from ONE evaluation of S (split path, batches; r03).
// The two passes above each spend 13 VALU cycles per entry on the running-maximum soft-max (two exponentials per
// three entries' rescale, the max, the exact division by the temperature) and are bound by that, not by their MFMAs.
// With ONE offset G per pair -- G = 10 max|f0s_i| max|f1s_j| >= every s_ij (Cauchy-Schwarz) -- e_ij = exp(s_ij - G)
// serves both soft-maxes: R_i = sum_j e_ij, C_j = sum_i e_ij, and the statistics the consumers read are (G, R_i) and
// (G, C_j): exp(s - G) / R_i is the same row soft-max.  One fma + one v_exp + two adds per entry.  What the offset
// costs is range: an entry more than ~87 below G underflows.  LoFTR's logits live in [-2, 3] with G ~ 28, but nothing
// bounds them in principle, so k_sim_finish checks every sum and flags a pair whose sums left [1e-30, 1e30]; a
// flagged pair is redone by the running-maximum passes (which skip every other pair).  Sums in a fixed order: a wave
// owns three row tiles over a third of the column tiles, leaves its row sums in rpart[pair][third][1200] and its column
// partials in cpart[pair][triple 0..24][1200]; k_sim_finish adds the 3 and the 25 in order.  Deterministic, no atomics.
// r05: k_sim_single leaves, per item and column tile, the largest entry of the tile's 48 x 16 -- as the exp(s - G) it forms
// anyway: positive floats, so integer maxima on their bits --; k_sim_cand3 -- the same items over the same tiles -- then
// evaluates only the tiles whose maximum can reach the smallest candidate limit of its rows.  A candidate needs
// d >= ld(row); the two kernels sum the same six products in another order, so their d differ by a few ulp: a tile is
// skipped only if its maximum is below what d = smallest limit - 1e-5 would give (1.4e-4 relative; v_exp_f32 is good to
// 1e-7).  An exp that underflows to 0 makes every tile pass.  With ~100 matches per pair four tiles in five hold no candidate.
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]: lane ^ 1
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]: lane ^ 2
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));    // row_half_mirror: the other quad of the 8
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));    // row_mirror: the other 8 of the 16
  const auto r16 = __builtin_amdgcn_permlane16_swap((uint32_t)v, (uint32_t)v, false, false);
  v = max((int)r16[0], (int)r16[1]);
  const auto r32 = __builtin_amdgcn_permlane32_swap((uint32_t)v, (uint32_t)v, false, false);
  return max((int)r32[0], (int)r32[1]);
}
constexpr int kSimParts = NTOK / 16 / kSimRT;         // 25 row-tile triples per pair
#define MSF_LOFTR_SIM_COLPARTS 3
constexpr int kSimColParts = MSF_LOFTR_SIM_COLPARTS;  // an item covers 1 / kSimColParts of the column tiles (1, 3, 5 measured)
constexpr int kSimColTiles = NTOK / 16 / kSimColParts;
static_assert(kSimColTiles * kSimColParts == NTOK / 16, "column thirds");
// Work items (pair, column third, row-tile triple), the triple running fastest, four per workgroup: a workgroup of FIVE
// waves (k_sim_stats3's shape) puts two of its waves on one SIMD, and at four waves per SIMD (104 VGPRs) only two such
// workgroups fit a CU -- 10 of 16 wave slots; measured occupancy of the 5-wave form: 46 %.
constexpr int kSimItemWaves = 4;
constexpr int kSimItemsPerPair = kSimColParts * kSimParts;    // 75
__global__ __launch_bounds__(256) void k_pair_bound(const float* __restrict__ f0s, const float* __restrict__ f1s,
                                                    long long pair_stride, float* __restrict__ gbound) {
  __shared__ float red[2][4];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m[2] = {0.f, 0.f};
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const float* F = (side ? f1s : f0s) + (long long)pair * pair_stride;
    for (int tkn = tid; tkn < NTOK; tkn += 256) {
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < DM; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(F + (long long)tkn * DM + c);
        q += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      m[side] = fmaxf(m[side], q);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m[side] = fmaxf(m[side], __shfl_xor(m[side], o));
    if (lane == 0) red[side][wave] = m[side];
  }
  __syncthreads();
  if (tid == 0) {
    const float a = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float b = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    // a little above 10 |a| |b|: the bound only has to be near the logits, its rounding is of no consequence
    gbound[pair] = 10.0001f * sqrtf(a) * sqrtf(b);
  }
}

__global__ __launch_bounds__(64 * kSimItemWaves) void k_sim_single(const __bf16* __restrict__ pa, const __bf16* __restrict__ pb,
                                                               long long pair_stride, const float* __restrict__ gbound,
                                                               float* __restrict__ rpart, float* __restrict__ cpart,
                                                               int n_items, int* __restrict__ tmax) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kSimItemWaves + wave;
  if (item >= n_items) return;
  const int pair = item / kSimItemsPerPair, rem = item - pair * kSimItemsPerPair;
  const int zpart = rem / kSimParts, part = rem - zpart * kSimParts, it0 = part * kSimRT;
  const bf16x8* PA = reinterpret_cast<const bf16x8*>(pa + (long long)pair * 3 * pair_stride);
  const bf16x8* PB = reinterpret_cast<const bf16x8*>(pb + (long long)pair * 3 * pair_stride);
  constexpr int kPlane = NTOK * DM / 8;
  bf16x8 a3[kSimRT][3];
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int pl = 0; pl < 3; pl++) a3[t][pl] = PA[pl * kPlane + ((it0 + t) * 16 + tl) * (DM / 8) + g];
  const float G = gbound[pair];
  // exp(d / 0.1 - G) = 2^(d * 10 log2(e) - G log2(e))
  const float c1 = 14.4269504089f, c0 = -G * 1.44269504089f;
  float rs[kSimRT][4];
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) rs[t][r] = 0.f;
  float* cp = cpart + ((long long)pair * kSimParts + part) * NTOK;
  // the column tile's three operand planes are requested one tile ahead: a step is 18 MFMAs and 12 exponentials, about
  // the L2 latency it would otherwise wait for at four waves per SIMD
  bf16x8 bn[3];
  // zpart = which third of the column tiles: 256 pairs are 6 400 (row-tile triple, all columns) items for 4 096 wave
  // slots, i.e. two rounds with the second 56 % full; in thirds it is 4.7 rounds of a third (1.67 instead of 2)
  const int jt_lo = zpart * kSimColTiles, jt_hi = jt_lo + kSimColTiles;
#pragma unroll
  for (int pl = 0; pl < 3; pl++) bn[pl] = PB[pl * kPlane + (jt_lo * 16 + tl) * (DM / 8) + g];
  for (int jt = jt_lo; jt < jt_hi; jt++) {
    bf16x8 b3[3];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) b3[pl] = bn[pl];
    const int jn = jt + 1 < jt_hi ? jt + 1 : jt;
#pragma unroll
    for (int pl = 0; pl < 3; pl++) bn[pl] = PB[pl * kPlane + (jn * 16 + tl) * (DM / 8) + g];
    float col = 0.f;
    int emax = 0;                                 // largest exp(s - G) of the tile: positive floats order like their bits
#pragma unroll
    for (int t = 0; t < kSimRT; t++) {
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][2], b3[0], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[2], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[0], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[0], d, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(d[r], c1, c0));
        rs[t][r] += e;
        col += e;
        emax = max(emax, __float_as_int(e));
      }
    }
    if (tmax) {                                   // (uniform) the tile's largest entry, for k_sim_cand3's skip
      const int km = wave_max_i32(emax);
      if (lane == 0) tmax[(long long)item * kSimColTiles + (jt - jt_lo)] = km;
    }
    // this lane's column jt * 16 + tl over the 12 rows it holds; the other three lane groups hold the other rows
    // (v_permlane16/32_swap: the cross-row sums stay in the VALU; as shuffles they are two LDS round trips per column
    // tile in the wave's critical path)
    {
      const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(col), __float_as_uint(col), false, false);
      col = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);               // lanes l and l ^ 16
      const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(col), __float_as_uint(col), false, false);
      col = __uint_as_float(r32[0]) + __uint_as_float(r32[1]);               // ... and l ^ 32: all four lane groups
    }
    if (g == 0) cp[jt * 16 + tl] = col;
  }
  float* rp = rpart + ((long long)pair * kSimColParts + zpart) * NTOK;
#pragma unroll
  for (int t = 0; t < kSimRT; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float v = rs[t][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
      if (tl == 0) rp[(it0 + t) * 16 + 4 * g + r] = v;
    }
}

// column sums from the 25 partials (fixed order), the candidate bound of every row, and the range check
__global__ __launch_bounds__(256) void k_sim_finish(const float* __restrict__ gbound, float* __restrict__ rstats,
                                                    const float* __restrict__ rpart,
                                                    float* __restrict__ cstats, long long stats_stride,
                                                    const float* __restrict__ cpart, float threshold,
                                                    float* __restrict__ lim, uint32_t* __restrict__ redo, int n_pairs,
                                                    int force_redo) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_pairs * NTOK) return;
  const int pair = idx / NTOK, i = idx - pair * NTOK;
  const float* cp = cpart + (long long)pair * kSimParts * NTOK + i;
  float c = 0.f;
#pragma unroll 5
  for (int w = 0; w < kSimParts; w++) c += cp[(long long)w * NTOK];
  const float G = gbound[pair];
  float* cs = cstats + (long long)pair * stats_stride;
  cs[i] = G;
  cs[NTOK + i] = c;
  const float* rp = rpart + (long long)pair * kSimColParts * NTOK + i;
  float r = rp[0];
#pragma unroll
  for (int z = 1; z < kSimColParts; z++) r += rp[(long long)z * NTOK];
  float* rsx = rstats + (long long)pair * stats_stride;
  rsx[i] = G;
  rsx[NTOK + i] = r;
  if (lim) lim[idx] = G + __logf(threshold * r) - 1e-2f;
  const bool ok = c > 1e-30f && c < 1e30f && r > 1e-30f && r < 1e30f;     // false for NaN as well
  if (!ok || force_redo) atomicOr(&redo[pair], 1u);
}

// The candidates of the pairs whose single-pass statistics held: entries with s_ij >= lim_i (see k_row_limits; the same
// predicate on the same s as the emitting k_sim_stats, so the same set).  A wave owns three row tiles of S (their twelve
// bounds per lane stay in registers) and walks the column tiles; the products are compared in d units first (a slack of
// 1e-3 covers the rounding of the division) and only a tile with a hit takes the exact division and the append.  Appends go
// to a wave-private LDS list (ballot + prefix, no atomics) that is flushed with ONE global atomic: the per-lane global
// atomics of the emitting k_sim_stats -- a memory round trip per tile with a hit, in the wave's critical path -- are
// what made its three-tiles-per-wave form slower than one tile per wave.
constexpr int kCandBuf = 512;                         // entries per wave; a step adds at most 9 x 256
__global__ __launch_bounds__(64 * kSimItemWaves) void k_sim_cand3(const __bf16* __restrict__ pa, const __bf16* __restrict__ pb,
                                                              long long pair_stride, const float* __restrict__ lim,
                                                              SimCand* __restrict__ cand, uint32_t* __restrict__ cand_cnt,
                                                              const uint32_t* __restrict__ skip_if, int n_items,
                                                              const int* __restrict__ tmax, const float* __restrict__ gbound) {
  __shared__ SimCand buf_s[kSimItemWaves][kCandBuf];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 15, g = lane >> 4;
  const int item = blockIdx.x * kSimItemWaves + wave;
  if (item >= n_items) return;
  const int pair = item / kSimItemsPerPair, rem = item - pair * kSimItemsPerPair;
  const int zpart = rem / kSimParts, it0 = (rem - zpart * kSimParts) * kSimRT;
  if (skip_if && skip_if[pair]) return;        // flagged: the running-maximum passes list this pair's candidates
  const bf16x8* PA = reinterpret_cast<const bf16x8*>(pa + (long long)pair * 3 * pair_stride);
  const bf16x8* PB = reinterpret_cast<const bf16x8*>(pb + (long long)pair * 3 * pair_stride);
  constexpr int kPlane = NTOK * DM / 8;
  bf16x8 a3[kSimRT][3];
  float li[kSimRT][4], ld[kSimRT][4];
#pragma unroll
  for (int t = 0; t < kSimRT; t++) {
#pragma unroll
    for (int pl = 0; pl < 3; pl++) a3[t][pl] = PA[pl * kPlane + ((it0 + t) * 16 + tl) * (DM / 8) + g];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      li[t][r] = lim[(long long)pair * NTOK + (it0 + t) * 16 + 4 * g + r];
      ld[t][r] = (li[t][r] - 1e-3f) * 0.1f - 1e-6f;
    }
  }
  SimCand* buf = buf_s[wave];
  SimCand* out = cand + (long long)pair * kCandCap;
  uint32_t nb = 0;                               // wave-uniform
  auto flush = [&]() {
    if (nb == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&cand_cnt[pair], nb);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (uint32_t k = lane; k < nb; k += 64)
      if (base + k < (uint32_t)kCandCap) out[base + k] = buf[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    nb = 0;
  };
  bf16x8 bn[3];
  const int jt_lo = zpart * kSimColTiles, jt_hi = jt_lo + kSimColTiles;            // a third of the columns (see k_sim_single)
  // the column tiles this item has to look at: all of them, or (tmax) those whose largest d can reach the smallest limit
  // of the item's 48 rows
  static_assert(kSimColTiles <= 64, "one mask bit per column tile");
  unsigned long long todo = kSimColTiles == 64 ? ~0ull : ((1ull << kSimColTiles) - 1ull);
  if (tmax) {
    float lmin = INFINITY;
#pragma unroll
    for (int t = 0; t < kSimRT; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) lmin = fminf(lmin, ld[t][r]);
    // k_sim_single's measure of an entry: exp(d / 0.1 - G) as 2^(d c1 + c0), monotonic in d
    const float emin = __builtin_amdgcn_exp2f(__builtin_fmaf(lmin - 1e-5f, 14.4269504089f, -gbound[pair] * 1.44269504089f));
    const int kmin = -wave_max_i32(-__float_as_int(emin));               // wave minimum (positive floats order like their bits)
    const int km = lane < kSimColTiles ? tmax[(long long)item * kSimColTiles + lane] : -1;
    todo &= __ballot(km >= kmin);
  }
  if (todo == 0ull) return;                      // (nothing buffered yet)
  int jt = jt_lo + (int)__builtin_ctzll(todo);
  todo &= todo - 1ull;
#pragma unroll
  for (int pl = 0; pl < 3; pl++) bn[pl] = PB[pl * kPlane + (jt * 16 + tl) * (DM / 8) + g];
  for (;;) {
    bf16x8 b3[3];
#pragma unroll
    for (int pl = 0; pl < 3; pl++) b3[pl] = bn[pl];
    const int jn = todo ? jt_lo + (int)__builtin_ctzll(todo) : jt;      // the next tile to look at, requested one ahead
#pragma unroll
    for (int pl = 0; pl < 3; pl++) bn[pl] = PB[pl * kPlane + (jn * 16 + tl) * (DM / 8) + g];
#pragma unroll
    for (int t = 0; t < kSimRT; t++) {
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      // the product order of the emitting k_sim_stats call (operands swapped there): the same s, bit for bit
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[2], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][2], b3[0], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[1], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][1], b3[0], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[t][0], b3[0], d, 0, 0, 0);
      const bool reach = d[0] >= ld[t][0] || d[1] >= ld[t][1] || d[2] >= ld[t][2] || d[3] >= ld[t][3];
      if (__any(reach)) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float sv = div_temperature(d[r]);
          const bool hit = sv >= li[t][r];
          const unsigned long long bal = __ballot(hit);
          if (bal) {
            if (hit) {
              const uint32_t k = nb + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
              buf[k] = SimCand{(uint32_t)((it0 + t) * 16 + 4 * g + r) | ((uint32_t)(jt * 16 + tl) << 16), sv};
            }
            nb += (uint32_t)__popcll(bal);
          }
        }
        if (nb > (uint32_t)(kCandBuf - 256)) flush();
      }
    }
    if (todo == 0ull) break;
    jt = jn;
    todo &= todo - 1ull;
  }
  flush();
}

// working copies of cached per-frame tokens for the pairs (slot_a[i], slot_b[i]): z = 0 -> f0, z = 1 -> f1
__global__ __launch_bounds__(256) void k_gather_tokens(const float* __restrict__ cache, const int32_t* __restrict__ slot_a,
                                                       const int32_t* __restrict__ slot_b, int n_slots,
                                                       float* __restrict__ f0, float* __restrict__ f1) {
  const int i = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
  if (i >= NTOK * DM / 4) return;
  int slot = blockIdx.z ? slot_b[pair] : slot_a[pair];
  slot = min(max(slot, 0), n_slots - 1);     // a slot outside the caller's range: some valid tokens, the pair is marked below
  const f32x4* src = reinterpret_cast<const f32x4*>(cache + (long long)slot * NTOK * DM);
  f32x4* dst = reinterpret_cast<f32x4*>((blockIdx.z ? f1 : f0) + (long long)pair * NTOK * DM);
  dst[i] = src[i];
}

// slot arrays come from the caller's device memory and cannot be checked on the host: a pair with a slot outside
// [0, slot_limit) reports n_out = -1 (msf_abi.h), whatever the clamped gather above made of it
__global__ __launch_bounds__(256) void k_mark_bad_slots(int n_pairs, const int32_t* __restrict__ slot_a,
                                                        const int32_t* __restrict__ slot_b, int slot_limit, int32_t* n_out) {
  const int pair = blockIdx.x * 256 + threadIdx.x;
  if (pair >= n_pairs) return;
  const int a = slot_a[pair], b = slot_b[pair];
  if (a < 0 || a >= slot_limit || b < 0 || b >= slot_limit) n_out[pair] = -1;
}

// conf_ij = softmax_i(s)_ij * softmax_j(s)_ij, '> threshold' -> bit mask (16-bit chunk per row and column tile);
// the 5.76 MB confidence matrix is never written (except for the debug pair).
__global__ __launch_bounds__(256) void k_conf_mask(const float* __restrict__ f0s, const float* __restrict__ f1s,
                                                   long long pair_stride, const float* __restrict__ rstats,
                                                   const float* __restrict__ cstats, long long stats_stride,
                                                   float threshold, uint32_t* __restrict__ mask, float* conf_dbg,
                                                   int dbg_pair) {
  const int pair = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tl = lane & 15, g = lane >> 4;
  const int it = blockIdx.x * 4 + wave;
  if (it >= NTOK / 16) return;
  const float* A = f0s + (long long)pair * pair_stride;
  const float* B = f1s + (long long)pair * pair_stride;
  const float* rs = rstats + (long long)pair * stats_stride;
  const float* cs = cstats + (long long)pair * stats_stride;
  uint16_t* mk = reinterpret_cast<uint16_t*>(mask + (long long)pair * NTOK * MASK_WORDS);
  float* dbg = (conf_dbg && pair == dbg_pair) ? conf_dbg : nullptr;
  float a[8], rm[4], rsum[4];
  load8(A + (long long)(it * 16 + tl) * DM + 8 * g, a);
#pragma unroll
  for (int r = 0; r < 4; r++) { rm[r] = rs[it * 16 + 4 * g + r]; rsum[r] = rs[NTOK + it * 16 + 4 * g + r]; }
  // conf <= softmax_j(s)_ij = exp(s - rm) / rsum, so s < rm + log(threshold * rsum) (minus a margin that covers
  // the rounding of this bound) can never pass: tiles where no lane reaches the bound skip the exact evaluation.
  float lim[4];
#pragma unroll
  for (int r = 0; r < 4; r++)
    lim[r] = threshold > 0.f ? rm[r] + __logf(threshold * rsum[r]) - 1e-2f : -__builtin_inff();
  for (int jt = 0; jt < NTOK / 16; jt++) {
    float b[8];
    const int j = jt * 16 + tl;
    load8(B + (long long)j * DM + 8 * g, b);
    const float cm = cs[j], csum = cs[NTOK + j];
    f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sI = 0; sI < 8; sI++) d = mfma4(a[sI], b[sI], d);
    const bool reach = d[0] * 10.f >= lim[0] || d[1] * 10.f >= lim[1] || d[2] * 10.f >= lim[2] || d[3] * 10.f >= lim[3];
    if (!dbg && !__any(reach)) {
      if (tl == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) mk[(long long)(it * 16 + 4 * g + r) * (2 * MASK_WORDS) + jt] = 0;
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float s = div_temperature(d[r]);
      const float conf = (__expf(s - cm) / csum) * (__expf(s - rm[r]) / rsum[r]);
      const unsigned long long bal = __ballot(conf > threshold);   // strict '>' (dnnfeaturematcher.cpp:75)
      if (tl == 0) mk[(long long)(it * 16 + 4 * g + r) * (2 * MASK_WORDS) + jt] = (uint16_t)(bal >> (16 * g));
      if (dbg) dbg[(long long)(it * 16 + 4 * g + r) * NTOK + j] = conf;
    }
  }
}

// Exact confidence of the candidates (same expression and operand values as k_conf_mask, so the same bits) and the
// '> threshold' bit of each into the (zeroed) mask.
__global__ __launch_bounds__(256) void k_conf_cand(const SimCand* __restrict__ cand, const uint32_t* __restrict__ cand_cnt,
                                                   const float* __restrict__ rstats, const float* __restrict__ cstats,
                                                   long long stats_stride, float threshold, uint32_t* __restrict__ mask) {
  const int pair = blockIdx.y;
  const uint32_t n = min(cand_cnt[pair], (uint32_t)kCandCap);
  const SimCand* c = cand + (long long)pair * kCandCap;
  const float* rs = rstats + (long long)pair * stats_stride;
  const float* cs = cstats + (long long)pair * stats_stride;
  uint32_t* mk = mask + (long long)pair * NTOK * MASK_WORDS;
  for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const SimCand e = c[k];
    const int i = e.ij & 0xFFFF, j = e.ij >> 16;
    const float s = e.s;
    const float conf = (__expf(s - cs[j]) / cs[NTOK + j]) * (__expf(s - rs[i]) / rs[NTOK + i]);
    if (conf > threshold) atomicOr(&mk[i * MASK_WORDS + (j >> 5)], 1u << (j & 31));
  }
}

// findNonZero row-major + decode (dnnfeaturematcher.cpp:80-99): one workgroup per pair.  Every thread owns a contiguous
// run of kDecPer mask words and requests all of them at once (one memory round trip for the 182 KB mask instead of a
// load + barrier per 4 KB chunk: 57 -> 8 us for a single pair, where the kernel is pure latency); one scan of the
// per-thread bit counts gives each thread its position in the row-major output.
constexpr int kDecThreads = 1024;
constexpr int kDecPer = ((NTOK * MASK_WORDS + kDecThreads - 1) / kDecThreads + 3) & ~3;   // words per thread, whole uint4s
static_assert((NTOK * MASK_WORDS) % 4 == 0, "mask is read four words at a time");
__global__ __launch_bounds__(kDecThreads) void k_decode(const uint32_t* __restrict__ mask, msf_match* __restrict__ out, int cap,
                                                        int32_t* __restrict__ n_out) {
  __shared__ uint32_t wsum[kDecThreads / 64];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t* mk = mask + (long long)pair * NTOK * MASK_WORDS;
  msf_match* o = out + (long long)pair * cap;
  constexpr int total = NTOK * MASK_WORDS;
  const int w_begin = tid * kDecPer;
  uint4 b[kDecPer / 4];
#pragma unroll
  for (int q = 0; q < kDecPer / 4; q++) {
    const int w = w_begin + 4 * q;
    b[q] = w < total ? *reinterpret_cast<const uint4*>(mk + w) : make_uint4(0u, 0u, 0u, 0u);
  }
  uint32_t c = 0;
#pragma unroll
  for (int q = 0; q < kDecPer / 4; q++) c += __popc(b[q].x) + __popc(b[q].y) + __popc(b[q].z) + __popc(b[q].w);
  uint32_t s = c;                                  // inclusive scan inside the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t v = __shfl_up(s, d);
    if (lane >= d) s += v;
  }
  if (lane == 63) wsum[wave] = s;
  __syncthreads();
  uint32_t base = 0, all = 0;
#pragma unroll
  for (int k = 0; k < kDecThreads / 64; k++) {
    const uint32_t v = wsum[k];
    all += v;
    if (k < wave) base += v;
  }
  if (tid == 0) n_out[pair] = (int32_t)all;
  if (c == 0) return;
  uint32_t pos = base + s - c;
#pragma unroll
  for (int q = 0; q < kDecPer / 4; q++) {
    const uint32_t wb[4] = {b[q].x, b[q].y, b[q].z, b[q].w};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      uint32_t bb = wb[r];
      if (!bb) continue;
      const int ww = w_begin + 4 * q + r;
      const int i = ww / MASK_WORDS, jw = (ww % MASK_WORDS) * 32;
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
  }
}

// ================================================================== host side
struct ConvDesc {
  int cin, cout, ks, stride, hin, win, hout, wout;
  float* d_w = nullptr;   // [KSTEPS*4][NPAD]
  float* d_w2 = nullptr;  // row-packed variant (RP = 2) for the 8 -> 8 layers
  uint16_t* d_wx = nullptr;  // split-bf16 fragments: 8 -> 8 layers (k_block8x) [hi | lo][kx][lane][8]; 32 -> 32 stride-1
                             // layers (k_convx) [tap][cout tile][hi | lo][lane][8]
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
  float *tok[4] = {nullptr, nullptr, nullptr, nullptr};  // f0 f1 t0 t1, each [max_pairs][1200][32]
  float* tok_cache = nullptr;  // [n_slots][1200][32] backbone tokens per frame slot (extract / match_slots)
  int n_slots = 0;             // 2 * max_pairs caller-visible slots + the handle's transparent frame cache
  float* fsc = nullptr;      // [2][max_pairs][1200][32] features / sqrt(32)
  __bf16* fsp = nullptr;     // [2][max_pairs][3][1200][32]: the same as three bf16 planes (split path)
  float* kv = nullptr;       // [max_pairs][1056]
  float* rstats = nullptr;   // [max_pairs][2][1200]
  float* cstats = nullptr;
  uint32_t* mask = nullptr;  // [max_pairs][1200][38]
  float* lim = nullptr;      // [max_pairs][1200] candidate bound per row of S
  SimCand* cand = nullptr;   // [max_pairs][kCandCap]
  uint32_t* cand_cnt = nullptr;
  float* gbound = nullptr;   // [max_pairs] single-pass statistics: the pair's exponent offset
  float* cpart = nullptr;    // [max_pairs][25][1200] column-sum partials
  float* rpart = nullptr;    // [max_pairs][3][1200] row-sum partials (one per third of the columns)
  uint32_t* sim_redo = nullptr;   // [max_pairs] pairs whose single-pass sums left the f32 range
  int* sim_tmax = nullptr;        // [max_pairs][75 items][25 column tiles] largest d per tile (k_sim_single -> k_sim_cand3); null: MSF_LOFTR_SIM_SKIP=0
  bool sim_skip = true;      // MSF_LOFTR_SIM_SKIP=0: the candidate pass evaluates every tile (tests compare: identical lists)
  bool out_fused = true;     // MSF_LOFTR_OUT_FUSED=0: the 1 x 1 output convolution and the token kernel as two passes (tests compare: identical)
  bool attn_pair = true;     // MSF_LOFTR_ATTN_PAIR=0: one launch per encoder block (tests compare: identical results)
  bool sim_single = true;    // MSF_LOFTR_SIM_SINGLE=0: the two running-maximum passes always
  bool sim_force_redo = false;    // MSF_LOFTR_SIM_FORCE_REDO=1 (tests): every pair takes the fallback
  bool dense_head = false;
  bool fuse_blocks = true;   // MSF_LOFTR_UNFUSED=1: one kernel per convolution (tests: bit-identical results)
  bool split_bf16 = true;    // MSF_LOFTR_F32=1: every convolution on the f32 MFMA (no split-bf16 kernels)
  bool down_stream = true;   // MSF_LOFTR_DOWN=0: layer2's first block as two kernels instead of the streaming k_down16x
  int strip_min_images = 64; // backbone passes of fewer images use the banded kernels (run_backbone)
  int strip_mode = 3;        // MSF_LOFTR_STRIP: non-zero (default) = the streaming strip kernels (stem + block 1 in one pass,
                             // then block 2, ...); 0 = the banded kernels calls of fewer than 64 images take anyway
  bool keep_debug = false;   // MSF_FLAG_KEEP_DEBUG: pair 0's confidence matrix + features for the parity tests
  float* conf_dbg = nullptr; // [1200][1200]
  float* feat_dbg = nullptr; // [2][1200][32]
  float* act_dbg[4] = {nullptr, nullptr, nullptr, nullptr};   // frame A of pair 0 after layer1..4: [8][240][320], [16][120][160], [32][60][80], [32][30][40]
  bool act_split[4] = {false, false, false, false};           // ... kept in the streaming kernels' split-pixel format (debug_get converts)
  int dbg_pair = 0;
  bool have_dbg = false;
  // Stage events (start, backbone done, transformer done, head done) as a ring of sets: a call records into the next free
  // set and nobody waits, so a caller can enqueue batches ahead of the device; stage_times() harvests the finished sets
  // and returns the sums since the last query (a query after every call sees that call's times, as before)
  static constexpr int kEvRing = 32;
  std::vector<hipEvent_t> ev;  // kEvRing x 4
  bool ev_set_rec[kEvRing] = {};
  int ev_cur = 0;
  float ev_acc[3] = {0.f, 0.f, 0.f};
  int ev_acc_calls = 0;
  bool ev_ok = false;
  hipEvent_t* ev_cur_set() { return ev.data() + 4 * ev_cur; }
  void ev_harvest(int i) {
    if (!ev_set_rec[i]) return;
    ev_set_rec[i] = false;
    hipEvent_t* e = ev.data() + 4 * i;
    if (hipEventSynchronize(e[3]) != hipSuccess) return;
    for (int k = 0; k < 3; k++) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, e[k], e[k + 1]) == hipSuccess) ev_acc[k] += t;
    }
    ev_acc_calls++;
  }
  hipEvent_t* ev_begin_call() {           // the set this call records into
    if (ev_set_rec[ev_cur]) {
      ev_cur = (ev_cur + 1) % kEvRing;
      ev_harvest(ev_cur);                  // the oldest set, kEvRing calls back: long finished
    }
    return ev_cur_set();
  }
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

std::string LoftrPipeline::init(const char* weights_path, int max_pairs, bool profile, bool keep_debug,
                                int extra_slots, bool f32_convs) {
  destroy();
  p_ = new Impl();
  Impl& P = *p_;
  P.max_pairs = max_pairs;
  P.n_slots = 2 * max_pairs + (extra_slots > 0 ? extra_slots : 0);
  {
    if (const char* d = getenv("MSF_LOFTR_DENSE_HEAD")) P.dense_head = atoi(d) != 0;   // tests: force the dense head
    if (const char* d = getenv("MSF_LOFTR_SIM_SINGLE")) P.sim_single = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_ATTN_PAIR")) P.attn_pair = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_OUT_FUSED")) P.out_fused = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_SIM_SKIP")) P.sim_skip = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_SIM_FORCE_REDO")) P.sim_force_redo = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_UNFUSED")) P.fuse_blocks = atoi(d) == 0;
    P.split_bf16 = !f32_convs;
    if (const char* d = getenv("MSF_LOFTR_F32")) P.split_bf16 = atoi(d) == 0;
    if (const char* d = getenv("MSF_LOFTR_STRIP")) P.strip_mode = atoi(d);
    if (const char* d = getenv("MSF_LOFTR_DOWN")) P.down_stream = atoi(d) != 0;
    if (const char* d = getenv("MSF_LOFTR_STRIP_MIN")) P.strip_min_images = atoi(d);   // tests: strips for a single pair
    // pairs per backbone pass (activation working set: 19.7 MB per pair).  Whole launches of 512 images fill the 512
    // workgroup slots of the fused block kernels in whole rounds (64 pairs: conv stack 8.41 ms, 128: 8.15, 256: 8.07)
    const char* e = getenv("MSF_LOFTR_CHUNK");
    const int want = e ? atoi(e) : 256;
    P.chunk = max_pairs < want ? max_pairs : (want > 0 ? want : 256);
  }
  P.profile = profile;
  P.keep_debug = keep_debug;
  // the model file the caller names (the reference's .onnx, dnnfeaturematcher.cpp:11-21) or the packed blob
  WeightMap blob;
  std::string err = load_weights(weights_path && weights_path[0] ? weights_path : default_weights(), &blob);
  if (!err.empty()) return err;
  auto need = [&](const std::string& n, size_t count) -> const std::vector<float>* {
    auto it = blob.find(n);
    if (it == blob.end() || it->second.data.size() != count) return nullptr;
    return &it->second.data;
  };
  auto upload = [&](const std::vector<float>& h, float** d) -> hipError_t {
    hipError_t e = hipMalloc(d, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    P.allocs.push_back(*d);
    return hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  };
  // backbone, index = execution order of the graph (and of the weight blob)
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
    const int grp = ksteps >= 4 ? 4 : ksteps, ksteps_pad = ((ksteps + grp - 1) / grp) * grp;   // = ConvCfg::NG * G
    std::vector<float> wb((size_t)ksteps_pad * 4 * npad, 0.f);
    for (int co = 0; co < c.cout; co++)
      for (int ci = 0; ci < c.cin; ci++)
        for (int ky = 0; ky < c.ks; ky++)
          for (int kx = 0; kx < c.ks; kx++) {
            const int k = (ky * c.ks + kx) * c.cin + ci;
            wb[(size_t)k * npad + co] = (*w)[(((size_t)co * c.cin + ci) * c.ks + ky) * c.ks + kx];
          }
    LF_TRY(upload(wb, &c.d_w));
    if (c.cout == 8 && (c.stride == 1 || c.cin == 1)) {
      // row-packed weights (RP = 2): k = ((kyy * ks + kx) * cin + ci) over ks + stride input rows,
      // column = row_sel * 8 + co; output row rs sees input rows stride * rs .. stride * rs + ks - 1
      const int kyy = c.ks + c.stride, kt2 = kyy * c.ks * c.cin, ks2 = (kt2 + 3) / 4;
      const int g2 = ks2 >= 4 ? 4 : ks2, ks2_pad = ((ks2 + g2 - 1) / g2) * g2;   // = ConvCfg::NG * G
      std::vector<float> w2((size_t)ks2_pad * 4 * 16, 0.f);
      for (int rs = 0; rs < 2; rs++)
        for (int co = 0; co < 8; co++)
          for (int ci = 0; ci < c.cin; ci++)
            for (int ky = 0; ky < c.ks; ky++)
              for (int kx = 0; kx < c.ks; kx++) {
                const int k = ((ky + c.stride * rs) * c.ks + kx) * c.cin + ci;
                w2[(size_t)k * 16 + rs * 8 + co] = (*w)[(((size_t)co * c.cin + ci) * c.ks + ky) * c.ks + kx];
              }
      LF_TRY(upload(w2, &c.d_w2));
    }
    auto to_bf16 = [](float f) -> uint16_t {
      uint32_t u;
      memcpy(&u, &f, 4);
      u += 0x7FFFu + ((u >> 16) & 1u);      // round to nearest even (weights are finite)
      return (uint16_t)(u >> 16);
    };
    auto from_bf16 = [](uint16_t h) -> float {
      const uint32_t u = (uint32_t)h << 16;
      float f;
      memcpy(&f, &u, 4);
      return f;
    };
    auto upload16 = [&](const std::vector<uint16_t>& h, uint16_t** d) -> hipError_t {
      hipError_t e = hipMalloc(reinterpret_cast<void**>(d), h.size() * sizeof(uint16_t));
      if (e != hipSuccess) return e;
      P.allocs.push_back(reinterpret_cast<float*>(*d));
      return hipMemcpy(*d, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    };
    if (c.cin == 1 && c.ks == 7) {
      // k_stem_strip8x stage 0: fragment (hi | lo, row group g): element j = kx of lane (idx = co + 8 rs, kq) is
      // w[co][ky = 4 g + kq - 2 rs][kx] / 255 (0 outside the 7 x 7 window); an output row pair spans image rows s = 0 .. 8
      const float k255 = (float)(1.0 / 255.0);
      std::vector<uint16_t> wx(stem8::WFRAG, 0);
      for (int g = 0; g < 3; g++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int co = l & 7, rs = (l >> 3) & 1, ky = 4 * g + (l >> 4) - 2 * rs;
            const float v = (ky >= 0 && ky <= 6 && j <= 6) ? (*w)[((size_t)co * 7 + ky) * 7 + j] * k255 : 0.f;
            const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
            wx[((size_t)(0 * 3 + g) * 64 + l) * 8 + j] = hi;
            wx[((size_t)(1 * 3 + g) * 64 + l) * 8 + j] = lo;
          }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 8 && c.cin == 8 && c.stride == 1) {
      // k_block8x: element j of lane (idx = co + 8 rs, input row s) of fragment kx is w[co][ci = j][ky = s - rs][kx]
      // (0 where output row rs does not see input row s), as hi = bf16(w) and lo = bf16(w - hi)
      std::vector<uint16_t> wx(blk8x::WFRAG, 0);
      for (int g = 0; g < 3; g++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int co = l & 7, rs = (l >> 3) & 1, s = l >> 4, ky = s - rs;
            const float v = (ky >= 0 && ky <= 2) ? (*w)[(((size_t)co * 8 + j) * 3 + ky) * 3 + g] : 0.f;
            const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
            wx[((size_t)(0 * 3 + g) * 64 + l) * 8 + j] = hi;
            wx[((size_t)(1 * 3 + g) * 64 + l) * 8 + j] = lo;
          }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 16 && c.cin == 8 && c.stride == 2 && c.ks == 3) {
      // k_down16x stage 1: fragment (hi | lo, ky): element j = ci of lane (cout l & 15, kx = l >> 4) is w[cout][ci][ky][kx]
      // (kx = 3: zero)
      std::vector<uint16_t> wx(down16::W1FRAG, 0);
      for (int g = 0; g < 3; g++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int co = l & 15, kx = l >> 4;
            const float v = kx < 3 ? (*w)[(((size_t)co * 8 + j) * 3 + g) * 3 + kx] : 0.f;
            const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
            wx[((size_t)(0 * 3 + g) * 64 + l) * 8 + j] = hi;
            wx[((size_t)(1 * 3 + g) * 64 + l) * 8 + j] = lo;
          }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 16 && c.cin == 8 && c.stride == 2 && c.ks == 1) {
      // k_down16x shortcut: rides on the ky = 1 fragments, whose kx = 1 block is the pixel (2Y, 2X): other blocks zero
      std::vector<uint16_t> wx(down16::WSFRAG, 0);
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) {
          const int co = l & 15, kx = l >> 4;
          const float v = kx == 1 ? (*w)[(size_t)co * 8 + j] : 0.f;
          const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
          wx[((size_t)0 * 64 + l) * 8 + j] = hi;
          wx[((size_t)1 * 64 + l) * 8 + j] = lo;
        }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 16 && c.cin == 16 && c.stride == 1) {
      // k_block16x: fragment (hi | lo, g): element j of lane (cout l & 15, kq = l >> 4) is
      // w[cout][ci = 8 (kq & 1) + j][tap 2 g + (kq >> 1)] (0 for the tenth tap)
      std::vector<uint16_t> wx(blk16x::WFRAG, 0);
      for (int g = 0; g < blk16x::G; g++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int co = l & 15, q = l >> 4, t = 2 * g + (q >> 1), ci = 8 * (q & 1) + j;
            const float v = t < 9 ? (*w)[(((size_t)co * 16 + ci) * 3 + t / 3) * 3 + t % 3] : 0.f;
            const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
            wx[((size_t)(0 * blk16x::G + g) * 64 + l) * 8 + j] = hi;
            wx[((size_t)(1 * blk16x::G + g) * 64 + l) * 8 + j] = lo;
          }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 32 && c.stride == 2 && c.ks == 3) {
      // k_convx2<CIN>: fragment (g, cout tile n, hi | lo): lane (cout 16 n + (l & 15), kq = l >> 4), element j:
      //   CIN = 32: w[cout][ci = 8 kq + j][tap g];  CIN = 16: w[cout][ci = 8 (kq & 1) + j][tap 2 g + (kq >> 1)] (tenth tap: 0)
      const int G = c.cin == 32 ? 9 : 5;
      std::vector<uint16_t> wx((size_t)G * 2 * 2 * 64 * 8, 0);
      for (int g = 0; g < G; g++)
        for (int n = 0; n < 2; n++)
          for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) {
              const int co = 16 * n + (l & 15), q = l >> 4;
              const int t = c.cin == 32 ? g : 2 * g + (q >> 1), ci = c.cin == 32 ? 8 * q + j : 8 * (q & 1) + j;
              const float v = t < 9 ? (*w)[(((size_t)co * c.cin + ci) * 3 + t / 3) * 3 + t % 3] : 0.f;
              const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
              wx[((size_t)((g * 2 + n) * 2 + 0) * 64 + l) * 8 + j] = hi;
              wx[((size_t)((g * 2 + n) * 2 + 1) * 64 + l) * 8 + j] = lo;
            }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 32 && c.stride == 2 && c.ks == 1) {
      // k_convx2 shortcut: rides on the centre tap's fragment: CIN = 32: every K block (channel block kq);
      // CIN = 16: the blocks of tap 4 = kq 0, 1 of group 2 (kq 2, 3 hold tap 5: zero)
      std::vector<uint16_t> wx((size_t)2 * 2 * 64 * 8, 0);
      for (int n = 0; n < 2; n++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 8; j++) {
            const int co = 16 * n + (l & 15), q = l >> 4;
            const bool live = c.cin == 32 || q < 2;
            const int ci = c.cin == 32 ? 8 * q + j : 8 * (q & 1) + j;
            const float v = live ? (*w)[(size_t)co * c.cin + ci] : 0.f;
            const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
            wx[((size_t)(n * 2 + 0) * 64 + l) * 8 + j] = hi;
            wx[((size_t)(n * 2 + 1) * 64 + l) * 8 + j] = lo;
          }
      LF_TRY(upload16(wx, &c.d_wx));
    }
    if (c.cout == 32 && c.cin == 32 && c.stride == 1 && c.ks == 3) {
      // k_convx<32>: fragment (tap g, cout tile n, hi | lo): element j of lane (cout 16 n + (l & 15), channel block l >> 4)
      // is w[cout][ci = 8 (l >> 4) + j][ky = g / 3][kx = g % 3]
      using F = cvx::Cfg<32>;
      std::vector<uint16_t> wx((size_t)F::WSLOTS * 8, 0);
      for (int g = 0; g < F::G; g++)
        for (int n = 0; n < F::NT; n++)
          for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) {
              const int co = 16 * n + (l & 15), ci = 8 * (l >> 4) + j;
              const float v = (*w)[(((size_t)co * 32 + ci) * 3 + g / 3) * 3 + g % 3];
              const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
              wx[((size_t)((g * F::NT + n) * 2 + 0) * 64 + l) * 8 + j] = hi;
              wx[((size_t)((g * F::NT + n) * 2 + 1) * 64 + l) * 8 + j] = lo;
            }
      LF_TRY(upload16(wx, &c.d_wx));
    }
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
    // slot orders (see the kernel comments): 0 = P8 (feature 8*kq + s), 1 = PD (feature 16*(s/4) + 4*kq + s%4);
    // `split`: wmlp0's 64 inputs are [x (P8, 8 slots) | merged message (PD, 8 slots)]
    struct { const char* n; int in, out; int order; const float** dst; } items[] = {
        {"wq", 32, 32, 0, &P.blk[b].wq_p}, {"wk", 32, 32, 0, &P.blk[b].wk_p}, {"wv", 32, 32, 0, &P.blk[b].wv_p},
        {"wmerge", 32, 32, 1, &P.blk[b].wm_p}, {"wmlp0", 64, 64, 2, &P.blk[b].w0_p}, {"wmlp1", 64, 32, 1, &P.blk[b].w1_p}};
    for (auto& it : items) {
      snprintf(nm, sizeof nm, "blk%d.%s", b, it.n);
      const auto* w = need(nm, (size_t)it.in * it.out);
      if (!w) return std::string("io: weights blob lacks ") + nm;
      const int slots = it.in / 4, mtiles = it.out / 16;
      std::vector<float> pk((size_t)it.in * it.out);
      for (int mt = 0; mt < mtiles; mt++)
        for (int sl = 0; sl < slots; sl++)
          for (int ln = 0; ln < 64; ln++) {
            const int kq = ln >> 4;
            int feat;
            if (it.order == 0) feat = 8 * kq + sl;
            else if (it.order == 1) feat = 16 * (sl >> 2) + 4 * kq + (sl & 3);
            else feat = sl < 8 ? 8 * kq + sl : 32 + 16 * ((sl - 8) >> 2) + 4 * kq + ((sl - 8) & 3);
            pk[((size_t)mt * slots + sl) * 64 + ln] = (*w)[(size_t)feat * it.out + 16 * mt + (ln & 15)];
          }
      float* d = nullptr;
      LF_TRY(upload(pk, &d));
      *it.dst = d;
      // split-bf16 fragments of the matrices k_attn_update_x multiplies by: element j of lane ln of (mtile, K group kg)
      // is slot 8 kg + j of the f32 packing above
      const uint16_t** xdst = !strcmp(it.n, "wq") ? &P.blk[b].wq_x : !strcmp(it.n, "wmerge") ? &P.blk[b].wm_x
                              : !strcmp(it.n, "wmlp0") ? &P.blk[b].w0_x : !strcmp(it.n, "wmlp1") ? &P.blk[b].w1_x : nullptr;
      if (xdst) {
        auto to_bf16 = [](float f) -> uint16_t {
          uint32_t u;
          memcpy(&u, &f, 4);
          u += 0x7FFFu + ((u >> 16) & 1u);
          return (uint16_t)(u >> 16);
        };
        auto from_bf16 = [](uint16_t h) -> float {
          const uint32_t u = (uint32_t)h << 16;
          float f;
          memcpy(&f, &u, 4);
          return f;
        };
        const int kgs = slots / 8;
        std::vector<uint16_t> fx((size_t)mtiles * kgs * 2 * 64 * 8);
        for (int mt = 0; mt < mtiles; mt++)
          for (int kg = 0; kg < kgs; kg++)
            for (int ln = 0; ln < 64; ln++)
              for (int j = 0; j < 8; j++) {
                const float v = pk[((size_t)mt * slots + kg * 8 + j) * 64 + ln];
                const uint16_t hi = to_bf16(v), lo = to_bf16(v - from_bf16(hi));
                fx[((((size_t)mt * kgs + kg) * 2 + 0) * 64 + ln) * 8 + j] = hi;
                fx[((((size_t)mt * kgs + kg) * 2 + 1) * 64 + ln) * 8 + j] = lo;
              }
        uint16_t* dx = nullptr;
        LF_TRY(hipMalloc(reinterpret_cast<void**>(&dx), fx.size() * sizeof(uint16_t)));
        P.allocs.push_back(reinterpret_cast<float*>(dx));
        LF_TRY(hipMemcpy(dx, fx.data(), fx.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        *xdst = dx;
      }
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
  for (int i = 0; i < 4; i++) LF_TRY(dalloc(&P.tok[i], (size_t)max_pairs * NTOK * DM));
  LF_TRY(dalloc(&P.tok_cache, (size_t)P.n_slots * NTOK * DM));
  LF_TRY(dalloc(&P.fsc, (size_t)2 * max_pairs * NTOK * DM));
  {
    float* m = nullptr;
    LF_TRY(dalloc(&m, (size_t)3 * max_pairs * NTOK * DM));   // 2 x 3 bf16 planes = 3 floats per feature
    P.fsp = reinterpret_cast<__bf16*>(m);
  }
  LF_TRY(dalloc(&P.kv, (size_t)2 * max_pairs * (DM * DM + DM)));     // two halves: a launch may carry two encoder blocks
  LF_TRY(dalloc(&P.rstats, (size_t)max_pairs * 2 * NTOK));
  LF_TRY(dalloc(&P.cstats, (size_t)max_pairs * 2 * NTOK));
  {
    float* m = nullptr;
    LF_TRY(dalloc(&m, (size_t)max_pairs * NTOK * MASK_WORDS));
    P.mask = reinterpret_cast<uint32_t*>(m);
    // the last 16-bit chunk of every row (bits 1200 .. 1215) is never written: keep it zero
    LF_TRY(hipMemset(P.mask, 0, (size_t)max_pairs * NTOK * MASK_WORDS * sizeof(uint32_t)));
  }
  LF_TRY(dalloc(&P.lim, (size_t)max_pairs * NTOK));
  LF_TRY(dalloc(&P.gbound, (size_t)max_pairs));
  LF_TRY(dalloc(&P.cpart, (size_t)max_pairs * kSimParts * NTOK));
  LF_TRY(dalloc(&P.rpart, (size_t)max_pairs * kSimColParts * NTOK));
  {
    float* m = nullptr;
    LF_TRY(dalloc(&m, (size_t)max_pairs));
    P.sim_redo = reinterpret_cast<uint32_t*>(m);
    if (P.sim_skip) {
      LF_TRY(dalloc(&m, (size_t)max_pairs * kSimItemsPerPair * kSimColTiles));
      P.sim_tmax = reinterpret_cast<int*>(m);
    }
  }
  {
    float* m = nullptr;
    LF_TRY(dalloc(&m, (size_t)max_pairs * kCandCap * 2));
    P.cand = reinterpret_cast<SimCand*>(m);
    LF_TRY(dalloc(&m, (size_t)max_pairs));
    P.cand_cnt = reinterpret_cast<uint32_t*>(m);
  }
  LF_TRY(dalloc(&P.conf_dbg, (size_t)NTOK * NTOK));
  LF_TRY(dalloc(&P.feat_dbg, (size_t)2 * NTOK * DM));
  if (keep_debug) {
    const size_t act_elems[4] = {8u * 240 * 320, 16u * 120 * 160, 32u * 60 * 80, 32u * 30 * 40};
    for (int l = 0; l < 4; l++) LF_TRY(dalloc(&P.act_dbg[l], act_elems[l]));
  }
  if (profile) {
    P.ev.resize(4 * Impl::kEvRing);
    for (auto& e : P.ev) LF_TRY(hipEventCreate(&e));
    P.ev_ok = true;
  }
  return "";
}

namespace {

template <int CIN, int COUT, int KS, int S, int OTW, bool RELU, bool RES, bool U8IN, int RP = 1, bool SC = false>
void launch_conv(const ConvDesc& c, const void* in, long long in_img_stride, int in_row_stride, const float* res,
                 float* out, int n_img, hipStream_t st, const ConvDesc* sc = nullptr, float* out_sc = nullptr) {
  using C = ConvCfg<CIN, COUT, KS, S, OTW, RELU, RES, U8IN, RP>;
  static_assert((C::PLANE % 32) == 16, "plane stride must be 16 mod 32");
  static_assert(C::PLANE >= C::RAW, "plane too small");
  const size_t lds = (size_t)CIN * C::PLANE * sizeof(float);
  auto kern = k_conv<CIN, COUT, KS, S, OTW, RELU, RES, U8IN, RP, SC>;
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    if (lds > 48 * 1024)
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int n_bands = (c.hout + C::OTH - 1) / C::OTH;     // a workgroup walks the x tiles of its row band
  hipLaunchKernelGGL(kern, dim3(n_bands * n_img), dim3(64 * kConvWaves), lds, st, in, in_img_stride, in_row_stride,
                     RP == 2 ? c.d_w2 : c.d_w, c.d_b, res, out, c.hin, c.win, c.hout, c.wout, sc ? sc->d_w : nullptr,
                     sc ? sc->d_b : nullptr, out_sc, n_bands);
}

// y = relu(conv_b(relu(conv_a(x))) + x) for an 8-channel, stride-1 BasicBlock at 240 x 320 (k_block8)
void launch_block8(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  const size_t lds = (size_t)blk8::LDS_FLOATS * sizeof(float);
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_block8), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int n_bands = ca.hout / blk8::R;      // 240 / 8
  hipLaunchKernelGGL(k_block8, dim3(n_bands * n_img), dim3(256), lds, st, in, ca.d_w2, ca.d_b, cb.d_w2, cb.d_b, out, ca.hout,
                     ca.wout, n_bands);
}

// the same block on split-bf16 MFMAs (k_block8x)
void launch_block8x(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_block8x), hipFuncAttributeMaxDynamicSharedMemorySize, blk8x::LDS_BYTES);
  });
  const int n_bands = ca.hout / blk8x::R;
  hipLaunchKernelGGL(k_block8x, dim3(n_bands * n_img), dim3(256), blk8x::LDS_BYTES, st, in, ca.d_wx, ca.d_b, cb.d_wx, cb.d_b,
                     out, ca.hout, ca.wout, n_bands);
}

// one 8-channel BasicBlock as a streaming pass (k_strip8x<1>): convolutions cv[0], cv[1]
void launch_strip8x(const ConvDesc* cv, const float* in, float* out, int n_img, hipStream_t st) {
  constexpr int NB = 1, WVW = strip8::WAVES;
#ifndef MSF_STRIP8_S
#define MSF_STRIP8_S strip8::S
#endif
  auto kern = k_strip8x<NB, WVW, MSF_STRIP8_S>;
  constexpr int lds = strip8::lds_bytes<NB>();
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  StripW sw{};
  for (int c = 0; c < 2 * NB; c++) { sw.wx[c] = cv[c].d_wx; sw.b[c] = cv[c].d_b; }
  const int Wd = cv[0].wout, Sd = MSF_STRIP8_S;
  const int n_strips = (Wd - Sd + Sd - 1) / Sd + 1;
  hipLaunchKernelGGL(kern, dim3(n_strips * n_img), dim3(64 * WVW), lds, st, in, sw, out, cv[0].hout, Wd, n_strips);
}

// stem + first 8-channel BasicBlock as one streaming pass over u8 frames (k_stem_strip8x): convolutions cv[0 .. 3)
void launch_stem_strip8x(const ConvDesc* cv, const uint8_t* framesA, int nA, const uint8_t* framesB, int nB,
                         long long frame_stride, int row_stride, float* out, hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_strip8x), hipFuncAttributeMaxDynamicSharedMemorySize, stem8::LDS_BYTES);
  });
  StripW sw{};
  for (int c = 0; c < 2; c++) { sw.wx[c] = cv[1 + c].d_wx; sw.b[c] = cv[1 + c].d_b; }
  const int n_strips = cv[1].wout / strip8::S;
  hipLaunchKernelGGL(k_stem_strip8x, dim3(n_strips * (nA + nB)), dim3(64 * strip8::WAVES), stem8::LDS_BYTES, st, framesA, nA,
                     framesB, frame_stride, row_stride, cv[0].d_wx, cv[0].d_b, sw, out, cv[1].hout, cv[1].wout, n_strips);
}

// down-sampling block 8 -> 16 as one streaming pass (k_down16x): cs2 = 3x3 stride 2, csc = 1x1 stride 2, c2 = 3x3
void launch_down16x(const ConvDesc& cs2, const ConvDesc& csc, const ConvDesc& c2, const float* in, float* out, int n_img,
                    hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_down16x), hipFuncAttributeMaxDynamicSharedMemorySize, down16::LDS_BYTES);
  });
  DownW dw{cs2.d_wx, csc.d_wx, c2.d_wx, cs2.d_b, csc.d_b, c2.d_b};
  const int n_strips = c2.wout / down16::S;
  hipLaunchKernelGGL(k_down16x, dim3(n_strips * n_img), dim3(64 * down16::WAVES), down16::LDS_BYTES, st, in, dw, out, c2.hout,
                     c2.wout, n_strips);
}

// the 16-channel BasicBlock on split-bf16 MFMAs (k_block16x)
void launch_block16x(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_block16x), hipFuncAttributeMaxDynamicSharedMemorySize, blk16x::LDS_BYTES);
  });
  const int n_bands = ca.hout / blk16x::R;
  hipLaunchKernelGGL(k_block16x, dim3(n_bands * n_img), dim3(256), blk16x::LDS_BYTES, st, in, ca.d_wx, ca.d_b, cb.d_wx, cb.d_b,
                     out, ca.hout, ca.wout, n_bands);
}

// the 16-channel BasicBlock as one streaming pass (k_strip16x)
void launch_strip16x(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_strip16x), hipFuncAttributeMaxDynamicSharedMemorySize, strip16::LDS_BYTES);
  });
  const int n_strips = ca.wout / strip16::S;
  hipLaunchKernelGGL(k_strip16x, dim3(n_strips * n_img), dim3(64 * strip16::WAVES), strip16::LDS_BYTES, st, in, ca.d_wx, ca.d_b,
                     cb.d_wx, cb.d_b, out, ca.hout, ca.wout, n_strips);
}

// down-sampling block 16 -> 32 as one streaming pass (k_down32x): cs2 = 3x3 stride 2, csc = 1x1 stride 2, c2 = 3x3
void launch_down32x(const ConvDesc& cs2, const ConvDesc& csc, const ConvDesc& c2, const float* in, float* out, int n_img,
                    hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_down32x), hipFuncAttributeMaxDynamicSharedMemorySize, down32::LDS_BYTES);
  });
  DownW dw{cs2.d_wx, csc.d_wx, c2.d_wx, cs2.d_b, csc.d_b, c2.d_b};
  const int n_strips = (c2.wout + down32::S - 1) / down32::S;
  hipLaunchKernelGGL(k_down32x, dim3(n_strips * n_img), dim3(64 * down32::WAVES), down32::LDS_BYTES, st, in, dw, out, c2.hout,
                     c2.wout, n_strips);
}

// a 32-channel BasicBlock as one streaming pass (k_strip32x)
void launch_strip32x(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_strip32x), hipFuncAttributeMaxDynamicSharedMemorySize, strip32::LDS_BYTES);
  });
  const int n_strips = (ca.wout + strip32::S - 1) / strip32::S;
  hipLaunchKernelGGL(k_strip32x, dim3(n_strips * n_img), dim3(64 * strip32::WAVES), strip32::LDS_BYTES, st, in, ca.d_wx, ca.d_b,
                     cb.d_wx, cb.d_b, out, ca.hout, ca.wout, n_strips);
}

// 3x3 stride-1 C -> C convolution (+ residual) + ReLU on split-bf16 MFMAs (k_convx)
template <int C, bool RES>
void launch_convx(const ConvDesc& c, const float* in, const float* res, float* out, int n_img, hipStream_t st) {
  using F = cvx::Cfg<C>;
  auto kern = k_convx<C, RES>;
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
  });
  const int n_bands = (c.hout + cvx::OTH - 1) / cvx::OTH;
  hipLaunchKernelGGL(kern, dim3(n_bands * n_img), dim3(256), F::LDS_BYTES, st, in, c.d_wx, c.d_b, res, out, c.hout, c.wout,
                     n_bands);
}

// 3x3 stride-2 CIN -> 32 convolution + ReLU with its 1x1 stride-2 shortcut on split-bf16 MFMAs (k_convx2)
template <int CIN>
void launch_convx2(const ConvDesc& c, const ConvDesc& sc, const float* in, float* out, float* out_sc, int n_img, hipStream_t st) {
  using F = cvx2::Cfg<CIN>;
  auto kern = k_convx2<CIN>;
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, F::LDS_BYTES);
  });
  const int n_bands = (c.hout + F::OTH - 1) / F::OTH;
  hipLaunchKernelGGL(kern, dim3(n_bands * n_img), dim3(256), F::LDS_BYTES, st, in, c.d_wx, c.d_b, sc.d_wx, sc.d_b, out, out_sc,
                     c.hin, c.win, c.hout, c.wout, n_bands);
}

// the same for a 16-channel, stride-1 BasicBlock at 120 x 160 (k_block16)
void launch_block16(const ConvDesc& ca, const ConvDesc& cb, const float* in, float* out, int n_img, hipStream_t st) {
  const size_t lds = (size_t)blk16::LDS_FLOATS * sizeof(float);
  static std::once_flag attr_once;     // handles on several host threads (msf_multi) may arrive here together
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_block16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  const int n_bands = ca.hout / blk16::R;     // 120 / 8
  hipLaunchKernelGGL(k_block16, dim3(n_bands * n_img), dim3(256), lds, st, in, ca.d_w, ca.d_b, cb.d_w, cb.d_b, out, ca.hout,
                     ca.wout, n_bands);
}

}  // namespace

namespace {

// Backbone (ConvertImageToFloat + 21 convs + positional encoding) for one chunk of images: nA frames from srcA
// followed by nB frames from srcB (either may be 0); their tokens go to tokA / tokB.  A frame's tokens do not depend
// on the frame it is paired with, which is what the per-frame token cache below rests on.
void run_backbone(LoftrPipeline::Impl& P, const uint8_t* srcA, int nA, float* tokA, const uint8_t* srcB, int nB,
                  float* tokB, long long frame_stride, int row_stride, hipStream_t st);

}  // namespace

hipError_t LoftrPipeline::match(int n_pairs, const uint8_t* d_a, const uint8_t* d_b, long long frame_stride,
                                int row_stride, float threshold, msf_match* d_out, int cap, int32_t* d_n_out,
                                hipStream_t st) {
  if (!p_) return hipErrorNotInitialized;
  Impl& P = *p_;
  if (n_pairs > P.max_pairs) return hipErrorInvalidValue;
  const long long ts = (long long)NTOK * DM;
  hipEvent_t* ev = P.ev_ok ? P.ev_begin_call() : nullptr;
  if (ev) hipEventRecord(ev[0], st);
  // ---- backbone, in chunks of pairs (activations are the big buffers); tokens of all pairs are kept
  for (int p0 = 0; p0 < n_pairs; p0 += P.chunk) {
    const int n = std::min(P.chunk, n_pairs - p0);
    run_backbone(P, d_a + (long long)p0 * frame_stride, n, P.tok[0] + p0 * ts, d_b + (long long)p0 * frame_stride, n,
                 P.tok[1] + p0 * ts, frame_stride, row_stride, st);
  }
  if (ev) hipEventRecord(ev[1], st);
  return transformer_and_head(n_pairs, threshold, d_out, cap, d_n_out, st);
}

// "next" row 1 of SURVEY.md 8f for LoFTR: the backbone runs once per frame, its tokens stay in a slot.
hipError_t LoftrPipeline::extract(int n_frames, const uint8_t* d_frames, long long frame_stride, int row_stride,
                                  int first_slot, hipStream_t st) {
  if (!p_) return hipErrorNotInitialized;
  Impl& P = *p_;
  if (n_frames < 0 || first_slot < 0 || first_slot + n_frames > P.n_slots) return hipErrorInvalidValue;
  const long long ts = (long long)NTOK * DM;
  for (int f0 = 0; f0 < n_frames; f0 += 2 * P.chunk) {
    const int n = std::min(2 * P.chunk, n_frames - f0);
    run_backbone(P, d_frames + (long long)f0 * frame_stride, n, P.tok_cache + (long long)(first_slot + f0) * ts, nullptr,
                 0, nullptr, frame_stride, row_stride, st);
  }
  return hipGetLastError();
}

hipError_t LoftrPipeline::match_slots(int n_pairs, const int32_t* d_slot_a, const int32_t* d_slot_b, float threshold,
                                      msf_match* d_out, int cap, int32_t* d_n_out, hipStream_t st, int slot_limit) {
  if (!p_) return hipErrorNotInitialized;
  Impl& P = *p_;
  if (n_pairs > P.max_pairs) return hipErrorInvalidValue;
  if (n_pairs <= 0) return hipSuccess;
  hipEvent_t* ev = P.ev_ok ? P.ev_begin_call() : nullptr;
  if (ev) hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(k_gather_tokens, dim3(NTOK * DM / 4 / 256 + 1, n_pairs, 2), dim3(256), 0, st, P.tok_cache,
                     d_slot_a, d_slot_b, P.n_slots, P.tok[0], P.tok[1]);
  if (ev) hipEventRecord(ev[1], st);
  const hipError_t e = transformer_and_head(n_pairs, threshold, d_out, cap, d_n_out, st);
  if (e != hipSuccess) return e;
  const int limit = slot_limit > 0 && slot_limit < P.n_slots ? slot_limit : P.n_slots;
  hipLaunchKernelGGL(k_mark_bad_slots, dim3((n_pairs + 255) / 256), dim3(256), 0, st, n_pairs, d_slot_a, d_slot_b, limit, d_n_out);
  return hipGetLastError();
}

int LoftrPipeline::max_slots() const { return p_ ? p_->n_slots : 0; }

namespace {

void run_backbone(LoftrPipeline::Impl& P, const uint8_t* srcA, int nA, float* tokA, const uint8_t* srcB, int nB,
                  float* tokB, long long frame_stride, int row_stride, hipStream_t st) {
  const int ni = nA + nB;
  const ConvDesc* c = P.conv;
  float *a = P.bufA, *b = P.bufB, *cc = P.bufC, *d = P.bufD;
  const long long s8 = 8LL * 240 * 320;
  // the stem reads u8 frames from up to two arrays: launch it per array
  // A streaming workgroup walks a whole 240-row strip (~70 us however few images there are): calls of fewer than 64
  // images -- the single-pair drop-in path -- keep the short banded workgroups (stateless pair: 0.84 vs 1.06 ms).
  const int strip_mode = ni >= P.strip_min_images ? P.strip_mode : 0;
  const bool stem_fused = P.fuse_blocks && P.split_bf16 && strip_mode != 0;   // stem + block 1 in one pass -> cc
  if (stem_fused) {
    launch_stem_strip8x(c, srcA, nA, srcB, nB, frame_stride, row_stride, cc, st);
  } else {
    if (nA) launch_conv<1, 8, 7, 2, 64, true, false, true, 2>(c[0], srcA, frame_stride, row_stride, nullptr, a, nA, st);
    if (nB) launch_conv<1, 8, 7, 2, 64, true, false, true, 2>(c[0], srcB, frame_stride, row_stride, nullptr, a + (long long)nA * s8, nB, st);
  }
  // layer1 @240x320, 8 ch
  if (P.fuse_blocks) {   // each BasicBlock in one kernel: the intermediate activation stays in LDS
    if (stem_fused) {
      launch_strip8x(c + 3, cc, a, ni, st);                                                          // a = 196
    } else if (P.split_bf16) {
      launch_block8x(c[1], c[2], a, cc, ni, st);
      launch_block8x(c[3], c[4], cc, a, ni, st);                                                     // a = 196
    } else {
      launch_block8(c[1], c[2], a, cc, ni, st);
      launch_block8(c[3], c[4], cc, a, ni, st);                                                      // a = 196
    }
  } else {
    launch_conv<8, 8, 3, 1, 64, true, false, false, 2>(c[1], a, s8, 0, nullptr, b, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, true, false, 2>(c[2], b, s8, 0, a, cc, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, false, false, 2>(c[3], cc, s8, 0, nullptr, b, ni, st);
    launch_conv<8, 8, 3, 1, 64, true, true, false, 2>(c[4], b, s8, 0, cc, a, ni, st);                 // a = 196
  }
  // MSF_FLAG_KEEP_DEBUG: the first image's activation (raw bytes; `split`: in the streaming kernels' split-pixel format)
  auto keep = [&](int l, const float* src, size_t elems, bool split) {
    if (P.act_dbg[l]) {
      hipMemcpyAsync(P.act_dbg[l], src, elems * sizeof(float), hipMemcpyDeviceToDevice, st);
      P.act_split[l] = split;
    }
  };
  keep(0, a, 8u * 240 * 320, P.fuse_blocks && stem_fused);
  // layer2 @120x160, 16 ch
  const long long s16 = 16LL * 120 * 160;
  if (P.fuse_blocks && P.split_bf16 && strip_mode != 0 && P.down_stream) {
    launch_down16x(c[5], c[7], c[6], a, cc, ni, st);                                               // cc = 205
  } else {
    launch_conv<8, 16, 3, 2, 32, true, false, false, 1, true>(c[5], a, s8, 0, nullptr, b, ni, st, &c[7], d);   // + shortcut -> d
    launch_conv<16, 16, 3, 1, 32, true, true, false>(c[6], b, s16, 0, d, cc, ni, st);              // cc = 205
  }
  if (P.fuse_blocks) {
    if (P.split_bf16 && strip_mode != 0 && P.down_stream) launch_strip16x(c[8], c[9], cc, a, ni, st);   // a = 212
    else if (P.split_bf16) launch_block16x(c[8], c[9], cc, a, ni, st);
    else launch_block16(c[8], c[9], cc, a, ni, st);
  } else {
    launch_conv<16, 16, 3, 1, 32, true, false, false>(c[8], cc, s16, 0, nullptr, b, ni, st);
    launch_conv<16, 16, 3, 1, 32, true, true, false>(c[9], b, s16, 0, cc, a, ni, st);              // a = 212
  }
  keep(1, a, 16u * 120 * 160, P.fuse_blocks && P.split_bf16 && strip_mode != 0 && P.down_stream);
  // layer3 @60x80, 32 ch
  const long long s32 = 32LL * 60 * 80;
  const bool down3 = P.split_bf16 && P.fuse_blocks && strip_mode != 0 && P.down_stream;
  if (down3) launch_down32x(c[10], c[12], c[11], a, cc, ni, st);                                   // cc = 221
  else if (P.split_bf16) launch_convx2<16>(c[10], c[12], a, b, d, ni, st);
  else launch_conv<16, 32, 3, 2, 16, true, false, false, 1, true>(c[10], a, s16, 0, nullptr, b, ni, st, &c[12], d);
  if (P.split_bf16) {
    if (!down3) launch_convx<32, true>(c[11], b, d, cc, ni, st);                                   // cc = 221
    if (strip_mode != 0 && P.down_stream) {
      launch_strip32x(c[13], c[14], cc, a, ni, st);                                                // a = 228
    } else {
      launch_convx<32, false>(c[13], cc, nullptr, b, ni, st);
      launch_convx<32, true>(c[14], b, cc, a, ni, st);                                             // a = 228
    }
  } else {
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[11], b, s32, 0, d, cc, ni, st);             // cc = 221
    launch_conv<32, 32, 3, 1, 16, true, false, false>(c[13], cc, s32, 0, nullptr, b, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[14], b, s32, 0, cc, a, ni, st);             // a = 228
  }
  keep(2, a, 32u * 60 * 80, false);
  // layer4 @30x40, 32 ch
  const long long s40 = 32LL * 30 * 40;
  if (P.split_bf16) launch_convx2<32>(c[15], c[17], a, b, d, ni, st);
  else launch_conv<32, 32, 3, 2, 16, true, false, false, 1, true>(c[15], a, s32, 0, nullptr, b, ni, st, &c[17], d);
  if (P.split_bf16) {
    launch_convx<32, true>(c[16], b, d, cc, ni, st);                                               // cc = 237
    // (k_strip32x at 30 x 40: 167 us against 165 for these two -- 2.5 strips of 16 columns, a third of the workgroups half empty)
    launch_convx<32, false>(c[18], cc, nullptr, b, ni, st);
    launch_convx<32, true>(c[19], b, cc, a, ni, st);                                               // a = 244
  } else {
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[16], b, s40, 0, d, cc, ni, st);             // cc = 237
    launch_conv<32, 32, 3, 1, 16, true, false, false>(c[18], cc, s40, 0, nullptr, b, ni, st);
    launch_conv<32, 32, 3, 1, 16, true, true, false>(c[19], b, s40, 0, cc, a, ni, st);             // a = 244
  }
  keep(3, a, 32u * 30 * 40, false);
  const int tok_tiles = (NTOK + kTokTile - 1) / kTokTile;
  if (P.out_fused) {
    hipLaunchKernelGGL(k_out_tokens, dim3(tok_tiles, ni), dim3(256), 0, st, a, c[20].d_w, c[20].d_b, P.d_pe, tokA, nA, tokB, ni);
    return;
  }
  launch_conv<32, 32, 1, 1, 16, false, false, false>(c[20], a, s40, 0, nullptr, b, ni, st);      // b = 245
  if (nA) hipLaunchKernelGGL(k_tokens, dim3(tok_tiles, nA), dim3(256), 0, st, b, P.d_pe, tokA, nA);
  if (nB) hipLaunchKernelGGL(k_tokens, dim3(tok_tiles, nB), dim3(256), 0, st, b + (long long)nA * 32LL * 30 * 40, P.d_pe, tokB, nB);
}

}  // namespace

hipError_t LoftrPipeline::transformer_and_head(int n_pairs, float threshold, msf_match* d_out, int cap,
                                               int32_t* d_n_out, hipStream_t st) {
  Impl& P = *p_;
  const long long ts = (long long)NTOK * DM;
  hipEvent_t* ev = P.ev_ok ? P.ev_cur_set() : nullptr;      // the set match() / match_slots() began
  // ---- 8 encoder blocks over all pairs: (x, source) -> dst   [self, self, cross, cross(updated feat0)] x 2
  const int n = n_pairs;
  float *f0 = P.tok[0], *f1 = P.tok[1], *t0 = P.tok[2], *t1 = P.tok[3];
  struct { const float* x; const float* s; float* o; } seq[8] = {
      {f0, f0, t0}, {f1, f1, t1}, {t0, t1, f0}, {t1, f0, f1}, {f0, f0, t0}, {f1, f1, t1}, {t0, t1, f0}, {t1, f0, f1}};
  const int upd_blocks = (NTOK / 16 + 4 * kUpdTilesPerWave - 1) / (4 * kUpdTilesPerWave);
  for (int bi = 0; bi < 8; bi++) {
    if (P.split_bf16) {
      // the two self-attention blocks of a layer pair (0 | 1, 4 | 5) read and write disjoint sequences: ONE launch of each
      // kernel for both (12 launches per call instead of 16; the second block's KV goes to the second half of P.kv)
      const bool pairup = (bi & 3) == 0 && P.attn_pair;
      const int bj = pairup ? bi + 1 : bi;
      float* kv2 = P.kv + (long long)P.max_pairs * (DM * DM + DM);
      hipLaunchKernelGGL(k_attn_kv_x, dim3(pairup ? 2 * n : n), dim3(64 * kKvWaves), 0, st, seq[bi].s, ts, P.blk[bi], P.kv, n,
                         seq[bj].s, P.blk[bj], kv2);
      const int n_items = n * upd_blocks;
      const int per_wg = 1;     // one item per workgroup (runs of 3 / 5 items with the weights staged once: 0.52 / 0.54 vs 0.49 ms)
      const int wgs = (n_items + per_wg - 1) / per_wg;
      // blocks 6 and 7 write the final f0 and f1: they also write the head's scaled features and bf16 planes
      float* fs_o = bi == 6 ? P.fsc : bi == 7 ? P.fsc + (long long)P.max_pairs * ts : nullptr;
      __bf16* pl_o = bi == 6 ? P.fsp : bi == 7 ? P.fsp + (long long)P.max_pairs * 3 * ts : nullptr;
      hipLaunchKernelGGL(k_attn_update_x, dim3(pairup ? 2 * wgs : wgs), dim3(256), 0, st, seq[bi].x, ts, P.kv, P.blk[bi],
                         seq[bi].o, ts, n_items, per_wg, fs_o, pl_o, wgs, seq[bj].x, kv2, P.blk[bj], seq[bj].o);
      if (pairup) bi++;
      continue;
    }
    hipLaunchKernelGGL(k_attn_kv, dim3(n), dim3(64 * kKvWaves), 0, st, seq[bi].s, ts, P.blk[bi], P.kv);
    hipLaunchKernelGGL(k_attn_update, dim3(upd_blocks, n), dim3(256), 0, st, seq[bi].x, ts, P.kv, P.blk[bi], seq[bi].o, ts);
  }
  if (ev) hipEventRecord(ev[2], st);
  // ---- matching head on (f0, f1)
  float* f0s = P.fsc;
  float* f1s = P.fsc + (long long)P.max_pairs * ts;
  __bf16* p0 = P.split_bf16 ? P.fsp : nullptr;      // split path: the scaled features also as three bf16 planes
  __bf16* p1 = P.split_bf16 ? P.fsp + (long long)P.max_pairs * 3 * ts : nullptr;
  if (!P.split_bf16) {      // (the split path's last two attention blocks wrote f0s / f1s and the planes themselves)
    hipLaunchKernelGGL(k_scale_feats, dim3((unsigned)((n * ts + 255) / 256)), dim3(256), 0, st, f0, f0s, n * ts, p0, ts);
    hipLaunchKernelGGL(k_scale_feats, dim3((unsigned)((n * ts + 255) / 256)), dim3(256), 0, st, f1, f1s, n * ts, p1, ts);
  }
  const int head_blocks = (NTOK / 16 + 3) / 4;
  const dim3 grid3((NTOK / 16 / kSimRT + kSimWaves - 1) / kSimWaves, n), block3(64 * kSimWaves);
  // a call of a few pairs is latency-bound: one row tile per wave (75 waves per pair) instead of three (25)
  const bool few = n < 8;
  const bool sparse = threshold >= kCandMinThreshold && !P.dense_head;
  const bool single = p0 && !few && P.sim_single;
  const uint32_t* redo = nullptr;
  if (single) {
    // one evaluation of S gives both statistics; a pair whose sums left the f32 range is flagged and redone below
    hipMemsetAsync(P.sim_redo, 0, (size_t)n * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_pair_bound, dim3(n), dim3(256), 0, st, f0s, f1s, ts, P.gbound);
    const int n_items = n * kSimItemsPerPair;
    const dim3 gridi((n_items + kSimItemWaves - 1) / kSimItemWaves), blocki(64 * kSimItemWaves);
    hipLaunchKernelGGL(k_sim_single, gridi, blocki, 0, st, p0, p1, ts, P.gbound, P.rpart, P.cpart, n_items,
                       sparse ? P.sim_tmax : nullptr);
    hipLaunchKernelGGL(k_sim_finish, dim3((n * NTOK + 255) / 256), dim3(256), 0, st, P.gbound, P.rstats, P.rpart, P.cstats, 2LL * NTOK,
                       P.cpart, threshold, sparse ? P.lim : nullptr, P.sim_redo, n, P.sim_force_redo ? 1 : 0);
    redo = P.sim_redo;
  }
  // row statistics by the running-maximum pass: everything without the single pass, only the flagged pairs with it
  if (p0 && !few) hipLaunchKernelGGL(k_sim_stats3<false>, grid3, block3, 0, st, p0, p1, ts, P.rstats, 2LL * NTOK, nullptr, nullptr, nullptr, redo);
  else if (p0) hipLaunchKernelGGL((k_sim_stats<false, true>), dim3(head_blocks, n), dim3(256), 0, st, f0s, f1s, ts, P.rstats, 2LL * NTOK,
                                  nullptr, nullptr, nullptr, p0, p1, nullptr);
  else hipLaunchKernelGGL((k_sim_stats<false, false>), dim3(head_blocks, n), dim3(256), 0, st, f0s, f1s, ts, P.rstats, 2LL * NTOK,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (sparse) {
    // sparse path: the column pass lists the few entries per row that can pass, k_conf_cand evaluates them exactly
    hipMemsetAsync(P.cand_cnt, 0, (size_t)n * sizeof(uint32_t), st);
    hipMemsetAsync(P.mask, 0, (size_t)n * NTOK * MASK_WORDS * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_row_limits, dim3((n * NTOK + 255) / 256), dim3(256), 0, st, P.rstats, 2LL * NTOK, threshold, P.lim, n, redo);
    // (the emitting pass stays one row tile per wave: three per wave serialises the candidate appends, 330 -> 381 us)
    if (single) {
      const int n_items = n * kSimItemsPerPair;
      hipLaunchKernelGGL(k_sim_cand3, dim3((n_items + kSimItemWaves - 1) / kSimItemWaves), dim3(64 * kSimItemWaves), 0, st, p0, p1, ts,
                         P.lim, P.cand, P.cand_cnt, redo, n_items, P.sim_tmax, P.gbound);
      hipLaunchKernelGGL((k_sim_stats<true, true>), dim3(head_blocks, n), dim3(256), 0, st, f1s, f0s, ts, P.cstats, 2LL * NTOK,
                         P.lim, P.cand, P.cand_cnt, p1, p0, redo);      // flagged pairs only
    } else if (p0) hipLaunchKernelGGL((k_sim_stats<true, true>), dim3(head_blocks, n), dim3(256), 0, st, f1s, f0s, ts, P.cstats, 2LL * NTOK,
                                    P.lim, P.cand, P.cand_cnt, p1, p0, nullptr);
    else hipLaunchKernelGGL((k_sim_stats<true, false>), dim3(head_blocks, n), dim3(256), 0, st, f1s, f0s, ts, P.cstats, 2LL * NTOK,
                            P.lim, P.cand, P.cand_cnt, nullptr, nullptr, nullptr);
    // pair 0's confidence matrix (+ its mask, densely) is kept for the parity tests
    if (P.keep_debug)
      hipLaunchKernelGGL(k_conf_mask, dim3(head_blocks, 1), dim3(256), 0, st, f0s, f1s, ts, P.rstats, P.cstats, 2LL * NTOK,
                         threshold, P.mask, P.conf_dbg, 0);
    hipLaunchKernelGGL(k_conf_cand, dim3(8, n), dim3(256), 0, st, P.cand, P.cand_cnt, P.rstats, P.cstats, 2LL * NTOK, threshold,
                       P.mask);
  } else {
    if (p0 && !few) hipLaunchKernelGGL(k_sim_stats3<false>, grid3, block3, 0, st, p1, p0, ts, P.cstats, 2LL * NTOK, nullptr, nullptr,
                                       nullptr, redo);
    else if (p0) hipLaunchKernelGGL((k_sim_stats<false, true>), dim3(head_blocks, n), dim3(256), 0, st, f1s, f0s, ts, P.cstats,
                                    2LL * NTOK, nullptr, nullptr, nullptr, p1, p0, nullptr);
    else hipLaunchKernelGGL((k_sim_stats<false, false>), dim3(head_blocks, n), dim3(256), 0, st, f1s, f0s, ts, P.cstats,
                            2LL * NTOK, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_conf_mask, dim3(head_blocks, n), dim3(256), 0, st, f0s, f1s, ts, P.rstats, P.cstats, 2LL * NTOK,
                       threshold, P.mask, P.keep_debug ? P.conf_dbg : nullptr, 0);
  }
  hipLaunchKernelGGL(k_decode, dim3(n), dim3(kDecThreads), 0, st, P.mask, d_out, cap, d_n_out);
  if (P.keep_debug) {
    hipMemcpyAsync(P.feat_dbg, f0, ts * sizeof(float), hipMemcpyDeviceToDevice, st);
    hipMemcpyAsync(P.feat_dbg + ts, f1, ts * sizeof(float), hipMemcpyDeviceToDevice, st);
    P.have_dbg = true;
  }
  if (ev) { hipEventRecord(ev[3], st); P.ev_set_rec[P.ev_cur] = true; }
  return hipGetLastError();
}

int LoftrPipeline::stage_times(const char** names, float* ms, int cap) {
  static const char* kNames[3] = {"backbone_convs", "transformer", "match_head"};
  if (!p_ || !p_->ev_ok) return 0;
  Impl& P = *p_;
  for (int k = 0; k < Impl::kEvRing; k++) P.ev_harvest((P.ev_cur + 1 + k) % Impl::kEvRing);   // oldest first
  if (P.ev_acc_calls == 0) return 0;
  int n = 0;
  for (int i = 0; i < 3 && n < cap; i++, n++) {
    names[n] = kNames[i];
    ms[n] = P.ev_acc[i];
    P.ev_acc[i] = 0.f;
  }
  P.ev_acc_calls = 0;
  return n;
}

int LoftrPipeline::debug_get(int what, int slot, int level, void* host_out, size_t cap, size_t* n_bytes,
                             std::string* err) {
  if (!p_ || !p_->have_dbg) { *err = "no LoFTR batch has run yet, or the handle was created without MSF_FLAG_KEEP_DEBUG"; return MSF_ERR_INVALID_ARG; }
  if (slot != 0) { *err = "LoFTR debug tensors are kept for pair 0 of the last call only"; return MSF_ERR_INVALID_ARG; }
  if (hipDeviceSynchronize() != hipSuccess) { *err = "hipDeviceSynchronize failed"; return MSF_ERR_HIP; }
  const float* src = nullptr;
  size_t bytes = 0;
  if (what == MSF_DBG_LOFTR_CONF) { src = p_->conf_dbg; bytes = (size_t)NTOK * NTOK * 4; }
  else if (what == MSF_DBG_LOFTR_FEAT) { src = p_->feat_dbg; bytes = (size_t)2 * NTOK * DM * 4; }
  else if (what == MSF_DBG_LOFTR_ACT && level >= 0 && level < 4 && p_->act_dbg[level]) {
    const size_t act_elems[4] = {8u * 240 * 320, 16u * 120 * 160, 32u * 60 * 80, 32u * 30 * 40};
    src = p_->act_dbg[level];
    bytes = act_elems[level] * 4;
  }
  else { *err = "unknown debug item for LoFTR"; return MSF_ERR_INVALID_ARG; }
  *n_bytes = bytes;
  const size_t n = bytes < cap ? bytes : cap;
  if (what == MSF_DBG_LOFTR_ACT && p_->act_split[level]) {
    // kept in split pixels ([y][hi | lo][channel block][x] x 8 bf16): back to the graph's NCHW f32, v = hi + lo
    const int C = level == 0 ? 8 : 16, H = level == 0 ? 240 : 120, W = level == 0 ? 320 : 160, CB = C / 8;
    std::vector<uint16_t> raw(bytes / 2);
    if (hipMemcpy(raw.data(), src, bytes, hipMemcpyDeviceToHost) != hipSuccess) { *err = "hipMemcpy failed"; return MSF_ERR_HIP; }
    std::vector<float> nchw(bytes / 4);
    auto bf = [](uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; };
    for (int y = 0; y < H; y++)
      for (int cb = 0; cb < CB; cb++)
        for (int x = 0; x < W; x++)
          for (int k = 0; k < 8; k++) {
            const size_t hi = ((((size_t)y * 2 + 0) * CB + cb) * W + x) * 8 + k, lo = ((((size_t)y * 2 + 1) * CB + cb) * W + x) * 8 + k;
            nchw[((size_t)(8 * cb + k) * H + y) * W + x] = bf(raw[hi]) + bf(raw[lo]);
          }
    memcpy(host_out, nchw.data(), n);
    return 0;
  }
  if (n && hipMemcpy(host_out, src, n, hipMemcpyDeviceToHost) != hipSuccess) { *err = "hipMemcpy failed"; return MSF_ERR_HIP; }
  return 0;
}

}  // namespace msf
