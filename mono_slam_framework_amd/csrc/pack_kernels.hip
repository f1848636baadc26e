// Packs per-pair match lists [n][cap] + counts [n] into one contiguous list + offsets [n+1] on the
// device: the payload of the multi-GPU gather and of the adapter's result copy.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "msf_abi.h"

namespace msf {

// single workgroup: exclusive scan of min(max(cnt,0), cap)
__global__ __launch_bounds__(1024) void k_pack_scan(int n, const int32_t* cnt, int cap, int32_t* offsets) {
  __shared__ int32_t part[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = min(n, lo + per);
  int32_t s = 0;
  for (int i = lo; i < hi; i++) s += min(max(cnt[i], 0), cap);
  part[tid] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int32_t v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int32_t run = tid ? part[tid - 1] : 0;
  for (int i = lo; i < hi; i++) {
    offsets[i] = run;
    run += min(max(cnt[i], 0), cap);
  }
  if (tid == 1023) offsets[n] = part[1023];
}

__global__ __launch_bounds__(256) void k_pack_copy(const msf_match* in, int cap, const int32_t* offsets, msf_match* packed) {
  const int p = blockIdx.x;
  const int32_t o = offsets[p], m = offsets[p + 1] - o;
  const int4* src = reinterpret_cast<const int4*>(in + (long long)p * cap);
  int4* dst = reinterpret_cast<int4*>(packed + o);
  for (int i = threadIdx.x; i < m; i += 256) dst[i] = src[i];
}

hipError_t pack_matches(int n, const msf_match* d_in, int cap, const int32_t* d_cnt, msf_match* d_packed,
                        int32_t* d_offsets, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, st, n, d_cnt, cap, d_offsets);
  hipLaunchKernelGGL(k_pack_copy, dim3(n), dim3(256), 0, st, d_in, cap, d_offsets, d_packed);
  return hipGetLastError();
}

// KeyFrameMatchDatabase::DetectLoopCandidate's inner loop (slam_pipeline/src/KeyFrameDatabase.cc:37-44): the number
// of matches whose two endpoints both carry a map point.  KeyPointMap::GetMapPoint (KeyPointMap.cc:56-87) reduces to
// an exact lookup of the key y*cols + x (its neighbourhood loop re-reads the centre), out-of-image points have none.
// One wave per pair; occupancy is a bitmap per frame.
__global__ __launch_bounds__(256) void k_count_mp(int n, const msf_match* __restrict__ matches, int cap,
                                                  const int32_t* __restrict__ cnt, const int32_t* __restrict__ map_a,
                                                  const int32_t* __restrict__ map_b, const uint32_t* __restrict__ maps,
                                                  int n_maps, int map_words, int width, int height,
                                                  int32_t* __restrict__ num_mp) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= n) return;
  const int m = min(max(cnt[p], 0), cap);
  const int sa = map_a[p], sb = map_b[p];
  int c = 0;
  if (sa >= 0 && sa < n_maps && sb >= 0 && sb < n_maps) {
    const uint32_t* A = maps + (long long)sa * map_words;
    const uint32_t* B = maps + (long long)sb * map_words;
    const int4* src = reinterpret_cast<const int4*>(matches + (long long)p * cap);
    for (int i = lane; i < m; i += 64) {
      const int4 q = src[i];   // x1, y1, x2, y2
      bool hit = (unsigned)q.x < (unsigned)width && (unsigned)q.y < (unsigned)height &&
                 (unsigned)q.z < (unsigned)width && (unsigned)q.w < (unsigned)height;
      if (hit) {
        const int ka = q.y * width + q.x, kb = q.w * width + q.z;
        hit = ((A[ka >> 5] >> (ka & 31)) & 1u) && ((B[kb >> 5] >> (kb & 31)) & 1u);
      }
      c += hit;
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) num_mp[p] = c;
}

hipError_t count_mappoint_matches(int n, const msf_match* d_matches, int cap, const int32_t* d_cnt,
                                  const int32_t* d_map_a, const int32_t* d_map_b, const uint32_t* d_maps, int n_maps,
                                  int map_words, int width, int height, int32_t* d_num_mp, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_count_mp, dim3((n + 3) / 4), dim3(256), 0, st, n, d_matches, cap, d_cnt, d_map_a, d_map_b, d_maps,
                     n_maps, map_words, width, height, d_num_mp);
  return hipGetLastError();
}

// Tracking::CreateCurrentMatchImage (slam_pipeline/src/Tracking.cc:899-940): the two gray frames side by side as RGB,
// a filled radius-3 circle on every match end point -- first the matches without a map point on either side
// (0, 255, 0), then, over them, the ones with a map point on either side (255, 0, 0).
// cv::circle(img, c, 3, color, FILLED) sets, per the midpoint loop of OpenCV's Circle(): row c.y: x-3..x+3,
// rows c.y +- 1 and c.y +- 2: x-2..x+2, rows c.y +- 3: x only; spans are clipped to the image.
__global__ __launch_bounds__(256) void k_gray2rgb_pair(const uint8_t* __restrict__ f1, const uint8_t* __restrict__ f2,
                                                       int w, int h, long long pitch, uint8_t* __restrict__ out,
                                                       long long out_stride) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= 2 * w) return;
  const uint8_t v = x < w ? f1[(long long)y * pitch + x] : f2[(long long)y * pitch + (x - w)];
  uint8_t* o = out + (long long)y * out_stride + 3ll * x;
  o[0] = v; o[1] = v; o[2] = v;      // cvtColor(GRAY2RGB)
}

__global__ __launch_bounds__(256) void k_match_circles(const msf_match* __restrict__ m, const uint8_t* __restrict__ mp1,
                                                       const uint8_t* __restrict__ mp2, int n, int want_mp, int w, int h,
                                                       uint8_t* __restrict__ out, long long out_stride) {
  // 29 pixels per circle, two circles per match
  constexpr int kPix = 29;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int i = idx / (2 * kPix), r = idx - i * (2 * kPix);
  if (i >= n) return;
  const bool has = (mp1 && mp1[i]) || (mp2 && mp2[i]);
  if ((int)has != want_mp) return;
  const int side = r / kPix, p = r - side * kPix;
  // pixel p of the shape: rows -3..3 with half widths 0, 2, 2, 3, 2, 2, 0 (1 + 5 + 5 + 7 + 5 + 5 + 1 = 29)
  int dy, dx;
  if (p < 1) { dy = -3; dx = 0; }
  else if (p < 6) { dy = -2; dx = p - 1 - 2; }
  else if (p < 11) { dy = -1; dx = p - 6 - 2; }
  else if (p < 18) { dy = 0; dx = p - 11 - 3; }
  else if (p < 23) { dy = 1; dx = p - 18 - 2; }
  else if (p < 28) { dy = 2; dx = p - 23 - 2; }
  else { dy = 3; dx = 0; }
  const msf_match q = m[i];
  const int cx = side ? q.x2 : q.x1, cy = side ? q.y2 : q.y1;
  const int x = cx + dx, y = cy + dy;
  if (x < 0 || x >= w || y < 0 || y >= h) return;   // clipped to the half image the circle is drawn into
  uint8_t* o = out + (long long)y * out_stride + 3ll * (x + (side ? w : 0));
  o[0] = want_mp ? 255 : 0; o[1] = want_mp ? 0 : 255; o[2] = 0;
}

hipError_t render_match_image(const uint8_t* d_f1, const uint8_t* d_f2, int w, int h, long long pitch,
                              const msf_match* d_m, const uint8_t* d_mp1, const uint8_t* d_mp2, int n, uint8_t* d_out,
                              long long out_stride, hipStream_t st) {
  hipLaunchKernelGGL(k_gray2rgb_pair, dim3((2 * w + 255) / 256, h), dim3(256), 0, st, d_f1, d_f2, w, h, pitch, d_out,
                     out_stride);
  if (n > 0) {
    const int blocks = (n * 58 + 255) / 256;
    hipLaunchKernelGGL(k_match_circles, dim3(blocks), dim3(256), 0, st, d_m, d_mp1, d_mp2, n, 0, w, h, d_out, out_stride);
    hipLaunchKernelGGL(k_match_circles, dim3(blocks), dim3(256), 0, st, d_m, d_mp1, d_mp2, n, 1, w, h, d_out, out_stride);
  }
  return hipGetLastError();
}

}  // namespace msf
