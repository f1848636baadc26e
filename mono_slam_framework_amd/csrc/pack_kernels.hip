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

}  // namespace msf
