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

}  // namespace msf
