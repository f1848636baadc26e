// Micro-benchmark (developer tool, not part of the product): issue cost of the VALU instructions the LoFTR epilogues
// could be rebuilt from -- v_add_f32, v_and + v_sub (today's bf16 unpack + subtract), v_dot2_f32_bf16 (bf16 half ->
// f32 add in one instruction), v_cvt_pk_bf16_f32, v_perm_b32 -- alone (4 waves per SIMD, independent chains) and beside a
// partner wave that issues v_mfma_f32_16x16x32_bf16 back to back.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, uint64_t* cyc, int iters, int mfma_waves) {
  const int wave = threadIdx.x >> 6;
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  uint32_t h = 0x3f803f80u + threadIdx.x;
  const uint32_t one = 0x00003f80u;
  f32x4 acc = {0, 0, 0, 0};
  bf16x8 fa, fb;
  for (int j = 0; j < 8; j++) { fa[j] = (__bf16)(float)(threadIdx.x & 3); fb[j] = (__bf16)1.f; }
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  if (wave < mfma_waves) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; it++) {
#define R8(X) X(a0) X(a1) X(a2) X(a3) X(a4) X(a5) X(a6) X(a7)
      if (OP == 0) {
#define X(a) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(a0));
        R8(X)
#undef X
      } else if (OP == 1) {
#define X(a) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a) : "v"(h), "v"(one));
        R8(X)
#undef X
      } else if (OP == 2) {
#define X(a) { uint32_t t; asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(t) : "v"(h)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(t)); }
        R8(X)
#undef X
      } else if (OP == 3) {
#define X(a) { uint32_t t; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(a0)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(h) : "v"(t)); }
        R8(X)
#undef X
      } else if (OP == 4) {
#define X(a) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(h), "v"(one));
        R8(X)
#undef X
      } else if (OP == 5) {
#define X(a) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(h));
        R8(X)
#undef X
      } else if (OP == 6) {
#define X(a) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a) : "v"(h), "v"(one));
        R8(X)
#undef X
      } else if (OP == 7) {
#define X(a) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a) : "v"(*(double*)&a0));
        X(a0) X(a2) X(a4) X(a6) X(a0) X(a2) X(a4) X(a6)
#undef X
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc[0] + __uint_as_float(h);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, uint64_t* cyc, uint64_t* hc) {
  const int iters = 4000, grid = 256 * 2;
  for (int mw : {0, 4}) {
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(512), 0, 0, out, cyc, iters, mw);
    hipDeviceSynchronize();
    hipMemcpy(hc, cyc, grid * 8 * 8, hipMemcpyDeviceToHost);
    double mv = 0, mm = 0;
    for (int b = 0; b < grid; b++)
      for (int w = 0; w < 8; w++) (w < mw ? mm : mv) += (double)hc[b * 8 + w];
    const int nv = grid * (8 - mw), nm = grid * mw;
    // a SIMD hosts 2 waves of this block (x2 blocks per CU if they fit): report wave-cycles per instruction
    printf("%-28s mfma partner waves %d: VALU wave %.2f cycles / instr", name, mw, mv / nv / (iters * 8.0));
    if (nm) printf("   | MFMA wave %.2f cycles / mfma", mm / nm / (iters * 8.0));
    printf("\n");
  }
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 512 * 512 * 4 * 2); hipMalloc(&cyc, 512 * 8 * 8 * 2);
  uint64_t* hc = new uint64_t[512 * 8 * 2];
  run<0>("v_add_f32", out, cyc, hc);
  run<1>("v_dot2_f32_bf16", out, cyc, hc);
  run<6>("v_dot2c_f32_bf16", out, cyc, hc);
  run<2>("v_and_b32 + v_sub_f32 (pair)", out, cyc, hc);
  run<3>("v_cvt_pk_bf16 + v_xor (pair)", out, cyc, hc);
  run<4>("v_perm_b32", out, cyc, hc);
  run<5>("v_max_i32", out, cyc, hc);
  run<7>("v_pk_add_f32", out, cyc, hc);
  return 0;
}
