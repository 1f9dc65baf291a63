// mfma_agpr.hip -- does it matter which register file the B operand of
// v_mfma_f32_32x32x2_f32 comes from?  (K2's compiler output keeps half of the query
// fragments in AGPRs.)  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool B_IN_AGPR>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float a = a0 + threadIdx.x;
  float b[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) b[t] = b0 + t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (B_IN_AGPR)
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "a"(b[t]));
        else
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b[t]));
      }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;
}

template <bool B>
int run(const char *name) {
  float *out;
  CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    k<B><<<blocks, 256>>>(out, iters, 1.0f, 2.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    if (rep == 2) printf("{\"kernel\": \"%s\", \"ms\": %.2f, \"TFLOPs\": %.1f}\n", name, ms, flops / ms / 1e9);
  }
  return 0;
}

int main() {
  run<false>("B operand in VGPR");
  run<true>("B operand in AGPR");
  return 0;
}
