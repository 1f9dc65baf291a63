// mfma_peak.hip -- what the FP32 matrix pipe sustains with nothing else going on:
// every wave issues independent v_mfma_f32_32x32x2_f32 back to back from registers.
// The yardstick beside the 157.3 TFLOP/s spec figure for K2 (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;
}

// the same stream with one ds_read_b128 after every MFMA of the first group of 8 in each
// 32 (K2's fragment reads), results consumed as B operands of the next iteration
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int READS>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters, float a0) {
  __shared__ __align__(16) float lds[4 * 64 * 4 * 9];
  for (int i = threadIdx.x; i < 4 * 64 * 4 * 9; i += 256) lds[i] = 1.0f;
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  f32x4 q[8], qn[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) q[t] = f32x4{1, 1, 1, 1};
  const float *base = lds + threadIdx.x * 4;
  float a = a0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, q[t][e], acc[t], 0, 0, 0);
        if (e == 0 && t < (READS < 8 ? READS : 8)) qn[t] = *reinterpret_cast<const f32x4 *>(base + t * 1024);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
    for (int t = 0; t < (READS < 8 ? READS : 8); ++t) q[t] = qn[t];
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;
}

template <int READS>
int run_lds(int waves_per_simd = 1) {
  float *out;
  CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000, blocks = 256 * waves_per_simd;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    k_lds<READS><<<blocks, 256>>>(out, iters, 1.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    if (rep == 2) printf("{\"kernel\": \"32 MFMA + %d ds_read_b128 per iteration\", \"waves_per_simd\": %d, \"ms\": %.2f, \"TFLOPs\": %.1f}\n", READS, waves_per_simd, ms, flops / ms / 1e9);
  }
  return 0;
}

// the bare stream again, but with operands that look like data (a different pseudo-random
// value per lane and per MFMA, accumulators away from zero): power, not issue, may set the rate
__global__ __launch_bounds__(256) void k_rand(float *out, int iters, unsigned seed) {
  f32x16 acc[8];
  unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  auto rnd = [&]() {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return (float)(int)(x & 0xFFFFFF) * (1.0f / 8388608.0f) - 1.0f;  // uniform(-1, 1)
  };
  float a[16], b[32];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = rnd();
#pragma unroll
  for (int i = 0; i < 32; ++i) b[i] = rnd();
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = rnd();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)  // all register indices are compile-time constants
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 8; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u * 4 + e], b[(e * 8 + t + u * 5) & 31], acc[t], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  if (s == 12345.678f) out[0] = s;
}

int run_rand() {
  float *out;
  CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 40000, blocks = 256;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    k_rand<<<blocks, 256>>>(out, iters, 12345u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    if (rep == 2) printf("{\"kernel\": \"32 MFMA per iteration, pseudo-random operands\", \"ms\": %.2f, \"TFLOPs\": %.1f}\n", ms, flops / ms / 1e9);
  }
  return 0;
}

template <int NACC>
int run(const char *name, int waves_per_simd) {
  float *out;
  CK(hipMalloc((void **)&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 4 waves per block = one per SIMD
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    k<NACC><<<blocks, 256>>>(out, iters, 1.0f, 2.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 4 * iters * 4 * NACC * 4096.0;
    if (rep == 2) printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.2f, \"TFLOPs\": %.1f}\n", name, waves_per_simd, ms, flops / ms / 1e9);
  }
  return 0;
}

int main() {
  run<8>("mfma_f32_32x32x2_f32 x8 independent accumulators", 1);
  run<8>("mfma_f32_32x32x2_f32 x8 independent accumulators", 2);
  run<4>("mfma_f32_32x32x2_f32 x4 independent accumulators", 1);
  run<2>("mfma_f32_32x32x2_f32 x2 independent accumulators", 1);
  run_rand();
  run_lds<0>();
  run_lds<4>();
  run_lds<8>();
  run_lds<9>(1);
  run_lds<0>(2);
  run_lds<8>(2);
  return 0;
}
