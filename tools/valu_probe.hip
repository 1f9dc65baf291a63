// tools/valu_probe.hip -- issue cost of the vector instructions K1m is made of, on gfx950, at 1, 2
// and 3 waves per SIMD: cycles per instruction and SIMD (s_memtime ticks = shader cycles).
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip && tools/valu_probe
// Diagnostic only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define BODYS(INS)                                                                        \
  for (int it = 0; it < iters; ++it) {                                                    \
    REP8(asm volatile(INS : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7) : "v"(c0), "v"(c1));) \
  }
#define BODY(INS)                                                                         \
  for (int it = 0; it < iters; ++it) {                                                    \
    REP8(asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));) \
  }

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, unsigned long long *cycles) {
  f32x2 a0{1.f, 1.f}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  f32x2 b0{1.0001f, 0.9999f}, b1{0.5f, 2.f};
  b0.x += threadIdx.x * 1e-9f;
  float s0 = 1.f, s1 = 1.f, s2 = 1.f, s3 = 1.f, s4 = 1.f, s5 = 1.f, s6 = 1.f, s7 = 1.f, c0 = b0.x, c1 = b1.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  // eight independent destinations per line, eight lines per iteration = 64 instructions
  if (KIND == 0) BODYS("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n")
  if (KIND == 1) BODY("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
  if (KIND == 2) BODY("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
  if (KIND == 3) BODYS("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
  if (KIND == 4) BODYS("v_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
  if (KIND == 5) BODY("v_pk_mul_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,1]\n")
  if (KIND == 6) BODYS("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
  if (KIND == 8) BODYS("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n")
  if (KIND == 9) BODYS("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n")
  if (KIND == 7) BODYS("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
  const unsigned long long t1 = __builtin_readcyclecounter();
  const f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char *name, float *out, unsigned long long *cyc, int cus) {
  const int iters = 20000;
  std::printf("%-28s", name);
  for (int per_cu = 1; per_cu <= 3; ++per_cu) {
    const int blocks = cus * per_cu;  // 256 threads = one wave per SIMD per block
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // instructions per SIMD = per_cu waves x iters x 64
    std::printf("  %d w/SIMD: %6.2f ticks/instr/SIMD (%.3f ms)", per_cu, avg / ((double)per_cu * iters * 64.0), ms);
  }
  std::printf("\n");
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, (size_t)cus * 3 * 256 * sizeof(float));
  hipMalloc(&cyc, (size_t)cus * 3 * sizeof(unsigned long long));
  std::printf("%d CUs, clock %d kHz; ticks are s_memtime counts (their rate against the shader clock: compare rows)\n", cus, p.clockRate);
  run<0>("v_mul_f32", out, cyc, cus);
  run<3>("v_add_f32", out, cyc, cus);
  run<7>("v_fma_f32", out, cyc, cus);
  run<4>("v_add_f32_dpp quad_perm", out, cyc, cus);
  run<6>("v_cndmask_b32 (vcc)", out, cyc, cus);
  run<8>("v_cndmask_b32_e64 (sgpr pair)", out, cyc, cus);
  run<9>("v_xor_b32", out, cyc, cus);
  run<1>("v_pk_mul_f32", out, cyc, cus);
  run<5>("v_pk_mul_f32 op_sel", out, cyc, cus);
  run<2>("v_pk_add_f32", out, cyc, cus);
  return 0;
}
