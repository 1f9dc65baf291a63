// k2s_probe.hip -- the bf16 nomination pass alone, on synthetic rows: K2s (from the bf16 shadow,
// vt_batch_shadow.hip) beside K2b (from the f32 rows, vt_batch_bf16.hip), interleaved in one process,
// and K2s under its timing switches (what do the barrier / the DMA / the fragment reads / the MFMAs cost?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVT_BATCH_TIMING_EXPERIMENTS -Ivettore_amd/csrc tools/k2s_probe.hip -o tools/k2s_probe
// Run:   tools/k2s_probe [rows] [d] [nq_pad] [tau]      one JSON line per variant
#define VT_ENV_IMPLEMENTATION  // (this program's own copy of the library's settings table: csrc/vt_env.h)
#include "../vettore_amd/csrc/vt_batch_bf16.hip"
#include "../vettore_amd/csrc/vt_batch_shadow.hip"

#include <cmath>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(float *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFF) / 32768.0f - 1.0f;
  }
}

int main(int argc, char **argv) {
  const uint32_t rows = argc > 1 ? (uint32_t)atoll(argv[1]) : 4000000u;
  const uint32_t d = argc > 2 ? (uint32_t)atoi(argv[2]) : 768u;
  const uint32_t nq_pad = argc > 3 ? (uint32_t)atoi(argv[3]) : 256u;
  const float tau = argc > 4 ? (float)atof(argv[4]) : 1e30f;
  const uint32_t ld = vt::padded_dim(d);
  const uint32_t rows_img = (rows + 255) / 256 * 256;
  float *X, *Q, *dtau, *sampA, *sampB;
  void *imgB, *imgS, *shadow;
  vt::BatchCand *cand;
  uint32_t *cnt;
  CK(hipMalloc(&X, (size_t)rows_img * ld * 4));
  CK(hipMalloc(&shadow, (size_t)rows_img * ld * 2));
  CK(hipMalloc(&Q, (size_t)256 * ld * 4));
  CK(hipMalloc(&imgB, vt::batch_bf16_image_bytes(ld)));
  CK(hipMalloc(&imgS, vt::batch_shadow_image_bytes(ld)));
  CK(hipMalloc(&dtau, 256 * 4));
  CK(hipMalloc(&cand, (size_t)256 * 8192 * sizeof(vt::BatchCand)));
  CK(hipMalloc(&cnt, 256 * 4));
  const uint32_t srows = 16384;
  CK(hipMalloc(&sampA, (size_t)256 * srows * 4));
  CK(hipMalloc(&sampB, (size_t)256 * srows * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, (size_t)rows_img * ld, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, Q, (size_t)256 * ld, 2u);
  std::vector<float> ht(256, tau);
  CK(hipMemcpy(dtau, ht.data(), 256 * 4, hipMemcpyHostToDevice));
  CK(vt::launch_batch_q_image(Q, ld, nq_pad, imgB, 0));
  CK(vt::launch_batch_q_image16(Q, ld, nq_pad, imgS, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  {
    CK(hipEventRecord(e0, 0));
    CK(vt::launch_shadow_build(X, ld, rows, rows_img, ld, shadow, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"shadow_build_ms\": %.3f, \"rows\": %u, \"d\": %u, \"GBps_read\": %.1f}\n", ms, rows, d, (double)rows * ld * 4 / (ms * 1e-3) / 1e9);
  }
  vt::BatchScoreArgs a{};
  a.X = X; a.stride = ld; a.Q = Q; a.ld = ld; a.nq_pad = nq_pad; a.n = rows; a.n_total = rows;
  a.tau = dtau; a.cand = cand; a.cand_count = cnt; a.cand_cap = 8192; a.Xshadow = shadow;
  // ---- the two passes must nominate from the same scores: dense mode over the first 16 384 rows
  {
    vt::BatchScoreArgs s = a;
    s.n = srows; s.sample_stride = 1; s.sample_rows = srows;
    s.Qimage = imgB; s.sample = sampA;
    CK(vt::launch_batch_scores_bf16(s, true, 64, 0));
    s.Qimage = imgS; s.sample = sampB;
    CK(vt::launch_batch_scores_shadow(s, true, 64, 0));
    std::vector<float> ha((size_t)nq_pad * srows), hb((size_t)nq_pad * srows);
    CK(hipMemcpy(ha.data(), sampA, ha.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), sampB, hb.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, big = 0;
    for (size_t i = 0; i < ha.size(); ++i) {
      worst = std::max(worst, (double)std::fabs(ha[i] - hb[i]));
      big = std::max(big, (double)std::fabs(ha[i]));
    }
    printf("{\"dense_check\": \"K2s vs K2b\", \"max_abs_diff\": %.3g, \"max_abs_score\": %.3g}\n", worst, big);
    if (!(worst <= 1e-3 * std::max(1.0, big))) {
      fprintf(stderr, "K2s and K2b disagree\n");
      return 2;
    }
  }
  const uint32_t ntiles = (rows + 255) / 256;
  const uint32_t grid = std::min<uint32_t>(ntiles, 256);
  auto run = [&](bool shadow_pass, uint32_t dbg, float *best_out) -> int {
    a.debug = dbg;
    a.Qimage = shadow_pass ? imgS : imgB;
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
      CK(hipMemsetAsync(cnt, 0, 256 * 4, 0));
      CK(hipEventRecord(e0, 0));
      if (shadow_pass) CK(vt::launch_batch_scores_shadow(a, false, grid, 0));
      else CK(vt::launch_batch_scores_bf16(a, false, grid, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    *best_out = best;
    return 0;
  };
  // clocks: the first few hundred milliseconds after start-up run slow
  a.debug = 0;
  a.Qimage = imgS;
  for (int rep = 0; rep < 100; ++rep) CK(vt::launch_batch_scores_shadow(a, false, grid, 0));
  CK(hipDeviceSynchronize());
  for (int round = 0; round < 3; ++round) {
    float ms_b = 0, ms_s = 0;
    if (run(false, 0, &ms_b)) return 1;
    if (run(true, 0, &ms_s)) return 1;
    std::vector<uint32_t> hc(256);
    CK(hipMemcpy(hc.data(), cnt, 256 * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0;
    for (uint32_t v : hc) tot += v;
    printf("{\"round\": %d, \"rows\": %u, \"d\": %u, \"nq_pad\": %u, \"k2b_ms\": %.4f, \"k2b_TBps_f32\": %.2f, \"k2s_ms\": %.4f, \"k2s_TBps_bf16\": %.2f, "
           "\"k2s_PFLOPs\": %.3f, \"appends\": %llu}\n",
           round, rows, d, nq_pad, ms_b, (double)rows * d * 4 / (ms_b * 1e-3) / 1e12, ms_s, (double)rows * d * 2 / (ms_s * 1e-3) / 1e12,
           2.0 * rows * (double)nq_pad * ld / (ms_s * 1e-3) / 1e15, (unsigned long long)tot);
    fflush(stdout);
  }
  // K2s under its timing switches (256 columns, five stages only)
  if (nq_pad == 256) {
    const uint32_t debugs[] = {0, 8, 2, 8 | 2, 1, 1 | 2, 1 | 4, 16, 16 | 4, 16 | 4 | 2, 32 | 16, 32 | 16 | 4, 32 | 16 | 4 | 2, 0};
    for (uint32_t dbg : debugs) {
      float ms = 0;
      if (run(true, dbg, &ms)) return 1;
      printf("{\"k2s_debug\": %u, \"ms\": %.4f, \"TBps_bf16\": %.2f, \"PFLOPs\": %.3f}\n", dbg, ms,
             (double)rows * d * 2 / (ms * 1e-3) / 1e12, 2.0 * rows * (double)nq_pad * ld / (ms * 1e-3) / 1e15);
      fflush(stdout);
    }
  }
  return 0;
}
