// k2b_probe.hip -- K2b's nominate pass alone, on synthetic rows, under the timing switches of
// vt_batch_bf16.hip (what does the barrier / the query DMA / the MFMAs / the append cost?).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVT_BATCH_TIMING_EXPERIMENTS -Ivettore_amd/csrc tools/k2b_probe.hip -o tools/k2b_probe
// Run:   tools/k2b_probe [rows] [d] [tau]      one line per switch combination
#define VT_ENV_IMPLEMENTATION  // (this program's own copy of the library's settings table: csrc/vt_env.h)
#include "../vettore_amd/csrc/vt_batch_bf16.hip"

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
  const float tau = argc > 3 ? (float)atof(argv[3]) : 1e30f;
  const uint32_t ld = vt::padded_dim(d);
  float *X, *Q, *dtau;
  void *img;
  vt::BatchCand *cand;
  uint32_t *cnt;
  CK(hipMalloc(&X, (size_t)rows * ld * 4));
  CK(hipMalloc(&Q, (size_t)256 * ld * 4));
  CK(hipMalloc(&img, vt::batch_bf16_image_bytes(ld)));
  CK(hipMalloc(&dtau, 256 * 4));
  CK(hipMalloc(&cand, (size_t)256 * 8192 * sizeof(vt::BatchCand)));
  CK(hipMalloc(&cnt, 256 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, (size_t)rows * ld, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, Q, (size_t)256 * ld, 2u);
  std::vector<float> ht(256, tau);
  CK(hipMemcpy(dtau, ht.data(), 256 * 4, hipMemcpyHostToDevice));
  CK(vt::launch_batch_q_image(Q, ld, 256, img, 0));
  vt::BatchScoreArgs a{};
  a.X = X; a.stride = ld; a.Q = Q; a.ld = ld; a.nq_pad = 256; a.n = rows; a.n_total = rows;
  a.tau = dtau; a.cand = cand; a.cand_count = cnt; a.cand_cap = 8192; a.Qimage = img;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const uint32_t ntiles = (rows + 255) / 256;
    const uint32_t debugs[] = {0, 8, 1, 1 | 4, 1 | 2, 4, 16, 16 | 4, 0};
  // clocks: the first few hundred milliseconds after start-up run slow
  a.debug = 0;
  for (int rep = 0; rep < 100; ++rep) CK(vt::launch_batch_scores_bf16(a, false, std::min<uint32_t>(ntiles, 256), 0));
  CK(hipDeviceSynchronize());
  for (uint32_t dbg : debugs) {
    a.debug = dbg;
    float best = 1e30f;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipMemset(cnt, 0, 256 * 4));
      CK(hipEventRecord(e0, 0));
      CK(vt::launch_batch_scores_bf16(a, false, std::min<uint32_t>(ntiles, 256), 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    std::vector<uint32_t> hc(256);
    CK(hipMemcpy(hc.data(), cnt, 256 * 4, hipMemcpyDeviceToHost));
    uint64_t tot = 0;
    for (uint32_t v : hc) tot += v;
    printf("{\"debug\": %u, \"rows\": %u, \"d\": %u, \"ms\": %.4f, \"GBps\": %.1f, \"TFLOPs\": %.1f, \"appends\": %llu}\n", dbg, rows, d, best,
           (double)rows * d * 4 / (best * 1e-3) / 1e9, 2.0 * rows * 256.0 * ld / (best * 1e-3) / 1e12, (unsigned long long)tot);
    fflush(stdout);
  }
  return 0;
}
