// tau_check.hip -- sample_tau_kernel (vt_batch.hip) against a sort on the host: the rank-th largest value of every row,
// ties and short samples included (fewer values than the rank: the smallest), ranks 1..40 (the small-rank shortcut and
// the histogram path), and its time at the batch path's shape (256 rows of 65 536).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivettore_amd/csrc tools/tau_check.hip -o tools/tau_check
#define VT_ENV_IMPLEMENTATION  // (this program's own copy of the library's settings table: csrc/vt_env.h)
#include "../vettore_amd/csrc/vt_batch.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  std::mt19937 rng(7);
  const uint32_t sizes[] = {1, 5, 63, 64, 1000, 4097, 65536};
  const uint32_t ranks[] = {1, 2, 3, 6, 7, 12, 16, 17, 27, 40};
  const uint32_t nq = 8;
  float *dS, *dT;
  CK(hipMalloc(&dS, (size_t)256 * 65536 * 4));
  CK(hipMalloc(&dT, 256 * 4));
  int bad = 0, checks = 0;
  for (int style = 0; style < 4; ++style)
    for (uint32_t n : sizes)
      for (uint32_t rank : ranks) {
        // (the histogram path, rank > 16, is only ever asked for a rank the sample can give: batch_group and funnel_group
        // clamp it to the sample's size; with fewer values it answers NaN -- no candidates, the query goes alone)
        if (rank > 16 && rank > n) continue;
        std::vector<float> h((size_t)nq * n);
        for (auto &v : h) {
          float x = std::uniform_real_distribution<float>(-50.f, 50.f)(rng);
          if (style == 1) x = std::round(x);                 // many ties
          if (style == 2) x = std::round(x / 25.f) * 25.f;   // a handful of distinct values
          if (style == 3 && (rng() % 7) == 0) x = -INFINITY; // invalid rows of a sample
          v = x;
        }
        if (style == 2 && n >= 64)
          for (uint32_t q = 0; q < nq; ++q)
            for (uint32_t j = 0; j < 8; ++j) h[(size_t)q * n + (rng() % 64)] = 75.f;  // several of the best in one thread's slots
        CK(hipMemcpy(dS, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(vt::launch_sample_tau(dS, n, nq, nq - 1, rank, dT, 0));  // (the last row is a padding column: +inf)
        float got[8];
        CK(hipMemcpy(got, dT, nq * 4, hipMemcpyDeviceToHost));
        for (uint32_t q = 0; q < nq; ++q) {
          float want;
          if (q == nq - 1) {
            want = INFINITY;
          } else {
            std::vector<float> row(h.begin() + (size_t)q * n, h.begin() + (size_t)(q + 1) * n);
            auto ord = [](float f) {  // f32::total_cmp as an unsigned key (the kernel's order: -0 below +0)
              uint32_t u;
              std::memcpy(&u, &f, 4);
              return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            };
            std::sort(row.begin(), row.end(), [&](float a, float b) { return ord(a) > ord(b); });
            want = row[std::min<uint32_t>(rank, n) - 1];
          }
          ++checks;
          if (std::memcmp(&want, &got[q], 4) != 0) {
            if (bad < 10) fprintf(stderr, "MISMATCH style %d n %u rank %u q %u: want %g got %g\n", style, n, rank, q, want, got[q]);
            ++bad;
          }
        }
      }
  printf("{\"checks\": %d, \"mismatches\": %d}\n", checks, bad);
  // sample_tau_groups_kernel (r05: the threshold from group maxima, <= 1 024 values per query, one wave per query):
  // the same question, the same answers -- the rank-th largest by total order, the smallest when there are fewer
  {
    int gbad = 0, gchecks = 0;
    const uint32_t gsizes[] = {1, 5, 63, 64, 65, 316, 1000, 1024};
    const uint32_t granks[] = {1, 2, 3, 6, 7, 12, 16, 27, 40, 64};
    const uint32_t gq = 9;  // (a block holds four waves: three blocks, the last one short)
    for (int style = 0; style < 4; ++style)
      for (uint32_t n : gsizes)
        for (uint32_t rank : granks) {
          std::vector<float> h((size_t)gq * n);
          for (auto &v : h) {
            float x = std::uniform_real_distribution<float>(-50.f, 50.f)(rng);
            if (style == 1) x = std::round(x);
            if (style == 2) x = std::round(x / 25.f) * 25.f;
            if (style == 3 && (rng() % 7) == 0) x = -INFINITY;
            if (style == 2 && (rng() % 11) == 0) x = (rng() & 1) ? 0.0f : -0.0f;
            v = x;
          }
          CK(hipMemcpy(dS, h.data(), h.size() * 4, hipMemcpyHostToDevice));
          CK(vt::launch_sample_tau_groups(dS, n, gq, gq - 1, rank, dT, 0));
          float got[9];
          CK(hipMemcpy(got, dT, gq * 4, hipMemcpyDeviceToHost));
          for (uint32_t q = 0; q < gq; ++q) {
            float want = INFINITY;
            if (q != gq - 1) {
              std::vector<float> row(h.begin() + (size_t)q * n, h.begin() + (size_t)(q + 1) * n);
              auto ord = [](float f) {
                uint32_t u;
                std::memcpy(&u, &f, 4);
                return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
              };
              std::sort(row.begin(), row.end(), [&](float a, float b) { return ord(a) > ord(b); });
              want = row[std::min<uint32_t>(rank, n) - 1];
            }
            ++gchecks;
            if (std::memcmp(&want, &got[q], 4) != 0) {
              if (gbad < 10) fprintf(stderr, "GROUPS MISMATCH style %d n %u rank %u q %u: want %g got %g\n", style, n, rank, q, want, got[q]);
              ++gbad;
            }
          }
        }
    printf("{\"groups_checks\": %d, \"groups_mismatches\": %d}\n", gchecks, gbad);
    bad += gbad;
    std::vector<float> h((size_t)256 * 1024);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : h) v = nd(rng);
    CK(hipMemcpy(dS, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      CK(hipEventRecord(e0, 0));
      CK(vt::launch_sample_tau_groups(dS, 1024, 256, 256, 6, dT, 0));
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("{\"groups\": 1024, \"rank\": 6, \"rows\": 256, \"us\": %.1f}\n", best * 1e3);
  }
  // time: 256 rows of 65 536 normal-ish values, rank 6 and rank 27
  {
    std::vector<float> h((size_t)256 * 65536);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : h) v = nd(rng);
    CK(hipMemcpy(dS, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (uint32_t rank : {6u, 27u})
      for (uint32_t rows : {8u, 256u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
          CK(hipEventRecord(e0, 0));
          CK(vt::launch_sample_tau(dS, 65536, rows, rows, rank, dT, 0));
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = std::min(best, ms);
        }
        printf("{\"rank\": %u, \"rows\": %u, \"us\": %.1f}\n", rank, rows, best * 1e3);
      }
  }
  return bad ? 2 : 0;
}
