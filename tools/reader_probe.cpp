// reader_probe.cpp -- native threads searching ONE handle through the C ABI (what BEAM dirty
// schedulers do): queries/s and mean latency with the coalescing of concurrent searches on and
// off (vt_debug_set "coalesce" 0), for several thread counts.  The Python twin (reader_probe.py) measures the
// same, but its threads also queue for the interpreter lock, which hides the cost of a hand-over
// on small corpora.
//   g++ -O2 -std=c++17 tools/reader_probe.cpp -Iinclude -Lvettore_amd/lib -lvettore_hip -lpthread \
//       -Wl,-rpath,$PWD/vettore_amd/lib -o tools/reader_probe_native
//   ./tools/reader_probe_native <rows> <dim> [seconds] [limit]
#include "vettore_flat.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000, d = argc > 2 ? strtoull(argv[2], nullptr, 10) : 768;
  const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
  const size_t limit = argc > 4 ? strtoull(argv[4], nullptr, 10) : 10;
  vt_flat *h = nullptr;
  if (vt_flat_new(VT_COSINE, 0, &h) != VT_OK) { fprintf(stderr, "vt_flat_new: %s\n", vt_last_error()); return 1; }
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  const size_t block = 100000;
  std::vector<float> x(std::min(rows, block) * d);
  std::string ids;
  std::vector<size_t> off;
  for (size_t r0 = 0; r0 < rows; r0 += block) {
    const size_t cnt = std::min(block, rows - r0);
    for (size_t i = 0; i < cnt * d; ++i) x[i] = u(rng);
    ids.clear();
    off.assign(1, 0);
    for (size_t i = 0; i < cnt; ++i) {
      char b[32];
      ids.append(b, (size_t)snprintf(b, sizeof b, "doc-%09zu", r0 + i));
      off.push_back(ids.size());
    }
    if (vt_flat_load_matrix(h, cnt, d, ids.data(), off.data(), x.data()) != VT_OK) { fprintf(stderr, "load: %s\n", vt_last_error()); return 1; }
  }
  std::vector<std::vector<float>> qs(256, std::vector<float>(d));
  for (auto &q : qs) for (auto &v : q) v = u(rng);
  { vt_hits *hits = nullptr; vt_flat_search(h, qs[0].data(), d, limit, &hits); vt_hits_free(hits); }
  const int thread_counts[] = {1, 2, 4, 8, 16, 32, 64};
  for (int T : thread_counts) {
    double qps[2] = {0, 0}, lat[2] = {0, 0};
    unsigned long long carried[2] = {0, 0};
    for (int mode = 0; mode < 2; ++mode) {
      vt_debug_set("coalesce", mode == 0 ? 1 : 0);  // (the library reads its environment once, at load: settings by name since r05)
      std::atomic<bool> stop{false};
      std::vector<unsigned long long> count(T, 0);
      std::vector<double> busy(T, 0.0);
      uint64_t b0 = 0, q0 = 0, b1 = 0, q1 = 0;
      vt_flat_coalesce_stats(h, &b0, &q0);
      std::vector<std::thread> th;
      const double t0 = now();
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          size_t i = t;
          while (!stop.load(std::memory_order_relaxed)) {
            vt_hits *hits = nullptr;
            const double a = now();
            if (vt_flat_search(h, qs[i % qs.size()].data(), d, limit, &hits) != VT_OK) { fprintf(stderr, "search: %s\n", vt_last_error()); exit(1); }
            busy[t] += now() - a;
            vt_hits_free(hits);
            count[t] += 1;
            i += T;
          }
        });
      std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
      stop = true;
      for (auto &t : th) t.join();
      const double dt = now() - t0;
      vt_flat_coalesce_stats(h, &b1, &q1);
      unsigned long long total = 0;
      double busy_total = 0;
      for (int t = 0; t < T; ++t) { total += count[t]; busy_total += busy[t]; }
      qps[mode] = total / dt;
      lat[mode] = busy_total / (total ? total : 1) * 1e3;
      carried[mode] = q1 - q0;
    }
    vt_debug_set("coalesce", 1);
    printf("{\"rows\": %zu, \"dim\": %zu, \"threads\": %d, \"coalesced_qps\": %.0f, \"coalesced_latency_ms\": %.3f, \"in_batches\": %llu, "
           "\"side_by_side_qps\": %.0f, \"side_by_side_latency_ms\": %.3f, \"gain\": %.2f}\n",
           rows, d, T, qps[0], lat[0], carried[0], qps[1], lat[1], qps[0] / (qps[1] > 0 ? qps[1] : 1));
    fflush(stdout);
  }
  vt_flat_free(h);
  return 0;
}
