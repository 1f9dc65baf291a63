// concurrency_check.cpp -- the host side's thread machinery (vettore_amd/csrc/host/vt_concurrency.h)
// instantiated with stub operations and run under ThreadSanitizer (tests/test_concurrency.py):
//   1. the coalescer: 64 threads x mixed limits and query lengths x injected failures (a query that is
//      invalid on its own, a batch that fails as a whole, a batch that throws) x a "writer" that
//      forces the disband path; every caller must get the answer of ITS query, nobody may be left
//      waiting, the slot count must come back to zero;
//   2. the context lease: 48 readers on a pool of at most 8 contexts, no context ever held twice;
//   3. the workers: concurrent callers posting to 4 workers, every worker sees the callers' jobs in
//      the same order (what matching collectives need), failures come back with their message;
//   4. the settings table (vettore_amd/csrc/vt_env.h): the path-choice reads of a search -- vt::env::get --
//      beside a thread that calls setenv / unsetenv all the time (what System.put_env/2 does to a NIF
//      library inside a BEAM) and one that stores settings: the table was filled once, before any of
//      them started, and a read never goes back to the environment;
//   5. a second context beside the one a reader holds (SpareLeaseT): never the primary, never held
//      twice, never a wait -- 48 readers that each hold one and ask for another must all come through.
// TEST INFRASTRUCTURE.  Prints "ok" and exits 0, or says what went wrong.
#include "../vettore_amd/csrc/host/vt_concurrency.h"
#define VT_ENV_IMPLEMENTATION
#include "../vettore_amd/csrc/vt_env.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <new>
#include <random>

struct vt_hits {
  uint64_t tag;
};

static std::atomic<long> g_live_hits{0};
static vt_hits *make_hits(uint64_t tag) {
  g_live_hits += 1;
  return new vt_hits{tag};
}
static void free_hits(vt_hits *h) {
  g_live_hits -= 1;
  delete h;
}

#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::fprintf(stderr, "concurrency_check: %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      std::exit(1);                                                          \
    }                                                                        \
  } while (0)

// ------------------------------------------------------------------ 1. coalescer
struct FakeHandle {
  vt_host::Coalescer co;
  std::atomic<unsigned> slots{1};
  std::atomic<bool> ranks_lazy{false};     // the writer's doing: batches must disband
  std::atomic<size_t> hold{0};             // the test hook: an idle handle's first caller waits for this many
  std::atomic<uint64_t> alone{0}, batched{0}, batches{0}, failed_batches{0}, thrown{0};
};

static thread_local std::string t_last_error;

// the "answer" of a query: a function of its content and limit only
static uint64_t answer(const float *q, size_t n, size_t limit) {
  uint64_t h = 1469598103934665603ull ^ limit;
  for (size_t i = 0; i < n; ++i) {
    uint32_t b;
    std::memcpy(&b, &q[i], 4);
    h = (h ^ b) * 1099511628211ull;
  }
  return h;
}
static void spin(int us) { std::this_thread::sleep_for(std::chrono::microseconds(us)); }

struct FakeOps {
  static constexpr int kOutOfMemory = 9;
  static vt_host::Coalescer &coalescer(FakeHandle *h) { return h->co; }
  static unsigned slots(FakeHandle *h) { return h->slots.load(); }
  static int search_direct(FakeHandle *h, int kind, size_t aux, const float *query, size_t n, size_t limit, vt_hits **out) {
    limit += 1000 * (size_t)kind + 100000 * aux;  // (the answer depends on the entry point and its parameter too)
    if (query[0] < 0.0f) {  // "vector contains a non-finite value"
      t_last_error = "bad query";
      return 3;
    }
    spin(30);
    h->alone += 1;
    *out = make_hits(answer(query, n, limit));
    return 0;
  }
  static void search_alone(FakeHandle *h, vt_host::Waiting *w) {
    w->status = search_direct(h, w->kind, w->aux, w->query, w->n, w->limit, w->out);
    if (w->status != 0) w->error = t_last_error;
  }
  static void judge(FakeHandle *, std::vector<vt_host::Waiting *> &members, std::vector<vt_host::Waiting *> *good) {
    for (vt_host::Waiting *w : members) {
      if (w->query[0] < 0.0f) {
        w->status = 3;
        w->error = "bad query";
      } else {
        good->push_back(w);
      }
    }
  }
  static int batch(FakeHandle *h, int kind, size_t aux, const float *queries, size_t nq, size_t n, size_t limit, vt_hits **outs) {
    limit += 1000 * (size_t)kind + 100000 * aux;
    spin(60);
    // injected: a batch that fails as a whole (one member's "metric overflow"), a batch that throws
    const uint64_t roll = answer(queries, n, limit + nq);
    if (roll % 29u == 0u) {
      h->failed_batches += 1;
      return 4;
    }
    if (roll % 211u == 1u) {
      h->thrown += 1;
      throw std::bad_alloc();
    }
    for (size_t i = 0; i < nq; ++i) outs[i] = make_hits(answer(queries + i * n, n, limit));
    h->batched += nq;
    h->batches += 1;
    return 0;
  }
  static bool must_disband(FakeHandle *h, int, size_t) { return h->ranks_lazy.load(); }
  static size_t capacity(FakeHandle *, int kind) { return kind ? 8 : 256; }
  static size_t hold_until(FakeHandle *h) { return h->hold.load(); }
  static void run(FakeHandle *h, std::vector<vt_host::Waiting *> &members) { vt_host::run_coalesced_t<FakeHandle, FakeOps>(h, members); }
  static void drop_hits(vt_hits *hits) { free_hits(hits); }
  static void set_last_error(const std::string &msg) { t_last_error = msg; }
};

static void check_coalescer(int threads, int per_thread) {
  FakeHandle h;
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> ok{0}, bad{0}, oom{0};
  std::thread writer([&] {  // flips the disband condition and the slot count while searches run
    std::mt19937 rng(7);
    while (!stop.load()) {
      h.ranks_lazy = (rng() & 7u) == 0u;
      h.slots = 1 + (rng() % 3u);
      spin(200);
    }
    h.ranks_lazy = false;
  });
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      std::mt19937 rng(1000 + t);
      for (int i = 0; i < per_thread; ++i) {
        const size_t n = (rng() & 1u) ? 16 : 24;
        const size_t limit = 1 + (rng() % 3u);
        const int kind = (int)(rng() % 2u);                       // two entry points (flat_search, quantized_search) ...
        const size_t aux = kind ? 10 * (1 + rng() % 2u) : 0;      // ... the second with a parameter of its own
        std::vector<float> q(n);
        for (auto &v : q) v = (float)(rng() % 1000u) / 10.0f;
        if ((rng() % 50u) == 0u) q[0] = -1.0f;  // invalid on its own
        vt_hits *out = nullptr;
        const int st = vt_host::coalesced_search_t<FakeHandle, FakeOps>(&h, q.data(), n, limit, &out, kind, aux);
        if (q[0] < 0.0f) {
          CHECK(st == 3 && out == nullptr && t_last_error == "bad query");
          bad += 1;
        } else if (st == FakeOps::kOutOfMemory) {
          CHECK(out == nullptr);
          oom += 1;
        } else {
          CHECK(st == 0 && out != nullptr && out->tag == answer(q.data(), n, limit + 1000 * (size_t)kind + 100000 * aux));
          free_hits(out);
          ok += 1;
        }
      }
    });
  for (auto &th : pool) th.join();
  stop = true;
  writer.join();
  {
    std::lock_guard<std::mutex> g(h.co.mu);
    CHECK(h.co.waiting.empty() && h.co.active == 0);
  }
  CHECK(ok + bad + oom == (uint64_t)threads * per_thread);
  CHECK(g_live_hits.load() == 0);
  CHECK(h.batches.load() > 0 && h.alone.load() > 0);
  std::fprintf(stderr, "coalescer: %llu ok (%llu in %llu batches, %llu alone), %llu invalid, %llu after a thrown batch, %llu failed batches\n",
               (unsigned long long)ok.load(), (unsigned long long)h.batched.load(), (unsigned long long)h.batches.load(),
               (unsigned long long)h.alone.load(), (unsigned long long)bad.load(), (unsigned long long)oom.load(),
               (unsigned long long)h.failed_batches.load());
}

// the test hook (Ops::hold_until): `threads` callers that leave a barrier together must travel as ONE batch, round after
// round -- what tests/test_gpu_coalesce.py relies on when it asserts which kernels ran
static void check_forced_meeting(int threads, int rounds) {
  FakeHandle h;
  h.hold = (size_t)threads;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  auto barrier = [&] {
    std::unique_lock<std::mutex> g(mu);
    const int gen = generation;
    if (++arrived == threads) {
      arrived = 0;
      generation += 1;
      cv.notify_all();
    } else {
      cv.wait(g, [&] { return generation != gen; });
    }
  };
  std::atomic<uint64_t> ok{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (int r = 0; r < rounds; ++r) {
        std::vector<float> q(16);
        for (size_t i = 0; i < q.size(); ++i) q[i] = (float)(1 + t * 131 + r * 17 + (int)i);   // (no injected roll hits these)
        barrier();
        vt_hits *out = nullptr;
        const int st = vt_host::coalesced_search_t<FakeHandle, FakeOps>(&h, q.data(), q.size(), 5, &out, 0, 0);
        if (st == 0) {
          CHECK(out != nullptr && out->tag == answer(q.data(), q.size(), 5));
          free_hits(out);
          ok += 1;
        } else {
          CHECK(st == 4 || st == FakeOps::kOutOfMemory || out != nullptr);
        }
      }
    });
  for (auto &th : pool) th.join();
  {
    std::lock_guard<std::mutex> g(h.co.mu);
    CHECK(h.co.waiting.empty() && h.co.active == 0);
    // every round: one batch of everybody (a batch the stub fails or throws still counts as one that met)
    CHECK(h.co.batches == (uint64_t)rounds && h.co.batched_queries == (uint64_t)rounds * threads);
  }
  std::fprintf(stderr, "forced meeting: %d rounds of %d callers, %llu batches\n", rounds, threads, (unsigned long long)h.co.batches);
}

// ------------------------------------------------------------------ 2. lease
struct FakeCtx {
  std::atomic<int> holders{0};
  int device = 0;
};
struct FakeShard {
  FakeCtx ctx;
  std::mutex pool_mu;
  std::condition_variable pool_cv;
  std::vector<std::unique_ptr<FakeCtx>> extra;
  std::vector<FakeCtx *> free_ctx;
  bool ctx0_busy = false;
};

static void check_lease(int threads, int per_thread) {
  FakeShard s;
  std::atomic<int> made{0}, failed{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      std::mt19937 rng(t);
      for (int i = 0; i < per_thread; ++i) {
        vt_host::LeaseT<FakeShard, FakeCtx> lease(&s, 8, [&](FakeShard *, int *status) -> std::unique_ptr<FakeCtx> {
          if ((rng() % 97u) == 0u) {  // a context that cannot be made (out of device memory)
            *status = 9;
            failed += 1;
            return nullptr;
          }
          made += 1;
          return std::make_unique<FakeCtx>();
        });
        if (!lease.c) {
          CHECK(lease.status == 9);
          continue;
        }
        CHECK(lease.c->holders.fetch_add(1) == 0);  // never held twice
        if ((rng() & 3u) == 0u) spin(5);
        CHECK(lease.c->holders.fetch_sub(1) == 1);
      }
    });
  for (auto &th : pool) th.join();
  CHECK(s.extra.size() + 1 <= 8 && !s.ctx0_busy && s.free_ctx.size() == s.extra.size());
  CHECK(made.load() == (int)s.extra.size());
  std::fprintf(stderr, "lease: %d contexts made, %d refused\n", made.load(), failed.load());
}

// ------------------------------------------------------------------ 3. workers
struct TestWorkerPolicy {
  static void thread_start(int) {}
  static int run(const std::function<int()> &fn, std::string *error) {
    try {
      const int st = fn();
      if (st != 0) *error = "job failed with " + std::to_string(st);
      return st;
    } catch (...) {
      *error = "job threw";
      return 11;
    }
  }
};
using TestWorker = vt_host::WorkerT<TestWorkerPolicy>;

static void check_workers(int callers, int per_caller) {
  constexpr size_t S = 4;
  std::vector<std::unique_ptr<TestWorker>> workers;
  for (size_t s = 0; s < S; ++s) {
    workers.push_back(std::make_unique<TestWorker>());
    workers.back()->start((int)s);
  }
  std::mutex post_mu;
  std::vector<std::vector<uint64_t>> seen(S);  // each touched by its worker thread only
  std::atomic<int> failures{0};
  std::vector<std::thread> pool;
  for (int c = 0; c < callers; ++c)
    pool.emplace_back([&, c] {
      for (int i = 0; i < per_caller; ++i) {
        const uint64_t id = ((uint64_t)c << 32) | (uint32_t)i;
        const bool fails = (i % 37) == 5;
        std::vector<size_t> all{0, 1, 2, 3};
        std::string msg;
        const int st = vt_host::run_on_workers(workers, post_mu, all,
                                               [&](size_t s) -> int {
                                                 seen[s].push_back(id);
                                                 if (fails && s == 2) return 6;
                                                 if (fails && s == 3) throw std::bad_alloc();
                                                 return 0;
                                               },
                                               [&](int status, const std::string &error) {
                                                 msg = error;
                                                 return status;
                                               });
        if (fails) {
          CHECK(st == 6 && msg == "job failed with 6");  // the first failing shard's
          failures += 1;
        } else {
          CHECK(st == 0);
        }
      }
    });
  for (auto &th : pool) th.join();
  workers.clear();  // joins
  for (size_t s = 1; s < S; ++s) CHECK(seen[s] == seen[0]);  // one order for everybody
  CHECK(seen[0].size() == (size_t)callers * per_caller);
  std::fprintf(stderr, "workers: %zu jobs per worker in one order, %d failures reported\n", seen[0].size(), failures.load());
}

// ------------------------------------------------------------------ 3b. the posting thread put to use
// batch_multi's shape: every worker settles its items one by one (plain store, then a release flag); the caller
// consumes an item once every worker has flagged it, while the jobs still run; whatever is left when they end it
// takes afterwards.  Under TSan a read of an item that was not published yet is a report.
static void check_workers_meanwhile(int callers, int per_caller) {
  constexpr size_t S = 3, N = 64;
  std::vector<std::unique_ptr<TestWorker>> workers;
  for (size_t s = 0; s < S; ++s) {
    workers.push_back(std::make_unique<TestWorker>());
    workers.back()->start((int)s);
  }
  std::mutex post_mu;
  std::atomic<uint64_t> early{0}, late{0};
  std::vector<std::thread> pool;
  for (int c = 0; c < callers; ++c)
    pool.emplace_back([&, c] {
      for (int it = 0; it < per_caller; ++it) {
        std::vector<uint64_t> item(S * N, 0);
        std::unique_ptr<std::atomic<unsigned char>[]> fin(new std::atomic<unsigned char>[S * N]);
        for (size_t i = 0; i < S * N; ++i) fin[i].store(0, std::memory_order_relaxed);
        std::vector<char> taken(N, 0);
        uint64_t sum = 0;
        const bool fails = (it % 29) == 7;
        auto take_ready = [&]() -> bool {
          bool any = false;
          for (size_t i = 0; i < N; ++i) {
            if (taken[i]) continue;
            bool ready = true;
            for (size_t s = 0; s < S && ready; ++s) ready = fin[s * N + i].load(std::memory_order_acquire) != 0;
            if (!ready) continue;
            for (size_t s = 0; s < S; ++s) sum += item[s * N + i];
            taken[i] = 1;
            any = true;
          }
          return any;
        };
        std::vector<size_t> all{0, 1, 2};
        const int st = vt_host::run_on_workers_meanwhile(
            workers, post_mu, all,
            [&](size_t s) -> int {
              for (size_t i = 0; i < N; ++i) {
                if (fails && s == 1 && i == N / 2) return 9;
                item[s * N + i] = (uint64_t)(c + 1) * 1000003u + i * (s + 1);
                if (i % 3 != 2) fin[s * N + i].store(1, std::memory_order_release);  // (every third one only at the end)
              }
              for (size_t i = 0; i < N; ++i) fin[s * N + i].store(1, std::memory_order_release);
              return 0;
            },
            take_ready, [](int status, const std::string &) { return status; });
        size_t before = 0;
        for (size_t i = 0; i < N; ++i) before += taken[i] ? 1 : 0;
        early += before;
        if (fails) {
          CHECK(st == 9);
          continue;
        }
        CHECK(st == 0);
        take_ready();
        late += N - before;
        uint64_t want = 0;
        for (size_t s = 0; s < S; ++s)
          for (size_t i = 0; i < N; ++i) want += (uint64_t)(c + 1) * 1000003u + i * (s + 1);
        CHECK(sum == want);
        for (size_t i = 0; i < N; ++i) CHECK(taken[i]);
      }
    });
  for (auto &th : pool) th.join();
  // a `meanwhile` that throws: the exception leaves only after every job has ended (they write into this frame)
  {
    std::vector<uint64_t> cell(S, 0);
    std::vector<size_t> all{0, 1, 2};
    bool caught = false;
    try {
      (void)vt_host::run_on_workers_meanwhile(
          workers, post_mu, all,
          [&](size_t s) -> int {
            for (int i = 0; i < 2000; ++i) cell[s] += (uint64_t)i;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            cell[s] += 1;
            return 0;
          },
          []() -> bool { throw std::runtime_error("meanwhile failed"); }, [](int status, const std::string &) { return status; });
    } catch (const std::runtime_error &) {
      caught = true;
    }
    CHECK(caught);
    for (size_t s = 0; s < S; ++s) CHECK(cell[s] == 1999000u + 1u);
  }
  workers.clear();
  std::fprintf(stderr, "workers (caller merging meanwhile): %llu items taken while the jobs ran, %llu after\n",
               (unsigned long long)early.load(), (unsigned long long)late.load());
}

// ------------------------------------------------------------------ 4. settings beside setenv
static void check_settings(int readers, int per_reader) {
  // what the table holds was decided when this program was loaded (VT_COALESCE_SLOTS=3 from the test's environment)
  CHECK(vt::env::get(vt::env::COALESCE_SLOTS) == 3);
  CHECK(vt::env::get(vt::env::BF16_MIN_RANK) == 6 && vt::env::get(vt::env::FORCE_BATCH_MFMA) == 0);
  CHECK(vt::env::find("force_batch_mfma") == (int)vt::env::FORCE_BATCH_MFMA && vt::env::find("VT_FORCE_BATCH_MFMA") < 0);
  CHECK(vt::env::find("test_refuse_shadow") < 0);  // (a fault hook: not in a build without -DVT_TEST_HOOKS)
  std::atomic<bool> stop{false};
  std::thread meddler([&] {  // System.put_env/2 on a scheduler thread
    for (unsigned i = 0; !stop.load(); ++i) {
      setenv("VT_COALESCE_SLOTS", (i & 1u) ? "7" : "9", 1);
      setenv("VT_BATCH_NO_MFMA", "1", 1);
      unsetenv("VT_BATCH_NO_MFMA");
      setenv("SOME_OTHER_VARIABLE_THAT_GROWS_THE_ENVIRONMENT", std::to_string(i).c_str(), 1);
    }
  });
  std::thread setter([&] {  // vt_debug_set from a test thread
    for (unsigned i = 0; !stop.load(); ++i) vt::env::set(vt::env::BF16_MIN_RANK, 4 + (long)(i & 7u));
  });
  std::atomic<long> sum{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < readers; ++t)
    pool.emplace_back([&] {
      long local = 0;
      for (int i = 0; i < per_reader; ++i) {
        // batch_uses_mfma / multi_scan_applies / sweep_group_applies / coalesce_slots, as the library asks them
        CHECK(!vt::env::on(vt::env::BATCH_NO_MFMA) && !vt::env::on(vt::env::FORCE_BATCH_MFMA));
        CHECK(!vt::env::on(vt::env::NO_MULTI_SCAN) && !vt::env::on(vt::env::NO_GROUP_PIPELINE));
        CHECK(vt::env::get(vt::env::COALESCE_SLOTS) == 3);  // not 7, not 9: the environment is not looked at again
        const long b = vt::env::get(vt::env::BF16_MIN_RANK);
        CHECK(b >= 4 && b <= 11);
        local += b;
      }
      sum += local;
    });
  for (auto &th : pool) th.join();
  stop = true;
  meddler.join();
  setter.join();
  std::fprintf(stderr, "settings: %d readers x %d rounds beside setenv (checksum %ld)\n", readers, per_reader, sum.load());
}

// ------------------------------------------------------------------ 5. a second context
static void check_spare_lease(int threads, int per_thread) {
  FakeShard s;
  std::atomic<int> got{0}, none{0};
  std::vector<std::thread> pool;
  auto make = [](FakeShard *, int *) -> std::unique_ptr<FakeCtx> { return std::make_unique<FakeCtx>(); };
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      std::mt19937 rng(1000 + t);
      for (int i = 0; i < per_thread; ++i) {
        vt_host::LeaseT<FakeShard, FakeCtx> lease(&s, 8, make);
        CHECK(lease.c != nullptr);
        CHECK(lease.c->holders.fetch_add(1) == 0);
        {
          vt_host::SpareLeaseT<FakeShard, FakeCtx> spare((rng() & 1u) ? &s : nullptr, 8, make);
          if (spare.c) {
            CHECK(spare.c != &s.ctx && spare.c != lease.c);
            CHECK(spare.c->holders.fetch_add(1) == 0);  // never held twice
            if ((rng() & 3u) == 0u) spin(3);
            CHECK(spare.c->holders.fetch_sub(1) == 1);
            got += 1;
          } else {
            none += 1;
          }
        }
        CHECK(lease.c->holders.fetch_sub(1) == 1);
      }
    });
  for (auto &th : pool) th.join();  // (returning at all is the point: nobody waits for a second context)
  CHECK(s.extra.size() + 1 <= 8 && !s.ctx0_busy && s.free_ctx.size() == s.extra.size());
  std::fprintf(stderr, "spare lease: %d second contexts handed out, %d times none to be had\n", got.load(), none.load());
}

int main(int argc, char **argv) {
  const int scale = argc > 1 ? std::atoi(argv[1]) : 1;
  check_settings(16, 20000 * scale);
  check_coalescer(64, 1600 * scale);  // 10^5 operations at scale 1
  check_forced_meeting(24, 40 * scale);
  check_lease(48, 4000 * scale);
  check_spare_lease(48, 2000 * scale);
  check_workers(8, 1500 * scale);
  check_workers_meanwhile(6, 400 * scale);
  std::printf("ok\n");
  return 0;
}
