// vt_concurrency.h -- the host side's three pieces of thread machinery, stand-alone: no HIP, no
// other header of this directory, every operation they wrap supplied by the instantiating side.
//   * WorkerT      one thread per shard of a multi-shard handle, a FIFO of jobs;
//   * LeaseT       a context for one reader of a shard (the primary one, a spare, a new one, or wait);
//   * coalesced_search_t / run_coalesced_t   searches that meet on one handle travel as one batch.
// vt_types.h and vt_coalesce.h instantiate them with the real operations (HIP contexts, the search
// and batch paths); tests/concurrency_check.cpp instantiates the same templates with stubs and
// runs them under ThreadSanitizer on a CPU box (tests/test_concurrency.py) -- two races in this
// code were found in round 2 by a test that failed one run in three; this is the check that does
// not need luck.
#pragma once

#include <algorithm>
#include <chrono>
#include <exception>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct vt_hits;

namespace vt_host {

// ---------------------------------------------------------------------------- worker
// One thread bound to a shard's device: the caller posts the same job to all the workers of a
// handle, so the launch overheads of the shards overlap and each shard's kernels are issued by a
// thread whose current device never changes.
//   Policy::thread_start(device)                       once, on the new thread
//   Policy::run(fn, &error) -> status                  runs a job; no exception may leave it
template <class Policy>
struct WorkerT {
  struct Job {
    std::function<int()> fn;
    int status = 0;
    std::string error;
    bool done = false;
  };
  std::thread th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::deque<Job *> queue;
  bool stop = false;
  int device = 0;

  void start(int dev) {
    device = dev;
    th = std::thread([this] { loop(); });
  }
  void loop() {
    Policy::thread_start(device);
    for (;;) {
      Job *job = nullptr;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [this] { return stop || !queue.empty(); });
        if (queue.empty()) return;  // stop
        job = queue.front();
        queue.pop_front();
      }
      std::string error;
      const int st = Policy::run(job->fn, &error);
      {
        std::lock_guard<std::mutex> g(mu);
        job->status = st;
        if (st != 0) job->error = std::move(error);
        job->done = true;
      }
      done_cv.notify_all();
    }
  }
  void post(Job *job) {
    {
      std::lock_guard<std::mutex> g(mu);
      queue.push_back(job);
    }
    cv.notify_one();
  }
  void wait(Job *job) {
    std::unique_lock<std::mutex> g(mu);
    done_cv.wait(g, [job] { return job->done; });
  }
  bool finished(Job *job) {
    std::lock_guard<std::mutex> g(mu);
    return job->done;
  }
  ~WorkerT() {
    {
      std::lock_guard<std::mutex> g(mu);
      stop = true;
    }
    cv.notify_one();
    if (th.joinable()) th.join();
  }
};

// The same job on several workers: posted under ONE mutex so that every worker sees the jobs of
// concurrent callers in the same order (collectives must match up), then waited for.  Returns the
// first failing job's status and hands its message to `on_error`.
template <class Worker, class F, class OnError>
int run_on_workers(std::vector<std::unique_ptr<Worker>> &workers, std::mutex &post_mu, const std::vector<size_t> &which, F fn,
                   OnError on_error) {
  std::vector<typename Worker::Job> jobs(which.size());
  for (size_t i = 0; i < which.size(); ++i) {
    const size_t s = which[i];
    jobs[i].fn = [&fn, s]() -> int { return fn(s); };
  }
  {
    std::lock_guard<std::mutex> g(post_mu);
    for (size_t i = 0; i < which.size(); ++i) workers[which[i]]->post(&jobs[i]);
  }
  for (size_t i = 0; i < which.size(); ++i) workers[which[i]]->wait(&jobs[i]);
  for (size_t i = 0; i < which.size(); ++i)
    if (jobs[i].status != 0) return on_error(jobs[i].status, jobs[i].error);
  return 0;
}

// The same, with the posting thread put to use while the workers run: `meanwhile()` is called over and over until
// every job is done (it returns whether it found something to do; when it did not, the thread naps for 100 us -- the
// jobs here are passes over a corpus, milliseconds long).  What `meanwhile` may touch of the jobs' results is the
// callers' business (batch_multi: hit lists published query by query through release stores).
template <class Worker, class F, class M, class OnError>
int run_on_workers_meanwhile(std::vector<std::unique_ptr<Worker>> &workers, std::mutex &post_mu, const std::vector<size_t> &which,
                             F fn, M meanwhile, OnError on_error) {
  std::vector<typename Worker::Job> jobs(which.size());
  for (size_t i = 0; i < which.size(); ++i) {
    const size_t s = which[i];
    jobs[i].fn = [&fn, s]() -> int { return fn(s); };
  }
  {
    std::lock_guard<std::mutex> g(post_mu);
    for (size_t i = 0; i < which.size(); ++i) workers[which[i]]->post(&jobs[i]);
  }
  // (the jobs point into this frame -- `jobs`, `fn` and whatever `fn` captured: nothing may unwind out of here before
  // the last of them has ended.  An exception of `meanwhile` is kept and rethrown then.)
  std::exception_ptr thrown;
  for (size_t next = 0; next < which.size();) {
    if (workers[which[next]]->finished(&jobs[next])) {
      ++next;
      continue;
    }
    bool worked = false;
    if (!thrown) {
      try {
        worked = meanwhile();
      } catch (...) {
        thrown = std::current_exception();
      }
    }
    if (!worked) std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  if (thrown) std::rethrow_exception(thrown);
  for (size_t i = 0; i < which.size(); ++i)
    if (jobs[i].status != 0) return on_error(jobs[i].status, jobs[i].error);
  return 0;
}

// ---------------------------------------------------------------------------- lease
// A context for one reader: the primary one if free, else a spare, else a new one (up to
// `max_contexts`), else wait.  `Holder` carries: pool_mu, pool_cv, ctx (the primary context),
// ctx0_busy, free_ctx (vector<Ctx *>), extra (vector<unique_ptr<Ctx>>).
//   make(holder, &status) -> unique_ptr<Ctx>           a new, initialised context (or null + status)
template <class Holder, class Ctx>
struct LeaseT {
  Holder *ix;
  Ctx *c = nullptr;
  int status = 0;
  template <class Make>
  LeaseT(Holder *s, size_t max_contexts, Make make) : ix(s) {
    std::unique_lock<std::mutex> g(ix->pool_mu);
    for (;;) {
      if (!ix->ctx0_busy) {
        ix->ctx0_busy = true;
        c = &ix->ctx;
        return;
      }
      if (!ix->free_ctx.empty()) {
        c = ix->free_ctx.back();
        ix->free_ctx.pop_back();
        return;
      }
      if (ix->extra.size() + 1 < max_contexts) {
        std::unique_ptr<Ctx> nc = make(ix, &status);
        if (!nc) return;
        c = nc.get();
        ix->extra.push_back(std::move(nc));
        return;
      }
      ix->pool_cv.wait(g);
    }
  }
  ~LeaseT() {
    if (!c) return;
    {
      std::lock_guard<std::mutex> g(ix->pool_mu);
      if (c == &ix->ctx) ix->ctx0_busy = false;
      else ix->free_ctx.push_back(c);
    }
    ix->pool_cv.notify_one();
  }
  LeaseT(const LeaseT &) = delete;
  LeaseT &operator=(const LeaseT &) = delete;
};

// A SECOND context for a reader that already holds one (consecutive groups of one batch call alternate
// between two: DESIGN 4.5): a spare or a new one -- never the primary (the holder may be using it without a
// lease, under the exclusive lock or on a shard's worker) and never a wait (two readers each holding one
// context and waiting for a second would wait for ever): `c` stays null when there is none to be had, and
// the caller goes on with the one context it has.
template <class Holder, class Ctx>
struct SpareLeaseT {
  Holder *ix;
  Ctx *c = nullptr;
  int status = 0;
  template <class Make>
  SpareLeaseT(Holder *s, size_t max_contexts, Make make) : ix(s) {  // (s == nullptr: none wanted)
    if (!ix) return;
    std::unique_lock<std::mutex> g(ix->pool_mu);
    if (!ix->free_ctx.empty()) {
      c = ix->free_ctx.back();
      ix->free_ctx.pop_back();
    } else if (ix->extra.size() + 1 < max_contexts) {
      std::unique_ptr<Ctx> nc = make(ix, &status);
      if (!nc) return;
      c = nc.get();
      ix->extra.push_back(std::move(nc));
    }
  }
  ~SpareLeaseT() {
    if (!c) return;
    {
      std::lock_guard<std::mutex> g(ix->pool_mu);
      ix->free_ctx.push_back(c);
    }
    ix->pool_cv.notify_one();
  }
  SpareLeaseT(const SpareLeaseT &) = delete;
  SpareLeaseT &operator=(const SpareLeaseT &) = delete;
};

// ---------------------------------------------------------------------------- coalescer
// Searches that arrive while another one is running wait for it and then go TOGETHER: one sweep
// of the corpus answers up to eight of them (K1m), a matrix-core pass up to 256 (K2b), where the
// same callers on their own streams would read the whole corpus once each.
struct Waiting {
  const float *query;
  size_t n, limit;
  int kind;    // which entry point (0: flat_search; the instantiating side names the others)
  size_t aux;  // its extra parameter (quantized_search: candidates); only equals travel together
  vt_hits **out;
  int status = 0;
  std::string error;
  enum { QUEUED, LEADS, ALONE, DONE } state = QUEUED;
  std::condition_variable wake;  // its own: a finished batch wakes exactly its members and the next leader
  Waiting(const float *q, size_t n_, size_t limit_, int kind_, size_t aux_, vt_hits **out_)
      : query(q), n(n_), limit(limit_), kind(kind_), aux(aux_), out(out_) {}
};
struct Coalescer {
  std::mutex mu;
  std::condition_variable gather;  // a leader waiting a moment for the callers it expects back
  std::deque<Waiting *> waiting;
  unsigned active = 0;        // searches / batches running
  size_t last_batch = 1;      // members of the last batch that ran
  double last_seconds = 0.0;  // what it took
  std::chrono::steady_clock::time_point last_end{};  // when it ended
  uint64_t batches = 0, batched_queries = 0;
};

constexpr size_t kCoalesceMax = 256;

// Runs the members of one batch (all with the leader's limit and query length).
//   Ops::search_alone(h, w)                            w's own search; fills w->status / w->error
//   Ops::judge(h, members, &good)                      every query judged on its own (flat.rs:97-101):
//                                                      the bad ones get their status, the others go to `good`
//   Ops::batch(h, kind, aux, queries, nq, n, limit, outs) -> st   one batch; outs[i] = query i's hits
template <class H, class Ops>
void run_coalesced_t(H *h, std::vector<Waiting *> &members) {
  const size_t limit = members[0]->limit, n = members[0]->n;
  if (members.size() == 1) {
    Ops::search_alone(h, members[0]);
    return;
  }
  std::vector<Waiting *> good;
  Ops::judge(h, members, &good);
  if (good.size() < 2) {
    for (Waiting *w : good) Ops::search_alone(h, w);
    return;
  }
  std::vector<float> qs(good.size() * n);
  for (size_t i = 0; i < good.size(); ++i) std::memcpy(&qs[i * n], good[i]->query, n * sizeof(float));
  std::vector<vt_hits *> outs(good.size(), nullptr);
  const int st = Ops::batch(h, members[0]->kind, members[0]->aux, qs.data(), good.size(), n, limit, outs.data());
  if (st == 0) {
    for (size_t i = 0; i < good.size(); ++i) *good[i]->out = outs[i];
    return;
  }
  // one query's failure ("metric overflow", a dimension that changed under us) is that query's own
  for (Waiting *w : good) Ops::search_alone(h, w);
}

//   Ops::coalescer(h) -> Coalescer &
//   Ops::slots(h) -> unsigned                          operations in flight before callers queue
//   Ops::search_direct(h, kind, aux, query, n, limit, out) -> st  a search outside the coalescer
//   Ops::must_disband(h, kind, limit) -> bool          a batch would force work a lone search avoids
//   Ops::capacity(h, kind) -> size_t                   callers one pass over the corpus carries at no extra cost
//   Ops::hold_until(h) -> size_t                       0; test builds: callers an idle handle's first caller waits for
//   Ops::run(h, members)                               normally run_coalesced_t<H, Ops>
//   Ops::drop_hits(hits)                               frees a hit list
//   Ops::set_last_error(msg)                           the calling thread's error text
//   Ops::kOutOfMemory                                  status of a batch that threw
template <class H, class Ops>
int coalesced_search_t(H *h, const float *query, size_t n, size_t limit, vt_hits **out, int kind = 0, size_t aux = 0) {
  Coalescer &co = Ops::coalescer(h);
  const unsigned max_active = Ops::slots(h);
  *out = nullptr;  // ("answered" is read off this pointer if a batch dies half way)
  Waiting me(query, n, limit, kind, aux, out);
  std::vector<Waiting *> members;
  members.reserve(kCoalesceMax);  // (no allocation once others depend on this caller)
  // Callers that have just been answered are about to come back: whoever starts the next pass
  // gives them a moment (a few % of a pass) before it runs without them.  `already`: callers the
  // starter can see now; it waits until as many more as the last batch had have queued up, or the
  // window closes.  (wait_until on the system clock = pthread_cond_timedwait, which every
  // ThreadSanitizer intercepts; wait_for's pthread_cond_clockwait is invisible to gcc 11's, which
  // then believes the mutex is still held.  A clock step during these <= 300 us only ends the wait.)
  // Only while the pass can still take them for free (Ops::capacity: callers one pass carries at no
  // extra cost -- 256 for a matrix-core batch, 8 where groups of eight share a sweep): beyond that
  // a batch takes as many passes as two smaller ones, and waiting buys nothing.
  auto gather_returning = [&](std::unique_lock<std::mutex> &lk) {
    const size_t capacity = std::min(kCoalesceMax, Ops::capacity(h, kind));
    if (co.waiting.size() + 1 >= capacity) return;
    // (a thread needs 20-50 us to wake up and call again: where one pass owns the card -- corpora of a
    // GiB and more, passes of 0.2 ms and more -- the window does not go below 30 us)
    double window = std::min(300e-6, 0.03 * co.last_seconds);
    if (max_active == 1) window = std::max(window, 30e-6);
    const size_t want = std::min(capacity - 1, co.waiting.size() + co.last_batch - 1);
    co.gather.wait_until(lk, std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::system_clock::duration>(
                                                                   std::chrono::duration<double>(window)),
                         [&] { return co.waiting.size() >= want; });
  };
  auto take_along = [&]() {
    for (auto it = co.waiting.begin(); it != co.waiting.end() && members.size() + 1 < kCoalesceMax;) {
      if ((*it)->limit == limit && (*it)->n == n && (*it)->kind == kind && (*it)->aux == aux) {
        members.push_back(*it);
        it = co.waiting.erase(it);
      } else {
        ++it;
      }
    }
  };
  {
    std::unique_lock<std::mutex> lk(co.mu);
    if (co.active < max_active && co.waiting.empty()) {
      co.active += 1;  // nobody to wait for ...
      // (builds with the test hooks only -- Ops::hold_until is the constant 0 in the product: the caller that finds the
      // handle idle keeps its slot until `hold` callers are here or five seconds have passed, so that a test's callers
      // MEET instead of hoping to; what happens once they have met is the code below, unchanged)
      if (const size_t hold = Ops::hold_until(h)) {
        co.gather.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds(5),
                             [&] { return co.waiting.size() + 1 >= hold; });
        take_along();
      } else
      // ... unless a batch has only just ended: its callers are on their way back, and the first
      // of them to arrive would otherwise run alone, the others queue behind it, and the handle
      // settles into passes of 1 and N - 1 callers (or two alternating halves) -- N callers per
      // TWO passes.  (Only when this caller is the one operation in flight: with more slots the
      // corpus is small and passes overlap anyway.)
      if (max_active == 1 && co.last_batch > 1 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - co.last_end).count() <
              std::max(30e-6, std::min(300e-6, 0.03 * co.last_seconds))) {
        gather_returning(lk);
        take_along();
      }
    } else {
      co.waiting.push_back(&me);
      co.gather.notify_one();
      me.wake.wait(lk, [&] { return me.state != Waiting::QUEUED; });
      if (me.state == Waiting::DONE) {
        if (me.status != 0) Ops::set_last_error(me.error);
        return me.status;
      }
      if (me.state == Waiting::ALONE) {
        lk.unlock();
        return Ops::search_direct(h, kind, aux, query, n, limit, out);
      }
      // LEADS (the operation that just finished passed its slot on: `active` already counts this one).
      // The callers of the batch that just ended are about to come back: wait a moment for them too
      // (through r03 the leader only waited while FEWER callers than the last batch had were queued:
      // two halves of the callers then took turns for good, each pass carrying half of them)
      if (co.last_batch > 1) gather_returning(lk);
      take_along();
    }
    members.insert(members.begin(), &me);
  }
  // a batch needs strictly current id ranks; a lone search after unsorted inserts does not
  // (lazy ranks): then nobody is made to wait for a re-rank -- everyone searches alone
  const bool disband = members.size() > 1 && Ops::must_disband(h, kind, limit);
  const auto t0 = std::chrono::steady_clock::now();
  if (disband) {
    {
      std::lock_guard<std::mutex> g(co.mu);
      for (size_t i = 1; i < members.size(); ++i) {
        members[i]->state = Waiting::ALONE;
        members[i]->wake.notify_one();
      }
    }
    members.resize(1);
  }
  try {
    Ops::run(h, members);
  } catch (...) {  // (host memory, most likely) -- nobody may be left waiting
    // whoever has been answered -- with hits or with an error of his own -- keeps that answer
    for (Waiting *w : members) {
      if (w->status != 0 || (w->out && *w->out)) continue;
      w->status = Ops::kOutOfMemory;
      w->error = "out of host memory";
    }
  }
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> g(co.mu);
    // (notified under the lock: a member may return -- and its Waiting leave the stack -- the
    // moment it can take the lock and see DONE)
    for (size_t i = 1; i < members.size(); ++i) {
      members[i]->state = Waiting::DONE;
      members[i]->wake.notify_one();
    }
    co.last_batch = members.size();
    co.last_seconds = seconds;
    co.last_end = std::chrono::steady_clock::now();
    if (members.size() > 1) {
      co.batches += 1;
      co.batched_queries += members.size();
    }
    // the longest-waiting caller leads next, in this operation's slot (it takes the others along)
    if (!co.waiting.empty()) {
      co.waiting.front()->state = Waiting::LEADS;
      co.waiting.front()->wake.notify_one();
      co.waiting.pop_front();
    } else {
      co.active -= 1;
    }
  }
  if (me.status != 0) Ops::set_last_error(me.error);
  return me.status;
}

}  // namespace vt_host
