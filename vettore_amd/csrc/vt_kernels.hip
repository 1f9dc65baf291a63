// vt_kernels.hip -- gfx950 kernels around the scan: K3 radix-select merge, K4
// packed-Hamming scan, K5 sign packing, K6 cosine rerank, K7 normalisation and
// the ingest helpers; plus the scan dispatch.  Written for gfx950 only.
#include "vt_scan.cuh"

namespace vt {

using namespace dev;

namespace {

// ---------------------------------------------------------------------------
// K3: top-k of the partial lists, sorted ascending.  Replaces `hits.sort()`
// (flat.rs:120-121, search.rs:107-110) and the cross-wave merge the reference's
// single heap never needed.  One 1024-thread block, MSD radix select on the u64
// keys with 8-bit digits starting at the highest bit in which the keys differ:
//   pass 1  range + count of the live keys,
//   pass 2  histogram of the first digit -> the bin holding the k-th key,
//   pass 3  keys below that bin are winners; keys in it move to an LDS list,
//   then the remaining digits are resolved on the LDS list only.
// Keys are distinct when every row carries its own id rank; rows that share the lazy
// mode's sentinel rank (and callers' duplicate ids) can carry EQUAL keys, so "<= threshold"
// may hold more than k: everything below the threshold is filed first, then its equals, and
// only equals are ever left out.  The winners are rank-sorted in LDS and written, with the
// status word of the scan, straight into the host-mapped result block.
// ---------------------------------------------------------------------------
constexpr uint32_t kSelCand = 4096;  // LDS candidate list capacity

struct SelectBin {
  uint32_t bin, below, count;
};

// Finds the histogram bin containing the krem-th smallest (1-based); wave 0 only.
__device__ __forceinline__ void select_find_bin(const uint32_t *hist, uint32_t krem, int lane, SelectBin *out) {
  const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
  const uint32_t mine = h0 + h1 + h2 + h3;
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
  const uint32_t excl = incl - mine;
  if (excl < krem && krem <= incl) {
    uint32_t below = excl, b = 4 * lane, c = h0;
    if (below + c < krem) {
      below += c; b += 1; c = h1;
      if (below + c < krem) {
        below += c; b += 1; c = h2;
        if (below + c < krem) { below += c; b += 1; c = h3; }
      }
    }
    out->bin = b;
    out->below = below;
    out->count = c;
  }
}

// Visits keys[i] for i = tid, tid + 1024, ... with 8 independent loads in flight
// per thread (one block has to stream up to a few MB out of L2 by itself).
template <typename F>
__device__ __forceinline__ void for_each_key(const uint64_t *__restrict__ keys, uint32_t m, uint32_t tid, F f) {
  constexpr uint32_t kStride = 1024, kUnroll = 8;
  uint32_t i = tid;
  for (; i + (kUnroll - 1) * kStride < m; i += kUnroll * kStride) {
    uint64_t v[kUnroll];
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) v[u] = keys[i + u * kStride];
#pragma unroll
    for (uint32_t u = 0; u < kUnroll; ++u) f(v[u], i + u * kStride);
  }
  for (; i < m; i += kStride) f(keys[i], i);
}

// With more than one block (first level of a two-level select over a long
// list) block b works on its own slice of `slice` keys and leaves its winners,
// unsorted and padded with kEmptyKey, at part_keys/part_pay[b * k ..).
__global__ __launch_bounds__(1024) void select_topk_kernel(const uint64_t *__restrict__ keys,
                                                           const Payload *__restrict__ pay, uint32_t m, uint32_t k,
                                                           uint64_t lo_key, int has_lo, int *dev_status,
                                                           ResultBlock *out, uint32_t slice,
                                                           uint64_t *__restrict__ part_keys,
                                                           Payload *__restrict__ part_pay,
                                                           const uint32_t *__restrict__ m_dev, uint32_t out_stride,
                                                           uint32_t skip_upto = 0) {
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t m_stride = m;
  // list length decided on the device (hamming_collect_kernel): one count per query
  if (m_dev) m = m_dev[blockIdx.y] < m ? m_dev[blockIdx.y] : m;
  if (m_dev && skip_upto && m <= skip_upto) return;  // (select_lists_spread_kernel, launched beside this one, takes those)
  if (gridDim.y > 1) {
    // one list of up to m keys per query (grid.y = queries): query y's winners go to the block
    // `out_stride` bytes after query y - 1's (header + k entries when packed tightly)
    keys += (size_t)blockIdx.y * m_stride;
    pay += (size_t)blockIdx.y * m_stride;
    out = reinterpret_cast<ResultBlock *>(reinterpret_cast<unsigned char *>(out) + (size_t)blockIdx.y * out_stride);
  }
  // part_keys != nullptr: the winners go, unsorted and padded with kEmptyKey, to
  // part_keys/part_pay[blockIdx.x * k ..) instead of a result block -- the first level of
  // a two-level select (several blocks, one slice each) or a device-resident list of up to
  // kSelListMax rows for a following stage (one block).
  const bool partial = part_keys != nullptr;
  if (gridDim.x > 1) {
    const uint32_t lo = blockIdx.x * slice;
    keys += lo;
    pay += lo;
    m = lo >= m ? 0u : (m - lo < slice ? m - lo : slice);
  }
  if (!partial && gridDim.x == 1 && m <= 1024) {
    // A short list -- the few hundred keys a threshold collect leaves, a candidate set being
    // reranked: one key per thread, its place found by counting the smaller ones.  None of
    // the radix passes' fixed cost (9-10 us -> ~3 us for 200 keys).
    uint64_t *sk = reinterpret_cast<uint64_t *>(smem);  // [1024] (the dynamic LDS holds 4096 + k keys)
    __shared__ uint32_t s_live;
    const uint32_t t = threadIdx.x;
    uint64_t key = kEmptyKey;
    if (t < m) {
      key = keys[t];
      if (has_lo && key <= lo_key) key = kEmptyKey;
    }
    sk[t] = key;
    if (t == 0) s_live = 0;
    __syncthreads();
    const bool alive = key != kEmptyKey;
    const uint64_t votes = __ballot(alive);
    if ((t & (kWave - 1)) == 0 && votes) atomicAdd(&s_live, (uint32_t)__popcll(votes));
    if (alive) {
      uint32_t pos = 0;
      for (uint32_t x = 0; x < m; ++x) {
        const uint64_t kx = sk[x];
        pos += (kx < key || (kx == key && x < t)) ? 1u : 0u;
      }
      if (pos < k) {
        const Payload p = pay[t];
        Entry e;
        e.key = key;
        e.row = p.row;
        e.raw = p.raw;
        out->e[pos] = e;
      }
    }
    __syncthreads();
    if (t == 0) {
      out->count = s_live < k ? s_live : k;
      out->status = dev_status ? *dev_status : 0;
      if (dev_status) *dev_status = 0;
    }
    return;
  }
  uint64_t *sel_key = reinterpret_cast<uint64_t *>(smem);  // [k]
  uint64_t *cand_key = sel_key + k;                        // [kSelCand]
  uint32_t *sel_idx = reinterpret_cast<uint32_t *>(cand_key + kSelCand);  // [k]
  uint32_t *cand_idx = sel_idx + k;                                       // [kSelCand]
  __shared__ uint32_t hist[256];
  __shared__ uint64_t red_min[16], red_max[16];
  __shared__ uint32_t red_cnt[16];
  __shared__ SelectBin s_bin;
  __shared__ uint32_t s_sel, s_ncand;
  const uint32_t tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;

  auto live = [&](uint64_t key) { return key != kEmptyKey && (!has_lo || key > lo_key); };

  // pass 1: count of the live keys and the bit positions in which they differ
  // (OR ^ AND): digits are cut from those positions only, so the long constant
  // runs of a key -- an id rank below 2^24 under a 32-bit score, the handful of
  // distinct Hamming distances -- cost no rounds
  uint64_t mn = ~0ull, mx = 0;  // AND, OR
  uint32_t cnt = 0;
  for_each_key(keys, m, tid, [&](uint64_t key, uint32_t) {
    if (live(key)) {
      mn &= key;
      mx |= key;
      cnt += 1;
    }
  });
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mn &= __shfl_xor(mn, o, kWave);
    mx |= __shfl_xor(mx, o, kWave);
    cnt += __shfl_xor(cnt, o, kWave);
  }
  if (lane == 0) {
    red_min[wave] = mn;
    red_max[wave] = mx;
    red_cnt[wave] = cnt;
  }
  if (tid == 0) {
    s_sel = 0;
    s_ncand = 0;
  }
  if (tid < 256) hist[tid] = 0;
  __syncthreads();
  mn = ~0ull;
  mx = 0;
  uint32_t nvalid = 0;
  for (int w = 0; w < 16; ++w) {
    mn &= red_min[w];
    mx |= red_max[w];
    nvalid += red_cnt[w];
  }
  const uint64_t var = nvalid ? (mn ^ mx) : 0ull;  // bits that differ among the live keys

  uint64_t T = ~0ull - 1;  // threshold over the global keys: select every live key <= T
  bool from_cand = false;  // the remaining winners are cand keys <= Tc
  uint64_t Tc = 0;
  if (nvalid > k && var != 0) {
    uint32_t krem = k;
    int hb = 63 - __clzll((long long)var);
    uint64_t mask = ~var;          // constant bits count as resolved
    uint64_t prefix = mx & mask;
    int width = hb + 1 < 8 ? hb + 1 : 8;
    int shift = hb + 1 - width;
    uint32_t dmask = (1u << width) - 1;
    // highest varying bit below `shift`, or -1: the next digit starts there
    auto next_hb = [&](int sh) -> int {
      const uint64_t rem = sh > 0 ? (var & ((1ull << sh) - 1)) : 0ull;
      return rem ? 63 - __clzll((long long)rem) : -1;
    };
    // pass 2: first digit
    for_each_key(keys, m, tid, [&](uint64_t key, uint32_t) {
      if (live(key)) atomicAdd(&hist[(uint32_t)(key >> shift) & dmask], 1u);
    });
    __syncthreads();
    if (wave == 0) select_find_bin(hist, krem, lane, &s_bin);
    __syncthreads();
    SelectBin sb = s_bin;
    krem -= sb.below;
    prefix |= (uint64_t)sb.bin << shift;
    mask |= (uint64_t)dmask << shift;
    if (sb.count == krem || next_hb(shift) < 0) {
      T = prefix | (shift ? ((1ull << shift) - 1) : 0ull);  // the whole bin is selected
    } else if (sb.count <= kSelCand) {
      // pass 3: winners below the bin, the bin itself into LDS
      const uint64_t bin_lo = prefix, bin_hi = prefix | ((1ull << shift) - 1);
      for_each_key(keys, m, tid, [&](uint64_t key, uint32_t i) {
        if (!live(key) || key > bin_hi) return;
        if (key < bin_lo) {
          const uint32_t pos = atomicAdd(&s_sel, 1u);
          if (pos < k) {
            sel_key[pos] = key;
            sel_idx[pos] = i;
          }
        } else {
          const uint32_t pos = atomicAdd(&s_ncand, 1u);
          if (pos < kSelCand) {
            cand_key[pos] = key;
            cand_idx[pos] = i;
          }
        }
      });
      __syncthreads();
      const uint32_t ncand = s_ncand < kSelCand ? s_ncand : kSelCand;
      // remaining digits on the LDS list
      hb = next_hb(shift);
      for (;;) {
        width = hb + 1 < 8 ? hb + 1 : 8;
        shift = hb + 1 - width;
        dmask = (1u << width) - 1;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < ncand; i += 1024) {
          const uint64_t key = cand_key[i];
          if ((key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & dmask], 1u);
        }
        __syncthreads();
        if (wave == 0) select_find_bin(hist, krem, lane, &s_bin);
        __syncthreads();
        sb = s_bin;
        krem -= sb.below;
        prefix |= (uint64_t)sb.bin << shift;
        mask |= (uint64_t)dmask << shift;
        if (sb.count == krem || next_hb(shift) < 0) {
          Tc = prefix | (shift ? ((1ull << shift) - 1) : 0ull);
          break;
        }
        hb = next_hb(shift);
      }
      from_cand = true;
      // (equal keys exist -- see WaveTopK::compact: everything below the threshold is filed
      // before its equals, so that only equals can be left out)
      for (int pass = 0; pass < 2; ++pass) {
        for (uint32_t i = tid; i < ncand; i += 1024) {
          const uint64_t key = cand_key[i];
          if (pass == 0 ? key < Tc : key == Tc) {
            const uint32_t pos = atomicAdd(&s_sel, 1u);
            if (pos < k) {
              sel_key[pos] = key;
              sel_idx[pos] = cand_idx[i];
            }
          }
        }
        __syncthreads();
      }
    } else {
      // crowded bin (more than kSelCand keys share the digit): keep resolving on the global keys
      hb = next_hb(shift);
      for (;;) {
        width = hb + 1 < 8 ? hb + 1 : 8;
        shift = hb + 1 - width;
        dmask = (1u << width) - 1;
        __syncthreads();
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for_each_key(keys, m, tid, [&](uint64_t key, uint32_t) {
          if (live(key) && (key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & dmask], 1u);
        });
        __syncthreads();
        if (wave == 0) select_find_bin(hist, krem, lane, &s_bin);
        __syncthreads();
        sb = s_bin;
        krem -= sb.below;
        prefix |= (uint64_t)sb.bin << shift;
        mask |= (uint64_t)dmask << shift;
        if (sb.count == krem || next_hb(shift) < 0) {
          T = prefix | (shift ? ((1ull << shift) - 1) : 0ull);
          break;
        }
        hb = next_hb(shift);
      }
    }
  }

  if (!from_cand) {
    // compaction of the winners straight from the global keys: below the threshold first, then
    // its equals
    for (int pass = 0; pass < 2; ++pass) {
      for_each_key(keys, m, tid, [&](uint64_t key, uint32_t i) {
        if (live(key) && (pass == 0 ? key < T : key == T)) {
          const uint32_t pos = atomicAdd(&s_sel, 1u);
          if (pos < k) {
            sel_key[pos] = key;
            sel_idx[pos] = i;
          }
        }
      });
      __syncthreads();
    }
  }
  __syncthreads();
  const uint32_t nsel = s_sel < k ? s_sel : k;
  if (partial) {
    for (uint32_t j = tid; j < k; j += 1024) {
      part_keys[(size_t)blockIdx.x * k + j] = j < nsel ? sel_key[j] : kEmptyKey;
      Payload pad;  // padding entries must still be safe to gather from: row 0
      pad.row = 0;
      pad.raw = 0.0f;
      part_pay[(size_t)blockIdx.x * k + j] = j < nsel ? pay[sel_idx[j]] : pad;
    }
    return;
  }
  // rank sort (keys distinct; ties only for caller-supplied duplicate ids)
  for (uint32_t j = tid; j < nsel; j += 1024) {
    const uint64_t kj = sel_key[j];
    uint32_t pos = 0;
    for (uint32_t x = 0; x < nsel; ++x) {
      const uint64_t kx = sel_key[x];
      pos += (kx < kj || (kx == kj && x < j)) ? 1u : 0u;
    }
    const Payload p = pay[sel_idx[j]];
    Entry e;
    e.key = kj;
    e.row = p.row;
    e.raw = p.raw;
    out->e[pos] = e;
  }
  if (tid == 0) {
    out->count = nsel;
    // dev_status == nullptr: an intermediate stage -- the flag stays where it is
    // and reaches the host with the last select of the chain
    out->status = dev_status ? *dev_status : 0;
    if (dev_status) *dev_status = 0;
  }
}

// The k best of each query's list, for lists of up to kSpreadKeys keys, on kSpreadBlocks blocks per list (r05).  A key's
// place among the winners is the number of smaller keys -- computable for every key on its own -- so each block stages the
// whole list in LDS (one sweep of <= 16 KB) and places ITS 128 keys; nothing is exchanged between blocks.  Against the
// one-block forms above: counting on one block is m x m / 64 wave-iterations on ONE CU (60 us at 1 000 keys), the radix
// form four or five DEPENDENT sweeps of the keys in global memory -- 47 us alone, 0.6 ms beside another context's sweep
// of the corpus, which is where a funnel group's list select runs (profiles/r05_funnel64_trace_excerpt.txt).  Longer
// lists return at once: select_topk_kernel is launched beside this kernel with skip_upto = kSpreadKeys.
constexpr uint32_t kSpreadKeys = 2048, kSpreadBlocks = 16, kSpreadThreads = kSpreadKeys / kSpreadBlocks;
__global__ __launch_bounds__(kSpreadThreads) void select_lists_spread_kernel(const uint64_t *__restrict__ keys,
                                                                            const Payload *__restrict__ pay, uint32_t m_stride,
                                                                            const uint32_t *__restrict__ m_dev, uint32_t k,
                                                                            ResultBlock *out, uint32_t out_stride) {
  __shared__ uint64_t sk[kSpreadKeys];
  __shared__ uint32_t s_live;
  const uint32_t y = blockIdx.y;
  const uint32_t m = m_dev[y] < m_stride ? m_dev[y] : m_stride;
  if (m > kSpreadKeys) return;
  keys += (size_t)y * m_stride;
  pay += (size_t)y * m_stride;
  out = reinterpret_cast<ResultBlock *>(reinterpret_cast<unsigned char *>(out) + (size_t)y * out_stride);
  const uint32_t t = threadIdx.x;
  if (t == 0) s_live = 0;
  for (uint32_t i = t; i < m; i += kSpreadThreads) sk[i] = keys[i];
  __syncthreads();
  if (blockIdx.x == 0) {  // the list's header
    uint32_t live = 0;
    for (uint32_t i = t; i < m; i += kSpreadThreads) live += sk[i] != kEmptyKey ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) live += __shfl_xor(live, o, kWave);
    if ((t & (kWave - 1)) == 0) atomicAdd(&s_live, live);
    __syncthreads();
    if (t == 0) {
      out->count = s_live < k ? s_live : k;
      out->status = 0;  // (an intermediate stage: the overflow flag reaches the host with the chain's last select)
    }
  }
  const uint32_t i = blockIdx.x * kSpreadThreads + t;
  const uint64_t key = i < m ? sk[i] : kEmptyKey;
  if (__ballot(key != kEmptyKey) == 0) return;  // (wave-uniform: a wave without keys does not walk the list)
  uint32_t pos = 0;
  for (uint32_t x = 0; x < m; ++x) {
    const uint64_t kx = sk[x];
    pos += (kx < key || (kx == key && x < i)) ? 1u : 0u;
  }
  if (key != kEmptyKey && pos < k) {
    const Payload p = pay[i];
    Entry e;
    e.key = key;
    e.row = p.row;
    e.raw = p.raw;
    out->e[pos] = e;
  }
}

// All of a short list (<= kSelListMax keys) in ascending key order, kEmptyKey entries dropped:
// the winners of a limit above kMaxFusedK leave the device in one launch.  One block, keys in
// LDS, rank sort (m^2 / 1024 compares per thread: ~55 us at m = 4096).
__global__ __launch_bounds__(1024) void sort_list_kernel(const uint64_t *__restrict__ keys, const Payload *__restrict__ pay,
                                                         uint32_t m, int *dev_status, BigResultHeader *head,
                                                         Entry *__restrict__ out) {
  extern __shared__ __align__(16) unsigned char sl_smem[];
  uint64_t *sk = reinterpret_cast<uint64_t *>(sl_smem);
  __shared__ uint32_t s_live;
  if (threadIdx.x == 0) s_live = 0;
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) sk[i] = keys[i];
  __syncthreads();
  uint32_t live = 0;
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
    const uint64_t ki = sk[i];
    if (ki == kEmptyKey) continue;
    live += 1;
    uint32_t pos = 0;
    for (uint32_t x = 0; x < m; ++x) {
      const uint64_t kx = sk[x];
      pos += (kx < ki || (kx == ki && x < i)) ? 1u : 0u;
    }
    const Payload p = pay[i];
    Entry e;
    e.key = ki;
    e.row = p.row;
    e.raw = p.raw;
    out[pos] = e;
  }
  atomicAdd(&s_live, live);
  __syncthreads();
  if (threadIdx.x == 0) {
    head->count = s_live;
    head->status = *dev_status;
    *dev_status = 0;
  }
}

// Cross-shard merge: world * k candidate entries (already sorted per shard) ->
// the k best overall by rank sort in LDS.  Keys are comparable across shards
// because every shard's id_rank column was taken from ONE ordering of all ids.
__global__ __launch_bounds__(256) void merge_blocks_kernel(const unsigned char *__restrict__ blocks, uint32_t world,
                                                           uint32_t k, uint32_t block_bytes, ResultBlock *out,
                                                           uint32_t *__restrict__ out_shard) {
  extern __shared__ __align__(16) unsigned char mb_smem[];
  uint64_t *keys = reinterpret_cast<uint64_t *>(mb_smem);  // [world * k]
  const uint32_t m = world * k;
  __shared__ uint32_t s_total;
  __shared__ int s_status;
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    int status = 0;
    for (uint32_t w = 0; w < world; ++w) {
      const ResultBlock *b = reinterpret_cast<const ResultBlock *>(blocks + (size_t)w * block_bytes);
      total += b->count < k ? b->count : k;
      status = b->status > status ? b->status : status;
    }
    s_total = total;
    s_status = status;
  }
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
    const uint32_t w = i / k, j = i - w * k;
    const ResultBlock *b = reinterpret_cast<const ResultBlock *>(blocks + (size_t)w * block_bytes);
    keys[i] = j < b->count ? b->e[j].key : kEmptyKey;
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
    const uint64_t ki = keys[i];
    if (ki == kEmptyKey) continue;
    uint32_t pos = 0;
    for (uint32_t x = 0; x < m; ++x) {
      const uint64_t kx = keys[x];
      pos += (kx < ki || (kx == ki && x < i)) ? 1u : 0u;
    }
    if (pos < k) {
      const uint32_t w = i / k, j = i - w * k;
      const ResultBlock *b = reinterpret_cast<const ResultBlock *>(blocks + (size_t)w * block_bytes);
      out->e[pos] = b->e[j];
      out_shard[pos] = w;
    }
  }
  if (threadIdx.x == 0) {
    out->count = s_total < k ? s_total : k;
    out->status = s_status;
  }
}

// ---------------------------------------------------------------------------
// K4: packed sign-bit Hamming scan + fused top-k.  Replaces binary_top_k
// (search.rs:76-92) + packed_hamming (distances.rs:426-437, word_mask :472-481).
//
// The bit matrix is the index's own derived structure, so it is stored the way
// a wave wants to read it: per tile of 64 rows, word pair j of all 64 rows is
// contiguous ([tile][pair][row][2] u64).  Lane r then reads row r's words with
// fully coalesced 16-B loads (1 KiB per wave instruction) and owns the row's
// whole popcount: no cross-lane step, no LDS.  The query words are wave-uniform
// (scalar loads, SGPR operands).  PAIRS > 0 unrolls the row completely so every
// load of a tile is in flight before the first popcount.
// ---------------------------------------------------------------------------
template <int CAP, int PAIRS>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void hamming_topk_kernel(const HammingArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t pairs = PAIRS > 0 ? (uint32_t)PAIRS : a.pairs;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kWave - 1) / kWave;
  const uint32_t last_word = a.words - 1;
  const uint32_t rem = a.d % 64;
  const uint64_t last_mask = rem ? ((1ull << rem) - 1) : ~0ull;  // distances.rs:472-481 word_mask
  const u64x2 *bits = reinterpret_cast<const u64x2 *>(a.bits);
  extern __shared__ __align__(16) unsigned char hsmem[];

  WaveTopK<CAP> tk;
  tk.init(hsmem + wib * WaveTopK<CAP>::lds_bytes(), a.k);
  // Unrolled builds: the query's words (the host hands over an even count, the odd one out zero)
  // are read once, through the constant address space (s_load), and stay in SGPRs; only the last
  // word pair can hold the word that needs distances.rs:472-481's mask (words is 2 PAIRS - 1 or
  // 2 PAIRS), and a pad word is zero on both sides.
  typedef const __attribute__((address_space(4))) uint64_t *cu64_p;
  uint64_t qw[PAIRS > 0 ? 2 * PAIRS : 1];
  if (PAIRS > 0) {
#pragma unroll
    for (int j = 0; j < 2 * PAIRS; ++j) qw[j] = ((cu64_p)(uintptr_t)a.qbits)[j];
  }
  const uint64_t mask_even = a.words == 2u * PAIRS - 1 ? last_mask : ~0ull;
  // (an odd word count: the last pair's second word is a zero pad on both sides for whole rows, and the first word
  // BEHIND a prefix -- which must not count -- for a prefix pass)
  const uint64_t mask_odd = a.words == 2u * PAIRS ? last_mask : 0ull;
  const uint32_t tile_pairs = a.tile_pairs ? a.tile_pairs : pairs;
  for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
    const u64x2 *base = bits + ((size_t)t * tile_pairs * kWave + lane);
    // `ham` counts the differing bits; in the pattern mode for jaccard `both` counts the bits set on
    // both sides as well
    uint32_t ham = 0, both = 0;
    if (PAIRS > 0) {
      u64x2 v[PAIRS > 0 ? PAIRS : 1];
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) v[j] = __builtin_nontemporal_load(base + (size_t)j * kWave);
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) {
        const uint64_t q0 = qw[2 * j], q1 = qw[2 * j + 1];
        if (j < PAIRS - 1) {
          ham += __popcll(v[j].x ^ q0) + __popcll(v[j].y ^ q1);
          if (a.jaccard) both += __popcll(v[j].x & q0) + __popcll(v[j].y & q1);
        } else {
          ham += __popcll((v[j].x ^ q0) & mask_even) + __popcll((v[j].y ^ q1) & mask_odd);
          if (a.jaccard) both += __popcll(v[j].x & q0 & mask_even) + __popcll(v[j].y & q1 & mask_odd);
        }
      }
    } else {
      for (uint32_t j = 0; j < pairs; ++j) {
        const u64x2 v = __builtin_nontemporal_load(base + (size_t)j * kWave);
        const uint32_t w0 = 2 * j, w1 = 2 * j + 1;
        const uint64_t q0 = a.qbits[w0], q1 = w1 < a.words ? a.qbits[w1] : 0ull;
        const uint64_t m0 = w0 == last_word ? last_mask : ~0ull, m1 = w1 < a.words ? (w1 == last_word ? last_mask : ~0ull) : 0ull;
        ham += __popcll((v.x ^ q0) & m0) + __popcll((v.y ^ q1) & m1);
        if (a.jaccard) both += __popcll(v.x & q0 & m0) + __popcll(v.y & q1 & m1);
      }
    }
    const uint32_t grow = t * kWave + lane;
    bool valid = grow < a.n;
    const uint32_t my_rank = (valid && a.id_rank) ? a.id_rank[grow] : grow;
    float raw = (float)ham;  // distance as f32 (distances.rs:436; :319-324 over non-zero bits)
    if (a.jaccard) {         // distances.rs:327-347: union = differing + common coordinates
      const uint32_t uni = ham + both;
      raw = uni == 0 ? 0.0f : 1.0f - (float)both / (float)uni;
    }
    const uint64_t key = ((uint64_t)orderable(raw) << 32) | my_rank;
    if (a.has_lo) valid = valid && key > a.lo_key;
    tk.offer(valid, key, grow, raw, lane);
  }
  __shared__ uint32_t s_counts[kWavesPerBlock];
  tk.merge_block(wib, kWavesPerBlock, s_counts, lane);
  if (wib == 0) tk.store(a.part_keys + (size_t)blockIdx.x * a.k, a.part_pay + (size_t)blockIdx.x * a.k, lane);
}

// ---------------------------------------------------------------------------
// K4p: flat_search under float hamming / jaccard for up to kPatternMultiMax queries in ONE sweep
// of the non-zero-bit column (batches; callers that met on a handle).  A lane owns a row as in
// K4 and keeps its words in registers; the queries' words are wave-uniform (scalar loads), so a
// query costs the popcounts and one offer to ITS wave list -- the tile is read once.  Scores and
// keys as in K4's pattern mode (distances.rs:319-347); lists of k <= kSmallK per (query, wave),
// merged per block and filed per query for launch_select_queries.  Padding bits are zero on both
// sides (K5 and the host's query packing write none), so no word needs a mask.
// ---------------------------------------------------------------------------
template <int PAIRS, bool JACCARD>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void pattern_topk_multi_kernel(const PatternMultiArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kWave - 1) / kWave;
  const u64x2 *bits = reinterpret_cast<const u64x2 *>(a.bits);
  extern __shared__ __align__(16) unsigned char hsmem[];
  constexpr size_t kList = WaveTopK<kCapSmall>::lds_bytes();
  uint32_t *s_counts = reinterpret_cast<uint32_t *>(hsmem + (size_t)kPatternMultiMax * kWavesPerBlock * kList);

  WaveTopK<kCapSmall> tk[kPatternMultiMax];
#pragma unroll
  for (int q = 0; q < (int)kPatternMultiMax; ++q) tk[q].init(hsmem + ((size_t)q * kWavesPerBlock + wib) * kList, a.k);
  // 64 KB of lists leave two blocks on a CU -- two waves per SIMD, too few to hide a trip to HBM
  // behind the other wave's popcounts -- so a wave keeps its next U tiles in flight while it works
  // on the current U (registers are what this kernel has to spare).
  // A query's words are scalar loads: their latency is paid once per query and U tiles, not per tile.
  constexpr int U = PAIRS <= 4 ? 4 : PAIRS <= 6 ? 3 : PAIRS <= 8 ? 2 : 1;
  u64x2 cur[U][PAIRS], nxt[U][PAIRS];
  auto load_tiles = [&](u64x2(&dst)[U][PAIRS], uint32_t t0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = t0 + (uint32_t)u * total_waves;
      if (t < ntiles) {
        const u64x2 *base = bits + ((size_t)t * PAIRS * kWave + lane);
#pragma unroll
        for (int j = 0; j < PAIRS; ++j) dst[u][j] = __builtin_nontemporal_load(base + (size_t)j * kWave);
      }
    }
  };
  load_tiles(cur, wave_global);
  for (uint32_t t0 = wave_global; t0 < ntiles; t0 += (uint32_t)U * total_waves) {
    // (the id ranks of the CURRENT tiles are asked for before the next tiles' words: loads return in order, so a rank
    // requested behind the prefetch could only be waited for together with it -- every iteration then drained what it
    // had just put in flight)
    uint32_t grow[U], my_rank[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = t0 + (uint32_t)u * total_waves;
      grow[u] = t * kWave + lane;
      valid[u] = t < ntiles && grow[u] < a.n;
      my_rank[u] = (valid[u] && a.id_rank) ? a.id_rank[grow[u]] : grow[u];
    }
    load_tiles(nxt, t0 + (uint32_t)U * total_waves);
    // jaccard: |x or q| = |x| + |q| - |x and q|, and |x| is the row's own (once per tile, not per
    // query), |q| a scalar: a query costs one v_and + one v_bcnt per 32 row bits, like hamming's xor
    uint32_t px[U];
    if (JACCARD) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        px[u] = 0;
#pragma unroll
        for (int j = 0; j < PAIRS; ++j) px[u] += __popcll(cur[u][j].x) + __popcll(cur[u][j].y);
      }
    }
#pragma unroll
    for (int q = 0; q < (int)kPatternMultiMax; ++q) {
      if ((uint32_t)q < a.nq) {  // (wave-uniform)
        // (read through the constant address space => s_load, and the words enter v_xor / v_and as
        // SGPR operands, as in K4hm: nothing writes them while the kernel runs)
        typedef const __attribute__((address_space(4))) uint64_t *cu64_p;
        const cu64_p qb = (cu64_p)(uintptr_t)a.qbits + (size_t)q * 2 * PAIRS;
        uint64_t qw[2 * PAIRS];
#pragma unroll
        for (int j = 0; j < 2 * PAIRS; ++j) qw[j] = qb[j];
        uint32_t pq = 0;
        if (JACCARD) {
#pragma unroll
          for (int j = 0; j < 2 * PAIRS; ++j) pq += __popcll(qw[j]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (t0 + (uint32_t)u * total_waves < ntiles) {  // (wave-uniform)
            uint32_t ham = 0, both = 0;
#pragma unroll
            for (int j = 0; j < PAIRS; ++j) {
              if (JACCARD) both += __popcll(cur[u][j].x & qw[2 * j]) + __popcll(cur[u][j].y & qw[2 * j + 1]);
              else ham += __popcll(cur[u][j].x ^ qw[2 * j]) + __popcll(cur[u][j].y ^ qw[2 * j + 1]);
            }
            float raw = (float)ham;
            if (JACCARD) {
              const uint32_t uni = px[u] + pq - both;
              raw = uni == 0 ? 0.0f : 1.0f - (float)both / (float)uni;
            }
            const uint64_t key = ((uint64_t)orderable(raw) << 32) | my_rank[u];
            tk[q].offer(valid[u], key, grow[u], raw, lane);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) cur[u][j] = nxt[u][j];
  }
#pragma unroll
  for (int q = 0; q < (int)kPatternMultiMax; ++q) {
    if ((uint32_t)q < a.nq) {
      tk[q].merge_block(wib, kWavesPerBlock, s_counts + q * kWavesPerBlock, lane);
      if (wib == 0) {
        const size_t list = ((size_t)(a.first_query + q) * gridDim.x + blockIdx.x) * a.k;
        tk[q].store(a.part_keys + list, a.part_pay + list, lane);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K4h: the Hamming candidate pass as a pure stream (resident corpus, k <= 256).
// Distances are integers 0..d, so the k-th smallest is found exactly from a
// histogram instead of carrying k-entry lists through the scan:
//   hamming_dist_kernel     popcounts as in K4; writes the 2-byte distance of every
//                           row and accumulates a (d+1)-bin histogram (LDS, flushed
//                           once per block);
//   hamming_collect_kernel  every block finds D* = the k-th smallest distance from
//                           the histogram, then the grid sweeps the distance column
//                           (2 bytes per row) and appends the rows with distance <=
//                           D* -- all winners plus the ties at D* -- to one list,
//                           keyed (distance, id rank); K3 selects the k best.
// The scan neither reads id ranks nor touches candidate buffers, so its time does
// not depend on k (K4: 168 us at k = 10, 249 us at k = 256 for 10M rows).
// Two histograms alternate between queries: the collect pass of one query clears
// the histogram of the next, block 0 of the distance pass clears the list counter.
// ---------------------------------------------------------------------------
template <int PAIRS>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void hamming_dist_kernel(const HammingHistArgs a) {
  extern __shared__ uint32_t hh_lds[];  // [d + 1]
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t pairs = PAIRS > 0 ? (uint32_t)PAIRS : a.pairs;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kWave - 1) / kWave;
  const uint32_t last_word = a.words - 1;
  const uint32_t rem = a.d % 64;
  const uint64_t last_mask = rem ? ((1ull << rem) - 1) : ~0ull;  // distances.rs:472-481 word_mask
  const u64x2 *bits = reinterpret_cast<const u64x2 *>(a.bits);
  for (uint32_t i = threadIdx.x; i <= a.d; i += blockDim.x) hh_lds[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.list_count = 0;
  __syncthreads();
  // (query words in SGPRs, a mask on the last word pair only: as in K4)
  typedef const __attribute__((address_space(4))) uint64_t *cu64_p;
  uint64_t qw[PAIRS > 0 ? 2 * PAIRS : 1];
  if (PAIRS > 0) {
#pragma unroll
    for (int j = 0; j < 2 * PAIRS; ++j) qw[j] = ((cu64_p)(uintptr_t)a.qbits)[j];
  }
  const uint64_t mask_even = a.words == 2u * PAIRS - 1 ? last_mask : ~0ull;
  const uint64_t mask_odd = a.words == 2u * PAIRS ? last_mask : ~0ull;
  for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
    const u64x2 *base = bits + ((size_t)t * pairs * kWave + lane);
    uint32_t ham = 0;
    if (PAIRS > 0) {
      u64x2 v[PAIRS > 0 ? PAIRS : 1];
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) v[j] = __builtin_nontemporal_load(base + (size_t)j * kWave);
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) {
        const uint64_t q0 = qw[2 * j], q1 = qw[2 * j + 1];
        if (j < PAIRS - 1) ham += __popcll(v[j].x ^ q0) + __popcll(v[j].y ^ q1);
        else ham += __popcll((v[j].x ^ q0) & mask_even) + __popcll((v[j].y ^ q1) & mask_odd);
      }
    } else {
      for (uint32_t j = 0; j < pairs; ++j) {
        const u64x2 v = __builtin_nontemporal_load(base + (size_t)j * kWave);
        const uint32_t w0 = 2 * j, w1 = 2 * j + 1;
        const uint64_t q0 = a.qbits[w0], q1 = w1 < a.words ? a.qbits[w1] : 0ull;
        const uint64_t m0 = w0 == last_word ? last_mask : ~0ull, m1 = w1 == last_word ? last_mask : ~0ull;
        ham += __popcll((v.x ^ q0) & m0) + __popcll((v.y ^ q1) & m1);
      }
    }
    const uint32_t grow = t * kWave + lane;
    if (grow < a.n) {
      a.dist[grow] = (uint16_t)ham;
      atomicAdd(&hh_lds[ham], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i <= a.d; i += blockDim.x) {
    const uint32_t c = hh_lds[i];
    if (c) atomicAdd(&a.hist[i], c);
  }
}

// K4h's distance pass for up to kHammingMultiMax (8) queries at once: the bit tiles are read
// ONCE, every lane popcounts its row against all the queries, writes the eight 2-byte distances
// of its row as one 16-byte store (dist[row][8]) and counts them in nq LDS histograms.
// Concurrent quantized_search callers (collection.ex:276-295 under the read lock) then share a
// sweep of the 0.96-GB bit matrix the way plain searches share a scan of the rows.
// The query words are wave-uniform: they come through the scalar cache (constant address space
// => s_load) and enter v_xor as SGPR operands -- with the words in LDS (a ds_read_b64 per word
// and query, then moves) eight queries were VALU-bound at 1.45x the single pass; this is two
// vector instructions per 32 row bits and query.
template <int PAIRS>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void hamming_dist_multi_kernel(const HammingMultiArgs a) {
  extern __shared__ __align__(16) uint32_t hm_lds[];  // [nq][d + 1] histograms
  typedef const __attribute__((address_space(4))) uint64_t *cu64_p;
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t pairs = PAIRS > 0 ? (uint32_t)PAIRS : a.pairs;
  const uint32_t bins = a.d + 1;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kWave - 1) / kWave;
  const uint32_t rem = a.d % 64;
  const uint64_t last_mask = rem ? ((1ull << rem) - 1) : ~0ull;  // distances.rs:472-481 word_mask
  for (uint32_t i = threadIdx.x; i < a.nq * bins; i += blockDim.x) hm_lds[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x < kHammingMultiMax) a.list_count[threadIdx.x] = 0;
  __syncthreads();
  // (the host packs the queries' words with their padding bits clear and an even word count:
  // qbits[q][2 * pairs])
  cu64_p qc = (cu64_p)(uintptr_t)a.qbits;
  const u64x2 *bits = reinterpret_cast<const u64x2 *>(a.bits);
  for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
    const u64x2 *base = bits + ((size_t)t * pairs * kWave + lane);
    const uint32_t grow = t * kWave + lane;
    auto row_words = [&](uint32_t j) -> u64x2 {
      u64x2 v = __builtin_nontemporal_load(base + (size_t)j * kWave);
      // (packed_hamming masks both operands' last word; the matrix's own padding is zero already)
      if (2 * j == a.words - 1) v.x &= last_mask;
      if (2 * j + 1 == a.words - 1) v.y &= last_mask;
      if (2 * j + 1 >= a.words) v.y = 0ull;
      return v;
    };
    // All eight query slots are computed (the host zero-fills the unused ones): no branches in
    // here.  The words of a query are fetched anew for every tile -- hoisted out of the tile loop
    // they are 192 SGPRs, which the compiler then parks in vector lanes (v_writelane / v_readlane
    // around every use: the pass was VALU-bound at 1.45x the single one).
    uint32_t ham[kHammingMultiMax];
    typedef const __attribute__((address_space(4))) uint32_t *cu32_p;
    if (PAIRS > 0) {
      // (no masking of the last word here: the matrix's padding bits and pad word are zero by
      // construction -- sign_pack writes bits j < d only, into zeroed words -- and so are the query's,
      // so they contribute nothing to the xor; the masks were 36 of ~600 vector instructions per tile)
      u64x2 v[PAIRS > 0 ? PAIRS : 1];
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) v[j] = __builtin_nontemporal_load(base + (size_t)j * kWave);
#pragma unroll
      for (uint32_t q = 0; q < kHammingMultiMax; ++q) {
        uint64_t qaddr = (uint64_t)(uintptr_t)a.qbits + (uint64_t)q * 2 * PAIRS * 8;
        asm volatile("" : "+s"(qaddr));  // (not loop-invariant as far as the compiler can tell)
        cu32_p w = (cu32_p)(uintptr_t)qaddr;
        uint32_t h = 0;
#pragma unroll
        for (int j = 0; j < PAIRS; ++j) {
          h = __builtin_popcount((uint32_t)v[j].x ^ w[4 * j]) + h;
          h = __builtin_popcount((uint32_t)(v[j].x >> 32) ^ w[4 * j + 1]) + h;
          h = __builtin_popcount((uint32_t)v[j].y ^ w[4 * j + 2]) + h;
          h = __builtin_popcount((uint32_t)(v[j].y >> 32) ^ w[4 * j + 3]) + h;
        }
        ham[q] = h;
      }
    } else {
#pragma unroll
      for (uint32_t q = 0; q < kHammingMultiMax; ++q) ham[q] = 0;
      for (uint32_t j = 0; j < pairs; ++j) {
        const u64x2 v = row_words(j);
#pragma unroll
        for (uint32_t q = 0; q < kHammingMultiMax; ++q) {
          cu64_p w = qc + (size_t)q * 2 * pairs;
          ham[q] += __popcll(v.x ^ w[2 * j]) + __popcll(v.y ^ w[2 * j + 1]);
        }
      }
    }
    if (grow < a.n) {
      uint32_t packed[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) packed[i] = ham[2 * i] | (ham[2 * i + 1] << 16);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      reinterpret_cast<u32x4 *>(a.dist)[grow] = u32x4{packed[0], packed[1], packed[2], packed[3]};
#pragma unroll
      for (uint32_t q = 0; q < kHammingMultiMax; ++q)
        if (q < a.nq) atomicAdd(&hm_lds[q * bins + ham[q]], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < a.nq * bins; i += blockDim.x) {
    const uint32_t c = hm_lds[i];
    if (c) {
      const uint32_t q = i / bins;
      atomicAdd(&a.hist[(size_t)q * a.hist_stride + (i - q * bins)], c);
    }
  }
}

// The collect pass for the nq queries of a group in ONE sweep of the interleaved distance column
// (16 bytes per row): every block finds the nq thresholds D*_q from the nq histograms, then each
// lane takes a row's eight distances and appends the row to the list of every query it
// qualifies for (keyed (distance, id rank), as hamming_collect_kernel does for one).
__global__ __launch_bounds__(256) void hamming_collect_multi_kernel(const HammingCollectArgs a, uint32_t nq) {
  extern __shared__ uint32_t hcm_lds[];  // [d + 1] one histogram at a time
  __shared__ uint32_t s_dstar[kHammingMultiMax];
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t bins = a.d + 1;
  for (uint32_t q = 0; q < nq; ++q) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) hcm_lds[i] = a.hist[(size_t)q * a.hist_stride + i];
    __syncthreads();
    if (threadIdx.x < kWave) {
      // D* = smallest D with count(distance <= D) >= k; lane l owns bins [l*B, (l+1)*B)
      const uint32_t B = (bins + kWave - 1) / kWave;
      uint32_t mine = 0;
      for (uint32_t j = 0; j < B; ++j) {
        const uint32_t b = lane * B + j;
        mine += b < bins ? hcm_lds[b] : 0u;
      }
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
      }
      const uint32_t excl = incl - mine;
      const uint32_t total = __shfl(incl, kWave - 1, kWave);
      if (lane == 0 && total < a.k) s_dstar[q] = a.d;  // fewer rows than k: everything qualifies
      if (excl < a.k && a.k <= incl) {
        uint32_t cum = excl, b = lane * B;
        for (;; ++b) {
          cum += hcm_lds[b];
          if (cum >= a.k) break;
        }
        s_dstar[q] = b;
      }
    }
  }
  __syncthreads();
  uint32_t dstar[kHammingMultiMax];
#pragma unroll
  for (uint32_t q = 0; q < kHammingMultiMax; ++q) dstar[q] = q < nq ? s_dstar[q] : 0u;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 *d4 = reinterpret_cast<const u32x4 *>(a.dist);
  for (uint32_t row = blockIdx.x * blockDim.x + threadIdx.x; row < a.n; row += gridDim.x * blockDim.x) {
    const u32x4 v = d4[row];
    uint32_t rk = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t q = 0; q < kHammingMultiMax; ++q) {
      const uint32_t dv = (v[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
      if (q < nq && dv <= dstar[q]) {
        if (rk == 0xFFFFFFFFu) rk = a.id_rank ? a.id_rank[row] : row;
        const uint32_t pos = atomicAdd(a.list_count + q, 1u);
        if (pos < a.cap) {
          const float raw = (float)dv;  // distance as f32 (distances.rs:436)
          a.keys[(size_t)q * a.cap + pos] = ((uint64_t)orderable(raw) << 32) | rk;
          Payload p;
          p.row = row;
          p.raw = raw;
          a.pay[(size_t)q * a.cap + pos] = p;
        } else {
          atomicMax(a.status, kStatusRetry);  // more ties than the list holds: the caller takes the queries one by one
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void hamming_collect_kernel(const HammingCollectArgs a0) {
  extern __shared__ uint32_t hc_lds[];  // [d + 1]
  __shared__ uint32_t s_dstar;
  const HammingCollectArgs &a = a0;
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t bins = a.d + 1;
  for (uint32_t i = threadIdx.x; i < bins; i += blockDim.x) hc_lds[i] = a.hist[i];
  // clear the other histogram for the next query (grid-wide, bins are few)
  if (a.hist_next)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < bins; i += gridDim.x * blockDim.x) a.hist_next[i] = 0;
  __syncthreads();
  if (threadIdx.x < kWave) {
    // D* = smallest D with count(distance <= D) >= k; lane l owns bins [l*B, (l+1)*B)
    const uint32_t B = (bins + kWave - 1) / kWave;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < B; ++j) {
      const uint32_t b = lane * B + j;
      mine += b < bins ? hc_lds[b] : 0u;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o, kWave);
      if (lane >= o) incl += t;
    }
    const uint32_t excl = incl - mine;
    const uint32_t total = __shfl(incl, kWave - 1, kWave);
    if (lane == 0 && total < a.k) s_dstar = a.d;  // fewer rows than k: everything qualifies
    if (excl < a.k && a.k <= incl) {
      uint32_t cum = excl, b = lane * B;
      for (;; ++b) {
        cum += hc_lds[b];
        if (cum >= a.k) break;
      }
      s_dstar = b;
    }
  }
  __syncthreads();
  const uint32_t dstar = s_dstar;
  // sweep the distance column, 8 rows (16 bytes) per load
  const uint32_t n8 = (a.n + 7) / 8;
  const u64x2 *d8 = reinterpret_cast<const u64x2 *>(a.dist);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x) {
    const u64x2 v = d8[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t row = i * 8 + j;
      const uint32_t dv = (uint32_t)(((j < 4 ? v.x : v.y) >> (16 * (j & 3))) & 0xFFFFu);
      if (row < a.n && dv <= dstar) {
        const uint32_t pos = atomicAdd(a.list_count, 1u);
        if (pos < a.cap) {
          const float raw = (float)dv;  // distance as f32 (distances.rs:436)
          const uint32_t rk = a.id_rank ? a.id_rank[row] : row;
          a.keys[pos] = ((uint64_t)orderable(raw) << 32) | rk;
          Payload p;
          p.row = row;
          p.raw = raw;
          a.pay[pos] = p;
        } else {
          atomicMax(a.status, kStatusRetry);  // more ties than the list holds: the caller takes the K4 path
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// K5: sign packing (compress_sign_bits, distances.rs:413-423).  One wave per
// 64 coordinates: lane j tests v[j] >= 0.0, the wave ballot IS the word.
// tiled != 0 writes K4's [tile][pair][row][2] layout, else plain [row][word].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sign_pack_kernel(const float *__restrict__ rows, size_t stride, uint32_t n,
                                                        uint32_t d, uint64_t *__restrict__ bits, int tiled, int nonzero) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t W = (d + 63) / 64;
  const uint32_t pairs = (W + 1) / 2;
  const uint64_t total = (uint64_t)n * W;
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  for (uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6); w < total; w += nwaves) {
    const uint32_t r = (uint32_t)(w / W), wi = (uint32_t)(w - (uint64_t)r * W);
    const uint32_t j = wi * 64 + lane;
    bool bit = false;
    if (j < d) {
      const float v = rows[(size_t)r * stride + j];
      bit = nonzero ? v != 0.0f : v >= 0.0f;
    }
    const uint64_t word = __ballot(bit);
    if (lane == 0) {
      const size_t at = tiled ? hamming_word_index(r, wi, pairs) : (size_t)w;
      bits[at] = word;
    }
  }
}

// K5 for the resident corpus (tiled layout, rows on the slab's 256-byte grid): a wave takes
// 256 consecutive floats of a row as one coalesced 1-KiB load (float4 per lane) -- four words of
// the row.  Component c of all 64 lanes is one ballot; word w of the four is the 16 ballot bits of
// lanes 16w..16w+15 of each component, interleaved (bit 4i + c), which lanes 0..3 do with shifts
// and masks.  One pass over the rows at streaming rate instead of a 256-byte load per wave.
__global__ __launch_bounds__(256) void sign_pack_tiled_kernel(const float *__restrict__ rows, size_t stride, uint32_t n,
                                                              uint32_t d, uint64_t *__restrict__ bits, int nonzero) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t W = (d + 63) / 64;
  const uint32_t pairs = (W + 1) / 2;
  const uint32_t segs = (d + 255) / 256;  // per row
  const uint64_t total = (uint64_t)n * segs;
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  constexpr int U = 4;  // segments in flight per wave (one 1-KiB load each)
  for (uint64_t g0 = ((uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6)) * U; g0 < total; g0 += nwaves * U) {
    f32x4 v[U];
    uint32_t r[U], sg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t g = g0 + u;
      r[u] = (uint32_t)(g / segs);
      sg[u] = (uint32_t)(g - (uint64_t)r[u] * segs);
      const uint32_t j0 = sg[u] * 256 + (uint32_t)lane * 4;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g < total && j0 < stride) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(rows + (size_t)r[u] * stride + j0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g0 + u >= total) break;  // (wave-uniform)
      const uint32_t j0 = sg[u] * 256 + (uint32_t)lane * 4;
      const float c[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      uint64_t m[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = __ballot(j0 + k < d && (nonzero ? c[k] != 0.0f : c[k] >= 0.0f));
      if (lane < 4) {
        const uint32_t wi = sg[u] * 4 + (uint32_t)lane;
        if (wi < W) {
          uint64_t word = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint64_t x = (m[k] >> (16 * lane)) & 0xffffull;  // bit i -> bit 4 i
            x = (x | (x << 24)) & 0x000000ff000000ffull;
            x = (x | (x << 12)) & 0x000f000f000f000full;
            x = (x | (x << 6)) & 0x0303030303030303ull;
            x = (x | (x << 3)) & 0x1111111111111111ull;
            word |= x << k;
          }
          bits[hamming_word_index(r[u], wi, pairs)] = word;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float *__restrict__ rows, size_t stride, uint32_t n,
                                                           uint32_t d, int *flag) {
  const uint64_t total = (uint64_t)n * d;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / d, c = i - r * d;
    bad |= !finite_f32(rows[r * stride + c]);
  }
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// K5 for a list of rows (bits of mutated rows patched in place, tiled layout).
__global__ __launch_bounds__(256) void sign_pack_rows_kernel(const float *__restrict__ rows, size_t stride,
                                                             const uint32_t *__restrict__ list, uint32_t count, uint32_t d,
                                                             uint64_t *__restrict__ bits, int nonzero) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t W = (d + 63) / 64;
  const uint32_t pairs = (W + 1) / 2;
  const uint64_t total = (uint64_t)count * W;
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  for (uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6); w < total; w += nwaves) {
    const uint32_t i = (uint32_t)(w / W), wi = (uint32_t)(w - (uint64_t)i * W);
    const uint32_t r = list[i];
    const uint32_t j = wi * 64 + lane;
    bool bit = false;
    if (j < d) {
      const float v = rows[(size_t)r * stride + j];
      bit = nonzero ? v != 0.0f : v >= 0.0f;
    }
    const uint64_t word = __ballot(bit);
    if (lane == 0) bits[hamming_word_index(r, wi, pairs)] = word;
  }
}

// dst[idx[i]] = val[i]: rank updates of a few rows without re-uploading the column.
__global__ __launch_bounds__(256) void scatter_u32_kernel(const uint32_t *__restrict__ pairs, uint32_t n,
                                                          uint32_t *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[pairs[2 * i]] = pairs[2 * i + 1];
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, uint32_t n, uint32_t d,
                                                       float *__restrict__ dst, size_t dst_stride) {
  const uint64_t total = (uint64_t)n * dst_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / dst_stride, c = i - r * dst_stride;
    dst[i] = c < d ? src[r * d + c] : 0.0f;
  }
}

// dst row map[2i + 1] <- src row map[2i] (first d columns; the rest of the dst row zeroed): rows
// of a device-resident batch that land scattered in the slab (upserts; a batch dealt to shards).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, uint32_t d,
                                                          const uint32_t *__restrict__ map, uint32_t count,
                                                          float *__restrict__ dst, size_t dst_stride) {
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    const float *from = src + (size_t)map[2 * i] * d;
    float *to = dst + (size_t)map[2 * i + 1] * dst_stride;
    for (uint32_t c = threadIdx.x; c < dst_stride; c += blockDim.x) to[c] = c < d ? from[c] : 0.0f;
  }
}

// A trickle of rows lands (host/vt_types.h, Shard::Landing): `stage` is a slot of PINNED HOST memory as the device sees it
// -- `count` rows of `ld` floats, already zero padded, then their slab rows (0xFFFFFFFF: an earlier occurrence of an id that
// comes again in the same batch -- skipped, the last one wins, flat.rs:270-281), then `nranks` id ranks for the rows
// rank_first .. of the rank column.  One launch instead of a copy per run of rows and another for the ranks: back to back
// on one stream a small copy costs as much device time as this whole kernel.  One block per row, 16-byte moves.
__global__ __launch_bounds__(256) void land_rows_kernel(const float *__restrict__ stage, uint32_t count, uint32_t ld,
                                                        float *__restrict__ X, uint32_t *__restrict__ rank_col,
                                                        uint32_t rank_first, uint32_t nranks) {
  const uint32_t *targets = reinterpret_cast<const uint32_t *>(stage + (size_t)count * ld);
  const uint32_t *ranks = targets + count;
  if (blockIdx.x == 0 && rank_col)
    for (uint32_t i = threadIdx.x; i < nranks; i += blockDim.x) rank_col[rank_first + i] = ranks[i];
  for (uint32_t j = blockIdx.x; j < count; j += gridDim.x) {
    const uint32_t t = targets[j];
    if (t == 0xFFFFFFFFu) continue;
    const float4 *from = reinterpret_cast<const float4 *>(stage + (size_t)j * ld);
    float4 *to = reinterpret_cast<float4 *>(X + (size_t)t * ld);
    for (uint32_t c = threadIdx.x; c < ld / 4; c += blockDim.x) to[c] = from[c];
  }
}

// flat.rs:88-93 on the slab: the last row moves into the hole (with its rank, when the rank column is current) and its old
// place is zeroed -- rows n..cap are scanned by the last tile and must stay defined.  r == last: only the zeroing.  One
// launch where the host used to queue a copy, a rank copy and a memset.
__global__ __launch_bounds__(256) void swap_delete_kernel(float *__restrict__ X, uint32_t ld, uint32_t r, uint32_t last,
                                                          uint32_t *__restrict__ rank_col) {
  float4 *hole = reinterpret_cast<float4 *>(X + (size_t)r * ld);
  float4 *tail = reinterpret_cast<float4 *>(X + (size_t)last * ld);
  for (uint32_t c = threadIdx.x; c < ld / 4; c += blockDim.x) {
    if (r != last) hole[c] = tail[c];
    tail[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (threadIdx.x == 0 && rank_col && r != last) rank_col[r] = rank_col[last];
}

// K6 (cosine part): exact rerank value of distances.rs:160-177.  One wave per
// candidate: the wave stages the row and the query in LDS with coalesced loads,
// then lanes 0..2 run the three sequential f64 sums |q|^2, |x|^2, q.x in index
// order side by side (distances.rs:179-185 f64_dot is a sequential fold; products
// of two f32 are exact in f64, so only the order of the additions matters).
__global__ __launch_bounds__(64) void cosine_rerank_kernel(const CosineRerankArgs a0) {
  extern __shared__ __align__(16) float crs[];  // [ld] query, [ld] row
  CosineRerankArgs a = a0;
  if (gridDim.y > 1) {  // query y of a batch (launch_cosine_rerank_batch)
    const uint32_t y = blockIdx.y;
    a.q += (size_t)y * a.q_stride;
    if (a.gather) a.gather += (size_t)y * a.gather_qstride;
    a.out_keys += (size_t)y * a.n;
    a.out_pay += (size_t)y * a.n;
  }
  const uint32_t i = blockIdx.x;
  const int lane = threadIdx.x;
  if (a.n_dev && i >= *a.n_dev) {  // (a list shorter than its buffer: nothing behind its end)
    if (lane == 0) a.out_keys[i] = kEmptyKey;
    return;
  }
  const uint32_t src = a.gather ? a.gather[(size_t)i * a.gather_stride] : i;
  const uint32_t ld4 = (a.d + 3) / 4 * 4;
  float *qs = crs, *xs = crs + ld4;
  const float *x = a.X + (size_t)src * a.stride;
  if ((a.stride & 3u) == 0 && a.stride >= ld4) {  // (the query buffer is always padded to padded_dim)
    // every 16-B load of the row and of the query goes out before the first LDS store
    const f32x4 *x4 = reinterpret_cast<const f32x4 *>(x);
    const f32x4 *q4 = reinterpret_cast<const f32x4 *>(a.q);
    const uint32_t n4 = ld4 / 4;
    for (uint32_t base = 0; base < n4; base += 4 * kWave) {
      f32x4 xv[4], qv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = base + u * kWave + lane;
        xv[u] = j < n4 ? x4[j] : f32x4{0, 0, 0, 0};
        qv[u] = j < n4 ? q4[j] : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t j = base + u * kWave + lane;
        if (j < n4) {
          *reinterpret_cast<f32x4 *>(xs + 4 * j) = xv[u];
          *reinterpret_cast<f32x4 *>(qs + 4 * j) = qv[u];
        }
      }
    }
  } else {
    for (uint32_t j = lane; j < a.d; j += kWave) {
      qs[j] = a.q[j];
      xs[j] = x[j];
    }
  }
  wave_lds_fence();
  double acc = 0.0;
  if (lane < 3) {
    const float *A = lane == 1 ? xs : qs;  // lane 0: q.q   lane 1: x.x   lane 2: q.x
    const float *B = lane == 0 ? qs : xs;
    // fma(x, y, acc) == acc + x*y here (the product of two f32 is exact in f64).  The chain is
    // one dependent f64 FMA per element; what it must never wait for is LDS: blocks of 16
    // elements, the NEXT block's eight ds_read_b128 issued before the current block's 16 FMAs
    // (one basic block: a plain `#pragma unroll` left an exit test between the steps and a
    // `s_waitcnt lgkmcnt(0)` in front of every four FMAs -- 30 cycles per element instead of 10).
    uint32_t j = 0;
    const uint32_t nblk = a.d / 16;
    if (nblk) {
      f32x4 a0[4], b0[4], a1[4], b1[4];  // two blocks in registers, filled and consumed in turn
      auto fetch = [&](f32x4(&ra)[4], f32x4(&rb)[4], uint32_t blk) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ra[u] = *reinterpret_cast<const f32x4 *>(A + 16 * blk + 4 * u);
          rb[u] = *reinterpret_cast<const f32x4 *>(B + 16 * blk + 4 * u);
        }
        __builtin_amdgcn_sched_barrier(0);  // the reads stay in front of the chain that hides them
      };
      // (the chain's first step is split off: the compiler waits for EVERY outstanding LDS read
      // before the first use of a block fetched an iteration ago, so the next fetch goes out
      // right after that step, not before it)
      auto head = [&](const f32x4(&ra)[4], const f32x4(&rb)[4]) {
        acc = __builtin_fma((double)ra[0].x, (double)rb[0].x, acc);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto rest = [&](const f32x4(&ra)[4], const f32x4(&rb)[4]) {
        acc = __builtin_fma((double)ra[0].y, (double)rb[0].y, acc);
        acc = __builtin_fma((double)ra[0].z, (double)rb[0].z, acc);
        acc = __builtin_fma((double)ra[0].w, (double)rb[0].w, acc);
#pragma unroll
        for (int u = 1; u < 4; ++u) {
          acc = __builtin_fma((double)ra[u].x, (double)rb[u].x, acc);
          acc = __builtin_fma((double)ra[u].y, (double)rb[u].y, acc);
          acc = __builtin_fma((double)ra[u].z, (double)rb[u].z, acc);
          acc = __builtin_fma((double)ra[u].w, (double)rb[u].w, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      fetch(a0, b0, 0);
      uint32_t blk = 0;
      for (; blk + 2 <= nblk; blk += 2) {
        head(a0, b0);
        fetch(a1, b1, blk + 1);
        rest(a0, b0);
        head(a1, b1);
        fetch(a0, b0, blk + 2 < nblk ? blk + 2 : nblk - 1);  // (no next block: the last one again, unused)
        rest(a1, b1);
      }
      if (blk < nblk) {
        head(a0, b0);
        rest(a0, b0);
      }
      j = nblk * 16;
    }
    for (; j < a.d; ++j) acc = __builtin_fma((double)A[j], (double)B[j], acc);
  }
  const double qq = __shfl(acc, 0, kWave), xx = __shfl(acc, 1, kWave), qx = __shfl(acc, 2, kWave);
  if (lane != 0) return;
  const double ln = sqrt(qq), rn = sqrt(xx);
  float raw = 0.0f;
  bool ok = true;
  if (!(ln == 0.0 || rn == 0.0)) {
    double sim = qx / (ln * rn);
    if (!isfinite(sim)) {
      ok = false;
      atomicMax(a.status, kErrOverflow);
    } else {
      sim = sim < -1.0 ? -1.0 : (sim > 1.0 ? 1.0 : sim);
      raw = (float)sim;
    }
  }
  const uint32_t rk = a.id_rank ? a.id_rank[src] : src;
  a.out_keys[i] = ok ? (((uint64_t)orderable(1.0f - raw) << 32) | rk) : kEmptyKey;
  Payload p;
  p.row = src;
  p.raw = raw;
  a.out_pay[i] = p;
}

// K6b: exact f64 cosine of a prefix of every row + fused top-k.  The reference
// folds |x|^2 and q.x sequentially in f64, element by element, so the K1 trick
// applies one level down: a wave owns 64 rows and walks them in panels of 64
// floats; a panel is read with coalesced 16-B loads (4 rows x 256 B per wave
// instruction), parked in a wave-private LDS panel S[64][68] (stride 4*odd), and
// lane r then runs row r's two f64 chains over it -- 64 chains in parallel.
// (r05: PANEL = 32 -- 8 wave loads of 8 rows x 128 B per panel, half the prefetch registers -- is what K1p, K6bm and this
// kernel run on, DESIGN_APPENDIX A.15; r06: the 64-float builds of r04 have left the library)
constexpr int kCsRows = 64, kCsPanelFloats = 32;

template <int CAP, int PANEL>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void cosine_scan_kernel(const CosineScanArgs a) {
  constexpr int kCsPanel = PANEL, kCsStride = PANEL + 4;
  constexpr int kLanesPerRow = PANEL / 4, kRowsPerLoad = 64 / kLanesPerRow, kLoads = kCsRows / kRowsPerLoad;
  extern __shared__ __align__(16) float cs_lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ldq = padded_dim(a.d);
  float *qs = cs_lds;
  float *S = cs_lds + ldq + wib * (kCsRows * kCsStride);
  unsigned char *tkbuf = reinterpret_cast<unsigned char *>(cs_lds + ldq + kWavesPerBlock * (kCsRows * kCsStride)) +
                         wib * WaveTopK<CAP>::lds_bytes();
  for (uint32_t i = threadIdx.x; i < ldq; i += blockDim.x) qs[i] = a.q[i];
  __syncthreads();

  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kCsRows - 1) / kCsRows;
  const uint32_t npanel = (a.d + kCsPanel - 1) / kCsPanel;
  const double ln = sqrt(a.qq);

  WaveTopK<CAP> tk;
  tk.init(tkbuf, a.k);
  // The panel after the one being summed is already on its way: its 16 loads are
  // issued as soon as the current panel has been parked in LDS, so a wave keeps
  // 16 KiB in flight through its f64 chain phase (without this the kernel sat at
  // 4.3 TB/s of prefix bytes; a plain strided read of the same bytes does 6.5).
  f32x4 v[kLoads];
  const int lrow = lane / kLanesPerRow, lcol = (lane % kLanesPerRow) * 4;  // this lane's row within a load, its column
  auto issue = [&](uint32_t t, uint32_t p) {
#pragma unroll
    for (int s = 0; s < kLoads; ++s) {
      uint32_t r = t * kCsRows + kRowsPerLoad * s + lrow;
      r = r < a.n ? r : a.n - 1;
      v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.X + (size_t)r * a.stride + p * kCsPanel + lcol));
    }
  };
  if (wave_global < ntiles) issue(wave_global, 0);
  for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
    const uint32_t grow = t * kCsRows + lane;
    const bool valid_row = grow < a.n;
    const uint32_t my_rank = (valid_row && a.id_rank) ? a.id_rank[grow] : grow;
    double xx = 0.0, qx = 0.0;
    for (uint32_t p = 0; p < npanel; ++p) {
#pragma unroll
      for (int s = 0; s < kLoads; ++s) *reinterpret_cast<f32x4 *>(S + (kRowsPerLoad * s + lrow) * kCsStride + lcol) = v[s];
      wave_lds_fence();
      if (p + 1 < npanel) issue(t, p + 1);
      else if (t + total_waves < ntiles) issue(t + total_waves, 0);
      const uint32_t cnt = a.d - p * kCsPanel < (uint32_t)kCsPanel ? a.d - p * kCsPanel : (uint32_t)kCsPanel;
      const float *Sr = S + lane * kCsStride;
      const float *qp = qs + p * kCsPanel;
      // fma(x, y, acc) == acc + x*y here: the product of two f32 is exact in f64
      uint32_t j = 0;
      for (; j + 4 <= cnt; j += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(Sr + j);
        const f32x4 qv = *reinterpret_cast<const f32x4 *>(qp + j);
        xx = __builtin_fma((double)xv.x, (double)xv.x, xx);
        qx = __builtin_fma((double)qv.x, (double)xv.x, qx);
        xx = __builtin_fma((double)xv.y, (double)xv.y, xx);
        qx = __builtin_fma((double)qv.y, (double)xv.y, qx);
        xx = __builtin_fma((double)xv.z, (double)xv.z, xx);
        qx = __builtin_fma((double)qv.z, (double)xv.z, qx);
        xx = __builtin_fma((double)xv.w, (double)xv.w, xx);
        qx = __builtin_fma((double)qv.w, (double)xv.w, qx);
      }
      for (; j < cnt; ++j) {
        const double xd = (double)Sr[j];
        xx = __builtin_fma(xd, xd, xx);
        qx = __builtin_fma((double)qp[j], xd, qx);
      }
      wave_lds_fence();
    }
    // distances.rs:160-177
    const double rn = sqrt(xx);
    float raw = 0.0f;
    bool valid = valid_row;
    if (!(ln == 0.0 || rn == 0.0)) {
      double sim = qx / (ln * rn);
      if (!isfinite(sim)) {
        if (valid) atomicMax(a.status, kErrOverflow);
        valid = false;
      } else {
        sim = sim < -1.0 ? -1.0 : (sim > 1.0 ? 1.0 : sim);
        raw = (float)sim;
      }
    }
    const uint64_t key = ((uint64_t)orderable(1.0f - raw) << 32) | my_rank;
    if (a.has_lo) valid = valid && key > a.lo_key;
    if (a.key_out) {
      if (valid_row) a.key_out[grow] = valid ? key : kEmptyKey;
    } else {
      tk.offer(valid, key, grow, raw, lane);
    }
  }
  if (a.key_out) return;
  __shared__ uint32_t s_counts[kWavesPerBlock];
  tk.merge_block(wib, kWavesPerBlock, s_counts, lane);
  if (wib == 0) tk.store(a.part_keys + (size_t)blockIdx.x * a.k, a.part_pay + (size_t)blockIdx.x * a.k, lane);
}

// ---------------------------------------------------------------------------
// Exact k-th smallest of a key column (limits above kMaxFusedK), no host decisions:
// three passes histogram 11-bit digits of the top 33 key bits among the keys that
// match the prefix resolved so far; every block re-derives that prefix from the
// previous passes' histograms (2 048 bins each), so a pass is one launch.  The
// collect pass appends the keys below the final prefix and those sharing it.
// ---------------------------------------------------------------------------
struct RadixPrefix {
  uint64_t prefix, mask;
  uint32_t krem;
};

// Digit q of a key: eleven bits from the top down, the sixth and last one the nine that remain
// (q = 0..2 cover the rank and the top id-rank bit, q = 3..5 the rest of the id rank).
__device__ __forceinline__ int radix_shift(int q) { return q < 5 ? 53 - 11 * q : 0; }
__device__ __forceinline__ uint32_t radix_digit_mask(int q) { return q < 5 ? kRadixBins - 1 : 511u; }

// Bin of `hist` holding the krem-th smallest (1-based), by one wave; updates krem.
__device__ __forceinline__ uint32_t radix_find_bin(const uint32_t *hist, uint32_t *krem, int lane, uint32_t last_bin) {
  constexpr uint32_t B = kRadixBins / kWave;  // bins per lane
  uint32_t mine = 0;
  for (uint32_t j = 0; j < B; ++j) mine += hist[lane * B + j];
  uint32_t incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += t;
  }
  const uint32_t excl = incl - mine;
  const uint32_t k = *krem;
  uint32_t bin = 0xFFFFFFFFu, below = 0;
  if (excl < k && k <= incl) {
    uint32_t cum = excl, b = lane * B;
    for (;; ++b) {
      if (cum + hist[b] >= k) break;
      cum += hist[b];
    }
    bin = b;
    below = cum;
  }
  // exactly one lane found it (or none when fewer than k keys exist: take the last bin)
  const uint64_t m = __ballot(bin != 0xFFFFFFFFu);
  const int src = m ? __ffsll((long long)m) - 1 : 0;
  const uint32_t rbin = __shfl(bin, src, kWave), rbelow = __shfl(below, src, kWave);
  if (!m) return last_bin;
  *krem = k - rbelow;
  return rbin;
}

// Prefix after `passes` resolved digits (wave 0 computes, everyone reads from LDS).
__device__ __forceinline__ RadixPrefix radix_prefix(const RadixArgs &a, int passes, uint32_t *lds_hist, RadixPrefix *s_out) {
  const int lane = threadIdx.x & (kWave - 1);
  RadixPrefix p;
  p.prefix = 0;
  p.mask = 0;
  p.krem = a.k;
  for (int q = 0; q < passes; ++q) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kRadixBins; i += blockDim.x) lds_hist[i] = a.hist[q * kRadixBins + i];
    __syncthreads();
    if (threadIdx.x < kWave) {
      uint32_t krem = p.krem;
      const uint32_t bin = radix_find_bin(lds_hist, &krem, lane, radix_digit_mask(q));
      const int shift = radix_shift(q);
      p.prefix |= (uint64_t)bin << shift;
      p.mask |= (uint64_t)radix_digit_mask(q) << shift;
      p.krem = krem;
      if (threadIdx.x == 0) *s_out = p;
    }
    __syncthreads();
    p = *s_out;
  }
  return p;
}

__global__ __launch_bounds__(256) void radix_pass_kernel(const RadixArgs a, int pass) {
  __shared__ uint32_t lds_hist[kRadixBins];
  __shared__ RadixPrefix s_p;
  if (pass == 0 && blockIdx.x == 0 && threadIdx.x == 0) *a.list_count = 0;
  const RadixPrefix p = radix_prefix(a, pass, lds_hist, &s_p);
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kRadixBins; i += blockDim.x) lds_hist[i] = 0;
  __syncthreads();
  const int shift = radix_shift(pass);
  const uint32_t dmask = radix_digit_mask(pass);
  const u64x2 *k2 = reinterpret_cast<const u64x2 *>(a.keys);
  const uint32_t n2 = a.n / 2;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) {
    const u64x2 v = k2[i];
    if ((v.x & p.mask) == p.prefix) atomicAdd(&lds_hist[(uint32_t)(v.x >> shift) & dmask], 1u);
    if ((v.y & p.mask) == p.prefix) atomicAdd(&lds_hist[(uint32_t)(v.y >> shift) & dmask], 1u);
  }
  if ((a.n & 1u) && blockIdx.x == 0 && threadIdx.x == 0) {
    const uint64_t v = a.keys[a.n - 1];
    if ((v & p.mask) == p.prefix) atomicAdd(&lds_hist[(uint32_t)(v >> shift) & dmask], 1u);
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < kRadixBins; i += blockDim.x) {
    const uint32_t c = lds_hist[i];
    if (c) atomicAdd(&a.hist[pass * kRadixBins + i], c);
  }
}

__global__ __launch_bounds__(256) void radix_collect_kernel(const RadixArgs a) {
  __shared__ uint32_t lds_hist[kRadixBins];
  __shared__ RadixPrefix s_p;
  const RadixPrefix p = radix_prefix(a, a.passes == 6 ? 6 : 3, lds_hist, &s_p);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
    const uint64_t v = a.keys[i];
    if (v != kEmptyKey && (v & p.mask) <= p.prefix) {
      const uint32_t pos = atomicAdd(a.list_count, 1u);
      if (pos < a.cap) {
        a.list_keys[pos] = v;
        Payload pv;
        pv.row = i;
        pv.raw = a.pay_col ? a.pay_col[i].raw : 0.0f;  // (limits above kSelListMax: the list leaves the device as it is)
        a.list_pay[pos] = pv;
      } else {
        atomicMax(a.status, kStatusRetry);
      }
    }
  }
}

// K7: normalize_l2 (distances.rs:350-361), one row per lane.
__global__ __launch_bounds__(64) void normalize_l2_kernel(const float *__restrict__ in, uint32_t n, uint32_t d,
                                                          float *__restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float *x = in + (size_t)r * d;
  float *y = out + (size_t)r * d;
  double acc = 0.0;
  for (uint32_t j = 0; j < d; ++j) {
    const double v = (double)x[j];
    acc += v * v;
  }
  const double norm = sqrt(acc);
  if (norm == 0.0) {
    for (uint32_t j = 0; j < d; ++j) y[j] = 0.0f;
  } else {
    for (uint32_t j = 0; j < d; ++j) y[j] = (float)((double)x[j] / norm);
  }
}

size_t scan_lds_for(const ScanShape &p, uint32_t k) {
  const size_t buf = k <= (uint32_t)kSmallK ? WaveTopK<kCapSmall>::lds_bytes() : WaveTopK<kCapLarge>::lds_bytes();
  return ((size_t)(p.q_global ? 0u : p.ld) + (size_t)kWavesPerBlock * p.tr * p.ss) * sizeof(float) + kWavesPerBlock * buf;
}

// A row too long for the query to share LDS with the panels: the query stays in global memory
// (ScanShape.q_global) and the run-time-op kernel serves the scan.  Returns the LDS bytes.
size_t scan_fit(ScanShape &p, uint32_t k) {
  size_t bytes = scan_lds_for(p, k);
  if (bytes > kMaxLds) {
    p.q_global = 1;
    bytes = scan_lds_for(p, k);
  }
  return bytes;
}

}  // namespace

size_t scan_lds_bytes(uint32_t d, uint32_t k) {
  ScanShape p;
  if (!make_scan_shape(d, 1, &p)) return 0;
  const size_t bytes = scan_fit(p, k);
  return bytes <= kMaxLds ? bytes : 0;
}

hipError_t launch_merge_blocks(const void *blocks, uint32_t world, uint32_t k, uint32_t block_bytes, ResultBlock *out,
                               uint32_t *out_shard, hipStream_t s) {
  if (world == 0 || k == 0 || k > (uint32_t)kMaxFusedK || (size_t)world * k * 8 > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(merge_blocks_kernel, dim3(1), dim3(256), (size_t)world * k * 8, s,
                     static_cast<const unsigned char *>(blocks), world, k, block_bytes, out, out_shard);
  return hipGetLastError();
}

size_t hamming_lds_bytes(uint32_t k) {
  return kWavesPerBlock * (k <= (uint32_t)kSmallK ? WaveTopK<kCapSmall>::lds_bytes() : WaveTopK<kCapLarge>::lds_bytes());
}

uint32_t scan_tile_rows(uint32_t n, uint32_t d, uint32_t resident_waves) {
  uint32_t want = kTileRows;
  if ((uint64_t)n <= (uint64_t)resident_waves * 8 * 8) want = 8;        // <= 8 tiles of 8 rows per wave
  else if ((uint64_t)n <= (uint64_t)resident_waves * 8 * 16) want = 16;
  ScanShape p;
  if (!make_scan_shape(d, n, &p, want)) return kTileRows;
  return p.tr;
}

hipError_t launch_scan(const ScanArgs &a, uint32_t blocks, hipStream_t s) {
  ScanDev sd;
  sd.a = a;
  if (!make_scan_shape(a.d, a.n, &sd.p, a.tile_rows ? a.tile_rows : (uint32_t)kTileRows)) return hipErrorInvalidValue;
  sd.p.tile_floats = sd.p.tr * (uint32_t)a.stride;
  const size_t lds = scan_fit(sd.p, a.k);
  if (lds > kMaxLds || a.k == 0 || a.k > (uint32_t)kMaxFusedK || a.stride < sd.p.ld || a.stride % 4 != 0)
    return hipErrorInvalidValue;
  if (a.gather != nullptr || sd.p.q_global) return launch_scan_general(sd, blocks, 1, lds, s);
  const bool padded = (a.d % kRowAlign) != 0;
  switch (metric_op(a.metric)) {
    case OP_DOT: return launch_scan_dot(sd, blocks, lds, padded, s);
    case OP_L2: return launch_scan_l2(sd, blocks, lds, padded, s);
    default: return launch_scan_misc(sd, blocks, lds, padded, s);
  }
}

hipError_t launch_scan_batch(const ScanArgs &a, uint32_t blocks, uint32_t nq, hipStream_t s) {
  ScanDev sd;
  sd.a = a;
  // candidate lists are short (tens to hundreds of rows per query): 8-row tiles spread a
  // list over more waves (a 32-row tile is ~25 us of one wave's load latency)
  if (!make_scan_shape(a.d, a.batch_cap, &sd.p, 8)) return hipErrorInvalidValue;
  sd.p.tile_floats = sd.p.tr * (uint32_t)a.stride;
  const size_t lds = scan_fit(sd.p, a.k);
  if (lds > kMaxLds || a.k == 0 || a.k > (uint32_t)kMaxFusedK || a.stride < sd.p.ld || !a.gather || !a.batch_counts)
    return hipErrorInvalidValue;
  return launch_scan_general(sd, blocks, nq, lds, s);
}

hipError_t launch_radix_pass(const RadixArgs &a, int pass, uint32_t blocks, hipStream_t s) {
  if (pass < 0 || pass >= (a.passes == 6 ? 6 : 3) || a.n == 0 || a.k == 0 || ((uintptr_t)a.keys & 15)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(radix_pass_kernel, dim3(blocks), dim3(256), 0, s, a, pass);
  return hipGetLastError();
}

hipError_t launch_radix_collect(const RadixArgs &a, uint32_t blocks, hipStream_t s) {
  hipLaunchKernelGGL(radix_collect_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_sort_list(const uint64_t *keys, const Payload *pay, uint32_t m, int *dev_status, BigResultHeader *head,
                            Entry *out, hipStream_t s) {
  if (m == 0 || m > kSelListMax) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sort_list_kernel, dim3(1), dim3(1024), (size_t)m * 8, s, keys, pay, m, dev_status, head, out);
  return hipGetLastError();
}

hipError_t launch_select_list(const uint64_t *keys, const Payload *pay, uint32_t m, const uint32_t *m_dev, uint32_t k,
                              uint64_t *out_keys, Payload *out_pay, hipStream_t s) {
  if (k == 0 || k > kSelListMax || !out_keys || !out_pay) return hipErrorInvalidValue;
  const size_t lds = ((size_t)k + kSelCand) * 12;
  hipError_t e = allow_lds(select_topk_kernel, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(select_topk_kernel, dim3(1), dim3(1024), lds, s, keys, pay, m, k, 0ull, 0, nullptr, nullptr, 0u,
                     out_keys, out_pay, m_dev, 0u);
  return hipGetLastError();
}

hipError_t launch_select_queries(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m, uint32_t k, void *out,
                                 uint32_t out_stride, hipStream_t s) {
  if (k == 0 || k > (uint32_t)kMaxFusedK || nq == 0 || nq > 65535 || out_stride < 16 + k * sizeof(Entry) || out_stride % 16)
    return hipErrorInvalidValue;
  const size_t lds = ((size_t)k + kSelCand) * 12;
  hipError_t e = allow_lds(select_topk_kernel, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(select_topk_kernel, dim3(1, nq), dim3(1024), lds, s, keys, pay, m, k, 0ull, 0, nullptr,
                     static_cast<ResultBlock *>(out), 0u, nullptr, nullptr, nullptr, out_stride);
  return hipGetLastError();
}

hipError_t launch_select(const uint64_t *keys, const Payload *pay, uint32_t m, uint32_t k, uint64_t lo_key, int has_lo,
                         int *dev_status, ResultBlock *out, uint64_t *scratch_keys, Payload *scratch_pay,
                         hipStream_t s, const uint32_t *m_dev) {
  if (k == 0 || k > (uint32_t)kMaxFusedK) return hipErrorInvalidValue;
  const size_t lds = ((size_t)k + kSelCand) * 12;
  if (m >= kSelTwoLevelMin && scratch_keys && scratch_pay && !m_dev) {
    // long lists (k = 100 leaves 51 200 partial keys): kSelGroups blocks select in
    // parallel on slices, one block finishes on kSelGroups * k keys
    const uint32_t slice = (m + kSelGroups - 1) / kSelGroups;
    hipLaunchKernelGGL(select_topk_kernel, dim3(kSelGroups), dim3(1024), lds, s, keys, pay, m, k, lo_key, has_lo,
                       dev_status, out, slice, scratch_keys, scratch_pay, nullptr, 0u);
    hipLaunchKernelGGL(select_topk_kernel, dim3(1), dim3(1024), lds, s, scratch_keys, scratch_pay, kSelGroups * k, k,
                       0ull, 0, dev_status, out, 0u, nullptr, nullptr, nullptr, 0u);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(select_topk_kernel, dim3(1), dim3(1024), lds, s, keys, pay, m, k, lo_key, has_lo, dev_status, out,
                     0u, nullptr, nullptr, m_dev, 0u);
  return hipGetLastError();
}

template <int CAP>
hipError_t launch_hamming_r(const HammingArgs &a, uint32_t blocks, hipStream_t s) {
  const size_t lds = kWavesPerBlock * WaveTopK<CAP>::lds_bytes();
#define VT_HAM_CASE(P)                                                                                        \
  case P:                                                                                                     \
    hipLaunchKernelGGL((hamming_topk_kernel<CAP, P>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a); \
    break;
  switch (a.pairs) {
    VT_HAM_CASE(1)
    VT_HAM_CASE(2)
    VT_HAM_CASE(3)
    VT_HAM_CASE(4)
    VT_HAM_CASE(6)
    VT_HAM_CASE(8)
    VT_HAM_CASE(12)
    VT_HAM_CASE(16)
    default:
      hipLaunchKernelGGL((hamming_topk_kernel<CAP, 0>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  }
#undef VT_HAM_CASE
  return hipGetLastError();
}

hipError_t launch_hamming(const HammingArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.k == 0 || a.k > (uint32_t)kMaxFusedK || a.words == 0 || a.pairs != (a.words + 1) / 2 || (a.tile_pairs && a.tile_pairs < a.pairs))
    return hipErrorInvalidValue;
  return a.k <= (uint32_t)kSmallK ? launch_hamming_r<kCapSmall>(a, blocks, s) : launch_hamming_r<kCapLarge>(a, blocks, s);
}

size_t pattern_multi_lds_bytes() {
  return (size_t)kPatternMultiMax * kWavesPerBlock * WaveTopK<kCapSmall>::lds_bytes() + kPatternMultiMax * kWavesPerBlock * sizeof(uint32_t);
}
bool pattern_multi_supports(uint32_t pairs) {
  return pairs == 1 || pairs == 2 || pairs == 3 || pairs == 4 || pairs == 6 || pairs == 8 || pairs == 12 || pairs == 16;
}
hipError_t launch_pattern_multi(const PatternMultiArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.k == 0 || a.k > (uint32_t)kSmallK || a.nq == 0 || a.nq > kPatternMultiMax || a.pairs != (a.words + 1) / 2 ||
      !pattern_multi_supports(a.pairs))
    return hipErrorInvalidValue;
  const size_t lds = pattern_multi_lds_bytes();
#define VT_PAT_CASE(P)                                                                                                   \
  case P: {                                                                                                              \
    hipError_t e = a.jaccard ? allow_lds(pattern_topk_multi_kernel<P, true>, lds)                                        \
                             : allow_lds(pattern_topk_multi_kernel<P, false>, lds);                                      \
    if (e != hipSuccess) return e;                                                                                       \
    if (a.jaccard)                                                                                                       \
      hipLaunchKernelGGL((pattern_topk_multi_kernel<P, true>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);   \
    else                                                                                                                 \
      hipLaunchKernelGGL((pattern_topk_multi_kernel<P, false>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);  \
    break;                                                                                                               \
  }
  switch (a.pairs) {
    VT_PAT_CASE(1)
    VT_PAT_CASE(2)
    VT_PAT_CASE(3)
    VT_PAT_CASE(4)
    VT_PAT_CASE(6)
    VT_PAT_CASE(8)
    VT_PAT_CASE(12)
    VT_PAT_CASE(16)
    default:
      return hipErrorInvalidValue;
  }
#undef VT_PAT_CASE
  return hipGetLastError();
}

size_t hamming_hist_lds_bytes(uint32_t d) { return ((size_t)d + 1) * sizeof(uint32_t); }

hipError_t launch_hamming_dist(const HammingHistArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.words == 0 || a.pairs != (a.words + 1) / 2 || a.d > kHammingHistMaxDim) return hipErrorInvalidValue;
  const size_t lds = hamming_hist_lds_bytes(a.d);
#define VT_HAMH_CASE(P)                                                                                   \
  case P:                                                                                                 \
    hipLaunchKernelGGL((hamming_dist_kernel<P>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a); \
    break;
  switch (a.pairs) {
    VT_HAMH_CASE(1)
    VT_HAMH_CASE(2)
    VT_HAMH_CASE(3)
    VT_HAMH_CASE(4)
    VT_HAMH_CASE(6)
    VT_HAMH_CASE(8)
    VT_HAMH_CASE(12)
    VT_HAMH_CASE(16)
    default:
      hipLaunchKernelGGL((hamming_dist_kernel<0>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  }
#undef VT_HAMH_CASE
  return hipGetLastError();
}

hipError_t launch_hamming_collect(const HammingCollectArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.d > kHammingHistMaxDim || a.k == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hamming_collect_kernel, dim3(blocks), dim3(256), hamming_hist_lds_bytes(a.d), s, a);
  return hipGetLastError();
}

hipError_t launch_hamming_collect_multi(const HammingCollectArgs &a, uint32_t blocks, uint32_t nq, hipStream_t s) {
  if (a.d > kHammingHistMaxDim || a.k == 0 || nq == 0 || nq > kHammingMultiMax || a.hist_next) return hipErrorInvalidValue;
  hipLaunchKernelGGL(hamming_collect_multi_kernel, dim3(blocks), dim3(256), hamming_hist_lds_bytes(a.d), s, a, nq);
  return hipGetLastError();
}

size_t hamming_multi_lds_bytes(uint32_t d, uint32_t words, uint32_t nq) {
  (void)words;
  return (size_t)nq * (d + 1) * sizeof(uint32_t);
}

hipError_t launch_hamming_dist_multi(const HammingMultiArgs &a, uint32_t blocks, hipStream_t s) {
  if (a.words == 0 || a.pairs != (a.words + 1) / 2 || a.nq == 0 || a.nq > kHammingMultiMax || a.hist_stride < a.d + 1 ||
      ((uintptr_t)a.dist & 15) || ((uintptr_t)a.qbits & 15))
    return hipErrorInvalidValue;
  const size_t lds = hamming_multi_lds_bytes(a.d, a.words, a.nq);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
#define VT_HAMM_CASE(P)                                                                                        \
  case P:                                                                                                      \
    hipLaunchKernelGGL((hamming_dist_multi_kernel<P>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a); \
    break;
  switch (a.pairs) {
    VT_HAMM_CASE(1)
    VT_HAMM_CASE(2)
    VT_HAMM_CASE(3)
    VT_HAMM_CASE(4)
    VT_HAMM_CASE(6)
    VT_HAMM_CASE(8)
    VT_HAMM_CASE(12)
    VT_HAMM_CASE(16)
    default:
      hipLaunchKernelGGL((hamming_dist_multi_kernel<0>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  }
#undef VT_HAMM_CASE
  return hipGetLastError();
}

hipError_t launch_select_lists(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m_stride, const uint32_t *m_dev,
                               uint32_t k, void *out, uint32_t out_stride, hipStream_t s, bool spread) {
  if (k == 0 || k > (uint32_t)kMaxFusedK || nq == 0 || nq > 65535 || !m_dev || out_stride < 16 + k * sizeof(Entry) || out_stride % 16)
    return hipErrorInvalidValue;
  const size_t lds = ((size_t)k + kSelCand) * 12;
  hipError_t e = allow_lds(select_topk_kernel, lds);
  if (e != hipSuccess) return e;
  // `spread` (lists of several hundred to a few thousand keys, a funnel group's): lists of up to kSpreadKeys keys go to
  // select_lists_spread_kernel, sixteen blocks each; the one-block kernel beside it takes the longer ones only
  const uint32_t skip = spread && m_stride > 1024 ? kSpreadKeys : 0u;
  if (skip)
    hipLaunchKernelGGL(select_lists_spread_kernel, dim3(kSpreadBlocks, nq), dim3(kSpreadThreads), 0, s, keys, pay, m_stride, m_dev, k,
                       static_cast<ResultBlock *>(out), out_stride);
  // (grid.y >= 2 is what makes the kernel index its lists by query: a batch of one goes through launch_select)
  hipLaunchKernelGGL(select_topk_kernel, dim3(1, nq), dim3(1024), lds, s, keys, pay, m_stride, k, 0ull, 0, nullptr,
                     static_cast<ResultBlock *>(out), 0u, nullptr, nullptr, m_dev, out_stride, skip);
  return hipGetLastError();
}

hipError_t launch_sign_pack(const float *rows, size_t stride, uint32_t n, uint32_t d, uint64_t *bits, int tiled,
                            hipStream_t s, int nonzero) {
  if (n == 0) return hipSuccess;
  if (tiled && stride % 4 == 0 && ((uintptr_t)rows & 15) == 0) {
    hipLaunchKernelGGL(sign_pack_tiled_kernel, dim3(4096), dim3(256), 0, s, rows, stride, n, d, bits, nonzero);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(sign_pack_kernel, dim3(2048), dim3(256), 0, s, rows, stride, n, d, bits, tiled, nonzero);
  return hipGetLastError();
}

hipError_t launch_check_finite(const float *rows, size_t stride, uint32_t n, uint32_t d, int *flag, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(check_finite_kernel, dim3(2048), dim3(256), 0, s, rows, stride, n, d, flag);
  return hipGetLastError();
}

hipError_t launch_sign_pack_rows(const float *rows, size_t stride, const uint32_t *list, uint32_t count, uint32_t d,
                                 uint64_t *bits, hipStream_t s, int nonzero) {
  if (count == 0) return hipSuccess;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, ((uint64_t)count * ((d + 63) / 64) + 3) / 4);
  hipLaunchKernelGGL(sign_pack_rows_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, rows, stride, list, count, d, bits,
                     nonzero);
  return hipGetLastError();
}

hipError_t launch_scatter_u32(const uint32_t *pairs, uint32_t n, uint32_t *dst, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pairs, n, dst);
  return hipGetLastError();
}

hipError_t launch_pad_rows(const float *src, uint32_t n, uint32_t d, float *dst, size_t dst_stride, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, src, n, d, dst, dst_stride);
  return hipGetLastError();
}

hipError_t launch_gather_rows(const float *src, uint32_t d, const uint32_t *map, uint32_t count, float *dst,
                              size_t dst_stride, hipStream_t s) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(count < 4096 ? count : 4096), dim3(256), 0, s, src, d, map, count, dst,
                     dst_stride);
  return hipGetLastError();
}

hipError_t launch_land_rows(const float *stage_dev, uint32_t count, uint32_t ld, float *X, uint32_t *rank_col, uint32_t rank_first,
                            uint32_t nranks, hipStream_t s) {
  if (count == 0 || ld % 4 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(land_rows_kernel, dim3(count), dim3(256), 0, s, stage_dev, count, ld, X, rank_col, rank_first, nranks);
  return hipGetLastError();
}

hipError_t launch_swap_delete(float *X, uint32_t ld, uint32_t r, uint32_t last, uint32_t *rank_col, hipStream_t s) {
  if (ld % 4 != 0 || r > last) return hipErrorInvalidValue;
  hipLaunchKernelGGL(swap_delete_kernel, dim3(1), dim3(256), 0, s, X, ld, r, last, rank_col);
  return hipGetLastError();
}

hipError_t launch_cosine_rerank(const CosineRerankArgs &a, hipStream_t s) { return launch_cosine_rerank_batch(a, 1, s); }

hipError_t launch_cosine_rerank_batch(const CosineRerankArgs &a, uint32_t nq, hipStream_t s) {
  if (a.n == 0 || nq == 0) return hipSuccess;
  const size_t lds = (size_t)2 * ((a.d + 3) / 4 * 4) * sizeof(float);
  if (lds > kMaxLds) return hipErrorInvalidValue;
  hipError_t e = allow_lds(cosine_rerank_kernel, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cosine_rerank_kernel, dim3(a.n, nq), dim3(64), lds, s, a);
  return hipGetLastError();
}

// K6b, several queries per sweep (CosineScanMultiArgs in vt_device.h).  The row walk is
// cosine_scan_kernel's: a wave parks a 64-row x 64-float panel in LDS, the next panel's 16 loads
// already on their way, then lane r walks row r -- one x.x chain and nq q.x chains, each the
// single kernel's sequence of f64 FMAs.  The queries are wave-uniform: they come as f64 through
// the scalar cache (constant address space => s_load) and enter the FMAs as SGPR operands -- read
// from LDS as f32 like the single kernel's one query, eight queries cost 8 LDS reads and 32
// conversions per 4 row elements and lane beside the 36 FMAs, and the pass was LDS / VALU bound
// at 3.5 TB/s of prefix bytes.
// PANEL (r05): K1p's finding carried over -- the same walk on 64 x 32-float panels (8 wave loads of 8 rows x 128 B, half
// the prefetch registers) at the same two blocks per CU; VT_CS_PANEL=64 is the r04 form (A/B, DESIGN_APPENDIX A.15).
template <int PANEL>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void cosine_scan_multi_kernel(const CosineScanMultiArgs a) {
  constexpr int kCsPanel = PANEL, kCsStride = PANEL + 4;  // (shadow the single kernel's 64 / 68)
  constexpr int kLanesPerRow = PANEL / 4, kRowsPerLoad = 64 / kLanesPerRow, kLoads = kCsRows / kRowsPerLoad;
  extern __shared__ __align__(16) float csm_lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ldq = padded_dim(a.d);
  typedef const __attribute__((address_space(4))) double *cd_p;
  cd_p qd = (cd_p)(uintptr_t)a.Qd;
  float *S = csm_lds + wib * (kCsRows * kCsStride);
  const bool dense = a.sample != nullptr;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles_all = (a.n + kCsRows - 1) / kCsRows;
  const uint32_t step = dense ? a.sample_stride : 1u;           // dense: every step-th tile
  const uint32_t ntiles = (ntiles_all + step - 1) / step;        // tiles this launch walks
  const uint32_t npanel = (a.d + kCsPanel - 1) / kCsPanel;
  // the thresholds, once, through the scalar cache (read per tile as vector loads each brought a wait for the NEXT
  // tile's sixteen panel loads into the epilogue: K1p's note, vt_prefix_multi.hip)
  typedef const __attribute__((address_space(4))) float *cf_p;
  float tauv[kCosineMultiMax];
#pragma unroll
  for (uint32_t q = 0; q < kCosineMultiMax; ++q) tauv[q] = INFINITY;
  if (!dense) {
    cf_p tp = (cf_p)(uintptr_t)a.tau;
#pragma unroll
    for (uint32_t q = 0; q < kCosineMultiMax; ++q)
      if (q < a.nq) tauv[q] = tp[q];
  }
  f32x4 v[kLoads];
  const int lrow = lane / kLanesPerRow, lcol = (lane % kLanesPerRow) * 4;  // this lane's row within a load, its column
  auto issue = [&](uint32_t ti, uint32_t p) {
    const uint32_t t = ti * step;
#pragma unroll
    for (int s = 0; s < kLoads; ++s) {
      uint32_t r = t * kCsRows + kRowsPerLoad * s + lrow;
      r = r < a.n ? r : a.n - 1;
      v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.X + (size_t)r * a.stride + p * kCsPanel + lcol));
    }
  };
  if (wave_global < ntiles) issue(wave_global, 0);
  for (uint32_t ti = wave_global; ti < ntiles; ti += total_waves) {
    const uint32_t grow = ti * step * kCsRows + lane;
    const bool valid_row = grow < a.n;
    double xx = 0.0, qx[kCosineMultiMax];
#pragma unroll
    for (uint32_t q = 0; q < kCosineMultiMax; ++q) qx[q] = 0.0;
    for (uint32_t p = 0; p < npanel; ++p) {
#pragma unroll
      for (int s = 0; s < kLoads; ++s) *reinterpret_cast<f32x4 *>(S + (kRowsPerLoad * s + lrow) * kCsStride + lcol) = v[s];
      wave_lds_fence();
      if (p + 1 < npanel) issue(ti, p + 1);
      else if (ti + total_waves < ntiles) issue(ti + total_waves, 0);
      const uint32_t cnt = a.d - p * kCsPanel < (uint32_t)kCsPanel ? a.d - p * kCsPanel : (uint32_t)kCsPanel;
      const float *Sr = S + lane * kCsStride;
      cd_p qp = qd + p * kCsPanel;
      // fma(x, y, acc) == acc + x*y here: the product of two f32 is exact in f64
      uint32_t j = 0;
      for (; j + 4 <= cnt; j += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(Sr + j);
        const double x0 = (double)xv.x, x1 = (double)xv.y, x2 = (double)xv.z, x3 = (double)xv.w;
        xx = __builtin_fma(x0, x0, xx);
        xx = __builtin_fma(x1, x1, xx);
        xx = __builtin_fma(x2, x2, xx);
        xx = __builtin_fma(x3, x3, xx);
        // (all eight slots, unused ones zero: a branch between the queries puts a wait behind every
        // scalar load; straight-line, the eight loads go out together)
        double w[kCosineMultiMax][4];
#pragma unroll
        for (uint32_t q = 0; q < kCosineMultiMax; ++q) {
          cd_p wp = qp + q * ldq + j;
          w[q][0] = wp[0];
          w[q][1] = wp[1];
          w[q][2] = wp[2];
          w[q][3] = wp[3];
        }
#pragma unroll
        for (uint32_t q = 0; q < kCosineMultiMax; ++q) {
          qx[q] = __builtin_fma(w[q][0], x0, qx[q]);
          qx[q] = __builtin_fma(w[q][1], x1, qx[q]);
          qx[q] = __builtin_fma(w[q][2], x2, qx[q]);
          qx[q] = __builtin_fma(w[q][3], x3, qx[q]);
        }
      }
      for (; j < cnt; ++j) {
        const double xd = (double)Sr[j];
        xx = __builtin_fma(xd, xd, xx);
#pragma unroll
        for (uint32_t q = 0; q < kCosineMultiMax; ++q) qx[q] = __builtin_fma(qp[q * ldq + j], xd, qx[q]);
      }
      wave_lds_fence();
    }
    // distances.rs:160-177, once per query
    const double rn = sqrt(xx);
#pragma unroll
    for (uint32_t q = 0; q < kCosineMultiMax; ++q) {
      if (q >= a.nq) break;
      const double ln = sqrt(a.qq[q]);
      float raw = 0.0f;
      bool valid = valid_row;
      if (!(ln == 0.0 || rn == 0.0)) {
        double sim = qx[q] / (ln * rn);
        if (!isfinite(sim)) {
          if (valid && !dense) atomicMax(a.status, kErrOverflow);
          valid = false;
        } else {
          sim = sim < -1.0 ? -1.0 : (sim > 1.0 ? 1.0 : sim);
          raw = (float)sim;
        }
      }
      if (dense) {
        if (a.sample_maxima) {  // the tile's best score, one value per query and tile
          float m = valid ? raw : -INFINITY;
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
          if (lane == 0 && ti < a.sample_rows) a.sample[(size_t)q * a.sample_rows + ti] = m;
          continue;
        }
        const uint32_t i = ti * kCsRows + lane;  // position in the sample
        if (i < a.sample_rows) a.sample[(size_t)q * a.sample_rows + i] = valid ? raw : -INFINITY;
        continue;
      }
      const bool hit = valid && raw >= tauv[q];
      const uint64_t m = __ballot(hit);
      if (m) {
        uint32_t base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&a.cand_count[q], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(m), kWave);
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (hit && pos < a.cand_cap) {
          const uint32_t my_rank = a.id_rank ? a.id_rank[grow] : grow;  // (only in the rare lanes that list a row)
          a.cand_keys[(size_t)q * a.cand_cap + pos] = ((uint64_t)orderable(1.0f - raw) << 32) | my_rank;
          Payload pv;
          pv.row = grow;
          pv.raw = raw;
          a.cand_pay[(size_t)q * a.cand_cap + pos] = pv;
        }
      }
    }
  }
}

size_t cosine_scan_multi_lds_bytes() { return (size_t)kWavesPerBlock * kCsRows * (kCsPanelFloats + 4) * sizeof(float); }

hipError_t launch_cosine_scan_multi(const CosineScanMultiArgs &a, uint32_t blocks, hipStream_t s) {
  const size_t lds = cosine_scan_multi_lds_bytes();
  if (!a.Qd || ((uintptr_t)a.Qd & 31) || a.nq == 0 || a.nq > kCosineMultiMax || a.n == 0 || a.d == 0) return hipErrorInvalidValue;
  if (a.sample ? (a.sample_stride == 0 || a.sample_rows == 0) : (!a.tau || !a.cand_keys || !a.cand_pay || !a.cand_count))
    return hipErrorInvalidValue;
  hipError_t e = allow_lds(cosine_scan_multi_kernel<kCsPanelFloats>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cosine_scan_multi_kernel<kCsPanelFloats>, dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  return hipGetLastError();
}

size_t cosine_scan_lds_bytes(uint32_t d, uint32_t k) {
  const size_t buf = k <= (uint32_t)kSmallK ? WaveTopK<kCapSmall>::lds_bytes() : WaveTopK<kCapLarge>::lds_bytes();
  const size_t bytes = ((size_t)padded_dim(d) + (size_t)kWavesPerBlock * kCsRows * (kCsPanelFloats + 4)) * sizeof(float) +
                       kWavesPerBlock * buf;
  return bytes <= kMaxLds ? bytes : 0;
}

hipError_t launch_cosine_scan(const CosineScanArgs &a, uint32_t blocks, hipStream_t s) {
  const size_t lds = cosine_scan_lds_bytes(a.d, a.k);
  if (lds == 0 || a.k == 0 || a.k > (uint32_t)kMaxFusedK || a.n == 0) return hipErrorInvalidValue;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
    return hipGetLastError();
  };
  if (a.k <= (uint32_t)kSmallK) return go(cosine_scan_kernel<kCapSmall, kCsPanelFloats>);
  return go(cosine_scan_kernel<kCapLarge, kCsPanelFloats>);
}

// The yardstick bench.py quotes beside the 8 TB/s spec figure (vt_device_read_peak): a read-only
// stream that does strictly less than any search kernel and reads the way the fastest of them do --
// the LDS-DMA ring of the batch passes (vt_batch_shadow.hip) with nobody consuming it.  A block of 8
// waves owns 384-KiB tiles of the buffer; wave w streams two 24-KiB runs of a tile KiB by KiB
// (global_load_lds_dwordx4 nt: 64 lanes x 16 B per piece) into its 2 KiB of each of five LDS stages,
// four chunks = 64 KiB per CU in flight behind a counted vmcnt.  Nothing is computed or written.
// (r03's yardstick -- 16-byte loads to registers strided over the whole grid, on a zero-filled buffer
// -- read SLOWER than the product's scan: a yardstick below the thing it measures is noise)
constexpr uint32_t kPeakStages = 5, kPeakRun = 24 * 1024, kPeakTile = 16 * kPeakRun;
__global__ __launch_bounds__(512, 1) void read_peak_kernel(const char *__restrict__ p, uint32_t ntiles) {
  extern __shared__ __align__(16) unsigned char peak_lds[];
  const uint32_t lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)peak_lds;
  const uint32_t off0 = (2 * wid) * kPeakRun + lane * 16, off1 = off0 + kPeakRun;
  uint32_t stage = 0;
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const char *base = p + (size_t)t * kPeakTile;
    for (uint32_t c = 0; c < kPeakRun / 1024; ++c) {
      stage = __builtin_amdgcn_readfirstlane(stage);
      const uint64_t b = reinterpret_cast<uint64_t>(base + (size_t)c * 1024);
      const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
      const uint32_t m0a = lds0 + stage * 16384 + (2 * wid) * 1024, m0b = m0a + 1024;
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this stage's previous two pieces (five chunks ago) have landed
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" : : "s"(m0a), "v"(off0), "s"(sb) : "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" : : "s"(m0b), "v"(off1), "s"(sb) : "memory");
      stage = stage + 1 == kPeakStages ? 0u : stage + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void peak_fill_kernel(uint32_t *p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u + 12345u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = (h & 0x807fffffu) | 0x3f000000u;  // floats of either sign in [0.5, 1): random mantissas, like rows
  }
}

hipError_t launch_read_peak(const void *buf, size_t bytes, float *sink, uint32_t blocks, hipStream_t s) {
  (void)sink;
  const uint32_t ntiles = (uint32_t)(bytes / kPeakTile);  // (whole tiles only: read_peak_bytes() says how much that is)
  const int lds_bytes = (int)(kPeakStages * 16384);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(read_peak_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(read_peak_kernel, dim3(blocks), dim3(512), lds_bytes, s, reinterpret_cast<const char *>(buf), ntiles);
  return hipGetLastError();
}
size_t read_peak_bytes(size_t bytes) { return bytes / kPeakTile * kPeakTile; }
hipError_t launch_peak_fill(void *buf, size_t bytes, hipStream_t s) {
  hipLaunchKernelGGL(peak_fill_kernel, dim3(4096), dim3(256), 0, s, reinterpret_cast<uint32_t *>(buf), bytes / 4);
  return hipGetLastError();
}

hipError_t launch_normalize_l2(const float *in, uint32_t n, uint32_t d, float *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(normalize_l2_kernel, dim3((n + 63) / 64), dim3(64), 0, s, in, n, d, out);
  return hipGetLastError();
}

}  // namespace vt
