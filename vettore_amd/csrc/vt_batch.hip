// vt_batch.hip -- K2: query batches on the FP32 matrix cores (gfx950).
//
// The reference answers B queries with B independent scans (flat.rs:96-124).
// Here S = X * Q^T (N x B) is a genuine dense GEMM, so it runs on
// v_mfma_f32_32x32x2_f32.  MFMA sums each dot product as one fmaf chain, which
// is NOT the reference's chunked order, so the MFMA pass only nominates
// candidates; the exact K1 arithmetic then re-scores them (vt_scan.cuh, batch
// mode), and the host accepts a query's result only if a rigorous bound proves
// that no row outside the candidate set can reach the top k (vt_index.cpp,
// vt_flat_search_batch).  Ranking and scores therefore stay bit-identical to
// the reference; S itself is never materialised.
//
//   pass 0  scores of a strided sample of row tiles, dense, -> per-query
//           threshold tau_b = 3rd best sample score (sample_tau_kernel);
//   pass 1  every row tile: S tile in accumulators, epilogue appends
//           (score, row) with score >= tau_b to the query's candidate list.
//
// Work split: a wave owns 32 rows x (NT*32) queries (NT*16 accumulator VGPRs).
// Both operands of a 32-wide k chunk go through LDS (a ring of three stages), filled by
// LDS-DMA (global_load_lds, 16 B/lane, whole 128-B lines, no staging
// registers): the Q chunk (all queries) is shared by the block, each wave's 32
// X rows are private to it.  The images are linear [row][8 slots of 16 B] and
// the slot index is XOR-swizzled with (row >> 1) & 7 -- applied to the per-lane
// SOURCE address on the way in and to the read address on the way out -- which
// makes every ds_read_b128 conflict-free.  (Fragment-shaped X loads straight to
// registers were measured 6-20 % slower: 32 B per row per instruction.)  k is
// permuted inside a chunk (lane half h owns k = 16h..16h+15) -- harmless for a sum
// that only nominates candidates.
#include "vt_common.cuh"

#include <algorithm>
#include <cstdlib>

// Timing experiments on K2 (tools/batch_debug.sh) remove barriers and waits from the hot
// loop: results are garbage with any bit set, and the host's acceptance test cannot tell.
// They exist only in builds made with -DVT_BATCH_TIMING_EXPERIMENTS (make EXPERIMENTS=1);
// the product library ignores VT_BATCH_DEBUG.
#ifdef VT_BATCH_TIMING_EXPERIMENTS
#define VT_DBG(a, bit) ((a).debug & (bit))
#else
#define VT_DBG(a, bit) false
#endif

namespace vt {

using namespace dev;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowWaves = 4;  // waves per block, each owning RT * 32 rows of the block's tile
constexpr int kQStride = 32;        // LDS floats per query row of a 32-k chunk (linear, swizzled slots)

// One LDS-DMA piece: 64 lanes x 16 B from (wave-uniform base + 32-bit lane offset) to
// LDS at lds_addr + lane * 16.  Written out so that the address is the SGPR-base form
// (no 64-bit VALU add per piece) and the m0 write sits right in front of the load.
// m0 is not listed as clobbered (the compiler reserves it); nothing else in these kernels
// uses it: gfx9+ LDS instructions do not, and there is no register-indexed access.
__device__ __forceinline__ void dma16(uint32_t lds_addr, const void *base, uint32_t lane_off) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(sl), "v"(lane_off), "s"(sb));
}

// Cold path of the epilogue: some score of this lane's 16 (one query column, 16
// rows) reaches the threshold.  Inlined into its (rare) branch: as an out-of-line call
// it made the wave save and restore its live registers, 4 % of the pass.
__device__ __forceinline__ void append_candidates(const BatchScoreArgs &a, f32x16 v, float tau, uint32_t qcol,
                                               uint32_t row0, int h) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t row = row0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    const float s = v[i];
    if (s >= tau && row < a.n_total) {
      const uint32_t pos = atomicAdd(&a.cand_count[qcol], 1u);
      if (pos < a.cand_cap) {
        BatchCand cnd;
        cnd.score = s;
        cnd.row = row;
        a.cand[(size_t)qcol * a.cand_cap + pos] = cnd;
      }
    }
  }
}

// NT: 32-query tiles in the batch (a wave covers them all: 32 rows x NT*32
// queries, NT*16 accumulator registers).
//
// The block's work is ONE sequence of 32-k chunks running through all of its row
// tiles, carried by a ring of three LDS stages:
//   * the DMA of chunk m+2 is issued in quarters under the four MFMA steps of
//     chunk m -- across tile boundaries too, so a new tile starts with its first
//     two chunks already on their way (a first version drained the ring per tile:
//     a full HBM latency without MFMAs every 24 chunks);
//   * after step 2 of chunk m every wave waits for ITS pieces of chunk m+1
//     (counted vmcnt: the three quarters of chunk m+2 just issued stay in flight)
//     and for its own fragment reads, then one raw s_barrier: chunk m+1 is
//     complete for everybody and nobody reads the stage of chunk m-1 any more;
//   * under step 3 the fragments of chunk m+1's first step are read, so the MFMA
//     stream does not stop at the chunk boundary either.
// (64 rows per wave with two stages was measured 10 % slower: no registers left to
// pipeline the fragment reads.)
template <int NT, bool DENSE>
__global__ __launch_bounds__(kRowWaves *kWave) void mfma_scores_kernel(const BatchScoreArgs a) {
  extern __shared__ __align__(16) float qlds[];  // [3][NT*32][32] queries, then [3][4 waves][32][32] rows
  constexpr int NS = 3;
  constexpr int NQ = NT * 32;
  constexpr int kTileRowsB = kRowWaves * 32;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t nchunk = a.ld / 32;
  const uint32_t ntiles = (a.n + kTileRowsB - 1) / kTileRowsB;
  if (blockIdx.x >= ntiles) return;
  const uint32_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;

  // thresholds of the query columns this lane sees (column = 32*t + r)
  float tau[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tau[t] = DENSE ? 0.f : a.tau[t * 32 + r];

  // Staging by LDS-DMA: one wave instruction fills 8 rows (1 KiB); lane L lands
  // in row L/8, physical slot L%8 and fetches logical slot (L%8) ^ ((row >> 1) & 7)
  // of that row.  Per-lane source pointers are set up once per tile and advanced
  // by the chunk.
  constexpr int kDmaQ = NQ / 8 / kRowWaves;  // Q pieces per wave per chunk
  constexpr int kDmaX = 4;                   // X pieces per wave per chunk (32 rows)
  static_assert(kDmaQ * 8 * kRowWaves == NQ, "query rows split evenly over the waves");
  constexpr int kDmaPerChunk = kDmaQ + kDmaX;
  // pieces of a chunk issued with steps 0..2 (piece i goes with step i % 4)
  constexpr int kDmaBy3 = (kDmaQ < 3 ? kDmaQ : (kDmaQ / 4) * 3 + (kDmaQ % 4 < 3 ? kDmaQ % 4 : 3)) + 3;
  // physical 16-B slot of logical slot s in row q
  auto qslot = [&](uint32_t q, uint32_t s) { return s ^ ((q >> 1) & 7); };
  // Addresses are (wave-uniform base) + (32-bit per-lane byte offset): the base moves
  // by the chunk in SGPRs, the lane offsets are set once per tile, so a piece costs
  // an m0 write and the load itself -- no per-piece VALU between the MFMAs.
  uint32_t qoff[kDmaQ];
#pragma unroll
  for (int i = 0; i < kDmaQ; ++i) {
    const uint32_t qrow = (uint32_t)(wid * kDmaQ + i) * 8 + (lane >> 3);
    qoff[i] = (qrow * a.ld + qslot(qrow, lane & 7) * 4) * 4;
  }
  const char *qbase = reinterpret_cast<const char *>(a.Q);
  // LDS byte address of the staging area (one address-space cast, integer arithmetic after it)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)qlds;
  auto dma_q = [&](int i, uint32_t c, int stage) {
    const uint32_t dst = lds0 + (uint32_t)(stage * (NQ * kQStride) + (wid * kDmaQ + i) * 8 * kQStride) * 4u;
    dma16(dst, qbase + (size_t)c * 128, qoff[i]);
  };
  float *xlds = qlds + NS * (NQ * kQStride) + wid * (32 * kQStride);
  uint32_t xoff[kDmaX];      // of the tile the DMA cursor is in, relative to xbase
  const char *xbase = nullptr;  // first row of that block tile (wave-uniform)
  auto dma_x = [&](int i, uint32_t c, int stage) {
    const uint32_t dst = lds0 + (uint32_t)(NS * (NQ * kQStride) + wid * (32 * kQStride) +
                                           stage * (kRowWaves * 32 * kQStride) + i * 8 * kQStride) * 4u;
    dma16(dst, xbase + (size_t)c * 128, xoff[i]);
  };
  auto tile_row0 = [&](uint32_t k) {
    const uint32_t tile = blockIdx.x + k * gridDim.x;
    return (DENSE ? tile * a.sample_stride : tile) * kTileRowsB + wid * 32;  // DENSE: a strided sample of the tiles
  };
  auto set_xsrc = [&](uint32_t k) {
    const uint32_t row0 = tile_row0(k);
    const uint32_t block0 = row0 - wid * 32;  // first row of the block tile (< n_total for every tile visited)
    xbase = reinterpret_cast<const char *>(a.X + (size_t)block0 * a.stride);
#pragma unroll
    for (int i = 0; i < kDmaX; ++i) {
      const uint32_t xr = (uint32_t)i * 8 + (lane >> 3);  // row within the wave's rows
      uint32_t grow = row0 + xr;
      grow = grow < a.n_total ? grow : a.n_total - 1;  // clamped for the load, masked in the epilogue
      xoff[i] = ((grow - block0) * (uint32_t)a.stride + qslot(xr, lane & 7) * 4) * 4;
    }
  };
  auto dma_chunk = [&](uint32_t c, int stage) {
#pragma unroll
    for (int i = 0; i < kDmaX; ++i) dma_x(i, c, stage);
#pragma unroll
    for (int i = 0; i < kDmaQ; ++i) dma_q(i, c, stage);
  };
  // DMA cursor: tile dk (of this block's), chunk dc
  uint32_t dk = 0, dc = 0;
  // past the end of the sequence the cursor stays on the last chunk: the steady
  // state keeps issuing (into a stage nobody reads any more) so that the loop body
  // has no branches and the vmcnt arithmetic never changes
  auto dma_advance = [&]() {
    if (dc + 1 == nchunk && dk + 1 == my_tiles) return;
    dc += 1;
    if (dc == nchunk) {
      dc = 0;
      dk += 1;
      set_xsrc(dk);
    }
  };

  f32x4 xa, qv[NT], xa_n, qv_n[NT];
  auto read_frags = [&](int stage, int j, f32x4 &xd, f32x4 *qd) {
    // the swizzle depends only on r (tile bases are multiples of 32 rows)
    const uint32_t so = qslot(r, 4 * h + j) * 4;
    const float *xb = xlds + stage * (kRowWaves * 32 * kQStride) + r * kQStride;
    const float *qb = qlds + stage * (NQ * kQStride) + r * kQStride;
    xd = *reinterpret_cast<const f32x4 *>(xb + so);
#pragma unroll
    for (int t = 0; t < NT; ++t) qd[t] = *reinterpret_cast<const f32x4 *>(qb + so + t * 32 * kQStride);
  };

  // ring prologue: chunks 0 and 1 of the sequence, fragments of (chunk 0, step 0)
  set_xsrc(0);
  dma_chunk(0, 0);
  dma_advance();
  dma_chunk(dc, 1);
  dma_advance();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerChunk) : "memory");
  __builtin_amdgcn_s_barrier();
  read_frags(0, 0, xa, qv);

  int stage = 0;
  for (uint32_t k = 0; k < my_tiles; ++k) {
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (uint32_t c = 0; c < nchunk; ++c) {
      stage = __builtin_amdgcn_readfirstlane(stage);             // keep the ring index (and every LDS base) scalar
      const int stage_n = stage == NS - 1 ? 0 : stage + 1;      // chunk m + 1
      const int stage_a = stage_n == NS - 1 ? 0 : stage_n + 1;  // chunk m + 2 == the stage chunk m - 1 was read from
      const uint32_t cdma = __builtin_amdgcn_readfirstlane(dc);
      // One step = 32-k quarter of the chunk: NT*4 MFMAs.  Everything else the step
      // has to issue is pinned between them, one item per MFMA, so that nothing
      // piles up in front of the matrix pipe: the NT+1 fragment reads of the next
      // step (under the first NT MFMAs) and this step's share of chunk m+2's DMA
      // (one piece under each of the three remaining groups).
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rs = j < 3 ? stage : stage_n;  // step 3 reads (chunk m+1, step 0)
        const uint32_t so = qslot(r, 4 * h + (j < 3 ? j + 1 : 0)) * 4;
        const float *xb = xlds + rs * (kRowWaves * 32 * kQStride) + r * kQStride + so;
        const float *qb = qlds + rs * (NQ * kQStride) + r * kQStride + so;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[e], qv[t][e], acc[t], 0, 0, 0);
            if (e == 0) {
              if (t == 0) xa_n = *reinterpret_cast<const f32x4 *>(xb);
              qv_n[t] = *reinterpret_cast<const f32x4 *>(qb + t * 32 * kQStride);
            } else if (t == 0) {
              if (e == 1) dma_x(j, cdma, stage_a);
              if (e == 2 && j < kDmaQ) dma_q(j, cdma, stage_a);
              if (e == 3 && j + 4 < kDmaQ) dma_q(j + 4, cdma, stage_a);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (j == 2) {
          // my pieces of chunk m+1 have landed, my reads of chunk m are done
          if (!VT_DBG(a, 4u)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kDmaBy3) : "memory");
          if (!VT_DBG(a, 2u)) __builtin_amdgcn_s_barrier();
        }
        xa = xa_n;
#pragma unroll
        for (int t = 0; t < NT; ++t) qv[t] = qv_n[t];
      }
      dma_advance();
      stage = stage_n;
    }

    // epilogue, one 32x32 tile at a time (C layout: column = lane & 31,
    // row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5))
    const uint32_t grow0 = tile_row0(k);
    float xn[16];
    if (a.xnorm2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t row = grow0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        xn[i] = a.xnorm2[row < a.n_total ? row : a.n_total - 1];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const uint32_t qcol = t * 32 + r;
      f32x16 v = acc[t];
      if (a.xnorm2) {
        // L2 family: rank by s = 2 q.x - |x|^2 (larger s <=> smaller |q - x|^2)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 2.0f * v[i] - xn[i];
      }
      if (DENSE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const uint32_t off = (i & 3) + 8 * (i >> 2) + 4 * h;
          // dense sample matrix [query][sample row]
          const uint32_t srow = (blockIdx.x + k * gridDim.x) * kTileRowsB + wid * 32 + off;
          a.sample[(size_t)qcol * a.sample_rows + srow] = grow0 + off < a.n_total ? v[i] : -INFINITY;
        }
      } else {
        float mx = v[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
        if (mx >= tau[t] && !VT_DBG(a, 8u)) append_candidates(a, v, tau[t], qcol, grow0, h);
      }
    }
  }
}

// (r06: the 128 x 128 register-tile variant for 256-query batches -- VT_BATCH_KERNEL=3: 140 TFLOP/s, no faster than the
// kernel above, DESIGN_APPENDIX A.3 -- has left the library)

// tau_b = the `rank`-th largest of the query's sample scores (one block per query), by an
// MSD radix select on the order-preserving u32 image of the scores: the block keeps its
// sample row in registers (<= 64 values per thread) and resolves four 8-bit digits with an
// LDS histogram each -- ~10 us whatever the rank (a first version took `rank` passes of
// "largest value not yet taken": 71 us at rank 12).
constexpr int kTauThreads = 1024, kTauPerThread = 64;

// descending order: larger score = smaller key
__device__ __forceinline__ uint32_t tau_key(float f) { return ~orderable(f); }

__global__ __launch_bounds__(kTauThreads) void sample_tau_kernel(const float *__restrict__ sample, uint32_t sample_rows,
                                                                 uint32_t rank, float *__restrict__ tau, uint32_t nq_real) {
  // padding columns of the batch (all-zero queries) must never nominate a row
  if (blockIdx.x >= nq_real) {
    if (threadIdx.x == 0) tau[blockIdx.x] = INFINITY;
    return;
  }
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_bin, s_below;
  const float *v = sample + (size_t)blockIdx.x * sample_rows;
  const int lane = threadIdx.x & (kWave - 1);
  // this thread's keys (index = threadIdx.x + u * kTauThreads); ~0 = absent
  uint32_t key[kTauPerThread];
#pragma unroll
  for (int u = 0; u < kTauPerThread; ++u) {
    const uint32_t i = threadIdx.x + (uint32_t)u * kTauThreads;
    key[u] = i < sample_rows ? tau_key(v[i]) : 0xFFFFFFFFu;
  }
  uint32_t prefix = 0, mask = 0, krem = rank;  // rank-th smallest key == rank-th largest score
  if (rank <= 16) {
    // Small ranks (K2b takes tau from rank 6 of 65 536 at N = 10 M): walk up the distinct keys from
    // the smallest -- block-wide minimum of the keys above the last one and its multiplicity, at
    // most `rank` rounds of ~2 us -- instead of four histogram passes whose LDS atomics pile up on
    // the few bins the best scores share (62 us -> 17 us per 256 queries).
    __shared__ uint32_t s_min[kTauThreads / kWave], s_cnt[kTauThreads / kWave];
    uint32_t above = 0;  // keys >= above are still in play
    uint32_t answer = 0xFFFFFFFFu;
    for (uint32_t round = 0; round < rank; ++round) {
      uint32_t m = 0xFFFFFFFFu;
#pragma unroll
      for (int u = 0; u < kTauPerThread; ++u) m = (key[u] >= above && key[u] < m) ? key[u] : m;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o, kWave);
        m = t < m ? t : m;
      }
      if (lane == 0) s_min[threadIdx.x / kWave] = m;
      __syncthreads();
      uint32_t gm = 0xFFFFFFFFu;
#pragma unroll
      for (int w = 0; w < kTauThreads / kWave; ++w) gm = s_min[w] < gm ? s_min[w] : gm;
      uint32_t cnt = 0;
#pragma unroll
      for (int u = 0; u < kTauPerThread; ++u) cnt += key[u] == gm ? 1u : 0u;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, kWave);
      if (lane == 0) s_cnt[threadIdx.x / kWave] = cnt;
      __syncthreads();
      uint32_t total = 0;
#pragma unroll
      for (int w = 0; w < kTauThreads / kWave; ++w) total += s_cnt[w];
      answer = gm;
      // absent slots hold ~0: when the sample runs out the smallest score (largest key seen) stays
      if (gm == 0xFFFFFFFFu || total >= krem) break;
      krem -= total;
      above = gm + 1;
      __syncthreads();
    }
    if (answer == 0xFFFFFFFFu) {  // fewer than rank values: the smallest score
      uint32_t mx = 0;
#pragma unroll
      for (int u = 0; u < kTauPerThread; ++u) mx = (key[u] != 0xFFFFFFFFu && key[u] > mx) ? key[u] : mx;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)mx, o, kWave);
        mx = t > mx ? t : mx;
      }
      __syncthreads();
      if (lane == 0) s_min[threadIdx.x / kWave] = mx;
      __syncthreads();
      answer = 0;
#pragma unroll
      for (int w = 0; w < kTauThreads / kWave; ++w) answer = s_min[w] > answer ? s_min[w] : answer;
    }
    if (threadIdx.x == 0) {
      const uint32_t o = ~answer;
      const uint32_t bits = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
      tau[blockIdx.x] = __uint_as_float(bits);
    }
    return;
  }
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kTauPerThread; ++u) {
      const uint32_t i = threadIdx.x + (uint32_t)u * kTauThreads;
      if (i < sample_rows && (key[u] & mask) == prefix) atomicAdd(&hist[(key[u] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kWave) {
      // bin holding the krem-th smallest: lane l owns bins 4l .. 4l+3
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
            if (below + c < krem) { below += c; b += 1; }
          }
        }
        s_bin = b;
        s_below = below;
      }
      if (lane == kWave - 1 && incl < krem) {  // fewer than rank values: the smallest score
        s_bin = 255;
        s_below = 0;
      }
    }
    __syncthreads();
    prefix |= s_bin << shift;
    mask |= 255u << shift;
    krem -= s_below;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // prefix is the key of the rank-th largest score
    const uint32_t o = ~prefix;
    const uint32_t bits = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    tau[blockIdx.x] = __uint_as_float(bits);
  }
}

// Per-row squared norms (f64 accumulation, stored as f32) and their maximum: the
// L2-family score term and the row-norm bound of the error margin.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ X, size_t stride, uint32_t n,
                                                         uint32_t d, float *__restrict__ xnorm2,
                                                         unsigned long long *out_bits) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  double best = 0.0;
  for (uint32_t row = wave; row < n; row += nwaves) {
    const float *x = X + (size_t)row * stride;
    double s = 0.0;
    for (uint32_t j = lane; j < d; j += kWave) s += (double)x[j] * (double)x[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
    if (lane == 0) xnorm2[row] = (float)s;
    best = s > best ? s : best;
  }
  if (lane == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(best));  // s >= 0: bits are monotone
}

// The same for a list of rows (norms of mutated rows patched in place).
__global__ __launch_bounds__(256) void row_sqnorm_rows_kernel(const float *__restrict__ X, size_t stride,
                                                              const uint32_t *__restrict__ list, uint32_t count, uint32_t d,
                                                              float *__restrict__ xnorm2, unsigned long long *out_bits) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
  double best = 0.0;
  for (uint32_t i = wave; i < count; i += nwaves) {
    const uint32_t row = list[i];
    const float *x = X + (size_t)row * stride;
    double s = 0.0;
    for (uint32_t j = lane; j < d; j += kWave) s += (double)x[j] * (double)x[j];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
    if (lane == 0) xnorm2[row] = (float)s;
    best = s > best ? s : best;
  }
  if (lane == 0) atomicMax(out_bits, (unsigned long long)__double_as_longlong(best));
}

// Batched K3: block b selects the k smallest of keys[b][0..m) (rank sort; m is small).
// `ex` (a query batch's last kernel, r05): what the host wants to read beside the lists -- every query's candidate
// count and threshold, the status word (moved out and cleared) -- goes out with them, so that `out`, `out_count` and
// these can all be host-mapped and no copy is queued behind the kernel (four blit launches per group before).
__global__ __launch_bounds__(256) void batch_select_kernel(const uint64_t *__restrict__ keys,
                                                           const Payload *__restrict__ pay, uint32_t m, uint32_t k,
                                                           Entry *__restrict__ out, uint32_t *__restrict__ out_count, BatchExport ex) {
  if (threadIdx.x == 0) {
    if (ex.cand_count_out) ex.cand_count_out[blockIdx.x] = ex.cand_count[blockIdx.x];
    if (ex.tau_out && ex.tau) ex.tau_out[blockIdx.x] = ex.tau[blockIdx.x];
    if (ex.status_out && blockIdx.x == 0) {
      *ex.status_out = *ex.status;
      *ex.status = 0;
    }
  }
  extern __shared__ __align__(16) unsigned char bsm[];
  uint64_t *sk = reinterpret_cast<uint64_t *>(bsm);  // [m]
  const uint64_t *kb = keys + (size_t)blockIdx.x * m;
  const Payload *pb = pay + (size_t)blockIdx.x * m;
  __shared__ uint32_t s_live;
  if (threadIdx.x == 0) s_live = 0;
  for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) sk[i] = kb[i];
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
    if (pos < k) {
      Entry e;
      e.key = ki;
      e.row = pb[i].row;
      e.raw = pb[i].raw;
      out[(size_t)blockIdx.x * k + pos] = e;
    }
  }
  atomicAdd(&s_live, live);
  __syncthreads();
  if (threadIdx.x == 0) out_count[blockIdx.x] = s_live < k ? s_live : k;
}

template <int NT>
hipError_t launch_scores_nt(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s) {
  const size_t lds = (size_t)3 * (NT * 32 + kRowWaves * 32) * kQStride * sizeof(float);
  const dim3 block(kRowWaves * kWave);
  if (dense) {
    auto kern = mfma_scores_kernel<NT, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), block, lds, s, a);
  } else {
    auto kern = mfma_scores_kernel<NT, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), block, lds, s, a);
  }
  return hipGetLastError();
}

}  // namespace

uint32_t batch_rows_per_block(uint32_t) { return kRowWaves * 32; }

hipError_t launch_batch_scores(const BatchScoreArgs &a0, bool dense, uint32_t blocks, hipStream_t s) {
  BatchScoreArgs a = a0;
#ifdef VT_BATCH_TIMING_EXPERIMENTS
  a.debug = dense ? 0u : (uint32_t)env::get(env::BATCH_DEBUG);
#else
  a.debug = 0u;
#endif
  if (a.ld % 32 != 0 || a.nq_pad % 32 != 0 || a.nq_pad == 0 || a.nq_pad > 256) return hipErrorInvalidValue;
  switch (a.nq_pad / 32) {
    case 1: return launch_scores_nt<1>(a, dense, blocks, s);
    case 2: return launch_scores_nt<2>(a, dense, blocks, s);
    case 4: return launch_scores_nt<4>(a, dense, blocks, s);
    case 8: return launch_scores_nt<8>(a, dense, blocks, s);
    default: return hipErrorInvalidValue;
  }
}

// tau_b from the sample's GROUP MAXIMA (K2s since r05: the sample pass files, per query, the best score of every
// 64-row group of its tiles -- `groups` <= 1 024 values -- instead of every score): the `rank`-th largest of them.
// Every maximum is one of the sample's scores, so this is at most the `rank`-th largest score of the whole sample
// (equal to it unless two of the `rank` best share a group: the host only comes here with rank * 16 <= groups <= 2 048) -- a
// threshold that is a little lower nominates a few more rows and changes nothing else.  One wave per query, 32 keys
// per lane in registers, the answer bit by bit from the top: 32 steps of 32 compares and a count -- 12-15 us per launch
// where the radix select over 65 536 scores took 45 (and wrote / read 64 MB around it).
constexpr int kTauGroupsPerLane = 32;
__global__ __launch_bounds__(256) void sample_tau_groups_kernel(const float *__restrict__ maxima, uint32_t groups, uint32_t rank,
                                                                float *__restrict__ tau, uint32_t nq, uint32_t nq_real) {
  const uint32_t q = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  if (q >= nq) return;
  if (q >= nq_real) {  // padding columns of the batch (all-zero queries) must never nominate a row
    if (lane == 0) tau[q] = INFINITY;
    return;
  }
  const float *v = maxima + (size_t)q * groups;
  uint32_t key[kTauGroupsPerLane];  // ascending with the score; 0 = absent (below every real score's key)
#pragma unroll
  for (int u = 0; u < kTauGroupsPerLane; ++u) {
    const uint32_t i = (uint32_t)lane + (uint32_t)u * kWave;
    key[u] = i < groups ? orderable(v[i]) : 0u;
  }
  uint32_t t = 0;
  for (int b = 31; b >= 0; --b) {
    const uint32_t cand = t | (1u << b);
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < kTauGroupsPerLane; ++u) c += (uint32_t)__builtin_popcountll(__ballot(key[u] >= cand));
    if (c >= rank) t = cand;  // (wave-uniform: every lane counted the same ballots)
  }
  // fewer than `rank` values (t stayed 0): the smallest score of the sample
  if (t == 0u) {
    uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
    for (int u = 0; u < kTauGroupsPerLane; ++u) mn = (key[u] != 0u && key[u] < mn) ? key[u] : mn;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t x = (uint32_t)__shfl_xor((int)mn, o, kWave);
      mn = x < mn ? x : mn;
    }
    t = mn;
  }
  if (lane == 0) tau[q] = __uint_as_float((t & 0x80000000u) ? (t & 0x7FFFFFFFu) : ~t);
}

hipError_t launch_sample_tau_groups(const float *maxima, uint32_t groups, uint32_t nq, uint32_t nq_real, uint32_t rank, float *tau,
                                    hipStream_t s) {
  if (groups == 0 || groups > (uint32_t)kTauGroupsPerLane * kWave || rank == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sample_tau_groups_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, maxima, groups, rank, tau, nq, nq_real);
  return hipGetLastError();
}

hipError_t launch_sample_tau(const float *sample, uint32_t sample_rows, uint32_t nq, uint32_t nq_real, uint32_t rank,
                             float *tau, hipStream_t s) {
  hipLaunchKernelGGL(sample_tau_kernel, dim3(nq), dim3(kTauThreads), 0, s, sample, sample_rows, rank, tau, nq_real);
  return hipGetLastError();
}

hipError_t launch_row_sqnorms(const float *X, size_t stride, uint32_t n, uint32_t d, float *xnorm2,
                              unsigned long long *out_bits, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(2048), dim3(256), 0, s, X, stride, n, d, xnorm2, out_bits);
  return hipGetLastError();
}

hipError_t launch_row_sqnorms_rows(const float *X, size_t stride, const uint32_t *list, uint32_t count, uint32_t d,
                                   float *xnorm2, unsigned long long *out_bits, hipStream_t s) {
  if (count == 0) return hipSuccess;
  const uint32_t blocks = std::min<uint32_t>(2048, (count + 3) / 4);
  hipLaunchKernelGGL(row_sqnorm_rows_kernel, dim3(blocks), dim3(256), 0, s, X, stride, list, count, d, xnorm2, out_bits);
  return hipGetLastError();
}

hipError_t launch_batch_select(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m, uint32_t k, Entry *out,
                               uint32_t *out_count, hipStream_t s, const BatchExport *ex) {
  if (m == 0 || m > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(batch_select_kernel, dim3(nq), dim3(256), (size_t)m * 8, s, keys, pay, m, k, out, out_count,
                     ex ? *ex : BatchExport{});
  return hipGetLastError();
}

}  // namespace vt
