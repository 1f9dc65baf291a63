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
// Both operands of a 32-wide k chunk go through LDS, double buffered, filled by
// LDS-DMA (global_load_lds, 16 B/lane, whole 128-B lines, no staging
// registers): the Q chunk (all queries) is shared by the block, each wave's 32
// X rows are private to it.  The images are linear [row][8 slots of 16 B] and
// the slot index is XOR-swizzled with (row >> 1) & 7 -- applied to the per-lane
// SOURCE address on the way in and to the read address on the way out -- which
// makes every ds_read_b128 conflict-free.  (Fragment-shaped X loads straight to
// registers were measured 6-20 % slower: 32 B per row per instruction.)  k is permuted inside a chunk
// (lane half h owns k = 16h..16h+15) -- harmless for a sum that only nominates
// candidates.
#include "vt_common.cuh"

#include <cstdlib>

namespace vt {

using namespace dev;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowWaves = 4;  // waves per block, each owning RT * 32 rows of the block's tile
constexpr int kQStride = 32;        // LDS floats per query row of a 32-k chunk (linear, swizzled slots)

// Cold path of the epilogue: some score of this lane's 16 (one query column, 16
// rows) reaches the threshold.  Kept out of line so the hot loop stays lean.
__device__ __noinline__ void append_candidates(const BatchScoreArgs &a, f32x16 v, float tau, uint32_t qcol,
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

// NT: 32-query tiles in the batch (a wave covers them all); RT: 32-row groups per
// wave (RT = 2: 64 rows x 256 queries = 256 accumulator registers, 256 MFMAs per
// barrier); NS: LDS stages (the DMA runs NS - 1 chunks ahead of the MFMAs).
template <int NT, int RT, int NS, bool DENSE>
__global__ __launch_bounds__(kRowWaves *kWave) void mfma_scores_kernel(const BatchScoreArgs a) {
  extern __shared__ __align__(16) float qlds[];  // [NS][NT*32][32] queries, then [NS][4 waves][RT*32][32] rows
  constexpr int NQ = NT * 32;
  constexpr int kWaveRows = RT * 32;
  constexpr int kTileRowsB = kRowWaves * kWaveRows;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t nchunk = a.ld / 32;
  const uint32_t ntiles = (a.n + kTileRowsB - 1) / kTileRowsB;

  // thresholds of the query columns this lane sees (column = 32*t + r)
  float tau[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tau[t] = DENSE ? 0.f : a.tau[t * 32 + r];

  // Staging by LDS-DMA: one wave instruction fills 8 rows (1 KiB); lane L lands
  // in row L/8, physical slot L%8 and fetches logical slot (L%8) ^ ((row >> 1) & 7)
  // of that row.  Per-lane source pointers are set up once and advanced by the
  // chunk; the pieces are issued one at a time so they can be spread between
  // the MFMAs of a chunk.
  constexpr int kDmaQ = NQ / 8 / kRowWaves;  // Q pieces per wave per chunk
  constexpr int kDmaX = RT * 4;              // X pieces per wave per chunk
  static_assert(kDmaQ * 8 * kRowWaves == NQ, "query rows split evenly over the waves");
  constexpr int kDmaPerChunk = kDmaQ + kDmaX;
  // physical 16-B slot of logical slot s in row q
  auto qslot = [&](uint32_t q, uint32_t s) { return s ^ ((q >> 1) & 7); };
  const float *qsrc[kDmaQ];
#pragma unroll
  for (int i = 0; i < kDmaQ; ++i) {
    const uint32_t qrow = (uint32_t)(wid * kDmaQ + i) * 8 + (lane >> 3);
    qsrc[i] = a.Q + (size_t)qrow * a.ld + qslot(qrow, lane & 7) * 4;
  }
  auto dma_q = [&](int i, uint32_t c, int stage) {
    float *dst = qlds + stage * (NQ * kQStride) + (wid * kDmaQ + i) * 8 * kQStride;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(qsrc[i] + c * 32),
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };
  // this wave's RT*32 X rows: RT*4 pieces of 8 whole lines each
  float *xlds = qlds + NS * (NQ * kQStride) + wid * (kWaveRows * kQStride);
  const float *xsrc[kDmaX];
  auto dma_x = [&](int i, uint32_t c, int stage) {
    float *dst = xlds + stage * (kRowWaves * kWaveRows * kQStride) + i * 8 * kQStride;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xsrc[i] + c * 32),
                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
  };
  // all of a chunk's pieces, or the share that goes with step j of the MFMA loop
  auto dma_chunk = [&](uint32_t c, int stage) {
#pragma unroll
    for (int i = 0; i < kDmaX; ++i) dma_x(i, c, stage);
#pragma unroll
    for (int i = 0; i < kDmaQ; ++i) dma_q(i, c, stage);
  };
  // With two stages the chunk must land within the iteration that issues it, so
  // its pieces go out with the first two steps; with three stages they spread
  // over all four.
  constexpr int kDmaSteps = NS == 2 ? 2 : 4;
  auto dma_step = [&](int j, uint32_t c, int stage) {
    if (j >= kDmaSteps) return;
#pragma unroll
    for (int i = 0; i < kDmaX; ++i)
      if (i % kDmaSteps == j) dma_x(i, c, stage);
#pragma unroll
    for (int i = 0; i < kDmaQ; ++i)
      if (i % kDmaSteps == j) dma_q(i, c, stage);
  };

  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // DENSE (pass 0) visits a strided sample of the tiles
    const uint32_t rtile = DENSE ? tile * a.sample_stride : tile;
    const uint32_t row0 = rtile * kTileRowsB + wid * kWaveRows;
    // rows past the end are clamped for the load and masked in the epilogue
#pragma unroll
    for (int i = 0; i < kDmaX; ++i) {
      const uint32_t xr = (uint32_t)i * 8 + (lane >> 3);  // row within the wave's rows
      uint32_t grow = row0 + xr;
      grow = grow < a.n_total ? grow : a.n_total - 1;
      xsrc[i] = a.X + (size_t)grow * a.stride + qslot(xr, lane & 7) * 4;
    }

    f32x16 acc[RT][NT];
#pragma unroll
    for (int g = 0; g < RT; ++g)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][t][i] = 0.f;

    // NS LDS stages, DMA NS-1 chunks ahead of the MFMAs.  Raw s_barrier + counted
    // vmcnt: a __syncthreads() would drain the DMA that is meant to stay in flight.
    __builtin_amdgcn_s_barrier();  // every wave is done reading the previous tile's stages
    dma_chunk(0, 0);
    if (NS == 3 && nchunk > 1) {
      dma_chunk(1, 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerChunk) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    for (uint32_t c = 0; c < nchunk; ++c) {
      const int stage_a = (stage + NS - 1) % NS;  // stage of chunk c + NS - 1 == the one read in iteration c - 1
      const bool ahead = c + (NS - 1) < nchunk;
      const float *xb = xlds + stage * (kRowWaves * kWaveRows * kQStride) + r * kQStride;
      // the swizzle depends only on r (tile bases are multiples of 32 rows), so the
      // per-tile address is a constant offset from four per-lane bases
      const float *qb = qlds + stage * (NQ * kQStride) + r * kQStride;
      // software pipeline over the four 8-k steps (RT == 1): fragments of step j+1
      // are read while step j's MFMAs run; with 64 rows per wave the accumulators
      // already take half the register file, so the fragments are read per step
      // (there are twice as many MFMAs per read to hide them behind).
      constexpr bool kPipe = RT == 1;
      f32x4 xa[RT], qv[NT], xa_n[kPipe ? RT : 1], qv_n[kPipe ? NT : 1];
      auto read_frags = [&](int j, f32x4 *xd, f32x4 *qd) {
        const uint32_t so = qslot(r, 4 * h + j) * 4;
#pragma unroll
        for (int g = 0; g < RT; ++g) xd[g] = *reinterpret_cast<const f32x4 *>(xb + so + g * 32 * kQStride);
#pragma unroll
        for (int t = 0; t < NT; ++t) qd[t] = *reinterpret_cast<const f32x4 *>(qb + so + t * 32 * kQStride);
      };
      if (kPipe) read_frags(0, xa, qv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (kPipe) {
          if (j < 3) read_frags(j + 1, xa_n, qv_n);
        } else {
          read_frags(j, xa, qv);
        }
        if (ahead) dma_step(j, c + (NS - 1), stage_a);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int g = 0; g < RT; ++g)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[g][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[g][e], qv[t][e], acc[g][t], 0, 0, 0);
        }
        if (kPipe && j < 3) {
#pragma unroll
          for (int g = 0; g < RT; ++g) xa[g] = xa_n[g];
#pragma unroll
          for (int t = 0; t < NT; ++t) qv[t] = qv_n[t];
        }
      }
      // chunk c + 1 must have landed for everyone before anyone reads it
      if (NS == 3 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDmaPerChunk) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      stage = stage == NS - 1 ? 0 : stage + 1;
    }

    // epilogue, one 32x32 tile at a time (C layout: column = lane & 31,
    // row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5))
#pragma unroll
    for (int g = 0; g < RT; ++g) {
      const uint32_t grow0 = row0 + g * 32;
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
        f32x16 v = acc[g][t];
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
            const uint32_t srow = tile * kTileRowsB + wid * kWaveRows + g * 32 + off;
            a.sample[(size_t)qcol * a.sample_rows + srow] = grow0 + off < a.n_total ? v[i] : -INFINITY;
          }
        } else {
          float mx = v[0];
#pragma unroll
          for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
          if (mx >= tau[t]) append_candidates(a, v, tau[t], qcol, grow0, h);
        }
      }
    }
  }
}

// tau_b = the `rank`-th largest of the query's sample scores (one block per query).
__global__ __launch_bounds__(256) void sample_tau_kernel(const float *__restrict__ sample, uint32_t sample_rows,
                                                         uint32_t rank, float *__restrict__ tau) {
  __shared__ float s_best[256];
  __shared__ uint32_t s_idx[256];
  __shared__ float s_cut;
  __shared__ uint32_t s_cutidx;
  const float *v = sample + (size_t)blockIdx.x * sample_rows;
  float cut = INFINITY;      // values >= cut (ties by index) were already taken
  uint32_t cutidx = 0xFFFFFFFFu;
  float result = -INFINITY;
  for (uint32_t round = 0; round < rank; ++round) {
    float best = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < sample_rows; i += blockDim.x) {
      const float x = v[i];
      const bool taken = x > cut || (x == cut && i <= cutidx);
      if (!taken && (x > best || (x == best && i < bi))) {
        best = x;
        bi = i;
      }
    }
    s_best[threadIdx.x] = best;
    s_idx[threadIdx.x] = bi;
    __syncthreads();
    if (threadIdx.x == 0) {
      float b = -INFINITY;
      uint32_t ix = 0xFFFFFFFFu;
      for (int t = 0; t < 256; ++t)
        if (s_best[t] > b || (s_best[t] == b && s_idx[t] < ix)) {
          b = s_best[t];
          ix = s_idx[t];
        }
      s_cut = b;
      s_cutidx = ix;
    }
    __syncthreads();
    cut = s_cut;
    cutidx = s_cutidx;
    result = cut;
    __syncthreads();
  }
  if (threadIdx.x == 0) tau[blockIdx.x] = result;
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

// Batched K3: block b selects the k smallest of keys[b][0..m) (rank sort; m is small).
__global__ __launch_bounds__(256) void batch_select_kernel(const uint64_t *__restrict__ keys,
                                                           const Payload *__restrict__ pay, uint32_t m, uint32_t k,
                                                           Entry *__restrict__ out, uint32_t *__restrict__ out_count) {
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

template <int NT, int RT, int NS>
hipError_t launch_scores_nt(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s) {
  const size_t lds = (size_t)NS * (NT * 32 + kRowWaves * RT * 32) * kQStride * sizeof(float);
  const dim3 block(kRowWaves * kWave);
  if (dense) {
    auto kern = mfma_scores_kernel<NT, RT, NS, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), block, lds, s, a);
  } else {
    auto kern = mfma_scores_kernel<NT, RT, NS, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), block, lds, s, a);
  }
  return hipGetLastError();
}

// 32 rows per wave with three stages measured 118 TF against 106 TF for 64 rows
// per wave with two (N=2M, B=256): the wide tile has no registers left to
// pipeline its fragment reads.  VT_BATCH_WIDE=1 selects it for experiments.
bool batch_wide() {
  static const bool wide = std::getenv("VT_BATCH_WIDE") != nullptr;
  return wide;
}

}  // namespace

uint32_t batch_rows_per_block() { return kRowWaves * 32 * (batch_wide() ? 2 : 1); }

hipError_t launch_batch_scores(const BatchScoreArgs &a, bool dense, uint32_t blocks, hipStream_t s) {
  if (a.ld % 32 != 0 || a.nq_pad % 32 != 0 || a.nq_pad == 0 || a.nq_pad > 256) return hipErrorInvalidValue;
  const bool wide = batch_wide();
  switch (a.nq_pad / 32) {
    case 1: return wide ? launch_scores_nt<1, 2, 2>(a, dense, blocks, s) : launch_scores_nt<1, 1, 3>(a, dense, blocks, s);
    case 2: return wide ? launch_scores_nt<2, 2, 2>(a, dense, blocks, s) : launch_scores_nt<2, 1, 3>(a, dense, blocks, s);
    case 4: return wide ? launch_scores_nt<4, 2, 2>(a, dense, blocks, s) : launch_scores_nt<4, 1, 3>(a, dense, blocks, s);
    case 8: return wide ? launch_scores_nt<8, 2, 2>(a, dense, blocks, s) : launch_scores_nt<8, 1, 3>(a, dense, blocks, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sample_tau(const float *sample, uint32_t sample_rows, uint32_t nq, uint32_t rank, float *tau,
                             hipStream_t s) {
  hipLaunchKernelGGL(sample_tau_kernel, dim3(nq), dim3(256), 0, s, sample, sample_rows, rank, tau);
  return hipGetLastError();
}

hipError_t launch_row_sqnorms(const float *X, size_t stride, uint32_t n, uint32_t d, float *xnorm2,
                              unsigned long long *out_bits, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(2048), dim3(256), 0, s, X, stride, n, d, xnorm2, out_bits);
  return hipGetLastError();
}

hipError_t launch_batch_select(const uint64_t *keys, const Payload *pay, uint32_t nq, uint32_t m, uint32_t k, Entry *out,
                               uint32_t *out_count, hipStream_t s) {
  if (m == 0 || m > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(batch_select_kernel, dim3(nq), dim3(256), (size_t)m * 8, s, keys, pay, m, k, out, out_count);
  return hipGetLastError();
}

}  // namespace vt
