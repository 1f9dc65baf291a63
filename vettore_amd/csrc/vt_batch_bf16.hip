// vt_batch_bf16.hip -- K2b: query batches nominated on the BF16 matrix cores (gfx950).
//
// K2 (vt_batch.hip) runs S = X * Q^T on v_mfma_f32_32x32x2_f32 and is bound by that pipe:
// 28 ms per 256 queries at N = 10 M, d = 768, six times what reading the rows takes.  But the
// matrix pass only NOMINATES candidates -- the exact K1 arithmetic re-scores them and the host
// certifies that nothing else can reach the top k (host/vt_batch.h) -- so its
// operands need not be exact: here the f32 rows are streamed as they lie in HBM, rounded to
// bf16 in registers (v_cvt_pk_bf16_f32) and fed to v_mfma_f32_32x32x16_bf16 (16x the FP32
// rate, f32 accumulators); the queries are rounded once (q_image_kernel).  The host's bound
// grows by the operand rounding (2^-8 each, see batch_group) and the candidate lists by a
// factor of a few; the pass becomes HBM-bound.  Ranking and scores stay bit-identical to the
// reference (flat.rs:96-124 run B times) because nothing this kernel computes is ever returned.
//
// Work split: a block of 8 waves owns 256 rows x 256 queries; wave w owns rows 32w..32w+31
// against all queries (8 accumulator tiles of 32 x 32 = 128 registers), two waves per SIMD.
// Per 32-k chunk and block: 32 KiB of rows (whole 128-B lines of 256 rows, f32) and 16 KiB of
// queries (bf16, already in fragment order) arrive in LDS by LDS-DMA, three stages deep, so
// two chunks = 64 KiB of rows are in flight per CU (what K1 keeps in flight).  One counted
// vmcnt + one raw s_barrier per chunk; 20 ds_read_b128, 8 conversions and 16 MFMAs per wave
// and chunk.  k is permuted inside a chunk (lane half h owns k = 16h..16h+15, MFMA step s the
// 8 of them at 8s) -- identically for both operands, so every product still meets its partner.
//
// What bounds it (tools/k2b_probe.hip, 6 M rows x 768): the DMA ring alone streams 6.7 TB/s,
// the MFMA work alone (rows never fetched) takes 1.85-2.0 ms = 1.2-1.3 PFLOP/s -- what an
// LDS-fed bf16 MFMA loop sustains on random data on this part, whose clock drops to ~1.6 GHz
// under it -- and together 3.1 ms = 5.9-6.0 TB/s: the pass is co-limited, memory at ~0.73 of
// the 8 TB/s peak with the matrix pipe busy ~60 % of the time.  Tried and dropped, each
// measured in the same run: the rows staged through registers four chunks ahead instead of
// LDS-DMA (128 KiB in flight per CU, hand-counted vmcnt like K1m: 3.19 ms against 3.08 -- depth
// of the ring is not the limit); fragment reads issued one MFMA step ahead of their use,
// across the barrier too (compute alone 2.06 ms against 2.07: not the fragment traffic either).
#include "vt_common.cuh"

#include <algorithm>
#include <cstdlib>

// Timing experiments (tools/k2b_probe.hip): only in builds made with -DVT_BATCH_TIMING_EXPERIMENTS;
// results are garbage with any bit set.  1: no fragment reads / MFMAs, 2: no barrier, 4: no query
// DMA, 8: no candidate append, 16: no row DMA.
#ifdef VT_BATCH_TIMING_EXPERIMENTS
#define VT_DBG(a, bit) ((a).debug & (bit))
#else
#define VT_DBG(a, bit) false
#endif

namespace vt {

using namespace dev;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWavesB = 8;              // waves per block
constexpr int kRowsB = kWavesB * 32;    // rows per block tile
// QT = 32-query tiles of the batch: 8 (256 queries), 4 (128) or 2 (64).  A batch of 64 pads to 64
// columns, not 256: a quarter of the MFMA work, and the pass that is co-limited by the matrix pipe
// at 256 queries is plainly HBM-bound there (r03: callers that meet on a handle are usually fewer
// than 256).
constexpr int kStages = 3;
constexpr uint32_t q_stage_bytes(int qt) { return (uint32_t)qt * 2 * 1024; }  // [t][s][lane] x 16 B
constexpr uint32_t kXStageBytes = kRowsB * 128;         // [row][8 slots of 16 B], slots XOR-swizzled
constexpr uint32_t kXWaveBytes = 32 * 128;
// DMA pieces per wave and chunk: 4 of rows; of the chunk's query image (QT x 2 KiB) every wave copies
// 2 KiB at QT = 8 and 1 KiB below (at QT = 2 the 4 KiB are copied twice, waves w and w + 4 the same
// KiB: every wave issues the same number of pieces, which is what the counted wait needs)
constexpr int q_pieces(int qt) { return qt == 8 ? 2 : 1; }

// One LDS-DMA piece: 64 lanes x 16 B from (wave-uniform base + 32-bit lane offset) to LDS at
// lds_addr + lane * 16 (see vt_batch.hip: SGPR-base addressing, m0 written right in front).
// NT: the rows are read once (nt), the query image is re-read by every block (default policy).
template <bool NT>
__device__ __forceinline__ void dma16(uint32_t lds_addr, const void *base, uint32_t lane_off) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" : : "s"(sl), "v"(lane_off), "s"(sb) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(sl), "v"(lane_off), "s"(sb) : "memory");
}

__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  const bf16x2 p0 = __builtin_convertvector((f32x2){lo[0], lo[1]}, bf16x2);
  const bf16x2 p1 = __builtin_convertvector((f32x2){lo[2], lo[3]}, bf16x2);
  const bf16x2 p2 = __builtin_convertvector((f32x2){hi[0], hi[1]}, bf16x2);
  const bf16x2 p3 = __builtin_convertvector((f32x2){hi[2], hi[3]}, bf16x2);
  bf16x8 o;
  o[0] = p0[0]; o[1] = p0[1]; o[2] = p1[0]; o[3] = p1[1];
  o[4] = p2[0]; o[5] = p2[1]; o[6] = p3[0]; o[7] = p3[1];
  return o;
}

// The queries as the B operand wants them: image[c][t][s][lane = 32h + r][e] =
// bf16(Q[32t + r][32c + 16h + 8s + e]) -- a fragment read is 1 KiB of consecutive lanes.
__global__ __launch_bounds__(256) void q_image_kernel(const float *__restrict__ Q, uint32_t ld, uint32_t nchunk, uint32_t qt,
                                                      bf16x8 *__restrict__ image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-B fragment slot each
  if (i >= nchunk * qt * 2 * 64) return;
  const uint32_t lane = i & 63, s = (i >> 6) & 1, t = (i >> 7) % qt, c = (i >> 7) / qt;
  const uint32_t r = lane & 31, h = lane >> 5;
  const float *src = Q + (size_t)(32 * t + r) * ld + 32 * c + 16 * h + 8 * s;
  image[i] = pack8(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
}

// Cold path of the epilogue (as in vt_batch.hip): a score of this lane's 16 reaches tau.
__device__ __forceinline__ void append_candidates(const BatchScoreArgs &a, f32x16 v, float tau, uint32_t qcol,
                                                  uint32_t row0, int h) {
  // (the list addresses are loop-invariant; hoisted out of the tile loop they cost 16 registers
  // this kernel does not have -- the compiler may not know where qcol comes from)
  asm volatile("" : "+v"(qcol));
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

// What a wave does with a finished 32-row x 256-query tile (C layout: column = lane & 31 =
// query, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)): pass 0 writes the dense sample
// matrix, pass 1 appends the scores that reach the query's threshold.
template <bool DENSE, int QT>
__device__ __forceinline__ void tile_epilogue(const BatchScoreArgs &a, f32x16 (&acc)[QT], const float (&tau)[QT],
                                              uint32_t grow0, uint32_t srow0, int r, int h) {
  // L2 family: the 32 row norms of this wave's tile.  The address is wave-uniform, so they come
  // through the scalar cache (constant address space => s_load): a vector load here would sit in
  // the same in-order queue as the row DMA, and the wait for it would drain the ring once per tile.
  // (the norm column is as long as the slab's row capacity, a multiple of 32: a tile that starts
  // below n_total ends inside it)
  float xn[16];
  if (a.xnorm2) {
    typedef const __attribute__((address_space(4))) float *cfloat_p;
    const uint32_t g0 = __builtin_amdgcn_readfirstlane(grow0);
    if (g0 < a.n_total) {
      cfloat_p xc = (cfloat_p)(uintptr_t)(a.xnorm2 + g0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int off = (i & 3) + 8 * (i >> 2);
        const float lo = xc[off], hi = xc[off + 4];
        xn[i] = h ? hi : lo;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) xn[i] = 0.f;
    }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const uint32_t qcol = t * 32 + r;
    f32x16 v = acc[t];
    if (a.xnorm2) {
      // L2 family: rank by s = 2 q.x - |x|^2 (larger s <=> smaller |q - x|^2)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = 2.0f * v[i] - xn[i];
    }
    if (DENSE) {
      uint32_t qc2 = qcol;
      asm volatile("" : "+v"(qc2));  // (as in append_candidates: no hoisted address arithmetic)
      const uint32_t qcol = qc2;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t off = (i & 3) + 8 * (i >> 2) + 4 * h;
        // dense sample matrix [query][sample row]
        a.sample[(size_t)qcol * a.sample_rows + srow0 + off] = grow0 + off < a.n_total ? v[i] : -INFINITY;
      }
    } else {
      float mx = v[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, v[i]);
      if (mx >= tau[t] && !VT_DBG(a, 8u)) append_candidates(a, v, tau[t], qcol, grow0, h);
    }
  }
}

template <bool DENSE, int QT>
__global__ __launch_bounds__(kWavesB *kWave, 1) void bf16_scores_kernel(const BatchScoreArgs a) {
  constexpr uint32_t kQStageBytes = q_stage_bytes(QT);
  constexpr int kPiecesPerChunk = 4 + q_pieces(QT);
  extern __shared__ __align__(16) unsigned char lds[];  // [3] query stages, then [3][8 waves] row stages
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const uint32_t nchunk = a.ld / 32;
  const uint32_t ntiles = (a.n + kRowsB - 1) / kRowsB;
  if (blockIdx.x >= ntiles) return;
  const uint32_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;

  float tau[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) tau[t] = DENSE ? 0.f : a.tau[t * 32 + r];

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  const uint32_t xlds0 = lds0 + kStages * kQStageBytes + wid * kXWaveBytes;
  auto xslot = [&](uint32_t row, uint32_t s) { return s ^ ((row >> 1) & 7); };

  // queries: the chunk's image is one run of QT x 2 KiB; this wave copies KiB 2w and 2w + 1 of it
  // (QT = 8), KiB w (QT = 4), KiB w mod 4 (QT = 2)
  const char *qimg = reinterpret_cast<const char *>(a.Qimage);
  const uint32_t qkib = QT == 8 ? (uint32_t)wid * 2 : QT == 4 ? (uint32_t)wid : (uint32_t)wid & 3u;
  const uint32_t qoff = qkib * 1024 + lane * 16;
  // rows: piece i = rows 8i .. 8i+7 of this wave's 32, lane L -> row L / 8, physical slot L % 8
  uint32_t xoff[4];
  const char *xbase = nullptr;  // first row of the DMA cursor's block tile (wave-uniform)
  auto tile_row0 = [&](uint32_t k) {
    const uint32_t tile = blockIdx.x + k * gridDim.x;
    return (DENSE ? tile * a.sample_stride : tile) * kRowsB + wid * 32;
  };
  auto set_xsrc = [&](uint32_t k) {
    const uint32_t row0 = tile_row0(k);
    const uint32_t block0 = row0 - wid * 32;
    xbase = reinterpret_cast<const char *>(a.X + (size_t)block0 * a.stride);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t xr = (uint32_t)i * 8 + (lane >> 3);
      uint32_t grow = row0 + xr;
      grow = grow < a.n_total ? grow : a.n_total - 1;  // clamped for the load, masked in the epilogue
      xoff[i] = ((grow - block0) * (uint32_t)a.stride + xslot(xr, lane & 7) * 4) * 4;
    }
  };
  auto dma_chunk = [&](uint32_t c, int stage) {
    if (!VT_DBG(a, 16u)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16<true>(xlds0 + stage * kXStageBytes + i * 1024, xbase + (size_t)c * 128, xoff[i]);
    }
    if (!VT_DBG(a, 4u)) {
#pragma unroll
      for (int i = 0; i < q_pieces(QT); ++i)
        dma16<false>(lds0 + stage * kQStageBytes + (qkib + i) * 1024, qimg + (size_t)c * kQStageBytes + i * 1024, qoff);
    }
  };
  // DMA cursor; past the end of this block's sequence it stays on the last chunk, so the loop
  // body has no branches and the vmcnt arithmetic never changes
  uint32_t dk = 0, dc = 0;
  auto dma_advance = [&]() {
    if (dc + 1 == nchunk && dk + 1 == my_tiles) return;
    dc += 1;
    if (dc == nchunk) {
      dc = 0;
      dk += 1;
      set_xsrc(dk);
    }
  };

  set_xsrc(0);
  dma_chunk(0, 0);
  dma_advance();
  dma_chunk(dc, 1);
  dma_advance();

  // fragment addresses of this lane inside a stage (the swizzle depends on r only)
  uint32_t xfrag[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xfrag[j] = xlds0 - lds0 + r * 128 + xslot(r, 4 * h + j) * 16;
  const uint32_t qfrag = lane * 16;

  int stage = 0;
  for (uint32_t k = 0; k < my_tiles; ++k) {
    f32x16 acc[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (uint32_t c = 0; c < nchunk; ++c) {
      stage = __builtin_amdgcn_readfirstlane(stage);
      const int stage_a = stage == 0 ? kStages - 1 : stage - 1;  // chunk m + 2 goes where chunk m - 1 was read
      // my pieces of chunk m have landed (chunk m + 1's stay in flight), my reads of chunk m - 1
      // are done; behind the barrier that holds for every wave of the block
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kPiecesPerChunk) : "memory");
      if (!VT_DBG(a, 2u)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      dma_chunk(__builtin_amdgcn_readfirstlane(dc), stage_a);
      dma_advance();
      if (VT_DBG(a, 1u)) {
        stage = stage == kStages - 1 ? 0 : stage + 1;
        continue;
      }
      // All 20 fragment reads of the chunk go out before its first MFMA.  (Measured, compute
      // alone on 6 M rows: reads a step or two ahead of their MFMAs 2.04 ms, all up front 2.07,
      // reads of the next step under the MFMAs of this one, across the barrier too, 2.06 -- the
      // fragment traffic is not what the time goes to: 1.3 PFLOP/s is what an LDS-fed bf16 MFMA
      // loop on random data sustains on this part, the clock drops to ~1.6 GHz under it.)
      const unsigned char *xs = lds + stage * kXStageBytes;
      const unsigned char *qs = lds + stage * kQStageBytes + qfrag;
      bf16x8 qv[2][QT];
      const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xs + xfrag[0]);
      const f32x4 x1 = *reinterpret_cast<const f32x4 *>(xs + xfrag[1]);
#pragma unroll
      for (int t = 0; t < QT; ++t) qv[0][t] = *reinterpret_cast<const bf16x8 *>(qs + (t * 2) * 1024);
      const f32x4 x2 = *reinterpret_cast<const f32x4 *>(xs + xfrag[2]);
      const f32x4 x3 = *reinterpret_cast<const f32x4 *>(xs + xfrag[3]);
#pragma unroll
      for (int t = 0; t < QT; ++t) qv[1][t] = *reinterpret_cast<const bf16x8 *>(qs + (t * 2 + 1) * 1024);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 xb0 = pack8(x0, x1), xb1 = pack8(x2, x3);
#pragma unroll
      for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb0, qv[0][t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb1, qv[1][t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      stage = stage == kStages - 1 ? 0 : stage + 1;
    }

    tile_epilogue<DENSE, QT>(a, acc, tau, tile_row0(k), (blockIdx.x + k * gridDim.x) * kRowsB + wid * 32, r, h);
  }
}


}  // namespace

uint32_t batch_bf16_rows_per_block() { return kRowsB; }
size_t batch_bf16_image_bytes(uint32_t ld) { return (size_t)(ld / 32) * q_stage_bytes(8); }
uint32_t batch_bf16_pad(uint32_t nq) { return nq <= 64 ? 64u : nq <= 128 ? 128u : 256u; }

hipError_t launch_batch_q_image(const float *Q, uint32_t ld, uint32_t nq_pad, void *image, hipStream_t s) {
  if (nq_pad != 64 && nq_pad != 128 && nq_pad != 256) return hipErrorInvalidValue;
  const uint32_t nchunk = ld / 32, qt = nq_pad / 32;
  const uint32_t slots = nchunk * qt * 2 * 64;
  hipLaunchKernelGGL(q_image_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, Q, ld, nchunk, qt,
                     reinterpret_cast<bf16x8 *>(image));
  return hipGetLastError();
}

template <class K>
hipError_t launch_one(K kern, size_t lds_bytes, const BatchScoreArgs &a, uint32_t blocks, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesB * kWave), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_batch_scores_bf16(const BatchScoreArgs &a0, bool dense, uint32_t blocks, hipStream_t s) {
  BatchScoreArgs a = a0;
#ifndef VT_BATCH_TIMING_EXPERIMENTS
  a.debug = 0u;
#endif
  if (a.ld % 32 != 0 || (a.nq_pad != 256 && a.nq_pad != 128 && a.nq_pad != 64) || a.Qimage == nullptr) return hipErrorInvalidValue;
  const int qt = (int)a.nq_pad / 32;
  const size_t lds_bytes = (size_t)kStages * (q_stage_bytes(qt) + kXStageBytes);
  if (qt == 8)
    return dense ? launch_one(bf16_scores_kernel<true, 8>, lds_bytes, a, blocks, s)
                 : launch_one(bf16_scores_kernel<false, 8>, lds_bytes, a, blocks, s);
  if (qt == 4)
    return dense ? launch_one(bf16_scores_kernel<true, 4>, lds_bytes, a, blocks, s)
                 : launch_one(bf16_scores_kernel<false, 4>, lds_bytes, a, blocks, s);
  return dense ? launch_one(bf16_scores_kernel<true, 2>, lds_bytes, a, blocks, s)
               : launch_one(bf16_scores_kernel<false, 2>, lds_bytes, a, blocks, s);
}

}  // namespace vt
