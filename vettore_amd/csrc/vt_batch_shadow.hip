// vt_batch_shadow.hip -- K2s: query batches nominated on the BF16 matrix cores from a bf16
// SHADOW of the rows (gfx950).
//
// K2b (vt_batch_bf16.hip) streams the f32 rows and rounds them to bf16 in registers: twice the
// bytes the pass needs, a conversion per element, and twice the LDS traffic (DMA
// alone 6.7 TB/s, MFMA alone 3.2 ms per 10 M rows, together 5.3 ms).  Here the rounding has been
// done once, when the rows arrived: beside the f32 slab the shard keeps -- when the card has
// room -- an image of the same rows in bf16, already in the order the matrix cores take their
// operands in (Shard::dShadow, built by shadow_build_kernel, patched per mutated row like the
// sign bits).  The pass reads N * ld * 2 bytes instead of N * ld * 4, converts nothing, and both
// operands reach LDS as whole fragments.  Same rounding (v_cvt_pk_bf16_f32, round to nearest
// even) => the same nominations as K2b's, the same bound in batch_group; nothing this kernel
// computes is ever returned -- the exact K1 arithmetic re-scores (flat.rs:96-124 run B times).
//
// Image layout (vt_device.h shadow_index): [row / 16][col / 32][(col / 8) % 4][row % 16][col % 8]
// -- 1 KiB per (16 rows, 32 columns) = one A operand of v_mfma_f32_16x16x32_bf16 with lane
// 16 g + r holding row r, k = 8 g .. 8 g + 7 (cdna_hip_programming.md section 3).  The query image
// is the same thing with queries for rows, chunk-major ([col / 32][query / 16]...).
//
// Work split: a block of 8 waves owns 256 rows x nq_pad queries; wave (rg, qh) = (w >> 1, w & 1)
// owns rows 64 rg .. 64 rg + 63 against half of the queries: 4 x QTW accumulator tiles of 16 x 16
// (QTW = 8 / 4 / 2 for 256 / 128 / 64 query columns; 128 accumulator registers at 256).  Per
// 32-column chunk 16 KiB of rows and nq_pad * 64 B of queries arrive in LDS by LDS-DMA (whole
// 1-KiB fragments: a piece is one wave instruction, its source one contiguous KiB), S stages
// deep; a wave reads 4 + QTW fragments (ds_read_b128, lane-linear: conflict-free) for 4 * QTW
// MFMAs -- 12 reads per 32 MFMAs where K2b needs 20 reads and 8 conversions per 16 (32x32x16).
// The chunk sequence runs through all of the block's tiles; fragments of chunk m + 1 are read
// under the MFMAs of chunk m (their registers rotate in place), so right behind the barrier both
// waves of a SIMD have matrix work at hand.  One counted vmcnt + one raw s_barrier per chunk.
#include "vt_common.cuh"

#include <algorithm>
#include <cstdlib>

// Timing experiments (tools/k2s_probe.hip): only with -DVT_BATCH_TIMING_EXPERIMENTS; results are
// garbage with any bit set.  1: no MFMAs, 2: no barrier, 4: no query DMA, 8: no candidate
// append, 16: no row DMA, 32: no fragment reads.
// (the switches are a template argument -- DBG -- so that a mode keeps the product kernel's schedule, minus
// what it leaves out: a run-time test in the hot loop would cut the MFMA cluster into basic blocks)
#define VT_SDBG(a, bit) ((DBG & (bit)) != 0)

namespace vt {

using namespace dev;

namespace {

typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));

constexpr int kWavesS = 8;               // waves per block: 4 row groups x 2 query halves
constexpr int kRowsS = 256;              // rows per block tile
constexpr int kRT = 4;                   // 16-row accumulator tiles per wave
constexpr uint32_t kAStage = 16 * 1024;  // a chunk of a block tile: 256 rows x 32 columns x 2 B
constexpr uint32_t b_stage(int qtw) { return (uint32_t)qtw * 2 * 1024; }  // 2 halves x QTW tiles x 1 KiB
constexpr int b_pieces(int qtw) { return qtw == 8 ? 2 : 1; }              // per wave and chunk (QTW = 2: waves w and w + 4 copy the same KiB)

__device__ __forceinline__ bf16x8s pack8s(f32x4 lo, f32x4 hi) {
  const bf16x2s p0 = __builtin_convertvector((f32x2s){lo[0], lo[1]}, bf16x2s);
  const bf16x2s p1 = __builtin_convertvector((f32x2s){lo[2], lo[3]}, bf16x2s);
  const bf16x2s p2 = __builtin_convertvector((f32x2s){hi[0], hi[1]}, bf16x2s);
  const bf16x2s p3 = __builtin_convertvector((f32x2s){hi[2], hi[3]}, bf16x2s);
  bf16x8s o;
  o[0] = p0[0]; o[1] = p0[1]; o[2] = p1[0]; o[3] = p1[1];
  o[4] = p2[0]; o[5] = p2[1]; o[6] = p3[0]; o[7] = p3[1];
  return o;
}

// ---- the image: built once, patched per mutated row ------------------------------------------
// One wave per (16-row tile, 32-column chunk): lane 16 g + r reads 8 floats of row r (the four
// lanes of a row 128 contiguous bytes), writes its 16 bytes of the fragment.  Rows >= rows_src do
// not exist in the slab (the image is padded to whole block tiles): zeros.
__global__ __launch_bounds__(256) void shadow_build_kernel(const float *__restrict__ X, size_t stride, uint32_t rows_src,
                                                           uint32_t rows_img, uint32_t ld, bf16x8s *__restrict__ img) {
  const uint32_t nchunk = ld >> 5;
  const size_t units = (size_t)(rows_img >> 4) * nchunk;
  const int lane = threadIdx.x & 63;
  const uint32_t r = lane & 15, g = lane >> 4;
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t u = wave0; u < units; u += nwaves) {
    const uint32_t t16 = (uint32_t)(u / nchunk), c = (uint32_t)(u % nchunk);
    const uint32_t row = t16 * 16 + r;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
    if (row < rows_src) {
      const float *src = X + (size_t)row * stride + c * 32 + g * 8;
      lo = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src));
      hi = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(src + 4));
    }
    img[u * 64 + lane] = pack8s(lo, hi);
  }
}

// The rows of `list`: one wave per row, lane p of it the row's 16-byte pieces p, p + 64, ...
__global__ __launch_bounds__(256) void shadow_rows_kernel(const float *__restrict__ X, size_t stride, const uint32_t *__restrict__ list,
                                                          uint32_t count, uint32_t ld, __bf16 *__restrict__ img) {
  const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w >= count) return;
  const int lane = threadIdx.x & 63;
  const uint32_t row = list[w];
  const float *src = X + (size_t)row * stride;
  for (uint32_t p = lane; p < ld / 8; p += 64) {
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(src + 8 * p), hi = *reinterpret_cast<const f32x4 *>(src + 8 * p + 4);
    *reinterpret_cast<bf16x8s *>(img + shadow_index(row, 8 * p, ld)) = pack8s(lo, hi);
  }
}

// The queries in the same fragment order, chunk-major: image[c][query / 16][g][query % 16][e] =
// bf16(Q[query][32 c + 8 g + e]) -- a chunk's B operand is one run of nq_pad * 64 bytes.
__global__ __launch_bounds__(256) void q_image16_kernel(const float *__restrict__ Q, uint32_t ld, uint32_t nchunk, uint32_t nqt,
                                                        bf16x8s *__restrict__ image) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-B fragment slot each
  if (i >= nchunk * nqt * 64) return;
  const uint32_t lane = i & 63, qt = (i >> 6) % nqt, c = (i >> 6) / nqt;
  const uint32_t r = lane & 15, g = lane >> 4;
  const float *src = Q + (size_t)(16 * qt + r) * ld + 32 * c + 8 * g;
  image[i] = pack8s(*reinterpret_cast<const f32x4 *>(src), *reinterpret_cast<const f32x4 *>(src + 4));
}

// ---- the pass ---------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void dma16s(uint32_t lds_addr, const void *base, uint32_t lane_off) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" : : "s"(sl), "v"(lane_off), "s"(sb) : "memory");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(sl), "v"(lane_off), "s"(sb) : "memory");
}

// Cold path of the epilogue: one of this lane's 4 scores of a 16 x 16 tile reaches tau.
__device__ __forceinline__ void append_candidates16(const BatchScoreArgs &a, f32x4 v, float tau, uint32_t qcol, uint32_t row0) {
  asm volatile("" : "+v"(qcol));  // (no list address arithmetic hoisted out of the tile loop: it would cost registers all the time)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t row = row0 + i;
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

// A wave's finished 64-row x (16 QTW)-query tile (C layout of 16x16x32: column = lane & 15 = query,
// row = 4 (lane >> 4) + register).  MODE 0 (the pass): scores that reach the query's threshold are appended.
// MODE 1 (dense): the whole sample matrix is written (tools/k2s_probe compares it with K2b's).  MODE 2 (r05, what the
// sample pass runs): per query only the LARGEST score of the wave's 64 rows -- sample[q][4 tile + rg]; the threshold
// is then taken from those maxima (sample_tau_groups_kernel), 4 KB per query instead of 256.
// grow0 / srow0: the wave's first row in the index / the sample; gslot: the wave's slot in the maxima.
enum { kModePass = 0, kModeDense = 1, kModeMaxima = 2 };
template <int MODE, int QTW, unsigned DBG>
__device__ __forceinline__ void tile_epilogue16(const BatchScoreArgs &a, f32x4 (&acc)[kRT][QTW], const float (&tau)[QTW],
                                                uint32_t grow0, uint32_t srow0, uint32_t gslot, uint32_t qbase, int lane) {
  const int c16 = lane & 15, g = lane >> 4;
  float gmax[QTW];
  if (MODE == kModeMaxima) {
#pragma unroll
    for (int j = 0; j < QTW; ++j) gmax[j] = -INFINITY;
  }
#pragma unroll
  for (int i = 0; i < kRT; ++i) {
    // L2 family: the 16 row norms of this sub-tile through the scalar cache (a vector load would sit in
    // the DMA's in-order queue, and waiting for it would drain the ring once per tile); the norm
    // column is as long as the slab's row capacity, a multiple of 32: a sub-tile that starts below
    // n_total ends inside it.
    float xn[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.xnorm2) {
      typedef const __attribute__((address_space(4))) float *cfloat_p;
      const uint32_t g0 = __builtin_amdgcn_readfirstlane(grow0 + 16 * i);
      if (g0 < a.n_total) {
        cfloat_p xc = (cfloat_p)(uintptr_t)(a.xnorm2 + g0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float s0 = xc[v], s1 = xc[4 + v], s2 = xc[8 + v], s3 = xc[12 + v];
          xn[v] = g < 2 ? (g == 0 ? s0 : s1) : (g == 2 ? s2 : s3);
        }
      }
    }
    const uint32_t row0 = grow0 + 16 * i + 4 * g;
#pragma unroll
    for (int j = 0; j < QTW; ++j) {
      f32x4 v = acc[i][j];
      if (a.xnorm2) {
        // L2 family: rank by s = 2 q.x - |x|^2 (larger s <=> smaller |q - x|^2)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 2.0f * v[e] - xn[e];
      }
      uint32_t qcol = qbase + 16 * j + c16;
      if (MODE == kModeDense) {
        asm volatile("" : "+v"(qcol));
#pragma unroll
        for (int e = 0; e < 4; ++e)
          a.sample[(size_t)qcol * a.sample_rows + srow0 + 16 * i + 4 * g + e] = row0 + e < a.n_total ? v[e] : -INFINITY;
      } else if (MODE == kModeMaxima) {
#pragma unroll
        for (int e = 0; e < 4; ++e) gmax[j] = fmaxf(gmax[j], row0 + e < a.n_total ? v[e] : -INFINITY);
      } else {
        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (mx >= tau[j] && !VT_SDBG(a, 8u)) append_candidates16(a, v, tau[j], qcol, row0);
      }
    }
  }
  if (MODE == kModeMaxima) {
    // the four lanes that hold a query's rows (c16, c16 + 16, + 32, + 48) meet; lane group 0 files the maximum
#pragma unroll
    for (int j = 0; j < QTW; ++j) {
      float m = gmax[j];
      m = fmaxf(m, __shfl_xor(m, 16, kWave));
      m = fmaxf(m, __shfl_xor(m, 32, kWave));
      if (g == 0) a.sample[(size_t)(qbase + 16 * j + c16) * a.sample_rows + gslot] = m;
    }
  }
}

template <int MODE, int QTW, int S, unsigned DBG>
__global__ __launch_bounds__(kWavesS *kWave, 1) void shadow_scores_kernel(const BatchScoreArgs a) {
  constexpr bool DENSE = MODE != kModePass;  // (the sample's tiles: every sample_stride-th, no thresholds)
  constexpr uint32_t kBStage = b_stage(QTW);
  constexpr uint32_t kStage = kAStage + kBStage;
  constexpr int kPieces = 2 + b_pieces(QTW);  // per wave and chunk
  extern __shared__ __align__(16) unsigned char lds[];  // [S] x {rows 16 KiB, queries}
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wid >> 1, qh = wid & 1;
  const uint32_t nchunk = a.ld / 32;
  const uint32_t ntiles = (a.n + kRowsS - 1) / kRowsS;
  if (blockIdx.x >= ntiles) return;
  const uint32_t my_tiles = (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const uint32_t qbase = (uint32_t)qh * QTW * 16;

  float tau[QTW];
#pragma unroll
  for (int j = 0; j < QTW; ++j) tau[j] = DENSE ? 0.f : a.tau[qbase + 16 * j + (lane & 15)];
  // The compiler counts these loads in vmcnt and never sees them retire (the DMA below is inline asm, its waits are
  // hand-placed): left alone it puts s_waitcnt vmcnt(7), (6), ... (0) in front of every tile's threshold compares -- the
  // last of them waits until NO piece of the DMA ring is in flight, once per tile.  Each value is therefore taken
  // through an asm operand right here: the wait happens once, before the first piece is issued.  (Measured: nothing --
  // 3.59-3.62 ms either way at 10 M x 768 x 256; the ring is full again before the pass misses it.  Kept for the ISA.)
  if (!DENSE) {
#pragma unroll
    for (int j = 0; j < QTW; ++j) asm volatile("" : "+v"(tau[j]));
  }

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  // DMA: this wave's pieces of a chunk.  Rows: 16-row tiles 2 w and 2 w + 1 of the block tile (their
  // KiB of chunk c sits nchunk KiB apart in the image); queries: KiB 2 w, 2 w + 1 (QTW = 8), w (4),
  // w mod 4 (2) of the chunk's run.
  const char *const ximg = reinterpret_cast<const char *>(a.Xshadow);
  const char *const qimg = reinterpret_cast<const char *>(a.Qimage);
  const uint32_t aoff0 = (uint32_t)(2 * wid) * nchunk * 1024u + (uint32_t)lane * 16u, aoff1 = aoff0 + nchunk * 1024u;
  const uint32_t qkib = QTW == 8 ? (uint32_t)wid * 2 : QTW == 4 ? (uint32_t)wid : (uint32_t)wid & 3u;
  const uint32_t qoff = qkib * 1024 + (uint32_t)lane * 16u;
  auto tile_index = [&](uint32_t k) {
    const uint32_t tile = blockIdx.x + k * gridDim.x;
    return DENSE ? tile * a.sample_stride : tile;
  };
  const char *xbase = nullptr;  // the DMA cursor's block tile in the image (wave-uniform)
  auto set_xsrc = [&](uint32_t k) { xbase = ximg + (size_t)tile_index(k) * 16 * nchunk * 1024; };
  auto dma_chunk = [&](uint32_t c, uint32_t stage) {
    const uint32_t sb = lds0 + stage * kStage;
    if (!VT_SDBG(a, 16u)) {
      dma16s<true>(sb + (uint32_t)(2 * wid) * 1024, xbase + (size_t)c * 1024, aoff0);
      dma16s<true>(sb + (uint32_t)(2 * wid + 1) * 1024, xbase + (size_t)c * 1024, aoff1);
    }
    if (!VT_SDBG(a, 4u)) {
#pragma unroll
      for (int i = 0; i < b_pieces(QTW); ++i)
        dma16s<false>(sb + kAStage + (qkib + i) * 1024, qimg + (size_t)c * kBStage + i * 1024, qoff);
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
#pragma unroll
  for (int s = 0; s < S; ++s) {
    dma_chunk(dc, s);
    dma_advance();
  }

  // this lane's fragment addresses inside a stage
  const uint32_t afrag = (uint32_t)(rg * kRT) * 1024 + (uint32_t)lane * 16;
  const uint32_t bfrag = kAStage + (uint32_t)(qh * QTW) * 1024 + (uint32_t)lane * 16;
  bf16x8s af[kRT], bq[QTW];
  auto read_a = [&](uint32_t stage, int i) {
    return *reinterpret_cast<const bf16x8s *>(lds + stage * kStage + afrag + i * 1024);
  };
  auto read_b = [&](uint32_t stage, int j) {
    return *reinterpret_cast<const bf16x8s *>(lds + stage * kStage + bfrag + j * 1024);
  };

  // chunk 0 has landed (the other S - 1 stay in flight) -- for every wave behind the barrier
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * kPieces) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < kRT; ++i) af[i] = read_a(0, i);
#pragma unroll
  for (int j = 0; j < QTW; ++j) bq[j] = read_b(0, j);

  uint32_t stage = 0;  // stage of the chunk whose fragments are in registers
  for (uint32_t k = 0; k < my_tiles; ++k) {
    f32x4 acc[kRT][QTW];
#pragma unroll
    for (int i = 0; i < kRT; ++i)
#pragma unroll
      for (int j = 0; j < QTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (uint32_t c = 0; c < nchunk; ++c) {
      stage = __builtin_amdgcn_readfirstlane(stage);
      const uint32_t stage_n = stage + 1 == (uint32_t)S ? 0u : stage + 1;  // next chunk's stage: read from under this chunk's MFMAs
      // My pieces of the next chunk have landed (S - 2 chunks stay in flight) and my reads of this
      // chunk's fragments are done; behind the barrier that holds for every wave of the block: the
      // next chunk may be read, this chunk's stage may be refilled.
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((S - 2) * kPieces) : "memory");
      if (!VT_SDBG(a, 2u)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      dma_chunk(__builtin_amdgcn_readfirstlane(dc), stage);
      dma_advance();
      __builtin_amdgcn_sched_barrier(0);
      if (!VT_SDBG(a, 1u)) {
        // 4 MFMAs per query tile, then that tile's fragment of the NEXT chunk into the registers
        // they have just been read from; the next chunk's row fragments go out first and land in
        // registers of their own (every MFMA of the cluster reads the current ones).
        // (Where the chunk's four DMA pieces are issued was measured three ways, 6 M rows x 768,
        // 256 columns: all of them right here behind the barrier 2.14-2.18 ms; one behind every other
        // MFMA group 2.14; every wave's four in a slot of its own of the cluster, so that the CU's
        // address unit never sees two waves at once, 2.27 -- the memory system likes its requests in
        // bursts.)
        bf16x8s an[kRT];
        if (!VT_SDBG(a, 32u)) {
#pragma unroll
          for (int i = 0; i < kRT; ++i) an[i] = read_a(stage_n, i);
        }
#pragma unroll
        for (int j = 0; j < QTW; ++j) {
#pragma unroll
          for (int i = 0; i < kRT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bq[j], acc[i][j], 0, 0, 0);
          if (!VT_SDBG(a, 32u)) bq[j] = read_b(stage_n, j);
        }
        if (!VT_SDBG(a, 32u)) {
#pragma unroll
          for (int i = 0; i < kRT; ++i) af[i] = an[i];
        }
        // the order above is the order wanted: left alone the compiler issues all 32 MFMAs first and
        // the 12 reads behind them -- which the next iteration's lgkmcnt(0) then waits for with the
        // matrix pipe idle
        if (!VT_SDBG(a, 32u)) {
          __builtin_amdgcn_sched_group_barrier(0x100, kRT, 0);
#pragma unroll
          for (int j = 0; j < QTW; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, kRT, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      stage = stage_n;
    }

    const uint32_t tile = tile_index(k);
    tile_epilogue16<MODE, QTW, DBG>(a, acc, tau, tile * kRowsS + rg * 64, (blockIdx.x + k * gridDim.x) * kRowsS + rg * 64,
                                    (blockIdx.x + k * gridDim.x) * 4 + rg, qbase, lane);
  }
}

template <class K>
hipError_t launch_one_s(K kern, size_t lds_bytes, const BatchScoreArgs &a, uint32_t blocks, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesS * kWave), lds_bytes, s, a);
  return hipGetLastError();
}

template <int QTW, int S>
hipError_t launch_qs(const BatchScoreArgs &a, int mode, uint32_t blocks, hipStream_t s) {
  const bool dense = mode != kModePass;
  const size_t lds_bytes = (size_t)S * (kAStage + b_stage(QTW));
#ifdef VT_BATCH_TIMING_EXPERIMENTS
  if (a.debug && !dense && QTW == 8 && S == 5) {  // (the modes tools/k2s_probe.hip asks for: 256 columns, five stages)
    switch (a.debug) {
#define VT_MODE(M) case M: return launch_one_s(shadow_scores_kernel<kModePass, 8, 5, M>, lds_bytes, a, blocks, s);
      VT_MODE(2u) VT_MODE(8u) VT_MODE(10u) VT_MODE(1u) VT_MODE(3u) VT_MODE(5u) VT_MODE(16u) VT_MODE(20u) VT_MODE(22u) VT_MODE(48u) VT_MODE(52u) VT_MODE(54u)
#undef VT_MODE
      default: return hipErrorInvalidValue;
    }
  }
#endif
  if (mode == kModeMaxima) return launch_one_s(shadow_scores_kernel<kModeMaxima, QTW, S, 0u>, lds_bytes, a, blocks, s);
  return dense ? launch_one_s(shadow_scores_kernel<kModeDense, QTW, S, 0u>, lds_bytes, a, blocks, s)
               : launch_one_s(shadow_scores_kernel<kModePass, QTW, S, 0u>, lds_bytes, a, blocks, s);
}

// stages of the LDS ring: 5 x 32 KiB is all of a CU's LDS at 256 query columns
int shadow_stages() {
  const long v = env::get(env::SHADOW_STAGES);  // (VT_SHADOW_STAGES=4|5)
  return v == 4 ? 4 : 5;
}

}  // namespace

uint32_t batch_shadow_rows_per_block() { return kRowsS; }
size_t batch_shadow_image_bytes(uint32_t ld) { return (size_t)(ld / 32) * b_stage(8); }
size_t shadow_elems(uint32_t rows, uint32_t ld) { return (size_t)round_up(rows, (uint32_t)kRowsS) * ld; }

hipError_t launch_shadow_build(const float *X, size_t stride, uint32_t rows_src, uint32_t rows_img, uint32_t ld, void *img,
                               hipStream_t s) {
  if (ld % 64 != 0 || rows_img % kRowsS != 0) return hipErrorInvalidValue;
  if (rows_img == 0) return hipSuccess;
  const size_t units = (size_t)(rows_img / 16) * (ld / 32);
  const uint32_t blocks = (uint32_t)std::min<size_t>((units + 3) / 4, (size_t)256 * 32);
  hipLaunchKernelGGL(shadow_build_kernel, dim3(blocks), dim3(256), 0, s, X, stride, rows_src, rows_img, ld,
                     reinterpret_cast<bf16x8s *>(img));
  return hipGetLastError();
}

hipError_t launch_shadow_rows(const float *X, size_t stride, const uint32_t *list, uint32_t count, uint32_t ld, void *img,
                              hipStream_t s) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(shadow_rows_kernel, dim3((count + 3) / 4), dim3(256), 0, s, X, stride, list, count, ld,
                     reinterpret_cast<__bf16 *>(img));
  return hipGetLastError();
}

hipError_t launch_batch_q_image16(const float *Q, uint32_t ld, uint32_t nq_pad, void *image, hipStream_t s) {
  if (nq_pad != 64 && nq_pad != 128 && nq_pad != 256) return hipErrorInvalidValue;
  const uint32_t nchunk = ld / 32, nqt = nq_pad / 16;
  const uint32_t slots = nchunk * nqt * 64;
  hipLaunchKernelGGL(q_image16_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, Q, ld, nchunk, nqt,
                     reinterpret_cast<bf16x8s *>(image));
  return hipGetLastError();
}

static hipError_t launch_shadow_mode(const BatchScoreArgs &a0, int mode, uint32_t blocks, hipStream_t s);
hipError_t launch_batch_scores_shadow(const BatchScoreArgs &a0, bool dense, uint32_t blocks, hipStream_t s) {
  return launch_shadow_mode(a0, dense ? kModeDense : kModePass, blocks, s);
}
// The sample pass as the library runs it since r05: a.sample is [nq_pad][a.sample_rows] with a.sample_rows = 4 x the
// launch's tiles -- the best score of every 64-row group of the sample (batch_shadow_sample_groups) -- and
// launch_sample_tau_groups takes the thresholds from it.
hipError_t launch_batch_sample_maxima_shadow(const BatchScoreArgs &a, uint32_t blocks, hipStream_t s) {
  return launch_shadow_mode(a, kModeMaxima, blocks, s);
}
uint32_t batch_shadow_sample_groups(uint32_t sample_tiles) { return sample_tiles * (kRowsS / 64); }

static hipError_t launch_shadow_mode(const BatchScoreArgs &a0, int dense, uint32_t blocks, hipStream_t s) {
  BatchScoreArgs a = a0;
#ifndef VT_BATCH_TIMING_EXPERIMENTS
  a.debug = 0u;
#endif
  if (a.ld % 64 != 0 || (a.nq_pad != 256 && a.nq_pad != 128 && a.nq_pad != 64) || a.Qimage == nullptr || a.Xshadow == nullptr)
    return hipErrorInvalidValue;
  // (five stages are all of a CU's LDS at 256 columns; four leave 32 KB -- room for the small kernels of the groups
  // around this pass, batch_ready)
  const bool five = (a.stages == 4 || a.stages == 5 ? (int)a.stages : shadow_stages()) == 5;
  if (a.nq_pad == 256) return five ? launch_qs<8, 5>(a, dense, blocks, s) : launch_qs<8, 4>(a, dense, blocks, s);
  if (a.nq_pad == 128) return five ? launch_qs<4, 5>(a, dense, blocks, s) : launch_qs<4, 4>(a, dense, blocks, s);
  return five ? launch_qs<2, 5>(a, dense, blocks, s) : launch_qs<2, 4>(a, dense, blocks, s);
}

}  // namespace vt
