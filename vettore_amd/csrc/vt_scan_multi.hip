// vt_scan_multi.hip -- K1m: the flat scan for SEVERAL queries in one sweep of the corpus,
// exact reference arithmetic, any metric (gfx950).
//
// Replaces NQ runs of the loop of FlatIndex::search (native/vettore/src/flat.rs:104-118).
// Every (query, row) value is computed exactly as K1 computes it (vt_scan.cuh): eight
// separately rounded products per 8-float chunk, one horizontal add in the selected lane
// order of wide::f32x8::reduce_add, `acc += chunk_sum` sequentially over the chunks, then
// the scalar tail -- so the hits of query q are bit-identical to its own flat_search.
//
// What changes is the shape of the walk.  The scan uses a few per cent of the vector ALUs,
// so one pass over HBM can carry several queries:
//   * a wave owns a tile of TR rows and walks it in column panels of 256 floats: ONE 1-KiB
//     wave load per row and panel (16 B per lane, nontemporal), kU loads in flight in a
//     register ring that runs ahead across panel and tile boundaries;
//   * within a panel a lane always sits on the same four columns, so its fragment of each
//     query is loop-invariant: NQ x 4 registers, fetched (from L1/L2: every wave of the chip
//     reads the same few KB) one panel ahead.  No query tile in LDS, no per-load LDS read;
//   * each load is multiplied into NQ = 8 chunk sums (lane pair = one chunk, DPP add), filed
//     in NQ small LDS panels S_q[TR = 8 rows][32 chunks] (the even lane of a pair stores the
//     even queries' sums, the odd lane the odd ones': four stores per load);
//   * after the 8 loads of a panel the wave re-reads the panels as 64 (query, row) pairs:
//     lane (q, r) advances the sequential chain of row r under query q by 32 chunks -- all
//     64 lanes busy, one chain each;
//   * after the last panel every lane owns one finished (query, row) value: finiteness / f64
//     recovery, rank key, then 8 offers into 8 wave-private top-k buffers (WaveTopK).
// LDS per block: 4 waves x 8 queries x (1.4 KiB panel + 1 KiB candidates) = 77 KB: two
// blocks per CU.  Two groups of 8 loads (16 KiB) are in flight per wave.
//
// Roofline: HBM-bound like K1 -- algorithmic bytes per launch = n * d * 4, whatever NQ is.
#include "vt_scan.cuh"

#include <cstdlib>

namespace vt {
namespace dev {

constexpr uint32_t kMqPanel = 256;  // floats per panel = one 1-KiB wave load per row
constexpr uint32_t kMqSS = 44;      // dwords per LDS panel row: 32 chunk sums, 8 tail products, pad (4 * odd)
// ... of the slim build: 32 chunk sums, no tail slots and no pad -- the 16-byte groups of row r are
// stored at group position (g ^ r) instead, which spreads the chain phase's reads (lane = (query,
// row), all lanes on the same group index) over the banks like the pad does
constexpr uint32_t kMqSSSlim = 32;
constexpr uint32_t kMqTail = 32;    // dword offset of the tail products in a panel row
constexpr int kMqNQ = (int)kMultiMaxQueries;  // queries per sweep
constexpr int kMqTR = kWave / kMqNQ;          // rows per tile: (query, row) pairs fill the wave exactly
constexpr int kMqCap = 64;                    // candidate slots per query and wave (k <= 32, 8 offers at a time)
constexpr int kMqCapSlim = 32;                // ... of the slim build (k <= 16)
constexpr uint32_t kMqSlimMaxK = 16;
// dwords between two queries' panels (the + 8 staggers their banks)
// (slim: + 4, and 8 replay slots: 50 KB per block.  At 54 KB three blocks did NOT become resident
// on a CU, whatever 3 x 54 144 <= 163 840 says: wave-cycle counters showed two, and the statically
// dealt tiles of the third ran as a tail, 8 % slower than the two-block build)
constexpr uint32_t mq_qs(bool slim) { return kMqTR * (slim ? kMqSSSlim : kMqSS) + (slim ? 4 : 8); }
// overflowed (query, row) pairs a wave can set aside for the f64 replay
constexpr uint32_t mq_redo(bool slim) { return slim ? 8u : 64u; }

// Per-element operation on PREPARED operands: for float hamming / jaccard the loaded row and
// the query fragments are first turned into 0/1 indicators (x != 0), after which the element
// is |q - x| like manhattan's (sums of small integers: exact in f32 in any order), jaccard
// adding 4096 per non-zero row coordinate (exact for d < 4096), as K1 does.
// The four elements of a lane's half chunk as two packed pairs (a = elements 0,1; b = 2,3):
// v_pk_mul_f32 / v_pk_add_f32 do two lanes' worth of f32 work per instruction, each half
// rounded on its own exactly like the scalar instruction (-ffp-contract=off: no fusing).
typedef float f32x2 __attribute__((ext_vector_type(2)));
// (q - x on both halves of a pair in one instruction: the compiler packs products and sums but
// turns every form of a packed difference back into two scalar subtractions)
__device__ __forceinline__ f32x2 pk_sub(f32x2 q, f32x2 x) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(q), "v"(x));
  return r;
}
template <int OP>
__device__ __forceinline__ void half_chunk(const f32x4 q, const f32x4 x, const f32x4 x4k, f32x2 &a, f32x2 &b) {
  const f32x2 qa{q.x, q.y}, qb{q.z, q.w}, xa{x.x, x.y}, xb{x.z, x.w};
  if (OP == OP_DOT) {
    a = qa * xa;
    b = qb * xb;
  } else if (OP == OP_L2) {
    const f32x2 ta = pk_sub(qa, xa), tb = pk_sub(qb, xb);
    a = ta * ta;
    b = tb * tb;
  } else if (OP == OP_JAC) {
    const f32x2 ta = pk_sub(qa, xa), tb = pk_sub(qb, xb);
    a = f32x2{fabsf(ta.x), fabsf(ta.y)} + f32x2{x4k.x, x4k.y};
    b = f32x2{fabsf(tb.x), fabsf(tb.y)} + f32x2{x4k.z, x4k.w};
  } else {
    // manhattan / chebyshev / (indicator) hamming: the signed differences; |.| is applied where
    // they are summed (an operand modifier of the scalar add / max: free)
    a = pk_sub(qa, xa);
    b = pk_sub(qb, xb);
  }
}
// wide::f32x8::reduce_add of the chunk held by a lane pair, from the packed halves (see
// chunk_sum in vt_scan.cuh for the four orders); both lanes return the chunk sum.
template <int SUM_OP, int ORDER, bool ABS>
__device__ __forceinline__ float chunk_sum_packed(int order_rt, const f32x2 a, const f32x2 b, int odd) {
  // ABS: the halves hold signed differences whose magnitudes are summed (half_chunk)
  if (ORDER < 0 || ORDER == 1) {
    if (ABS) return chunk_sum<SUM_OP, ORDER>(SUM_OP, order_rt, fabsf(a.x), fabsf(a.y), fabsf(b.x), fabsf(b.y), odd);
    return chunk_sum<SUM_OP, ORDER>(SUM_OP, order_rt, a.x, a.y, b.x, b.y, odd);
  }
  float e;
  if (SUM_OP == OP_LINF) {
    e = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(b.x), fabsf(b.y)));
  } else if (ABS) {
    if (ORDER == 3) e = (fabsf(a.x) + fabsf(b.x)) + (fabsf(a.y) + fabsf(b.y));
    else if (ORDER == 2) e = ((fabsf(a.x) + fabsf(a.y)) + fabsf(b.x)) + fabsf(b.y);
    else e = (fabsf(a.x) + fabsf(a.y)) + (fabsf(b.x) + fabsf(b.y));
  } else if (ORDER == 3) {  // SSE2: ((l0+l2)+(l1+l3)) + ((l4+l6)+(l5+l7)): one packed add, one add
    const f32x2 t = a + b;
    e = t.x + t.y;
  } else if (ORDER == 2) {  // SEQ
    e = ((a.x + a.y) + b.x) + b.y;
  } else {  // PAIR
    e = (a.x + a.y) + (b.x + b.y);
  }
  return comb<SUM_OP>(SUM_OP, e, dpp_xor1(e));
}

template <int OP>
__device__ __forceinline__ f32x4 indicator(f32x4 v) {
  if (OP != OP_HAM && OP != OP_JAC) return v;
  return f32x4{v.x != 0.0f ? 1.0f : 0.0f, v.y != 0.0f ? 1.0f : 0.0f, v.z != 0.0f ? 1.0f : 0.0f, v.w != 0.0f ? 1.0f : 0.0f};
}

// `b` in the odd lanes, `a` in the even ones.  Written as the instruction: left to the
// compiler, a select between two elements of a register array becomes an indexed access to a
// copy of the array in scratch memory -- vector-memory traffic whose waits also drain the
// corpus loads that are in flight behind it.
__device__ __forceinline__ float pick_odd(float a, float b) {
  float r;
  asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(0xAAAAAAAAAAAAAAAAull));
  return r;
}

// ---- loads the compiler does not schedule (FAST path) --------------------------------------
// Vector loads return in order and `s_waitcnt vmcnt(N)` waits for all but the N youngest, so a
// stream of loads that is consumed in issue order never stalls on a younger load -- provided
// every wait carries the exact count.  The compiler's own counting gives up at loop-carried
// and conditional loads (it then waits for vmcnt(0), i.e. for the group that was just
// requested: the pipeline drains once per step).  So the FAST path issues its loads as bare
// instructions and places the counted waits itself; each wait names the registers it
// releases, which is what keeps their uses behind it.  (The compiler's own vector-memory
// instructions elsewhere in the kernel stay correct: more loads in flight than it knows of
// can only make its waits longer.)
__device__ __forceinline__ void issue_load_nt(f32x4 &dst, const float *p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue_load(f32x4 &dst, const float *p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void issue_load_u32(uint32_t &dst, const uint32_t *p) {
  asm volatile("global_load_dword %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_all_but(f32x4 &released) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(released) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_all_but(uint32_t &released) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(released) : "n"(N) : "memory");
}
__device__ __forceinline__ void released_too(f32x4 &v) { asm volatile("" : "+v"(v)); }

// FAST: d % 64 == 0 -- rows carry no padding and no scalar tail, and ORDER is a compile-time
// constant.  The last panel of a row may still be partial (d % 256 != 0): its surplus lanes
// re-read columns of the same row (no extra HBM lines) and file sums the chain never reads.
// Otherwise (ORDER = -1) bounds, tail and lane order are run-time and the loads the compiler's.
// SLIM (FAST, no tail, k <= 16): three blocks per CU instead of two.  What that takes: <= 168
// VGPRs -- ONE set of query fragments instead of two (the next panel's are requested into the
// same registers as soon as the panel's last row has been multiplied: the chain phase covers
// their trip from L1/L2) -- and <= 53 KB of LDS per block: panel rows without tail slots, 32
// candidate slots per (wave, query) instead of 64, 8 replay slots per wave instead of 64.
// PACK (slim builds; 2 or 4): the row's last panel is 256 / PACK floats wide (d % 256 = 128 or 64:
// d = 384, 128, 64, 320 ...), and instead of leaving (PACK - 1) / PACK of every load's lanes to
// re-read columns nobody sums, one load carries that panel of PACK consecutive rows: lane l sits
// on row u * PACK + l / (64 / PACK), columns 4 * (l % (64 / PACK)) ... of the panel -- TR / PACK loads,
// and as many rounds of products, for the step instead of TR.  The group still ISSUES TR loads -- the
// surplus ones fetch the (cache-resident) query rows -- and every load of the loop stays
// unconditional: with groups of two sizes the asm loads sat in two branches, the ring registers
// met in phis, and the compiler resolved those with copies (`v_mov_b64` of ring registers on the
// loop's back edge, in front of the counted waits) of registers whose loads were still in flight:
// garbage scores, timing-dependent.  The same goes for two `s_waitcnt` statements in an if / else
// (the copy of the released register was placed above one of them).  One size, one wait, no phi.
template <int OP, int ORDER, bool FAST, bool TAILED, bool SLIM, int PACK = 1>
__global__ __launch_bounds__(kWavesPerBlock *kWave, SLIM ? 3 : 2) void scan_multi_kernel(const MultiScanArgs a) {
  static_assert(!SLIM || (FAST && !TAILED), "the slim build carries no tail");
  static_assert(PACK == 1 || (SLIM && (PACK == 2 || PACK == 4)), "row packing lives in the slim builds");
  constexpr bool kTail = TAILED || !FAST;  // rows may end in chunks short of a group of four and in a scalar tail
  constexpr int NQ = kMqNQ, TR = kMqTR, CAP = SLIM ? kMqCapSlim : kMqCap;
  constexpr uint32_t kMqSS = SLIM ? kMqSSSlim : dev::kMqSS, kMqQS = mq_qs(SLIM), kMqRedo = mq_redo(SLIM);
  constexpr int SUM_OP = (OP == OP_HAM || OP == OP_JAC) ? OP_L1 : OP;  // how chunk sums combine
  constexpr bool kAbs = OP == OP_L1 || OP == OP_LINF || OP == OP_HAM;     // half_chunk leaves signed differences
  extern __shared__ __align__(16) float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int odd = lane & 1;
  // LDS: per wave NQ panels, then per query the four waves' candidate buffers side by side
  // (merge_block expects wave w's buffer 2 * CAP u64 after wave 0's)
  float *S = lds + wib * (NQ * kMqQS);
  unsigned char *tkbase = reinterpret_cast<unsigned char *>(lds + kWavesPerBlock * (NQ * kMqQS));
  __shared__ uint32_t s_counts[NQ][kWavesPerBlock];
  // (query, row) pairs whose f32 value came out non-finite: re-evaluated in f64 after the sweep
  // (distances.rs:59-67), so that the loop itself holds no vector-memory instruction of the
  // compiler's (its waits would drain the load stream).  More of them than fit: the launch
  // reports kStatusRetry and the host takes the queries one by one.
  __shared__ uint32_t s_redo[kWavesPerBlock][kMqRedo], s_redo_q[kWavesPerBlock][kMqRedo];
  __shared__ uint32_t s_thr_hi[SLIM ? kWavesPerBlock : 1][SLIM ? kWave : 1];  // (slim builds: see the tile's end)
  uint32_t nredo = 0;  // wave-uniform

  const uint32_t ld = a.ld;
  const uint32_t cfull = a.d / 8, tail = kTail ? a.d % 8 : 0u;
  const uint32_t npanel = (ld + kMqPanel - 1) / kMqPanel;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + TR - 1) / TR;
  const uint32_t lane_col = (uint32_t)lane * 4u;
  // packed last panel: this lane's row within a load and its columns within the panel
  // (derived from the lane id where they are used, behind an opaque copy of it: hoisted out of the
  // loop they are two more live registers, and the packed builds sit exactly on the 168 that three
  // blocks per CU allow -- one spilled vector register fails the build, tools/check_scratch.py)
  constexpr uint32_t kPkLanes = kWave / PACK, kPkG = kMqTR / PACK;
  auto pk_position = [&](uint32_t &sub, uint32_t &col) {
    uint32_t l = (uint32_t)lane;
    asm volatile("" : "+v"(l));
    sub = l / kPkLanes;
    col = (l % kPkLanes) * 4u;
  };
  // compute phase: a lane pair owns chunk (lane >> 1) of the loaded row; the even lane files
  // the sums of the even queries, the odd lane those of the odd queries (one store per pair of queries)
  float *Sstore = S + odd * kMqQS + (lane >> 1);
  // chain phase: lane = (query, row) -- 64 sequential chains advance side by side
  const int cq = lane / TR, cr = lane % TR;
  const float *Schain = S + cq * kMqQS + cr * kMqSS;

  WaveTopK<CAP> tk[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) tk[q].init(tkbase + (size_t)(q * kWavesPerBlock + wib) * WaveTopK<CAP>::lds_bytes(), a.k);

  if (wave_global < ntiles) {
    const uint32_t my_tiles = (ntiles - wave_global + total_waves - 1) / total_waves;
    const uint32_t last_tile = wave_global + (my_tiles - 1) * total_waves;
    const uint32_t total_steps = my_tiles * npanel;  // one step = one panel of one tile = TR loads
    // the load stream runs one step ahead of the compute stream
    uint32_t lt = wave_global, lp = 0;
    auto load_group = [&](f32x4 *buf) {
      const uint32_t t = lt < last_tile ? lt : last_tile;  // clamp at the end of the stream
      uint32_t colf = lp * kMqPanel + lane_col;
      if (FAST && colf >= ld) colf %= ld;  // surplus lane of a partial panel: any column of the row will do
      const float *base = a.X + (size_t)t * TR * a.stride + colf;
      // (PACK) the short last panel: PACK rows per load in the first TR / PACK loads; the rest repeat
      // the last of those (lines already on their way: no new HBM traffic) -- one base and one stride
      // are SELECTED, the load instructions are the same eight
      const bool pk = PACK > 1 && lp + 1 == npanel;
      const float *base_sel = base;
      size_t stride_sel = a.stride;
      if constexpr (PACK > 1) {
        uint32_t pk_sub, pk_col;
        pk_position(pk_sub, pk_col);
        const float *base_pk = a.X + ((size_t)t * TR + pk_sub) * a.stride + lp * kMqPanel + pk_col;
        base_sel = pk ? base_pk : base;
        stride_sel = pk ? (size_t)PACK * a.stride : a.stride;
      }
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        if constexpr (PACK > 1) {
          const uint32_t u_eff = (pk && u >= (int)kPkG) ? kPkG - 1 : (uint32_t)u;
          issue_load_nt(buf[u], base_sel + (size_t)u_eff * stride_sel);
        } else if constexpr (FAST) {
          issue_load_nt(buf[u], base + (size_t)u * a.stride);
        } else {
          // (rows up to the slab's capacity exist and are zero; a panel's lanes beyond the row end do not load)
          buf[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (colf < ld) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)u * a.stride));
        }
      }
      lp += 1;
      if (lp == npanel) {
        lp = 0;
        lt += total_waves;
      }
    };
    auto query_fragment = [&](uint32_t panel, f32x4 *qv) {
      uint32_t colf = panel * kMqPanel + lane_col;
      if (PACK > 1 && panel + 1 == npanel) {
        uint32_t pk_sub, pk_col;
        pk_position(pk_sub, pk_col);
        colf = panel * kMqPanel + pk_col;
      }
      if (FAST && colf >= ld) colf %= ld;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if constexpr (FAST) {
          issue_load(qv[q], a.Q + (size_t)q * ld + colf);
        } else {
          qv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (colf < ld) qv[q] = *reinterpret_cast<const f32x4 *>(a.Q + (size_t)q * ld + colf);
        }
      }
    };

    // Two register sets for the query fragments and two groups of TR corpus loads, swapping
    // roles every step: the step loop is unrolled by two, so every register index is static
    // and nothing is copied.  Every vector-memory instruction of the loop is unconditional and
    // issued in the order its data is needed -- loads return in order, so a wait for the oldest
    // never has to wait for a younger one.
    f32x4 qa[NQ], qb[SLIM ? 1 : NQ];
    query_fragment(0, qa);
    f32x4 bufa[TR], bufb[TR];
    load_group(bufa);

    uint32_t t = wave_global, pc = 0;  // tile and panel of the step being consumed
    float acc = 0.0f;
    uint32_t my_thr_hi = 0xFFFFFFFFu;  // score half of the threshold of this lane's query's list (see the tile's end)
    if constexpr (SLIM) s_thr_hi[wib][lane] = 0xFFFFFFFFu;
    (void)my_thr_hi;
    for (uint32_t step = 0; step < total_steps; step += 2) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (step + half >= total_steps) break;
        f32x4 *cur = half ? bufb : bufa;
        f32x4 *nxt = half ? bufa : bufb;
        f32x4 *qcur = (half && !SLIM) ? qb : qa;
        f32x4 *qnext = (half || SLIM) ? qa : qb;
        const uint32_t grow = t * TR + cr;  // the row of this lane's chain
        const bool row_valid = grow < a.n;
        // per step, in this order: the row's id rank (the column covers the slab's capacity), the
        // next panel's query fragments, the next group of corpus loads -- 1 + NQ + TR loads
        uint32_t my_rank = 0;
        if constexpr (FAST) issue_load_u32(my_rank, a.id_rank + grow);
        if constexpr (!SLIM) query_fragment(pc + 1 == npanel ? 0 : pc + 1, qnext);
        load_group(nxt);
        const uint32_t c = pc * (kMqPanel / 8) + ((uint32_t)lane >> 1);  // this lane pair's chunk of the row
        // (PACK) is this step's panel the short one, and the one just requested?
        const uint32_t nxt_panel = pc + 1 == npanel ? 0u : pc + 1;
        const bool cur_pk = PACK > 1 && pc + 1 == npanel;
        const int last_u = cur_pk ? (int)kPkG - 1 : TR - 1;
#pragma unroll
        for (int u = 0; u < TR; ++u) {
          if (PACK > 1 && u > last_u) break;
          if constexpr (SLIM) {
            // in flight, oldest first: cur's group, the fragments (requested at the end of the
            // previous step), this step's rank and group.  One wait for all but the last 1 + TR
            // releases the group and the fragments together.
            if (u == 0) {
#ifdef VT_MULTI_TIMING_EXPERIMENTS
              if (a.dbg & 8u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (8: every load in before a step's first use)
#endif
              wait_all_but<1 + TR>(cur[0]);
#pragma unroll
              for (int q = 0; q < NQ; ++q) released_too(qcur[q]);
            } else {
              released_too(cur[u]);
            }
          } else if constexpr (FAST) {
            // younger than cur[u]: the rest of its group and this step's 1 + NQ + TR loads
            switch (u) {
              case 0: wait_all_but<TR - 1 + 1 + NQ + TR>(cur[0]); break;
              case 1: wait_all_but<TR - 2 + 1 + NQ + TR>(cur[1]); break;
              case 2: wait_all_but<TR - 3 + 1 + NQ + TR>(cur[2]); break;
              case 3: wait_all_but<TR - 4 + 1 + NQ + TR>(cur[3]); break;
              case 4: wait_all_but<TR - 5 + 1 + NQ + TR>(cur[4]); break;
              case 5: wait_all_but<TR - 6 + 1 + NQ + TR>(cur[5]); break;
              case 6: wait_all_but<TR - 7 + 1 + NQ + TR>(cur[6]); break;
              default: wait_all_but<1 + NQ + TR>(cur[7]); break;
            }
            if (u == 0) {  // (the fragments were requested before cur's group: they are in as well)
#pragma unroll
              for (int q = 0; q < NQ; ++q) released_too(qcur[q]);
            }
          }
#ifdef VT_MULTI_TIMING_EXPERIMENTS
          // (wrong results, timing only: `make mqdbg`) 1: no products / sums / stores, 2: no chain phase, 4: no stores
          if (a.dbg & 1u) {
            asm volatile("" ::"v"(cur[u]));
            if constexpr (SLIM && PACK == 1) {
              if (u == TR - 1) query_fragment(nxt_panel, qnext);
            }
            continue;
          }
#endif
          const f32x4 x = indicator<OP>(cur[u]);
          f32x4 x4k = x;
          if (OP == OP_JAC) x4k = f32x4{x.x * 4096.0f, x.y * 4096.0f, x.z * 4096.0f, x.w * 4096.0f};
          // the eight queries' chains are independent: written level by level (all products, then
          // all sums) so that the instruction stream interleaves them instead of walking one
          // dependent chain after the other
          float sum[NQ];
          f32x2 pa[NQ], pb[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q) half_chunk<OP>(indicator<OP>(qcur[q]), x, x4k, pa[q], pb[q]);
          if constexpr (SLIM && PACK == 1) {
            // the panel's last row has been multiplied: the fragments' registers take the next panel's
            if (u == TR - 1) {
#pragma unroll
              for (int q = 0; q < NQ; ++q) asm volatile("" ::"v"(pa[q]), "v"(pb[q]));  // (products first)
              query_fragment(nxt_panel, qnext);
            }
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q) sum[q] = chunk_sum_packed<SUM_OP, ORDER, kAbs>(a.order, pa[q], pb[q], odd);
          if (kTail && tail && c == cfull) {  // tail chunk: the reference adds these products one by one
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              *reinterpret_cast<f32x4 *>(S + q * kMqQS + u * kMqSS + kMqTail + odd * 4) =
                  kAbs ? f32x4{fabsf(pa[q].x), fabsf(pa[q].y), fabsf(pb[q].x), fabsf(pb[q].y)}
                       : f32x4{pa[q].x, pa[q].y, pb[q].x, pb[q].y};
          }
#ifdef VT_MULTI_TIMING_EXPERIMENTS
          if (a.dbg & 4u) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) asm volatile("" ::"v"(sum[q]));
            continue;
          }
#endif
          // (sums of padding chunks land in slots the chain never reads)
#pragma unroll
          for (int qq = 0; qq < NQ; qq += 2) {
            if constexpr (SLIM) {
              // row of this lane's sums and its chunk within the panel; row r keeps 16-byte group g at g ^ r
              uint32_t srow = (uint32_t)u, schunk = (uint32_t)lane >> 1;
              if (cur_pk) {
                uint32_t pk_sub, pk_col;
                pk_position(pk_sub, pk_col);
                srow = (uint32_t)u * PACK + pk_sub;
                schunk = pk_col >> 3;
              }
              (S + odd * kMqQS)[qq * kMqQS + srow * kMqSS + (schunk ^ (srow << 2))] = pick_odd(sum[qq], sum[qq + 1]);
            } else {
              Sstore[qq * kMqQS + u * kMqSS] = pick_odd(sum[qq], sum[qq + 1]);
            }
          }
        }
        // (PACK: the step's rounds end at a run-time count, so the fragments' reload sits behind the
        // loop, at ONE place in the program -- see the note on phis at the top of the kernel; the
        // sums' LDS stores above cannot move below it, so every product has been formed)
        if constexpr (SLIM && PACK > 1) query_fragment(nxt_panel, qnext);
        // The rank was requested first in this step: it has long arrived.  The wait is placed in
        // EVERY step, used or not -- a register whose load is still on its way must not look dead
        // to the register allocator, or the late write lands in whatever was put there since.
        if constexpr (FAST) wait_all_but<NQ + TR>(my_rank);
        // panel complete: every (query, row) chain advances by the panel's chunks
        wave_lds_fence();
#ifdef VT_MULTI_TIMING_EXPERIMENTS
        if (!(a.dbg & 2u))
#endif
        {
          const uint32_t c0 = pc * (kMqPanel / 8);
          float v = pc == 0 ? 0.0f : acc;
          if (FAST && !TAILED) {
            // chunks of this panel: 32, or what is left of the row (a multiple of 8 chunks)
            const uint32_t left = cfull - c0;
            const uint32_t swz = SLIM ? (uint32_t)cr << 2 : 0u;  // (slim: row r keeps group g at g ^ r)
            if (left >= kMqPanel / 8) {
#pragma unroll
              for (uint32_t i = 0; i < kMqPanel / 8; i += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(Schain + (i ^ swz));
                v = comb<SUM_OP>(SUM_OP, v, w.x);
                v = comb<SUM_OP>(SUM_OP, v, w.y);
                v = comb<SUM_OP>(SUM_OP, v, w.z);
                v = comb<SUM_OP>(SUM_OP, v, w.w);
              }
            } else {
              for (uint32_t i = 0; i < left; i += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(Schain + (i ^ swz));
                v = comb<SUM_OP>(SUM_OP, v, w.x);
                v = comb<SUM_OP>(SUM_OP, v, w.y);
                v = comb<SUM_OP>(SUM_OP, v, w.z);
                v = comb<SUM_OP>(SUM_OP, v, w.w);
              }
            }
          } else if (FAST) {
            // chunks of this panel: 32, or what is left of the row (its full chunks, then the
            // scalar tail's products one by one: distances.rs:236-270)
            const uint32_t left = cfull > c0 ? cfull - c0 : 0u;
            if (left >= kMqPanel / 8) {
#pragma unroll
              for (uint32_t i = 0; i < kMqPanel / 8; i += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(Schain + i);
                v = comb<SUM_OP>(SUM_OP, v, w.x);
                v = comb<SUM_OP>(SUM_OP, v, w.y);
                v = comb<SUM_OP>(SUM_OP, v, w.z);
                v = comb<SUM_OP>(SUM_OP, v, w.w);
              }
            } else {
              uint32_t i = 0;
              for (; i + 4 <= left; i += 4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(Schain + i);
                v = comb<SUM_OP>(SUM_OP, v, w.x);
                v = comb<SUM_OP>(SUM_OP, v, w.y);
                v = comb<SUM_OP>(SUM_OP, v, w.z);
                v = comb<SUM_OP>(SUM_OP, v, w.w);
              }
              for (; i < left; ++i) v = comb<SUM_OP>(SUM_OP, v, Schain[i]);
            }
            if (tail && cfull >= c0 && cfull < c0 + kMqPanel / 8)
              for (uint32_t j = 0; j < tail; ++j) v = comb<SUM_OP>(SUM_OP, v, Schain[kMqTail + j]);
          } else {
            const uint32_t nsum = cfull > c0 ? (cfull - c0 < kMqPanel / 8 ? cfull - c0 : kMqPanel / 8) : 0u;
            uint32_t i = 0;
            for (; i + 4 <= nsum; i += 4) {
              const f32x4 w = *reinterpret_cast<const f32x4 *>(Schain + i);
              v = comb<SUM_OP>(SUM_OP, v, w.x);
              v = comb<SUM_OP>(SUM_OP, v, w.y);
              v = comb<SUM_OP>(SUM_OP, v, w.z);
              v = comb<SUM_OP>(SUM_OP, v, w.w);
            }
            for (; i < nsum; ++i) v = comb<SUM_OP>(SUM_OP, v, Schain[i]);
            if (tail && cfull >= c0 && cfull < c0 + kMqPanel / 8)
              for (uint32_t j = 0; j < tail; ++j) v = comb<SUM_OP>(SUM_OP, v, Schain[kMqTail + j]);
          }
          acc = v;
        }
        wave_lds_fence();
        pc += 1;
        if (pc == npanel) {
          // the tile is done: distances.rs:42-68 compute() for this lane's (query, row) -- value,
          // finiteness, f64 recovery -- then distances.rs:113-119 rank_value and the key (flat.rs:34-40)
          const int metric = a.metric;
          float raw = acc;
          if (metric == M_NIP) raw = -acc;
          else if (metric == M_L2) raw = finite_f32(acc) ? __builtin_sqrtf(acc) : acc;
          else if (OP == OP_JAC) {
            uint32_t qnz = a.q_nonzero[0];
#pragma unroll
            for (int q = 1; q < NQ; ++q) qnz = cq == q ? a.q_nonzero[q] : qnz;
            const uint32_t tot = (uint32_t)acc;
            const uint32_t xnz = tot >> 12, ham = tot & 4095u;
            const uint32_t uni = (qnz + xnz + ham) >> 1;
            const uint32_t inter = (qnz + xnz - ham) >> 1;
            raw = uni == 0 ? 0.0f : 1.0f - (float)inter / (float)uni;
          }
          bool valid = row_valid && (uint32_t)cq < a.nq;  // (padding queries of a short group offer nothing)
          const bool redo = valid && !finite_f32(raw);
          const uint64_t redo_mask = __ballot(redo);
          if (redo_mask) {
            const uint32_t pos = nredo + __popcll(redo_mask & ((1ull << lane) - 1));
            if (redo && pos < kMqRedo) s_redo[wib][pos] = grow;  // (the query is the lane's own: see the replay)
            if (redo && pos < kMqRedo) s_redo_q[wib][pos] = (uint32_t)cq;
            nredo += __popcll(redo_mask);
            valid = valid && !redo;
          }
          float rank = raw;
          if (metric == M_COS) rank = 1.0f - raw;
          else if (metric == M_IP) rank = -raw;
          if constexpr (!FAST) my_rank = row_valid ? a.id_rank[grow] : 0u;
          const uint64_t key = ((uint64_t)orderable(rank) << 32) | my_rank;
          // One compare and one ballot decide whether ANY of the eight lists takes anything from this
          // tile: each lane keeps the high half (the score part) of the threshold of its own query's
          // list -- an equal score goes on to the full comparison -- and thresholds only move inside
          // an offer, so the copy is refreshed behind the offers.  (The slim builds have no register
          // left for it -- the packed ones sit on the 168 that three blocks per CU allow -- and keep
          // it in LDS: one read per tile.)  Once the lists have warmed up nearly every tile ends
          // here -- eight compare / ballot / branch rounds were ~50 of the ~280 vector instructions
          // of a tile at d = 128, ~1 600 at d = 768.
          const uint32_t thr_hi = SLIM ? s_thr_hi[wib][lane] : my_thr_hi;
          if (__ballot(valid && (uint32_t)(key >> 32) <= thr_hi) != 0ull) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) tk[q].offer(valid && cq == q, key, grow, raw, lane);
            uint32_t now = (uint32_t)(tk[0].thr >> 32);
#pragma unroll
            for (int q = 1; q < NQ; ++q) now = cq == q ? (uint32_t)(tk[q].thr >> 32) : now;
            if constexpr (SLIM) s_thr_hi[wib][lane] = now;
            else my_thr_hi = now;
          }
          pc = 0;
          t += total_waves;
        }
      }
    }
    // nothing of the load stream may still be on its way into registers that get new owners below
    if constexpr (FAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the overflowed pairs, one per lane: f64 re-evaluation, then the same offer as everyone else
    if (nredo) {
      wave_lds_fence();
      if (nredo > kMqRedo) {
        if (lane == 0) atomicMax(a.status, kStatusRetry);
      } else {
        const bool mine = (uint32_t)lane < nredo;
        const uint32_t row = mine ? s_redo[wib][lane] : 0u, rq = mine ? s_redo_q[wib][lane] : 0u;
        float raw = 0.0f;
        bool valid = mine;
        if (mine) {
          raw = recover_overflow(a.metric, a.Q + (size_t)rq * ld, a.X + (size_t)row * a.stride, a.d);
          if (raw != raw) {  // (NaN: no f32 holds the value, distances.rs:92-98)
            atomicMax(a.status, kErrOverflow);
            valid = false;
            raw = 0.0f;
          }
        }
        float rank = raw;
        if (a.metric == M_COS) rank = 1.0f - raw;
        else if (a.metric == M_IP) rank = -raw;
        const uint64_t key = ((uint64_t)orderable(rank) << 32) | (mine ? a.id_rank[row] : 0u);
#pragma unroll
        for (int q = 0; q < NQ; ++q) tk[q].offer(valid && rq == (uint32_t)q, key, row, raw, lane);
      }
    }
  }
  // One list per block and query.  Every wave compacts its eight buffers; then the queries are
  // dealt to the waves (query q to wave q % 4), each absorbing the other three waves' buffers of
  // its queries and writing their lists -- the four waves merge side by side instead of wave 0
  // doing all eight in turn (a third of a small sweep's time went there).
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if ((uint32_t)q >= a.nq) break;
    tk[q].compact(lane);
    if (lane == 0) s_counts[q][wib] = tk[q].n;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if ((uint32_t)q >= a.nq) break;
    if ((q % kWavesPerBlock) != wib) continue;
    for (int w = 0; w < kWavesPerBlock; ++w) {
      if (w == wib) continue;
      // (the four waves' buffers of one query lie 2 * CAP u64 apart, wave 0's first)
      tk[q].absorb(tk[q].bk + ((ptrdiff_t)w - (ptrdiff_t)wib) * 2 * CAP, s_counts[q][w], lane);
    }
    const size_t list = (size_t)(a.first_query + q) * gridDim.x + blockIdx.x;
    tk[q].store(a.part_keys + list * a.k, a.part_pay + list * a.k, lane);
  }
}

constexpr size_t mq_lds(bool slim) {
  return (size_t)kWavesPerBlock * kMqNQ *
         ((size_t)mq_qs(slim) * 4 + (slim ? WaveTopK<kMqCapSlim>::lds_bytes() : WaveTopK<kMqCap>::lds_bytes()));
}
// which launches take the slim build: see launch_multi_op
inline bool mq_slim(uint32_t d, uint32_t k, int metric) {
  const int op = metric_op(metric);
  return d % kRowAlign == 0 && k <= kMqSlimMaxK && (op == OP_DOT || op == OP_L2 || op == OP_L1 || op == OP_LINF);
}

template <int OP, int ORDER, bool FAST, bool TAILED = false, bool SLIM = false, int PACK = 1>
static hipError_t launch_multi_t(const MultiScanArgs &a, uint32_t blocks, hipStream_t s) {
  auto kern = scan_multi_kernel<OP, ORDER, FAST, TAILED, SLIM, PACK>;
  hipError_t e = allow_lds(kern, mq_lds(SLIM));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * kWave), mq_lds(SLIM), s, a);
  return hipGetLastError();
}

// ORDERED: the metric's value depends on the lane order of reduce_add (sums); max and the
// small-integer sums of float hamming / jaccard do not.
template <int OP, bool ORDERED>
static hipError_t launch_multi_op(const MultiScanArgs &a, uint32_t blocks, hipStream_t s) {
  // Rows whose length is a multiple of 64 floats: no padding, no scalar tail, chunk counts in
  // groups of four -- the variants measured in DESIGN 4.4.  Other lengths (d = 100, 300, 1000 ...)
  // take the same scheduled-load kernels compiled with the tail handling in (it costs the
  // aligned shapes 9 %, so they do not carry it): padding is zero in rows and queries alike,
  // the scalar tail's products are filed beside the chunk sums.  (The variant with run-time lane order and
  // compiler-scheduled loads -- 2.2-2.7x slower, VT_MULTI_GENERAL through r05 -- has left the library.)
  if (a.d % kRowAlign != 0) {
    if (!ORDERED || a.order == 0) return launch_multi_t<OP, 0, true, true>(a, blocks, s);
    if (a.order == 1) return launch_multi_t<OP, ORDERED ? 1 : 0, true, true>(a, blocks, s);
    if (a.order == 2) return launch_multi_t<OP, ORDERED ? 2 : 0, true, true>(a, blocks, s);
    return launch_multi_t<OP, ORDERED ? 3 : 0, true, true>(a, blocks, s);
  }
  if constexpr (OP == OP_DOT || OP == OP_L2 || OP == OP_L1 || OP == OP_LINF) {
    if (mq_slim(a.d, a.k, a.metric)) {
      // a last panel of 128 or 64 floats (d = 384, 128, 64, 320 ...): two or four rows per load there
      // (the default lane order only: two more builds per operation)
      constexpr int kOrd = ORDERED ? kDefaultReduceOrder : 0;
      if (!ORDERED || a.order == kDefaultReduceOrder) {
        if (a.ld % kMqPanel == 128) return launch_multi_t<OP, kOrd, true, false, true, 2>(a, blocks, s);
        if (a.ld % kMqPanel == 64) return launch_multi_t<OP, kOrd, true, false, true, 4>(a, blocks, s);
      }
      if (!ORDERED || a.order == 0) return launch_multi_t<OP, 0, true, false, true>(a, blocks, s);
      if (a.order == 1) return launch_multi_t<OP, ORDERED ? 1 : 0, true, false, true>(a, blocks, s);
      if (a.order == 2) return launch_multi_t<OP, ORDERED ? 2 : 0, true, false, true>(a, blocks, s);
      return launch_multi_t<OP, ORDERED ? 3 : 0, true, false, true>(a, blocks, s);
    }
  }
  if (!ORDERED || a.order == 0) return launch_multi_t<OP, 0, true>(a, blocks, s);
  if (a.order == 1) return launch_multi_t<OP, ORDERED ? 1 : 0, true>(a, blocks, s);
  if (a.order == 2) return launch_multi_t<OP, ORDERED ? 2 : 0, true>(a, blocks, s);
  return launch_multi_t<OP, ORDERED ? 3 : 0, true>(a, blocks, s);
}

}  // namespace dev

uint32_t scan_multi_max_k(uint32_t) { return 32u; }
uint32_t scan_multi_tile_rows(uint32_t) { return (uint32_t)dev::kMqTR; }
size_t scan_multi_lds_bytes(uint32_t d, uint32_t k, int metric) { return dev::mq_lds(dev::mq_slim(d, k, metric)); }
int scan_multi_blocks_per_cu(uint32_t d, uint32_t k, int metric) { return dev::mq_slim(d, k, metric) ? 3 : 2; }

hipError_t launch_scan_multi(const MultiScanArgs &a, uint32_t blocks, hipStream_t s) {
  using namespace dev;
  if (a.nq == 0 || a.nq > kMultiMaxQueries || a.k == 0 || a.k > scan_multi_max_k(a.nq) || a.ld % kRowAlign != 0 || !a.id_rank ||
      a.ld < a.d || a.stride < a.ld || (a.metric == M_JAC && a.d >= 4096))
    return hipErrorInvalidValue;
  switch (metric_op(a.metric)) {
    case OP_DOT: return launch_multi_op<OP_DOT, true>(a, blocks, s);
    case OP_L2: return launch_multi_op<OP_L2, true>(a, blocks, s);
    case OP_L1: return launch_multi_op<OP_L1, true>(a, blocks, s);
    case OP_LINF: return launch_multi_op<OP_LINF, false>(a, blocks, s);
    case OP_HAM: return launch_multi_op<OP_HAM, false>(a, blocks, s);
    default: return launch_multi_op<OP_JAC, false>(a, blocks, s);
  }
}

}  // namespace vt
