// vt_scan.cuh -- K1: flat scan + fused wave64 top-k (gfx950).
//
// Replaces the hot loop of FlatIndex::search (native/vettore/src/flat.rs:104-118)
// and of vector_top_k (search.rs:49-70) together with the distance kernels
// (distances.rs:197-347).
//
// Arithmetic contract: every raw metric value is computed with exactly the
// reference's f32 operation order: per 8-float chunk eight separately rounded
// products (no FMA), one horizontal add in the lane order of
// wide::f32x8::reduce_add, then `acc += chunk_sum` sequentially over the chunks,
// then the scalar tail.  The result is bit-identical to the CPU oracle for the
// selected order, not merely within tolerance.
//
// How that is made HBM-bound:
//   * the corpus is one row-major slab (row stride a multiple of 256 B).  A wave
//     owns a tile of 32 rows and walks it in column panels of <= 96 chunks; a
//     panel is read as a run of fully coalesced 1-KiB wave loads (16 B/lane,
//     nontemporal), kU deep in flight in a register ring that runs ahead across
//     panel and tile boundaries;
//   * the query sits in LDS; each lane pair holds one 8-float chunk, so a chunk
//     sum is 4 v_mul + 3..7 v_add, one of them a DPP quad_perm add;
//   * chunk sums go to a per-wave LDS panel S[32 rows][<=96 chunks] (row stride
//     4*odd dwords: conflict-free ds_read_b128).  After a panel's last load lane
//     r walks row r's sums in order -- the reference's sequential `acc +=` chain
//     -- so 32 chains advance in parallel, once per panel (~1% of its time);
//   * after the last panel each lane owns a finished row: finiteness / f64
//     recovery, rank key, compare with the wave's k-th best (wave-uniform
//     threshold); rows that pass are appended, all at once, to a wave-private LDS
//     buffer that is compacted to its k best when it fills (WaveTopK).
// LDS per block is fixed (4 waves x (13.5 KiB panel + 2 or 5 KiB candidate buffer)
// + the query), whatever d is.
#pragma once
#include "vt_common.cuh"

namespace vt {
namespace dev {

constexpr uint32_t kPanelChunks = 96;  // max chunks per column panel (768 floats, 3 KiB per row)

struct ScanShape {
  uint32_t ld;         // floats read per row = padded_dim(d), multiple of 64
  uint32_t cfull;      // full 8-float chunks per row (d / 8)
  uint32_t tail;       // d % 8
  uint32_t npanel;     // column panels per row
  uint32_t pc;         // chunks per panel (multiple of 8)
  uint32_t pc_last;    // chunks in the last panel (multiple of 8)
  uint32_t ss;         // dwords per LDS panel row, 4 * odd
  uint32_t tail_base;  // dword offset of the 8 tail-product slots in a panel row
  uint32_t ntiles;
  uint32_t tr;         // rows per tile: 32, or 16 / 8 for small row counts (more tiles than waves)
  uint32_t segs;       // 1-KiB segments per panel of a tile = pc * tr / 32 (multiple of kU)
  uint32_t segs_last;
  uint32_t tile_floats;  // tr * row stride (contiguous scans)
  // Rows too long for the query to sit in LDS beside the panels (d beyond ~24 000: the reference
  // answers any d): the run-time-op kernel reads its query fragments from global memory instead
  // (4 d bytes, cache-resident) -- slower per load, no bound on d.
  uint32_t q_global;
};

// `tile_rows`: 32, 16 or 8 requested by the caller; a height whose panels are not a
// whole number of kU-segment groups falls back to the next larger one.
inline bool make_scan_shape(uint32_t d, uint32_t n, ScanShape *out, uint32_t tile_rows = kTileRows) {
  if (d == 0) return false;
  ScanShape p;
  p.ld = padded_dim(d);
  p.cfull = d / 8;
  p.tail = d % 8;
  const uint32_t cpr = p.ld / 8;
  uint32_t np = (cpr + kPanelChunks - 1) / kPanelChunks;
  uint32_t pc = round_up((cpr + np - 1) / np, 8);
  while (np > 1 && (np - 1) * pc >= cpr) {
    --np;
    pc = round_up((cpr + np - 1) / np, 8);
  }
  if (pc > kPanelChunks + 8) return false;
  p.npanel = np;
  p.pc = pc;
  p.pc_last = cpr - (np - 1) * pc;
  p.tail_base = pc;
  p.ss = pc + 12;  // pc % 8 == 0 -> (pc + 12) / 4 is odd
  uint32_t tr = tile_rows == 8 || tile_rows == 16 ? tile_rows : (uint32_t)kTileRows;
  while (tr < (uint32_t)kTileRows && ((p.pc * tr / 32) % kU != 0 || (p.pc_last * tr / 32) % kU != 0 ||
                                      (p.pc * tr) % 32 != 0 || (p.pc_last * tr) % 32 != 0))
    tr *= 2;
  p.tr = tr;
  p.segs = p.pc * tr / 32;
  p.segs_last = p.pc_last * tr / 32;
  p.tile_floats = 0;  // set by the launcher (needs the row stride)
  p.q_global = 0;
  p.ntiles = (n + tr - 1) / tr;
  *out = p;
  return true;
}

struct ScanDev {
  ScanArgs a;
  ScanShape p;
};

// ---- per-element operation of each metric family ---------------------------
template <int OP>
__device__ __forceinline__ float elem(int op_rt, float q, float x) {
  const int op = OP >= 0 ? OP : op_rt;
  switch (op) {
    case OP_DOT: return q * x;
    case OP_L2: {
      const float t = q - x;
      return t * t;
    }
    case OP_L1:
    case OP_LINF: return fabsf(q - x);
    case OP_HAM: return ((q != 0.0f) != (x != 0.0f)) ? 1.0f : 0.0f;
    default:  // OP_JAC: hamming count + 4096 * (x != 0); exact in f32 for d < 4096
      return (((q != 0.0f) != (x != 0.0f)) ? 1.0f : 0.0f) + ((x != 0.0f) ? 4096.0f : 0.0f);
  }
}

// the four products of one lane's 16 bytes: ONE branch on a run-time operation, not four
template <int OP>
__device__ __forceinline__ f32x4 elem4(int op_rt, const f32x4 q, const f32x4 x) {
  const int op = OP >= 0 ? OP : op_rt;
  switch (op) {
    case OP_DOT: return f32x4{elem<OP_DOT>(0, q.x, x.x), elem<OP_DOT>(0, q.y, x.y), elem<OP_DOT>(0, q.z, x.z), elem<OP_DOT>(0, q.w, x.w)};
    case OP_L2: return f32x4{elem<OP_L2>(0, q.x, x.x), elem<OP_L2>(0, q.y, x.y), elem<OP_L2>(0, q.z, x.z), elem<OP_L2>(0, q.w, x.w)};
    case OP_L1:
    case OP_LINF: return f32x4{elem<OP_L1>(0, q.x, x.x), elem<OP_L1>(0, q.y, x.y), elem<OP_L1>(0, q.z, x.z), elem<OP_L1>(0, q.w, x.w)};
    case OP_HAM: return f32x4{elem<OP_HAM>(0, q.x, x.x), elem<OP_HAM>(0, q.y, x.y), elem<OP_HAM>(0, q.z, x.z), elem<OP_HAM>(0, q.w, x.w)};
    default: return f32x4{elem<OP_JAC>(0, q.x, x.x), elem<OP_JAC>(0, q.y, x.y), elem<OP_JAC>(0, q.z, x.z), elem<OP_JAC>(0, q.w, x.w)};
  }
}

template <int OP>
__device__ __forceinline__ float comb(int op_rt, float a, float b) {
  const int op = OP >= 0 ? OP : op_rt;
  return op == OP_LINF ? fmaxf(a, b) : a + b;
}

// wide::f32x8::reduce_add of the chunk held by a lane pair (even lane: l0..l3,
// odd lane: l4..l7); both lanes return the chunk sum.
template <int OP, int ORDER>
__device__ __forceinline__ float chunk_sum(int op_rt, int order_rt, float p0, float p1, float p2, float p3, int odd) {
  const int order = ORDER >= 0 ? ORDER : order_rt;
  if (order == 1) {  // AVX: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))
    const float u0 = comb<OP>(op_rt, p0, dpp_xor1(p0));
    const float u1 = comb<OP>(op_rt, p1, dpp_xor1(p1));
    const float u2 = comb<OP>(op_rt, p2, dpp_xor1(p2));
    const float u3 = comb<OP>(op_rt, p3, dpp_xor1(p3));
    const float t = comb<OP>(op_rt, odd ? u1 : u0, odd ? u3 : u2);
    return comb<OP>(op_rt, t, dpp_xor1(t));
  }
  float e;
  if (order == 2)  // SEQ: (((l0+l1)+l2)+l3) + (((l4+l5)+l6)+l7)
    e = comb<OP>(op_rt, comb<OP>(op_rt, comb<OP>(op_rt, p0, p1), p2), p3);
  else if (order == 3)  // SSE2: ((l0+l2)+(l1+l3)) + ((l4+l6)+(l5+l7))
    e = comb<OP>(op_rt, comb<OP>(op_rt, p0, p2), comb<OP>(op_rt, p1, p3));
  else  // PAIR: ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))
    e = comb<OP>(op_rt, comb<OP>(op_rt, p0, p1), comb<OP>(op_rt, p2, p3));
  return comb<OP>(op_rt, e, dpp_xor1(e));
}

// distances.rs:70-90 recover_metric_overflow (+ the f64 branch of l2(),
// distances.rs:140-147), run by the one lane whose f32 result was non-finite.
// The recovered value comes back BY VALUE -- NaN where there is none (a recovered value is finite by
// construction, f64_to_f32): an out-parameter of a function that is not inlined lives in scratch memory, every
// launch of a kernel that may call it then sets up a scratch segment, and the reload behind the (rare) branch puts
// an s_waitcnt vmcnt(0) into every tile's epilogue (K1p, r04; K1 and K1m carried 16 bytes per lane until r05).
__device__ __forceinline__ float f64_as_f32_or_nan(double v) {  // distances.rs:92-98 f64_to_f32
  return isfinite(v) && v >= -(double)FLT_MAX && v <= (double)FLT_MAX ? (float)v : __builtin_nanf("");
}
__device__ __noinline__ static float recover_overflow(int metric, const float *q, const float *x, uint32_t d) {
  double acc = 0.0;
  switch (metric) {
    case M_L2:
    case M_L2SQ:
      for (uint32_t i = 0; i < d; ++i) {
        const double t = (double)q[i] - (double)x[i];
        acc += t * t;
      }
      if (metric == M_L2) {
        // l2(): (f64 sqrt) as f32, accepted when finite; compute()'s later
        // recovery reaches the same value or fails identically.
        const float v = (float)sqrt(acc);
        return finite_f32(v) ? v : __builtin_nanf("");
      }
      return f64_as_f32_or_nan(acc);
    case M_COS:
    case M_IP:
    case M_NIP:
      for (uint32_t i = 0; i < d; ++i) acc += (double)q[i] * (double)x[i];
      return f64_as_f32_or_nan(metric == M_NIP ? -acc : acc);
    case M_L1:
      for (uint32_t i = 0; i < d; ++i) acc += fabs((double)q[i] - (double)x[i]);
      return f64_as_f32_or_nan(acc);
    case M_LINF:
      for (uint32_t i = 0; i < d; ++i) acc = fmax(acc, fabs((double)q[i] - (double)x[i]));
      return f64_as_f32_or_nan(acc);
    default: return __builtin_nanf("");
  }
}

// Position of a wave in its stream of 1-KiB segments: tile t, column panel p,
// segment s of that panel (all wave-uniform), and this lane's (row in tile,
// column in panel) of the 4 floats it loads there.
struct Cursor {
  uint32_t t, p, s;
  uint32_t rowi, col;
};

// OP / ORDER < 0: taken from the arguments at run time.
// GENERAL: rows addressed through `gather` and/or a stride != ld (prefix scan).
// PADDED: d % 64 != 0 -- some chunks of a row are padding or the scalar tail.
// QGLOBAL (ScanShape.q_global): the query is read from global memory, not LDS.  A COMPILE-time
// property: a query pointer that may be either is a generic pointer, its loads are flat_load, a
// flat_load counts against vmcnt as well as lgkmcnt, and the wait for the query fragment then
// drains the whole register ring at every segment (r03, measured: gathered scans ran one load deep).
template <int OP, int ORDER, int CAP, bool GENERAL, bool PADDED, bool QGLOBAL = false>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void scan_topk_kernel(const ScanDev sd) {
  extern __shared__ __align__(16) float lds[];
  const ScanArgs &a = sd.a;
  const ScanShape &p = sd.p;
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int odd = lane & 1;
  const uint32_t q_lds = QGLOBAL ? 0u : p.ld;
  float *S = lds + q_lds + wib * (p.tr * p.ss);  // (a panel is as tall as the tiles: short tiles, more blocks per CU)
  unsigned char *tkbuf = reinterpret_cast<unsigned char *>(lds + q_lds + kWavesPerBlock * (p.tr * p.ss)) +
                         wib * WaveTopK<CAP>::lds_bytes();

  // batch mode (gathered scans): blockIdx.y selects the query and its row list
  const float *qsrc = a.q;
  const uint32_t *gather = a.gather;
  uint32_t nrows = a.n;
  uint32_t list_base = 0;
  if (GENERAL && a.batch_counts) {
    const uint32_t b = blockIdx.y;
    const uint32_t c = a.batch_counts[b];
    nrows = c < a.batch_cap ? c : a.batch_cap;
    qsrc += (size_t)b * (a.batch_qstride ? a.batch_qstride : p.ld);
    gather += (size_t)b * (a.batch_gather_stride ? a.batch_gather_stride : a.batch_cap * a.gather_stride);
    list_base = b * gridDim.x;
  }
  if (!QGLOBAL)
    for (uint32_t i = threadIdx.x; i < p.ld; i += blockDim.x) lds[i] = qsrc[i];
  __syncthreads();

  const int op_rt = metric_op(a.metric);
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = GENERAL ? (nrows + p.tr - 1) / p.tr : p.ntiles;

  WaveTopK<CAP> tk;
  tk.init(tkbuf, a.k);

  if (wave_global < ntiles) {
    const uint32_t last_tile = wave_global + ((ntiles - 1 - wave_global) / total_waves) * total_waves;
    // the two panel widths (floats) and this lane's start/step within them
    const uint32_t pw0 = p.pc * 8, pw1 = p.pc_last * 8;
    const uint32_t l4 = (uint32_t)lane * 4u;
    const uint32_t ir0 = l4 / pw0, ic0 = l4 - ir0 * pw0;
    const uint32_t ir1 = l4 / pw1, ic1 = l4 - ir1 * pw1;
    const uint32_t sr0 = 256u / pw0, sc0 = 256u - sr0 * pw0;
    const uint32_t sr1 = 256u / pw1, sc1 = 256u - sr1 * pw1;
    const uint32_t lastp = p.npanel - 1;

    auto enter_panel = [&](Cursor &c) {
      const bool last = c.p == lastp;
      c.rowi = last ? ir1 : ir0;
      c.col = last ? ic1 : ic0;
    };
    auto step_lane = [&](Cursor &c) {
      const bool last = c.p == lastp;
      const uint32_t pw = last ? pw1 : pw0;
      c.rowi += last ? sr1 : sr0;
      c.col += last ? sc1 : sc0;
      if (c.col >= pw) {
        c.col -= pw;
        c.rowi += 1;
      }
    };
    // GENERAL: lane r holds the source row of row r of the load cursor's tile (pf_idx), of the
    // tile before it (pf_idx_prev: the compute side may still be there) and of the tile after it
    // (pf_idx_next), fetched a whole tile ahead -- through the SCALAR cache: a tile's indices sit at
    // a wave-uniform address (constant address space => s_load).  (Through r02 every load of a
    // gathered scan first fetched its own index with a vector load: two dependent trips to memory
    // per 1-KiB load -- and any conditional vector load in this loop makes the compiler wait for
    // vmcnt(0) at every use of the ring: the eight-deep ring was one deep, 55 us per 8-row tile,
    // 0.68 ms to re-score the 154 000 candidate rows of a K2b batch.)
    uint32_t pf_idx = 0, pf_idx_prev = 0, pf_idx_next = 0;
    auto tile_index = [&](uint32_t t) -> uint32_t {
      const uint32_t g0 = t * p.tr;
      if (g0 >= nrows || g0 / p.tr != t) return 0u;  // (rows past the end read row 0: harmless)
      if (!gather) return ((uint32_t)lane < p.tr && g0 + (uint32_t)lane < nrows) ? g0 + (uint32_t)lane : 0u;
      typedef const __attribute__((address_space(4))) uint32_t *cu32_p;
      cu32_p sg = (cu32_p)(uintptr_t)(gather + (size_t)g0 * a.gather_stride);
      uint32_t v = 0;
      const uint32_t left = nrows - g0;  // >= 1; rows past the end repeat the last one (never used)
      for (uint32_t i0 = 0; i0 < p.tr; i0 += 8) {  // (tr is 8, 16 or 32) eight scalar loads in flight, one wait
        uint32_t sv[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
          const uint32_t i = i0 + j < left ? i0 + j : left - 1;
          sv[j] = sg[(size_t)i * a.gather_stride];
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) v = (uint32_t)lane == i0 + j ? sv[j] : v;
      }
      return v;
    };
    auto advance = [&](Cursor &c) {
      const uint32_t nseg = c.p == lastp ? p.segs_last : p.segs;
      c.s += 1;
      if (c.s == nseg) {
        c.s = 0;
        c.p += 1;
        if (c.p == p.npanel) {
          c.p = 0;
          c.t += total_waves;
          if (GENERAL) {
            pf_idx_prev = pf_idx;
            pf_idx = pf_idx_next;
            pf_idx_next = tile_index(c.t + total_waves);
          }
        }
        enter_panel(c);
      } else {
        step_lane(c);
      }
    };
    auto load_at = [&](const Cursor &c) -> f32x4 {
      const uint32_t t = c.t < last_tile ? c.t : last_tile;  // clamp at the end of the stream
      const uint32_t colf = c.p * pw0 + c.col;
      if (!GENERAL) {
        const float *base = a.X + (size_t)t * p.tile_floats;
        return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (c.rowi * (uint32_t)a.stride + colf)));
      }
      (void)t;
      const uint32_t src = (uint32_t)__shfl((int)pf_idx, (int)c.rowi, kWave);
      return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.X + (size_t)src * a.stride + colf));
    };

    Cursor pf;
    pf.t = wave_global;
    pf.p = 0;
    pf.s = 0;
    enter_panel(pf);
    if (GENERAL) {
      pf_idx = tile_index(wave_global);
      pf_idx_next = tile_index(wave_global + total_waves);
    }
    f32x4 buf[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      buf[u] = load_at(pf);
      advance(pf);
    }

    for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
      const uint32_t grow = t * p.tr + lane;  // lanes 0..tr-1 own a row
      const bool row_valid = (uint32_t)lane < p.tr && grow < nrows;
      uint32_t src_row = grow;
      if (GENERAL && gather) src_row = pf.t == t ? pf_idx : pf_idx_prev;  // (the load cursor is in this tile or the next)
      uint32_t my_rank = src_row;
      if (row_valid && a.id_rank) my_rank = a.id_rank[src_row];

      float acc = 0.0f;
      // jaccard: (hamming count + 4096 * non-zero count) is exact in f32 within a panel (<= 776
      // columns); across panels the two counts are carried as integers, so d is not bounded by it
      uint32_t jac_ham = 0, jac_xnz = 0;
      const bool jaccard = OP == OP_JAC || (OP < 0 && a.metric == M_JAC);
      Cursor cc;
      cc.t = t;
      for (cc.p = 0; cc.p < p.npanel; ++cc.p) {
        const uint32_t nseg = cc.p == lastp ? p.pc_last : p.pc;       // chunks per row in this panel
        const uint32_t nload = cc.p == lastp ? p.segs_last : p.segs;  // 1-KiB segments of the tile's panel
        const uint32_t c0 = cc.p * p.pc;  // first chunk of the panel
        enter_panel(cc);
        for (uint32_t s = 0; s < nload; s += kU) {
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const f32x4 x = buf[u];
            buf[u] = load_at(pf);
            advance(pf);

            f32x4 qv;
            if (QGLOBAL) qv = *reinterpret_cast<const f32x4 *>(qsrc + c0 * 8 + cc.col);
            else qv = *reinterpret_cast<const f32x4 *>(lds + c0 * 8 + cc.col);
            const f32x4 pr = elem4<OP>(op_rt, qv, x);
            const float p0 = pr.x, p1 = pr.y, p2 = pr.z, p3 = pr.w;
            const uint32_t cl = cc.col >> 3;  // chunk within the panel
            float *Srow = S + cc.rowi * p.ss;
            if (!PADDED || c0 + cl < p.cfull) {
              const float sum = chunk_sum<OP, ORDER>(op_rt, a.order, p0, p1, p2, p3, odd);
              if (!odd) Srow[cl] = sum;
            } else if (c0 + cl == p.cfull) {
              // tail chunk: the reference adds these products one by one
              *reinterpret_cast<f32x4 *>(Srow + p.tail_base + odd * 4) = f32x4{p0, p1, p2, p3};
            }
            step_lane(cc);
          }
        }

        // panel complete: lane r continues row r's sequential chain
        wave_lds_fence();
        if ((uint32_t)lane < p.tr) {
          const float *Sr = S + lane * p.ss;
          uint32_t nsum = nseg;
          if (PADDED) nsum = p.cfull > c0 ? (p.cfull - c0 < nseg ? p.cfull - c0 : nseg) : 0;
          uint32_t c = 0;
          for (; c + 4 <= nsum; c += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(Sr + c);
            acc = comb<OP>(op_rt, acc, v.x);
            acc = comb<OP>(op_rt, acc, v.y);
            acc = comb<OP>(op_rt, acc, v.z);
            acc = comb<OP>(op_rt, acc, v.w);
          }
          if (PADDED) {
            for (; c < nsum; ++c) acc = comb<OP>(op_rt, acc, Sr[c]);
            if (p.tail && p.cfull >= c0 && p.cfull < c0 + nseg)
              for (uint32_t j = 0; j < p.tail; ++j) acc = comb<OP>(op_rt, acc, Sr[p.tail_base + j]);
          }
          if (jaccard) {  // the panel's two counts leave the float while they are still exact in it
            const uint32_t tot = (uint32_t)acc;
            jac_xnz += tot >> 12;
            jac_ham += tot & 4095u;
            acc = 0.0f;
          }
        }
        wave_lds_fence();
      }

      // distances.rs:42-68 compute(): value, finiteness, f64 recovery
      const int metric = a.metric;
      float raw = acc;
      if (metric == M_NIP) raw = -acc;
      else if (metric == M_L2) raw = finite_f32(acc) ? __builtin_sqrtf(acc) : acc;
      else if (jaccard) {
        const uint32_t xnz = jac_xnz, ham = jac_ham;
        const uint32_t uni = (a.q_nonzero + xnz + ham) >> 1;
        const uint32_t inter = (a.q_nonzero + xnz - ham) >> 1;
        raw = uni == 0 ? 0.0f : 1.0f - (float)inter / (float)uni;
      }
      bool valid = row_valid;
      if (valid && !finite_f32(raw)) {
        const float rec = recover_overflow(metric, QGLOBAL ? qsrc : lds, a.X + (size_t)src_row * a.stride, a.d);
        if (rec == rec) {
          raw = rec;
        } else {
          atomicMax(a.status, kErrOverflow);
          valid = false;
        }
      }
      // distances.rs:113-119 rank_value, flat.rs:34-40 ordering
      float rank = raw;
      if (metric == M_COS) rank = 1.0f - raw;
      else if (metric == M_IP) rank = -raw;
      const uint64_t key = ((uint64_t)orderable(rank) << 32) | my_rank;
      if (a.has_lo) valid = valid && key > a.lo_key;
      if (a.key_out) {  // key-column mode (wave-uniform)
        if (row_valid) {
          a.key_out[grow] = valid ? key : kEmptyKey;
          if (a.pay_out) {
            Payload pv;
            pv.row = src_row;
            pv.raw = raw;
            a.pay_out[grow] = pv;
          }
        }
      } else {
        tk.offer(valid, key, src_row, raw, lane);
      }
    }
  }
  // one list per block: wave 0 absorbs the other waves' buffers
  if (a.key_out) return;
  __shared__ uint32_t s_counts[kWavesPerBlock];
  tk.merge_block(wib, kWavesPerBlock, s_counts, lane);
  if (wib == 0)
    tk.store(a.part_keys + (size_t)(list_base + blockIdx.x) * a.k, a.part_pay + (size_t)(list_base + blockIdx.x) * a.k,
             lane);
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

template <int OP, int ORDER, int CAP, bool GENERAL, bool PADDED, bool QGLOBAL = false>
inline hipError_t launch_scan_t(const ScanDev &sd, uint32_t blocks, size_t lds, hipStream_t s, uint32_t nq = 1) {
  auto kern = scan_topk_kernel<OP, ORDER, CAP, GENERAL, PADDED, QGLOBAL>;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks, nq), dim3(kWavesPerBlock * kWave), lds, s, sd);
  return hipGetLastError();
}

// One translation unit per operation family instantiates its kernels.
hipError_t launch_scan_dot(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s);
hipError_t launch_scan_l2(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s);
hipError_t launch_scan_misc(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s);
hipError_t launch_scan_l1(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s);
hipError_t launch_scan_general(const ScanDev &sd, uint32_t blocks, uint32_t nq, size_t lds, hipStream_t s);
hipError_t launch_scan_gather(const ScanDev &sd, uint32_t blocks, uint32_t nq, size_t lds, hipStream_t s);
constexpr int kDefaultReduceOrder = 3;  // VT_ORDER_SSE2 (include/vettore_flat.h)

// order (0..3) x candidate buffer (k <= kSmallK -> small) x padded
#define VT_SCAN_ORDERS(OPV, CAPV, PADV)                                                            \
  do {                                                                                             \
    if (order == 0) return launch_scan_t<OPV, 0, CAPV, false, PADV>(sd, blocks, lds, s);           \
    if (order == 1) return launch_scan_t<OPV, 1, CAPV, false, PADV>(sd, blocks, lds, s);           \
    if (order == 2) return launch_scan_t<OPV, 2, CAPV, false, PADV>(sd, blocks, lds, s);           \
    return launch_scan_t<OPV, 3, CAPV, false, PADV>(sd, blocks, lds, s);                           \
  } while (0)
#define VT_SCAN_DISPATCH_ORDERED(OPV)                                                              \
  do {                                                                                             \
    const int order = sd.a.order;                                                                  \
    const bool big = sd.a.k > kSmallK;                                                             \
    if (!padded) {                                                                                 \
      if (!big) VT_SCAN_ORDERS(OPV, kCapSmall, false);                                             \
      VT_SCAN_ORDERS(OPV, kCapLarge, false);                                                       \
    }                                                                                              \
    if (!big) VT_SCAN_ORDERS(OPV, kCapSmall, true);                                                \
    VT_SCAN_ORDERS(OPV, kCapLarge, true);                                                          \
  } while (0)

}  // namespace dev
}  // namespace vt
