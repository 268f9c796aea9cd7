// kernels.hip -- hand-written gfx950 kernels of the aligner hot path: the ICP passes and their control steps.
// (ingest, grid build and sorting, once per set_fixed / set_moving, are in kernels_prep.hip)
//
//   icp_step<DIM,PLANE>      : ONE kernel per slice per ICP iteration = finder->compute()
//                              (aligner_slice_processor_impl.cpp:39-48) fused with the factor's
//                              per-correspondence linearisation + robustifier + JtJ/Jtr reduction
//                              ([EXT] FactorCorrespondenceDriven_, SURVEY.md A1+A3+A4)
//   icp_control              : the rest of one _runSolver iteration on device: association check,
//                              prior factors, 6x6 solve, X <- X [+] dx, IterationStats, termination
//                              (multi_aligner_impl.cpp:104-127; aligner_termination_criteria_impl.cpp:24-65)
//   icp_init / icp_post / icp_finalize : compute() prologue/epilogue (multi_aligner_impl.cpp:52-95,163-181)
//
// Arithmetic follows DESIGN.md "arithmetic specification": float32 geometry in a fixed operation
// order without FMA contraction (-ffp-contract=off), products widened to double, sums carried as
// exact 64-bit fixed point so that the reduction order (lanes, waves, blocks, GPUs) cannot change
// a single bit of H, b or the statistics.
#include <cstdlib>
#include <algorithm>

#include "kernels.h"

#include <type_traits>

#include "det_math.h"

// Timing knobs (SRRG2_AMD_TUNE bits 1, 2, 8, 16, 32, 64, 128, 256, 1024, 2048) switch parts of the kernels OFF to see what
// they cost: WRONG results, so they only exist in profiling builds (make EXTRA=-DSRRG2_TIMING_KNOBS).  The other bits
// choose between exact strategies and stay available.
#ifdef SRRG2_TIMING_KNOBS
#define KNOB(t, bit) (((t) & (bit)) != 0)
#else
#define KNOB(t, bit) false
#endif

#include "device_util.h"

// ============================================================================================
// the fused ICP step kernel
// ============================================================================================
namespace {

// Fixed-point terms: every product (w 2^k J_ra) * J_rb is rounded ONCE, to nearest-even, onto the
// integer grid by a fused multiply-add onto FX_MAGIC = 1.5 * 2^52 (a double in [2^52, 2^53) has ulp 1, so the result is
// FX_MAGIC + integer); the rows of one correspondence are chained on the same accumulator.  The integer is the bit
// pattern minus that of FX_MAGIC.  |integer| < 2^51 is guaranteed by dm::fixed_point_exponent.  One v_fma_f64 per term.
#define FX_MAGIC 6755399441055744.0               // 1.5 * 2^52
#define FX_MAGIC_BITS 0x4338000000000000ll        // its bit pattern
__device__ __forceinline__ long long fx_bits(double biased) { return __double_as_longlong(biased) - FX_MAGIC_BITS; }

// One candidate.  Squared distance in the specified float32 operation order, then the running minimum of the 64-bit key
// (d2 bits << 32 | fixed index) -- d2 >= 0, so the bit pattern orders like the value and the low word breaks ties
// towards the smaller index -- taken with ONE v_min_f64: read as a double the key is a positive finite number (the
// exponent field of a float never reaches 0x7ff in the top 11 bits of the pair; denormals are enabled for float64 and
// compare exactly), and positive doubles order like their bit patterns.  No branch, no compare + selects.
// The position of the winner in the sorted array is not tracked (two selects per candidate): finish_point looks it up
// once through GridDev::pos_of.  An invalid candidate (the masked tail of a four-wide group) becomes the neutral key.
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long) a)), "v"(__longlong_as_double((long long) b)));
  return (unsigned long long) __double_as_longlong(r);
}

template <int DIM>
__device__ __forceinline__ float cand_d2(const float4 f, float qx, float qy, float qz) {
  const float dx = f.x - qx, dy = f.y - qy;
  float d2       = dx * dx + dy * dy;
  if (DIM == 3) {
    const float dz = f.z - qz;
    d2             = d2 + dz * dz;
  }
  return d2;
}

template <int DIM>
__device__ __forceinline__ void test_candidate(const float4 f, float qx, float qy, float qz, bool valid,
                                               unsigned long long& bkey) {
  const float d2 = valid ? cand_d2<DIM>(f, qx, qy, qz) : INFINITY;
  const unsigned idx = valid ? (unsigned) __float_as_int(f.w) : (unsigned) NO_MATCH;
  bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d2) << 32) | idx);
}

// ... and the squared distance of the runner-up (b2 = second smallest d2 over the candidates seen), for the exclusion
// radius: with best <= b2 the new runner-up is the median of (best, b2, d2) -- one v_med3_f32.
template <int DIM>
__device__ __forceinline__ void test_candidate2(const float4 f, float qx, float qy, float qz, bool valid,
                                                unsigned long long& bkey, float& b2) {
  const float d2 = valid ? cand_d2<DIM>(f, qx, qy, qz) : INFINITY;
  const unsigned idx = valid ? (unsigned) __float_as_int(f.w) : (unsigned) NO_MATCH;
  b2   = __builtin_amdgcn_fmed3f(__uint_as_float((unsigned) (bkey >> 32)), b2, d2);
  bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d2) << 32) | idx);
}

// -DSRRG2_TIMELINE: per-wave timeline of the step kernel (tools/timeline.py); compiled out of the product build
#ifdef SRRG2_TIMELINE
#define STAMP(slot, k)                                                       \
  do {                                                                       \
    if (slot) {                                                              \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            \
      if ((threadIdx.x & 63) == 0) (slot)[k] = (unsigned long long) wall_clock64(); \
    }                                                                        \
  } while (0)
#else
#define STAMP(slot, k) do { } while (0)
#endif
// margin of a trimmed scan beyond the previous neighbour's distance: 2 |q - q'| capped at PAD_CAP cells + 2% of a
// cell.  A margin that turns out too small only means one more (trimmed) scan next iteration; measured on C2 / C4:
// no cap 27.4k / 196k, 0.25: 28.1k / 199k, 0.1: 28.7k / 202k it/s (round 2, grid scans).  Round 4, searches over the cell
// neighbour lists (a search is cheaper, a candidate is not): C4-256 0.4 / 0.2 / 0.1 / 0.05 / 0.025 / 0: 539 / 575 / 596 / 611 /
// 612 / 615 k it/s, C4-32 423 / 444 / 458 / 457 / 464 / 470 k, C2 43.2 / 44.8 / 45.4 / 45.9 / 45.7 / 45.5 k; tracker cycle, small
// clouds and the 60 % overlap unchanged (profiles/archive/r4w_ab_scan_margin.txt): the margin is the 2 % of a cell alone.
#ifndef PAD_CAP
#define PAD_CAP 0.0f
#endif
#ifndef PAD_MIN
#define PAD_MIN 0.02f  // the part of the margin that does not depend on the motion, in cells
#endif
#ifndef SCAN_W0
#define SCAN_W0 1
#endif
#define NO_KEY ((0x7f800000ull << 32) | (unsigned long long) NO_MATCH)  // (+inf, NO_MATCH)
__device__ __forceinline__ unsigned long long make_key(float best, int bidx) {
  return ((unsigned long long) __float_as_uint(best) << 32) | (unsigned) bidx;
}
__device__ __forceinline__ float key_best(unsigned long long k) { return __uint_as_float((unsigned) (k >> 32)); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int) (unsigned) k; }

// scan the contiguous candidates [j, e): full groups of four independent 16-byte loads without masks, then the tail
// (1-3 candidates; reads up to 2 entries past e, masked out: the sorted arrays are allocated with slack).
template <int DIM>
__device__ __forceinline__ void scan_range2(const float4* __restrict__ pts, int j, int e, float qx, float qy, float qz,
                                            unsigned long long& bkey, float& b2) {
  for (; j + 4 <= e; j += 4) {
    const float4 f0 = pts[j];
    const float4 f1 = pts[j + 1];
    const float4 f2 = pts[j + 2];
    const float4 f3 = pts[j + 3];
    test_candidate2<DIM>(f0, qx, qy, qz, true, bkey, b2);
    test_candidate2<DIM>(f1, qx, qy, qz, true, bkey, b2);
    test_candidate2<DIM>(f2, qx, qy, qz, true, bkey, b2);
    test_candidate2<DIM>(f3, qx, qy, qz, true, bkey, b2);
  }
  if (j < e) {
    const float4 f0 = pts[j];
    const float4 f1 = pts[j + 1];
    const float4 f2 = pts[j + 2];
    test_candidate2<DIM>(f0, qx, qy, qz, true, bkey, b2);
    test_candidate2<DIM>(f1, qx, qy, qz, j + 1 < e, bkey, b2);
    test_candidate2<DIM>(f2, qx, qy, qz, j + 2 < e, bkey, b2);
  }
}

// Cells that can hold a fixed point f with d2(f, q) <= r2 (d2 as computed by test_candidate), per axis:
// [cell(q - s), cell(q + s)] with s = r inflated by 1e-5 relative + 4.8e-7 of the coordinate magnitude.  The cell of a
// point is a monotone function of its coordinate (fl(x - o), fl(. * inv_h), floor: all monotone), and the float32
// roundings of d2 and of q -+ s are far inside the inflation, so the range is conservative: trimming a scan to it
// can never drop the exact nearest neighbour.  r2 = +inf gives the unbounded range.
__device__ __forceinline__ float ball_radius(float r2) { return sqrtf(r2) * 1.00001f; }
__device__ __forceinline__ void axis_range(float q, float rr, float o, float inv_h, int clo, int chi, int n, int& lo,
                                           int& hi) {
  const float s = rr + (fabsf(q) + rr) * 4.8e-7f;
  lo            = max(max(cell_coord(q - s, o, inv_h), clo), 0);
  hi            = min(min(cell_coord(q + s, o, inv_h), chi), n - 1);
}

// first search phase: the 3^DIM cells around the query, trimmed to the ball of squared radius r2box (an upper bound
// of the nearest-neighbour distance known beforehand, +inf if none).  All row ranges are fetched up front
// (independent loads), then the candidates of each row are streamed four at a time.
template <int DIM>
__device__ __forceinline__ void scan_radius1(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                             float r2box, unsigned long long& bkey, float& b2,
                                             float& complete2, unsigned long long* tl = nullptr) {
  constexpr int NROWS = DIM == 3 ? 9 : 3;
  complete2 = r2box;
  const float rr = ball_radius(r2box);
  int x0, x1, y0, y1, z0 = 0, z1 = 0;
  axis_range(qx, rr, g.ox, g.inv_h, cx - 1, cx + 1, g.nx, x0, x1);
  axis_range(qy, rr, g.oy, g.inv_h, cy - 1, cy + 1, g.ny, y0, y1);
  if (DIM == 3) axis_range(qz, rr, g.oz, g.inv_h, cz - 1, cz + 1, g.nz, z0, z1);
  if (x0 > x1) return;
  // The step kernel is bound by the rate at which a CU's texture path accepts scattered per-lane requests (one per
  // lane per load instruction; measured with the per-wave timeline, tools/timeline.py) -- so every load below is
  // predicated per lane: rows outside the ball and empty rows issue no request at all.
  int rs[NROWS], re[NROWS];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    const int y = cy + (r % 3) - 1;
    const int z = DIM == 3 ? cz + (r / 3) - 1 : 0;
    rs[r] = re[r] = 0;
    if (y >= y0 && y <= y1 && z >= z0 && z <= z1) {
      const int row = (z * g.ny + y) * g.nx;
      rs[r] = g.cell_start[row + x0];
      re[r] = g.cell_start[row + x1 + 1];
    }
  }
  STAMP(tl, 2);  // row ranges arrived
  // The row through the query's own cell first: once the estimate is roughly right the nearest neighbour is there, and
  // its distance (+ a pad, so that the scan still proves an exclusion margin) prunes the other rows by their distance
  // to the query.  complete2 = squared radius inside which this scan has seen every fixed point of the block.
  constexpr int RC = DIM == 3 ? 4 : 1;
  scan_range2<DIM>(g.pts, rs[RC], re[RC], qx, qy, qz, bkey, b2);
  rs[RC] = re[RC] = 0;
  complete2 = r2box;
  if (key_idx(bkey) != NO_MATCH) {
    const float rb = (sqrtf(key_best(bkey)) + (PAD_CAP + PAD_MIN) * g.h) * 1.00001f;
    complete2      = fminf(complete2, rb * rb);
  }
  if (complete2 < 3.0e38f) {
    const float rb = sqrtf(complete2);
#pragma unroll
    for (int r = 0; r < NROWS; ++r) {
      if (r == RC) continue;
      const int y     = cy + (r % 3) - 1;
      const float ylo = g.oy + (float) y * g.h;
      float dy        = fmaxf(fmaxf(ylo - qy, qy - (ylo + g.h)), 0.f);
      dy              = fmaxf(dy - (0.01f * g.h + (fabsf(qy) + rb) * 2e-6f), 0.f);
      float rem       = complete2 * 1.00002f - dy * dy;
      if (DIM == 3) {
        const int z     = cz + (r / 3) - 1;
        const float zlo = g.oz + (float) z * g.h;
        float dz        = fmaxf(fmaxf(zlo - qz, qz - (zlo + g.h)), 0.f);
        dz              = fmaxf(dz - (0.01f * g.h + (fabsf(qz) + rb) * 2e-6f), 0.f);
        rem             = rem - dz * dz;
      }
      if (rem < 0.f) rs[r] = re[r] = 0;  // every point of this row is farther than the pruning radius
    }
  }
  // first W0 candidates of ALL remaining rows in flight together (one round trip instead of one per row), then the
  // rows that hold more
  constexpr int W0 = SCAN_W0;
  float4 c[NROWS][W0];
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
#pragma unroll
    for (int k = 0; k < W0; ++k) {
      c[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rs[r] + k < re[r]) c[r][k] = g.pts[rs[r] + k];
    }
  }
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    if (__any(rs[r] < re[r])) {
#pragma unroll
      for (int k = 0; k < W0; ++k)
        test_candidate2<DIM>(c[r][k], qx, qy, qz, rs[r] + k < re[r], bkey, b2);
    }
  }
  STAMP(tl, 3);  // first W0 candidates of every row tested
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    scan_range2<DIM>(g.pts, rs[r] + W0, re[r], qx, qy, qz, bkey, b2);
  }
}

// radius-2 cube (trimmed to the ball of the best candidate so far), one z-layer (5 rows) at a time: the 10 range
// fetches of a layer are independent, so a lane pays one fetch latency per layer instead of one per row.  Re-visits
// the radius-1 block (harmless: the minimum is idempotent).
template <int DIM>
__device__ __forceinline__ void scan_radius2(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                             float r2box, unsigned long long& bkey, float& b2) {
  const float rr = ball_radius(r2box);
  int x0, x1, y0, y1, z0 = 0, z1 = 0;
  axis_range(qx, rr, g.ox, g.inv_h, cx - 2, cx + 2, g.nx, x0, x1);
  axis_range(qy, rr, g.oy, g.inv_h, cy - 2, cy + 2, g.ny, y0, y1);
  if (DIM == 3) axis_range(qz, rr, g.oz, g.inv_h, cz - 2, cz + 2, g.nz, z0, z1);
  if (x0 > x1) return;
  for (int z = z0; z <= z1; ++z) {
    int rs[5], re[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int y   = cy + r - 2;
      const bool ok = y >= y0 && y <= y1;
      const int row = ok ? (z * g.ny + y) * g.nx : 0;
      rs[r] = ok ? g.cell_start[row + x0] : 0;
      re[r] = ok ? g.cell_start[row + x1 + 1] : 0;
    }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      scan_range2<DIM>(g.pts, rs[r], re[r], qx, qy, qz, bkey, b2);
    }
  }
}

// The SHELL of the radius-2 cube: the cells of the 5^DIM cube that are not in the 3^DIM block (which scan_radius1 has
// seen, trimmed to its own ball), trimmed to the ball of squared radius r2box.  Continues the running (key, runner-up)
// pair of the first phase: every fixed point of cube ∩ ball is met exactly once over both phases.  A lane that found a
// candidate in the block but could not settle on it (the usual case of a misaligned first pass: ~40 % of the lanes)
// only has to look at the few shell cells its candidate's ball reaches, instead of re-reading the whole block.
template <int DIM>
__device__ __forceinline__ void scan_shell2(const GridDev& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                            float r2box, unsigned long long& bkey, float& b2) {
  const float rr = ball_radius(r2box);
  int x0, x1, y0, y1, z0 = 0, z1 = 0;
  axis_range(qx, rr, g.ox, g.inv_h, cx - 2, cx + 2, g.nx, x0, x1);
  axis_range(qy, rr, g.oy, g.inv_h, cy - 2, cy + 2, g.ny, y0, y1);
  if (DIM == 3) axis_range(qz, rr, g.oz, g.inv_h, cz - 2, cz + 2, g.nz, z0, z1);
  if (x0 > x1) return;
  // inner rows (|y - cy| <= 1 and |z - cz| <= 1) keep only the cells cx - 2 and cx + 2; the x-extent of the block itself
  // is [cx - 1, cx + 1] clipped to the grid exactly as scan_radius1 clipped it
  const int bx0 = max(cx - 1, 0), bx1 = min(cx + 1, g.nx - 1);
  for (int z = z0; z <= z1; ++z) {
    const bool zin = DIM == 3 ? (z >= cz - 1 && z <= cz + 1) : true;
    int rs[10], re[10];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const int y     = cy + r - 2;
      const bool ok   = y >= y0 && y <= y1;
      const bool inner = zin && r >= 1 && r <= 3;
      const int row   = ok ? (z * g.ny + y) * g.nx : 0;
      // left part: [x0, min(x1, bx0 - 1)] for inner rows, the whole [x0, x1] otherwise; right part: inner rows only
      const int la = x0, lb = inner ? min(x1, bx0 - 1) : x1;
      const int ra = max(x0, bx1 + 1), rb = x1;
      const bool lok = ok && la <= lb, rok = ok && inner && ra <= rb;
      rs[2 * r]     = lok ? g.cell_start[row + la] : 0;
      re[2 * r]     = lok ? g.cell_start[row + lb + 1] : 0;
      rs[2 * r + 1] = rok ? g.cell_start[row + ra] : 0;
      re[2 * r + 1] = rok ? g.cell_start[row + rb + 1] : 0;
    }
#pragma unroll
    for (int r = 0; r < 10; ++r) scan_range2<DIM>(g.pts, rs[r], re[r], qx, qy, qz, bkey, b2);
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum 32 per-lane values over the 64 lanes of a wave with a transposing butterfly: at every step a lane
// hands half of its values to its partner, so the reduction costs 32 exchanges instead of 32 x 6.  The two
// widest steps use gfx950's v_permlane32_swap / v_permlane16_swap (pure VALU, no LDS crossbar, no selects).
// On return lane L holds the wave total of value index
//   16*bit5(L) + 8*bit4(L) + 4*bit3(L) + 2*bit2(L) + bit1(L);   lanes 2a and 2a+1 hold the same total.
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long swap_add32(long long a, long long b) {
  // a' = [a.lo32lanes, b.lo32lanes], b' = [a.hi32lanes, b.hi32lanes]; a' + b'
  v2u lo = __builtin_amdgcn_permlane32_swap((unsigned) a, (unsigned) b, false, false);
  v2u hi = __builtin_amdgcn_permlane32_swap((unsigned) ((unsigned long long) a >> 32),
                                            (unsigned) ((unsigned long long) b >> 32), false, false);
  long long x = (long long) (((unsigned long long) hi.x << 32) | lo.x);
  long long y = (long long) (((unsigned long long) hi.y << 32) | lo.y);
  return x + y;
}
__device__ __forceinline__ long long swap_add16(long long a, long long b) {
  v2u lo = __builtin_amdgcn_permlane16_swap((unsigned) a, (unsigned) b, false, false);
  v2u hi = __builtin_amdgcn_permlane16_swap((unsigned) ((unsigned long long) a >> 32),
                                            (unsigned) ((unsigned long long) b >> 32), false, false);
  long long x = (long long) (((unsigned long long) hi.x << 32) | lo.x);
  long long y = (long long) (((unsigned long long) hi.y << 32) | lo.y);
  return x + y;
}

__device__ __forceinline__ long long wave_transpose_reduce(long long (&v)[ACC_N], int lane, int& index_out) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = swap_add32(v[i], v[i + 16]);  // lanes <32 keep i, lanes >=32 keep i+16
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = swap_add16(v[i], v[i + 8]);    // bit4 == 0 keep i, bit4 == 1 keep i+8
  {  // xor 8: 4 values kept
    const bool upper = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long send = upper ? v[i] : v[i + 4];
      const long long keep = upper ? v[i + 4] : v[i];
      v[i]                 = keep + __shfl_xor(send, 8);
    }
  }
  {  // xor 4: 2 values kept
    const bool upper = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long send = upper ? v[i] : v[i + 2];
      const long long keep = upper ? v[i + 2] : v[i];
      v[i]                 = keep + __shfl_xor(send, 4);
    }
  }
  {  // xor 2: 1 value kept
    const bool upper     = (lane & 2) != 0;
    const long long send = upper ? v[0] : v[1];
    const long long keep = upper ? v[1] : v[0];
    v[0]                 = keep + __shfl_xor(send, 2);
  }
  v[0] += __shfl_xor(v[0], 1);
  index_out = ((lane >> 5) & 1) * 16 + ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 +
              ((lane >> 1) & 1);
  return v[0];
}

// per-correspondence factor arithmetic shared by every cue slice: chi, robustifier, the 27 (9 in 2D) fixed-point
// terms of w J^T J and w J^T e, and the statistics.  Returns the srrg2_factor_status of the correspondence.
template <int D, int ROWS>
__device__ __forceinline__ uint8_t factor_accumulate(const float (&J)[ROWS][D], const float (&e)[ROWS], bool invalid,
                                                     int rk, float thr, double scale, bool skip_terms,
                                                     long long (&acc)[ACC_N]) {
  float chi = e[0] * e[0];
#pragma unroll
  for (int r = 1; r < ROWS; ++r) chi = chi + e[r] * e[r];
  acc[ACC_N_CORR] += 1;
  if (!isfinite(chi) || invalid) return SRRG2_FACTOR_SUPPRESSED;
  float w         = 1.f;
  bool kernelized = false;
  if (rk != SRRG2_ROBUST_NONE && !(chi < thr)) {
    kernelized = true;
    w = rk == SRRG2_ROBUST_CLAMP ? 0.f : (rk == SRRG2_ROBUST_SATURATED ? thr / chi : 1.0f / (1.0f + chi / thr));
  }
  const long long chi_fx = fx_bits(__fma_rn((double) chi, scale, FX_MAGIC));  // (chi 2^k is exact: rne to the grid)
  // (branch-free on purpose: an if/else here gets merged into a dynamically indexed acc[] access = scratch memory)
  acc[ACC_N_OUT] += kernelized ? 1 : 0;
  acc[ACC_CHI_OUT] += kernelized ? chi_fx : 0;
  acc[ACC_N_IN] += kernelized ? 0 : 1;
  acc[ACC_CHI_IN] += kernelized ? 0 : chi_fx;
  if (w != 0.f && !skip_terms) {
    // (w * 2^k) is exact, (w * 2^k) * J is exact (24 x 24 bits); the fma rounds each product once onto the grid
    const double ws = (double) w * scale;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double wj[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) wj[r] = ws * (double) J[r][a];
#pragma unroll
      for (int b = a; b < D; ++b) {
        double t = FX_MAGIC;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) t = __fma_rn(wj[r], (double) J[r][b], t);
        acc[hidx(a, b)] += fx_bits(t);
      }
      double t = FX_MAGIC;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) t = __fma_rn(wj[r], (double) e[r], t);
      acc[ACC_B + a] += fx_bits(t);
    }
  }
  return kernelized ? SRRG2_FACTOR_KERNELIZED : SRRG2_FACTOR_INLIER;
}

// block reduction: transposing butterfly per wave, 4 waves through LDS, then the block adds its 32 totals into one of
// PARTIAL_SLOTS slot sets of its problem with 64-bit atomics.  Integer addition is exact, so the order of the atomic
// adds cannot change a bit; spreading the blocks over 32 slot sets keeps same-address contention (~10 ns per atomic)
// at <= ceil(blocks/32) per address, and the control kernel has 32 x 32 values to sum instead of blocks x 32.
// (History: one atomic target per entry serialised 391 blocks -> ~90 us; plain per-block partials made the control
// kernel read 100 KB -> ~10 us.  profiles/archive/r1a, r1b.)
// NW = waves per workgroup.  local != null: the workgroup owns the whole problem (k_icp_small) and adds into its LDS sums.
template <int NW>
__device__ __forceinline__ void block_reduce_store(long long (&acc)[ACC_N], long long* __restrict__ partials, int prob,
                                                   int block, long long* local = nullptr) {
  __shared__ long long red[NW][ACC_N];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int my_index;
  const long long total = wave_transpose_reduce(acc, lane, my_index);
  if ((lane & 1) == 0) red[wid][my_index] = total;
  __syncthreads();
  if (threadIdx.x < ACC_N) {
    long long v = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][threadIdx.x];
    if (local)
      local[threadIdx.x] += v;
    else if (v != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(partials) +
                  ((size_t) prob * PARTIAL_SLOTS + (block & (PARTIAL_SLOTS - 1))) * ACC_N + threadIdx.x,
                (unsigned long long) v);
  }
}

}  // namespace

// ---- pieces shared by the step kernel and the deferred-search kernel --------------------------------------------
namespace {

struct QEntry {  // a moving point whose search did not settle inside the 3^DIM block (deferred to k_icp_step_queue)
  int i;         // index in the (Morton-sorted) moving array of the problem
  int r2;        // cube radius that still has to be scanned
  float best;    // best so far
  int bidx, bpos;
  float qx, qy, qz;  // the transformed point (saves the dependent reload + transform in the queue kernel)
  float ball2;       // squared radius of the ball the scan can be trimmed to (best so far, prior bound or the gate)
  int pad_;
};
static_assert(sizeof(QEntry) == 40, "host reserves 10 words per entry");

// finder->setLocalMapInSensor(robot_in_sensor * X), aligner_slice_processor_impl.cpp:35.  Computed once per iteration
// by the init / control kernels (ProblemState::Tf, Tfprev) -- two float64 3x4 compositions per WAVE were a fifth of the
// vector instructions of a converged pass -- and read here through the scalar cache.
__device__ __forceinline__ void finder_transform_of(const float* Sinv, int dim, const float* X, float* T) {
  if (dim == 3) {
    dm::se3_compose(Sinv, X, T);
  } else {
    float t9[9];
    dm::se2_compose(Sinv, X, t9);
    // spread the 3x3 into the 3x4 slots used by the kernels: rows [r0 r1 . t]
    T[0] = t9[0]; T[1] = t9[1]; T[2] = 0.f; T[3] = t9[2];
    T[4] = t9[3]; T[5] = t9[4]; T[6] = 0.f; T[7] = t9[5];
    T[8] = 0.f; T[9] = 0.f; T[10] = 1.f; T[11] = 0.f;
  }
}
__device__ __forceinline__ void load_T(const float* src, float* T) {
#pragma unroll
  for (int i = 0; i < 12; ++i) T[i] = src[i];
}

template <int DIM>
__device__ __forceinline__ void transform_point(const float* T, const float4 p, float& qx, float& qy, float& qz) {
  if constexpr (DIM == 3) {
    qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
    qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
    qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  } else {
    qx = (T[0] * p.x + T[1] * p.y) + T[3];
    qy = (T[4] * p.x + T[5] * p.y) + T[7];
    qz = 0.f;
  }
}

// Team-cooperative exact scan of a cube of radius sr cells: a team = TW consecutive lanes (TW = 64: the whole wave,
// TW = 16: four independent queries per wave) sharing ONE query (sqx..scz, sr are team-uniform; sr < 0 = idle team).
// The team's lanes fetch all row ranges of the cube together, prefix-sum the counts, take equal contiguous shares of
// the flattened candidate list and reduce to the lexicographic minimum of (d2, fixed index).  Every lane of the team
// returns the same (best, idx, pos); idx == NO_MATCH when the cube is empty.  lds: 4*TW + 8 ints per team.
template <int DIM, int TW>
__device__ __forceinline__ void coop_scan(const GridDev& g, int lane, int* lds_wave, float sqx, float sqy, float sqz,
                                          int scx, int scy, int scz, int sr, float sr2, float& wbest, int& widx,
                                          int& wpos, float& wexcl2) {
  constexpr int ROWS_PER_CHUNK = 2 * TW;
  const int lt   = lane & (TW - 1);
  const int team = lane / TW;
  int* flat      = lds_wave + team * (4 * TW + 8);  // flattened candidate offset at which each row starts (+ total)
  int* first     = flat + ROWS_PER_CHUNK + 4;       // sorted-array index of each row's first candidate
  // the cube of radius sr cells, trimmed to the ball of squared radius sr2 (the team's best candidate so far or the
  // gate): first to the ball's bounding box, then row by row to the chord of the ball through that row of cells
  const float rr = ball_radius(sr2);
  int x0, x1, y0, y1, z0 = 0, z1 = 0;
  axis_range(sqx, rr, g.ox, g.inv_h, scx - sr, scx + sr, g.nx, x0, x1);
  axis_range(sqy, rr, g.oy, g.inv_h, scy - sr, scy + sr, g.ny, y0, y1);
  if (DIM == 3) axis_range(sqz, rr, g.oz, g.inv_h, scz - sr, scz + sr, g.nz, z0, z1);
  // distance of the query to a row of cells, shrunk by 1% of a cell + 2e-6 of the coordinate magnitude: covers the
  // float32 rounding of the cell boundaries (a point's cell is floor(fl(fl(x - o) * inv_h)))
  auto row_range = [&](int rrow, int ny_r, int& rs, int& re) {
    const int y = y0 + rrow % ny_r, z = z0 + rrow / ny_r;
    const float ylo = g.oy + (float) y * g.h;
    float dy        = fmaxf(fmaxf(ylo - sqy, sqy - (ylo + g.h)), 0.f);
    dy              = fmaxf(dy - (0.01f * g.h + (fabsf(sqy) + rr) * 2e-6f), 0.f);
    float rem       = rr * rr - dy * dy;
    if (DIM == 3) {
      const float zlo = g.oz + (float) z * g.h;
      float dz        = fmaxf(fmaxf(zlo - sqz, sqz - (zlo + g.h)), 0.f);
      dz              = fmaxf(dz - (0.01f * g.h + (fabsf(sqz) + rr) * 2e-6f), 0.f);
      rem             = rem - dz * dz;
    }
    rs = re = 0;
    if (!(rem < 0.f)) {  // (+inf radius: rem = +inf or NaN -> full row)
      int xa = x0, xb = x1;
      if (rem < 3.0e38f) axis_range(sqx, ball_radius(rem), g.ox, g.inv_h, x0, x1, g.nx, xa, xb);
      if (xa <= xb) {
        const int row = (z * g.ny + y) * g.nx;
        rs = g.cell_start[row + xa];
        re = g.cell_start[row + xb + 1];
      }
    }
  };
  unsigned long long key = NO_KEY;
  float lb2              = INFINITY;
  const bool any = sr >= 0 && x0 <= x1 && y0 <= y1 && z0 <= z1;
  const int ny_r = any ? y1 - y0 + 1 : 1;
  const int rows = any ? ny_r * (z1 - z0 + 1) : 0;
  // all teams of the wave iterate the same number of chunks (wave-uniform control flow for the LDS syncs)
  int max_rows = rows;
  if (TW < 64) {
#pragma unroll
    for (int off = TW; off < 64; off <<= 1) max_rows = max(max_rows, __shfl_xor(max_rows, off));
  }
  for (int row0 = 0; row0 < max_rows; row0 += ROWS_PER_CHUNK) {
    // (1) every lane fetches the [start, end) ranges of two rows: all row fetches of the chunk in flight at once
    int sA = 0, eA = 0, sB = 0, eB = 0;
    const int rA = row0 + lt, rB = row0 + TW + lt;
    if (rA < rows) row_range(rA, ny_r, sA, eA);
    if (rB < rows) row_range(rB, ny_r, sB, eB);
    // (2) team prefix sum of both counts at once (packed in 64 bits)
    const unsigned long long pk = (unsigned long long) (unsigned) (eA - sA) | ((unsigned long long) (unsigned) (eB - sB) << 32);
    unsigned long long inc = pk;
#pragma unroll
    for (int off = 1; off < TW; off <<= 1) {
      unsigned long long t = __shfl_up(inc, off, TW);
      if (lt >= off) inc += t;
    }
    const unsigned long long tot = __shfl(inc, TW - 1, TW);
    const int totA = (int) (unsigned) tot, totB = (int) (tot >> 32);
    const unsigned long long exc = inc - pk;
    flat[lt]       = (int) (unsigned) exc;
    flat[TW + lt]  = totA + (int) (exc >> 32);
    first[lt]      = sA;
    first[TW + lt] = sB;
    if (lt == 0) flat[ROWS_PER_CHUNK] = totA + totB;
    wave_lds_sync();
    // (3) every lane takes an equal contiguous share of the flattened candidate list
    const int total = totA + totB;
    const int share = (total + TW - 1) / TW;
    int t           = lt * share;
    const int tend  = min(t + share, total);
    if (t < tend) {
      int lo = 0, hi = ROWS_PER_CHUNK - 1;  // last row whose start offset is <= t
#pragma unroll
      for (int it = 0; it < 7; ++it) {
        if ((1 << it) >= ROWS_PER_CHUNK) break;
        const int mid = (lo + hi + 1) >> 1;
        if (flat[mid] <= t) lo = mid; else hi = mid - 1;
      }
      int r    = lo;
      int next = flat[r + 1];
      while (t < tend) {
        while (t >= next) {
          ++r;
          next = flat[r + 1];
        }
        const int j   = first[r] + (t - flat[r]);
        const int run = min(next, tend) - t;  // candidates of this row in my share: consecutive in memory
        scan_range2<DIM>(g.pts, j, j + run, sqx, sqy, sqz, key, lb2);
        t += run;
      }
    }
    wave_lds_sync();
  }
  // team minimum of the 64-bit key (d2 bits, index): d2 >= 0 so the float bit pattern orders like the value
  unsigned long long kmin = key;
#pragma unroll
  for (int off = TW / 2; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(kmin, off);
    kmin = o < kmin ? o : kmin;
  }
  wpos  = 0;  // (not tracked: GridDev::pos_of gives the winner's position when it is needed)
  wbest = __uint_as_float((unsigned) (kmin >> 32));
  widx  = (int) (unsigned) kmin;
  // squared exclusion radius: the runner-up of the team (a lane that does not hold the winner contributes its own
  // best), capped by the region the scan was complete in: the ball and the cube
  float v = key == kmin ? lb2 : fminf(lb2, __uint_as_float((unsigned) (key >> 32)));
#pragma unroll
  for (int off = TW / 2; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  wexcl2 = fminf(v, fminf(sr2, bound2_of(max(sr, 0), g.h)));
}

// ============================================================================================
// Searches over CELL NEIGHBOUR LISTS (round 4): cnl_search, used by k_icp_step_cnl (the search passes) and by the converged-
// pass kernel for the points whose certificate failed.
//
// The grid kernels enumerate CELLS -- the rows of the 3^DIM block, then the shell of the 5^DIM cube, then cubes of growing
// radius -- and most of what they enumerate is empty: a cloud is a surface, one cell in six of a block holds points.  On the
// tiles the bookkeeping of (mostly empty) rows, the second staging for the shell and the cooperative scans behind it were
// 46 % of the first pass' vector instructions (profiles/archive/r3n_tile_kernel_knob_attribution.txt), the candidates themselves a
// fifth.  The fixed cloud is set once and searched by every iteration of every alignment, so the enumeration is done ONCE
// per fixed cloud: every cell of the (extended) grid gets the list of OCCUPIED cells that can hold a point within the
// extended gate of a query in it (GridDev::list_*, ~20 entries of 8 bytes per cell on C4, at most 16 points per entry),
// sorted by the class of the cell pair (squared separation in cells), then by centre distance.  A search is then:
//   0. the first entry (the query's own cell when it is occupied) is scanned; its best candidate (+ the pad that lets the
//      scan prove an exclusion margin), or the ball of the previous neighbour, is the ball everything else is pruned to;
//   A. the entry headers, four loads in flight per lane: stop at the first class whose separation exceeds the ball; an
//      entry whose cell lies outside the ball (three slab distances, in cell units) is dropped; the survivors of ALL lanes
//      are appended to ONE pool of the wave in LDS (ballot + mbcnt: compact, in order);
//   B. the wave works the pool off together, item s by lane s mod 64 -- a wave is as slow as its busiest lane, and with one
//      list per lane a single point without a neighbour inside the gate (all ~20 entries survive) kept the other 63 lanes
//      waiting: 28 group iterations per wave on the first pass for ~8 per lane on average (profiles/archive/r4c_*).  The worker
//      merges its item's (key, runner-up) into the owner's slot with two LDS atomics: min of the 64-bit key, and the loser
//      of that min -- the larger of the old and the new key -- is a candidate for the runner-up.
// TEAM lanes share one query (TEAM = 1: throughput, one query per lane; TEAM = 4: a single alignment is a chain of
// dependent round trips on a half-empty chip -- four lanes split the first entry's candidates and the headers, and a wave's
// pool is a quarter as long).
// One mechanism covers the whole gate ball: no second phase, no staging, no tile that may not fit, no deferred-search
// queue.  Same candidates-within-the-ball, same key minimum (d2 bits << 32 | index) => the same exact nearest neighbour
// as every other path; runner-up and completeness radius leave an exclusion radius that is as valid.
// ============================================================================================
namespace {

// -DSRRG2_CNL_REDEAL=1: the search pass re-deals the queries of a workgroup by the estimated length of their header walk
// (k_icp_step_cnl).  An experiment, OFF: exact, but 2-6 % slower per search pass on C4 (profiles/r6b_*): the walk lengths
// are spread evenly between 4 and 28 headers rather than in two groups, the estimate ranks them poorly (6.98 -> 6.26 steps
// per wave where a perfect sort would reach ~4.8) and the exchange costs what that saves.
#ifndef SRRG2_CNL_REDEAL
#define SRRG2_CNL_REDEAL 0
#endif
// -DSRRG2_CNL_STATS: census of the lane utilisation of the list search per iteration (tools/cnl_stats.py); compiled out of the
// product build
#ifdef SRRG2_CNL_STATS
__device__ unsigned long long g_cnl_stats[4 * 16];
#define CNL_STAT(k, v)                                                                                   \
  do {                                                                                                   \
    const unsigned long long v_ = (unsigned long long) (v); /* (by every lane: v may hold a ballot) */   \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_cnl_stats[(stat_it < 3 ? stat_it : 3) * 16 + (k)], v_);    \
  } while (0)
#else
#define CNL_STAT(k, v) do { } while (0)
#endif
constexpr int CNL_POOL = 512;  // survivor entries of a wave between two rounds of phase B (a step of phase A adds <= 256)

struct CnlWave {               // per wave, in LDS
  uint2 pool[CNL_POOL];        // {position of the first candidate, owner lane | count << 8}
  unsigned long long key[64];  // running minimum key of every lane's query
  unsigned b2[64];             // ... and the squared distance of its runner-up (float bits: d2 >= 0 orders like the bits)
  float q[3][64];              // the queries
};

// up to four candidates pts[j .. j + 3], the first `cnt` of them valid; lanes without work issue no load and run no test
// (reads up to 3 entries past the range, masked out: the sorted array has slack behind it).  Inside the branch everything
// is straight-line: four loads in flight, four distances, the masks as selects -- written with the ternaries of
// test_candidate2 the compiler branched around every candidate and sank the second load into its branch (two dependent
// round trips per group).
template <int DIM>
__device__ __forceinline__ void test_group_glb(const float4* __restrict__ pts, int j, int cnt, float qx, float qy, float qz,
                                               unsigned long long& bkey, float& b2) {
  if (cnt > 0) {
    const float4 a0 = pts[j], a1 = pts[j + 1], a2 = pts[j + 2], a3 = pts[j + 3];
    const float d0 = cand_d2<DIM>(a0, qx, qy, qz);
    float d1 = cand_d2<DIM>(a1, qx, qy, qz), d2 = cand_d2<DIM>(a2, qx, qy, qz), d3 = cand_d2<DIM>(a3, qx, qy, qz);
    unsigned i1 = (unsigned) __float_as_int(a1.w), i2 = (unsigned) __float_as_int(a2.w), i3 = (unsigned) __float_as_int(a3.w);
    d1 = cnt > 1 ? d1 : INFINITY;
    d2 = cnt > 2 ? d2 : INFINITY;
    d3 = cnt > 3 ? d3 : INFINITY;
    i1 = cnt > 1 ? i1 : (unsigned) NO_MATCH;
    i2 = cnt > 2 ? i2 : (unsigned) NO_MATCH;
    i3 = cnt > 3 ? i3 : (unsigned) NO_MATCH;
    b2   = __builtin_amdgcn_fmed3f(__uint_as_float((unsigned) (bkey >> 32)), b2, d0);
    bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d0) << 32) | (unsigned) __float_as_int(a0.w));
    b2   = __builtin_amdgcn_fmed3f(__uint_as_float((unsigned) (bkey >> 32)), b2, d1);
    bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d1) << 32) | i1);
    b2   = __builtin_amdgcn_fmed3f(__uint_as_float((unsigned) (bkey >> 32)), b2, d2);
    bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d2) << 32) | i2);
    b2   = __builtin_amdgcn_fmed3f(__uint_as_float((unsigned) (bkey >> 32)), b2, d3);
    bkey = key_min(bkey, ((unsigned long long) __float_as_uint(d3) << 32) | i3);
  }
}

// Phase B: the wave works off the n pooled survivors; results are merged into the owners' slots.
template <int DIM>
__device__ __forceinline__ void cnl_drain_pool(const float4* __restrict__ pts, CnlWave& w, int lane, int n, int stat_it = 3) {
  wave_lds_sync();  // (pool entries, queries and slots written by other lanes)
  CNL_STAT(6, n);
  for (int s0 = 0; s0 < n; s0 += 64) {
    CNL_STAT(7, 1);
    const int s     = s0 + lane;
    const bool have = s < n;
    uint2 it        = make_uint2(0u, 0u);
    if (have) it = w.pool[s];
    const int owner = (int) (it.y & 63u);
    int cnt         = have ? (int) (it.y >> 8) : 0;
    int j           = (int) it.x;
    const float qx = w.q[0][owner], qy = w.q[1][owner], qz = DIM == 3 ? w.q[2][owner] : 0.f;
    unsigned long long key = NO_KEY;
    float lb2              = INFINITY;
    while (__any(cnt > 0)) {
      CNL_STAT(8, 1);
      CNL_STAT(9, __popcll(__ballot(cnt > 0)));
      test_group_glb<DIM>(pts, j, cnt, qx, qy, qz, key, lb2);
      j += 4;
      cnt -= 4;
    }
    if (have) {
      const unsigned long long old   = atomicMin(&w.key[owner], key);
      const unsigned long long loser = old > key ? old : key;  // (not the minimum any more: a runner-up candidate)
      const unsigned l2              = min((unsigned) (loser >> 32), __float_as_uint(lb2));
      atomicMin(&w.b2[owner], l2);
    }
  }
  wave_lds_sync();
}

// The exact gated nearest neighbour of the wave's queries over the cell neighbour lists.  Called by every lane of the wave
// (wave-uniform control flow); the TEAM lanes of a team pass the same query and the same `need` / `r2box`.
//   r2box : squared radius of a ball known to hold the nearest neighbour (the previous neighbour's), or +inf
//   gfar  : squared radius the search reaches to (the gate, or the extended gate once priors exist)
// Returns, in the first lane of every team with `need`: bkey = the minimum key over the fixed points inside the ball
// (NO_KEY: none), b2 = squared distance of the runner-up among the points examined, L = squared radius inside which every
// fixed point was examined (every point not examined is farther than min(L, extended gate)).
// gl: the grid's list header -- the search-pass kernel gets it by value in its kernel arguments (scalar registers), the
// converged-pass kernel, which searches rarely, reads it through GridDev::lists.
// Phase 0 of cnl_search (below): the list of the query's cell and its first entry; e / eend = what is left of the list.
template <int DIM, int TEAM>
__device__ __forceinline__ void cnl_phase0(const GridDev& g, const GridLists& gl, int lane, bool need, float qx, float qy, float qz,
                                           float r2box, float gfar, unsigned long long& bkey, float& b2, float& L, int& e,
                                           int& eend, int stat_it = 3) {
  const int R  = gl.R;
  const int tl = lane & (TEAM - 1);   // lane within its team
  const int cx = cell_coord(qx, g.ox, g.inv_h);
  const int cy = cell_coord(qy, g.oy, g.inv_h);
  const int cz = DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0;
  const int lx = cx + R, ly = cy + R, lz = DIM == 3 ? cz + R : 0;
  // (a query outside the extended grid is farther than the extended gate from every fixed point: no list, no match)
  const bool inl = need && lx >= 0 && lx < gl.lnx && ly >= 0 && ly < gl.lny && lz >= 0 && lz < gl.lnz;
  e = 0;
  eend = 0;
  if (inl) {
    const int lc = (lz * gl.lny + ly) * gl.lnx + lx;
    e            = gl.start[lc];
    eend         = gl.start[lc + 1];
  }
  bkey = NO_KEY;
  b2   = INFINITY;
  L    = fminf(r2box, gfar);  // squared radius of the ball the search is pruned to
  // ---- phase 0: the first entry of the list; its best candidate (+ pad) bounds the rest
  {
    const bool p0 = e < eend;
    uint4 h0      = make_uint4(0u, 0u, 0u, 0u);
    if (p0) h0 = gl.ent[e];
    int j = (int) h0.x + 4 * tl, cnt = p0 ? (int) (h0.y & 15u) + 1 - 4 * tl : 0;
    CNL_STAT(10, __popcll(__ballot(need)));
    while (__any(cnt > 0)) {
      CNL_STAT(1, 1);
      CNL_STAT(2, __popcll(__ballot(cnt > 0)));
      test_group_glb<DIM>(g.pts, j, cnt, qx, qy, qz, bkey, b2);
      j += 4 * TEAM;
      cnt -= 4 * TEAM;
    }
#pragma unroll
    for (int off = 1; off < TEAM; off <<= 1) {  // the team's (key, runner-up)
      const unsigned long long ok = __shfl_xor(bkey, off);
      const float ob2             = __shfl_xor(b2, off);
      const unsigned long long lo = ok < bkey ? ok : bkey, hi = ok < bkey ? bkey : ok;
      b2   = fminf(fminf(b2, ob2), __uint_as_float((unsigned) (hi >> 32)));
      bkey = lo;
    }
    if (p0) {
      ++e;
      if (key_idx(bkey) != NO_MATCH) {
        const float rb = (sqrtf(key_best(bkey)) + (PAD_CAP + PAD_MIN) * g.h) * 1.00001f;
        L              = fminf(L, rb * rb);
      }
    }
  }
}

// Phases A and B of cnl_search: the headers of the rest of the list, the pooled survivors.  Takes what cnl_phase0 left --
// (bkey, b2, L, e, eend) of the query (qx, qy, qz) -- whichever lane computed it (k_icp_step_cnl re-deals the queries of a
// workgroup between the two).  Returns bkey / b2 in the first lane of every team.
template <int DIM, int TEAM>
__device__ __forceinline__ void cnl_phaseAB(const GridDev& g, const GridLists& gl, CnlWave& w, int lane, float qx, float qy,
                                            float qz, unsigned long long& bkey, float& b2, float L, int e, int eend,
                                            int stat_it = 3) {
  const int R  = gl.R;
  const int tl = lane & (TEAM - 1);   // lane within its team
  const int owner_lane = lane - tl;   // the team's slot
  // the query in cell units: cell c + fraction u along every axis
  const float ux0 = (qx - g.ox) * g.inv_h, uy0 = (qy - g.oy) * g.inv_h, uz0 = DIM == 3 ? (qz - g.oz) * g.inv_h : 0.f;
  const int cx = cell_coord(qx, g.ox, g.inv_h);
  const int cy = cell_coord(qy, g.oy, g.inv_h);
  const int cz = DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0;
  if (tl == 0) {
    w.key[lane]  = bkey;
    w.b2[lane]   = __float_as_uint(b2);
    w.q[0][lane] = qx;
    w.q[1][lane] = qy;
    if (DIM == 3) w.q[2][lane] = qz;
  }
  // ---- phase A: the headers.  Every entry carries the box of ITS points in sixteenths of a cell, as bytes (offset + R) * 16 +
  // sixteenths per axis (k_cnl_boxes): lower corner lo, upper corner hi.  With u = the query's fraction of its own cell the box
  // is max(lo / 16 - (R + u), (R + u) - hi / 16, 0) away along an axis -- shrunk by m = 1.1 % of a cell + 1e-6 of the
  // coordinate, which covers the float32 rounding of cell assignments (a point's cell is floor(fl(fl(x - o) * inv_h))) and of
  // these expressions; the box itself is rounded outwards.  Two conversions (v_cvt_f32_ubyte), two subtractions from
  // per-lane constants, one v_max3 per axis; everything in sixteenths (squared distances x 256).
  const float Lc  = L * 1.00002f;
  const float Lcc = ((Lc * g.inv_h) * g.inv_h * 1.0001f) * 256.f;
  int mmax = -1;  // largest class whose cells can reach into the ball
#pragma unroll
  for (int m = 0; m <= CNL_MAX_CLASS; ++m) mmax += gl.cls_b2[m] <= Lc ? 1 : 0;
  const float fR = (float) R;
  const float mx = 0.011f + fabsf(ux0) * 1e-6f, my = 0.011f + fabsf(uy0) * 1e-6f, mz = 0.011f + fabsf(uz0) * 1e-6f;
  const float ux = ux0 - (float) cx, uy = uy0 - (float) cy, uz = uz0 - (float) cz;
  // t = max(lo - ca, cb - hi, 0):  lo / 16 - (R + u + m),  (R + u - m) - hi / 16, in sixteenths
  const float cax = ((fR + ux) + mx) * 16.f, cbx = ((fR + ux) - mx) * 16.f;
  const float cay = ((fR + uy) + my) * 16.f, cby = ((fR + uy) - my) * 16.f;
  const float caz = ((fR + uz) + mz) * 16.f, cbz = ((fR + uz) - mz) * 16.f;
  int pool_n = 0;  // (wave-uniform)
  e += 4 * tl;     // the lanes of a team take the headers four at a time, in turn
  CNL_STAT(0, 1);
  CNL_STAT(11, __popcll(__ballot(e < eend)));
  while (__any(e < eend)) {
    CNL_STAT(3, 1);
    CNL_STAT(4, __popcll(__ballot(e < eend)));
    uint4 hd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hd[k] = make_uint4(0u, 0xf0u, 0u, 0u);  // (class 15: beyond every ball)
      if (e + k < eend) hd[k] = gl.ent[e + k];
    }
    bool stop = e >= eend;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned code = hd[k].y;
      // (sorted by class: nothing behind the first entry of a class beyond the ball can reach into it)
      stop = stop || (int) ((code >> 4) & 15u) > mmax;
      const unsigned blo = hd[k].z, bhi = hd[k].w;
      const float lx_ = (float) (blo & 255u), ly_ = (float) ((blo >> 8) & 255u);
      const float hx_ = (float) (bhi & 255u), hy_ = (float) ((bhi >> 8) & 255u);
      const float tx = fmaxf(fmaxf(lx_ - cax, cbx - hx_), 0.f);
      const float ty = fmaxf(fmaxf(ly_ - cay, cby - hy_), 0.f);
      float d2c      = __fmaf_rn(ty, ty, tx * tx);
      if (DIM == 3) {
        const float lz_ = (float) ((blo >> 16) & 255u), hz_ = (float) ((bhi >> 16) & 255u);
        const float tz  = fmaxf(fmaxf(lz_ - caz, cbz - hz_), 0.f);
        d2c             = __fmaf_rn(tz, tz, d2c);
      }
      const bool surv             = !stop && !(d2c > Lcc);
      const unsigned long long bm = __ballot(surv);
      if (surv) {
        const int slot = pool_n + (int) __builtin_amdgcn_mbcnt_hi((unsigned) (bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) bm, 0u));
        w.pool[slot]   = make_uint2(hd[k].x, (unsigned) owner_lane | (((code & 15u) + 1u) << 8));
      }
      pool_n += __popcll(bm);
    }
    e = stop ? eend : min(e + 4 * TEAM, eend);
    if (pool_n > CNL_POOL - 256) {  // (the next step may add 4 x 64)
      cnl_drain_pool<DIM>(g.pts, w, lane, pool_n, stat_it);
      pool_n = 0;
    }
  }
  cnl_drain_pool<DIM>(g.pts, w, lane, pool_n, stat_it);  // (also the barrier before the slots are read back)
  bkey = w.key[owner_lane];
  b2   = __uint_as_float(w.b2[owner_lane]);
}

template <int DIM, int TEAM>
__device__ __forceinline__ void cnl_search(const GridDev& g, const GridLists& gl, CnlWave& w, int lane, bool need, float qx,
                                           float qy, float qz, float r2box, float gfar, unsigned long long& bkey, float& b2,
                                           float& L, int stat_it = 3) {
  int e, eend;
  cnl_phase0<DIM, TEAM>(g, gl, lane, need, qx, qy, qz, r2box, gfar, bkey, b2, L, e, eend, stat_it);
  cnl_phaseAB<DIM, TEAM>(g, gl, w, lane, qx, qy, qz, bkey, b2, L, e, eend, stat_it);
}

}  // namespace

// gates, normals, residual rows, factor arithmetic and the per-point outputs of ONE searched moving point
template <int DIM, bool PLANE>
__device__ __forceinline__ void finish_point(const SliceDev& S, const float* T, int rk, float thr, float kk, double scale,
                                             bool inrange, bool active, int gi, int oi, const float4 p, float qx, float qy,
                                             float qz, float best, int bidx, int bpos, float excl, bool kept,
                                             const float4 kept_f, const float4 kept_n, const float4 early_nm, bool have_nm,
                                             long long (&acc)[ACC_N]) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  const GridDev& g   = S.grid;
  int match     = -1;
  float resp    = 0.f;
  uint8_t fstat = SRRG2_FACTOR_SUPPRESSED;
  float4 fm     = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));  // the nearest fixed point {x, y, z, index}
  float4 nf     = make_float4(0.f, 0.f, 0.f, 0.f);                        // ... and its normal
  if (inrange) {
    if (active) {
      bool found = bidx != NO_MATCH && best <= g.gate2;
      if (KNOB(S.tune, 8)) found = false;
      // (a kept neighbour comes with its normal, loaded together with the prior: one round trip less on the chain; the
      // normal of a neighbour beyond the gate is fetched too: it is stored with the neighbour for the next iteration)
      if (!kept && bidx != NO_MATCH) bpos = g.pos_of[bidx];  // (searches do not track positions: one 4-byte gather here)
      if (bidx != NO_MATCH && (PLANE || S.use_normal_gate))
        nf = KNOB(S.tune, 32) ? make_float4(0.f, 0.f, 1.f, 0.f) : (kept ? kept_n : g.nrm[bpos]);
      // (with nf: one round trip, not one after the normal gate; a kept neighbour comes with its coordinates)
      if (bidx != NO_MATCH) fm = kept ? kept_f : g.pts[bpos];
      if (found && S.use_normal_gate) {
        const float4 nm = KNOB(S.tune, 128) ? make_float4(0.f, 0.f, 1.f, 0.f) : (have_nm ? early_nm : S.mnrm[gi]);
        float dot;
        if constexpr (DIM == 3) {
          float rx = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
          float ry = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
          float rz = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
          dot      = (nf.x * rx + nf.y * ry) + nf.z * rz;
        } else {
          float rx = T[0] * nm.x + T[1] * nm.y;
          float ry = T[4] * nm.x + T[5] * nm.y;
          dot      = nf.x * rx + nf.y * ry;
        }
        if (!(dot > S.normal_cos)) found = false;
      }
      if (found) {
        match          = bidx;
        resp           = best;
        const float4 f = fm;
        float J[ROWS][D];
        float e[ROWS];
        if constexpr (DIM == 3) {
          float m[ROWS][3];
          if (PLANE) {
            e[0]    = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
            m[0][0] = (T[0] * nf.x + T[4] * nf.y) + T[8] * nf.z;
            m[0][1] = (T[1] * nf.x + T[5] * nf.y) + T[9] * nf.z;
            m[0][2] = (T[2] * nf.x + T[6] * nf.y) + T[10] * nf.z;
          } else {
            const float q[3]  = {qx, qy, qz};
            const float ff[3] = {f.x, f.y, f.z};
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              e[r]    = q[r] - ff[r];
              m[r][0] = T[r * 4 + 0];
              m[r][1] = T[r * 4 + 1];
              m[r][2] = T[r * 4 + 2];
            }
          }
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            J[r][0]     = m[r][0];
            J[r][1]     = m[r][1];
            J[r][2]     = m[r][2];
            J[r][D - 3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
            J[r][D - 2] = kk * (p.z * m[r][0] - p.x * m[r][2]);
            J[r][D - 1] = kk * (p.x * m[r][1] - p.y * m[r][0]);
          }
        } else {
          float m[ROWS][2];
          if (PLANE) {
            e[0]    = nf.x * (qx - f.x) + nf.y * (qy - f.y);
            m[0][0] = T[0] * nf.x + T[4] * nf.y;
            m[0][1] = T[1] * nf.x + T[5] * nf.y;
          } else {
            const float q[2]  = {qx, qy};
            const float ff[2] = {f.x, f.y};
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
              e[r]    = q[r] - ff[r];
              m[r][0] = T[r * 4 + 0];
              m[r][1] = T[r * 4 + 1];
            }
          }
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            J[r][0] = m[r][0];
            J[r][1] = m[r][1];
            J[r][2] = m[r][1] * p.x - m[r][0] * p.y;
          }
        }
        fstat = factor_accumulate<D, ROWS>(J, e, false, rk, thr, scale, KNOB(S.tune, 64), acc);
      }
    }
    if (!kept) {  // (a skipped search keeps its neighbour: only the exclusion radius changes)
      S.prev_pos[gi] = (active && bidx != NO_MATCH) ? bpos : -1;
      if (!S.gather_prev) {  // (with gather_prev the position is all the next pass reads)
        S.prev_f[gi] = fm;  // (.w = NO_MATCH: none)
        if (PLANE || S.use_normal_gate) S.prev_n[gi] = nf;
      }
    }
    S.prev_m[gi] = excl;
    // (the correspondence record {match, resp, fstat} is not stored: k_icp_outputs derives it on demand)
    (void) match; (void) resp; (void) fstat;
  }

}

}  // namespace

// One moving point per thread.  Lanes whose nearest neighbour is not settled by the 3^DIM block are either pushed
// to the per-problem queue (S.queue != null; k_icp_step_queue finishes them with all waves of the chip sharing
// the work) or, without a queue, handled here: radius-2 scan per lane, then wave-cooperative scans.
// (body shared by k_icp_step -- 4 waves, tile = blockIdx.x, sums to the global slot sets -- and k_icp_small -- NW waves
// looping over the tiles of its problem, state and sums in LDS)
// What the body needs of the state: from ProblemState (k_icp_step, k_icp_small), or -- fused control steps, round 6: the grid
// kernel of a first compute() on a new fixed cloud carries the control step of the previous iteration like the list kernels do --
// from the published record (step_view_of_record below the fused prologue's definitions).
struct StepView {
  float T[12], Tprev[12];
  int kexp, nstats;
  bool phase1, prior, qmode;
};
__device__ __forceinline__ void step_view_of_state(const SliceDev& S, const ProblemState* st, StepView& v) {
#pragma unroll
  for (int i = 0; i < 12; ++i) v.T[i] = st->Tf[S.slice_idx][i];
#pragma unroll
  for (int i = 0; i < 12; ++i) v.Tprev[i] = st->Tfprev[S.slice_idx][i];
  v.kexp   = st->kexp[S.slice_idx];
  v.nstats = st->nstats;
  v.phase1 = st->phase == 1;
  v.prior  = st->nstats > 0 || st->phase == 1;
  v.qmode  = st->qmode[S.slice_idx] != 0;
}
template <int DIM, bool PLANE, int NW>
__device__ __forceinline__ void icp_step_body(const SliceDev& S, const ProblemDev pd, const StepView& sv, int prob,
                                              int tile, int ntiles, int nprob, long long* local_sums) {
  float T[12];
  load_T(sv.T, T);
  const int kexp     = sv.kexp;
  const double scale = dm::pow2(kexp);
  const int rk       = (sv.phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  const float b2_1   = bound2_of(1, g.h);
  const bool use_prior = sv.prior && !(S.tune & 4);
  // open points go to the deferred-search queue, or are finished here (no queue; or the control kernel saw that the
  // queue stays nearly empty: st->qmode, mirrored by the host which then drops the deferred-search launch)
  const bool use_q = S.queue != nullptr && sv.qmode;
  // Scans of points without a neighbour inside the gate reach 25% beyond it once a prior exists: what they find (a
  // point just outside the gate, or nothing) then certifies "no match" for the following iterations without a search.
  const float gfar = (use_prior && !(S.tune & 65536)) ? g.gate2_ext : g.gate2;
  // cube radius that covers the ball of radius sqrt(gfar) (both computed by the host with the same bound)
  const int rfar = (use_prior && !(S.tune & 65536)) ? g.rmax : g.rfar_gate;
  float Tprev[12];  // the finder transform of the previous iteration (its queries: q' = Tprev * p)
  load_T(sv.Tprev, Tprev);

  long long acc[ACC_N];
#pragma unroll
  for (int a = 0; a < ACC_N; ++a) acc[a] = 0;
#ifdef SRRG2_TIMELINE
  unsigned long long* tl = (S.dbg && sv.nstats < 32 && !sv.phase1)
                             ? S.dbg + (((size_t) sv.nstats * nprob + prob) * ntiles * NW + tile * NW + (threadIdx.x >> 6)) * 16
                             : nullptr;
#else
  unsigned long long* tl = nullptr;
#endif
  STAMP(tl, 0);  // state + transform loaded

  const int i        = tile * (NW * 64) + threadIdx.x;
  const int lane     = threadIdx.x & 63;
  const int wid      = threadIdx.x >> 6;
  const bool inrange = i < pd.nm;
  const int gi       = pd.moff + (inrange ? i : 0);
  float4 p           = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pf          = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
  float4 pn          = make_float4(0.f, 0.f, 0.f, 0.f);   // its normal
  float4 pnm         = make_float4(0.f, 0.f, 0.f, 0.f);   // this moving point's normal
  float pm           = 0.f;
  if (inrange) {
    p = S.mpts[gi];
    if (use_prior) {  // previous nearest neighbour {x, y, z, index}, its normal, exclusion radius: independent loads
      pm   = S.prev_m[gi];
      if (S.use_normal_gate) pnm = S.mnrm[gi];
      if (S.gather_prev) {  // (batches: through its position in the L2-resident fixed cloud, 8 instead of 40 streamed bytes)
        const int ppos = S.prev_pos[gi];
        if (ppos >= 0 && ppos < S.grid.n) {
          pf = S.grid.pts[ppos];
          if (PLANE || S.use_normal_gate) pn = S.grid.nrm[ppos];
        }
      } else {
        pf = S.prev_f[gi];
        if (PLANE || S.use_normal_gate) pn = S.prev_n[gi];
      }
    }
  }
  const bool has_prev = __float_as_int(pf.w) != NO_MATCH;
  // moving points are stored spatially sorted; p.w carries the caller's index within the problem
  const int oi      = pd.moff + __float_as_int(p.w);
  const bool active = inrange && finite3(p.x, p.y, p.z);
  float qx = 0.f, qy = 0.f, qz = 0.f;
  int cx = 0, cy = 0, cz = 0;
  float best = INFINITY;
  int bidx = NO_MATCH, bpos = 0;
  int r2 = 0;  // > 1: this lane needs a cube of radius r2
  float excl = 0.f;  // exclusion radius left behind for the next iteration (0: none)
  float r2box  = INFINITY;  // squared radius of the ball the first search phase can be trimmed to
  float ball2  = INFINITY;  // ... and the wider scans
  bool skipped = false;     // the previous nearest neighbour is provably still the nearest
  float excl_wide = 0.f;
  float pad       = PAD_MIN * g.h;  // margin of the scans beyond the nearest neighbour (grows with the motion)
  __shared__ int coop_lds[NW][288];  // (64-lane scans need 264 ints, four 16-lane teams 4 x 72)
  STAMP(tl, 1);  // moving point + prior loaded
  if (active && !KNOB(S.tune, 16)) {
    transform_point<DIM>(T, p, qx, qy, qz);
    // Temporal coherence, exactly.  The previous iteration of this compute() left, per moving point, its nearest
    // neighbour f* and an exclusion radius m: no OTHER fixed point lies within m of the previous query q'.
    //  (a) d(q, f*) + |q - q'| < m  =>  every other point is farther than f* (triangle inequality): f* is still the
    //      unique nearest neighbour and the search is skipped; the exclusion radius shrinks by |q - q'|.
    //  (b) otherwise the search is trimmed to the ball of radius d(q, f*) + pad around q: it contains f*, finds the
    //      exact (d2, index) minimum, and the runner-up distance / the ball / the 3^DIM block leave a new m behind.
    // All float32 roundings here are relative (~1e-6: differences and squares of exact float coordinates); the
    // factors 1.00001 / 0.99999 keep every inequality on the safe side.  The oracle searches from scratch.
    if (use_prior && has_prev) {
      unsigned long long k1 = NO_KEY;
      test_candidate<DIM>(pf, qx, qy, qz, true, k1);
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      const float d1 = sqrtf(key_best(k1));
      if (d1 * 1.00001f + dl * 1.00001f < pm * 0.99999f && !(S.tune & 4096)) {
        skipped = true;
        best    = key_best(k1);
        bidx    = key_idx(k1);
        bpos    = 0;  // (unused: a kept neighbour comes with its coordinates and normal)
        // (the stored radius shrinks by the motion and by its own rounding -- 0.9999999 > one ulp -- not by the
        // comparison's 1e-5 safety factor, which alone ate the margin of a few points per iteration, each of which then
        // made its whole wave pay a search)
        excl    = pm * 0.9999999f - dl * 1.00001f;
      } else {
        pad            = fminf(2.f * dl, PAD_CAP * g.h) + PAD_MIN * g.h;
        const float rr = (d1 + pad) * 1.00001f;
        r2box          = fminf(rr * rr, gfar);
      }
    } else if (use_prior && !has_prev && pm > 0.f && !(S.tune & (4096 | 65536))) {
      // (c) no fixed point at all within m of q' (an empty scan left m behind): if gate + |q - q'| < m there is still
      //     none within the gate of q: no match, no search
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      if (sqrtf(g.gate2) * 1.00001f + dl * 1.00001f < pm * 0.99999f) {
        skipped = true;
        excl    = pm * 0.9999999f - dl * 1.00001f;  // (best = inf, bidx = NO_MATCH: stays "none")
      }
    }
  }
  // Converged iterations: a handful of lanes per wave still need a search (near-ties, lost exclusion radius) and would
  // make the whole wave pay the search latency.  Hand them to the deferred-search kernel, which is launched anyway.
  // running (key, runner-up) pair and completeness radius of the first search phase (continued by the shell scan)
  unsigned long long bkey = NO_KEY;
  float b2                = INFINITY;
  float complete2         = INFINITY;
  bool straggler = false;
  if (use_q && use_prior && rfar > 1 && !KNOB(S.tune, 16) && !(S.tune & 8192)) {
    const bool need  = active && !skipped;
    const int n_need = __popcll(__ballot(need));
    if (need && n_need <= 8) {
      straggler = true;
      ball2     = fminf(r2box, gfar);
      r2        = ball2 <= bound2_of(2, g.h) ? 2 : rfar;
    }
  }
  // Workgroup-level compaction of the per-lane scans (batch mode): a wave pays for its slowest lane, and on the passes
  // before convergence only a fraction of the lanes still needs the first phase (the rest hold a certificate), and ~40 %
  // of those the second.  The lanes that need a scan leave their query in LDS; the first n threads of the workgroup each
  // take one and hand the (key, runner-up, completeness) triple back through LDS: whole waves skip the phase.
  // (everything here is uniform over the workgroup: use_q and the tune word are per problem)
  const bool compact = !use_q && !(S.tune & 2097152);
  constexpr int COMPACT_MAX = NW * 48;  // above this the scan runs in place (nothing to win, two barriers to lose)
  __shared__ float4 s_q[NW * 64];
  __shared__ unsigned long long s_key[NW * 64];
  __shared__ float s_b2[NW * 64], s_c2[NW * 64];
  __shared__ int s_wq[NW * 64], s_cnt1[NW], s_cnt2[NW];
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool need1 = active && !KNOB(S.tune, 16) && !skipped && !straggler;
  {
    bool in_lds = false;
    int n1 = 0;
    if (compact) {
      const unsigned long long m = __ballot(need1);
      if (lane == 0) s_cnt1[wid] = __popcll(m);
      __syncthreads();
      int base = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int c = s_cnt1[w];
        if (w < wid) base += c;
        n1 += c;
      }
      in_lds = n1 > 0 && n1 <= COMPACT_MAX;
      if (in_lds) {
        if (need1) {
          s_wq[base + __popcll(m & below)] = threadIdx.x;
          s_q[threadIdx.x]                 = make_float4(qx, qy, qz, r2box);
        }
        __syncthreads();
      }
    }
    bool wvalid = need1;
    float wqx = qx, wqy = qy, wqz = qz, wr2 = r2box;
    int owner = threadIdx.x;
    if (in_lds) {
      wvalid = (int) threadIdx.x < n1;
      if (wvalid) {
        owner          = s_wq[threadIdx.x];
        const float4 v = s_q[owner];
        wqx = v.x; wqy = v.y; wqz = v.z; wr2 = v.w;
      }
    }
    unsigned long long rkey = NO_KEY;
    float rb2 = INFINITY, rc2 = INFINITY;
    if (wvalid) {
      const int wcx = cell_coord(wqx, g.ox, g.inv_h);
      const int wcy = cell_coord(wqy, g.oy, g.inv_h);
      const int wcz = DIM == 3 ? cell_coord(wqz, g.oz, g.inv_h) : 0;
      scan_radius1<DIM>(g, wqx, wqy, wqz, wcx, wcy, wcz, wr2, rkey, rb2, rc2, tl);
    }
    if (in_lds) {
      if (wvalid) {
        s_key[owner] = rkey;
        s_b2[owner]  = rb2;
        s_c2[owner]  = rc2;
      }
      __syncthreads();
      if (need1) {
        rkey = s_key[threadIdx.x];
        rb2  = s_b2[threadIdx.x];
        rc2  = s_c2[threadIdx.x];
      }
    }
    if (need1) {
      bkey      = rkey;
      b2        = rb2;
      complete2 = rc2;
      cx = cell_coord(qx, g.ox, g.inv_h);
      cy = cell_coord(qy, g.oy, g.inv_h);
      cz = DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0;
      best = key_best(bkey);
      bidx = key_idx(bkey);
      const bool found1 = bidx != NO_MATCH && best <= gfar;  // a candidate that can bound the wider scan
      ball2             = gfar;
      if (!(found1 && best <= b2_1) && rfar > 1) {
        r2 = rfar;
        if (found1) {
          // the wider scan covers the ball of the candidate plus a pad, so that it leaves an exclusion radius
          // larger than the neighbour's distance behind (otherwise these points would be searched every iteration)
          const float rr = (sqrtf(best) + pad) * 1.00001f;
          ball2          = fminf(rr * rr, gfar);
          r2             = 1;
          _Pragma("clang loop vectorize(disable) unroll(disable)") while (r2 < rfar && bound2_of(r2, g.h) < ball2) ++r2;
        }
      } else {
        // settled: the scan was complete inside min(ball, block); nothing but the winner is closer than this
        excl = sqrtf(fminf(fminf(b2, complete2), b2_1)) * 0.99999f;
      }
    }
  }
#ifdef SRRG2_TIMELINE
  if (tl) {  // per-wave census of this iteration (tools/timeline.py prints the sums)
    const unsigned long long b_skip_a = __ballot(skipped && bidx != NO_MATCH), b_skip_c = __ballot(skipped && bidx == NO_MATCH);
    const unsigned long long b_strag = __ballot(straggler), b_near = __ballot(!straggler && r2 == 2), b_far = __ballot(!straggler && r2 > 2);
    const unsigned long long b_act = __ballot(active);
    if ((threadIdx.x & 63) == 0) {
      tl[8] = __popcll(b_act); tl[9] = __popcll(b_skip_a); tl[10] = __popcll(b_skip_c); tl[11] = __popcll(b_strag);
      tl[12] = __popcll(b_near); tl[13] = __popcll(b_far);
    }
  }
#endif
  STAMP(tl, 4);  // first search phase done
  bool deferred = false;
  if (use_q) {
    // push the open lanes to the problem's queues: near entries (radius 2) grow from the front of the problem's
    // region, far entries (radius > 2) from its back; one atomic per wave and kind, entries in lane order
    const unsigned long long need_near = __ballot(r2 == 2);
    const unsigned long long need_far  = __ballot(r2 > 2);
    // one atomic per BLOCK and kind (same-address device atomics serialise at ~10 ns each: per-wave atomics cost
    // ~15 us on the misaligned first iteration of C2)
    __shared__ int q_cnt[2][NW], q_base[2];
    if (lane == 0) {
      q_cnt[0][wid] = __popcll(need_near);
      q_cnt[1][wid] = __popcll(need_far);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      int tot = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += q_cnt[threadIdx.x][w];
      q_base[threadIdx.x] = tot ? atomicAdd(&S.qcount[2 * prob + threadIdx.x], tot) : 0;
    }
    __syncthreads();
    {
      int base_near = q_base[0], base_far = q_base[1];
      for (int w = 0; w < wid; ++w) {
        base_near += q_cnt[0][w];
        base_far += q_cnt[1][w];
      }
      if (r2 > 1) {
        QEntry e;
        e.i = i; e.r2 = r2; e.best = best; e.bidx = bidx; e.bpos = bpos;
        e.qx = qx; e.qy = qy; e.qz = qz;
        e.ball2 = ball2; e.pad_ = 0;
        QEntry* qbase = reinterpret_cast<QEntry*>(S.queue) + pd.moff;
        if (r2 == 2)
          qbase[base_near + __popcll(need_near & below)] = e;
        else
          qbase[pd.nm - 1 - (base_far + __popcll(need_far & below))] = e;
        deferred = true;
      }
    }
  } else {
    // radius-2 cube per lane, then the cooperative scan for what is still open
    // the shell of the radius-2 cube, continuing the first phase's (key, runner-up) pair: both phases together have met
    // every fixed point of cube(2) within min(ball, first phase's completeness radius) exactly once
    // (SRRG2_AMD_TUNE bit 1048576: the whole cube from scratch, as round 1 did)
    const bool need2 = r2 > 1 && rfar >= 2 && !KNOB(S.tune, 2);
    {
      bool in_lds = false;
      int n2 = 0;
      if (compact) {
        const unsigned long long m = __ballot(need2);
        if (lane == 0) s_cnt2[wid] = __popcll(m);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const int c = s_cnt2[w];
          if (w < wid) base += c;
          n2 += c;
        }
        in_lds = n2 > 0 && n2 <= COMPACT_MAX;
        if (in_lds) {
          if (need2) {
            s_wq[base + __popcll(m & below)] = threadIdx.x;
            s_q[threadIdx.x]                 = make_float4(qx, qy, qz, ball2);
            s_key[threadIdx.x]               = bkey;
            s_b2[threadIdx.x]                = b2;
          }
          __syncthreads();
        }
      }
      bool wvalid = need2;
      float wqx = qx, wqy = qy, wqz = qz, wball = ball2;
      unsigned long long rkey = bkey;
      float rb2 = b2;
      int owner = threadIdx.x;
      if (in_lds) {
        wvalid = (int) threadIdx.x < n2;
        if (wvalid) {
          owner          = s_wq[threadIdx.x];
          const float4 v = s_q[owner];
          wqx = v.x; wqy = v.y; wqz = v.z; wball = v.w;
          rkey = s_key[owner];
          rb2  = s_b2[owner];
        }
      }
      if (wvalid) {
        const int wcx = cell_coord(wqx, g.ox, g.inv_h);
        const int wcy = cell_coord(wqy, g.oy, g.inv_h);
        const int wcz = DIM == 3 ? cell_coord(wqz, g.oz, g.inv_h) : 0;
        if (S.tune & 1048576) {
          rkey = NO_KEY;
          rb2  = INFINITY;
          scan_radius2<DIM>(g, wqx, wqy, wqz, wcx, wcy, wcz, wball, rkey, rb2);
        } else {
          scan_shell2<DIM>(g, wqx, wqy, wqz, wcx, wcy, wcz, wball, rkey, rb2);
        }
      }
      if (in_lds) {
        __syncthreads();  // (every worker has read its item: the slots may be overwritten)
        if (wvalid) {
          s_key[owner] = rkey;
          s_b2[owner]  = rb2;
        }
        __syncthreads();
        if (need2) {
          rkey = s_key[threadIdx.x];
          rb2  = s_b2[threadIdx.x];
        }
      }
      if (need2) {
        bkey = rkey;
        b2   = rb2;
        if (S.tune & 1048576) complete2 = INFINITY;
        best = key_best(bkey);
        bidx = key_idx(bkey);
        const bool found2 = bidx != NO_MATCH && best <= gfar;
        if ((found2 && best <= bound2_of(2, g.h)) || rfar == 2) {
          r2        = 0;
          excl_wide = sqrtf(fminf(fminf(fminf(b2, ball2), complete2), bound2_of(2, g.h))) * 0.99999f;
        } else {
          r2    = rfar;
          ball2 = gfar;
          if (found2) {
            const float rr = (sqrtf(best) + pad) * 1.00001f;
            ball2          = fminf(rr * rr, gfar);
            r2             = 2;
            _Pragma("clang loop vectorize(disable) unroll(disable)") while (r2 < rfar && bound2_of(r2, g.h) < ball2) ++r2;
          }
        }
      }
    }
    unsigned long long need = __ballot(r2 > 1);
    if (KNOB(S.tune, 1)) need = 0;
    // Four open points per pass, a team of 16 lanes each (the cost of a cooperative scan is its fixed part -- row ranges,
    // prefix sum, two LDS round trips -- not its candidates: on the misaligned first pass of a batch a wave has a dozen
    // such points, most of them with nothing inside the gate); one 64-lane scan at a time when a single point is left.
    while (need) {
      if (__popcll(need) == 1 || (S.tune & 524288)) {
        const int src = __ffsll((long long) need) - 1;
        need &= need - 1;
        float wbest, wexcl2;
        int widx, wpos;
        coop_scan<DIM, 64>(g, lane, coop_lds[wid], __shfl(qx, src), __shfl(qy, src), __shfl(qz, src), __shfl(cx, src),
                           __shfl(cy, src), __shfl(cz, src), __shfl(r2, src), __shfl(ball2, src), wbest, widx, wpos, wexcl2);
        if (lane == src) excl_wide = sqrtf(wexcl2) * 0.99999f;
        if (lane == src && (wbest < best || (wbest == best && widx < bidx))) {
          best = wbest;
          bidx = widx;
        }
        continue;
      }
      int src[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        src[t] = need ? __ffsll((long long) need) - 1 : -1;
        need &= need - 1;
      }
      const int team = lane >> 4;
      const int mine = team == 0 ? src[0] : (team == 1 ? src[1] : (team == 2 ? src[2] : src[3]));
      const int from = mine >= 0 ? mine : lane;
      float wbest, wexcl2;
      int widx, wpos;
      // (every shuffle outside the select: the owner of a point may sit in a team that has nothing to do this pass)
      const int sr_from = __shfl(r2, from);
      coop_scan<DIM, 16>(g, lane, coop_lds[wid], __shfl(qx, from), __shfl(qy, from), __shfl(qz, from), __shfl(cx, from),
                         __shfl(cy, from), __shfl(cz, from), mine >= 0 ? sr_from : -1, __shfl(ball2, from), wbest, widx,
                         wpos, wexcl2);
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // the result of team t goes to the lane that owns the point
        const float rb = __shfl(wbest, 16 * t), re = __shfl(wexcl2, 16 * t);
        const int ri = __shfl(widx, 16 * t);
        if (lane == src[t]) {
          excl_wide = sqrtf(re) * 0.99999f;
          if (rb < best || (rb == best && ri < bidx)) {
            best = rb;
            bidx = ri;
          }
        }
      }
    }
  }
  STAMP(tl, 5);  // open lanes pushed / searched
  if (excl_wide != 0.f || r2 != 0) excl = excl_wide;  // finished by the wider scans above
  if (!deferred)
    finish_point<DIM, PLANE>(S, T, rk, thr, kk, scale, inrange, active, gi, oi, p, qx, qy, qz, best, bidx, bpos, excl, skipped,
                             pf, pn, pnm, use_prior && S.use_normal_gate, acc);
  STAMP(tl, 6);  // gates, rows, factor arithmetic, per-point outputs
  block_reduce_store<NW>(acc, S.partials, prob, tile, local_sums);
  STAMP(tl, 7);  // reduction + atomics issued
}

template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_step(SliceDev S, const ProblemDev* __restrict__ probs,
                                                  ProblemState* __restrict__ states) {
  const int prob   = blockIdx.y + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  ProblemState* st = &states[prob];
  if (st->done || st->finished) return;
  StepView sv;
  step_view_of_state(S, st, sv);
  icp_step_body<DIM, PLANE, 4>(S, probs[prob], sv, prob, blockIdx.x, gridDim.x, gridDim.y, nullptr);
}

// ============================================================================================
// Search passes of batches with the neighbourhood of every WAVE staged in LDS (k_icp_step_tile).
//
// What bounds k_icp_step in the throughput regime (profiles/archive/r2u_*, r3a_*): the texture path is busy ~70 % of a search
// pass -- it moves 64 bytes per clock of REQUESTED lane data, so a 16-byte candidate gather costs a wave 16 of its cycles
// however well its lanes coalesce (a finer Morton order of the moving cloud changes nothing: profiles/archive/r3a_ab_msort.txt) --
// the vector ALUs ~65 %, at four waves per SIMD.  Here every wave fetches the candidates it needs ONCE, coalesced, into
// LDS: the lanes of a wave are neighbours in space (Morton order), their 3^DIM blocks overlap, the union is a box of a
// few dozen rows of cells.  The lanes then scan their own rows with ds_read_b128 (LDS: 128 bytes per clock, its own
// pipe, ~100 cycles of latency instead of ~500-2000), without the prefetch registers the global path needs to hide
// that latency, without the workgroup-level compaction and its barriers.  Same candidates, same arithmetic, same exact
// minimum of the key: bit-identical results.  A wave whose box does not fit (TILE_CAP candidates, TILE_ROWS rows,
// TILE_NX cells per row: Morton-curve jumps, very dense spots) takes the global path for that phase.
// ============================================================================================
namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 64;   // rows of cells (y, z) of a wave's box (one lane per row)
constexpr int TILE_NX   = 12;   // cells per row
constexpr int TILE_CSW  = TILE_NX + 1;
constexpr int TILE_LIST = 12;   // ranges a lane can queue before it scans them

template <int CAP>
struct WaveTile {             // per wave, in LDS
  static_assert(CAP + 4 <= 512, "range starts are packed in 9 bits");
  f4v pts[CAP + 4];           // the candidates of the box, row after row (+ slack: groups of four read past the end, masked)
  unsigned short cs[TILE_ROWS * TILE_CSW];  // cs[r][k]: offset in pts[] at which cell X0 + k of row r starts
  union {
    struct {                     // while the tile is staged:
      int rowbase[TILE_ROWS];     //   index of cell (X0, y, z) in the grid
      int rowA[TILE_ROWS];        //   position of the row's first candidate in grid.pts
      int rowoff[TILE_ROWS + 1];  //   ... and in pts[] (exclusive prefix sum of the row sizes; [nrows] = total)
    } st;
    unsigned short list[TILE_LIST * 64];  // while it is scanned: list[k][lane] = start | count << 9 of the lane's k-th range
  } u;
};

// One group of four candidates from LDS, the first `cnt` of them valid (cnt >= 4: all).  (Measured without the masks --
// a group that runs past its range then tests a few more real fixed points, which cannot change the minimum: 2 % fewer
// vector instructions, but the candidates seen early shrink the ball the rest of the scan is pruned to, and with it the
// exclusion radius left behind: the first converged pass went from 34 to 52 us.  profiles/archive/r3h_*)
template <int DIM>
__device__ __forceinline__ void test_group_lds(const f4v* pts, int j, int cnt, float qx, float qy, float qz,
                                               unsigned long long& bkey, float& b2) {
  const f4v a0 = pts[j], a1 = pts[j + 1], a2 = pts[j + 2], a3 = pts[j + 3];
  test_candidate2<DIM>(make_float4(a0.x, a0.y, a0.z, a0.w), qx, qy, qz, cnt > 0, bkey, b2);
  test_candidate2<DIM>(make_float4(a1.x, a1.y, a1.z, a1.w), qx, qy, qz, cnt > 1, bkey, b2);
  test_candidate2<DIM>(make_float4(a2.x, a2.y, a2.z, a2.w), qx, qy, qz, cnt > 2, bkey, b2);
  test_candidate2<DIM>(make_float4(a3.x, a3.y, a3.z, a3.w), qx, qy, qz, cnt > 3, bkey, b2);
}

// candidates [j, e) of the tile, four at a time, the last group masked (no separate tail code: with the rows of 64 lanes
// of different lengths the tail block ran for nearly every row)
template <int DIM>
__device__ __forceinline__ void scan_range_lds(const f4v* pts, int j, int e, float qx, float qy, float qz,
                                               unsigned long long& bkey, float& b2) {
  for (; j < e; j += 4) test_group_lds<DIM>(pts, j, e - j, qx, qy, qz, bkey, b2);
}

// Every lane walks ITS OWN list of n ranges as one flattened sequence of groups of four: the wave iterates as often as
// its busiest lane has groups -- not, as with one loop per row, the sum over the rows of the busiest lane of each row
// (measured on C4, tools/tile_stats.py: 10.7 instead of 19.4 group iterations per wave and search pass).
template <int DIM>
__device__ __forceinline__ void scan_list_lds(const f4v* pts, const unsigned short* list, int lane, int n, float qx, float qy,
                                              float qz, unsigned long long& bkey, float& b2) {
  int cur = 0, j = 0, cnt = 0;
  if (n > 0) {
    const unsigned u = list[lane];
    j   = (int) (u & 511u);
    cnt = (int) (u >> 9);
  }
  while (__any(cnt > 0)) {
    test_group_lds<DIM>(pts, j, cnt, qx, qy, qz, bkey, b2);  // (a lane that has finished: group 0, all masked)
    j += 4;
    cnt -= 4;
    if (cnt <= 0) {
      cnt = 0;
      j   = 0;
      if (++cur < n) {
        const unsigned u = list[cur * 64 + lane];
        j   = (int) (u & 511u);
        cnt = (int) (u >> 9);
      }
    }
  }
}

// Queue the range [rs, re) of the tile on the lane's list.  An entry holds up to 127 candidates; the (rare: a very dense
// spot) longer range is scanned on the spot instead.  The caller leaves room for the entry (n < TILE_LIST).
template <int DIM>
__device__ __forceinline__ void list_push(const f4v* pts, unsigned short* list, int lane, int& n, int rs, int re, float qx,
                                          float qy, float qz, unsigned long long& bkey, float& b2) {
  if (re - rs > 127) {
    scan_range_lds<DIM>(pts, rs, re, qx, qy, qz, bkey, b2);
  } else if (rs < re) {
    list[n * 64 + lane] = (unsigned short) (rs | ((re - rs) << 9));
    ++n;
  }
}

// wave-wide minimum / maximum of a per-lane int, as a wave-uniform value (invalid lanes pass the neutral element)
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = min(v, __shfl_xor(v, off));
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = max(v, __shfl_xor(v, off));
  return __builtin_amdgcn_readfirstlane(v);
}

struct TileBox {
  int X0, Y0, Z0, nxb, nyb, nzb;
  bool ok;
  int why;  // 0 = staged, 1 = too many cells per row, 2 = too many rows, 3 = too many candidates (statistics builds)
  int total;
};

// -DSRRG2_TILE_STATS: census of the wave tiles per iteration (tools/tile_stats.py); compiled out of the product build
#ifdef SRRG2_TILE_STATS
__device__ unsigned long long g_tile_stats[4 * 16];
#define TILE_STAT(it, k, v)                                                                             \
  do {                                                                                                  \
    const unsigned long long v_ = (unsigned long long) (v); /* (by every lane: v may hold a ballot) */  \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_tile_stats[((it) < 3 ? (it) : 3) * 16 + (k)], v_);        \
  } while (0)
#else
#define TILE_STAT(it, k, v) do { } while (0)
#endif

// Stage the box [x0, x1] x [y0, y1] x [z0, z1] (union over the lanes with `want`) of the grid into the wave's tile.
// Wave-uniform control flow; returns ok = false (nothing staged) when the box does not fit.
template <int CAP>
__device__ __forceinline__ TileBox stage_tile(const GridDev& g, WaveTile<CAP>& t, int lane, bool want, int x0, int x1, int y0,
                                              int y1, int z0, int z1) {
  TileBox b;
  b.ok = false;
  b.why = 1;
  b.total = 0;
  // the box: six wave reductions, two per step (min of v and of -v packed in one 64-bit value would need a 64-bit min:
  // the lower corner is reduced as it is, the upper corner negated -- one min chain of three values each)
  int lo0 = want ? x0 : 0x7fffffff, lo1 = want ? y0 : 0x7fffffff, lo2 = want ? z0 : 0x7fffffff;
  int hi0 = want ? -x1 : 0x7fffffff, hi1 = want ? -y1 : 0x7fffffff, hi2 = want ? -z1 : 0x7fffffff;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    lo0 = min(lo0, __shfl_xor(lo0, off));
    lo1 = min(lo1, __shfl_xor(lo1, off));
    lo2 = min(lo2, __shfl_xor(lo2, off));
    hi0 = min(hi0, __shfl_xor(hi0, off));
    hi1 = min(hi1, __shfl_xor(hi1, off));
    hi2 = min(hi2, __shfl_xor(hi2, off));
  }
  b.X0 = __builtin_amdgcn_readfirstlane(lo0);
  b.Y0 = __builtin_amdgcn_readfirstlane(lo1);
  b.Z0 = __builtin_amdgcn_readfirstlane(lo2);
  const int X1 = -__builtin_amdgcn_readfirstlane(hi0), Y1 = -__builtin_amdgcn_readfirstlane(hi1),
            Z1 = -__builtin_amdgcn_readfirstlane(hi2);
  b.nxb = X1 - b.X0 + 1;
  b.nyb = Y1 - b.Y0 + 1;
  b.nzb = Z1 - b.Z0 + 1;
  if (b.X0 == 0x7fffffff || b.nxb > TILE_NX) return b;
  b.why = 2;
  if (b.nyb > TILE_ROWS || b.nzb > TILE_ROWS || b.nyb * b.nzb > TILE_ROWS) return b;
  const int nrows = b.nyb * b.nzb;
  // one lane per row: where the row starts in the grid and in the sorted cloud, how many candidates it holds
  int rbase = 0, A = 0, cnt = 0;
  if (lane < nrows) {
    const int rz = (int) (((float) lane + 0.5f) * (1.0f / (float) b.nyb));  // lane / nyb (exact: lane < 64)
    const int ry = lane - rz * b.nyb;
    rbase        = ((b.Z0 + rz) * g.ny + (b.Y0 + ry)) * g.nx + b.X0;
    A            = g.cell_start[rbase];
    cnt          = g.cell_start[rbase + b.nxb] - A;
  }
  int incl = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(incl, off);
    if (lane >= off) incl += u;
  }
  const int total = __shfl(incl, 63);
  b.why   = 3;
  b.total = total;
  if (total > CAP) return b;
  if (lane < nrows) {
    t.u.st.rowbase[lane] = rbase;
    t.u.st.rowA[lane]    = A;
    t.u.st.rowoff[lane]  = incl - cnt;
  }
  if (lane == 0) t.u.st.rowoff[nrows] = total;
  wave_lds_sync();
  // the cell table: four rows per step, 16 lanes each (TILE_CSW <= 16 entries per row, contiguous in the grid)
  {
    const int k = lane & 15;
    for (int r = lane >> 4; r < nrows; r += 4)
      if (k <= b.nxb)
        t.cs[r * TILE_CSW + k] = (unsigned short) (g.cell_start[t.u.st.rowbase[r] + k] - t.u.st.rowA[r] + t.u.st.rowoff[r]);
  }
  // the candidates: the flattened list, 64 at a time (coalesced within a row); the row of a slot by bisection
  for (int s0 = 0; s0 < total; s0 += 64) {
    const int s = s0 + lane;
    if (s < total) {
      int lo = 0, hi = nrows - 1;  // last row whose offset is <= s
#pragma unroll
      for (int it = 0; it < 6; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.u.st.rowoff[mid] <= s) lo = mid; else hi = mid - 1;
      }
      const float4 c = g.pts[t.u.st.rowA[lo] + (s - t.u.st.rowoff[lo])];
      f4v v;
      v.x = c.x; v.y = c.y; v.z = c.z; v.w = c.w;
      t.pts[s] = v;
    }
  }
  wave_lds_sync();  // (from here on the row tables are dead: their memory becomes the lanes' range lists)
  b.ok  = true;
  b.why = 0;
  return b;
}

// squared distance of q to the slab of cells with coordinate c along one axis, shrunk by the rounding of the cell
// boundaries (1 % of a cell + 2e-6 of the coordinate magnitude: a point's cell is floor(fl(fl(x - o) * inv_h)))
__device__ __forceinline__ float slab_dist2(float q, int c, float o, float h, float rb) {
  const float lo = o + (float) c * h;
  float d        = fmaxf(fmaxf(lo - q, q - (lo + h)), 0.f);
  d              = fmaxf(d - (0.01f * h + (fabsf(q) + rb) * 2e-6f), 0.f);
  return d * d;
}

// The first search phase (scan_radius1) on a staged tile: the 3^DIM cells around the query trimmed to the ball of squared
// radius r2box, the row through the query's own cell first, the other rows pruned by the distance of that row's best and
// then walked as ONE flattened list.  (x0 .. z1: the lane's cell ranges, as computed for the staging; the same candidates
// as scan_radius1, hence the same key minimum; runner-up and completeness radius are as valid.)  Called by every lane
// of the wave: lanes without a search (want == false) walk empty ranges.
template <int DIM, int CAP>
__device__ __forceinline__ void scan_radius1_tile(const GridDev& g, WaveTile<CAP>& t, const TileBox& b, int lane, float qx,
                                                  float qy, float qz, int cx, int cy, int cz, float r2box, bool want, int x0,
                                                  int x1, int y0, int y1, int z0, int z1, unsigned long long& bkey, float& b2,
                                                  float& complete2, bool centre_only = false) {
  constexpr int NROWS = DIM == 3 ? 9 : 3;
  constexpr int RC    = DIM == 3 ? 4 : 1;
  complete2 = r2box;
  const int kx0 = x0 - b.X0, kx1 = x1 + 1 - b.X0;
  auto row_range = [&](int r, int& rs, int& re) {
    const int y = cy + (r % 3) - 1;
    const int z = DIM == 3 ? cz + (r / 3) - 1 : 0;
    rs = re = 0;
    if (want && y >= y0 && y <= y1 && z >= z0 && z <= z1) {
      const int tr = (z - b.Z0) * b.nyb + (y - b.Y0);
      rs = t.cs[tr * TILE_CSW + kx0];
      re = t.cs[tr * TILE_CSW + kx1];
    }
  };
  // the row through the query's own cell first: its best candidate (+ a pad, so that the scan still proves an exclusion
  // margin) prunes the other rows.  (With the ball of the previous neighbour as bound all nine rows could be pruned
  // against that ball and walked as one list: measured 5 % MORE vector instructions on the passes with a prior.)
  {
    int rs, re;
    row_range(RC, rs, re);
    scan_range_lds<DIM>(t.pts, rs, re, qx, qy, qz, bkey, b2);
    if (key_idx(bkey) != NO_MATCH) {
      const float rb = (sqrtf(key_best(bkey)) + (PAD_CAP + PAD_MIN) * g.h) * 1.00001f;
      complete2      = fminf(complete2, rb * rb);
    }
  }
  if (centre_only) return;  // (timing knob)
  // the other rows: every point of a row is at least (dy, dz) away -- three slab distances per axis instead of one
  // rectangle distance per row
  const bool prune = complete2 < 3.0e38f;
  const float rb   = prune ? sqrtf(complete2) : 0.f;
  const float c2   = complete2 * 1.00002f;
  float dy2[3], dz2[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dy2[k] = slab_dist2(qy, cy + k - 1, g.oy, g.h, rb);
    if (DIM == 3) dz2[k] = slab_dist2(qz, cz + k - 1, g.oz, g.h, rb);
  }
  int n = 0;
#pragma unroll
  for (int r = 0; r < NROWS; ++r) {
    if (r == RC) continue;
    const float rem = (c2 - dy2[r % 3]) - dz2[DIM == 3 ? r / 3 : 0];
    int rs = 0, re = 0;
    if (!(prune && rem < 0.f)) row_range(r, rs, re);
    list_push<DIM>(t.pts, t.u.list, lane, n, rs, re, qx, qy, qz, bkey, b2);  // (<= 8 entries)
  }
  static_assert(TILE_LIST >= 8, "the first phase queues up to eight rows");
  wave_lds_sync();
  scan_list_lds<DIM>(t.pts, t.u.list, lane, n, qx, qy, qz, bkey, b2);
}

// The shell of the radius-2 cube (scan_shell2) on a staged tile; x0 .. z1: the lane's ranges of the 5^DIM cube trimmed to
// the ball (as computed for the staging).  The lane's ranges are queued TILE_LIST at a time and walked as flattened lists;
// rows farther from the query than the ball are skipped (the chord test of coop_scan: scan_shell2 only trims to the box
// of the ball).  Called by every lane of the wave.
template <int DIM, int CAP>
__device__ __forceinline__ void scan_shell2_tile(const GridDev& g, WaveTile<CAP>& t, const TileBox& b, int lane, float qx,
                                                 float qy, float qz, int cx, int cy, int cz, float ball2, bool want, int x0,
                                                 int x1, int y0, int y1, int z0, int z1, unsigned long long& bkey, float& b2) {
  const int bx0 = max(cx - 1, 0), bx1 = min(cx + 1, g.nx - 1);
  const float rr = ball_radius(ball2);
  const float r2 = rr * rr;
  // walk the rows (y, z) of the lane's box in a fixed order; `pos` = the next row to queue
  const int ny_r = want ? y1 - y0 + 1 : 0, nz_r = want ? z1 - z0 + 1 : 0;
  const int nrow = ny_r * nz_r;
  int pos = 0;
  int y = y0, z = z0;
  while (__any(pos < nrow)) {
    int n = 0;
    while (pos < nrow && n + 2 <= TILE_LIST) {  // (a row queues at most two ranges)
      const bool zin   = DIM == 3 ? (z >= cz - 1 && z <= cz + 1) : true;
      const bool inner = zin && y >= cy - 1 && y <= cy + 1;
      float rem        = r2 - slab_dist2(qy, y, g.oy, g.h, rr);
      if (DIM == 3) rem -= slab_dist2(qz, z, g.oz, g.h, rr);
      if (!(rem < 0.f)) {
        const int tr = (z - b.Z0) * b.nyb + (y - b.Y0);
        const unsigned short* row = t.cs + tr * TILE_CSW - b.X0;  // row[x] = offset of cell x
        const int la = x0, lb = inner ? min(x1, bx0 - 1) : x1;
        const int ra = max(x0, bx1 + 1), rbx = x1;
        if (la <= lb) list_push<DIM>(t.pts, t.u.list, lane, n, row[la], row[lb + 1], qx, qy, qz, bkey, b2);
        if (inner && ra <= rbx) list_push<DIM>(t.pts, t.u.list, lane, n, row[ra], row[rbx + 1], qx, qy, qz, bkey, b2);
      }
      ++pos;
      if (++y > y1) {
        y = y0;
        ++z;
      }
    }
    wave_lds_sync();
    scan_list_lds<DIM>(t.pts, t.u.list, lane, n, qx, qy, qz, bkey, b2);
    wave_lds_sync();  // (the list is rewritten by the next round)
  }
}

}  // namespace

// One moving point per thread, one alignment per blockIdx.y; no deferred-search queue (the throughput regime).
template <int DIM, bool PLANE, int CAP>
__global__ __launch_bounds__(256) void k_icp_step_tile(SliceDev S, const ProblemDev* __restrict__ probs,
                                                       ProblemState* __restrict__ states) {
  constexpr int NW = 4;
  const int prob   = blockIdx.y + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  const ProblemState* st = &states[prob];
  if (st->done || st->finished) return;
  const ProblemDev pd = probs[prob];
  const int tile      = blockIdx.x;
  float T[12];
  load_T(st->Tf[S.slice_idx], T);
  const double scale = dm::pow2(st->kexp[S.slice_idx]);
  const int rk       = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  const float b2_1   = bound2_of(1, g.h);
  const bool use_prior = (st->nstats > 0 || st->phase == 1) && !(S.tune & 4);
  const float gfar = (use_prior && !(S.tune & 65536)) ? g.gate2_ext : g.gate2;
  const int rfar   = (use_prior && !(S.tune & 65536)) ? g.rmax : g.rfar_gate;
  float Tprev[12];
  load_T(st->Tfprev[S.slice_idx], Tprev);

  // the tile of a wave and the row tables of its cooperative scans are never live together
  union WaveLds {
    WaveTile<CAP> tile;
    int coop[288];
  };
  __shared__ WaveLds wlds[NW];

  const int i        = tile * (NW * 64) + threadIdx.x;
  const int lane     = threadIdx.x & 63;
  const int wid      = threadIdx.x >> 6;
  const bool inrange = i < pd.nm;
  const int gi       = pd.moff + (inrange ? i : 0);
  float4 p           = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pf          = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
  float4 pn          = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pnm         = make_float4(0.f, 0.f, 0.f, 0.f);
  float pm           = 0.f;
  if (inrange) {
    p = S.mpts[gi];
    if (use_prior) {
      pm = S.prev_m[gi];
      if (S.use_normal_gate) pnm = S.mnrm[gi];
      if (S.gather_prev) {  // (as in icp_step_body)
        const int ppos = S.prev_pos[gi];
        if (ppos >= 0 && ppos < S.grid.n) {
          pf = S.grid.pts[ppos];
          if (PLANE || S.use_normal_gate) pn = S.grid.nrm[ppos];
        }
      } else {
        pf = S.prev_f[gi];
        if (PLANE || S.use_normal_gate) pn = S.prev_n[gi];
      }
    }
  }
  const bool has_prev = __float_as_int(pf.w) != NO_MATCH;
  const int oi        = pd.moff + __float_as_int(p.w);
  const bool active   = inrange && finite3(p.x, p.y, p.z);
  float qx = 0.f, qy = 0.f, qz = 0.f;
  float best = INFINITY;
  int bidx = NO_MATCH, bpos = 0;
  int r2 = 0;
  float excl = 0.f, r2box = INFINITY, ball2 = INFINITY, excl_wide = 0.f;
  bool skipped = false;
  float pad    = PAD_MIN * g.h;
  if (active) {
    transform_point<DIM>(T, p, qx, qy, qz);
    // temporal coherence, exactly as in icp_step_body: (a) the previous neighbour is provably still the nearest,
    // (b) the search is trimmed to its ball, (c) still nothing within the gate
    if (use_prior && has_prev) {
      unsigned long long k1 = NO_KEY;
      test_candidate<DIM>(pf, qx, qy, qz, true, k1);
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      const float d1 = sqrtf(key_best(k1));
      if (d1 * 1.00001f + dl * 1.00001f < pm * 0.99999f && !(S.tune & 4096)) {
        skipped = true;
        best    = key_best(k1);
        bidx    = key_idx(k1);
        excl    = pm * 0.9999999f - dl * 1.00001f;
      } else {
        pad            = fminf(2.f * dl, PAD_CAP * g.h) + PAD_MIN * g.h;
        const float rr = (d1 + pad) * 1.00001f;
        r2box          = fminf(rr * rr, gfar);
      }
    } else if (use_prior && !has_prev && pm > 0.f && !(S.tune & (4096 | 65536))) {
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      if (sqrtf(g.gate2) * 1.00001f + dl * 1.00001f < pm * 0.99999f) {
        skipped = true;
        excl    = pm * 0.9999999f - dl * 1.00001f;
      }
    }
  }
  const int cx = cell_coord(qx, g.ox, g.inv_h);
  const int cy = cell_coord(qy, g.oy, g.inv_h);
  const int cz = DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0;
  unsigned long long bkey = NO_KEY;
  float b2 = INFINITY, complete2 = INFINITY;
  // ---- first phase: the 3^DIM block, trimmed to the ball
  // (timing knobs, profiling builds only: 16 = no search, 33554432 = stage the tiles without scanning them,
  // 67108864 = the row through the query's cell only, 2 = no shell phase, 8 = no linearisation)
  const bool need1 = active && !skipped && !KNOB(S.tune, 16);
  if (__any(need1)) {
    const float rr = ball_radius(r2box);
    int x0, x1, y0, y1, z0 = 0, z1 = 0;
    axis_range(qx, rr, g.ox, g.inv_h, cx - 1, cx + 1, g.nx, x0, x1);
    axis_range(qy, rr, g.oy, g.inv_h, cy - 1, cy + 1, g.ny, y0, y1);
    if (DIM == 3) axis_range(qz, rr, g.oz, g.inv_h, cz - 1, cz + 1, g.nz, z0, z1);
    const bool want = need1 && x0 <= x1 && y0 <= y1 && z0 <= z1;
    const TileBox tb = stage_tile<CAP>(g, wlds[wid].tile, lane, want, x0, x1, y0, y1, z0, z1);
    TILE_STAT(st->nstats, 0, 1);
    TILE_STAT(st->nstats, 1 + tb.why, 1);
    TILE_STAT(st->nstats, 5, tb.total);
    TILE_STAT(st->nstats, 6, tb.nyb * tb.nzb);
    TILE_STAT(st->nstats, 7, __popcll(__ballot(need1)));
    if (tb.ok && !KNOB(S.tune, 33554432))
      scan_radius1_tile<DIM, CAP>(g, wlds[wid].tile, tb, lane, qx, qy, qz, cx, cy, cz, r2box, want, x0, x1, y0, y1, z0, z1, bkey,
                                  b2, complete2, KNOB(S.tune, 67108864));
    else if (need1)
      scan_radius1<DIM>(g, qx, qy, qz, cx, cy, cz, r2box, bkey, b2, complete2);
    if (need1) {
      best = key_best(bkey);
      bidx = key_idx(bkey);
      const bool found1 = bidx != NO_MATCH && best <= gfar;
      ball2             = gfar;
      if (!(found1 && best <= b2_1) && rfar > 1) {
        r2 = rfar;
        if (found1) {
          const float rr2 = (sqrtf(best) + pad) * 1.00001f;
          ball2           = fminf(rr2 * rr2, gfar);
          r2              = 1;
          _Pragma("clang loop vectorize(disable) unroll(disable)") while (r2 < rfar && bound2_of(r2, g.h) < ball2) ++r2;
        }
      } else {
        excl = sqrtf(fminf(fminf(b2, complete2), b2_1)) * 0.99999f;
      }
    }
  }
  // ---- second phase: the shell of the radius-2 cube, continuing the (key, runner-up) pair of the first
  const bool need2 = r2 > 1 && rfar >= 2 && !KNOB(S.tune, 2);
  if (__any(need2)) {
    wave_lds_sync();  // (the first phase's tile is dead)
    const float rr = ball_radius(ball2);
    int x0, x1, y0, y1, z0 = 0, z1 = 0;
    axis_range(qx, rr, g.ox, g.inv_h, cx - 2, cx + 2, g.nx, x0, x1);
    axis_range(qy, rr, g.oy, g.inv_h, cy - 2, cy + 2, g.ny, y0, y1);
    if (DIM == 3) axis_range(qz, rr, g.oz, g.inv_h, cz - 2, cz + 2, g.nz, z0, z1);
    const bool want = need2 && x0 <= x1 && y0 <= y1 && z0 <= z1;
    const TileBox tb = stage_tile<CAP>(g, wlds[wid].tile, lane, want, x0, x1, y0, y1, z0, z1);
    TILE_STAT(st->nstats, 8, 1);
    TILE_STAT(st->nstats, 9 + tb.why, 1);
    TILE_STAT(st->nstats, 13, tb.total);
    TILE_STAT(st->nstats, 14, __popcll(__ballot(need2)));
    if (tb.ok)
      scan_shell2_tile<DIM, CAP>(g, wlds[wid].tile, tb, lane, qx, qy, qz, cx, cy, cz, ball2, want, x0, x1, y0, y1, z0, z1, bkey, b2);
    else if (need2)
      scan_shell2<DIM>(g, qx, qy, qz, cx, cy, cz, ball2, bkey, b2);
    if (need2) {
      best = key_best(bkey);
      bidx = key_idx(bkey);
      const bool found2 = bidx != NO_MATCH && best <= gfar;
      if ((found2 && best <= bound2_of(2, g.h)) || rfar == 2) {
        r2        = 0;
        excl_wide = sqrtf(fminf(fminf(fminf(b2, ball2), complete2), bound2_of(2, g.h))) * 0.99999f;
      } else {
        r2    = rfar;
        ball2 = gfar;
        if (found2) {
          const float rr2 = (sqrtf(best) + pad) * 1.00001f;
          ball2           = fminf(rr2 * rr2, gfar);
          r2              = 2;
          _Pragma("clang loop vectorize(disable) unroll(disable)") while (r2 < rfar && bound2_of(r2, g.h) < ball2) ++r2;
        }
      }
    }
  }
  // ---- what is still open: cooperative scans (teams of 16 lanes, four points per pass), as in icp_step_body
  unsigned long long need = __ballot(r2 > 1);
  TILE_STAT(st->nstats, 15, __popcll(need));
  if (need) wave_lds_sync();  // (the tile is dead: its memory now holds the row tables of the scans)
  while (need) {
    if (__popcll(need) == 1) {
      const int src = __ffsll((long long) need) - 1;
      need &= need - 1;
      float wbest, wexcl2;
      int widx, wpos;
      coop_scan<DIM, 64>(g, lane, wlds[wid].coop, __shfl(qx, src), __shfl(qy, src), __shfl(qz, src), __shfl(cx, src),
                         __shfl(cy, src), __shfl(cz, src), __shfl(r2, src), __shfl(ball2, src), wbest, widx, wpos, wexcl2);
      if (lane == src) excl_wide = sqrtf(wexcl2) * 0.99999f;
      if (lane == src && (wbest < best || (wbest == best && widx < bidx))) {
        best = wbest;
        bidx = widx;
      }
      continue;
    }
    int src[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      src[t] = need ? __ffsll((long long) need) - 1 : -1;
      need &= need - 1;
    }
    const int team = lane >> 4;
    const int mine = team == 0 ? src[0] : (team == 1 ? src[1] : (team == 2 ? src[2] : src[3]));
    const int from = mine >= 0 ? mine : lane;
    float wbest, wexcl2;
    int widx, wpos;
    const int sr_from = __shfl(r2, from);
    coop_scan<DIM, 16>(g, lane, wlds[wid].coop, __shfl(qx, from), __shfl(qy, from), __shfl(qz, from), __shfl(cx, from),
                       __shfl(cy, from), __shfl(cz, from), mine >= 0 ? sr_from : -1, __shfl(ball2, from), wbest, widx, wpos,
                       wexcl2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float rb = __shfl(wbest, 16 * t), re = __shfl(wexcl2, 16 * t);
      const int ri = __shfl(widx, 16 * t);
      if (lane == src[t]) {
        excl_wide = sqrtf(re) * 0.99999f;
        if (rb < best || (rb == best && ri < bidx)) {
          best = rb;
          bidx = ri;
        }
      }
    }
  }
  if (excl_wide != 0.f || r2 != 0) excl = excl_wide;
  long long acc[ACC_N];
#pragma unroll
  for (int a = 0; a < ACC_N; ++a) acc[a] = 0;
  finish_point<DIM, PLANE>(S, T, rk, thr, kk, scale, inrange, active, gi, oi, p, qx, qy, qz, best, bidx, bpos, excl, skipped,
                           pf, pn, pnm, use_prior && S.use_normal_gate, acc);
  block_reduce_store<NW>(acc, S.partials, prob, tile, nullptr);
}

// ============================================================================================
// The converged pass.  From the second iteration of a compute() on, nearly every moving point keeps its nearest
// neighbour (exclusion-radius certificate, see icp_step_body): such a pass is a streaming kernel -- load the point, its
// previous neighbour and the neighbour's normal, prove that the neighbour is unchanged, linearise, reduce -- and in
// k_icp_step it pays for the generality of the search code around it (941 vector instructions per wave and point
// on converged C4 passes, the pass is VALU-issue bound; profiles/archive/r2a).  k_icp_step_fast is that pass alone:
//   * PPT moving points per thread share ONE 32-value transposing reduction (the reduction is ~230 of the ~500
//     vector instructions of a point; the accumulators are plain int64 registers, no search state is live);
//   * the fixed-point terms stay in their fma-biased form (bit pattern of FX_MAGIC + integer) and are summed as
//     integers; the bias FX_MAGIC_BITS x (number of contributions) is removed once per workgroup from the counters;
//   * points whose certificate fails are handed to the deferred-search kernel (S.queue) exactly like the stragglers of
//     k_icp_step, or -- without a queue -- searched here by the whole wave (coop_scan), one at a time.
// Same per-point arithmetic, same exact integer sums: bit-identical results.
// ============================================================================================
// ============================================================================================
// Fused control steps (FusedCtl, device_types.h): the control step of an ICP iteration on ONE wave, in the prologue of the
// first pass kernel of the next iteration.  wave_control() is k_icp_control's body (control_body below: the reference's
// multi_aligner_impl.cpp:106-126) for aligners whose cue slices have a nearest-neighbour or projective finder -- with up to two
// prior slices next to them, wave_prior: the PRIORS instantiations -- (no given correspondences, no deferred-search queue: those
// keep the control launch), written so that it
// can live inside a pass kernel: no LDS, no barrier, and matrices spread over the LANES of the wave (H(r, c) in lane
// r D + c, vectors in lanes 0 .. D - 1) instead of over 238 registers of one thread.  Every arithmetic statement is the one
// of control_body / dm::solve / dm::box_plus / dm::se3_compose with the same operands in the same order, executed by the
// lane that owns the result (operands fetched by v_readlane / ds_bpermute): the same bits.
// ============================================================================================
// -DSRRG2_PASS_TIMELINE: where the time of a fused pass launch goes (tools/pass_timeline.py): constant-rate clock readings
// (100 MHz) of every workgroup of problem 0 at its start, after the control step (designated wave), after the record arrived, at
// its end: one plain store each (atomics on shared words serialise and stretch a 12-us launch to 85 us).  Compiled out of the
// product build.
#ifdef SRRG2_PASS_TIMELINE
__device__ unsigned long long g_pass_ts[16 * 512 * 8];  // [epoch][workgroup (tile)][stamp]: plain stores, reduced on the host
#define PASS_TS_ANY(epoch, k) /* (by lane 0 of any wave: the last writer wins) */                      \
  do {                                                                                                 \
    if ((threadIdx.x & 63) == 0 && (epoch) >= 0 && (epoch) < 16 && blockIdx.y < 512 && blockIdx.x == 0) \
      g_pass_ts[(((epoch) * 512) + blockIdx.y) * 8 + (k)] = wall_clock64();                            \
  } while (0)
#define PASS_TS_W(epoch, k, dep) /* (inside wave_control: after `dep` has been computed) */             \
  do {                                                                                                 \
    if (threadIdx.x == 0 && (epoch) >= 0 && (epoch) < 16 && blockIdx.x == 0) {                         \
      unsigned long long t_ = wall_clock64();                                                          \
      if ((dep) != (dep)) t_ = 0; /* (a data dependence: the stamp cannot be scheduled ahead) */        \
      g_pass_ts[(((epoch) * 512) + 0) * 8 + (k)] = t_;                                                  \
    }                                                                                                  \
  } while (0)
#define PASS_TS(epoch, k)                                                                              \
  do {                                                                                                 \
    if (threadIdx.x == 0 && (epoch) >= 0 && (epoch) < 16 && blockIdx.y < 512 && blockIdx.x == 0)       \
      g_pass_ts[(((epoch) * 512) + blockIdx.y) * 8 + (k)] = wall_clock64();                            \
  } while (0)
#else
#define PASS_TS(epoch, k) do { } while (0)
#define PASS_TS_ANY(epoch, k) do { } while (0)
#define PASS_TS_W(epoch, k, dep) do { } while (0)
#endif
namespace {

__device__ float robust_weight(int kind, float thr, float chi, bool& kernelized);  // (defined with the control kernels below)
__device__ __forceinline__ int slice_exponent(const CtlParams& C, const SliceCtl& s, int prob, int nm, const float* X);
template <bool ONLY_INLINE>
__device__ __forceinline__ void init_problem_thread0(const CtlParams& C, int prob, const ProblemDev* probs_host, ProblemDev* probs,
                                                     ProblemState* states, const float* guesses_host, int tsize,
                                                     const InitInline& inl, unsigned (*init_gran)[PUB_SLICE_GRANULES]);

__device__ __forceinline__ double rl_d(double v, int k) {  // lane k's value, k wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long rl_ll(long long v, int k) {
  const int lo = __builtin_amdgcn_readlane((int) (unsigned) v, k);
  const int hi = __builtin_amdgcn_readlane((int) (unsigned) ((unsigned long long) v >> 32), k);
  return (long long) (((unsigned long long) (unsigned) hi << 32) | (unsigned) lo);
}
__device__ __forceinline__ float rl_f(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v              = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double o = __shfl_xor(v, off);
    v              = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned long long pub_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void pub_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the granule of lane `lane` of the record of (problem, slice s), from the state
__device__ __forceinline__ unsigned pub_granule_of(const CtlParams& C, const ProblemState* st, int s, int lane) {
  unsigned v = 0u;
  if (lane < 12) v = __float_as_uint(st->Tf[s][lane]);
  else if (lane < 24) v = __float_as_uint(st->Tfprev[s][lane - 12]);
  else if (lane == PUB_G_KEXP) v = (unsigned) st->kexp[s];
  else if (lane == PUB_G_FLAGS)
    v = ((st->done || st->finished) ? PUB_FLAG_STOP : 0u) | (st->phase == 1 ? PUB_FLAG_PHASE1 : 0u) |
        ((st->nstats > 0 || st->phase == 1) ? PUB_FLAG_PRIOR : 0u);
  else if (lane == PUB_G_NSTATS) v = (unsigned) st->nstats;
  else if (lane == PUB_G_WCOUNT) v = (unsigned) st->w_count;
  else if (lane == PUB_G_NPASSES) v = (unsigned) st->npasses;
  else if (lane >= PUB_G_X && lane < PUB_G_X + 12) v = __float_as_uint(st->X[lane - PUB_G_X]);
  return v;
}
__device__ __forceinline__ void pub_write_epoch(unsigned* pub_epoch, int prob, int lane, unsigned epoch) {
  if (lane < PUB_EPOCH_REPLICAS)
    __hip_atomic_store(pub_epoch + ((size_t) prob * PUB_EPOCH_REPLICAS + lane) * PUB_EPOCH_STRIDE, epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
// all records of a problem from its state, by the lanes 0 .. 63 of one wave (the control / init / post kernels)
__device__ __forceinline__ void pub_publish_state(const CtlParams& C, const ProblemState* st, int prob, int lane, unsigned epoch) {
  for (int s = 0; s < C.nslices; ++s) {
    if (C.slices[s].kind == SRRG2_SLICE_PRIOR) continue;
    pub_store(C.pub + ((size_t) prob * SRRG2_MAX_SLICES + s) * PUB_SLICE_GRANULES + lane,
              ((unsigned long long) epoch << 32) | pub_granule_of(C, st, s, lane));
  }
}

// The control step on one wave, for an aligner of `ns` <= MAXS cue slices (+ prior slices: PRIORS) (Sv: their SliceDev records in
// slice order -- the calling pass kernel's own argument, or the pack of the projective kernels).  Everything it reads
// arrives in ONE round trip: the records of the previous epoch (`g[z]`: this lane's granule of slice z's record, already
// loaded by the caller) carry the state the step needs, the slot sets are addressed from the kernel arguments.  Nothing waits
// for its plain stores (state, statistics, zeroed slot sets: read after the next kernel boundary); the new records and the
// epoch words are self-contained 8-byte stores.  D = 3 | 6.
// PUBLISH = false: the step of a POLLING wave (pass_view_fused / records_fused: the designated wave has not published within
// the poll limit).  Everything the step reads is read-only for the whole launch -- the records of the previous epoch, the
// slot sets of the previous round (FusedCtl::prev_partials: zeroed two rounds later, not here), the termination windows but
// for the entry of this iteration, which comes from the registers -- so any wave computes the same records from them; this
// variant stores NOTHING (workgroup (problem, 0) stays the only writer of state, statistics, slot sets and records) and
// returns this lane's granule of every slice's new record in `out`.
// A prior slice's factor (SE2PriorErrorFactor / SE3PriorErrorFactorAD: prior_linearize below, statement for statement) on the
// lanes of the control wave: every lane computes the error vector e and E = Z^-1 X itself (the same scalar code: the same bits
// in every lane), lane r D + c then forms ITS entry w sum_k (J(k, r) info(k)) J(k, c) of the factor's H, lane a its entry of b --
// the loops of prior_linearize over k, in the same order, with the entries of J picked per lane instead of stored (36 doubles).
// Returns the factor's status; pH / pb = 0 for a suppressed factor (prior_linearize returns zeros there).
// (prior_Z / prior_info / robust_thr: the slice's SliceCtl fields, fetched by the caller)
template <int D>
__device__ __forceinline__ int wave_prior(const float (&prior_Z)[12], const float (&prior_info)[6], float robust_thr, int rk,
                                          const float (&X)[12], int lane, double& pH, double& pb, double& chi_out) {
  const int hr = lane / D, hc = lane - hr * D;
  double e[D];
  float Zinv[12], E[12];
  double qw = 0.0;
  if constexpr (D == 6) {
    dm::se3_inverse(prior_Z, Zinv);
    dm::se3_compose(Zinv, X, E);
    dm::se3_t2v_quat(E, e);
    const double n2 = (e[3] * e[3] + e[4] * e[4]) + e[5] * e[5];
    qw              = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
  } else {
    dm::se2_inverse(prior_Z, Zinv);
    dm::se2_compose(Zinv, X, E);
    dm::se2_t2v(E, e);
  }
  // J(k, col), k uniform, col this lane's
  auto Jat = [&](int k, int col) -> double {
    if constexpr (D == 6) {
      if (k < 3) return col == 0 ? (double) E[k * 4 + 0] : (col == 1 ? (double) E[k * 4 + 1] : (col == 2 ? (double) E[k * 4 + 2] : 0.0));
      const double m0 = k == 3 ? qw : (k == 4 ? e[5] : -e[4]);
      const double m1 = k == 3 ? -e[5] : (k == 4 ? qw : e[3]);
      const double m2 = k == 3 ? e[4] : (k == 4 ? -e[3] : qw);
      return col == 3 ? m0 : (col == 4 ? m1 : (col == 5 ? m2 : 0.0));
    } else {
      if (k < 2) return col == 0 ? (double) E[k * 3 + 0] : (col == 1 ? (double) E[k * 3 + 1] : 0.0);
      return col == 2 ? 1.0 : 0.0;
    }
  };
  double chi = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) chi = chi + (e[i] * (double) prior_info[i]) * e[i];
  bool kernelized;
  const float w = robust_weight(rk, robust_thr, (float) chi, kernelized);
  chi_out       = chi;
  const int status = !isfinite(chi) ? (int) SRRG2_FACTOR_SUPPRESSED
                                    : (kernelized ? (int) SRRG2_FACTOR_KERNELIZED : (int) SRRG2_FACTOR_INLIER);
  pH = 0.0;
  pb = 0.0;
  if (status == SRRG2_FACTOR_SUPPRESSED) return status;
  double th = 0.0, tb = 0.0;
#pragma unroll
  for (int k = 0; k < D; ++k) {
    th = th + (Jat(k, hr) * (double) prior_info[k]) * Jat(k, hc);
    tb = tb + (Jat(k, lane) * (double) prior_info[k]) * e[k];
  }
  pH = (double) w * th;
  pb = (double) w * tb;
  return status;
}

// What the LAST control step of a compute() leaves in the registers of its wave for the finalize step behind it
// (k_icp_final_wave): filled only when the step ran in full (`applied`; a run that had already stopped, or stops here for want of
// correspondences, is finalized from ProblemState as before).
struct FinalRegs {
  bool applied;
  float Xl;        // lanes [0, 12): X after this step
  double Hl;       // lane r D + c: H(r, c) of this step
  int nstats;      // IterationStats appended so far, this step's included
  int num_in;      // inliers of this step (what the post step checks against min_num_inliers)
  int sw;          // lanes [0, 8): the words of this step's IterationStats record
  bool stats_written;
  int nc[4], ni[4];  // correspondences / inliers of the launch's cue slices
};

template <int D, int MAXS, bool PUBLISH = true, bool PRIORS = false, bool FINAL = false>
__device__ __forceinline__ void wave_control(const SliceDev* __restrict__ Sv, int ns, ProblemState* __restrict__ states, int prob,
                                             const unsigned long long (&g)[MAXS], unsigned (*out)[MAXS] = nullptr,
                                             FinalRegs* fin = nullptr) {
  if constexpr (FINAL) fin->applied = false;
  const FusedCtl& F  = Sv[0].fc;
  const int lane     = threadIdx.x & 63;
  ProblemState* st   = &states[prob];
  const unsigned epoch = (unsigned) F.epoch;
  constexpr int TS   = D == 3 ? 9 : 12;  // words of X
  // (prior slices: the head of their SliceCtl records -- kind .. prior_info, 25 words -- one word per lane, requested BEFORE the
  // slot sets: loads return in order, and the priors are linearised while the slot sets are still on their way)
  int pw[SRRG2_MAX_SLICES];
#pragma unroll
  for (int q = 0; q < SRRG2_MAX_SLICES; ++q) {
    pw[q] = 0;
    if constexpr (PRIORS)
      if (((unsigned) F.prior_mask >> q) & 1u) pw[q] = reinterpret_cast<const int*>(&F.ctl->slices[q])[lane & 31];
  }
  // ---- the slot sets of the passes of the previous epoch (buffer (epoch - 1) & 1) of every slice: requested first
  long long v[MAXS];
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    v[z] = 0;
    if (z < ns) {
      const long long* p = Sv[z].fc.prev_partials + (size_t) prob * PARTIAL_SLOTS * ACC_N;
#pragma unroll
      for (int q = 0; q < PARTIAL_SLOTS * ACC_N / 64; ++q) v[z] += p[q * 64 + lane];
    }
  }
  const int w          = (int) (unsigned) g[0];  // the value of this lane's granule (slice 0: the shared part of the state)
  float Xl             = __int_as_float(__shfl(w, (PUB_G_X + lane) & 63));  // lanes [0, 12): X
  const unsigned fl0   = (unsigned) __builtin_amdgcn_readlane(w, PUB_G_FLAGS);
  const int nstats0    = __builtin_amdgcn_readlane(w, PUB_G_NSTATS);
  const int wc         = __builtin_amdgcn_readlane(w, PUB_G_WCOUNT);
  const int npasses0   = __builtin_amdgcn_readlane(w, PUB_G_NPASSES);
  float told[MAXS];  // lanes [0, 12): Tf of the passes just run, per slice
  int kexp[MAXS];
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    told[z] = __int_as_float((int) (unsigned) g[z]);
    kexp[z] = __builtin_amdgcn_readlane((int) (unsigned) g[z], PUB_G_KEXP);
  }
  if (fl0 & PUB_FLAG_STOP) {  // (the passes return at their first instruction; the records only move to the new epoch)
#pragma unroll
    for (int z = 0; z < MAXS; ++z)
      if (z < ns) {
        if constexpr (PUBLISH)
          pub_store(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane,
                    ((unsigned long long) epoch << 32) | (unsigned) g[z]);
        else
          (*out)[z] = (unsigned) g[z];
      }
    if constexpr (PUBLISH) pub_write_epoch(F.pub_epoch, prob, lane, epoch);
    return;
  }
  // (termination criterion: the windows, one entry per lane, in flight with the slot sets)
  double wo = 0.0, wi = 0.0, wx = 0.0;
  int W = 1;
  if (F.has_term) {
    W = F.ctl->term.window_size;
    if (lane < W) {
      wo = st->w_out[lane];
      wi = st->w_inl[lane];
      wx = st->w_chi[lane];
    }
  }
  // Prior factors (at most two: run_compute), linearised AHEAD of the sums: they depend on X only, which the record carries -- their
  // float64 chain (a 4 x 4 inverse and product, the quaternion of the error, a division and a square root or two) runs while the
  // slot sets are still on their way; the slice loop below adds the results in slice order.
  int ps0 = -1, ps1 = -1, pt0 = 0, pt1 = 0;
  double pH0 = 0.0, pb0 = 0.0, pc0 = 0.0, pH1 = 0.0, pb1 = 0.0, pc1 = 0.0;
  if constexpr (PRIORS) {
    const unsigned pm0 = (unsigned) F.prior_mask, pm1 = pm0 & (pm0 - 1u);
    ps0 = pm0 ? __builtin_ctz(pm0) : -1;
    ps1 = pm1 ? __builtin_ctz(pm1) : -1;
    float Xa[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Xa[i] = i < TS ? rl_f(Xl, i) : 0.f;
#pragma unroll 1
    for (int j = 0; j < 2; ++j) {
      const int sl = j == 0 ? ps0 : ps1;
      if (sl < 0) break;
      int pword = pw[0];
#pragma unroll
      for (int q = 1; q < SRRG2_MAX_SLICES; ++q) pword = sl == q ? pw[q] : pword;
      constexpr int W_RK = (int) (offsetof(SliceCtl, robust_kind) / 4), W_THR = (int) (offsetof(SliceCtl, robust_thr) / 4),
                    W_Z = (int) (offsetof(SliceCtl, prior_Z) / 4), W_INFO = (int) (offsetof(SliceCtl, prior_info) / 4);
      static_assert(W_INFO + 6 <= 32, "the prefetched head of SliceCtl covers the prior's fields");
      const int robust_kind  = __builtin_amdgcn_readlane(pword, W_RK);
      const float robust_thr = __int_as_float(__builtin_amdgcn_readlane(pword, W_THR));
      float pZ[12], pinfo[6];
#pragma unroll
      for (int i = 0; i < 12; ++i) pZ[i] = __int_as_float(__builtin_amdgcn_readlane(pword, W_Z + i));
#pragma unroll
      for (int i = 0; i < 6; ++i) pinfo[i] = __int_as_float(__builtin_amdgcn_readlane(pword, W_INFO + i));
      const int rk = ((fl0 & PUB_FLAG_PHASE1) && robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : robust_kind;
      double pH, pb, pchi;
      const int pstat = wave_prior<D>(pZ, pinfo, robust_thr, rk, Xa, lane, pH, pb, pchi);
      if (j == 0) {
        pH0 = pH; pb0 = pb; pc0 = pchi; pt0 = pstat;
      } else {
        pH1 = pH; pb1 = pb; pc1 = pchi; pt1 = pstat;
      }
    }
  }
  const int hr = lane / D, hc = lane - hr * D;
  const int hsrc = hidx(hr < hc ? hr : hc, hr < hc ? hc : hr) & 31;
  double Hl = 0.0, bl = 0.0;  // H(r, c) in lane r D + c; b(a) in lane a
  int num_in = 0, num_out = 0, num_sup = 0, num_corr = 0;
  double chi_in = 0.0, chi_out = 0.0;
  bool good = false;
  // The slices in slice order (the order of control_body's sums): the cue slices of the launch are consecutive (run_compute checks),
  // prior slices may stand before and behind them (FusedCtl::prior_mask; none: the loop runs its cue part once).
  const unsigned pmask = PRIORS ? (unsigned) F.prior_mask : 0u;  // (PRIORS: the kernel instantiations of aligners with prior slices)
  const int first_cue  = Sv[0].slice_idx;
  const int s_end      = pmask ? F.nslices : first_cue + 1;
#pragma unroll 1
  for (int sl = pmask ? 0 : first_cue; sl < s_end; ++sl) {
  if ((pmask >> sl) & 1u) {
    const bool pfirst = sl == ps0;
    const double pH = pfirst ? pH0 : pH1, pb = pfirst ? pb0 : pb1, pchi = pfirst ? pc0 : pc1;
    const int pstat = pfirst ? pt0 : pt1;
    good = true;  // aligner_slice_processor_prior.h:66-68
    Hl   = Hl + pH;
    bl   = bl + pb;
    if (pstat == SRRG2_FACTOR_INLIER) {
      num_in += 1;
      chi_in = chi_in + pchi;
    } else if (pstat == SRRG2_FACTOR_KERNELIZED) {
      num_out += 1;
      chi_out = chi_out + pchi;
    } else {
      num_sup += 1;
    }
    num_corr += 1;  // (num_correspondences(): a prior slice counts one)
    if constexpr (PUBLISH)
      if (lane == 0) st->ninl[sl] = pstat == SRRG2_FACTOR_INLIER ? 1 : 0;
    continue;
  }
  if (sl != first_cue) continue;  // (the cue slices: all of them at the first one's place)
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    if (z >= ns) break;  // (uniform)
    v[z] += __shfl_xor(v[z], 32);  // the total of entry (lane & 31)
    // the slot sets are accumulated with atomics by the passes: the buffer that the round AFTER this launch's adds into is
    // zeroed here (it was read by the control step before this one); the buffer just read stays as it is until the launch
    // has retired -- a polling wave may still have to read it (PUBLISH = false)
    if constexpr (PUBLISH) {
      long long* p = Sv[z].fc.zero_partials + (size_t) prob * PARTIAL_SLOTS * ACC_N;
#pragma unroll
      for (int q = 0; q < PARTIAL_SLOTS * ACC_N / 64; ++q) p[q * 64 + lane] = 0;
    }
    const double scaled = (double) v[z] * dm::pow2(-kexp[z]);
    const int nc = (int) rl_ll(v[z], ACC_N_CORR), n_in = (int) rl_ll(v[z], ACC_N_IN), n_out = (int) rl_ll(v[z], ACC_N_OUT);
    good |= nc > Sv[z].fc.min_num_correspondences;  // aligner_slice_processor_impl.cpp:77-79
    if constexpr (FINAL)
      if (z < 4) {
        fin->nc[z] = nc;
        fin->ni[z] = n_in;
      }
    Hl = Hl + __shfl(scaled, hsrc);
    bl = bl + __shfl(scaled, (ACC_B + lane) & 31);
    num_in += n_in;
    num_out += n_out;
    num_sup += nc - n_in - n_out;
    num_corr += nc >= 0 ? nc : 0;
    chi_in  = chi_in + rl_d(scaled, ACC_CHI_IN);
    chi_out = chi_out + rl_d(scaled, ACC_CHI_OUT);
    if constexpr (PUBLISH) {
      const int s = Sv[z].slice_idx;
      if (lane == 0) {
        st->ncorr[s] = nc;
        st->ninl[s]  = n_in;
      }
      if (lane < 12) st->Tlast[s][lane] = told[z];  // the transforms the passes of this iteration ran with
    }
  }
  }  // (slices)
  if constexpr (PUBLISH)
    if (lane == 0) st->npasses = npasses0 + 1;
  if (!good) {  // multi_aligner_impl.cpp:107-111
    if constexpr (PUBLISH)
      if (lane == 0) {
        st->status = SRRG2_NOT_ENOUGH_CORRESPONDENCES;
        st->done   = 1;
      }
#pragma unroll
    for (int z = 0; z < MAXS; ++z) {
      if (z >= ns) break;
      unsigned nv = (unsigned) g[z];
      if (lane == PUB_G_FLAGS) nv = fl0 | PUB_FLAG_STOP;
      if (lane == PUB_G_NPASSES) nv = (unsigned) (npasses0 + 1);
      if constexpr (PUBLISH)
        pub_store(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane,
                  ((unsigned long long) epoch << 32) | nv);
      else
        (*out)[z] = nv;
    }
    if constexpr (PUBLISH) pub_write_epoch(F.pub_epoch, prob, lane, epoch);
    return;
  }
  PASS_TS_W(F.epoch, 5, Hl);
  // ---- dm::solve<D>: L D L^T, statement for statement; L(i, j) in lane i D + j (i > j), d(j) in lane j D + j of Dv,
  //      1 / d(j) in the same lane of Iv
  //      RIGHT-LOOKING: every lane (i, j) carries its own running value H(i, j) - sum_{k done} (L(i, k) L(j, k)) d(k) and takes
  //      the term of column k as soon as that column is final -- the same terms in the same (ascending k) order for every
  //      entry, hence the same bits as dm::solve's loops, but one dependent step per pivot instead of 2 j: the critical
  //      path of the step is this wave's chain of float64 operations (profiles/r6l).
  double Lv = 0.0, Dv = 0.0, Iv = 0.0;
  double Tv = Hl;
  bool bad  = false;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const double sj = rl_d(Tv, j * D + j);
    if (!(sj > 0.0)) {
      bad = true;
      break;
    }
    const double inv = 1.0 / sj;
    if (lane == j * D + j) {
      Dv = sj;
      Iv = inv;
    }
    if (hc == j && hr > j && lane < D * D) Lv = Tv * inv;
    if (j + 1 < D) {  // entry (i, j'), j' > j: - (L(i, j) L(j', j)) d(j)   (the lanes left of / above are not read again)
      const double lij = __shfl(Lv, (hr * D + j) & 63);
      const double ljj = __shfl(Lv, (hc * D + j) & 63);
      Tv               = Tv - (lij * ljj) * sj;
    }
  }
  double yv = 0.0, dxv = 0.0;  // y(i), dx(i) in lane i
  if (!bad) {
    // forward substitution by columns: lane i carries -b(i) - sum_{k done} L(i, k) y(k), final once k reaches i
    double tv = -bl;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const double yk = rl_d(tv, k);
      if (lane == k) yv = tv;
      if (k + 1 < D) tv = tv - __shfl(Lv, (lane * D + k) & 63) * yk;
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
      double t = rl_d(yv, i) * rl_d(Iv, i * D + i);
#pragma unroll
      for (int k = i + 1; k < D; ++k) t = t - rl_d(Lv, k * D + i) * rl_d(dxv, k);
      if (lane == i) dxv = t;
    }
    const bool nf = lane < D && (!(dxv == dxv) || dxv > 1e300 || dxv < -1e300);
    bad           = __any(nf);
  }
  PASS_TS_W(F.epoch, 6, dxv);
  // ---- what get_information / the batch records read: H, b, dx of this Gauss-Newton iteration
  if constexpr (PUBLISH) {
    if (lane < D * D) st->last_H[lane] = Hl;
    if (lane < D) {
      st->last_b[lane]  = bl;
      st->last_dx[lane] = bad ? 0.0 : dxv;
    }
    if (lane < 12) st->Xprev[lane] = Xl;
  }
  // ---- X <- X * v2t(dx) (dm::box_plus), the element of lane l < TS
  if (!bad) {
    if constexpr (D == 3) {
      double sn, cs;
      dm::sincos(rl_d(dxv, 2), sn, cs);
      const int i = lane / 3, j = lane - i * 3;
      const double a0 = (double) __shfl(Xl, (i * 3 + 0) & 63), a1 = (double) __shfl(Xl, (i * 3 + 1) & 63),
                   a2 = (double) __shfl(Xl, (i * 3 + 2) & 63);
      const double o0 = a0 * cs + a1 * sn;
      const double o1 = a0 * (-sn) + a1 * cs;
      const double o2 = (a0 * rl_d(dxv, 0) + a1 * rl_d(dxv, 1)) + a2;
      float out       = (float) (j == 0 ? o0 : (j == 1 ? o1 : o2));
      if (lane >= 6) out = lane == 8 ? 1.f : 0.f;
      if (lane < 9) Xl = out;
    } else {
      // dm::se3_v2t on named scalars (through its arrays, filled on three control-flow paths, the compiler keeps part of R on
      // the stack: a kernel that owns scratch memory pays for it in every wave)
      const double v0 = rl_d(dxv, 0), v1 = rl_d(dxv, 1), v2 = rl_d(dxv, 2), v3 = rl_d(dxv, 3), v4 = rl_d(dxv, 4), v5 = rl_d(dxv, 5);
      double R0, R1, R2, R3, R4, R5, R6, R7, R8;
      if (Sv[0].variable_kind == 2) {
        const double n2  = (v3 * v3 + v4 * v4) + v5 * v5;
        const bool small = n2 < 1.0;
        const double qw  = sqrt(small ? 1.0 - n2 : 1.0);
        // dm::quat_to_R(qw, v3, v4, v5)
        const double xx = v3 * v3, yy = v4 * v4, zz = v5 * v5;
        const double xy = v3 * v4, xz = v3 * v5, yz = v4 * v5;
        const double wx_ = qw * v3, wy_ = qw * v4, wz_ = qw * v5;
        R0 = small ? 1.0 - 2.0 * (yy + zz) : 1.0; R1 = small ? 2.0 * (xy - wz_) : 0.0;      R2 = small ? 2.0 * (xz + wy_) : 0.0;
        R3 = small ? 2.0 * (xy + wz_) : 0.0;      R4 = small ? 1.0 - 2.0 * (xx + zz) : 1.0; R5 = small ? 2.0 * (yz - wx_) : 0.0;
        R6 = small ? 2.0 * (xz - wy_) : 0.0;      R7 = small ? 2.0 * (yz + wx_) : 0.0;      R8 = small ? 1.0 - 2.0 * (xx + yy) : 1.0;
      } else {
        double sa, ca, sb, cb, sc, cc;
        dm::sincos(v3, sa, ca);
        dm::sincos(v4, sb, cb);
        dm::sincos(v5, sc, cc);
        R0 = cb * cc;                   R1 = -(cb * sc);                R2 = sb;
        R3 = ca * sc + (sa * sb) * cc;  R4 = ca * cc - (sa * sb) * sc;  R5 = -(sa * cb);
        R6 = sa * sc - (ca * sb) * cc;  R7 = sa * cc + (ca * sb) * sc;  R8 = ca * cb;
      }
      const int i = lane >> 2, j = lane & 3;
      const double a0 = (double) __shfl(Xl, (i * 4 + 0) & 63), a1 = (double) __shfl(Xl, (i * 4 + 1) & 63),
                   a2 = (double) __shfl(Xl, (i * 4 + 2) & 63), a3 = (double) __shfl(Xl, (i * 4 + 3) & 63);
      double r0 = v0, r1 = v1, r2 = v2;  // (t = the translation part of dx)
      if (j == 0) { r0 = R0; r1 = R3; r2 = R6; }
      if (j == 1) { r0 = R1; r1 = R4; r2 = R7; }
      if (j == 2) { r0 = R2; r1 = R5; r2 = R8; }
      double o = (a0 * r0 + a1 * r1) + a2 * r2;
      if (j == 3) o = o + a3;
      if (lane < 12) Xl = (float) o;
    }
    if constexpr (PUBLISH)
      if (lane < TS) st->X[lane] = Xl;
  }
  PASS_TS_W(F.epoch, 7, (double) Xl);
  // ---- IterationStats of this iteration (multi_aligner_impl.cpp:113-115): one 4-byte word of the record per lane
  const float chi_in_f = (float) chi_in, chi_out_f = (float) chi_out;
  static_assert(sizeof(srrg2_iteration_stats) == 32, "eight words");
  const int sw = lane == 0 ? nstats0 : lane == 1 ? num_in : lane == 2 ? num_out : lane == 3 ? num_sup : lane == 4 ? num_corr
               : lane == 5 ? (bad ? 1 : 0) : lane == 6 ? __float_as_int(chi_in_f) : __float_as_int(chi_out_f);
  if (PUBLISH && nstats0 < F.max_stats && lane < 8) reinterpret_cast<int*>(F.stats + (size_t) prob * F.max_stats + nstats0)[lane] = sw;
  if constexpr (FINAL) {
    fin->applied = true;
    fin->Xl = Xl;
    fin->Hl = Hl;
    fin->nstats = nstats0 + 1;
    fin->num_in = num_in;
    fin->sw = sw;
    fin->stats_written = nstats0 < F.max_stats;
  }
  if constexpr (PUBLISH)
    if (lane == 0) st->nstats = nstats0 + 1;
  // ---- AlignerTerminationCriteriaStandard_::hasToStop (aligner_termination_criteria_impl.cpp:24-65, has_to_stop below)
  bool stop = false;
  int wc1   = wc;
  if (F.has_term && num_in != 0) {
    const srrg2_termination_params& tp = F.ctl->term;
    const float chi = chi_in_f / (float) num_in;
    const int slot  = wc % W;
    const int n     = wc + 1 < W ? wc + 1 : W;
    wc1             = wc + 1;
    if (n >= W) {  // (W <= TERM_WINDOW_MAX = 64: one entry per lane; the entry of this iteration comes from the registers)
      const bool in = lane < n;
      if (lane == slot) {
        wo = (double) num_out;
        wi = (double) num_in;
        wx = (double) chi;
      }
      const double first_o = rl_d(wo, 0), first_i = rl_d(wi, 0), first_x = rl_d(wx, 0);
      const double omax = wave_max_d(in ? wo : first_o), omin = wave_min_d(in ? wo : first_o);
      const double imax = wave_max_d(in ? wi : first_i), imin = wave_min_d(in ? wi : first_i);
      const double xmax = wave_max_d(in ? wx : first_x), xmin = wave_min_d(in ? wx : first_x);
      stop = true;
      if (omax - omin > (double) tp.num_correspondences_range) stop = false;  // :46
      if (imax - imin > (double) tp.num_inliers_range) stop = false;
      const float chi_range = (float) (xmax - xmin);
      if (chi_range > (float) tp.num_outliers_range) stop = false;  // :53
      if (chi_range / (float) xmax > tp.chi_epsilon) stop = false;
    }
    if (PUBLISH && lane == 0) {
      st->w_corr[slot] = num_corr;
      st->w_inl[slot]  = num_in;
      st->w_out[slot]  = num_out;
      st->w_chi[slot]  = (double) chi;
      st->w_count      = wc + 1;
    }
  }
  if (PUBLISH && stop && lane == 0) st->done = 1;  // :124-126
  // ---- finder transforms robot_in_sensor * X (finder_transform_of) of the slices, the previous ones kept; the records of
  //      the new epoch, then the epoch words
  const unsigned x_up = __float_as_uint(__shfl(Xl, (lane - PUB_G_X) & 63));
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    if (z >= ns) break;  // (uniform)
    const int s = Sv[z].slice_idx;
    float tnew  = told[z];
    if (!bad) {
      const float* A = Sv[z].Sinv;
      const int i = (lane >> 2) % 3, j = lane & 3;  // slot (i, j) of the 3 x 4 layout
      const float A0 = i == 0 ? A[0] : (i == 1 ? A[D == 6 ? 4 : 3] : A[D == 6 ? 8 : 6]);
      const float A1 = i == 0 ? A[1] : (i == 1 ? A[D == 6 ? 5 : 4] : A[D == 6 ? 9 : 7]);
      const float A2 = i == 0 ? A[2] : (i == 1 ? A[D == 6 ? 6 : 5] : A[D == 6 ? 10 : 8]);
      if constexpr (D == 6) {
        const float A3  = i == 0 ? A[3] : (i == 1 ? A[7] : A[11]);
        const double b0 = (double) __shfl(Xl, (0 * 4 + j) & 63), b1 = (double) __shfl(Xl, (1 * 4 + j) & 63),
                     b2 = (double) __shfl(Xl, (2 * 4 + j) & 63);
        double o = ((double) A0 * b0 + (double) A1 * b1) + (double) A2 * b2;
        if (j == 3) o = o + (double) A3;
        tnew = (float) o;
      } else {
        // se2_compose into t9, spread into the 3 x 4 slots: T = [t0 t1 0 t2; t3 t4 0 t5; 0 0 1 0]
        const int jj = j == 3 ? 2 : j;  // column of the 3 x 3 product the slot holds (j = 2: none)
        const double b0 = (double) __shfl(Xl, (0 * 3 + jj) & 63), b1 = (double) __shfl(Xl, (1 * 3 + jj) & 63);
        double o = (double) A0 * b0 + (double) A1 * b1;  // (rows i < 2; row 2 is overwritten below)
        if (j == 3) o = o + (double) A2;
        float f = (float) o;
        if (j == 2) f = 0.f;
        if (i == 2) f = j == 2 ? 1.f : 0.f;
        tnew = f;
      }
      if (PUBLISH && lane < 12) st->Tf[s][lane] = tnew;
    }
    if (PUBLISH && lane < 12) st->Tfprev[s][lane] = told[z];
    const unsigned told_up = __float_as_uint(__shfl(told[z], (lane - 12) & 63));  // (every lane takes part in the shuffle)
    unsigned nv = 0u;
    if (lane < 12) nv = __float_as_uint(tnew);
    if (lane >= 12 && lane < 24) nv = told_up;
    if (lane == PUB_G_KEXP) nv = (unsigned) kexp[z];
    if (lane == PUB_G_FLAGS) nv = (stop ? PUB_FLAG_STOP : 0u) | (fl0 & PUB_FLAG_PHASE1) | PUB_FLAG_PRIOR;
    if (lane == PUB_G_NSTATS) nv = (unsigned) (nstats0 + 1);
    if (lane == PUB_G_WCOUNT) nv = (unsigned) wc1;
    if (lane == PUB_G_NPASSES) nv = (unsigned) (npasses0 + 1);
    if (lane >= PUB_G_X && lane < PUB_G_X + 12) nv = x_up;
    if constexpr (PUBLISH)
      pub_store(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + s) * PUB_SLICE_GRANULES + lane, ((unsigned long long) epoch << 32) | nv);
    else
      (*out)[z] = nv;
  }
  if constexpr (PUBLISH) pub_write_epoch(F.pub_epoch, prob, lane, epoch);
}

// What a pass kernel needs of the state.  Legacy (FUSED = false: its own kernel instantiations, the code of round 4): read from
// ProblemState.  Fused control steps: from the record of (problem, slice).  ONE wave per workgroup reads it (agent-scope
// loads: the record changes during the launch and other XCDs' L2s are not coherent) and hands the values to the others
// through LDS; a stale record => the control step is applied here (designated wave) or waited for.  Called AFTER the
// workgroup has requested its points: those loads do not depend on the state and are in flight while the record arrives.
struct PassView {
  float T[12], Tprev[12];
  int kexp;
  bool stop, phase1, prior;
};

__device__ __forceinline__ void pass_view_legacy(const SliceDev& S, const ProblemState* __restrict__ st, PassView& v) {
#pragma unroll
  for (int i = 0; i < 12; ++i) v.T[i] = st->Tf[S.slice_idx][i];
#pragma unroll
  for (int i = 0; i < 12; ++i) v.Tprev[i] = st->Tfprev[S.slice_idx][i];
  v.kexp   = st->kexp[S.slice_idx];
  v.stop   = st->done || st->finished;
  v.phase1 = st->phase == 1;
  v.prior  = st->nstats > 0 || st->phase == 1;
}

// Top of a fused pass kernel: wave 0 of workgroup (0, problem) applies the control step of the previous iteration if the
// record still stands at the previous epoch (this wave is the only writer of the record during the launch).  Before the
// workgroup requests its points: the step's registers are free again when the pass needs its own.
// How long a polling wave waits for the designated one before it applies the step itself (polls of ~1 us each: an
// agent-scope load and a short sleep).  Nothing in HIP promises that workgroup (problem, 0) is dispatched before its siblings;
// it is in practice, and when it is not -- or it is slow: a debugger, a profiler that serialises, a shared GPU -- the pollers
// proceed on their own computation of the same record instead of spinning (round 5 trapped after 2^24 polls: process-fatal).
#ifndef SRRG2_FUSED_POLL_LIMIT
#define SRRG2_FUSED_POLL_LIMIT 64
#endif
// -DSRRG2_FUSED_STALL=<n>: the designated wave sleeps n x ~3.4 us before its control step, so that every polling wave runs into
// the limit and takes the fallback (tests/test_gpu_fused_control.py builds such a library and compares bits); the counter says
// how many waves did.
#ifdef SRRG2_FUSED_STALL
__device__ unsigned long long g_fused_fallbacks;
extern "C" int srrg2_amd_debug_fused_fallbacks(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_fallbacks), sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fused_fallbacks), &zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
#define FUSED_STALL()                                                              \
  do {                                                                             \
    for (int stall_i = 0; stall_i < (SRRG2_FUSED_STALL); ++stall_i) __builtin_amdgcn_s_sleep(127); \
  } while (0)
#define FUSED_FALLBACK_COUNT()                                                     \
  do {                                                                             \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_fused_fallbacks, 1ull);              \
  } while (0)
#else
#define FUSED_STALL() do { } while (0)
#define FUSED_FALLBACK_COUNT() do { } while (0)
#endif

template <int DIM, bool PRIORS = false>
__device__ __forceinline__ void fused_control_if_due(const SliceDev& S, ProblemState* __restrict__ states, int prob) {
  if (blockIdx.y != 0 || threadIdx.x >= 64) return;  // (fused launches: x = problem, y = tile)
  const unsigned long long g[1] = {
    pub_load(S.fc.pub + ((size_t) prob * SRRG2_MAX_SLICES + S.slice_idx) * PUB_SLICE_GRANULES + (threadIdx.x & 63))};
  if (!__all((unsigned) (g[0] >> 32) == (unsigned) S.fc.epoch)) {
    FUSED_STALL();
    wave_control<DIM == 3 ? 6 : 3, 1, true, PRIORS>(&S, 1, states, prob, g);
  }
  PASS_TS(S.fc.epoch, 1);
}

template <int DIM, bool PRIORS = false>
__device__ __forceinline__ void pass_view_fused(const SliceDev& S, ProblemState* __restrict__ states, int prob, PassView& v) {
  __shared__ unsigned rec_lds[PUB_SLICE_GRANULES];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const unsigned long long* rec = S.fc.pub + ((size_t) prob * SRRG2_MAX_SLICES + S.slice_idx) * PUB_SLICE_GRANULES + lane;
    // A workgroup of the first round of the launch starts before the control steps have published: agent-scope load
    // (no copy of the stale record left in this XCD's L2).  A later one starts when a first-round workgroup has retired,
    // i.e. after the control steps (they all run in the first K workgroups of the launch): an ordinary cached load finds the
    // new record -- and if it ever does not, the tags say so and the agent-scope path below takes over.
    const bool late      = (int) (blockIdx.y * gridDim.x + blockIdx.x) >= S.fc.first_round;
    unsigned long long g = late ? *rec : pub_load(rec);
    if (!__all((unsigned) (g >> 32) == (unsigned) S.fc.epoch)) {
      // (workgroup (0, problem) applied the control step at its very top -- fused_control_if_due -- before it came here)
      const unsigned* ep = S.fc.pub_epoch + ((size_t) prob * PUB_EPOCH_REPLICAS + (blockIdx.y & (PUB_EPOCH_REPLICAS - 1))) * PUB_EPOCH_STRIDE;
      // Poll the epoch word, then the record (the epoch words are written after it).  Past the limit: if the record still
      // stands WHOLE at the previous epoch, this wave computes the new one itself, into registers (wave_control<.., false>:
      // same inputs, all read-only during this launch, same bits; nothing is stored); a record in the middle of being
      // published is simply waited for -- its writer is running.
      int spins = 0;
      for (;;) {
        if ((int) __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= S.fc.epoch) {
          g = pub_load(rec);
          if (__all((unsigned) (g >> 32) == (unsigned) S.fc.epoch)) break;
        }
        __builtin_amdgcn_s_sleep(4);
        if (++spins > SRRG2_FUSED_POLL_LIMIT) {
          g = pub_load(rec);
          if (__all((unsigned) (g >> 32) == (unsigned) S.fc.epoch)) break;
          if (__all((unsigned) (g >> 32) == (unsigned) (S.fc.epoch - 1))) {
            const unsigned long long gv[1] = {g};
            unsigned nv[1];
            wave_control<DIM == 3 ? 6 : 3, 1, false, PRIORS>(&S, 1, states, prob, gv, &nv);
            g = (unsigned long long) nv[0];
            FUSED_FALLBACK_COUNT();
            break;
          }
          spins = 0;
        }
      }
    }
    rec_lds[lane] = (unsigned) g;
  }
  __syncthreads();
  PASS_TS(S.fc.epoch, 2);
#pragma unroll
  for (int i = 0; i < 12; ++i) v.T[i] = __int_as_float(__builtin_amdgcn_readfirstlane((int) rec_lds[i]));
#pragma unroll
  for (int i = 0; i < 12; ++i) v.Tprev[i] = __int_as_float(__builtin_amdgcn_readfirstlane((int) rec_lds[12 + i]));
  v.kexp = __builtin_amdgcn_readfirstlane((int) rec_lds[PUB_G_KEXP]);
  const unsigned fl = (unsigned) __builtin_amdgcn_readfirstlane((int) rec_lds[PUB_G_FLAGS]);
  v.stop   = (fl & PUB_FLAG_STOP) != 0;
  v.phase1 = (fl & PUB_FLAG_PHASE1) != 0;
  v.prior  = (fl & PUB_FLAG_PRIOR) != 0;
}

// compute()'s prologue INSIDE the first pass kernel of a single alignment (round 6, late; k_icp_init otherwise: a launch of ~9 us
// in front of every compute()).  The first pass needs three things of that prologue -- the finder transform of the initial guess,
// the fixed-point exponent of the slice, zeroed slot sets -- and none of the rest (state, records, device copy of the control
// parameters, problem table: read from the second pass on, behind a kernel boundary).  So every thread of the pass derives the
// first two from the kernel arguments (the same statements as init_problem_thread0: same bits; two scalar loads of the clouds'
// norms, a few dozen operations), the slot sets are left zeroed by the final step of the handle's previous compute()
// (k_icp_final_wave; run_compute takes this path only then), and wave 0 of workgroup (problem, 0) writes the rest on its way.
template <int DIM>
__device__ __forceinline__ void pass_view_init(const SliceDev& S, const CtlParams& C, const InitInline& inl, int prob, int nm,
                                               PassView& v, const InitBatch* bat = nullptr) {
  const SliceCtl& sc = C.slices[S.slice_idx];
  float X[12];
  if (bat) {  // (a part of a batch: the problem's row of the launch's own table)
    const int lp = prob - C.prob0;
#pragma unroll
    for (int i = 0; i < 12; ++i) X[i] = bat->guess[lp][i];
  } else {
#pragma unroll
    for (int i = 0; i < 12; ++i) X[i] = inl.guess[i];  // (run_compute: a prior slice's override of the guess already applied)
  }
  const int nm_of = sc.nm_global > 0 ? sc.nm_global : nm;
  v.kexp = slice_exponent(C, sc, prob, nm_of, X);
  float T[12];
  finder_transform_of(sc.Sinv, DIM, X, T);
#pragma unroll
  for (int i = 0; i < 12; ++i) v.T[i] = v.Tprev[i] = T[i];
  v.stop = v.phase1 = v.prior = false;
}
// ... the rest of the prologue (k_icp_init's body without the zeroing), by one wave of that pass -- the workgroup behind the last
// tile, which holds no point.  Lane-distributed stores of what init_problem_thread0 computes with one thread: the transform and
// the exponent are the ones pass_view_init has just derived (`v`; the aligners that take this path have ONE cue slice, `cue`), the
// guess and the table rows come from the kernel arguments.  (The one-thread body inlined here took the pass kernel from 74 to
// 130 registers -- six waves per SIMD to three, and the first pass of a 32-batch is bound by its instruction issue: 0.535 ->
// 0.58 ms; this form leaves pass_view_init's float64 exponent as the widest point, 91.)
__device__ __forceinline__ void fused_init_tail(const CtlParams& C, const InitInline& inl, int prob, ProblemDev* __restrict__ probs,
                                                ProblemState* __restrict__ states, const InitBatch* bat, const PassView& v, int cue) {
  const int lane   = threadIdx.x & 63;
  const int lp     = prob - C.prob0;
  ProblemState* st = &states[prob];
  if (C.ctl_dev && blockIdx.x == 0) {  // (once per launch: the first problem's tail)
    const int* src = reinterpret_cast<const int*>(&C);
    int* dst       = reinterpret_cast<int*>(C.ctl_dev);
    for (int k = lane; k < (int) (sizeof(CtlParams) / sizeof(int)); k += 64) dst[k] = src[k];
  }
  // the initial guess (zero-padded beyond the variable's words by the host), element (lane mod 32) -- lanes [0, 12) store it into
  // the state, lanes [PUB_G_X, PUB_G_X + 12) into the record
  static_assert(PUB_G_X == 32, "the record's X granules sit one half-wave above the state's");
  // (selects over statically indexed kernel arguments: a lane-indexed argument array becomes a stack copy, and a kernel that owns
  // scratch memory pays for it in every wave)
  float x = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float gi = bat ? bat->guess[lp][i] : inl.guess[i];
    x              = (lane & 31) == i ? gi : x;
  }
  // the finder transform, element (lane mod 12) in lanes [0, 24) (taken through readfirstlane: selected straight from the view's
  // array the compiler moved the view to the stack)
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const float ti = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v.T[i])));
    t              = (lane == i || lane == 12 + i) ? ti : t;
  }
  if (lane < 12) {
    st->X[lane]          = x;
    st->Xprev[lane]      = x;
    st->Tf[cue][lane]     = t;
    st->Tfprev[cue][lane] = t;
  }
  if (lane == 0) {
    st->status   = SRRG2_FAIL;
    st->done     = 0;
    st->finished = 0;
    st->nstats   = 0;
    st->phase    = 0;
    st->w_count  = 0;
    st->npasses  = 0;
  }
  if (lane < SRRG2_MAX_SLICES) st->qmode[lane] = 1;
#pragma unroll
  for (int s = 0; s < SRRG2_MAX_SLICES; ++s) {  // (one slice per lane)
    if (s >= C.nslices) break;  // (uniform)
    ProblemDev pd{0, 0};
    if (bat) {
      if (s == cue) pd = bat->pd[lp];
    } else {
      pd = inl.pd[s];
    }
    int* qc = C.slices[s].qcount;
    if (lane == s) {
      probs[(size_t) s * C.K + prob] = pd;
      st->ncorr[s] = 0;
      st->ninl[s]  = 0;
      st->kexp[s]  = s == cue ? v.kexp : 0;
      if (qc) qc[2 * prob] = qc[2 * prob + 1] = 0;
    }
  }
  if (!C.pub) return;
  unsigned gran = 0u;  // (flags, nstats, w_count, npasses: 0)
  if (lane < 24) gran = __float_as_uint(t);
  if (lane == PUB_G_KEXP) gran = (unsigned) v.kexp;
  if (lane >= PUB_G_X && lane < PUB_G_X + 12) gran = __float_as_uint(x);
  pub_store(C.pub + ((size_t) prob * SRRG2_MAX_SLICES + cue) * PUB_SLICE_GRANULES + lane, (unsigned long long) gran);
  pub_write_epoch(C.pub_epoch, prob, lane, 0u);
}

// ... for an aligner of SEVERAL cue slices (a pack of projective slices, k_proj_zbuf_fz_init): k_icp_init's one-thread body as it is
// (the z-buffer kernel it rides in is not short of registers), the records staged in LDS for the wave that publishes them
__device__ __forceinline__ void fused_init_tail_full(const CtlParams& C, const InitInline& inl, int prob, ProblemDev* __restrict__ probs,
                                                     ProblemState* __restrict__ states) {
  __shared__ unsigned init_gran_tail[SRRG2_MAX_SLICES][PUB_SLICE_GRANULES];
  const int lane = threadIdx.x & 63;
  if (C.ctl_dev && blockIdx.y == 0) {
    const int* src = reinterpret_cast<const int*>(&C);
    int* dst       = reinterpret_cast<int*>(C.ctl_dev);
    for (int k = lane; k < (int) (sizeof(CtlParams) / sizeof(int)); k += 64) dst[k] = src[k];
  }
  if (lane == 0)
    init_problem_thread0<true>(C, prob, nullptr, probs, states, nullptr, C.variable_kind == SRRG2_SE2_RIGHT ? 9 : 12, inl, init_gran_tail);
  wave_lds_sync();
  if (!C.pub) return;
  for (int s = 0; s < C.nslices; ++s)
    if (C.slices[s].kind != SRRG2_SLICE_PRIOR)
      pub_store(C.pub + ((size_t) prob * SRRG2_MAX_SLICES + s) * PUB_SLICE_GRANULES + lane, (unsigned long long) init_gran_tail[s][lane]);
  pub_write_epoch(C.pub_epoch, prob, lane, 0u);
}

// The search pass on the GRID (no cell neighbour lists yet: the first compute() on a new fixed cloud, a tracker's every frame)
// with the control step of the previous iteration in its prologue (round 6).  Without the deferred-search queue: the open points
// are finished inside the kernel (the queue's kernel and its counters belong to the control LAUNCH: run_compute keeps both for
// clouds large enough for the queue to pay).  x = problem, y = tile, like the other fused launches.
// (MODE: 0 = plain, 1 = the aligner has prior slices, 2 = the first pass of a single alignment's compute() with the prologue inside)
template <int DIM, bool PLANE, int MODE>
__device__ __forceinline__ void grid_fused_body(const SliceDev& S, const ProblemDev* __restrict__ probs,
                                                ProblemState* __restrict__ states, const CtlParams* Ci = nullptr,
                                                const InitInline* inl = nullptr) {
  const int prob = blockIdx.x + S.prob0;
  const int tile = (int) blockIdx.y;
  if constexpr (MODE != 2) fused_control_if_due<DIM, MODE == 1>(S, states, prob);
  ProblemDev pd;
  if constexpr (MODE == 2) {
    pd = inl->pd[S.slice_idx];
  } else {
    pd = probs[prob];
  }
  const bool tail_wg = MODE == 2 && tile == (int) gridDim.y - 1;  // (one workgroup more than tiles: cnl_pass_body)
  if (tile * 256 >= pd.nm && (MODE == 2 ? !tail_wg : tile != 0)) return;
  PassView pv;
  if constexpr (MODE == 2) {
    pass_view_init<DIM>(S, *Ci, *inl, prob, pd.nm, pv);
    if (tail_wg && threadIdx.x < 64)  // (probs: the SLICE's table)
      fused_init_tail(*Ci, *inl, prob, const_cast<ProblemDev*>(probs) - (size_t) S.slice_idx * Ci->K, states, nullptr, pv, S.slice_idx);
  } else {
    pass_view_fused<DIM, MODE == 1>(S, states, prob, pv);
  }
  if (pv.stop || tile * 256 >= pd.nm) return;
  StepView sv;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    sv.T[i]     = pv.T[i];
    sv.Tprev[i] = pv.Tprev[i];
  }
  sv.kexp   = pv.kexp;
  sv.nstats = 0;  // (the -DSRRG2_TIMELINE stamps only)
  sv.phase1 = pv.phase1;
  sv.prior  = pv.prior;
  sv.qmode  = false;
  icp_step_body<DIM, PLANE, 4>(S, pd, sv, prob, tile, (int) gridDim.y - (MODE == 2 ? 1 : 0), (int) gridDim.x, nullptr);
}
template <int DIM, bool PLANE, bool PRIORS = false>
__global__ __launch_bounds__(256) void k_icp_step_fused(SliceDev S, const ProblemDev* __restrict__ probs,
                                                        ProblemState* __restrict__ states) {
  grid_fused_body<DIM, PLANE, PRIORS ? 1 : 0>(S, probs, states);
}
template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_step_fused_init(SliceDev S, CtlParams C, InitInline inl, ProblemDev* __restrict__ probs,
                                                             ProblemState* __restrict__ states) {
  grid_fused_body<DIM, PLANE, 2>(S, probs, states, &C, &inl);
}

// Several slices (the projective kernels: a pack of up to four slices that share one association): the control step of all
// of them at the top of the iteration's first kernel, and the records of `ns` slices staged in LDS by wave 0.
template <int MAXS, bool PRIORS = false>
__device__ __forceinline__ void fused_control_if_due_multi(const SliceDev* __restrict__ Sv, int ns, ProblemState* __restrict__ states,
                                                           int prob) {
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  unsigned long long g[MAXS];
  bool stale = false;
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    g[z] = 0ull;
    if (z < ns) {
      g[z]  = pub_load(Sv[0].fc.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + (threadIdx.x & 63));
      stale = stale || (unsigned) (g[z] >> 32) != (unsigned) Sv[0].fc.epoch;
    }
  }
  if (__any(stale)) {
    FUSED_STALL();
    wave_control<6, MAXS, true, PRIORS>(Sv, ns, states, prob, g);  // (projective finders: SE(3))
  }
}
// (ns_all: the slices of the aligner -- the control step is one step for all of them, whatever this kernel reads.
// FIRST = the iteration's first kernel, the one that carries the control step: only there can a record be stale -- the
// step kernel behind it starts behind a kernel boundary that the designated wave's stores have crossed, reads current
// records and carries no fallback: wave_control for four slices inside it took it from 98 to 203 registers)
template <int MAXS, bool FIRST, bool PRIORS = false>
__device__ __forceinline__ void records_fused(const SliceDev* __restrict__ Sv, int ns, int prob,
                                              unsigned (&rec)[MAXS][PUB_SLICE_GRANULES], ProblemState* __restrict__ states,
                                              int ns_all) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    const FusedCtl& F = Sv[0].fc;
    unsigned long long g[MAXS];
    bool stale = false, fallback = false;
#pragma unroll
    for (int z = 0; z < MAXS; ++z) {
      g[z] = 0ull;
      if (z < ns) {
        g[z]  = pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane);
        stale = stale || (unsigned) (g[z] >> 32) != (unsigned) F.epoch;
      }
    }
    (void) fallback;
    if (__any(stale)) {
      const unsigned* ep = F.pub_epoch + ((size_t) prob * PUB_EPOCH_REPLICAS + (blockIdx.x & (PUB_EPOCH_REPLICAS - 1))) * PUB_EPOCH_STRIDE;
      // (as pass_view_fused: poll, and past the limit apply the step in registers if every record stands whole at the
      // previous epoch)
      int spins = 0;
      for (;;) {
        const bool published = (int) __hip_atomic_load(ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= F.epoch;
        const bool limit     = ++spins > SRRG2_FUSED_POLL_LIMIT;
        if (published || limit) {
          stale = false;
#pragma unroll
          for (int z = 0; z < MAXS; ++z)
            if (z < ns) {
              g[z]  = pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane);
              stale = stale || (unsigned) (g[z] >> 32) != (unsigned) F.epoch;
            }
          if (!__any(stale)) break;
        }
        if (limit && !FIRST) break;  // (unreachable: see above)
        if constexpr (FIRST) if (limit) {
          // (the step itself BEHIND the loop: inside it, the loop-invariant scalar loads of four slices' records are hoisted
          // in front of the loop and spill the kernel's scalar registers into vector ones -- 60 -> 197 of them)
          bool old = true;
#pragma unroll
          for (int z = 0; z < 4; ++z)
            if (z < ns_all)
              old = old && (unsigned) (pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane) >> 32) ==
                             (unsigned) (F.epoch - 1);
          if (__all(old)) {
            fallback = true;
            break;
          }
          spins = 0;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    if constexpr (FIRST) {
      if (fallback) {
        unsigned long long ga[4];
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          ga[z] = 0ull;
          if (z < ns_all) ga[z] = pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane);
        }
        bool old = true;
#pragma unroll
        for (int z = 0; z < 4; ++z)
          if (z < ns_all) old = old && (unsigned) (ga[z] >> 32) == (unsigned) (F.epoch - 1);
        if (__all(old)) {  // (still whole at the previous epoch: computed here)
          unsigned nv[4];
          wave_control<6, 4, false, PRIORS>(Sv, ns_all, states, prob, ga, &nv);
#pragma unroll
          for (int z = 0; z < MAXS; ++z)
            if (z < ns) g[z] = (unsigned long long) nv[z];
          FUSED_FALLBACK_COUNT();
        } else {  // (its writer has started in the meantime: it finishes within microseconds)
          do {
            stale = false;
#pragma unroll
            for (int z = 0; z < MAXS; ++z)
              if (z < ns) {
                g[z]  = pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane);
                stale = stale || (unsigned) (g[z] >> 32) != (unsigned) F.epoch;
              }
          } while (__any(stale));
        }
      }
    }
#pragma unroll
    for (int z = 0; z < MAXS; ++z)
      if (z < ns) rec[z][lane] = (unsigned) g[z];
  }
  __syncthreads();
}
__device__ __forceinline__ void view_of_record(const unsigned* rec, PassView& v) {
#pragma unroll
  for (int i = 0; i < 12; ++i) v.T[i] = __int_as_float(__builtin_amdgcn_readfirstlane((int) rec[i]));
#pragma unroll
  for (int i = 0; i < 12; ++i) v.Tprev[i] = __int_as_float(__builtin_amdgcn_readfirstlane((int) rec[12 + i]));
  v.kexp = __builtin_amdgcn_readfirstlane((int) rec[PUB_G_KEXP]);
  const unsigned fl = (unsigned) __builtin_amdgcn_readfirstlane((int) rec[PUB_G_FLAGS]);
  v.stop   = (fl & PUB_FLAG_STOP) != 0;
  v.phase1 = (fl & PUB_FLAG_PHASE1) != 0;
  v.prior  = (fl & PUB_FLAG_PRIOR) != 0;
}

}  // namespace

namespace {

// NW = waves that reduce together (through LDS); NW == 1: every wave on its own, no barrier (the rare second phase of
// the converged pass).  Every lane contributed `per_lane` biased values to each of the entries [0, ACC_CHI_IN).
template <int NW>
__device__ __forceinline__ long long (&block_reduce_lds())[4][ACC_N] {
  __shared__ long long red[4][ACC_N];  // (one array per NW: the stages of a split reduction meet in it)
  return red;
}
// (STAGE: 0 = everything; 1 = only the wave's part -- its totals into LDS, after which the accumulators are dead --, 2 = only
// the workgroup's part: barrier, sum over the waves, atomics.  The converged pass runs its second phase between the two.)
template <int NW, int STAGE = 0>
__device__ __forceinline__ void block_reduce_store_biased(long long (&acc)[ACC_N], long long* __restrict__ partials,
                                                          int prob, int block, int per_lane) {
  long long (&red)[4][ACC_N] = block_reduce_lds<NW>();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if constexpr (STAGE != 2) {
    int my_index;
    const long long total = wave_transpose_reduce(acc, lane, my_index);
    if ((lane & 1) == 0) red[wid][my_index] = total;
  }
  if constexpr (STAGE == 1) return;
  if (NW > 1)
    __syncthreads();
  else
    wave_lds_sync();
  const int t = NW > 1 ? (int) threadIdx.x : lane;
  if (t < ACC_N) {
    long long v = 0;
    if (NW > 1) {
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][t];
    } else {
      v = red[wid][t];
    }
    // remove the bias (wrapping arithmetic: the biased sums may have wrapped, the true sums fit by construction)
    unsigned long long u = (unsigned long long) v;
    if (t < ACC_CHI_IN) u -= (unsigned long long) FX_MAGIC_BITS * (unsigned long long) (NW * 64 * per_lane);
    if (u != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(partials) +
                  ((size_t) prob * PARTIAL_SLOTS + ((block + wid * (NW == 1 ? 7 : 0)) & (PARTIAL_SLOTS - 1))) * ACC_N + t, u);
  }
}

// The factor arithmetic of factor_accumulate, straight-line: EVERY lane adds one biased value (the bit pattern of
// FX_MAGIC + integer) to each of the H / b entries -- a lane without a contribution adds FX_MAGIC itself (weight 0, rows
// forced to 0) --, so that the compiler sees no control flow around the 32 accumulators (with branches it re-materialises
// all of them on every path: ~150 register moves per point) and the bias is the same for every lane.
// FIRST: the accumulators are written, not added to (one point per thread: no adds at all).
template <int D, int ROWS, bool FIRST>
__device__ __forceinline__ uint8_t factor_accumulate_flat(float (&J)[ROWS][D], float (&e)[ROWS], bool found, int rk, float thr,
                                                          double scale, long long (&acc)[ACC_N], bool valid = true) {
  // (valid == false: a correspondence whose factor is suppressed -- counted, not linearised: factor_accumulate's `invalid`)
  float chi = e[0] * e[0];
#pragma unroll
  for (int r = 1; r < ROWS; ++r) chi = chi + e[r] * e[r];
  const bool ok         = found && valid && isfinite(chi);
  const bool kernelized = ok && rk != SRRG2_ROBUST_NONE && !(chi < thr);
  float w               = 1.f;
  if (__any(kernelized)) {  // (wave-uniform branch around the divisions; one value merges)
    const float wk = rk == SRRG2_ROBUST_CLAMP ? 0.f : (rk == SRRG2_ROBUST_SATURATED ? thr / chi : 1.0f / (1.0f + chi / thr));
    w              = kernelized ? wk : 1.f;
  }
  const bool contrib = ok && w != 0.f;
  const long long chi_fx = fx_bits(__fma_rn((double) (ok ? chi : 0.f), scale, FX_MAGIC));
  const long long c_corr = found ? 1 : 0, c_in = (ok && !kernelized) ? 1 : 0, c_out = kernelized ? 1 : 0;
  const long long x_in = (ok && !kernelized) ? chi_fx : 0, x_out = kernelized ? chi_fx : 0;
  if (FIRST) {
    acc[ACC_N_CORR] = c_corr; acc[ACC_N_IN] = c_in; acc[ACC_N_OUT] = c_out; acc[ACC_CHI_IN] = x_in; acc[ACC_CHI_OUT] = x_out;
  } else {
    acc[ACC_N_CORR] += c_corr; acc[ACC_N_IN] += c_in; acc[ACC_N_OUT] += c_out; acc[ACC_CHI_IN] += x_in; acc[ACC_CHI_OUT] += x_out;
  }
  const double ws = contrib ? (double) w * scale : 0.0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
#pragma unroll
    for (int a = 0; a < D; ++a) J[r][a] = contrib ? J[r][a] : 0.f;
    e[r] = contrib ? e[r] : 0.f;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double wj[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) wj[r] = ws * (double) J[r][a];
#pragma unroll
    for (int b = a; b < D; ++b) {
      double t = FX_MAGIC;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) t = __fma_rn(wj[r], (double) J[r][b], t);
      if (FIRST) acc[hidx(a, b)] = __double_as_longlong(t); else acc[hidx(a, b)] += __double_as_longlong(t);
    }
    double t = FX_MAGIC;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) t = __fma_rn(wj[r], (double) e[r], t);
    if (FIRST) acc[ACC_B + a] = __double_as_longlong(t); else acc[ACC_B + a] += __double_as_longlong(t);
  }
  if (D == 3) {  // (the 3-dof layout leaves entries of the 6 x 6 table unused: they carry the bias too)
#pragma unroll
    for (int k = 0; k < ACC_CHI_IN; ++k) {
      bool used = false;
#pragma unroll
      for (int a = 0; a < D; ++a) {
#pragma unroll
        for (int b = a; b < D; ++b) used |= k == hidx(a, b);
        used |= k == ACC_B + a;
      }
      if (!used) { if (FIRST) acc[k] = FX_MAGIC_BITS; else acc[k] += FX_MAGIC_BITS; }
    }
  }
  return !ok ? SRRG2_FACTOR_SUPPRESSED : (kernelized ? SRRG2_FACTOR_KERNELIZED : SRRG2_FACTOR_INLIER);
}

// residual rows of one matched point (the arithmetic of finish_point)
template <int DIM, bool PLANE>
__device__ __forceinline__ void point_rows(const float* T, float kk, const float4 p, float qx, float qy, float qz,
                                           const float4 f, const float4 nf,
                                           float (&J)[PLANE ? 1 : DIM][DIM == 3 ? 6 : 3], float (&e)[PLANE ? 1 : DIM]) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  if constexpr (DIM == 3) {
    float m[ROWS][3];
    if (PLANE) {
      e[0]    = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
      m[0][0] = (T[0] * nf.x + T[4] * nf.y) + T[8] * nf.z;
      m[0][1] = (T[1] * nf.x + T[5] * nf.y) + T[9] * nf.z;
      m[0][2] = (T[2] * nf.x + T[6] * nf.y) + T[10] * nf.z;
    } else {
      const float q[3]  = {qx, qy, qz};
      const float ff[3] = {f.x, f.y, f.z};
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        e[r]    = q[r] - ff[r];
        m[r][0] = T[r * 4 + 0];
        m[r][1] = T[r * 4 + 1];
        m[r][2] = T[r * 4 + 2];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      J[r][0]     = m[r][0];
      J[r][1]     = m[r][1];
      J[r][2]     = m[r][2];
      J[r][D - 3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
      J[r][D - 2] = kk * (p.z * m[r][0] - p.x * m[r][2]);
      J[r][D - 1] = kk * (p.x * m[r][1] - p.y * m[r][0]);
    }
  } else {
    float m[ROWS][2];
    if (PLANE) {
      e[0]    = nf.x * (qx - f.x) + nf.y * (qy - f.y);
      m[0][0] = T[0] * nf.x + T[4] * nf.y;
      m[0][1] = T[1] * nf.x + T[5] * nf.y;
    } else {
      const float q[2]  = {qx, qy};
      const float ff[2] = {f.x, f.y};
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        e[r]    = q[r] - ff[r];
        m[r][0] = T[r * 4 + 0];
        m[r][1] = T[r * 4 + 1];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      J[r][0] = m[r][0];
      J[r][1] = m[r][1];
      J[r][2] = m[r][1] * p.x - m[r][0] * p.y;
    }
  }
}

}  // namespace

// (experiment, -DSRRG2_NT_LOADS: the per-point arrays of the converged pass -- read once per pass, 512 MB per pass of a
// 256-batch -- loaded non-temporally so that they do not evict the shared fixed clouds from L2 / MALL)
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#ifdef SRRG2_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#ifdef SRRG2_NT_LOADS
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
// GATHER: the previous neighbour and its normal are gathered from the fixed cloud through prev_pos (batches: the cloud
// is shared by all alignments and stays in L2; 8 instead of 36 streamed bytes per point) instead of read from prev_f / prev_n
// (single alignments: no dependent load on the chain).
template <int DIM, bool PLANE, int PPT, bool GATHER, int FUSED>  // (FUSED: 0 = control launches, 1 = fused control steps, 2 = ... of an aligner with prior slices)
__device__ __forceinline__ void icp_step_fast_body(const SliceDev& S, const ProblemDev* __restrict__ probs,
                                                   ProblemState* __restrict__ states) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  // (a launch may cover a sub-range of the batch: SliceDev::prob0.  Fused control steps: x = problem, y = tile -- the
  // dispatcher walks x first, so the workgroups (problem, tile 0) that carry the control steps of ALL problems start first)
  const int prob     = (FUSED ? blockIdx.x : blockIdx.y) + S.prob0;
  const int tile     = FUSED ? (int) blockIdx.y : (int) blockIdx.x;
  const ProblemState* st = &states[prob];
  PassView pv;
  if constexpr (!FUSED) {
    pass_view_legacy(S, st, pv);
    if (pv.stop) return;
  } else {
    PASS_TS(S.fc.epoch, 0);
    fused_control_if_due<DIM, FUSED == 2>(S, states, prob);
  }
  const ProblemDev pd = probs[prob];
  // (batches of unequal clouds; fused control steps: workgroup (0, problem) carries the control step of the previous
  // iteration whatever its share of the points)
  if (tile * (256 * PPT) >= pd.nm && (!FUSED || tile != 0)) return;
  float T[12], Tprev[12];
  double scale;
  int rk;
  if constexpr (!FUSED) {
    load_T(pv.T, T);
    load_T(pv.Tprev, Tprev);
    scale = dm::pow2(pv.kexp);
    rk    = (pv.phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  }
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  const bool ext     = !(S.tune & 65536);
  const float gfar   = ext ? g.gate2_ext : g.gate2;
  const int rfar     = ext ? g.rmax : g.rfar_gate;
  const float gate_r = S.gate * 1.000001f;  // (>= sqrt(gate2))
  const bool ngate   = S.use_normal_gate != 0;
  const bool use_q   = !FUSED && S.queue != nullptr && st->qmode[S.slice_idx] != 0;  // (fused control steps: no queue)
  const bool cert_a  = !(S.tune & 4096), cert_c = !(S.tune & (4096 | 65536));
  const int lane     = threadIdx.x & 63;
  const int wid      = threadIdx.x >> 6;
  // (the row tables of the cooperative grid scans -- 64-lane scans need 264 ints, four 16-lane teams 4 x 72 -- or, when the
  // grid has cell neighbour lists, the wave's pool of cnl_search: never live together)
  union FastWaveLds {
    CnlWave cnl;
    int coop[288];
  };
  __shared__ FastWaveLds fast_lds[4];
  int* const coop_lds_w = fast_lds[wid].coop;

  // gates, rows and factor terms of a point whose nearest neighbour {fk, nk} is known (valid); straight-line
  auto linearize = [&](auto first, long long (&acc)[ACC_N], bool valid, const float4 pk, const float4 fk, const float4 nk,
                       const float4 nmk, float qx, float qy, float qz, float best) {
    bool found = valid && __float_as_int(fk.w) != NO_MATCH && best <= g.gate2;
    if (ngate) {
      float dot;
      if constexpr (DIM == 3) {
        const float rx = (T[0] * nmk.x + T[1] * nmk.y) + T[2] * nmk.z;
        const float ry = (T[4] * nmk.x + T[5] * nmk.y) + T[6] * nmk.z;
        const float rz = (T[8] * nmk.x + T[9] * nmk.y) + T[10] * nmk.z;
        dot            = (nk.x * rx + nk.y * ry) + nk.z * rz;
      } else {
        const float rx = T[0] * nmk.x + T[1] * nmk.y;
        const float ry = T[4] * nmk.x + T[5] * nmk.y;
        dot            = nk.x * rx + nk.y * ry;
      }
      found = found && dot > S.normal_cos;
    }
    float J[ROWS][D], e[ROWS];
    point_rows<DIM, PLANE>(T, kk, pk, qx, qy, qz, fk, nk, J, e);
    (void) factor_accumulate_flat<D, ROWS, decltype(first)::value>(J, e, found, rk, thr, scale, acc);
  };

  long long acc[ACC_N];

  // all loads of the PPT points first (independent: one round trip for the lot)
  float4 p[PPT], pf[PPT], pnm[PPT], pn[PPT];
  float pm[PPT];
  int gi_[PPT];
  bool inr[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int i = (tile * PPT + k) * 256 + (int) threadIdx.x;
    inr[k]      = i < pd.nm;
    gi_[k]      = pd.moff + (inr[k] ? i : 0);  // (out of range: the loads below read point 0 of the problem, masked out later)
    int ppos    = -1;
    p[k]        = ld_stream(S.mpts + gi_[k]);
    pm[k]       = ld_stream(S.prev_m + gi_[k]);
    pf[k]       = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
    pn[k]       = make_float4(0.f, 0.f, 0.f, 0.f);
    pnm[k]      = make_float4(0.f, 0.f, 0.f, 0.f);
    if (GATHER) {
      ppos = ld_stream(S.prev_pos + gi_[k]);
    } else {
      pf[k] = S.prev_f[gi_[k]];
      if (PLANE || ngate) pn[k] = S.prev_n[gi_[k]];
    }
    if (ngate) pnm[k] = ld_stream(S.mnrm + gi_[k]);
    if (GATHER && ppos >= 0 && ppos < g.n) {
      pf[k] = g.pts[ppos];
      if (PLANE || ngate) pn[k] = g.nrm[ppos];
    }
  }
  if constexpr (FUSED) {  // (the points are on their way: now the record, or the control step it still waits for)
    pass_view_fused<DIM, FUSED == 2>(S, states, prob, pv);
    if (pv.stop || tile * (256 * PPT) >= pd.nm) return;
    load_T(pv.T, T);
    load_T(pv.Tprev, Tprev);
    scale = dm::pow2(pv.kexp);
    rk    = (pv.phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  }

  float open_ball2[PPT];  // squared radius of the ball to search; < 0: not open
  bool any_open = false;
  // the searches of the points whose certificate failed (wave-uniform control flow; results in the lanes that own the points)
  auto search_open = [&](float (&sbest)[PPT], int (&sidx)[PPT], float (&sexcl)[PPT], int (&spos)[PPT]) {
  #pragma unroll
    for (int k = 0; k < PPT; ++k) {
      sbest[k] = INFINITY; sexcl[k] = 0.f; sidx[k] = NO_MATCH; spos[k] = 0;
      const bool open = open_ball2[k] >= 0.f;
      if (!__ballot(open)) continue;
      float qx = 0.f, qy = 0.f, qz = 0.f;
      if (open) transform_point<DIM>(T, p[k], qx, qy, qz);
      // The grid has cell neighbour lists and the wave has more than a few open points (a pass before convergence, a partial
      // overlap: C2 at 60 % overlap 24.9 -> 43.0 k it/s): all of them in one pooled search.  A settled pass leaves a wave one
      // or two: the 16-lane cooperative scans below finish those in fewer dependent round trips (C2 converged pass 8.6 us
      // against 11 us with the pooled search for every open point).
      if (g.list_R > 0 && __popcll(__ballot(open)) > 4) {
        unsigned long long skey;
        float sb2, sL;
        cnl_search<DIM, 1>(g, *g.lists, fast_lds[wid].cnl, lane, open, qx, qy, qz, open_ball2[k], gfar, skey, sb2, sL);
        if (open) {
          sbest[k] = key_best(skey);
          sidx[k]  = key_idx(skey);
          sexcl[k] = sqrtf(fminf(sb2, fminf(sL, g.gate2_ext))) * 0.99999f;
        }
        continue;
      }
      // No lists (the first compute() on a fixed cloud) and MANY open points -- the estimate still moves at iteration 3, or a
      // 2-D point-to-point alignment that creeps towards its solution: 2 700 of 3 000 certificates fail in every pass of a
      // 3000-beam scan pair, and four-at-a-time team scans took 42 us per pass where the search pass takes 16.  Every open lane
      // scans the 3^DIM block around its own query first (scan_radius1, the first phase of k_icp_step, same acceptance rule:
      // a candidate inside the radius the block is guaranteed to cover); what that does not settle goes to the teams.
      bool open_k = open;
      if constexpr (PPT == 1) {
        if (!(g.list_R > 0) && __popcll(__ballot(open)) > 4) {
          unsigned long long lkey = NO_KEY;
          float lb2 = INFINITY, lc2 = INFINITY;
          if (open)
            scan_radius1<DIM>(g, qx, qy, qz, cell_coord(qx, g.ox, g.inv_h), cell_coord(qy, g.oy, g.inv_h),
                              DIM == 3 ? cell_coord(qz, g.oz, g.inv_h) : 0, open_ball2[k], lkey, lb2, lc2);
          const float b2_1 = bound2_of(1, g.h);
          if (open && key_idx(lkey) != NO_MATCH && key_best(lkey) <= gfar && key_best(lkey) <= b2_1) {
            sbest[k] = key_best(lkey);
            sidx[k]  = key_idx(lkey);
            sexcl[k] = sqrtf(fminf(fminf(lb2, lc2), b2_1)) * 0.99999f;
            open_k   = false;
          }
        }
      }
      const int r2 = open_ball2[k] <= bound2_of(2, g.h) ? 2 : max(rfar, 3);
      // small balls (the usual case: the ball of the previous neighbour): four searches per pass, 16 lanes each
      unsigned long long near = __ballot(open_k && r2 == 2);
      while (near) {
        int src[4];
  #pragma unroll
        for (int t = 0; t < 4; ++t) {
          src[t] = near ? __ffsll((long long) near) - 1 : -1;
          near &= near - 1;
        }
        const int team = lane >> 4;
        const int mine = team == 0 ? src[0] : (team == 1 ? src[1] : (team == 2 ? src[2] : src[3]));
        const int from = mine >= 0 ? mine : lane;
        const float sqx = __shfl(qx, from), sqy = __shfl(qy, from), sqz = __shfl(qz, from), sball = __shfl(open_ball2[k], from);
        const int scx = cell_coord(sqx, g.ox, g.inv_h), scy = cell_coord(sqy, g.oy, g.inv_h);
        const int scz = DIM == 3 ? cell_coord(sqz, g.oz, g.inv_h) : 0;
        float wbest, wexcl2;
        int widx, wpos;
        coop_scan<DIM, 16>(g, lane, coop_lds_w, sqx, sqy, sqz, scx, scy, scz, mine >= 0 ? 2 : -1, sball, wbest, widx, wpos, wexcl2);
  #pragma unroll
        for (int t = 0; t < 4; ++t) {  // the result of team t goes to the lane that owns the point
          const float rb = __shfl(wbest, 16 * t), re = __shfl(wexcl2, 16 * t);
          const int ri = __shfl(widx, 16 * t), rp = __shfl(wpos, 16 * t);
          if (lane == src[t]) {
            sbest[k] = rb;
            sidx[k]  = ri;
            spos[k]  = rp;
            sexcl[k] = sqrtf(re) * 0.99999f;
          }
        }
      }
      unsigned long long todo = __ballot(open_k && r2 != 2);
      while (todo) {
        const int src = __ffsll((long long) todo) - 1;
        todo &= todo - 1;
        const float sqx = __shfl(qx, src), sqy = __shfl(qy, src), sqz = __shfl(qz, src);
        const int scx = cell_coord(sqx, g.ox, g.inv_h), scy = cell_coord(sqy, g.oy, g.inv_h);
        const int scz = DIM == 3 ? cell_coord(sqz, g.oz, g.inv_h) : 0;
        float wbest, wexcl2;
        int widx, wpos;
        coop_scan<DIM, 64>(g, lane, coop_lds_w, sqx, sqy, sqz, scx, scy, scz, __shfl(r2, src), __shfl(open_ball2[k], src),
                           wbest, widx, wpos, wexcl2);
        if (lane == src) {
          sbest[k] = wbest;
          sidx[k]  = widx;
          spos[k]  = wpos;
          sexcl[k] = sqrtf(wexcl2) * 0.99999f;
        }
      }
    }
  };
  constexpr bool defer = PPT == 1 && FUSED;  // (fused control steps: no deferred-search queue)
  float d_qx = 0.f, d_qy = 0.f, d_qz = 0.f, d_best = INFINITY;
  bool d_valid = false;
  // Phase 1: the certificates; points that keep their neighbour are linearised.  A failed certificate leaves the squared
  // radius of the ball to search (the ball contains the previous neighbour, hence the nearest one) behind.
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const bool active = inr[k] && finite3(p[k].x, p[k].y, p[k].z);
    float qx, qy, qz, px, py, pz;
    transform_point<DIM>(T, p[k], qx, qy, qz);
    transform_point<DIM>(Tprev, p[k], px, py, pz);
    const float ex = qx - px, ey = qy - py, ez = qz - pz;
    // (hardware square roots, ~1 ulp: every use below carries a 1e-5 safety factor)
    const float dl   = __builtin_amdgcn_sqrtf((ex * ex + ey * ey) + ez * ez);
    const bool hasp  = __float_as_int(pf[k].w) != NO_MATCH;
    unsigned long long k1 = NO_KEY;
    test_candidate<DIM>(pf[k], qx, qy, qz, true, k1);
    const float best = key_best(k1);
    const float d1   = __builtin_amdgcn_sqrtf(best);
    const float rhs  = pm[k] * 0.99999f;
    // (a) of icp_step_body: d(q, f*) + |q - q'| < m  =>  f* is still the unique nearest neighbour
    // (c): nothing within m of q'; gate + |q - q'| < m  =>  still no match
    const bool ca   = hasp && cert_a && (d1 * 1.00001f + dl * 1.00001f < rhs);
    const bool cc   = !hasp && cert_c && pm[k] > 0.f && (gate_r * 1.00001f + dl * 1.00001f < rhs);
    const bool have = active && (ca || cc);
    const float excl = pm[k] * 0.9999999f - dl * 1.00001f;
    const float pad  = fminf(2.f * dl, PAD_CAP * g.h) + PAD_MIN * g.h;
    const float rr   = (d1 + pad) * 1.00001f;
    const float r2box = hasp ? fminf(rr * rr, gfar) : gfar;
    open_ball2[k]     = (active && !have) ? r2box : -1.f;
    any_open |= active && !have;
#ifdef SRRG2_PASS_TIMELINE
    if constexpr (FUSED) {
      if (active && !have && S.fc.epoch >= 0 && S.fc.epoch < 16 && blockIdx.x == 0) {
        // census of the failed certificates: [epoch][499..511][5..7] of the stamp array are never stamped
        unsigned long long* c = &g_pass_ts[((size_t) S.fc.epoch * 512 + 500) * 8];
        const float gap = (pm[k] - d1) / g.h;  // margin left, in cells
        atomicAdd(&c[0], 1ull);
        if (!hasp) atomicAdd(&c[1], 1ull);
        else if (gap < 1e-4f) atomicAdd(&c[2], 1ull);
        else if (gap < 0.0202f) atomicAdd(&c[3], 1ull);
        else atomicAdd(&c[4], 1ull);
        if (dl > 1e-3f * g.h) atomicAdd(&c[5], 1ull);
        if (pm[k] <= 0.f) atomicAdd(&c[6], 1ull);
      }
    }
#endif
    if constexpr (defer) {  // (linearised below, together with what the searches find)
      d_qx = qx; d_qy = qy; d_qz = qz; d_best = best; d_valid = have && ca;
    } else if (k == 0) {
      linearize(std::true_type{}, acc, have && ca, p[k], pf[k], pn[k], pnm[k], qx, qy, qz, best);
    } else {
      linearize(std::false_type{}, acc, have && ca, p[k], pf[k], pn[k], pnm[k], qx, qy, qz, best);
    }
    if (have) S.prev_m[gi_[k]] = excl;
  }
  if constexpr (defer) {
    {
      // One point per thread (single alignments, small batches: the launch waits for its last wave): the searches FIRST, then
      // ONE linearisation of all 64 lanes -- kept and newly found neighbours alike -- and one reduction.  A settled pass
      // still has a handful of near-ties to search for (C2: 8 - 15 of 100 000 points, profiles/r6e); as a second phase with
      // its own linearisation, wave reduction and atomics they kept ten lone waves busy ~4 us after their siblings.
      if (__any(any_open)) {
        float sbest[PPT], sexcl[PPT];
        int sidx[PPT], spos[PPT];
        search_open(sbest, sidx, sexcl, spos);
        const bool open = open_ball2[0] >= 0.f;
        // (a near-tie searched again mostly finds the neighbour it had: coordinates, normal and position are in place)
        const bool same = open && sidx[0] != NO_MATCH && sidx[0] == __float_as_int(pf[0].w);
        if (open && !same) {
          pf[0] = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
          pn[0] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (sidx[0] != NO_MATCH) {  // the new neighbour and its normal
            spos[0] = g.pos_of[sidx[0]];
            pf[0]   = g.pts[spos[0]];
            if (PLANE || ngate) pn[0] = g.nrm[spos[0]];
          }
          S.prev_pos[gi_[0]] = sidx[0] != NO_MATCH ? spos[0] : -1;
          if (!GATHER) {
            S.prev_f[gi_[0]] = pf[0];
            if (PLANE || ngate) S.prev_n[gi_[0]] = pn[0];
          }
        }
        if (open) {
          S.prev_m[gi_[0]] = sexcl[0];
          d_best  = sbest[0];
          d_valid = true;
        }
      }
      linearize(std::true_type{}, acc, d_valid, p[0], pf[0], pn[0], pnm[0], d_qx, d_qy, d_qz, d_best);
      block_reduce_store_biased<4>(acc, S.partials, prob, tile, PPT);
      if constexpr (FUSED) PASS_TS(S.fc.epoch, 3);
      return;
    }
  }
  if (use_q) {
    // failed certificates go to the deferred-search kernel like the stragglers of k_icp_step
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const bool open               = open_ball2[k] >= 0.f;
      const unsigned long long need = __ballot(open);
      if (!need) continue;
      const int r2 = open_ball2[k] <= bound2_of(2, g.h) ? 2 : max(rfar, 3);
      const unsigned long long need_near = __ballot(open && r2 == 2);
      const unsigned long long need_far  = need & ~need_near;
      int base_near = 0, base_far = 0;
      if (lane == 0) {  // one atomic per wave and kind (rare once the estimate has settled)
        if (need_near) base_near = atomicAdd(&S.qcount[2 * prob], __popcll(need_near));
        if (need_far) base_far = atomicAdd(&S.qcount[2 * prob + 1], __popcll(need_far));
      }
      base_near = __shfl(base_near, 0);
      base_far  = __shfl(base_far, 0);
      if (open) {
        const unsigned long long below = (1ull << lane) - 1ull;
        QEntry q;
        q.i = (tile * PPT + k) * 256 + (int) threadIdx.x;
        q.r2 = r2; q.best = INFINITY; q.bidx = NO_MATCH; q.bpos = 0;
        transform_point<DIM>(T, p[k], q.qx, q.qy, q.qz);
        q.ball2 = open_ball2[k]; q.pad_ = 0;
        QEntry* qbase = reinterpret_cast<QEntry*>(S.queue) + pd.moff;
        if (r2 == 2)
          qbase[base_near + __popcll(need_near & below)] = q;
        else
          qbase[pd.nm - 1 - (base_far + __popcll(need_far & below))] = q;
      }
    }
  }
  // (the wave's totals go to LDS now -- the accumulators are dead from here on --; the workgroup adds them up AFTER the second
  // phase: a wave with failed certificates starts its searches without waiting for its siblings, which have nothing else
  // to do but wait for it anyway)
  block_reduce_store_biased<4, 1>(acc, S.partials, prob, tile, PPT);
  if (use_q || !__any(any_open)) {
    block_reduce_store_biased<4, 2>(acc, S.partials, prob, tile, PPT);
    if constexpr (FUSED) PASS_TS(S.fc.epoch, 3);
    return;
  }

  // Phase 2 (waves with a failed certificate and no queue; rare once the estimate has settled): the whole wave searches
  // the ball of each such point, one at a time, then the points are linearised into a second set of sums which the wave
  // reduces and adds on its own.  Phase 1's accumulators are dead by now: the search runs at the register footprint
  // of the scan, not on top of 64 accumulator registers.
  float sbest[PPT], sexcl[PPT];
  int sidx[PPT], spos[PPT];
  search_open(sbest, sidx, sexcl, spos);
  long long acc2[ACC_N];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const bool open = open_ball2[k] >= 0.f;
    float4 fk = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH)), nk = make_float4(0.f, 0.f, 0.f, 0.f);
    // (a settled pass searches again for near-ties, and finds the neighbour it had: coordinates, normal and position are in
    // the registers / in place -- two dependent round trips less at the tail of the launch)
    // (one point per thread only: with two the kept neighbours of both would stay live through the searches -- 16 registers
    // the batch kernels do not have)
    bool same = false;
    if constexpr (PPT == 1) same = open && sidx[k] != NO_MATCH && sidx[k] == __float_as_int(pf[k].w);
    if (same) {
      fk = pf[k];
      nk = pn[k];
    } else if (open && sidx[k] != NO_MATCH) {  // the new neighbour and its normal
      spos[k] = g.pos_of[sidx[k]];
      fk      = g.pts[spos[k]];
      if (PLANE || ngate) nk = g.nrm[spos[k]];
    }
    float qx, qy, qz;
    transform_point<DIM>(T, p[k], qx, qy, qz);
    if (k == 0)
      linearize(std::true_type{}, acc2, open, p[k], fk, nk, pnm[k], qx, qy, qz, sbest[k]);
    else
      linearize(std::false_type{}, acc2, open, p[k], fk, nk, pnm[k], qx, qy, qz, sbest[k]);
    if (open) {
      S.prev_m[gi_[k]] = sexcl[k];
      if (!same) {
        S.prev_pos[gi_[k]] = sidx[k] != NO_MATCH ? spos[k] : -1;
        if (!GATHER) {
          S.prev_f[gi_[k]] = fk;
          if (PLANE || ngate) S.prev_n[gi_[k]] = nk;
        }
      }
    }
  }
  block_reduce_store_biased<1>(acc2, S.partials, prob, tile, PPT);
  if constexpr (FUSED) PASS_TS_ANY(S.fc.epoch, 4);  // (a wave that ran the second phase)
  block_reduce_store_biased<4, 2>(acc, S.partials, prob, tile, PPT);  // (phase 1's totals of the workgroup)
  if constexpr (FUSED) PASS_TS(S.fc.epoch, 3);
}

template <int DIM, bool PLANE, int PPT, bool GATHER, int FUSED>  // (FUSED: 0 = control launches, 1 = fused control steps, 2 = ... of an aligner with prior slices)
__global__ __launch_bounds__(256) void k_icp_step_fast(SliceDev S, const ProblemDev* __restrict__ probs,
                                                       ProblemState* __restrict__ states) {
  icp_step_fast_body<DIM, PLANE, PPT, GATHER, FUSED>(S, probs, states);
}
// ============================================================================================
// The search pass over the cell neighbour lists (cnl_search above): TEAM lanes per moving point.
// ============================================================================================
// (FUSED: as k_icp_step_fast; 3 = the FIRST pass of a single alignment's compute() with the prologue inside, k_icp_step_cnl_init:
// Ci / inl = the kernel's extra arguments, pass_view_init / fused_init_tail)
template <int DIM, bool PLANE, int TEAM, int FUSED>
__device__ __forceinline__ void cnl_pass_body(const SliceDev& S, const GridLists& GL, const ProblemDev* __restrict__ probs,
                                              ProblemState* __restrict__ states, const CtlParams* Ci = nullptr,
                                              const InitInline* inl = nullptr, const InitBatch* bat = nullptr) {
  constexpr int NW  = 4;
  constexpr int PPB = NW * 64 / TEAM;  // moving points per workgroup
  const int prob    = (FUSED ? blockIdx.x : blockIdx.y) + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  PassView pv;
  if constexpr (!FUSED) {
    pass_view_legacy(S, &states[prob], pv);
    if (pv.stop) return;
  } else if constexpr (FUSED != 3) {
    PASS_TS(S.fc.epoch, 0);
    fused_control_if_due<DIM, FUSED == 2>(S, states, prob);
  }
  ProblemDev pd;
  if constexpr (FUSED == 3) {
    if (bat)
      pd = bat->pd[prob - Ci->prob0];  // (a part of a batch: the launch's own table)
    else
      pd = inl->pd[S.slice_idx];
  } else {
    pd = probs[prob];
  }
  const int tile      = FUSED ? blockIdx.y : blockIdx.x;
  // (batches of unequal clouds; fused control steps: workgroup (0, problem) carries the control step of the previous
  // iteration whatever its share of the points.  FUSED = 3: no step is due; the launch has ONE workgroup more than tiles, which
  // holds no point and writes the rest of the prologue while the others run -- on workgroup 0 that tail sat on the launch's
  // critical path, ~3 us)
  const bool tail_wg = FUSED == 3 && tile == (int) gridDim.y - 1;
  if (tile * PPB >= pd.nm && (!FUSED || (FUSED == 3 ? !tail_wg : tile != 0))) return;
  float T[12], Tprev[12];
  double scale;
  int rk;
  bool use_prior;
  if constexpr (!FUSED) {
    load_T(pv.T, T);
    load_T(pv.Tprev, Tprev);
    scale     = dm::pow2(pv.kexp);
    rk        = (pv.phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
    use_prior = pv.prior && !(S.tune & 4);
  } else {
    use_prior = S.fc.prior != 0 && !(S.tune & 4);  // (the host knows: every pass but the first of the first run)
  }
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  const float gfar = (use_prior && !(S.tune & 65536)) ? g.gate2_ext : g.gate2;

  __shared__ CnlWave wlds[NW];

  const int i        = tile * PPB + (int) threadIdx.x / TEAM;
  const int lane     = threadIdx.x & 63;
  const int wid      = threadIdx.x >> 6;
  const bool leader  = (threadIdx.x & (TEAM - 1)) == 0;
  const bool inrange = i < pd.nm;
  const int gi       = pd.moff + (inrange ? i : 0);
  float4 p           = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pf          = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
  float4 pn          = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 pnm         = make_float4(0.f, 0.f, 0.f, 0.f);
  float pm           = 0.f;
  if (inrange) {
    p = S.mpts[gi];
    if (use_prior) {
      pm = S.prev_m[gi];
      if (S.use_normal_gate) pnm = S.mnrm[gi];
      if (S.gather_prev) {  // (as in icp_step_body)
        const int ppos = S.prev_pos[gi];
        if (ppos >= 0 && ppos < S.grid.n) {
          pf = S.grid.pts[ppos];
          if (PLANE || S.use_normal_gate) pn = S.grid.nrm[ppos];
        }
      } else {
        pf = S.prev_f[gi];
        if (PLANE || S.use_normal_gate) pn = S.prev_n[gi];
      }
    }
  }
  if constexpr (FUSED) {  // (the points are on their way: now the record, or the control step it still waits for)
    if constexpr (FUSED == 3) {
      pass_view_init<DIM>(S, *Ci, *inl, prob, pd.nm, pv, bat);
      if (tail_wg && threadIdx.x < 64)  // (probs: the SLICE's table)
        fused_init_tail(*Ci, *inl, prob, const_cast<ProblemDev*>(probs) - (size_t) S.slice_idx * Ci->K, states, bat, pv, S.slice_idx);
    } else {
      pass_view_fused<DIM, FUSED == 2>(S, states, prob, pv);
    }
    if (pv.stop || tile * PPB >= pd.nm) return;
    load_T(pv.T, T);
    load_T(pv.Tprev, Tprev);
    scale = dm::pow2(pv.kexp);
    rk    = (pv.phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  }
  const bool has_prev = __float_as_int(pf.w) != NO_MATCH;
  const bool active   = inrange && finite3(p.x, p.y, p.z);
  float qx = 0.f, qy = 0.f, qz = 0.f;
  float best = INFINITY;
  int bidx = NO_MATCH;
  float excl = 0.f, r2box = INFINITY;
  bool skipped = false;
  if (active) {
    transform_point<DIM>(T, p, qx, qy, qz);
    // temporal coherence, exactly as in icp_step_body: (a) the previous neighbour is provably still the nearest,
    // (b) the search is trimmed to its ball, (c) still nothing within the gate
    if (use_prior && has_prev) {
      unsigned long long k1 = NO_KEY;
      test_candidate<DIM>(pf, qx, qy, qz, true, k1);
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      const float d1 = sqrtf(key_best(k1));
      if (d1 * 1.00001f + dl * 1.00001f < pm * 0.99999f && !(S.tune & 4096)) {
        skipped = true;
        best    = key_best(k1);
        bidx    = key_idx(k1);
        excl    = pm * 0.9999999f - dl * 1.00001f;
      } else {
        const float pad = fminf(2.f * dl, PAD_CAP * g.h) + PAD_MIN * g.h;
        const float rr  = (d1 + pad) * 1.00001f;
        r2box           = fminf(rr * rr, gfar);
      }
    } else if (use_prior && !has_prev && pm > 0.f && !(S.tune & (4096 | 65536))) {
      float px, py, pz;
      transform_point<DIM>(Tprev, p, px, py, pz);
      const float ex = qx - px, ey = qy - py, ez = qz - pz;
      const float dl = sqrtf((ex * ex + ey * ey) + ez * ez);
      if (sqrtf(g.gate2) * 1.00001f + dl * 1.00001f < pm * 0.99999f) {
        skipped = true;
        excl    = pm * 0.9999999f - dl * 1.00001f;
      }
    }
  }
  const bool need = active && !skipped && !KNOB(S.tune, 16);
  if constexpr (TEAM == 1 && SRRG2_CNL_REDEAL != 0) {
    // (experiment, see SRRG2_CNL_REDEAL) The queries of the workgroup are RE-DEALT between phase 0 and the header walk.  A
    // wave walks the headers as long as its longest list lasts: 6.98 steps of four headers on the first pass of C4 with
    // 39.8 of 64 lanes busy (tools/cnl_stats.py).  After phase 0 the length of a walk can be estimated (what is left of the
    // list x the share of the gate ball the pruning ball covers): the 256 queries are sorted into four length classes, the
    // short ones into the first waves, which then finish early and leave their issue slots to the other workgroups of
    // the SIMD while the last wave(s) take the long walks with more lanes busy.  The adopting lane gets (q, e, eend, L) through LDS -- the tuple of slot s lies in the pool of the wave
    // that adopts it, which is idle until its phase A -- and starts from an empty minimum; the owner merges what its
    // phase 0 found with the slot of the adopting lane afterwards: a minimum and a runner-up over the same candidates
    // whoever examined them, so the result is the same to the bit.
    __shared__ unsigned long long wcnt[NW];
    unsigned long long k0;
    float b20, L;
    int e, eend;
    cnl_phase0<DIM, 1>(g, GL, lane, need, qx, qy, qz, r2box, gfar, k0, b20, L, e, eend, S.fc.epoch);
    const float est = (float) (eend - e) * fminf(L * (1.f / gfar), 1.f);
    const int cls   = (est > 4.f ? 1 : 0) + (est > 8.f ? 1 : 0) + (est > 16.f ? 1 : 0);
    const unsigned long long m1 = __ballot(cls == 1), m2 = __ballot(cls == 2), m3 = __ballot(cls == 3);
    const unsigned long long m0 = ~(m1 | m2 | m3);
    if (lane == 0)
      wcnt[wid] = (unsigned long long) __popcll(m0) | ((unsigned long long) __popcll(m1) << 16) |
                  ((unsigned long long) __popcll(m2) << 32) | ((unsigned long long) __popcll(m3) << 48);
    __syncthreads();
    int slot;
    {
      unsigned long long tot = 0ull, before = 0ull;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const unsigned long long c = wcnt[k];
        tot += c;
        before += k < wid ? c : 0ull;
      }
      const unsigned long long mm = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
      // classes before mine, waves before mine within my class, lanes before mine within my wave
      const unsigned long long below = tot & ((1ull << (16 * cls)) - 1ull);  // (fields of the lower classes)
      slot = (int) ((below & 0xffffull) + ((below >> 16) & 0xffffull) + ((below >> 32) & 0xffffull)) +
             (int) ((before >> (16 * cls)) & 0xffffull) +
             (int) __builtin_amdgcn_mbcnt_hi((unsigned) (mm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned) mm, 0u));
    }
    {
      float* tw = reinterpret_cast<float*>(wlds[slot >> 6].pool);  // [6][64]
      const int sl = slot & 63;
      tw[sl]       = qx;
      tw[64 + sl]  = qy;
      tw[128 + sl] = qz;
      tw[192 + sl] = L;
      tw[256 + sl] = __int_as_float(e);
      tw[320 + sl] = __int_as_float(eend);
    }
    __syncthreads();
    {
      const float* tr = reinterpret_cast<const float*>(wlds[wid].pool);
      const float ax = tr[lane], ay = tr[64 + lane], az = tr[128 + lane], aL = tr[192 + lane];
      const int ae = __float_as_int(tr[256 + lane]), aend = __float_as_int(tr[320 + lane]);
      wave_lds_sync();  // (the pool is written next)
      unsigned long long ka = NO_KEY;
      float ba              = INFINITY;
      cnl_phaseAB<DIM, 1>(g, GL, wlds[wid], lane, ax, ay, az, ka, ba, aL, ae, aend, S.fc.epoch);
    }
    __syncthreads();
    if (need) {
      const unsigned long long ka = wlds[slot >> 6].key[slot & 63];
      const float ba              = __uint_as_float(wlds[slot >> 6].b2[slot & 63]);
      const unsigned long long lo = ka < k0 ? ka : k0, hi = ka < k0 ? k0 : ka;
      const float b2              = fminf(fminf(b20, ba), __uint_as_float((unsigned) (hi >> 32)));
      best = key_best(lo);
      bidx = key_idx(lo);
      excl = sqrtf(fminf(b2, fminf(L, g.gate2_ext))) * 0.99999f;
    }
  } else if (__any(need)) {
    unsigned long long bkey;
    float b2, L;
    cnl_search<DIM, TEAM>(g, GL, wlds[wid], lane, need, qx, qy, qz, r2box, gfar, bkey, b2, L, S.fc.epoch);
    if (need) {
      best = key_best(bkey);
      bidx = key_idx(bkey);
      // every fixed point that was not examined is farther than the ball (a pruned or unreached cell) or than the extended
      // gate (a cell outside the list): nothing but the winner is closer than this
      excl = sqrtf(fminf(b2, fminf(L, g.gate2_ext))) * 0.99999f;
    }
  }
  // ---- gates, rows, factor terms: straight-line as in the converged-pass kernel (every lane produces the 32 biased values,
  // a lane without a correspondence with weight 0: no control flow around the accumulators -- finish_point's branches made
  // the compiler re-materialise them on every path, ~150 register moves per point).  One lane of a team carries the point.
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  const bool ngate   = S.use_normal_gate != 0;
  const bool has     = active && leader && bidx != NO_MATCH;
  int bpos           = -1;
  float4 fm          = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
  float4 nf          = make_float4(0.f, 0.f, 0.f, 0.f);
  if (skipped) {  // (a kept neighbour comes with its coordinates and normal)
    fm = pf;
    nf = pn;
  } else if (has) {
    bpos = g.pos_of[bidx];  // (the searches track keys, not positions)
    fm   = g.pts[bpos];
    if (PLANE || ngate) nf = g.nrm[bpos];
  }
  bool found = has && best <= g.gate2;
  if (ngate) {
    float4 nmk = pnm;
    if (!use_prior && found) nmk = S.mnrm[gi];  // (with a prior the normal was loaded together with it)
    float dot;
    if constexpr (DIM == 3) {
      const float rx = (T[0] * nmk.x + T[1] * nmk.y) + T[2] * nmk.z;
      const float ry = (T[4] * nmk.x + T[5] * nmk.y) + T[6] * nmk.z;
      const float rz = (T[8] * nmk.x + T[9] * nmk.y) + T[10] * nmk.z;
      dot            = (nf.x * rx + nf.y * ry) + nf.z * rz;
    } else {
      const float rx = T[0] * nmk.x + T[1] * nmk.y;
      const float ry = T[4] * nmk.x + T[5] * nmk.y;
      dot            = nf.x * rx + nf.y * ry;
    }
    found = found && dot > S.normal_cos;
  }
  long long acc[ACC_N];
  {
    float J[ROWS][D], er[ROWS];
    point_rows<DIM, PLANE>(T, kk, p, qx, qy, qz, fm, nf, J, er);
    (void) factor_accumulate_flat<D, ROWS, true>(J, er, found, rk, thr, scale, acc);
  }
  if (inrange && leader) {
    if (!skipped) {  // (a skipped search keeps its neighbour: only the exclusion radius changes)
      S.prev_pos[gi] = has ? bpos : -1;
      if (!S.gather_prev) {
        S.prev_f[gi] = fm;  // (.w = NO_MATCH: none)
        if (PLANE || ngate) S.prev_n[gi] = nf;
      }
    }
    S.prev_m[gi] = excl;
  }
  block_reduce_store_biased<NW>(acc, S.partials, prob, tile, 1);
  if constexpr (FUSED) PASS_TS(S.fc.epoch, 3);
}
template <int DIM, bool PLANE, int TEAM, int FUSED>
__global__ __launch_bounds__(256) void k_icp_step_cnl(SliceDev S, GridLists GL, const ProblemDev* __restrict__ probs,
                                                      ProblemState* __restrict__ states) {
  cnl_pass_body<DIM, PLANE, TEAM, FUSED>(S, GL, probs, states);
}
// the first pass of a single alignment's compute() with the prologue inside (pass_view_init, fused_init_tail)
template <int DIM, bool PLANE, int TEAM>
__global__ __launch_bounds__(256) void k_icp_step_cnl_init(SliceDev S, GridLists GL, CtlParams C, InitInline inl,
                                                           ProblemDev* __restrict__ probs, ProblemState* __restrict__ states) {
  cnl_pass_body<DIM, PLANE, TEAM, 3>(S, GL, probs, states, &C, &inl);
}
// ... of up to INIT_BATCH_MAX alignments of one launch (a part of a pipelined batch): guesses and table rows in the arguments
template <int DIM, bool PLANE, int TEAM>
__global__ __launch_bounds__(256) void k_icp_step_cnl_init_batch(SliceDev S, GridLists GL, CtlParams C, InitBatch B,
                                                                 ProblemDev* __restrict__ probs, ProblemState* __restrict__ states) {
  const InitInline none{};
  cnl_pass_body<DIM, PLANE, TEAM, 3>(S, GL, probs, states, &C, &none, &B);
}

// The correspondence records of the nearest-neighbour passes, on demand (get_correspondences, factor status, the scene
// merger): slice->correspondences() and the factor statistics of the last linearisation (multi_aligner_impl.cpp:215,244)
// re-derived from what the last executed pass left behind -- every point's nearest neighbour (prev_f / prev_n) and the
// transform that pass ran with (ProblemState::Tlast) -- with the arithmetic of the pass: same bits.
template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_outputs(SliceDev S, const ProblemDev* __restrict__ probs,
                                                     const ProblemState* __restrict__ states) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  const int prob     = blockIdx.y + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  const ProblemState* st = &states[prob];
  const ProblemDev pd    = probs[prob];
  const int i            = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pd.nm) return;
  const int gi  = pd.moff + i;
  int match     = -1;
  float resp    = 0.f;
  uint8_t fstat = SRRG2_FACTOR_SUPPRESSED;
  const float4 p = S.mpts[gi];
  if (st->npasses > 0 && finite3(p.x, p.y, p.z)) {
    float4 f = make_float4(0.f, 0.f, 0.f, __int_as_float(NO_MATCH));
    int fpos = -1;
    if (S.gather_prev) {
      fpos = S.prev_pos[gi];
      if (fpos >= 0 && fpos < S.grid.n) f = S.grid.pts[fpos];
    } else {
      f = S.prev_f[gi];
    }
    if (__float_as_int(f.w) != NO_MATCH) {
      float T[12];
      load_T(st->Tlast[S.slice_idx], T);
      float qx, qy, qz;
      transform_point<DIM>(T, p, qx, qy, qz);
      unsigned long long k1 = NO_KEY;
      test_candidate<DIM>(f, qx, qy, qz, true, k1);
      const float best = key_best(k1);
      bool found       = best <= S.grid.gate2;
      float4 nf        = make_float4(0.f, 0.f, 0.f, 0.f);
      if (found && (PLANE || S.use_normal_gate)) nf = S.gather_prev ? S.grid.nrm[fpos] : S.prev_n[gi];
      if (found && S.use_normal_gate) {
        const float4 nm = S.mnrm[gi];
        float dot;
        if constexpr (DIM == 3) {
          const float rx = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
          const float ry = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
          const float rz = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
          dot            = (nf.x * rx + nf.y * ry) + nf.z * rz;
        } else {
          const float rx = T[0] * nm.x + T[1] * nm.y;
          const float ry = T[4] * nm.x + T[5] * nm.y;
          dot            = nf.x * rx + nf.y * ry;
        }
        if (!(dot > S.normal_cos)) found = false;
      }
      if (found) {
        match = key_idx(k1);
        resp  = best;
        const float kk = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
        float J[ROWS][D], e[ROWS];
        point_rows<DIM, PLANE>(T, kk, p, qx, qy, qz, f, nf, J, e);
        float chi = e[0] * e[0];
#pragma unroll
        for (int r = 1; r < ROWS; ++r) chi = chi + e[r] * e[r];
        const int rk = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
        if (isfinite(chi))
          fstat = (rk != SRRG2_ROBUST_NONE && !(chi < S.robust_thr)) ? SRRG2_FACTOR_KERNELIZED : SRRG2_FACTOR_INLIER;
      }
    }
  }
  S.corr_fixed[gi] = match;
  S.corr_resp[gi]  = resp;
  S.corr_stat[gi]  = fstat;
}

// Deferred searches: every wave takes queue entries w, w + W, ... of its problem, runs the cooperative exact scan
// for each (64 lanes per query), parks the result in the lane with that ordinal and, after at most 64 of them,
// finishes all parked points in SIMT.
template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_step_queue(SliceDev S, const ProblemDev* __restrict__ probs,
                                                        ProblemState* __restrict__ states) {
  const int prob   = blockIdx.y + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  ProblemState* st = &states[prob];
  if (st->done || st->finished) return;
  const ProblemDev pd = probs[prob];
  float T[12];
  load_T(st->Tf[S.slice_idx], T);
  const int kexp     = st->kexp[S.slice_idx];
  const double scale = dm::pow2(kexp);
  const int rk       = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float thr    = S.robust_thr;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const GridDev& g   = S.grid;
  __shared__ long long wave_acc[4][ACC_N];
  if ((threadIdx.x & 63) < ACC_N) wave_acc[threadIdx.x >> 6][threadIdx.x & 63] = 0;
  __shared__ int coop_lds[4][4 * (4 * 16 + 8)];  // = 288 ints >= 264 needed by the 64-lane scan
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int count_near = S.qcount[2 * prob], count_far = S.qcount[2 * prob + 1];
  const int W     = gridDim.x * 4;
  const QEntry* queue = reinterpret_cast<const QEntry*>(S.queue) + pd.moff;
  // Near entries (radius 2): four per pass, one per team of 16 lanes.  Far entries (radius > 2, large cubes): one
  // per pass with all 64 lanes.  Results are parked one per lane (slot = ordinal of the entry in this wave) and
  // finished in SIMT once 64 are parked or the queues are exhausted.
  bool have = false;
  int my_i = 0, my_bidx = NO_MATCH, my_bpos = 0;
  float my_best = INFINITY, my_excl = 0.f;
  int parked = 0;
  // The 32 fixed-point accumulators live in LDS between flushes (one set per wave), not in registers: the search
  // loop keeps its register budget (occupancy) and a flush costs one wave reduction.
  auto flush = [&]() {
    float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
    float mx = 0.f, my = 0.f, mz = 0.f;
    if (have) {
      mp = S.mpts[pd.moff + my_i];
      transform_point<DIM>(T, mp, mx, my, mz);
    }
    long long acc[ACC_N];
#pragma unroll
    for (int a = 0; a < ACC_N; ++a) acc[a] = 0;
    finish_point<DIM, PLANE>(S, T, rk, thr, kk, scale, have, have, pd.moff + my_i, pd.moff + __float_as_int(mp.w), mp, mx, my,
                             mz, my_best, my_bidx, my_bpos, my_excl, false, make_float4(0.f, 0.f, 0.f, 0.f),
                             make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), false, acc);
    int my_index;
    const long long total = wave_transpose_reduce(acc, lane, my_index);
    if ((lane & 1) == 0) wave_acc[wid][my_index] += total;
    have   = false;
    parked = 0;
  };
  {
    constexpr int TW = 16, TEAMS = 64 / TW;
    const int team = lane / TW;
    for (int e0 = (blockIdx.x * 4 + wid) * TEAMS; e0 < count_near; e0 += W * TEAMS) {
      const int e     = e0 + team;
      const bool live = e < count_near;
      QEntry q;
      q.i = 0; q.r2 = -1; q.best = INFINITY; q.bidx = NO_MATCH; q.bpos = 0; q.qx = q.qy = q.qz = 0.f;
      q.ball2 = 0.f; q.pad_ = 0;
      if (live) q = queue[e];
      const bool skip = KNOB(S.tune, 2048);
      const int cx = cell_coord(q.qx, g.ox, g.inv_h);
      const int cy = cell_coord(q.qy, g.oy, g.inv_h);
      const int cz = DIM == 3 ? cell_coord(q.qz, g.oz, g.inv_h) : 0;
      float wbest, wexcl2;
      int widx, wpos;
      coop_scan<DIM, TW>(g, lane, coop_lds[wid], q.qx, q.qy, q.qz, cx, cy, cz, (live && !skip) ? q.r2 : -1, q.ball2, wbest,
                         widx, wpos, wexcl2);
      const float qexcl = skip ? 0.f : sqrtf(wexcl2) * 0.99999f;
      if (wbest < q.best || (wbest == q.best && widx < q.bidx)) {
        q.best = wbest;
        q.bidx = widx;
        q.bpos = wpos;
      }
      // park: lane (parked + t) takes the result of team t
      const int slot_team = lane - parked;
      const bool mine     = slot_team >= 0 && slot_team < TEAMS;
      const int src       = mine ? slot_team * TW : 0;
      const int pi = __shfl(q.i, src), pbi = __shfl(q.bidx, src), pbp = __shfl(q.bpos, src), plive = __shfl((int) live, src);
      const float pb = __shfl(q.best, src), pex = __shfl(qexcl, src);
      if (mine && plive) {
        have    = true;
        my_i    = pi;
        my_best = pb;
        my_bidx = pbi;
        my_bpos = pbp;
        my_excl = pex;
      }
      parked += TEAMS;
      if (parked == 64) flush();
    }
  }
  for (int e = blockIdx.x * 4 + wid; e < count_far; e += W) {
    const QEntry q  = queue[pd.nm - 1 - e];
    const bool skip = KNOB(S.tune, 1024);
    const int cx = cell_coord(q.qx, g.ox, g.inv_h);
    const int cy = cell_coord(q.qy, g.oy, g.inv_h);
    const int cz = DIM == 3 ? cell_coord(q.qz, g.oz, g.inv_h) : 0;
    float wbest, wexcl2;
    int widx, wpos;
    int sr      = q.r2;
    float ball2 = q.ball2;
    if (q.bidx == NO_MATCH && q.r2 >= 5 && !skip && !(S.tune & 262144)) {  // (small cubes: one pass is cheaper)
      // Nothing is known about this point's neighbourhood and the ball is the whole gate: in a dense cloud that is
      // thousands of candidates.  Grow the cube instead (radius 2, 4, 8, ...) until something turns up, then scan once
      // more with the ball of that candidate (+ pad, for the exclusion radius).  Every pass is exact inside
      // min(ball, cube), so the final result is the same minimum; the passes are wave-uniform.
      for (int r = 2; r < q.r2; r *= 2) {
        const float b2 = fminf(bound2_of(r, g.h), q.ball2);
        coop_scan<DIM, 64>(g, lane, coop_lds[wid], q.qx, q.qy, q.qz, cx, cy, cz, r, b2, wbest, widx, wpos, wexcl2);
        if (widx != NO_MATCH) {  // (wave-uniform: every lane gets the same result)
          const float rr = (sqrtf(wbest) + (PAD_CAP + PAD_MIN) * g.h) * 1.00001f;
          ball2          = fminf(rr * rr, q.ball2);
          sr             = 1;
          while (sr < q.r2 && bound2_of(sr, g.h) < ball2) ++sr;
          break;
        }
      }
    }
    coop_scan<DIM, 64>(g, lane, coop_lds[wid], q.qx, q.qy, q.qz, cx, cy, cz, skip ? -1 : sr, ball2, wbest, widx, wpos,
                       wexcl2);
    if (lane == parked) {
      have    = true;
      my_excl = skip ? 0.f : sqrtf(wexcl2) * 0.99999f;
      my_i    = q.i;
      my_best = q.best;
      my_bidx = q.bidx;
      my_bpos = q.bpos;
      if (wbest < my_best || (wbest == my_best && widx < my_bidx)) {
        my_best = wbest;
        my_bidx = widx;
        my_bpos = wpos;
      }
    }
    if (++parked == 64) flush();
  }
  if (parked > 0) flush();
  __syncthreads();
  if (threadIdx.x < ACC_N) {
    const long long v = (wave_acc[0][threadIdx.x] + wave_acc[1][threadIdx.x]) + (wave_acc[2][threadIdx.x] + wave_acc[3][threadIdx.x]);
    if (v != 0)
      atomicAdd(reinterpret_cast<unsigned long long*>(S.partials) +
                  ((size_t) prob * PARTIAL_SLOTS + (blockIdx.x & (PARTIAL_SLOTS - 1))) * ACC_N + threadIdx.x,
                (unsigned long long) v);
  }
}

// ============================================================================================
// given correspondences (SRRG2_FINDER_CORRESPONDENCES): no search, one thread per correspondence of the problem;
// clouds in ingest order (the indices are the caller's).  MultiLoopDetectorHBST_::_computeAlignments,
// multi_loop_detector_hbst_impl.cpp:320-352.
// ============================================================================================
template <int DIM, bool PLANE>
__global__ __launch_bounds__(256) void k_icp_step_corr(SliceDev S, const ProblemDev* __restrict__ probs,
                                                       ProblemState* __restrict__ states) {
  constexpr int D    = DIM == 3 ? 6 : 3;
  constexpr int ROWS = PLANE ? 1 : DIM;
  const int prob     = blockIdx.y + S.prob0;  // (a launch may cover a sub-range of the batch: SliceDev::prob0)
  ProblemState* st   = &states[prob];
  if (st->done || st->finished) return;
  const ProblemDev pd = probs[prob];
  float T[12];
  load_T(st->Tf[S.slice_idx], T);
  const double scale = dm::pow2(st->kexp[S.slice_idx]);
  const int rk       = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  long long acc[ACC_N];
#pragma unroll
  for (int a = 0; a < ACC_N; ++a) acc[a] = 0;
  const int c0 = S.gcorr_off[prob], c1 = S.gcorr_off[prob + 1];
  const int c  = c0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (c < c1) {
    const srrg2_correspondence k = S.gcorr[c];
    const float4 p  = S.moving_raw[pd.moff + k.moving_idx];
    const float4 f  = S.fixed_org[k.fixed_idx];
    float qx, qy, qz;
    transform_point<DIM>(T, p, qx, qy, qz);
    float J[ROWS][D];
    float e[ROWS];
    if constexpr (DIM == 3) {
      float m[ROWS][3];
      if (PLANE) {
        const float4 nf = S.fixed_org_nrm[k.fixed_idx];
        e[0]    = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
        m[0][0] = (T[0] * nf.x + T[4] * nf.y) + T[8] * nf.z;
        m[0][1] = (T[1] * nf.x + T[5] * nf.y) + T[9] * nf.z;
        m[0][2] = (T[2] * nf.x + T[6] * nf.y) + T[10] * nf.z;
      } else {
        const float q[3]  = {qx, qy, qz};
        const float ff[3] = {f.x, f.y, f.z};
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          e[r]    = q[r] - ff[r];
          m[r][0] = T[r * 4 + 0];
          m[r][1] = T[r * 4 + 1];
          m[r][2] = T[r * 4 + 2];
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        J[r][0]     = m[r][0];
        J[r][1]     = m[r][1];
        J[r][2]     = m[r][2];
        J[r][D - 3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
        J[r][D - 2] = kk * (p.z * m[r][0] - p.x * m[r][2]);
        J[r][D - 1] = kk * (p.x * m[r][1] - p.y * m[r][0]);
      }
    } else {
      float m[ROWS][2];
      if (PLANE) {
        const float4 nf = S.fixed_org_nrm[k.fixed_idx];
        e[0]    = nf.x * (qx - f.x) + nf.y * (qy - f.y);
        m[0][0] = T[0] * nf.x + T[4] * nf.y;
        m[0][1] = T[1] * nf.x + T[5] * nf.y;
      } else {
        const float q[2]  = {qx, qy};
        const float ff[2] = {f.x, f.y};
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          e[r]    = q[r] - ff[r];
          m[r][0] = T[r * 4 + 0];
          m[r][1] = T[r * 4 + 1];
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        J[r][0] = m[r][0];
        J[r][1] = m[r][1];
        J[r][2] = m[r][1] * p.x - m[r][0] * p.y;
      }
    }
    // (a pair with a non-finite point gives a non-finite chi: Suppressed)
    S.gcorr_stat[c] = factor_accumulate<D, ROWS>(J, e, false, rk, S.robust_thr, scale, false, acc);
  }
  block_reduce_store<4>(acc, S.partials, prob, blockIdx.x);
}

// ============================================================================================
// projective finder + factors on an organised fixed cloud (BASELINE config C3)
//   k_proj_zbuf        z-buffer: per pixel the transformed moving point of minimum depth, ties -> smaller caller
//                      index, as ONE 64-bit atomicMin on the key (depth bits << 32 | index)  (deterministic)
//   k_icp_step_proj<R> per moving point: re-project, keep it iff it won its pixel, gates, point-to-plane (R = false)
//                      or pinhole reprojection (R = true) rows, shared factor arithmetic + reduction
// ============================================================================================
namespace {

#define PIX_BOUND 8.0f

__device__ __forceinline__ int project_point(const SliceDev& S, float qx, float qy, float qz, float& u, float& v) {
  if (!finite3(qx, qy, qz)) return -1;
  if (!(qz >= S.depth_min) || !(qz <= S.depth_max)) return -1;
  u              = (S.K[0] * qx) / qz + S.K[2];
  v              = (S.K[4] * qy) / qz + S.K[5];
  const float uf = u + 0.5f, vf = v + 0.5f;
  if (!(uf >= 0.f) || !(uf < (float) S.cols) || !(vf >= 0.f) || !(vf < (float) S.rows)) return -1;
  return (int) floorf(vf) * S.cols + (int) floorf(uf);
}

__device__ __forceinline__ void finder_transform3(const SliceDev& S, const ProblemState* st, float* T) {
  load_T(st->Tf[S.slice_idx], T);  // robot_in_sensor * X (finder_transform_of)
}

}  // namespace

// the projective slices of one aligner in ONE launch (blockIdx.z = slice): their passes are independent until the
// control step, and at ~5 us of launch floor per kernel two slices cost two launches less per iteration
struct SlicePack {
  SliceDev s[4];
  const ProblemDev* probs[4];
};

// LAST: the z-buffer of the last EXECUTED pass again (ProblemState::Tlast), into buffer 0, whatever the alignment's flags
// say: the correspondence records of the projective slices are derived on demand (k_proj_records)
template <bool LAST = false>
__device__ __forceinline__ void proj_zbuf_body(const SliceDev& S, const ProblemDev* __restrict__ probs,
                                               ProblemState* __restrict__ states) {
  const int prob   = blockIdx.y;
  ProblemState* st = &states[prob];
  if (LAST ? st->npasses <= 0 : (st->done || st->finished)) return;
  const ProblemDev pd = probs[prob];
  const int i         = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pd.nm) return;
  float T[12];
  if (LAST)
    load_T(st->Tlast[S.slice_idx], T);
  else
    finder_transform3(S, st, T);
  const float4 p = S.mpts[pd.moff + i];
  if (!finite3(p.x, p.y, p.z)) return;
  const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
  const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
  const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  float u, v;
  const int pix = project_point(S, qx, qy, qz, u, v);
  if (pix < 0) return;
  const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) __float_as_int(p.w);
  atomicMin(&S.zbuf[((size_t) (LAST ? 0 : S.zbuf_parity) * gridDim.y + prob) * S.rows * S.cols + pix], key);
}

__global__ __launch_bounds__(256) void k_proj_zbuf(SliceDev S, const ProblemDev* __restrict__ probs,
                                                   ProblemState* __restrict__ states) {
  proj_zbuf_body(S, probs, states);
}
__global__ __launch_bounds__(256) void k_proj_zbuf_pack(SlicePack P, ProblemState* __restrict__ states) {
  proj_zbuf_body(P.s[blockIdx.z], P.probs[blockIdx.z], states);
}
// Fused control steps (projective slices that share their association): the z-buffer pass is the first kernel of an
// iteration -- wave 0 of workgroup (0, problem) applies the control step of ALL slices of the previous iteration at its top;
// the transform comes from the record of the first slice.
// (PRIORS: the instantiation for aligners with prior slices next to the pack -- a motion model beside the projective slices of an
// RGB-D tracker --, linearised by the control wave: wave_prior)
template <bool PRIORS>
__global__ __launch_bounds__(256) void k_proj_zbuf_fz(SlicePack P, int nslices, ProblemState* __restrict__ states) {
  const SliceDev& S = P.s[0];
  const int prob    = blockIdx.y;
  fused_control_if_due_multi<4, PRIORS>(P.s, nslices, states, prob);
  const ProblemDev pd = P.probs[0][prob];
  const int i         = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inr      = i < pd.nm;
  float4 p            = make_float4(NAN, 0.f, 0.f, 0.f);
  if (inr) p = S.mpts[pd.moff + i];  // (requested before the record: it does not depend on the state)
  __shared__ unsigned rec[1][PUB_SLICE_GRANULES];
  records_fused<1, true, PRIORS>(P.s, 1, prob, rec, states, nslices);
  PassView pv;
  view_of_record(rec[0], pv);
  if (pv.stop || !inr) return;
  const float (&T)[12] = pv.T;
  if (!finite3(p.x, p.y, p.z)) return;
  const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
  const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
  const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  float u, v;
  const int pix = project_point(S, qx, qy, qz, u, v);
  if (pix < 0) return;
  const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) __float_as_int(p.w);
  atomicMin(&S.zbuf[((size_t) S.zbuf_parity * gridDim.y + prob) * S.rows * S.cols + pix], key);
}
// The z-buffer pass of the FIRST iteration of a compute() with the prologue inside (round 6, last; cnl_pass_body's FUSED = 3 for the
// projective packs): the finder transform from the kernel arguments, the rest of k_icp_init's work -- all slices' records, state,
// tables, the device copy of the control parameters -- by one wave of the block behind the last tile; the step kernel that follows
// reads it behind the kernel boundary.  (One slice's record in the arguments instead of the pack: pack + parameters exceed 4 KB.)
__global__ __launch_bounds__(256) void k_proj_zbuf_fz_init(SliceDev S, CtlParams C, InitInline inl, ProblemDev* __restrict__ probs,
                                                          ProblemState* __restrict__ states) {
  const int prob      = blockIdx.y;
  const ProblemDev pd = inl.pd[S.slice_idx];
  const int i         = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inr      = i < pd.nm;
  float4 p            = make_float4(NAN, 0.f, 0.f, 0.f);
  if (inr) p = S.mpts[pd.moff + i];
  float X[12], T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) X[k] = inl.guess[k];
  finder_transform_of(C.slices[S.slice_idx].Sinv, 3, X, T);
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 64) fused_init_tail_full(C, inl, prob, probs, states);
  if (!inr) return;
  if (!finite3(p.x, p.y, p.z)) return;
  const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
  const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
  const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  float u, v;
  const int pix = project_point(S, qx, qy, qz, u, v);
  if (pix < 0) return;
  const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) __float_as_int(p.w);
  atomicMin(&S.zbuf[((size_t) S.zbuf_parity * gridDim.y + prob) * S.rows * S.cols + pix], key);
}
__global__ __launch_bounds__(256) void k_proj_zbuf_last(SliceDev S, const ProblemDev* __restrict__ probs,
                                                        ProblemState* __restrict__ states) {
  proj_zbuf_body<true>(S, probs, states);
}

// Correspondence records {matched pixel, |dz|, factor status} of a projective slice, derived on demand from the z-buffer of
// the last executed pass (k_proj_zbuf_last has rebuilt it in buffer 0 of the slice that OWNS the association: S0 = S, or
// the first slice of a group that shares clouds and finder parameters) with the arithmetic of the pass: the same bits the
// pass used to store every iteration (9 bytes per point and slice: C3 wrote 5.5 of its 39.8 MB per iteration for them).
__global__ __launch_bounds__(256) void k_proj_records(SliceDev S0, SliceDev S, const ProblemDev* __restrict__ probs,
                                                      const ProblemState* __restrict__ states) {
  const int prob         = blockIdx.y;
  const ProblemState* st = &states[prob];
  const ProblemDev pd    = probs[prob];
  const int i            = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pd.nm) return;
  const int gi  = pd.moff + i;
  int match     = -1;
  float resp    = 0.f;
  uint8_t fstat = SRRG2_FACTOR_SUPPRESSED;
  const float4 p = S0.mpts[gi];
  if (st->npasses > 0 && finite3(p.x, p.y, p.z)) {
    float T[12];
    load_T(st->Tlast[S0.slice_idx], T);
    const unsigned long long* zcur = S0.zbuf + (size_t) prob * S0.rows * S0.cols;
    const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
    const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
    const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
    float u, v;
    const int pix = project_point(S0, qx, qy, qz, u, v);
    const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) __float_as_int(p.w);
    bool found = false;
    float4 f   = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix >= 0 && zcur[pix] == key) {
      f     = S0.fixed_org[pix];
      found = finite3(f.x, f.y, f.z);
    }
    const float dd = fabsf(f.z - qz);
    {
      const float dx = f.x - qx, dy = f.y - qy, dz = f.z - qz;
      const float d2 = (dx * dx + dy * dy) + dz * dz;
      const float g2 = (2.f * S0.gate) * (2.f * S0.gate);
      found          = found && dd <= S0.gate && d2 <= g2;
    }
    const bool repro = S.factor == SRRG2_SLICE_REPROJECTION;
    float4 nf        = make_float4(0.f, 0.f, 0.f, 0.f);
    if (found && (!repro || S.use_normal_gate) && S0.fixed_org_nrm) nf = S0.fixed_org_nrm[pix];
    if (S.use_normal_gate) {
      float4 nm = make_float4(0.f, 0.f, 0.f, 0.f);
      if (found && S0.mnrm) nm = S0.mnrm[gi];
      const float rx  = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
      const float ry  = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
      const float rz  = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
      const float dot = (nf.x * rx + nf.y * ry) + nf.z * rz;
      found           = found && dot > S.normal_cos;
    }
    if (found) {
      match = pix;
      resp  = dd;
      bool valid = true;
      float chi;
      if (repro) {
        valid          = f.z > 0.f;
        const float uq = (S.K[0] * qx) / qz + S.K[2], vq = (S.K[4] * qy) / qz + S.K[5];
        const float uf = (S.K[0] * f.x) / f.z + S.K[2], vf = (S.K[4] * f.y) / f.z + S.K[5];
        const float e0 = valid ? uq - uf : 0.f, e1 = valid ? vq - vf : 0.f;
        if (!(fabsf(e0) <= PIX_BOUND) || !(fabsf(e1) <= PIX_BOUND)) valid = false;
        chi = e0 * e0;
        chi = chi + e1 * e1;
      } else {
        const float e0 = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
        chi            = e0 * e0;
      }
      const int rk = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
      if (valid && isfinite(chi))
        fstat = (rk != SRRG2_ROBUST_NONE && !(chi < S.robust_thr)) ? SRRG2_FACTOR_KERNELIZED : SRRG2_FACTOR_INLIER;
    }
  }
  S.corr_fixed[gi] = match;
  S.corr_resp[gi]  = resp;
  S.corr_stat[gi]  = fstat;
}

template <bool REPRO>
__device__ __forceinline__ void step_proj_body(const SliceDev& S, const ProblemDev* __restrict__ probs,
                                               ProblemState* __restrict__ states) {
  constexpr int D    = 6;
  constexpr int ROWS = REPRO ? 2 : 1;
  const int prob     = blockIdx.y;
  ProblemState* st   = &states[prob];
  {  // the z-buffers ping-pong: this pass reads buffer `zbuf_parity` and resets the other one for the next pass (no
     // memset launch per iteration); done before the early exit so that a later phase finds a clean buffer
    const size_t npix        = (size_t) S.rows * S.cols;
    unsigned long long* next = S.zbuf + ((size_t) (1 - S.zbuf_parity) * gridDim.y + prob) * npix;
    for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < npix; k += (size_t) gridDim.x * blockDim.x)
      next[k] = ~0ull;
  }
  if (st->done || st->finished) return;
  const ProblemDev pd = probs[prob];
  float T[12];
  finder_transform3(S, st, T);
  const unsigned long long* zcur = S.zbuf + ((size_t) S.zbuf_parity * gridDim.y + prob) * S.rows * S.cols;
  const int kexp     = st->kexp[S.slice_idx];
  const double scale = dm::pow2(kexp);
  const int rk       = (st->phase == 1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
  const float kk     = S.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  // Straight-line like the converged pass of the nearest-neighbour slices (k_icp_step_fast): the match, the gates and the
  // validity of the factor are predicates, every lane produces all 32 values (weight 0 adds the bias alone), so that no
  // control flow surrounds the accumulators -- with branches the compiler re-materialises them on every path.
  long long acc[ACC_N];
  const int i    = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inr = i < pd.nm;
  const int gi   = pd.moff + (inr ? i : 0);
  float4 p       = make_float4(NAN, 0.f, 0.f, 0.f);
  if (inr) p = S.mpts[gi];
  const int ci      = __float_as_int(p.w);  // caller's index within the problem
  const bool active = inr && finite3(p.x, p.y, p.z);
  const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
  const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
  const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  float u, v;
  int pix = -1;
  if (active) pix = project_point(S, qx, qy, qz, u, v);
  const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) ci;
  bool found = false;
  float4 f   = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pix >= 0 && zcur[pix] == key) {
    f     = S.fixed_org[pix];
    found = finite3(f.x, f.y, f.z);
  }
  const float dd = fabsf(f.z - qz);
  {
    const float dx = f.x - qx, dy = f.y - qy, dz = f.z - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const float g2 = (2.f * S.gate) * (2.f * S.gate);
    found          = found && dd <= S.gate && d2 <= g2;
  }
  float4 nf = make_float4(0.f, 0.f, 0.f, 0.f);
  if (found && (!REPRO || S.use_normal_gate) && S.fixed_org_nrm) nf = S.fixed_org_nrm[pix];
  if (S.use_normal_gate) {  // (uniform)
    float4 nm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (found) nm = S.mnrm[gi];
    const float rx  = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
    const float ry  = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
    const float rz  = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
    const float dot = (nf.x * rx + nf.y * ry) + nf.z * rz;
    found           = found && dot > S.normal_cos;
  }
  float J[ROWS][D];
  float e[ROWS];
  float m[ROWS][3];
  bool valid = true;
  if (REPRO) {
    // (lanes without a match compute on zeros: their rows are forced to 0 by factor_accumulate_flat)
    valid          = f.z > 0.f;
    const float uq = (S.K[0] * qx) / qz + S.K[2], vq = (S.K[4] * qy) / qz + S.K[5];
    const float uf = (S.K[0] * f.x) / f.z + S.K[2], vf = (S.K[4] * f.y) / f.z + S.K[5];
    e[0]           = valid ? uq - uf : 0.f;
    e[ROWS - 1]    = valid ? vq - vf : 0.f;
    if (!(fabsf(e[0]) <= PIX_BOUND) || !(fabsf(e[ROWS - 1]) <= PIX_BOUND)) valid = false;
    const float iz = 1.0f / qz;
    const float g[2][3] = {{S.K[0] * iz, 0.f, -(((S.K[0] * qx) * iz) * iz)},
                           {0.f, S.K[4] * iz, -(((S.K[4] * qy) * iz) * iz)}};
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) m[r][k] = (T[0 * 4 + k] * g[r][0] + T[1 * 4 + k] * g[r][1]) + T[2 * 4 + k] * g[r][2];
  } else {
    e[0]    = (nf.x * (qx - f.x) + nf.y * (qy - f.y)) + nf.z * (qz - f.z);
    m[0][0] = (T[0] * nf.x + T[4] * nf.y) + T[8] * nf.z;
    m[0][1] = (T[1] * nf.x + T[5] * nf.y) + T[9] * nf.z;
    m[0][2] = (T[2] * nf.x + T[6] * nf.y) + T[10] * nf.z;
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    J[r][0] = m[r][0];
    J[r][1] = m[r][1];
    J[r][2] = m[r][2];
    J[r][3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
    J[r][4] = kk * (p.z * m[r][0] - p.x * m[r][2]);
    J[r][5] = kk * (p.x * m[r][1] - p.y * m[r][0]);
  }
  // (the correspondence record {pixel, |dz|, status} is not stored: k_proj_records derives it on demand)
  (void) factor_accumulate_flat<D, ROWS, true>(J, e, found, rk, S.robust_thr, scale, acc, valid);
  (void) dd;
  block_reduce_store_biased<4>(acc, S.partials, prob, blockIdx.x, 1);
}

// Projective slices that SHARE their clouds and their finder parameters (srrg2_aligner_share_clouds: in the reference two
// slices with the same fixed_slice_name / moving_slice_name read the same cloud objects of the scene,
// aligner_slice_processor_base_impl.cpp:27-50) also share their association: every slice's finder would project the same
// points with the same transform into an identical z-buffer.  ONE z-buffer pass (the first slice's) and ONE step launch:
// a thread projects its point, reads the z-buffer and the matched pixel once and evaluates every slice's factor rows in
// turn (own robustifier, own normal gate, own exponent), each reduced into that slice's slot sets.  Same pixel, same
// gates, same rows per slice => the same bits as the slices run one by one; C3 moves 36 instead of 68 MB per iteration.
template <bool FUSED>
__global__ __launch_bounds__(256) void k_icp_step_proj_fused(SlicePack P, int nslices, ProblemState* __restrict__ states) {
  constexpr int D     = 6;
  const SliceDev& S0  = P.s[0];  // the association is the first slice's
  const int prob      = blockIdx.y;
  ProblemState* st    = &states[prob];
  // (fused control steps: the state comes from the slices' records -- current: the z-buffer kernel ran before this one)
  __shared__ unsigned rec[FUSED ? 4 : 1][PUB_SLICE_GRANULES];
  PassView pv0;
  if constexpr (FUSED) {
    records_fused<4, false>(P.s, nslices, prob, rec, states, nslices);
    view_of_record(rec[0], pv0);
  }
  {  // ping-pong reset of the (one) z-buffer, as in step_proj_body
    const size_t npix        = (size_t) S0.rows * S0.cols;
    unsigned long long* next = S0.zbuf + ((size_t) (1 - S0.zbuf_parity) * gridDim.y + prob) * npix;
    for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < npix; k += (size_t) gridDim.x * blockDim.x)
      next[k] = ~0ull;
  }
  if (FUSED ? pv0.stop : (st->done || st->finished)) return;
  const ProblemDev pd = P.probs[0][prob];
  float T[12];
  if constexpr (FUSED)
    load_T(pv0.T, T);
  else
    finder_transform3(S0, st, T);
  const unsigned long long* zcur = S0.zbuf + ((size_t) S0.zbuf_parity * gridDim.y + prob) * S0.rows * S0.cols;
  const float kk = S0.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.f : 1.f;
  const int i    = blockIdx.x * blockDim.x + threadIdx.x;
  const bool inr = i < pd.nm;
  const int gi   = pd.moff + (inr ? i : 0);
  float4 p       = make_float4(NAN, 0.f, 0.f, 0.f);
  if (inr) p = S0.mpts[gi];
  const int ci      = __float_as_int(p.w);
  const bool active = inr && finite3(p.x, p.y, p.z);
  const float qx = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3];
  const float qy = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7];
  const float qz = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11];
  float u, v;
  int pix = -1;
  if (active) pix = project_point(S0, qx, qy, qz, u, v);
  const unsigned long long key = ((unsigned long long) __float_as_uint(qz) << 32) | (unsigned) ci;
  bool found0 = false;
  float4 f    = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pix >= 0 && zcur[pix] == key) {
    f      = S0.fixed_org[pix];
    found0 = finite3(f.x, f.y, f.z);
  }
  const float dd = fabsf(f.z - qz);
  {
    const float dx = f.x - qx, dy = f.y - qy, dz = f.z - qz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    const float g2 = (2.f * S0.gate) * (2.f * S0.gate);
    found0         = found0 && dd <= S0.gate && d2 <= g2;
  }
  // the matched pixel's normal and the point's own normal: loaded once if any slice wants them
  bool want_nf = false, want_nm = false;
#pragma unroll
  for (int z = 0; z < 4; ++z)
    if (z < nslices) {
      want_nf = want_nf || P.s[z].factor != SRRG2_SLICE_REPROJECTION || P.s[z].use_normal_gate;
      want_nm = want_nm || P.s[z].use_normal_gate;
    }
  float4 nf = make_float4(0.f, 0.f, 0.f, 0.f), nm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (found0 && want_nf && S0.fixed_org_nrm) nf = S0.fixed_org_nrm[pix];
  if (found0 && want_nm && S0.mnrm) nm = S0.mnrm[gi];
  float ndot = 0.f;
  {
    const float rx = (T[0] * nm.x + T[1] * nm.y) + T[2] * nm.z;
    const float ry = (T[4] * nm.x + T[5] * nm.y) + T[6] * nm.z;
    const float rz = (T[8] * nm.x + T[9] * nm.y) + T[10] * nm.z;
    ndot           = (nf.x * rx + nf.y * ry) + nf.z * rz;
  }
#pragma unroll
  for (int z = 0; z < 4; ++z) {
    if (z >= nslices) break;  // (uniform)
    const SliceDev& S  = P.s[z];
    const bool repro   = S.factor == SRRG2_SLICE_REPROJECTION;
    const int kexp_z   = FUSED ? __builtin_amdgcn_readfirstlane((int) rec[FUSED ? z : 0][PUB_G_KEXP]) : st->kexp[S.slice_idx];
    const bool phase1  = FUSED ? pv0.phase1 : st->phase == 1;
    const double scale = dm::pow2(kexp_z);
    const int rk       = (phase1 && S.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : S.robust_kind;
    bool found         = found0;
    // (a slice without normals of its own kind sees zeros, as step_proj_body does)
    const float4 nfz = (!repro || S.use_normal_gate) ? nf : make_float4(0.f, 0.f, 0.f, 0.f);
    if (S.use_normal_gate) found = found && ndot > S.normal_cos;
    long long acc[ACC_N];
    if (repro) {
      float J[2][D], e[2], m[2][3];
      bool valid     = f.z > 0.f;
      const float uq = (S.K[0] * qx) / qz + S.K[2], vq = (S.K[4] * qy) / qz + S.K[5];
      const float uf = (S.K[0] * f.x) / f.z + S.K[2], vf = (S.K[4] * f.y) / f.z + S.K[5];
      e[0]           = valid ? uq - uf : 0.f;
      e[1]           = valid ? vq - vf : 0.f;
      if (!(fabsf(e[0]) <= PIX_BOUND) || !(fabsf(e[1]) <= PIX_BOUND)) valid = false;
      const float iz = 1.0f / qz;
      const float g[2][3] = {{S.K[0] * iz, 0.f, -(((S.K[0] * qx) * iz) * iz)},
                             {0.f, S.K[4] * iz, -(((S.K[4] * qy) * iz) * iz)}};
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) m[r][k] = (T[0 * 4 + k] * g[r][0] + T[1 * 4 + k] * g[r][1]) + T[2 * 4 + k] * g[r][2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        J[r][0] = m[r][0]; J[r][1] = m[r][1]; J[r][2] = m[r][2];
        J[r][3] = kk * (p.y * m[r][2] - p.z * m[r][1]);
        J[r][4] = kk * (p.z * m[r][0] - p.x * m[r][2]);
        J[r][5] = kk * (p.x * m[r][1] - p.y * m[r][0]);
      }
      (void) factor_accumulate_flat<D, 2, true>(J, e, found, rk, S.robust_thr, scale, acc, valid);
    } else {
      float J[1][D], e[1], m[3];
      e[0] = (nfz.x * (qx - f.x) + nfz.y * (qy - f.y)) + nfz.z * (qz - f.z);
      m[0] = (T[0] * nfz.x + T[4] * nfz.y) + T[8] * nfz.z;
      m[1] = (T[1] * nfz.x + T[5] * nfz.y) + T[9] * nfz.z;
      m[2] = (T[2] * nfz.x + T[6] * nfz.y) + T[10] * nfz.z;
      J[0][0] = m[0]; J[0][1] = m[1]; J[0][2] = m[2];
      J[0][3] = kk * (p.y * m[2] - p.z * m[1]);
      J[0][4] = kk * (p.z * m[0] - p.x * m[2]);
      J[0][5] = kk * (p.x * m[1] - p.y * m[0]);
      (void) factor_accumulate_flat<D, 1, true>(J, e, found, rk, S.robust_thr, scale, acc, true);
    }
    // (no correspondence records: k_proj_records derives them on demand)
    if (z > 0) __syncthreads();  // (the reduction's scratch in LDS is the previous slice's until every thread has left it)
    block_reduce_store_biased<4>(acc, S.partials, prob, blockIdx.x, 1);
  }
}

template <bool REPRO>
__global__ __launch_bounds__(256) void k_icp_step_proj(SliceDev S, const ProblemDev* __restrict__ probs,
                                                       ProblemState* __restrict__ states) {
  step_proj_body<REPRO>(S, probs, states);
}
__global__ __launch_bounds__(256) void k_icp_step_proj_pack(SlicePack P, ProblemState* __restrict__ states) {
  const SliceDev& S = P.s[blockIdx.z];
  if (S.factor == SRRG2_SLICE_REPROJECTION)
    step_proj_body<true>(S, P.probs[blockIdx.z], states);
  else
    step_proj_body<false>(S, P.probs[blockIdx.z], states);
}

// ============================================================================================
// control kernels (one thread per problem)
// ============================================================================================
namespace {

__device__ __forceinline__ int slice_exponent(const CtlParams& C, const SliceCtl& s, int prob, int nm, const float* X) {
  const bool plane = s.kind == SRRG2_SLICE_P2PLANE;
  const bool repro = s.kind == SRRG2_SLICE_REPROJECTION;
  const bool proj  = s.finder == SRRG2_FINDER_PROJECTIVE;
  const double kk  = C.variable_kind == SRRG2_SE3_QUAT_RIGHT ? 2.0 : 1.0;
  const int dim    = C.variable_kind == SRRG2_SE2_RIGHT ? 2 : 3;
  const int rows   = plane ? 1 : (repro ? 2 : dim);
  const float pinf = __uint_as_float(s.pinf_bits[prob]);
  const float ninf = __uint_as_float(s.ninf_bits[0]);
  double mb        = plane ? (1.7320508075688772 * (double) ninf) * 1.01 : 1.01;
  if (repro) {
    const double K0 = (double) s.K0, K4 = (double) s.K4;
    const double tx = (double) s.cols / K0, ty = (double) s.rows / K4;
    const double gb = (((K0 > K4 ? K0 : K4) / (double) s.depth_min) * (1.0 + (tx > ty ? tx : ty))) * 1.01;
    mb              = (1.7320508075688772 * gb) * 1.01;
  }
  double pf = (2.0 * kk) * (double) pinf;
  double jb = mb * (pf > 1.0 ? pf : 1.0);
  double eb = (mb * (double) s.gate) * 1.01;
  if (proj) eb = (mb * (2.0 * (double) s.gate)) * 1.01;
  if (repro) eb = (double) PIX_BOUND * 1.01;
  int n_terms = nm;
  if (s.finder == SRRG2_FINDER_CORRESPONDENCES) {
    // no gate bounds the residual: |e_r| <= sqrt3 (sqrt3 |p|inf + |t|inf + |f|inf) with the CURRENT estimate
    double tmax = 0.0;
    if (dim == 3) {
      for (int r = 0; r < 3; ++r) tmax = fabs((double) X[r * 4 + 3]) > tmax ? fabs((double) X[r * 4 + 3]) : tmax;
    } else {
      for (int r = 0; r < 2; ++r) tmax = fabs((double) X[r * 3 + 2]) > tmax ? fabs((double) X[r * 3 + 2]) : tmax;
    }
    const float finf = __uint_as_float(s.finf_bits[0]);
    eb          = ((mb * 1.7320508075688772) * ((1.7320508075688772 * (double) pinf + tmax) + (double) finf)) * 1.01;
    const int ng = s.gcorr_off[prob + 1] - s.gcorr_off[prob];
    n_terms     = ng > 1 ? ng : 1;
  }
  double mx = jb > eb ? jb : eb;
  double B  = (double) rows * (mx * mx);
  return dm::fixed_point_exponent(n_terms, B);
}

__device__ float robust_weight(int kind, float thr, float chi, bool& kernelized) {
  if (kind == SRRG2_ROBUST_NONE || chi < thr) {
    kernelized = false;
    return 1.f;
  }
  kernelized = true;
  if (kind == SRRG2_ROBUST_CLAMP) return 0.f;
  if (kind == SRRG2_ROBUST_SATURATED) return thr / chi;
  return 1.0f / (1.0f + chi / thr);
}

// SE2PriorErrorFactor / SE3PriorErrorFactorAD ([EXT]) at the current estimate
template <int D>
__device__ void prior_linearize(int variable_kind, const SliceCtl& s, int rk, const float* X, double* H, double* b,
                                double& chi_out, int& status) {
  double e[D], J[D * D];
  for (int i = 0; i < D * D; ++i) J[i] = 0.0;
  float Zinv[12], E[12];
  if (D == 6) {
    dm::se3_inverse(s.prior_Z, Zinv);
    dm::se3_compose(Zinv, X, E);
    dm::se3_t2v_quat(E, e);
    double n2 = (e[3] * e[3] + e[4] * e[4]) + e[5] * e[5];
    double w  = n2 < 1.0 ? sqrt(1.0 - n2) : 0.0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) J[i * 6 + j] = (double) E[i * 4 + j];
    J[3 * 6 + 3] = w;      J[3 * 6 + 4] = -e[5];  J[3 * 6 + 5] = e[4];
    J[4 * 6 + 3] = e[5];   J[4 * 6 + 4] = w;      J[4 * 6 + 5] = -e[3];
    J[5 * 6 + 3] = -e[4];  J[5 * 6 + 4] = e[3];   J[5 * 6 + 5] = w;
  } else {
    dm::se2_inverse(s.prior_Z, Zinv);
    dm::se2_compose(Zinv, X, E);
    dm::se2_t2v(E, e);
    J[0] = (double) E[0]; J[1] = (double) E[1];
    J[3] = (double) E[3]; J[4] = (double) E[4];
    J[8] = 1.0;
  }
  double chi = 0.0;
  for (int i = 0; i < D; ++i) chi = chi + (e[i] * (double) s.prior_info[i]) * e[i];
  bool kernelized;
  float w = robust_weight(rk, s.robust_thr, (float) chi, kernelized);
  chi_out = chi;
  status  = !isfinite(chi) ? (int) SRRG2_FACTOR_SUPPRESSED
                           : (kernelized ? (int) SRRG2_FACTOR_KERNELIZED : (int) SRRG2_FACTOR_INLIER);
  for (int i = 0; i < D * D; ++i) H[i] = 0.0;
  for (int i = 0; i < D; ++i) b[i] = 0.0;
  if (status == SRRG2_FACTOR_SUPPRESSED) return;
  for (int a = 0; a < D; ++a) {
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
      for (int r = 0; r < D; ++r) t = t + (J[r * D + a] * (double) s.prior_info[r]) * J[r * D + c];
      H[a * D + c] = (double) w * t;
    }
    double t = 0.0;
    for (int r = 0; r < D; ++r) t = t + (J[r * D + a] * (double) s.prior_info[r]) * e[r];
    b[a] = (double) w * t;
  }
}

__device__ int num_correspondences(const CtlParams& C, const ProblemState* st) {
  int n = 0;
  for (int s = 0; s < C.nslices; ++s) {
    int c = C.slices[s].kind == SRRG2_SLICE_PRIOR ? 1 : st->ncorr[s];
    if (c >= 0) n += c;
  }
  return n;
}

__device__ double win_max(const double* b, int n) {
  double m = b[0];
  for (int i = 1; i < n; ++i)
    if (b[i] > m) m = b[i];
  return m;
}
__device__ double win_min(const double* b, int n) {
  double m = b[0];
  for (int i = 1; i < n; ++i)
    if (b[i] < m) m = b[i];
  return m;
}

// AlignerTerminationCriteriaStandard_::hasToStop, aligner_termination_criteria_impl.cpp:24-65
__device__ bool has_to_stop(const CtlParams& C, ProblemState* st, const srrg2_iteration_stats& cur) {
  int ncorr = num_correspondences(C, st);
  int ninl  = cur.num_inliers;
  int nout  = cur.num_outliers;
  float chi = cur.chi_inliers / (float) ninl;
  if (!ninl) return false;
  const int W = C.term.window_size;
  int slot    = st->w_count % W;
  st->w_corr[slot] = ncorr;
  st->w_inl[slot]  = ninl;
  st->w_out[slot]  = nout;
  st->w_chi[slot]  = (double) chi;
  st->w_count++;
  int n = st->w_count < W ? st->w_count : W;
  if (n < W) return false;
  if (win_max(st->w_out, n) - win_min(st->w_out, n) > (double) C.term.num_correspondences_range) return false;  // :46
  if (win_max(st->w_inl, n) - win_min(st->w_inl, n) > (double) C.term.num_inliers_range) return false;
  float chi_range = (float) (win_max(st->w_chi, n) - win_min(st->w_chi, n));
  if (chi_range > (float) C.term.num_outliers_range) return false;  // :53
  if (chi_range / (float) win_max(st->w_chi, n) > C.term.chi_epsilon) return false;
  return true;
}

template <int D>
// sums: the exact integer sums of every cue slice; scaled: (double) sums[s][k] * 2^-kexp[s], converted by one lane per
// entry before this (sequential) body runs -- the same operation it used to do itself, 30 times in a row.
__device__ void control_body(const CtlParams& C, ProblemState* st, srrg2_iteration_stats* stats, int prob,
                             const long long (*sums)[ACC_N], const double (*scaled)[ACC_N]) {
  st->npasses++;
  for (int s = 0; s < C.nslices; ++s)  // the transforms the passes of this iteration ran with (k_icp_outputs)
    if (C.slices[s].kind != SRRG2_SLICE_PRIOR)
      for (int i = 0; i < 12; ++i) st->Tlast[s][i] = st->Tf[s][i];
  // association check: association_good |= slice->correspondencesGood(), multi_aligner.h:126-138
  bool good = false;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      good = true;  // aligner_slice_processor_prior.h:66-68
      continue;
    }
    const long long* acc = sums[s];
    int nc               = (int) acc[ACC_N_CORR];
    st->ncorr[s] = nc;
    good |= nc > sc.min_num_correspondences;  // aligner_slice_processor_impl.cpp:77-79
  }
  if (!good && !KNOB(C.tune, 256)) {
    for (int s = 0; s < C.nslices; ++s)
      if (C.slices[s].qcount) C.slices[s].qcount[2 * prob] = C.slices[s].qcount[2 * prob + 1] = 0;
    st->status = SRRG2_NOT_ENOUGH_CORRESPONDENCES;  // multi_aligner_impl.cpp:107-111
    st->done   = 1;
    return;
  }
  // solver->compute(): one Gauss-Newton iteration over all slices' factors
  double H[D * D], b[D], dx[D];
#pragma unroll
  for (int i = 0; i < D * D; ++i) H[i] = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) b[i] = 0.0;
  srrg2_iteration_stats cur;
  cur.iteration = st->nstats;
  cur.num_inliers = cur.num_outliers = cur.num_suppressed = 0;
  double chi_in = 0.0, chi_out = 0.0;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    const int rk = (st->phase == 1 && sc.robust_kind != SRRG2_ROBUST_NONE) ? (int) SRRG2_ROBUST_CLAMP : sc.robust_kind;
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      double pH[D * D], pb[D], pchi;
      int pstat;
      prior_linearize<D>(C.variable_kind, sc, rk, st->X, pH, pb, pchi, pstat);
#pragma unroll
      for (int i = 0; i < D * D; ++i) H[i] = H[i] + pH[i];
#pragma unroll
      for (int i = 0; i < D; ++i) b[i] = b[i] + pb[i];
      if (pstat == SRRG2_FACTOR_INLIER) {
        cur.num_inliers++;
        chi_in = chi_in + pchi;
      } else if (pstat == SRRG2_FACTOR_KERNELIZED) {
        cur.num_outliers++;
        chi_out = chi_out + pchi;
      } else {
        cur.num_suppressed++;
      }
      st->ninl[s] = pstat == SRRG2_FACTOR_INLIER ? 1 : 0;
      continue;
    }
    const long long* acc = sums[s];
    const double* sc_    = scaled[s];
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
      for (int c = a; c < D; ++c) {
        double v     = sc_[hidx(a, c)];
        H[a * D + c] = H[a * D + c] + v;
        if (c != a) H[c * D + a] = H[c * D + a] + v;
      }
      b[a] = b[a] + sc_[ACC_B + a];
    }
    int n_in = (int) acc[ACC_N_IN], n_out = (int) acc[ACC_N_OUT];
    int n_c  = (int) acc[ACC_N_CORR];
    cur.num_inliers += n_in;
    cur.num_outliers += n_out;
    cur.num_suppressed += n_c - n_in - n_out;
    chi_in      = chi_in + sc_[ACC_CHI_IN];
    chi_out     = chi_out + sc_[ACC_CHI_OUT];
    st->ninl[s] = n_in;
  }
  cur.num_correspondences = num_correspondences(C, st);
  cur.chi_inliers         = (float) chi_in;
  cur.chi_outliers        = (float) chi_out;
  // (timing, profiling builds: bits 27..29 of the mask select how far the step runs -- 1: without the solve and
  // everything behind it, 2: up to and including the solve, 3: + the stores of H, b, dx, 5: + box-plus, 6: + the finder
  // transforms, 4: everything but the termination criterion and the queue bookkeeping)
#ifdef SRRG2_TIMING_KNOBS
  const int stage = (C.tune >> 27) & 7;
#else
  constexpr int stage = 0;
#endif
  if (stage == 1) {
    st->nstats++;
    st->last_H[0] = H[0] + b[0];
    return;
  }
  int bad                 = dm::solve<D>(H, b, dx);
  cur.solver_status       = bad ? 1 : 0;
  if (stage == 2) {
    st->nstats++;
    st->last_dx[0] = dx[0] + dx[D - 1];
    return;
  }
#pragma unroll
  for (int i = 0; i < D * D; ++i) st->last_H[i] = H[i];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    st->last_b[i]  = b[i];
    st->last_dx[i] = bad ? 0.0 : dx[i];
  }
  if (stage == 3) {
    st->nstats++;
    return;
  }
  for (int i = 0; i < 12; ++i) st->Xprev[i] = st->X[i];  // the estimate this iteration's finder passes ran with
  if (!bad) dm::box_plus(C.variable_kind, st->X, dx);  // solver Success: multi_aligner_impl.cpp:118-121
  if (stage == 5) {
    st->nstats++;
    return;
  }
  for (int s = 0; s < C.nslices; ++s) {
    if (C.slices[s].kind == SRRG2_SLICE_PRIOR) continue;
    for (int i = 0; i < 12; ++i) st->Tfprev[s][i] = st->Tf[s][i];
    if (!bad) finder_transform_of(C.slices[s].Sinv, C.variable_kind == SRRG2_SE2_RIGHT ? 2 : 3, st->X, st->Tf[s]);
  }
  if (stage == 6) {
    st->nstats++;
    return;
  }
  for (int s = 0; s < C.nslices; ++s)  // the fixed-point scale of given-correspondences slices follows the estimate
    if (C.slices[s].finder == SRRG2_FINDER_CORRESPONDENCES && C.slices[s].kind != SRRG2_SLICE_PRIOR)
      st->kexp[s] = slice_exponent(C, C.slices[s], prob, 0, st->X);
  if (st->nstats < C.max_stats) stats[(size_t) prob * C.max_stats + st->nstats] = cur;
  st->nstats++;
  if (stage == 4) return;  // (timing: without the termination criterion and the queue bookkeeping)
  if (C.has_term && has_to_stop(C, st, cur)) st->done = 1;  // :124-126
  for (int s = 0; s < C.nslices; ++s)
    if (C.slices[s].qcount) {
      // how much was deferred this iteration, for the host (pinned memory): it decides after a few iterations whether
      // the deferred-search kernel is still worth its launch
      if (C.slices[s].qprobe_host && st->phase == 0 && st->nstats == C.probe_it + 1) {
        const int near = C.slices[s].qcount[2 * prob], far = C.slices[s].qcount[2 * prob + 1];
        // (the host applies the same rule to the mirrored counters: srrg2amd::queue_stays_small)
        if (far <= 32 && near <= max(1024, C.slices[s].probs[prob].nm / 64)) st->qmode[s] = 0;
        C.slices[s].qprobe_host[2 * prob]     = near;
        C.slices[s].qprobe_host[2 * prob + 1] = far;
      }
      C.slices[s].qcount[2 * prob] = C.slices[s].qcount[2 * prob + 1] = 0;  // queues start empty next iteration
    }
}

}  // namespace

// thread 0 of compute()'s prologue: the state of the problem, the device copy of its problem table and -- fused control steps -- the
// records of epoch 0, staged in `init_gran` for the wave that publishes them (k_icp_init; fused_init_tail inside a first pass)
namespace {
// (ONLY_INLINE: guess and problem table from `inl` whatever inl.use says -- the pinned tables are not even passed)
template <bool ONLY_INLINE>
__device__ __forceinline__ void init_problem_thread0(const CtlParams& C, int prob, const ProblemDev* probs_host, ProblemDev* probs,
                                     ProblemState* states, const float* guesses_host, int tsize, const InitInline& inl,
                                     unsigned (*init_gran)[PUB_SLICE_GRANULES]) {
  ProblemState* st = &states[prob];
  for (int i = 0; i < 12; ++i) st->X[i] = i < tsize ? ((ONLY_INLINE || inl.use) ? inl.guess[i] : guesses_host[(size_t) prob * tsize + i]) : 0.f;
  for (int i = 0; i < 12; ++i) st->Xprev[i] = st->X[i];
  st->status   = SRRG2_FAIL;
  st->done     = 0;
  st->finished = 0;
  st->nstats   = 0;
  st->phase    = 0;
  st->w_count  = 0;
  st->npasses  = 0;
  for (int s = 0; s < SRRG2_MAX_SLICES; ++s) st->qmode[s] = 1;
  int nm_of[SRRG2_MAX_SLICES];
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    nm_of[s]           = 0;
    ProblemDev pd;
    if constexpr (ONLY_INLINE) {
      pd = inl.pd[s];
    } else {
      pd = inl.use ? inl.pd[s] : probs_host[(size_t) s * C.K + prob];
    }
    probs[(size_t) s * C.K + prob] = pd;
    st->ncorr[s]       = 0;
    st->ninl[s]        = 0;
    if (sc.qcount) sc.qcount[2 * prob] = sc.qcount[2 * prob + 1] = 0;
    if (sc.kind == SRRG2_SLICE_PRIOR) {
      st->kexp[s] = 0;
      if (sc.prior_sets_initial_guess) {  // aligner_slice_odometry_prior.cpp:19,34; aligner_slice_motion_model.hpp:69-70
        for (int i = 0; i < tsize; ++i) st->X[i] = sc.prior_Z[i];
      }
    } else {
      nm_of[s] = sc.nm_global > 0 ? sc.nm_global : pd.nm;
    }
  }
  for (int s = 0; s < C.nslices; ++s) {  // (after the loop: a prior slice may have replaced the initial guess)
    if (C.slices[s].kind == SRRG2_SLICE_PRIOR) continue;
    float Xloc[12], Tloc[12];
    for (int i = 0; i < 12; ++i) Xloc[i] = st->X[i];
    const int kx = slice_exponent(C, C.slices[s], prob, nm_of[s], Xloc);
    finder_transform_of(C.slices[s].Sinv, C.variable_kind == SRRG2_SE2_RIGHT ? 2 : 3, Xloc, Tloc);
    st->kexp[s] = kx;
    for (int i = 0; i < 12; ++i) st->Tf[s][i] = st->Tfprev[s][i] = Tloc[i];
    if (C.pub) {  // fused control steps: the record of epoch 0, staged for the wave (from the registers: read back from the
                  // state just stored, 64 dependent loads took the kernel from 7 to 17 us)
      for (int l = 0; l < PUB_SLICE_GRANULES; ++l) init_gran[s][l] = 0u;  // (flags, nstats, w_count, npasses: 0)
      for (int i = 0; i < 12; ++i) {
        init_gran[s][i]           = __float_as_uint(Tloc[i]);
        init_gran[s][12 + i]      = __float_as_uint(Tloc[i]);
        init_gran[s][PUB_G_X + i] = __float_as_uint(Xloc[i]);
      }
      init_gran[s][PUB_G_KEXP] = (unsigned) kx;
    }
  }
}
}  // namespace

// compute() prologue: term_crit->init, stats clear, _preCompute (prior init overrides the guess).  One block per
// problem.  The initial guesses and the problem tables are read straight from pinned host memory (no staging copies on
// the stream), the partial-sum slots and queue counters are zeroed here (no memsets on the stream).
// (inl.use: a single alignment's guess and problem table travel in the kernel arguments -- 116 bytes -- instead of being read
// from pinned host memory: one PCIe round trip less at the head of every compute(); batches keep the pinned tables)
__global__ __launch_bounds__(64) void k_icp_init(CtlParams C, const ProblemDev* __restrict__ probs_host,
                                                 ProblemDev* __restrict__ probs, ProblemState* __restrict__ states,
                                                 const float* __restrict__ guesses_host, int tsize, InitInline inl) {
  const int prob = blockIdx.x + C.prob0;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    if (sc.kind == SRRG2_SLICE_PRIOR || !sc.partials) continue;
    for (int buf = 0; buf < (C.pub ? 3 : 1); ++buf) {  // (buffers 1, 2: fused control steps, round k adds into buffer k % 3)
      long long* p = const_cast<long long*>(sc.partials) + ((size_t) buf * C.K + prob) * PARTIAL_SLOTS * ACC_N;
      for (int k = threadIdx.x; k < PARTIAL_SLOTS * ACC_N; k += blockDim.x) p[k] = 0;
    }
  }
  if (C.ctl_dev && blockIdx.x == 0) {  // the control parameters where the fused control steps find them
    static_assert(sizeof(CtlParams) % sizeof(int) == 0, "copied word-wise");
    const int* src = reinterpret_cast<const int*>(&C);
    int* dst       = reinterpret_cast<int*>(C.ctl_dev);
    for (int k = threadIdx.x; k < (int) (sizeof(CtlParams) / sizeof(int)); k += blockDim.x) dst[k] = src[k];
  }
  // (fused control steps: the records of epoch 0 are written by all 64 lanes from a staged copy of what thread 0 computes;
  // one thread storing 64 granules per slice one after the other took k_icp_init from 6 to 25 us)
  __shared__ unsigned init_gran[SRRG2_MAX_SLICES][PUB_SLICE_GRANULES];
  if (threadIdx.x == 0) init_problem_thread0<false>(C, prob, probs_host, probs, states, guesses_host, tsize, inl, init_gran);
  if (!C.pub) return;
  __syncthreads();
  for (int s = 0; s < C.nslices; ++s)
    if (C.slices[s].kind != SRRG2_SLICE_PRIOR)
      pub_store(C.pub + ((size_t) prob * SRRG2_MAX_SLICES + s) * PUB_SLICE_GRANULES + threadIdx.x, (unsigned long long) init_gran[s][threadIdx.x]);
  pub_write_epoch(C.pub_epoch, prob, threadIdx.x, 0u);
}

constexpr int STATE_WORDS = (int) (sizeof(ProblemState) / sizeof(int));
static_assert(sizeof(ProblemState) % sizeof(int) == 0, "the state is copied word-wise");
__device__ __forceinline__ void state_to_lds(ProblemState* lds, const ProblemState* st) {
  for (int k = threadIdx.x; k < STATE_WORDS; k += blockDim.x)
    reinterpret_cast<int*>(lds)[k] = reinterpret_cast<const int*>(st)[k];
}
__device__ __forceinline__ void state_from_lds(ProblemState* st, const ProblemState* lds) {
  for (int k = threadIdx.x; k < STATE_WORDS; k += blockDim.x)
    reinterpret_cast<int*>(st)[k] = reinterpret_cast<const int*>(lds)[k];
}

// The slot sets of the first two cue slices, every thread's share, loaded at the very top of the control kernels: in flight
// TOGETHER with the state record instead of behind it.  (The step was a chain of three dependent round trips -- the `done`
// word, the state record, the slot sets -- each ~1.5 us on a cold cache behind a kernel boundary: round 4.)
struct PrePartials {
  int s0, s1;  // the slices they belong to (-1: none)
  long long v0, v1;
};
__device__ __forceinline__ PrePartials prefetch_partials(const CtlParams& C, int prob) {
  PrePartials P;
  P.s0 = P.s1 = -1;
  P.v0 = P.v1 = 0;
  for (int s = 0; s < C.nslices; ++s) {  // (uniform: kernel arguments)
    if (C.slices[s].kind == SRRG2_SLICE_PRIOR) continue;
    if (P.s0 < 0) P.s0 = s; else if (P.s1 < 0) P.s1 = s;
  }
  const int a = threadIdx.x & 31, c = threadIdx.x >> 5;
  if (P.s0 >= 0) {
    const long long* p = C.slices[P.s0].partials + ((size_t) C.parity * C.K + prob) * PARTIAL_SLOTS * ACC_N;
#pragma unroll
    for (int q = 0; q < PARTIAL_SLOTS / 8; ++q) P.v0 += p[(size_t) (c + 8 * q) * ACC_N + a];
  }
  if (P.s1 >= 0) {
    const long long* p = C.slices[P.s1].partials + ((size_t) C.parity * C.K + prob) * PARTIAL_SLOTS * ACC_N;
#pragma unroll
    for (int q = 0; q < PARTIAL_SLOTS / 8; ++q) P.v1 += p[(size_t) (c + 8 * q) * ACC_N + a];
  }
  return P;
}

// one 256-thread block per problem: sum the per-block partials of every cue slice (exact integer sums, any
// order), then thread 0 runs the sequential part of the iteration
__device__ void icp_control_block(const CtlParams& C, ProblemState* st, srrg2_iteration_stats* stats, int prob,
                                  const PrePartials& pre) {
  __shared__ long long sums[SRRG2_MAX_SLICES][ACC_N];
  __shared__ double scaled[SRRG2_MAX_SLICES][ACC_N];
  __shared__ long long part[8][ACC_N];
  const int a = threadIdx.x & 31, c = threadIdx.x >> 5;
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    if (sc.kind == SRRG2_SLICE_PRIOR) continue;
    long long* p = const_cast<long long*>(sc.partials) + ((size_t) C.parity * C.K + prob) * PARTIAL_SLOTS * ACC_N;
    long long v  = 0;
    if (s == pre.s0) {
      v = pre.v0;
    } else if (s == pre.s1) {
      v = pre.v1;
    } else {
#pragma unroll
      for (int q = 0; q < PARTIAL_SLOTS / 8; ++q) v += p[(size_t) (c + 8 * q) * ACC_N + a];
    }
    part[c][a] = v;
    __syncthreads();
    // the slot sets are accumulated with atomics by the step kernels: reset them for the next iteration
    // (after the barrier: every thread of the block has finished reading; fused control steps: not the buffer just read but
    // the one the round after next adds into -- FusedCtl::prev_partials --, legacy: zero_parity == parity == 0)
    long long* pz = const_cast<long long*>(sc.partials) + ((size_t) C.zero_parity * C.K + prob) * PARTIAL_SLOTS * ACC_N;
#pragma unroll
    for (int q = 0; q < PARTIAL_SLOTS / 8; ++q) pz[(size_t) (c + 8 * q) * ACC_N + a] = 0;
    if (threadIdx.x < ACC_N) {
      long long t = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
      sums[s][threadIdx.x]   = t;
      scaled[s][threadIdx.x] = (double) t * dm::pow2(-st->kexp[s]);
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if (KNOB(C.tune, 67108864)) {  // (timing: everything but the sequential part; the iteration counter must still advance)
    st->nstats++;
    return;
  }
  if (C.variable_kind == SRRG2_SE2_RIGHT)
    control_body<3>(C, st, stats, prob, sums, scaled);
  else
    control_body<6>(C, st, stats, prob, sums, scaled);
}

__global__ __launch_bounds__(256) void k_icp_control(CtlParams C, ProblemState* __restrict__ states,
                                                     srrg2_iteration_stats* __restrict__ stats) {
  const int prob   = blockIdx.x + C.prob0;
  ProblemState* st = &states[prob];
  // The sequential part reads and writes ~100 words of the state, with stores to other arrays in between that the
  // compiler must assume to alias: in global memory that is a chain of exposed L2 round trips.  The workgroup stages the
  // record in LDS (one coalesced load), thread 0 works there, the workgroup writes it back.  The slot sets are requested
  // before the record and `done` is read from the staged copy: one round trip for all three.
  const PrePartials pre = prefetch_partials(C, prob);
  __shared__ ProblemState sst;
  state_to_lds(&sst, st);
  __syncthreads();
  if (sst.done || sst.finished) {  // (uniform: LDS; nothing was modified)
    if (C.pub && threadIdx.x < 64) {  // (fused control steps: the records still move to this step's epoch)
      pub_publish_state(C, &sst, prob, threadIdx.x, (unsigned) C.epoch);
      pub_write_epoch(C.pub_epoch, prob, threadIdx.x, (unsigned) C.epoch);
    }
    return;
  }
  icp_control_block(C, &sst, stats, prob, pre);
  __syncthreads();
  state_from_lds(st, &sst);
  if (C.pub && threadIdx.x < 64) {
    pub_publish_state(C, &sst, prob, threadIdx.x, (unsigned) C.epoch);
    pub_write_epoch(C.pub_epoch, prob, threadIdx.x, (unsigned) C.epoch);
  }
}

// after the main _runSolver: multi_aligner_impl.cpp:75-85 and the start of _postCompute (:165-171)
__device__ void icp_post_one(const CtlParams& C, ProblemState* st, const srrg2_iteration_stats* stats, int prob) {
  if (st->nstats == 0) {
    st->status   = SRRG2_FAIL;
    st->finished = 1;
    return;
  }
  int last = st->nstats < C.max_stats ? st->nstats - 1 : C.max_stats - 1;
  if (stats[(size_t) prob * C.max_stats + last].num_inliers < C.params.min_num_inliers) {
    st->status   = SRRG2_NOT_ENOUGH_INLIERS;
    st->finished = 1;
    return;
  }
  if (C.params.enable_inlier_only_runs) {
    st->phase = 1;  // _setClampRobustifiers
    st->done  = 0;
  }
}

// (one wave per problem)
__global__ __launch_bounds__(64) void k_icp_post(CtlParams C, ProblemState* __restrict__ states,
                                                 const srrg2_iteration_stats* __restrict__ stats) {
  const int prob = blockIdx.x + C.prob0;
  if (threadIdx.x == 0) icp_post_one(C, &states[prob], stats, prob);
  if (!C.pub) return;
  // fused control steps.  The slot sets: a run that stopped early (termination criterion, too few correspondences) left the
  // sums of its last round in the buffer that round added into -- the control steps zero the buffer of the round after next,
  // not the one they read (FusedCtl::prev_partials) -- and the inlier-only run starts on zeroed buffers like the first one.
  for (int s = 0; s < C.nslices; ++s) {
    const SliceCtl& sc = C.slices[s];
    if (sc.kind == SRRG2_SLICE_PRIOR || !sc.partials) continue;
    for (int buf = 0; buf < 3; ++buf) {
      long long* p = const_cast<long long*>(sc.partials) + ((size_t) buf * C.K + prob) * PARTIAL_SLOTS * ACC_N;
      for (int k = threadIdx.x; k < PARTIAL_SLOTS * ACC_N; k += 64) p[k] = 0;
    }
  }
  __syncthreads();  // (phase / done changed: thread 0's stores to the state, read back by the wave)
  pub_publish_state(C, &states[prob], prob, threadIdx.x, (unsigned) C.epoch);
  pub_write_epoch(C.pub_epoch, prob, threadIdx.x, (unsigned) C.epoch);
}

// end of compute(): _pruneCorrespondences bookkeeping, fixTransform, Success (:88-94).  One block per problem; the
// results (ProblemOut + the IterationStats appended by this compute()) are written straight into pinned host memory:
// the host only waits for the stream, there are no device-to-host copies.
__device__ void icp_finalize_block(const CtlParams& C, ProblemState* st, const srrg2_iteration_stats* stats,
                                   ProblemOut* outs_host, srrg2_iteration_stats* stats_host, int prob, bool with_post) {
  if (with_post) {  // without an inlier-only run nothing lies between the two steps: one launch
    if (threadIdx.x == 0) icp_post_one(C, st, stats, prob);
    __syncthreads();
  }
  {  // the iteration statistics, 8 words each
    const int n       = min(st->nstats, C.max_stats) * (int) (sizeof(srrg2_iteration_stats) / sizeof(int));
    const int* src    = reinterpret_cast<const int*>(stats + (size_t) prob * C.max_stats);
    int* dst          = reinterpret_cast<int*>(stats_host + (size_t) prob * C.max_stats);
    for (int k = threadIdx.x; k < n; k += blockDim.x) dst[k] = src[k];
  }
  __threadfence_system();  // (the statistics above are in host memory before the completion flag below)
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (!st->finished) {
    if (C.params.keep_only_inlier_correspondences) {
      for (int s = 0; s < C.nslices; ++s)
        if (C.slices[s].kind != SRRG2_SLICE_PRIOR) st->ncorr[s] = st->ninl[s];
    }
    if (C.variable_kind == SRRG2_SE2_RIGHT)
      dm::se2_fix_transform(st->X);
    else
      dm::se3_fix_transform(st->X);
    st->status = SRRG2_SUCCESS;
  }
  ProblemOut* o = &outs_host[prob];
  for (int i = 0; i < 12; ++i) o->X[i] = st->X[i];
  o->status = st->status;
  o->nstats = st->nstats;
  for (int s = 0; s < SRRG2_MAX_SLICES; ++s) o->ncorr[s] = st->ncorr[s];
  {
    const int DD = C.variable_kind == SRRG2_SE2_RIGHT ? 9 : 36;
    for (int i = 0; i < 36; ++i) o->H[i] = (i < DD && st->nstats > 0) ? (float) st->last_H[i] : 0.f;
  }
  __threadfence_system();
  *reinterpret_cast<volatile int*>(&o->seq) = C.seq;  // the host polls this word instead of waiting for the stream
}

__global__ __launch_bounds__(64) void k_icp_finalize(CtlParams C, ProblemState* __restrict__ states,
                                                     const srrg2_iteration_stats* __restrict__ stats,
                                                     ProblemOut* __restrict__ outs_host,
                                                     srrg2_iteration_stats* __restrict__ stats_host, int with_post) {
  icp_finalize_block(C, &states[blockIdx.x + C.prob0], stats, outs_host, stats_host, blockIdx.x + C.prob0, with_post != 0);
}

// The control step of the LAST iteration of compute() with the (post and) finalize steps behind it: nothing lies
// between them, two launches less per compute().
__global__ __launch_bounds__(256) void k_icp_control_final(CtlParams C, ProblemState* __restrict__ states,
                                                           srrg2_iteration_stats* __restrict__ stats,
                                                           ProblemOut* __restrict__ outs_host,
                                                           srrg2_iteration_stats* __restrict__ stats_host, int with_post) {
  const int prob = blockIdx.x + C.prob0;
  const PrePartials pre = prefetch_partials(C, prob);  // (see k_icp_control)
  __shared__ ProblemState sst;
  state_to_lds(&sst, &states[prob]);
  __syncthreads();
  if (!sst.done && !sst.finished) icp_control_block(C, &sst, stats, prob, pre);  // (uniform: LDS)
  __threadfence();  // (thread 0's statistics, read by the whole block below)
  __syncthreads();
  icp_finalize_block(C, &sst, stats, outs_host, stats_host, prob, with_post != 0);
  __syncthreads();
  state_from_lds(&states[prob], &sst);
}

// ============================================================================================
// Small alignments (laser scans, landmark maps: up to ~1000 moving points).  With one launch per pass such a compute()
// is ~24 launch floors and little else (C1: 0.33 ms on the GPU against 0.66 ms on one CPU core; 0.23 ms here).  ONE
// workgroup owns the problem for the whole compute(): the state and the sums live in LDS, the passes loop over the
// problem's tiles, thread 0 runs the control step between them -- the same per-point and control code as the
// launch-per-pass path, so the same bits.  A batch of small problems runs one workgroup per problem.
// ============================================================================================
template <int DIM, bool PLANE>
__global__ __launch_bounds__(512) void k_icp_small(SliceDev S, CtlParams C, const ProblemDev* __restrict__ probs,
                                                   ProblemState* __restrict__ states, srrg2_iteration_stats* __restrict__ stats,
                                                   ProblemOut* __restrict__ outs_host,
                                                   srrg2_iteration_stats* __restrict__ stats_host) {
  constexpr int NW = 8;  // (16 waves would cap the registers at 128 per lane: the SE(3) control step spills, measured slower)
  constexpr int D  = DIM == 3 ? 6 : 3;
  const int prob   = blockIdx.x;
  __shared__ ProblemState sst;
  __shared__ long long sums[SRRG2_MAX_SLICES][ACC_N];
  __shared__ double scaled[SRRG2_MAX_SLICES][ACC_N];
  state_to_lds(&sst, &states[prob]);
  __syncthreads();
  const ProblemDev pd = probs[prob];
  const int ntiles    = (pd.nm + NW * 64 - 1) / (NW * 64);
  const int nrun      = C.params.enable_inlier_only_runs ? 2 : 1;
  for (int run = 0; run < nrun; ++run) {
    for (int it = 0; it < C.params.max_iterations; ++it) {
      if (sst.done || sst.finished) break;  // (uniform: LDS)
      if (threadIdx.x < ACC_N) sums[S.slice_idx][threadIdx.x] = 0;
      __syncthreads();
      for (int tile = 0; tile < ntiles; ++tile) {
        StepView sv;
        step_view_of_state(S, &sst, sv);
        icp_step_body<DIM, PLANE, NW>(S, pd, sv, prob, tile, ntiles, C.K, sums[S.slice_idx]);
        __syncthreads();  // (the body's shared scratch is reused by the next tile)
      }
      if (threadIdx.x < ACC_N)
        scaled[S.slice_idx][threadIdx.x] = (double) sums[S.slice_idx][threadIdx.x] * dm::pow2(-sst.kexp[S.slice_idx]);
      __syncthreads();
      if (threadIdx.x == 0) control_body<D>(C, &sst, stats, prob, sums, scaled);
      __syncthreads();
    }
    if (run == 0 && nrun == 2) {  // multi_aligner_impl.cpp:75-85, then the inlier-only run
      if (threadIdx.x == 0) icp_post_one(C, &sst, stats, prob);
      __syncthreads();
    }
  }
  __threadfence();  // (the statistics thread 0 wrote are read back by the whole workgroup below)
  __syncthreads();
  icp_finalize_block(C, &sst, stats, outs_host, stats_host, prob, nrun == 1);
  __syncthreads();
  state_from_lds(&states[prob], &sst);
}

// ============================================================================================
// launchers
// ============================================================================================
namespace srrg2amd {

int icp_step_blocks(int max_nm) {
  return (max_nm + 255) / 256;  // one moving point per thread
}

int icp_queue_blocks(int max_nm, int K) {
  // waves that share the deferred searches of one problem: up to 4096 for a single alignment, fewer per problem in a batch
  int b = icp_step_blocks(max_nm);
  int cap = K <= 1 ? 1024 : (K <= 8 ? 256 : 64);
  return b < cap ? b : cap;
}

static void launch_icp_queue(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                             int max_nm, hipStream_t s) {
  const int qb = icp_queue_blocks(max_nm, K);
  dim3 qgrid(qb, K);
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_queue<3, true>), qgrid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_queue<3, false>), qgrid, dim3(256), 0, s, S, probs, states);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_queue<2, true>), qgrid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_queue<2, false>), qgrid, dim3(256), 0, s, S, probs, states);
  }
}

void launch_icp_step(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                     int max_nm, hipStream_t s, const CtlParams* init_C, const InitInline* init_inl) {
  if (K <= 0) return;
  if (S.fc.pub && init_C) {  // the first pass of a single alignment with compute()'s prologue inside (even for an empty cloud)
    dim3 fgrid(K, (max_nm + 255) / 256 + 1);  // (+ the workgroup that writes the rest of the prologue)
    ProblemDev* pw = const_cast<ProblemDev*>(probs);
    if (dim == 3) {
      if (plane)
        hipLaunchKernelGGL((k_icp_step_fused_init<3, true>), fgrid, dim3(256), 0, s, S, *init_C, *init_inl, pw, states);
      else
        hipLaunchKernelGGL((k_icp_step_fused_init<3, false>), fgrid, dim3(256), 0, s, S, *init_C, *init_inl, pw, states);
    } else {
      if (plane)
        hipLaunchKernelGGL((k_icp_step_fused_init<2, true>), fgrid, dim3(256), 0, s, S, *init_C, *init_inl, pw, states);
      else
        hipLaunchKernelGGL((k_icp_step_fused_init<2, false>), fgrid, dim3(256), 0, s, S, *init_C, *init_inl, pw, states);
    }
    return;
  }
  if (max_nm <= 0) return;
  if (S.fc.pub) {  // fused control steps: the record instead of ProblemState, no deferred-search queue
    dim3 fgrid(K, (max_nm + 255) / 256);
#define FUSED_GRID_LAUNCH(PRIORS)                                                                              \
  do {                                                                                                         \
    if (dim == 3) {                                                                                            \
      if (plane)                                                                                               \
        hipLaunchKernelGGL((k_icp_step_fused<3, true, PRIORS>), fgrid, dim3(256), 0, s, S, probs, states);     \
      else                                                                                                     \
        hipLaunchKernelGGL((k_icp_step_fused<3, false, PRIORS>), fgrid, dim3(256), 0, s, S, probs, states);    \
    } else {                                                                                                   \
      if (plane)                                                                                               \
        hipLaunchKernelGGL((k_icp_step_fused<2, true, PRIORS>), fgrid, dim3(256), 0, s, S, probs, states);     \
      else                                                                                                     \
        hipLaunchKernelGGL((k_icp_step_fused<2, false, PRIORS>), fgrid, dim3(256), 0, s, S, probs, states);    \
    }                                                                                                          \
  } while (0)
    if (S.fc.prior_mask)
      FUSED_GRID_LAUNCH(true);
    else
      FUSED_GRID_LAUNCH(false);
#undef FUSED_GRID_LAUNCH
    return;
  }
  int bx = (max_nm + 255) / 256;  // one moving point per thread
  dim3 grid(bx, K);
  // (experiment: SRRG2_AMD_STEP_LDS = bytes of unused dynamic LDS per workgroup, to lower the occupancy on purpose)
  static const unsigned dyn = getenv("SRRG2_AMD_STEP_LDS") ? (unsigned) atoi(getenv("SRRG2_AMD_STEP_LDS")) : 0u;
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_step<3, true>), grid, dim3(256), dyn, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step<3, false>), grid, dim3(256), dyn, s, S, probs, states);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_step<2, true>), grid, dim3(256), dyn, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step<2, false>), grid, dim3(256), dyn, s, S, probs, states);
  }
  if (S.queue) launch_icp_queue(dim, plane, S, probs, states, K, max_nm, s);
}

#ifdef SRRG2_PASS_TIMELINE
extern "C" int srrg2_amd_debug_pass_timeline(unsigned long long* out, int reset) {
  const size_t bytes = sizeof(unsigned long long) * 16 * 512 * 8;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pass_ts), bytes) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long zero[16 * 512 * 8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pass_ts), zero, bytes) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#ifdef SRRG2_CNL_STATS
extern "C" int srrg2_amd_debug_cnl_stats(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cnl_stats), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long zero[64];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_cnl_stats), zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#ifdef SRRG2_TILE_STATS
extern "C" int srrg2_amd_debug_tile_stats(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stats), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long zero[64];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stats), zero, sizeof(zero)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

// the search pass of a batch with wave tiles in LDS (cap: candidates per wave tile, 416: four workgroups per CU, 504: three)
void launch_icp_step_tile(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                          int max_nm, int cap, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  dim3 grid((max_nm + 255) / 256, K);
#define TILE_LAUNCH(D, P)                                                                            \
  do {                                                                                               \
    if (cap > 450)                                                                                   \
      hipLaunchKernelGGL((k_icp_step_tile<D, P, 504>), grid, dim3(256), 0, s, S, probs, states);      \
    else                                                                                             \
      hipLaunchKernelGGL((k_icp_step_tile<D, P, 416>), grid, dim3(256), 0, s, S, probs, states);      \
  } while (0)
  if (dim == 3) {
    if (plane) TILE_LAUNCH(3, true); else TILE_LAUNCH(3, false);
  } else {
    if (plane) TILE_LAUNCH(2, true); else TILE_LAUNCH(2, false);
  }
#undef TILE_LAUNCH
}

// the search pass over the cell neighbour lists of the grid (S.grid.list_R > 0); team = lanes per moving point (1 or 4)
void launch_icp_step_cnl(int dim, bool plane, const SliceDev& S, const GridLists& GL, const ProblemDev* probs,
                         ProblemState* states, int K, int max_nm, int team, hipStream_t s, const CtlParams* init_C,
                         const InitInline* init_inl, const InitBatch* init_bat) {
  if (K <= 0) return;
  if (S.fc.pub && init_C && init_bat) {  // a part of a batch (K <= INIT_BATCH_MAX alignments) with the prologue in its first pass
    ProblemDev* pw = const_cast<ProblemDev*>(probs);
#define CNL_INITB_LAUNCH(TEAM)                                                                                                        \
  do {                                                                                                                                \
    dim3 grid(K, (max_nm * TEAM + 255) / 256 + 1);                                                                                    \
    if (dim == 3) {                                                                                                                   \
      if (plane)                                                                                                                      \
        hipLaunchKernelGGL((k_icp_step_cnl_init_batch<3, true, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_bat, pw, states);  \
      else                                                                                                                            \
        hipLaunchKernelGGL((k_icp_step_cnl_init_batch<3, false, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_bat, pw, states); \
    } else {                                                                                                                          \
      if (plane)                                                                                                                      \
        hipLaunchKernelGGL((k_icp_step_cnl_init_batch<2, true, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_bat, pw, states);  \
      else                                                                                                                            \
        hipLaunchKernelGGL((k_icp_step_cnl_init_batch<2, false, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_bat, pw, states); \
    }                                                                                                                                 \
  } while (0)
    if (team >= 4)
      CNL_INITB_LAUNCH(4);
    else
      CNL_INITB_LAUNCH(1);
#undef CNL_INITB_LAUNCH
    return;
  }
  if (S.fc.pub && init_C) {  // the first pass of a single alignment with compute()'s prologue inside (even for an empty cloud)
    ProblemDev* pw = const_cast<ProblemDev*>(probs);
#define CNL_INIT_LAUNCH(TEAM)                                                                                                   \
  do {                                                                                                                          \
    dim3 grid(K, (max_nm * TEAM + 255) / 256 + 1); /* (+ the workgroup that writes the rest of the prologue) */                  \
    if (dim == 3) {                                                                                                             \
      if (plane)                                                                                                                \
        hipLaunchKernelGGL((k_icp_step_cnl_init<3, true, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_inl, pw, states);  \
      else                                                                                                                      \
        hipLaunchKernelGGL((k_icp_step_cnl_init<3, false, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_inl, pw, states); \
    } else {                                                                                                                    \
      if (plane)                                                                                                                \
        hipLaunchKernelGGL((k_icp_step_cnl_init<2, true, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_inl, pw, states);  \
      else                                                                                                                      \
        hipLaunchKernelGGL((k_icp_step_cnl_init<2, false, TEAM>), grid, dim3(256), 0, s, S, GL, *init_C, *init_inl, pw, states); \
    }                                                                                                                           \
  } while (0)
    if (team >= 4)
      CNL_INIT_LAUNCH(4);
    else
      CNL_INIT_LAUNCH(1);
#undef CNL_INIT_LAUNCH
    return;
  }
  if (max_nm <= 0) return;
#define CNL_LAUNCH(TEAM, FUSED)                                                                                   \
  do {                                                                                                            \
    dim3 grid((max_nm * TEAM + 255) / 256, K);                                                                    \
    if (FUSED) grid = dim3(K, (max_nm * TEAM + 255) / 256); /* x = problem, y = tile */                            \
    if (dim == 3) {                                                                                               \
      if (plane)                                                                                                  \
        hipLaunchKernelGGL((k_icp_step_cnl<3, true, TEAM, FUSED>), grid, dim3(256), 0, s, S, GL, probs, states);  \
      else                                                                                                        \
        hipLaunchKernelGGL((k_icp_step_cnl<3, false, TEAM, FUSED>), grid, dim3(256), 0, s, S, GL, probs, states); \
    } else {                                                                                                      \
      if (plane)                                                                                                  \
        hipLaunchKernelGGL((k_icp_step_cnl<2, true, TEAM, FUSED>), grid, dim3(256), 0, s, S, GL, probs, states);  \
      else                                                                                                        \
        hipLaunchKernelGGL((k_icp_step_cnl<2, false, TEAM, FUSED>), grid, dim3(256), 0, s, S, GL, probs, states); \
    }                                                                                                             \
  } while (0)
  // (fused control steps -- S.fc.pub: the instantiations that read the state from the published record)
  if (S.fc.pub && S.fc.prior_mask) {  // (... of an aligner with prior slices: the control wave linearises them, wave_prior)
    if (team >= 4)
      CNL_LAUNCH(4, 2);
    else
      CNL_LAUNCH(1, 2);
  } else if (S.fc.pub) {
    if (team >= 4)
      CNL_LAUNCH(4, 1);
    else
      CNL_LAUNCH(1, 1);
  } else {
    if (team >= 4)
      CNL_LAUNCH(4, 0);
    else
      CNL_LAUNCH(1, 0);
  }
#undef CNL_LAUNCH
}
template <int PPT, bool GATHER, int FUSED>
static void launch_fast_ppt(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                            int max_nm, hipStream_t s) {
  dim3 grid((max_nm + 256 * PPT - 1) / (256 * PPT), K);
  if (FUSED) grid = dim3(K, (max_nm + 256 * PPT - 1) / (256 * PPT));  // (x = problem, y = tile)
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_fast<3, true, PPT, GATHER, FUSED>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_fast<3, false, PPT, GATHER, FUSED>), grid, dim3(256), 0, s, S, probs, states);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_fast<2, true, PPT, GATHER, FUSED>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_fast<2, false, PPT, GATHER, FUSED>), grid, dim3(256), 0, s, S, probs, states);
  }
}
void launch_icp_step_fast(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                          int max_nm, int ppt, bool gather, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  if (S.fc.pub && S.fc.prior_mask) {  // ... of an aligner with prior slices (one point per thread: the instantiations that exist)
    if (gather)
      launch_fast_ppt<1, true, 2>(dim, plane, S, probs, states, K, max_nm, s);
    else
      launch_fast_ppt<1, false, 2>(dim, plane, S, probs, states, K, max_nm, s);
    return;
  }
  if (S.fc.pub) {  // fused control steps (one or two points per thread)
    if (gather) {
      if (ppt >= 2)
        launch_fast_ppt<2, true, 1>(dim, plane, S, probs, states, K, max_nm, s);
      else
        launch_fast_ppt<1, true, 1>(dim, plane, S, probs, states, K, max_nm, s);
    } else {
      if (ppt >= 2)
        launch_fast_ppt<2, false, 1>(dim, plane, S, probs, states, K, max_nm, s);
      else
        launch_fast_ppt<1, false, 1>(dim, plane, S, probs, states, K, max_nm, s);
    }
    return;
  }
  if (gather) {
    if (ppt >= 4)
      launch_fast_ppt<4, true, 0>(dim, plane, S, probs, states, K, max_nm, s);
    else if (ppt >= 2)
      launch_fast_ppt<2, true, 0>(dim, plane, S, probs, states, K, max_nm, s);
    else
      launch_fast_ppt<1, true, 0>(dim, plane, S, probs, states, K, max_nm, s);
  } else {
    if (ppt >= 4)
      launch_fast_ppt<4, false, 0>(dim, plane, S, probs, states, K, max_nm, s);
    else if (ppt >= 2)
      launch_fast_ppt<2, false, 0>(dim, plane, S, probs, states, K, max_nm, s);
    else
      launch_fast_ppt<1, false, 0>(dim, plane, S, probs, states, K, max_nm, s);
  }
  if (S.queue) launch_icp_queue(dim, plane, S, probs, states, K, max_nm, s);
}

void launch_icp_outputs(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, const ProblemState* states, int K,
                        int max_nm, hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  dim3 grid((max_nm + 255) / 256, K);
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_outputs<3, true>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_outputs<3, false>), grid, dim3(256), 0, s, S, probs, states);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_outputs<2, true>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_outputs<2, false>), grid, dim3(256), 0, s, S, probs, states);
  }
}

void launch_corr_step(int dim, bool plane, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K,
                      int max_ncorr, hipStream_t s) {
  if (K <= 0 || max_ncorr <= 0) return;
  dim3 grid((max_ncorr + 255) / 256, K);
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_corr<3, true>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_corr<3, false>), grid, dim3(256), 0, s, S, probs, states);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_step_corr<2, true>), grid, dim3(256), 0, s, S, probs, states);
    else
      hipLaunchKernelGGL((k_icp_step_corr<2, false>), grid, dim3(256), 0, s, S, probs, states);
  }
}

void launch_proj_step(bool repro, const SliceDev& S, const ProblemDev* probs, ProblemState* states, int K, int max_nm,
                      hipStream_t s) {
  if (K <= 0 || max_nm <= 0) return;
  dim3 grid(icp_step_blocks(max_nm), K);
  // (no memset: the z-buffers ping-pong, k_icp_step_proj resets the one the next pass will use)
  hipLaunchKernelGGL(k_proj_zbuf, grid, dim3(256), 0, s, S, probs, states);
  if (repro)
    hipLaunchKernelGGL((k_icp_step_proj<true>), grid, dim3(256), 0, s, S, probs, states);
  else
    hipLaunchKernelGGL((k_icp_step_proj<false>), grid, dim3(256), 0, s, S, probs, states);
}

void launch_proj_step_pack(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states, int K,
                           int max_nm, hipStream_t s) {
  if (K <= 0 || max_nm <= 0 || nslices <= 0 || nslices > 4) return;
  SlicePack P;
  for (int z = 0; z < nslices; ++z) {
    P.s[z]     = slices[z];
    P.probs[z] = probs[z];
  }
  for (int z = nslices; z < 4; ++z) {
    P.s[z]     = slices[0];
    P.probs[z] = probs[0];
  }
  dim3 grid(icp_step_blocks(max_nm), K, nslices);
  hipLaunchKernelGGL(k_proj_zbuf_pack, grid, dim3(256), 0, s, P, states);
  hipLaunchKernelGGL(k_icp_step_proj_pack, grid, dim3(256), 0, s, P, states);
}

// the projective slices of one aligner that share clouds and finder parameters: ONE z-buffer pass, ONE step launch
void launch_proj_step_fused(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states, int K,
                            int max_nm, hipStream_t s, const CtlParams* init_C, const InitInline* init_inl, ProblemDev* probs_base) {
  if (K <= 0 || max_nm <= 0 || nslices <= 0 || nslices > 4) return;
  SlicePack P;
  for (int z = 0; z < 4; ++z) {
    P.s[z]     = slices[z < nslices ? z : 0];
    P.probs[z] = probs[z < nslices ? z : 0];
  }
  dim3 grid(icp_step_blocks(max_nm), K);
  if (P.s[0].fc.pub && init_C && init_inl && probs_base) {  // the first iteration of a compute() with the prologue inside
    hipLaunchKernelGGL(k_proj_zbuf_fz_init, dim3(grid.x + 1, K), dim3(256), 0, s, P.s[0], *init_C, *init_inl, probs_base, states);
    hipLaunchKernelGGL(k_icp_step_proj_fused<true>, grid, dim3(256), 0, s, P, nslices, states);
    return;
  }
  if (P.s[0].fc.pub) {  // fused control steps
    if (P.s[0].fc.prior_mask)
      hipLaunchKernelGGL(k_proj_zbuf_fz<true>, grid, dim3(256), 0, s, P, nslices, states);
    else
      hipLaunchKernelGGL(k_proj_zbuf_fz<false>, grid, dim3(256), 0, s, P, nslices, states);
    hipLaunchKernelGGL(k_icp_step_proj_fused<true>, grid, dim3(256), 0, s, P, nslices, states);
    return;
  }
  hipLaunchKernelGGL(k_proj_zbuf, grid, dim3(256), 0, s, P.s[0], P.probs[0], states);
  hipLaunchKernelGGL(k_icp_step_proj_fused<false>, grid, dim3(256), 0, s, P, nslices, states);
}

// the correspondence records of projective slice S after a compute(): the z-buffer of the last executed pass is rebuilt in
// buffer 0 of the slice that owns the association (S0; once per owner: `rebuild`), then read by k_proj_records
void launch_proj_records(const SliceDev& S0, const SliceDev& S, const ProblemDev* probs0, const ProblemDev* probs,
                         ProblemState* states, int K, int max_nm, bool rebuild, hipStream_t s) {
  if (K <= 0 || max_nm <= 0 || !S0.zbuf) return;
  dim3 grid(icp_step_blocks(max_nm), K);
  if (rebuild) {
    (void) hipMemsetAsync(S0.zbuf, 0xff, (size_t) K * S0.rows * S0.cols * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_proj_zbuf_last, grid, dim3(256), 0, s, S0, probs0, states);
  }
  hipLaunchKernelGGL(k_proj_records, grid, dim3(256), 0, s, S0, S, probs, (const ProblemState*) states);
}

bool make_init_inline(const CtlParams& C, const ProblemDev* probs_host, const float* guesses_host, int tsize, InitInline* inl) {
  *inl = InitInline{};
  if (C.K != 1) return false;
  inl->use = 1;  // (read on the host, sent with the launch)
  for (int i = 0; i < 12; ++i) inl->guess[i] = i < tsize ? guesses_host[i] : 0.f;
  for (int sl = 0; sl < C.nslices && sl < SRRG2_MAX_SLICES; ++sl) inl->pd[sl] = probs_host[sl];
  return true;
}
void launch_icp_init(const CtlParams& C, const ProblemDev* probs_host, ProblemDev* probs, ProblemState* states,
                     const float* guesses_host, int tsize, hipStream_t s) {
  InitInline inl{};
  (void) make_init_inline(C, probs_host, guesses_host, tsize, &inl);
  hipLaunchKernelGGL(k_icp_init, dim3(C.nprob > 0 ? C.nprob : C.K), dim3(64), 0, s, C, probs_host, probs, states, guesses_host,
                     tsize, inl);
}
void launch_icp_control(const CtlParams& C, ProblemState* states, srrg2_iteration_stats* stats, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_control, dim3(C.nprob > 0 ? C.nprob : C.K), dim3(256), 0, s, C, states, stats);
}
void launch_icp_small(int dim, bool plane, const SliceDev& S, const CtlParams& C, const ProblemDev* probs, ProblemState* states,
                      srrg2_iteration_stats* stats, ProblemOut* outs_host, srrg2_iteration_stats* stats_host, hipStream_t s) {
  if (C.K <= 0) return;
  if (dim == 3) {
    if (plane)
      hipLaunchKernelGGL((k_icp_small<3, true>), dim3(C.K), dim3(512), 0, s, S, C, probs, states, stats, outs_host, stats_host);
    else
      hipLaunchKernelGGL((k_icp_small<3, false>), dim3(C.K), dim3(512), 0, s, S, C, probs, states, stats, outs_host, stats_host);
  } else {
    if (plane)
      hipLaunchKernelGGL((k_icp_small<2, true>), dim3(C.K), dim3(512), 0, s, S, C, probs, states, stats, outs_host, stats_host);
    else
      hipLaunchKernelGGL((k_icp_small<2, false>), dim3(C.K), dim3(512), 0, s, S, C, probs, states, stats, outs_host, stats_host);
  }
}
// The LAST control step of a compute() with fused control steps, on one wave (wave_control) with the post / finalize steps
// behind it: the 256-thread k_icp_control_final stages the 3.4 KB state through LDS around a 238-register body; this one reads
// the record + the slot sets, runs the lane-distributed step and finalizes from what that step left in its registers
// (FinalRegs); only a run that had stopped earlier lets icp_finalize_block read the state back (its stores are complete
// behind the fence; this kernel has not loaded those lines before, so no stale copy can be hit).
// (MAXS cue slices: one nearest-neighbour slice, or a pack of projective slices that share one association)
template <int D, int MAXS, bool PRIORS>
__device__ __forceinline__ void final_wave_body(const CtlParams& C, const SliceDev* __restrict__ Sv, int ns,
                                                ProblemState* __restrict__ states, srrg2_iteration_stats* __restrict__ stats,
                                                ProblemOut* __restrict__ outs_host, srrg2_iteration_stats* __restrict__ stats_host,
                                                int with_post) {
  const FusedCtl& F = Sv[0].fc;
  const int prob = blockIdx.x + C.prob0;
  const int lane = threadIdx.x & 63;
  ProblemState* st = &states[prob];
  unsigned long long g[MAXS];
  bool stale = false;
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    g[z] = 0ull;
    if (z < ns) {
      g[z]  = pub_load(F.pub + ((size_t) prob * SRRG2_MAX_SLICES + Sv[z].slice_idx) * PUB_SLICE_GRANULES + lane);
      stale = stale || (unsigned) (g[z] >> 32) != (unsigned) F.epoch;
    }
  }
  // (what the fast path below needs of the state and does not find in the step's registers: the correspondence counts of the
  // slices the step does not write -- requested with the records)
  int ncorr_l = 0;
  if (lane < SRRG2_MAX_SLICES) ncorr_l = st->ncorr[lane];
  FinalRegs fin;
  fin.applied = false;
  if (__any(stale)) wave_control<D, MAXS, true, PRIORS, true>(Sv, ns, states, prob, g, nullptr, &fin);
  // The slot sets of the slices, all three buffers, are left ZEROED for the handle's next compute(): its first pass may then
  // carry the prologue (k_icp_step_cnl_init / k_icp_step_fused_init) instead of following a k_icp_init launch that zeroes them.
  // (this step was the last reader; a run that stopped early leaves sums in the buffers its control steps did not get to)
#pragma unroll
  for (int z = 0; z < MAXS; ++z) {
    if (z >= ns) break;
    if (const long long* base = C.slices[Sv[z].slice_idx].partials) {
#pragma unroll 1
      for (int buf = 0; buf < 3; ++buf) {
        long long* p = const_cast<long long*>(base) + ((size_t) buf * C.K + prob) * PARTIAL_SLOTS * ACC_N;
#pragma unroll
        for (int q = 0; q < PARTIAL_SLOTS * ACC_N / 64; ++q) p[q * 64 + lane] = 0;
      }
    }
  }
  if (!fin.applied || !fin.stats_written) {  // (uniform.  A run that had stopped before, or stops here: from the state, as before)
    __threadfence();
    icp_finalize_block(C, st, stats, outs_host, stats_host, prob, with_post != 0);
    return;
  }
  // ---- the step ran in full: post (multi_aligner_impl.cpp:75-85) and finalize (:88-94, :214-263) from its registers, the record
  //      for the host by the lanes of the wave (one 4-byte word each), one system-scope fence in front of the completion flag
  int status    = SRRG2_SUCCESS;
  bool finished = false;
  if (with_post && fin.num_in < C.params.min_num_inliers) {  // icp_post_one: nstats >= 1 here, the last record is this step's
    status   = SRRG2_NOT_ENOUGH_INLIERS;
    finished = true;
  }
  float Xa[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Xa[i] = rl_f(fin.Xl, i);
  const bool keep_inl = !finished && C.params.keep_only_inlier_correspondences;
  if (!finished) {
    if constexpr (D == 3)
      dm::se2_fix_transform(Xa);
    else
      dm::se3_fix_transform(Xa);
  }
  float xl = Xa[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) xl = lane == i ? Xa[i] : xl;
  // the correspondence counts of the cue slices, by slice index (lane = slice): the others keep what the state holds
  int ncorr_s = ncorr_l;
#pragma unroll
  for (int z = 0; z < MAXS; ++z)
    if (z < ns && lane == Sv[z].slice_idx) ncorr_s = keep_inl ? fin.ni[z] : fin.nc[z];
  // the state, for what reads it after compute() (k_icp_outputs, get_information, the next compute() of the handle)
  if (lane < 12) st->X[lane] = xl;
  if (lane < SRRG2_MAX_SLICES) st->ncorr[lane] = ncorr_s;
  if (lane == 0) {
    st->status = status;
    if (finished) st->finished = 1;
  }
  // ProblemOut: X[12] | status | nstats | ncorr[SRRG2_MAX_SLICES] | H[36] | seq
  constexpr int W_STATUS = 12, W_NSTATS = 13, W_NCORR = 14, W_H = W_NCORR + SRRG2_MAX_SLICES, W_SEQ = W_H + 36;
  static_assert(offsetof(ProblemOut, status) == 4 * W_STATUS && offsetof(ProblemOut, ncorr) == 4 * W_NCORR &&
                  offsetof(ProblemOut, H) == 4 * W_H && offsetof(ProblemOut, seq) == 4 * W_SEQ && W_SEQ < 64,
                "one word of the record per lane");
  const int ncorr_w = __shfl(ncorr_s, (lane - W_NCORR) & 63);
  const float h_w   = (float) __shfl(fin.Hl, (lane - W_H) & 63);
  int word = __float_as_int(xl);
  if (lane == W_STATUS) word = status;
  if (lane == W_NSTATS) word = fin.nstats;
  if (lane >= W_NCORR && lane < W_H) word = ncorr_w;
  if (lane >= W_H && lane < W_SEQ) word = lane - W_H < D * D ? __float_as_int(h_w) : __float_as_int(0.f);
  if (lane < W_SEQ) reinterpret_cast<int*>(&outs_host[prob])[lane] = word;
  {  // the iteration statistics, 8 words each: the earlier records from memory (earlier kernels wrote them), this step's from its lanes
    const int nrec  = min(fin.nstats, C.max_stats);
    const int n     = nrec * 8;
    const int* src  = reinterpret_cast<const int*>(stats + (size_t) prob * C.max_stats);
    int* dst        = reinterpret_cast<int*>(stats_host + (size_t) prob * C.max_stats);
    const int lastw = __shfl(fin.sw, lane & 7);
    for (int k = lane; k < n; k += 64) dst[k] = (k >> 3) == fin.nstats - 1 ? lastw : src[k];
  }
  __threadfence_system();
  if (lane == 0) *reinterpret_cast<volatile int*>(&outs_host[prob].seq) = C.seq;  // the host polls this word
}
template <int D, bool PRIORS>
__global__ __launch_bounds__(64) void k_icp_final_wave(CtlParams C, SliceDev S, ProblemState* __restrict__ states,
                                                        srrg2_iteration_stats* __restrict__ stats,
                                                        ProblemOut* __restrict__ outs_host,
                                                        srrg2_iteration_stats* __restrict__ stats_host, int with_post) {
  final_wave_body<D, 1, PRIORS>(C, &S, 1, states, stats, outs_host, stats_host, with_post);
}
// ... of a pack of projective slices that share one association (SE(3)).  A four-slice pack and the control parameters do not fit
// one kernel's 4 KB of arguments: the parameters are read from their device copy (FusedCtl::ctl, written by this compute()'s
// prologue: the same record but for the fields of the control launches -- epoch, parity --, which this kernel takes from the pack).
template <bool PRIORS>
__global__ __launch_bounds__(64) void k_icp_final_wave_pack(SlicePack P, int ns, ProblemState* __restrict__ states,
                                                             srrg2_iteration_stats* __restrict__ stats,
                                                             ProblemOut* __restrict__ outs_host,
                                                             srrg2_iteration_stats* __restrict__ stats_host, int with_post) {
  final_wave_body<6, 4, PRIORS>(*P.s[0].fc.ctl, P.s, ns, states, stats, outs_host, stats_host, with_post);
}
void launch_icp_final_wave(const CtlParams& C, const SliceDev& S, ProblemState* states, srrg2_iteration_stats* stats,
                           ProblemOut* outs_host, srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s) {
  const dim3 grid(C.nprob > 0 ? C.nprob : C.K);
#define FINAL_WAVE_LAUNCH(D_, PRIORS_) \
  hipLaunchKernelGGL((k_icp_final_wave<D_, PRIORS_>), grid, dim3(64), 0, s, C, S, states, stats, outs_host, stats_host, with_post ? 1 : 0)
  if (C.variable_kind == SRRG2_SE2_RIGHT) {
    if (S.fc.prior_mask)
      FINAL_WAVE_LAUNCH(3, true);
    else
      FINAL_WAVE_LAUNCH(3, false);
  } else {
    if (S.fc.prior_mask)
      FINAL_WAVE_LAUNCH(6, true);
    else
      FINAL_WAVE_LAUNCH(6, false);
  }
#undef FINAL_WAVE_LAUNCH
}
void launch_icp_final_wave_pack(const SliceDev* slices, const ProblemDev* const* probs, int nslices, ProblemState* states,
                                srrg2_iteration_stats* stats, ProblemOut* outs_host, srrg2_iteration_stats* stats_host,
                                bool with_post, hipStream_t s) {
  if (nslices <= 0 || nslices > 4) return;
  SlicePack P;
  for (int z = 0; z < 4; ++z) {
    P.s[z]     = slices[z < nslices ? z : 0];
    P.probs[z] = probs[z < nslices ? z : 0];
  }
  if (P.s[0].fc.prior_mask)
    hipLaunchKernelGGL(k_icp_final_wave_pack<true>, dim3(1), dim3(64), 0, s, P, nslices, states, stats, outs_host, stats_host, with_post ? 1 : 0);
  else
    hipLaunchKernelGGL(k_icp_final_wave_pack<false>, dim3(1), dim3(64), 0, s, P, nslices, states, stats, outs_host, stats_host, with_post ? 1 : 0);
}
void launch_icp_control_final(const CtlParams& C, ProblemState* states, srrg2_iteration_stats* stats, ProblemOut* outs_host,
                               srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_control_final, dim3(C.nprob > 0 ? C.nprob : C.K), dim3(256), 0, s, C, states, stats, outs_host, stats_host,
                     with_post ? 1 : 0);
}
void launch_icp_post(const CtlParams& C, ProblemState* states, const srrg2_iteration_stats* stats, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_post, dim3(C.nprob > 0 ? C.nprob : C.K), dim3(64), 0, s, C, states, stats);
}
void launch_icp_finalize(const CtlParams& C, ProblemState* states, const srrg2_iteration_stats* stats,
                         ProblemOut* outs_host, srrg2_iteration_stats* stats_host, bool with_post, hipStream_t s) {
  hipLaunchKernelGGL(k_icp_finalize, dim3(C.nprob > 0 ? C.nprob : C.K), dim3(64), 0, s, C, states, stats, outs_host, stats_host, with_post ? 1 : 0);
}

}  // namespace srrg2amd
