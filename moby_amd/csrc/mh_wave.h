// Wavefront-level building blocks for the many-worlds kernels (gfx950, wave64).
// One wavefront owns one world; lane i owns variable/row i.  Everything that
// steers control flow is made wave-uniform (SGPR) with readfirstlane so the
// pivoting loops compile to scalar branches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MH_WAVE 64
#define MH_DEV __device__ __forceinline__

namespace mh {

MH_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

MH_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
MH_DEV unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
MH_DEV double uni(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
MH_DEV uint64_t uni(uint64_t v) {
  unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffull));
  unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

// value held by lane `src` (src must be wave-uniform)
MH_DEV double read_lane(double v, int src) {
  long long b = __double_as_longlong(v);
  int s = __builtin_amdgcn_readfirstlane(src);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), s);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), s);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
MH_DEV int read_lane(int v, int src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }
MH_DEV unsigned read_lane(unsigned v, int src) { return (unsigned)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }

MH_DEV uint64_t ballot(bool p) { return __ballot(p); }
MH_DEV uint64_t lanes_below(int lane) { return (lane >= 64) ? ~0ull : ((1ull << lane) - 1ull); }
MH_DEV uint64_t bit(int i) { return 1ull << i; }
MH_DEV int popc(uint64_t m) { return __popcll(m); }
MH_DEV int ctz(uint64_t m) { return __ffsll((long long)m) - 1; }
// index of the r-th (0-based) set bit of m; m must have > r bits (uniform inputs)
MH_DEV int nth_set_bit(uint64_t m, int r) {
  for (int i = 0; i < r; i++) m &= m - 1;
  return ctz(m);
}

// ---- DPP wave reductions (VALU-latency, no LDS crossbar) -------------------
// Classic gfx9 pattern: xor-1 / xor-2 inside a quad, half-mirror (8), mirror
// (16), then row_bcast15 / row_bcast31 to fold the four rows; lane 63 ends up
// with the reduction over all 64 lanes.
template <int CTRL, int ROW_MASK>
MH_DEV double dpp_move(double v) {
  long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
#define MH_DPP_REDUCE(v, PICK)                                            \
  { double o_;                                                            \
    o_ = dpp_move<0xB1, 0xf>(v);  v = PICK(o_, v);  /* quad_perm [1,0,3,2] */ \
    o_ = dpp_move<0x4E, 0xf>(v);  v = PICK(o_, v);  /* quad_perm [2,3,0,1] */ \
    o_ = dpp_move<0x141, 0xf>(v); v = PICK(o_, v);  /* row_half_mirror    */ \
    o_ = dpp_move<0x140, 0xf>(v); v = PICK(o_, v);  /* row_mirror         */ \
    o_ = dpp_move<0x142, 0xa>(v); v = PICK(o_, v);  /* row_bcast15        */ \
    o_ = dpp_move<0x143, 0xc>(v); v = PICK(o_, v);  /* row_bcast31        */ }
// v_min_f64 / v_max_f64 (one instruction) instead of compare + two selects.  Their treatment of NaN is part of the contract: fmin / fmax
// return the OTHER operand when one is NaN, so a reduction skips NaNs the way std::min_element's `<` scan skips one that is not first
// -- lcp_lemke's ratio test feeds NaN ratios into wave_min (mh_lcp_wave.h) and relies on exactly that (the case "the FIRST ratio is
// NaN" is handled there separately).  A -0 / +0 tie only decides the sign of a zero that is then compared, never stored.
#define MH_PICK_MIN(o, v) __builtin_fmin((o), (v))
#define MH_PICK_MAX(o, v) __builtin_fmax((o), (v))

MH_DEV double wave_max(double v) { MH_DPP_REDUCE(v, MH_PICK_MAX); return read_lane(v, 63); }
MH_DEV double wave_min(double v) { MH_DPP_REDUCE(v, MH_PICK_MIN); return read_lane(v, 63); }

// first-index argmin over the lanes with valid==true; returns uniform (value, lane).
// Ties resolve to the lowest lane, as std::min_element does: reduce the value,
// then ballot the lanes that hold it.
MH_DEV void argmin_first(double v, bool valid, double& vmin, int& imin) {
  // std::min_element keeps its FIRST element when that is a NaN (nothing compares less than a NaN), and skips a NaN anywhere else -- the reduction below skips them all
  // (v_min_f64 returns the other operand), so the first-lane case is put right by hand; it only arises in worlds whose velocities have overflowed (tests/tools/fuzz_throw.py)
  const uint64_t nanm = ballot(valid && v != v);
  const uint64_t vm = nanm ? ballot(valid) : 0ull;
  if (!valid) v = __longlong_as_double(0x7ff0000000000000ll); // +inf
  vmin = wave_min(v);
  const uint64_t m = ballot(valid && v == vmin);
  imin = m ? ctz(m) : 0x7fffffff;
  if (nanm & (vm & (0ull - vm))) { imin = ctz(vm); vmin = __longlong_as_double(0x7ff8000000000000ll); }
}
// *std::min_element / *std::max_element over the lanes with valid == true, NaN semantics included (a NaN is skipped unless it is the FIRST element, which then stays)
MH_DEV double min_element_value(double v, bool valid) { double m; int i; argmin_first(v, valid, m, i); return m; }
MH_DEV double max_element_value(double v, bool valid) {
  const uint64_t nanm = ballot(valid && v != v);
  const uint64_t vm = nanm ? ballot(valid) : 0ull;
  if (nanm & (vm & (0ull - vm))) return __longlong_as_double(0x7ff8000000000000ll);
  if (!valid) v = __longlong_as_double(0xfff0000000000000ll); // -inf
  MH_DPP_REDUCE(v, MH_PICK_MAX);
  return read_lane(v, 63);
}
// first-index argmax (idamax over |a| supplied by the caller, a >= 0)
MH_DEV void argmax_first(double v, bool valid, double& vmax, int& imax) {
  if (!valid) v = -1.0;
  vmax = wave_max(v);
  const uint64_t m = ballot(valid && v == vmax);
  imax = m ? ctz(m) : 0x7fffffff;
}

// x / d for 0 <= x < 4096 and a small (wave-uniform, run-time) divisor 1 <= d <= 64: one multiply and a
// shift instead of the ~30-instruction expansion of an integer division by a non-constant.
// m = floor(2^18 / d) + 1 is exact for x * d < 2^18.
MH_DEV int small_div(int x, int d) { const unsigned m = (262144u / (unsigned)d) + 1u; return (int)(((unsigned)x * m) >> 18); }
MH_DEV unsigned small_div_magic(int d) { return (262144u / (unsigned)d) + 1u; }
MH_DEV int small_div_m(int x, unsigned m) { return (int)(((unsigned)x * m) >> 18); }

// forward permute: every lane sends v to lane dest (dest must be a permutation
// of 0..63 across the wave)
MH_DEV double push_to(double v, int dest) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_ds_permute(dest << 2, (int)(b & 0xffffffffll));
  int hi = __builtin_amdgcn_ds_permute(dest << 2, (int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// LDS visibility point between lanes of the SAME wavefront (kernels here run one wave per workgroup,
// __launch_bounds__(64)).
// For a one-wave workgroup __syncthreads() compiles to `s_waitcnt lgkmcnt(0)` (no s_barrier).  A
// wavefront-scope release/acquire fence pair would drop even that wait (the LDS unit executes one
// wave's ds_ operations in issue order); measured on the world kernel it is worth 0.6 % (21.60 vs
// 21.73 ms), so the conservative form stays the default (-DMH_WAVE_SYNC_FENCE selects the other).
MH_DEV void wave_sync() {
#ifdef MH_WAVE_SYNC_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  __syncthreads();
#endif
}

// ---------------------------------------------------------------------------
// glibc rand() (TYPE_3) with the 31-word ring spread over lanes 0..30: lane s
// holds ring slot s; `idx` (uniform) is the slot the next output overwrites.
// Same layout as oracle/glibc_rand.h so states can be exchanged verbatim.
// Reference consumers: src/LCP.cpp:208,620,641,688.
struct WaveRand {
  unsigned r;    // per-lane ring word (lanes 0..30)
  int idx;       // uniform
  MH_DEV void load(const uint32_t* st) { int l = lane_id(); r = (l < 31) ? st[l] : 0u; idx = uni((int)st[31]); }
  MH_DEV void store(uint32_t* st) const { int l = lane_id(); if (l < 31) st[l] = r; if (l == 31) st[31] = (unsigned)idx; }
  MH_DEV int next() {
    int i3 = idx + 28; if (i3 >= 31) i3 -= 31;
    unsigned a = read_lane(r, idx), b = read_lane(r, i3);
    unsigned v = a + b;
    if (lane_id() == idx) r = v;
    idx = (idx + 1 == 31) ? 0 : idx + 1;
    return (int)(v >> 1);
  }
  MH_DEV void skip(unsigned m) { for (unsigned i = 0; i < m; i++) (void)next(); }     // m draws whose values nobody looks at
};

} // namespace mh
