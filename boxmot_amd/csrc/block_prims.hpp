// Workgroup / wavefront primitives used by the tracker kernels (gfx950, wave64).
//
// Everything here is plain HIP device code.  Wave-level exchanges use the
// 64-lane shuffles / ballots; workgroup-level results go through a small LDS
// array and two barriers.  Ordered list semantics of the reference tracker
// (Python lists, botsort_utils.py:10-52) are realised with order-preserving
// stream compaction (`block_append_if`).
#pragma once

#include "kernel_macros.hpp"

namespace bm {

constexpr int WAVE = 64;
constexpr int MAX_WAVES = 16;   // workgroups of up to 1024 threads

struct Ctx {
    int tid, nthr, lane, wave, nwaves;
    int* s_int;       // [MAX_WAVES + 1] LDS
    double* s_dbl;    // [MAX_WAVES] LDS
};

__device__ inline Ctx make_ctx(int* s_int, double* s_dbl) {
    Ctx c;
    c.tid = threadIdx.x;
    c.nthr = blockDim.x;
    c.lane = c.tid & (WAVE - 1);
    c.wave = c.tid / WAVE;
    c.nwaves = c.nthr / WAVE;
    c.s_int = s_int;
    c.s_dbl = s_dbl;
    return c;
}

__device__ inline double wave_sum(double v) {
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

__device__ inline float wave_sum(float v) {
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// Exclusive prefix of 0/1 flags over the whole workgroup, in thread order.
// Returns this thread's prefix; `total` is workgroup-uniform.  2 barriers.
__device__ inline int block_scan_flags(const Ctx& c, bool flag, int& total) {
    const unsigned long long m = __ballot(flag);
    const int in_wave = __popcll(m & ((1ull << c.lane) - 1ull));
    if (c.lane == 0) c.s_int[c.wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < c.nwaves; ++w) {
        const int n = c.s_int[w];
        if (w < c.wave) base += n;
        tot += n;
    }
    __syncthreads();
    total = tot;
    return base + in_wave;
}

// dst[dst_n + k] = value(i) for the k-th i in [0, n) (ascending) with pred(i).
// Returns the new length (uniform).  Results are visible after the call.
template <class Pred, class Val>
__device__ inline int block_append_if(const Ctx& c, int n, Pred pred, Val value, int* dst, int dst_n) {
    for (int base = 0; base < n; base += c.nthr) {
        const int i = base + c.tid;
        const bool f = (i < n) && pred(i);
        int total;
        const int pos = block_scan_flags(c, f, total);
        if (f) dst[dst_n + pos] = value(i);
        dst_n += total;
    }
    __syncthreads();
    return dst_n;
}

// Wave minima on the DPP path (row rotations by 8, 4, 2, 1 inside the 16-lane rows, then the four row results as scalars): a
// minimum does not depend on the order it is formed in, so these return exactly what a shuffle butterfly would, at a fraction of
// its latency (six dependent LDS-crossbar round trips per 32-bit half).
template <int CTRL> __device__ inline double dpp_ror(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = BM_DPP_U32(0u, (unsigned)b, CTRL, false), hi = BM_DPP_U32(0u, (unsigned)(b >> 32), CTRL, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <int LANE> __device__ inline double lane_of(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = BM_READLANE_U32((unsigned)b, LANE), hi = BM_READLANE_U32((unsigned)(b >> 32), LANE);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ inline double min_f64(double a, double b) { return b < a ? b : a; }
__device__ inline double wave_min(double v) {
    v = min_f64(v, dpp_ror<0x128>(v)); v = min_f64(v, dpp_ror<0x124>(v)); v = min_f64(v, dpp_ror<0x122>(v)); v = min_f64(v, dpp_ror<0x121>(v));
    return min_f64(min_f64(lane_of<0>(v), lane_of<16>(v)), min_f64(lane_of<32>(v), lane_of<48>(v)));
}
__device__ inline int wave_min(int v) {
    int o;
    o = (int)BM_DPP_U32(0u, (unsigned)v, 0x128, false); v = o < v ? o : v; o = (int)BM_DPP_U32(0u, (unsigned)v, 0x124, false); v = o < v ? o : v;
    o = (int)BM_DPP_U32(0u, (unsigned)v, 0x122, false); v = o < v ? o : v; o = (int)BM_DPP_U32(0u, (unsigned)v, 0x121, false); v = o < v ? o : v;
    const int a = (int)BM_READLANE_U32((unsigned)v, 0), b = (int)BM_READLANE_U32((unsigned)v, 16), c = (int)BM_READLANE_U32((unsigned)v, 32),
              d = (int)BM_READLANE_U32((unsigned)v, 48);
    const int ab = b < a ? b : a, cd = d < c ? d : c;
    return cd < ab ? cd : ab;
}

// Lexicographic (value, index) minimum over the workgroup.  Threads without a
// candidate pass idx < 0.  Result is uniform; idx < 0 when nobody had one.
// (The value minimum over the candidates, then the lowest index holding it -- in the wave on the DPP path.  Threads without a
// candidate take part with +infinity, so a candidate of ANY value up to and including +infinity is returned -- a wave whose only
// candidates are +inf yields the lowest of their indices, as the shuffle version did (round-4 advisor finding: with DBL_MAX as the
// filler an infinite candidate lost to the filler and came back as "none").  NaN is outside the contract (the minimum of a wave that
// holds one is unspecified): the callers compare with `<` before they offer a value, which a NaN never passes.)
__device__ inline void block_argmin(const Ctx& c, double v, int idx, double& out_v, int& out_i) {
    const double NONE_V = __builtin_inf();
    constexpr int NONE_I = 0x7fffffff;
    const bool any = __ballot(idx >= 0) != 0ull;
    const double wv = wave_min(idx >= 0 ? v : NONE_V);
    const int wi = wave_min((idx >= 0 && v == wv) ? idx : NONE_I);
    if (c.lane == 0) { c.s_dbl[c.wave] = wv; c.s_int[c.wave] = (any && wi != NONE_I) ? wi : -1; }
    __syncthreads();
    double bv = 0.0;
    int bi = -1;
    for (int w = 0; w < c.nwaves; ++w) {
        const double ov = c.s_dbl[w];
        const int oi = c.s_int[w];
        if ((oi >= 0) && (bi < 0 || ov < bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    __syncthreads();
    out_v = bv;
    out_i = bi;
}

}  // namespace bm
