// Workgroup / wavefront primitives used by the tracker kernels (gfx950, wave64).
//
// Everything here is plain HIP device code.  Wave-level exchanges use the
// 64-lane shuffles / ballots; workgroup-level results go through a small LDS
// array and two barriers.  Ordered list semantics of the reference tracker
// (Python lists, botsort_utils.py:10-52) are realised with order-preserving
// stream compaction (`block_append_if`).
#pragma once

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

// Lexicographic (value, index) minimum over the workgroup.  Threads without a
// candidate pass idx < 0.  Result is uniform; idx < 0 when nobody had one.
__device__ inline void block_argmin(const Ctx& c, double v, int idx, double& out_v, int& out_i) {
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, WAVE);
        const int oi = __shfl_xor(idx, off, WAVE);
        const bool take = (oi >= 0) && (idx < 0 || ov < v || (ov == v && oi < idx));
        if (take) { v = ov; idx = oi; }
    }
    if (c.lane == 0) { c.s_dbl[c.wave] = v; c.s_int[c.wave] = idx; }
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
