// Camera-motion estimation on the device: the sparse-optical-flow estimator (boxmot/motion/cmc/sof.py:14-147) -- BoT-SORT's YAML
// default cmc_method and the estimator DeepOCSORT constructs (deepocsort.py:297) -- with the reference's fixed arguments:
// BaseCMC.preprocess / generate_mask (base_cmc.py:30-105, scale 0.15), goodFeaturesToTrack(1000, 0.01, minDistance 1, blockSize 3),
// cornerSubPix((5, 5), 30 / 0.01) on the initialising frame, calcOpticalFlowPyrLK(21 x 21, maxLevel 3, 30 / 0.01) and
// estimateAffinePartial2D(RANSAC, 3.0) + the inlier test of sof.py:131-138.
//
// The reference delegates the numerics to OpenCV.  What is built here are those algorithms in the structure of OpenCV's sources
// (featureselect.cpp / corner.cpp, cornersubpix.cpp + getRectSubPix, lkpyramid.cpp, ptsetreg.cpp + levmarq.cpp; DESIGN.md 4.9 says what
// the kernels are tested against and what is and is not pinned -- OpenCV itself is absent offline), one stream per grid row:
//
//   k_sof_preprocess   BGR frame -> BGR2GRAY -> INTER_LINEAR resize by `scale` -> 8-bit level 0 of the frame's pyramid
//   k_sof_pyrdown      one pyramid level: 5 x 5 [1 4 6 4 1]^2 / 256 with reflected borders
//   k_sof_scharr       Scharr 3/10/3 derivatives of a level as int16 pairs (kept with the frame: they are the PREVIOUS frame's
//                      derivatives when the next frame is tracked)
//   k_sof_lk           one WAVEFRONT per keypoint, all pyramid levels and Newton steps inside: the 21 x 21 window is spread over the
//                      lanes (7 pixels each), the template patch and its derivatives stay in registers, the 2 x 2 structure tensor and
//                      the mismatch vector are exact integer sums reduced across the wave, the 14-bit fixed-point bilinear weights
//                      and every fp32 cast are where lkpyramid.cpp has them
//   k_sof_estimate     one workgroup per stream: ordered compaction of the tracked points, RANSAC over 2-point similarity models
//                      with cv::RNG(-1)'s draw sequence (every thread carries the generator; the inlier count of a model is one
//                      workgroup reduction), RANSACUpdateNumIters, the Levenberg-Marquardt refinement of the 4-parameter model over
//                      the inliers in fp64, the min_inliers / min_inlier_ratio test, translation divided by `scale`
//   k_sof_eigen        minimum-eigenvalue map (Sobel / 3060, 3 x 3 box sums) + the detection mask (central 96 %, minus the boxes)
//                      + the masked maximum (one atomic per workgroup on an order-preserving key)
//   k_sof_candidates   THRESH_TOZERO at 0.01 max, 3 x 3 local-maximum test on interior pixels -> unordered candidate list
//   k_sof_rank         rank of every candidate by (value, address) descending; ranks < 1000 are written in order
//   k_sof_subpix       cornerSubPix, one wavefront per corner (initialising frame only)
//   k_sof_finalize     the estimator's state machine: which keypoints the next frame tracks, `initialized`
// Nothing returns to the host between the kernels: the per-stream state word decides which of them do work.
#pragma once

#include <stdint.h>

#include "cmc_ecc.hpp"

namespace bm {

constexpr int SOF_MAX_CORNERS = 1000, SOF_WIN = 21, SOF_MAX_LEVELS = 4, SOF_THREADS = 256;
constexpr int SOF_W_BITS = 14;

struct SofLevels {                      // pyramid geometry (host: sof_levels): level l is h[l] x w[l] at element offset off[l]
    int n;
    int h[SOF_MAX_LEVELS], w[SOF_MAX_LEVELS];
    long off[SOF_MAX_LEVELS];
    long total;
};

struct SofState {                       // per stream
    int initialized, n_prev;            // carried from frame to frame
    int mode;                           // this frame: 0 initialising, 1 tracked, 2 too few tracked points (sof.py:142-147)
    int n_valid, n_inliers, ransac_iters, estimated;
    int n_cand, n_kps;
    unsigned max_key;
    int pad[2];
};

struct SofParams { double scale; int min_inliers; double min_inlier_ratio; double thresh; };

__device__ inline int sof_r101(int i, int n) {           // BORDER_REFLECT_101, |overshoot| < n
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// buildOpticalFlowPyramid's level rule: a level is added while the NEXT size stays larger than the window in both directions
inline SofLevels sof_levels(int h, int w, int max_level = 3) {
    SofLevels lv{};
    long off = 0;
    for (int level = 0; level <= max_level && level < SOF_MAX_LEVELS; ++level) {
        lv.h[level] = h; lv.w[level] = w; lv.off[level] = off; lv.n = level + 1;
        off += (long)h * w;
        h = (h + 1) / 2; w = (w + 1) / 2;
        if (w <= SOF_WIN || h <= SOF_WIN) break;
    }
    lv.total = off;
    return lv;
}

__global__ void __launch_bounds__(256) k_sof_preprocess(const uint8_t* const* __restrict__ frames, uint8_t* __restrict__ out, long out_stride,
                                                        int rows, int cols, int h, int w, double inv_scale) {
    const int s = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= h * w) return;
    const int dy = e / w, dx = e - dy * w;
    const uint8_t* src = frames[s];
    const ResizeAxis ax = ecc_axis(dx, cols, inv_scale, true), ay = ecc_axis(dy, rows, inv_scale, false);
    const uint8_t* r0 = src + (long)ay.s0 * cols * 3;
    const uint8_t* r1 = src + (long)ay.s1 * cols * 3;
    const int S0 = ecc_gray(r0 + ax.s0 * 3) * ax.a0 + ecc_gray(r0 + ax.s1 * 3) * ax.a1;
    const int S1 = ecc_gray(r1 + ax.s0 * 3) * ax.a0 + ecc_gray(r1 + ax.s1 * 3) * ax.a1;
    int v = (((ay.a0 * (S0 >> 4)) >> 16) + ((ay.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[(long)s * out_stride + e] = (uint8_t)v;
}

// pyr: [S][lv.total] uint8; builds level `level` from level - 1
__global__ void __launch_bounds__(256) k_sof_pyrdown(uint8_t* __restrict__ pyr, long stride, SofLevels lv, int level) {
    const int dh = lv.h[level], dw = lv.w[level], sh = lv.h[level - 1], sw = lv.w[level - 1];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= dh * dw) return;
    const uint8_t* src = pyr + (long)blockIdx.y * stride + lv.off[level - 1];
    const int y = e / dw, x = e - y * dw;
    const int k[5] = {1, 4, 6, 4, 1};
    int v = 0;
    for (int r = 0; r < 5; ++r) {
        const uint8_t* row = src + (long)sof_r101(2 * y + r - 2, sh) * sw;
        int a = 0;
        for (int c = 0; c < 5; ++c) a += k[c] * row[sof_r101(2 * x + c - 2, sw)];
        v += k[r] * a;
    }
    pyr[(long)blockIdx.y * stride + lv.off[level] + e] = (uint8_t)((v + 128) >> 8);
}

// der: [S][lv.total][2] int16 (dx, dy) of every level
__global__ void __launch_bounds__(256) k_sof_scharr(const uint8_t* __restrict__ pyr, long stride, short* __restrict__ der, SofLevels lv, int level) {
    const int h = lv.h[level], w = lv.w[level];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= h * w) return;
    const uint8_t* im = pyr + (long)blockIdx.y * stride + lv.off[level];
    const int y = e / w, x = e - y * w;
    const int yu = sof_r101(y - 1, h), yd = sof_r101(y + 1, h);
    int t0[3], t1[3];
    for (int c = 0; c < 3; ++c) {
        const int xx = sof_r101(x + c - 1, w);
        const int a = im[yu * w + xx], m = im[y * w + xx], b = im[yd * w + xx];
        t0[c] = (a + b) * 3 + m * 10;
        t1[c] = b - a;
    }
    short* o = der + ((long)blockIdx.y * stride + lv.off[level] + e) * 2;
    o[0] = (short)(t0[2] - t0[0]);
    o[1] = (short)((t1[2] + t1[0]) * 3 + t1[1] * 10);
}

__device__ inline long long sof_wave_sum(long long v) {
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

struct SofWeights { int w00, w01, w10, w11; };
__device__ inline SofWeights sof_weights(float a, float b) {     // cvRound((1 - a)(1 - b) 2^14) ...; the last takes the remainder
    SofWeights q;
    const float sc = (float)(1 << SOF_W_BITS);
    q.w00 = (int)rintf((1.f - a) * (1.f - b) * sc);
    q.w01 = (int)rintf(a * (1.f - b) * sc);
    q.w10 = (int)rintf((1.f - a) * b * sc);
    q.w11 = (1 << SOF_W_BITS) - q.w00 - q.w01 - q.w10;
    return q;
}
__device__ inline int sof_descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

// One wavefront per keypoint.  prev_kps: [S][1000][2] fp32; next_pts likewise; status: [S][1000] uint8
__global__ void __launch_bounds__(SOF_THREADS) k_sof_lk(const uint8_t* __restrict__ pyr_prev, const uint8_t* __restrict__ pyr_next, long stride,
                                                        const short* __restrict__ der_prev, SofLevels lv, const float* __restrict__ prev_kps,
                                                        float* __restrict__ next_pts, uint8_t* __restrict__ status, const SofState* __restrict__ st,
                                                        int max_count, double eps2, double min_eig) {
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * (SOF_THREADS / 64) + (threadIdx.x >> 6);
    if (!st[s].initialized || k >= st[s].n_prev) return;                     // wavefront-uniform
    const float px = prev_kps[((long)s * SOF_MAX_CORNERS + k) * 2], py = prev_kps[((long)s * SOF_MAX_CORNERS + k) * 2 + 1];
    constexpr int NPIX = SOF_WIN * SOF_WIN, PER = (NPIX + 63) / 64;
    const float half = (float)((SOF_WIN - 1) * 0.5f);
    float ox = 0.f, oy = 0.f;                                                // nextPts[ptidx]
    int ok = 1;
    for (int level = lv.n - 1; level >= 0; --level) {
        const int rows = lv.h[level], cols = lv.w[level];
        const uint8_t* I = pyr_prev + (long)s * stride + lv.off[level];
        const uint8_t* J = pyr_next + (long)s * stride + lv.off[level];
        const short* dI = der_prev + ((long)s * stride + lv.off[level]) * 2;
        const float inv = (float)(1.0 / (double)(1 << level));
        float ppx = px * inv, ppy = py * inv, nx, ny;
        if (level == lv.n - 1) { nx = ppx; ny = ppy; }
        else { nx = ox * 2.f; ny = oy * 2.f; }
        ox = nx; oy = ny;
        ppx = ppx - half; ppy = ppy - half;
        const int ix = (int)floorf(ppx), iy = (int)floorf(ppy);
        if (ix < -SOF_WIN || ix >= cols || iy < -SOF_WIN || iy >= rows) { if (level == 0) ok = 0; continue; }
        SofWeights q = sof_weights(ppx - (float)ix, ppy - (float)iy);
        int Ip[PER], Ix[PER], Iy[PER];
        long long a11 = 0, a12 = 0, a22 = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = lane + 64 * i;
            Ip[i] = Ix[i] = Iy[i] = 0;
            if (e < NPIX) {
                const int wy = e / SOF_WIN, wx = e - wy * SOF_WIN;
                const int y0 = iy + wy, x0 = ix + wx;
                const int r0 = sof_r101(y0, rows), r1 = sof_r101(y0 + 1, rows), c0 = sof_r101(x0, cols), c1 = sof_r101(x0 + 1, cols);
                Ip[i] = sof_descale(I[r0 * cols + c0] * q.w00 + I[r0 * cols + c1] * q.w01 + I[r1 * cols + c0] * q.w10 + I[r1 * cols + c1] * q.w11,
                                    SOF_W_BITS - 5);
                // derivatives: constant-zero border
                auto d = [&](int yy, int xx, int c) { return (yy >= 0 && yy < rows && xx >= 0 && xx < cols) ? (int)dI[((long)yy * cols + xx) * 2 + c] : 0; };
                Ix[i] = sof_descale(d(y0, x0, 0) * q.w00 + d(y0, x0 + 1, 0) * q.w01 + d(y0 + 1, x0, 0) * q.w10 + d(y0 + 1, x0 + 1, 0) * q.w11, SOF_W_BITS);
                Iy[i] = sof_descale(d(y0, x0, 1) * q.w00 + d(y0, x0 + 1, 1) * q.w01 + d(y0 + 1, x0, 1) * q.w10 + d(y0 + 1, x0 + 1, 1) * q.w11, SOF_W_BITS);
                a11 += (long long)Ix[i] * Ix[i]; a12 += (long long)Ix[i] * Iy[i]; a22 += (long long)Iy[i] * Iy[i];
            }
        }
        const float FLT_SCALE = 1.f / (float)(1 << 20);
        const float A11 = (float)sof_wave_sum(a11) * FLT_SCALE, A12 = (float)sof_wave_sum(a12) * FLT_SCALE, A22 = (float)sof_wave_sum(a22) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dif = A11 - A22;
        const float min_e = (A22 + A11 - sqrtf(dif * dif + 4.f * A12 * A12)) / (float)(2 * SOF_WIN * SOF_WIN);
        if ((double)min_e < min_eig || D < 1.1920928955078125e-7f) { if (level == 0) ok = 0; continue; }
        D = 1.f / D;
        nx = nx - half; ny = ny - half;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            const int jx = (int)floorf(nx), jy = (int)floorf(ny);
            if (jx < -SOF_WIN || jx >= cols || jy < -SOF_WIN || jy >= rows) { if (level == 0) ok = 0; break; }
            q = sof_weights(nx - (float)jx, ny - (float)jy);
            long long b1 = 0, b2 = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = lane + 64 * i;
                if (e < NPIX) {
                    const int wy = e / SOF_WIN, wx = e - wy * SOF_WIN;
                    const int r0 = sof_r101(jy + wy, rows), r1 = sof_r101(jy + wy + 1, rows), c0 = sof_r101(jx + wx, cols), c1 = sof_r101(jx + wx + 1, cols);
                    const int diff = sof_descale(J[r0 * cols + c0] * q.w00 + J[r0 * cols + c1] * q.w01 + J[r1 * cols + c0] * q.w10 + J[r1 * cols + c1] * q.w11,
                                                 SOF_W_BITS - 5) - Ip[i];
                    b1 += (long long)diff * Ix[i]; b2 += (long long)diff * Iy[i];
                }
            }
            const float fb1 = (float)sof_wave_sum(b1) * FLT_SCALE, fb2 = (float)sof_wave_sum(b2) * FLT_SCALE;
            const float ddx = (A12 * fb2 - A22 * fb1) * D, ddy = (A12 * fb1 - A11 * fb2) * D;
            nx = nx + ddx; ny = ny + ddy;
            ox = nx + half; oy = ny + half;
            if ((double)ddx * (double)ddx + (double)ddy * (double)ddy <= eps2) break;
            if (j > 0 && fabs((double)(ddx + pdx)) < 0.01 && fabs((double)(ddy + pdy)) < 0.01) { ox = ox - ddx * 0.5f; oy = oy - ddy * 0.5f; break; }
            pdx = ddx; pdy = ddy;
        }
        if (ok && level == 0) {                                              // the error pass re-tests the final window
            const int jx = (int)floorf(ox - half), jy = (int)floorf(oy - half);
            if (jx < -SOF_WIN || jx >= cols || jy < -SOF_WIN || jy >= rows) ok = 0;
        }
    }
    if (lane == 0) {
        next_pts[((long)s * SOF_MAX_CORNERS + k) * 2] = ox; next_pts[((long)s * SOF_MAX_CORNERS + k) * 2 + 1] = oy;
        status[(long)s * SOF_MAX_CORNERS + k] = (uint8_t)ok;
    }
}

// ---- estimateAffinePartial2D --------------------------------------------------------------------------------------------------
struct SofRng {                                             // cv::RNG: multiply-with-carry
    unsigned long long state;
    __device__ unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a)) + a; }
};

// d = M^-1 v for a 4 x 4 system, Gaussian elimination with partial pivoting; a vanishing pivot leaves that component 0
__device__ inline void sof_solve4(const double (&M)[4][4], const double (&v)[4], double (&d)[4]) {
    double a[4][5];
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) a[i][j] = M[i][j]; a[i][4] = v[i]; }
    int piv_ok[4];
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (p != c) for (int j = 0; j < 5; ++j) { const double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        piv_ok[c] = fabs(a[c][c]) > 1e-300;
        if (!piv_ok[c]) continue;
        for (int r = c + 1; r < 4; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int j = c; j < 5; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int c = 3; c >= 0; --c) {
        double t = a[c][4];
        for (int j = c + 1; j < 4; ++j) t -= a[c][j] * d[j];
        d[c] = piv_ok[c] ? t / a[c][c] : 0.0;
    }
}

__device__ inline int sof_block_count(int v, int* red) {    // workgroup sum of a per-thread count, returned to every thread
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    __syncthreads();
    return t;
}

// ordered compaction of the thread's PT consecutive-stride points: element e = tid + 256 i keeps its order (chunks of 256)
__device__ inline int sof_block_prefix(int flag, int* red) {     // exclusive prefix of a 0 / 1 flag over the workgroup + the total in red[8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) red[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { if (i < wave) base += red[i]; tot += red[i]; }
    const int pre = base + __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (threadIdx.x == 0) red[8] = tot;
    __syncthreads();
    return pre;
}

__device__ inline void sof_errors_model(const double (&M)[6], float (&F)[6]) { for (int i = 0; i < 6; ++i) F[i] = (float)M[i]; }
__device__ inline float sof_err(const float (&F)[6], float fx, float fy, float tx, float ty) {
    const float a = ((F[0] * fx + F[1] * fy) + F[2]) - tx;
    const float b = ((F[3] * fx + F[4] * fy) + F[5]) - ty;
    return a * a + b * b;
}

// prev_kps / next_pts / status as in k_sof_lk; valid_to: [S][1000][2] (the tracked points that survive, sof.py:123-125);
// out_warp: fp64 [S][6], row-major 2 x 3, translation in FULL-RESOLUTION pixels
__global__ void __launch_bounds__(SOF_THREADS) k_sof_estimate(const float* __restrict__ prev_kps, const float* __restrict__ next_pts,
                                                              const uint8_t* __restrict__ status, float* __restrict__ valid_to,
                                                              SofState* __restrict__ st_all, double* __restrict__ out_warp, SofParams prm) {
    __shared__ float fx[SOF_MAX_CORNERS], fy[SOF_MAX_CORNERS], tx[SOF_MAX_CORNERS], ty[SOF_MAX_CORNERS];
    __shared__ int red[16];
    __shared__ double dred[8 * 16];
    const int s = blockIdx.x, tid = threadIdx.x;
    SofState* st = st_all + s;
    double* W = out_warp + (long)s * 6;
    if (tid == 0) { W[0] = 1; W[1] = 0; W[2] = 0; W[3] = 0; W[4] = 1; W[5] = 0; st->n_valid = 0; st->n_inliers = 0; st->ransac_iters = 0; st->estimated = 0; }
    if (!st->initialized) { if (tid == 0) st->mode = 0; return; }
    const int n_prev = st->n_prev;
    // ---- status == 1 points, in order
    int n = 0;
    for (int base = 0; base < n_prev; base += SOF_THREADS) {
        const int e = base + tid;
        const int f = e < n_prev && status[(long)s * SOF_MAX_CORNERS + e] == 1;
        const int pre = sof_block_prefix(f, red);
        if (f) {
            const long g = ((long)s * SOF_MAX_CORNERS + e) * 2;
            fx[n + pre] = prev_kps[g]; fy[n + pre] = prev_kps[g + 1]; tx[n + pre] = next_pts[g]; ty[n + pre] = next_pts[g + 1];
        }
        n += red[8];
        __syncthreads();
    }
    for (int e = tid; e < n; e += SOF_THREADS) { valid_to[((long)s * SOF_MAX_CORNERS + e) * 2] = tx[e]; valid_to[((long)s * SOF_MAX_CORNERS + e) * 2 + 1] = ty[e]; }
    if (tid == 0) st->n_valid = n;
    if (n < 4) { if (tid == 0) st->mode = 2; return; }
    if (tid == 0) st->mode = 1;
    // ---- RANSAC (RANSACPointSetRegistrator::run, model points 2, confidence 0.99, <= 2000 iterations)
    SofRng rng{0xFFFFFFFFFFFFFFFFull};
    const float t2 = (float)(prm.thresh * prm.thresh);
    int niters = 2000, best_count = 0, iters_done = 0;
    double best[6] = {1, 0, 0, 0, 1, 0};
    for (int it = 0; it < niters; ++it) {
        int i0 = rng.uniform(0, n), i1;
        do { i1 = rng.uniform(0, n); } while (i1 == i0);                     // getSubset: a repeated index is drawn again
        const double x1 = fx[i0], y1 = fy[i0], x2 = fx[i1], y2 = fy[i1], X1 = tx[i0], Y1 = ty[i0], X2 = tx[i1], Y2 = ty[i1];
        const double d = 1.0 / ((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
        const double S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2));
        const double S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2));
        const double S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2));
        const double S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2));
        const double M[6] = {S0, -S1, S2, S1, S0, S3};
        float F[6];
        sof_errors_model(M, F);
        int c = 0;
        for (int e = tid; e < n; e += SOF_THREADS) c += sof_err(F, fx[e], fy[e], tx[e], ty[e]) <= t2 ? 1 : 0;
        const int good = sof_block_count(c, red);
        iters_done = it + 1;
        if (good > (best_count > 1 ? best_count : 1)) {
            for (int i = 0; i < 6; ++i) best[i] = M[i];
            best_count = good;
            // RANSACUpdateNumIters(0.99, outlier ratio, 2, niters)
            double ep = (double)(n - good) / (double)n;
            ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
            const double num = 1.0 - 0.99 > 2.2250738585072014e-308 ? 1.0 - 0.99 : 2.2250738585072014e-308;
            const double denom = 1.0 - (1.0 - ep) * (1.0 - ep);
            if (denom < 2.2250738585072014e-308) niters = 0;
            else {
                const double ln = log(num), ld = log(denom);
                niters = (ld >= 0 || -ln >= niters * (-ld)) ? niters : (int)rint(ln / ld);
            }
        }
    }
    if (tid == 0) st->ransac_iters = iters_done;
    if (best_count == 0) return;                                             // no model: identity (sof.py:107-115)
    // ---- inliers of the best model to the front, in order
    float bF[6];
    sof_errors_model(best, bF);
    int n_in = 0;
    {
        constexpr int PT = (SOF_MAX_CORNERS + SOF_THREADS - 1) / SOF_THREADS;
        float hx[PT], hy[PT], gx[PT], gy[PT];
        int pos[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int e = tid + SOF_THREADS * i;
            const int f = e < n && sof_err(bF, fx[e], fy[e], tx[e], ty[e]) <= t2;
            if (e < n) { hx[i] = fx[e]; hy[i] = fy[e]; gx[i] = tx[e]; gy[i] = ty[e]; }
            const int pre = sof_block_prefix(f, red);
            pos[i] = f ? n_in + pre : -1;
            n_in += red[8];
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) if (pos[i] >= 0) { fx[pos[i]] = hx[i]; fy[pos[i]] = hy[i]; tx[pos[i]] = gx[i]; ty[pos[i]] = gy[i]; }
        __syncthreads();
    }
    // ---- Levenberg-Marquardt refinement (LMSolverImpl::run, 10 iterations, eps = FLT_EPSILON) of (a, b, tx, ty), [[a -b tx] [b a ty]]
    double x[4] = {best[0], best[3], best[2], best[5]};
    {
        const double EPS = 1.1920928955078125e-7, DEPS = 2.220446049250313e-16;
        // J^T J of the linear model: sums of Mx^2 + My^2, Mx, My, 1
        double q[3] = {0, 0, 0};
        for (int e = tid; e < n_in; e += SOF_THREADS) { const double mx = fx[e], my = fy[e]; q[0] += mx * mx + my * my; q[1] += mx; q[2] += my; }
        ecc_reduce<3>(q, dred);
        const double nn = (double)n_in;
        const double A[4][4] = {{q[0], 0, q[1], q[2]}, {0, q[0], -q[2], q[1]}, {q[1], -q[2], nn, 0}, {q[2], q[1], 0, nn}};
        auto residual_sums = [&](const double (&h)[4], double (&o)[7]) {     // S, J^T r (4), max |r|, unused
            double a[6] = {0, 0, 0, 0, 0, 0};
            double mr = 0;
            for (int e = tid; e < n_in; e += SOF_THREADS) {
                const double mx = fx[e], my = fy[e];
                const double rx = h[0] * mx - h[1] * my + h[2] - (double)tx[e], ry = h[1] * mx + h[0] * my + h[3] - (double)ty[e];
                a[0] += rx * rx + ry * ry;
                a[1] += mx * rx + my * ry; a[2] += -my * rx + mx * ry; a[3] += rx; a[4] += ry;
                mr = fmax(mr, fmax(fabs(rx), fabs(ry)));
            }
            ecc_reduce<6>(a, dred);
            // the maximum: a workgroup max through the same buffer
            double m = mr;
            for (int k = 32; k > 0; k >>= 1) m = fmax(m, __shfl_xor(m, k, 64));
            if ((tid & 63) == 0) dred[tid >> 6] = m;
            __syncthreads();
            m = 0;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m = fmax(m, dred[i]);
            __syncthreads();
            for (int i = 0; i < 5; ++i) o[i] = a[i];
            o[5] = m; o[6] = 0;
        };
        double o[7];
        residual_sums(x, o);
        double S = o[0], v[4] = {o[1], o[2], o[3], o[4]}, rmax = o[5];
        const double Dg[4] = {A[0][0], A[1][1], A[2][2], A[3][3]};
        double lam = 1.0, lc = 0.75;
        for (int iter = 0;;) {
            double Ap[4][4], d[4], xd[4];
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) Ap[i][j] = A[i][j] + (i == j ? lam * Dg[i] : 0.0);
            sof_solve4(Ap, v, d);
            for (int i = 0; i < 4; ++i) xd[i] = x[i] - d[i];
            double od[7];
            residual_sums(xd, od);
            const double Sd = od[0];
            double dS = 0, tdv = 0;
            for (int i = 0; i < 4; ++i) { double Ad = 0; for (int j = 0; j < 4; ++j) Ad += A[i][j] * d[j]; dS += d[i] * (2.0 * v[i] - Ad); tdv += d[i] * v[i]; }
            const double R = (S - Sd) / (fabs(dS) > DEPS ? dS : 1.0);
            if (R > 0.75) { lam *= 0.5; if (lam < lc) lam = 0.0; }
            else if (R < 0.25) {
                double nu = (Sd - S) / (fabs(tdv) > DEPS ? tdv : 1.0) + 2.0;
                nu = nu < 2.0 ? 2.0 : (nu > 10.0 ? 10.0 : nu);
                if (lam == 0.0) {
                    double maxval = DEPS;
                    for (int c = 0; c < 4; ++c) {
                        double e4[4] = {0, 0, 0, 0}, col[4];
                        e4[c] = 1.0;
                        sof_solve4(A, e4, col);
                        maxval = fmax(maxval, fabs(col[c]));
                    }
                    lam = lc = 1.0 / maxval;
                    nu *= 0.5;
                }
                lam *= nu;
            }
            if (Sd < S) { S = Sd; for (int i = 0; i < 4; ++i) { x[i] = xd[i]; v[i] = od[1 + i]; } rmax = od[5]; }
            ++iter;
            double dmax = 0;
            for (int i = 0; i < 4; ++i) dmax = fmax(dmax, fabs(d[i]));
            if (!(iter < 10 && dmax >= EPS && rmax >= EPS)) break;
        }
    }
    if (tid == 0) {
        st->n_inliers = best_count;
        if (best_count >= prm.min_inliers && (double)best_count / (double)n >= prm.min_inlier_ratio) {
            const float H[6] = {(float)x[0], (float)(-x[1]), (float)x[2], (float)x[1], (float)x[0], (float)x[3]};
            W[0] = H[0]; W[1] = H[1]; W[3] = H[3]; W[4] = H[4];
            const float fs = (float)prm.scale;
            W[2] = prm.scale < 1.0 ? H[2] / fs : H[2];                       // fp32 division, sof.py:116-120
            W[5] = prm.scale < 1.0 ? H[5] / fs : H[5];
            st->estimated = 1;
        }
    }
}

// ---- goodFeaturesToTrack -------------------------------------------------------------------------------------------------------
__device__ inline unsigned sof_float_key(float v) {          // order-preserving: larger float <-> larger key; 0 is below every float
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float sof_key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ inline void sof_sobel(const uint8_t* im, int y, int x, int h, int w, float& dx, float& dy) {
    const float s = (float)(1.0 / (4.0 * 3.0 * 255.0)), k0 = 2.0f * s, k1 = s;
    const int yu = sof_r101(y - 1, h), yd = sof_r101(y + 1, h), xl = sof_r101(x - 1, w), xr = sof_r101(x + 1, w);
    const float a00 = im[yu * w + xl], a01 = im[yu * w + x], a02 = im[yu * w + xr];
    const float a10 = im[y * w + xl], a11 = im[y * w + x], a12 = im[y * w + xr];
    const float a20 = im[yd * w + xl], a21 = im[yd * w + x], a22 = im[yd * w + xr];
    const float rx0 = a02 - a00, rx1 = a12 - a10, rx2 = a22 - a20;           // rows [-1 0 1]
    dx = (rx0 + rx2) * k1 + rx1 * k0;                                        // columns [1 2 1] s
    const float ry0 = a01 * k0 + (a00 + a02) * k1, ry2 = a21 * k0 + (a20 + a22) * k1;
    (void)a11;
    dy = ry2 - ry0;
}

// eig: fp32 [S][h w]; mask: uint8 [S][h w]; dets: [S][max_dets][det_stride] fp32 tlbr in frame pixels, n_dets[S]
__global__ void __launch_bounds__(256) k_sof_eigen(const uint8_t* __restrict__ pyr, long stride, float* __restrict__ eig, uint8_t* __restrict__ mask,
                                                   const float* __restrict__ dets, const int* __restrict__ n_dets, int max_dets, int det_stride,
                                                   SofState* __restrict__ st, int h, int w, float scale) {
    __shared__ unsigned wmax[4];
    const int s = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned key = 0;
    if (e < h * w) {
        const uint8_t* im = pyr + (long)s * stride;
        const int y = e / w, x = e - y * w;
        float sxx[3], sxy[3], syy[3];
        for (int r = 0; r < 3; ++r) {
            const int yy = sof_r101(y + r - 1, h);
            float pxx[3], pxy[3], pyy[3];
            for (int c = 0; c < 3; ++c) {
                float dx, dy;
                sof_sobel(im, yy, sof_r101(x + c - 1, w), h, w, dx, dy);
                pxx[c] = dx * dx; pxy[c] = dx * dy; pyy[c] = dy * dy;
            }
            sxx[r] = (pxx[0] + pxx[1]) + pxx[2]; sxy[r] = (pxy[0] + pxy[1]) + pxy[2]; syy[r] = (pyy[0] + pyy[1]) + pyy[2];
        }
        const float a = ((sxx[0] + sxx[1]) + sxx[2]) * 0.5f, b = (sxy[0] + sxy[1]) + sxy[2], c = ((syy[0] + syy[1]) + syy[2]) * 0.5f;
        const float v = (a + c) - sqrtf((a - c) * (a - c) + b * b);
        eig[(long)s * h * w + e] = v;
        // generate_mask: the central region, minus the detections (truncated fp32 products, clipped to the image)
        int m = y >= (int)(0.02 * h) && y < (int)(0.98 * h) && x >= (int)(0.02 * w) && x < (int)(0.98 * w);
        const int nd = n_dets ? n_dets[s] : 0;
        for (int k = 0; k < nd && m; ++k) {
            const float* d = dets + ((long)s * max_dets + k) * det_stride;
            int x1 = (int)(d[0] * scale), y1 = (int)(d[1] * scale), x2 = (int)(d[2] * scale), y2 = (int)(d[3] * scale);
            x1 = x1 < 0 ? 0 : (x1 > w ? w : x1); x2 = x2 < 0 ? 0 : (x2 > w ? w : x2);
            y1 = y1 < 0 ? 0 : (y1 > h ? h : y1); y2 = y2 < 0 ? 0 : (y2 > h ? h : y2);
            if (x2 > x1 && y2 > y1 && x >= x1 && x < x2 && y >= y1 && y < y2) m = 0;
        }
        mask[(long)s * h * w + e] = (uint8_t)(m ? 255 : 0);
        if (m) key = sof_float_key(v);
    }
    for (int k = 32; k > 0; k >>= 1) { const unsigned o = __shfl_xor(key, k, 64); key = o > key ? o : key; }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) key = wmax[i] > key ? wmax[i] : key;
        if (key) atomicMax(&st[s].max_key, key);
    }
}

__device__ inline float sof_thresholded(float v, float thr) { return v > thr ? v : 0.0f; }

// cand: int [S][h w] addresses (unordered), count in st.n_cand
__global__ void __launch_bounds__(256) k_sof_candidates(const float* __restrict__ eig, const uint8_t* __restrict__ mask, int* __restrict__ cand,
                                                        SofState* __restrict__ st, int h, int w) {
    const int s = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= h * w) return;
    const int y = e / w, x = e - y * w;
    if (y < 1 || y >= h - 1 || x < 1 || x >= w - 1) return;
    const float* g = eig + (long)s * h * w;
    const unsigned mk = st[s].max_key;
    const double max_val = mk ? (double)sof_key_float(mk) : 0.0;
    const float thr = (float)(max_val * 0.01);
    const float v = sof_thresholded(g[e], thr);
    if (v == 0.0f || !mask[(long)s * h * w + e]) return;
    float m = v;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) m = fmaxf(m, sof_thresholded(g[(y + dy) * w + x + dx], thr));
    if (v != m) return;
    cand[(long)s * h * w + atomicAdd(&st[s].n_cand, 1)] = e;
}

// new_kps: [S][1000][2] fp32 (x, y), strongest first: value descending, then address descending
__global__ void __launch_bounds__(256) k_sof_rank(const float* __restrict__ eig, const int* __restrict__ cand, float* __restrict__ new_kps,
                                                  SofState* __restrict__ st, int h, int w) {
    __shared__ float tv[256];
    __shared__ int ta[256];
    const int s = blockIdx.y, n = st[s].n_cand;
    if (blockIdx.x == 0 && threadIdx.x == 0) st[s].n_kps = n < SOF_MAX_CORNERS ? n : SOF_MAX_CORNERS;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;                         // workgroup-uniform
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float* g = eig + (long)s * h * w;
    const int* cs = cand + (long)s * h * w;
    const int ai = i < n ? cs[i] : -1;
    const float vi = i < n ? g[ai] : 0.f;
    int rank = 0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + (int)threadIdx.x;
        ta[threadIdx.x] = j < n ? cs[j] : -1;
        tv[threadIdx.x] = j < n ? g[cs[j]] : 0.f;
        __syncthreads();
        const int lim = n - base < 256 ? n - base : 256;
        for (int k = 0; k < lim; ++k) rank += (tv[k] > vi || (tv[k] == vi && ta[k] > ai)) ? 1 : 0;
        __syncthreads();
    }
    if (i < n && rank < SOF_MAX_CORNERS) {
        float* o = new_kps + ((long)s * SOF_MAX_CORNERS + rank) * 2;
        o[0] = (float)(ai % w); o[1] = (float)(ai / w);
    }
}

// ---- cornerSubPix((5, 5), (-1, -1), 30 / 0.01): one wavefront per corner, in place, on the initialising frame ------------------------
__device__ inline float sof_rect_sample(const uint8_t* im, int h, int w, int ipx, int ipy, float a, float b, int rx0, int rx1, int ry0, int ry1,
                                        int i, int j) {      // getRectSubPix element (i, j) of a window whose top-left integer corner is (ipx, ipy)
    int r0 = ipy + i; r0 = r0 < 0 ? 0 : (r0 > h - 1 ? h - 1 : r0);
    int r1 = (i >= ry0 && i < ry1) ? r0 + 1 : r0; r1 = r1 > h - 1 ? h - 1 : r1;
    const float b1 = 1.f - b, b2 = b;
    if (j < rx0 || j >= rx1) {
        int c = ipx + (j < rx0 ? rx0 : rx1); c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
        return (float)im[r0 * w + c] * b1 + (float)im[r1 * w + c] * b2;
    }
    const int c = ipx + j;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b;
    return (((float)im[r0 * w + c] * a11 + (float)im[r0 * w + c + 1] * a12) + (float)im[r1 * w + c] * a21) + (float)im[r1 * w + c + 1] * a22;
}

__global__ void __launch_bounds__(SOF_THREADS) k_sof_subpix(const uint8_t* __restrict__ pyr, long stride, float* __restrict__ new_kps,
                                                            const SofState* __restrict__ st, int h, int w) {
    const int s = blockIdx.y, lane = threadIdx.x & 63;
    const int k = blockIdx.x * (SOF_THREADS / 64) + (threadIdx.x >> 6);
    if (st[s].mode != 0 || st[s].n_kps < 4 || k >= st[s].n_kps) return;    // wavefront-uniform
    const uint8_t* im = pyr + (long)s * stride;
    // exp(-(k / 5)^2) in fp32, k = 0..5: cornerSubPix's separable window, the literal the CPU restatement of the tests holds too
    const float wt[6] = {1.0f, 0.96078944f, 0.85214376f, 0.69767630f, 0.52729243f, 0.36787945f};
    constexpr int HALF = 5, N = 11, WIN = N + 2;
    float* p = new_kps + ((long)s * SOF_MAX_CORNERS + k) * 2;
    const float ctx = p[0], cty = p[1];
    float cix = ctx, ciy = cty;
    int iter = 0;
    for (;;) {
        const float cx = cix - (float)((WIN - 1) * 0.5f), cy = ciy - (float)((WIN - 1) * 0.5f);
        const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
        const float a = cx - (float)ipx, b = cy - (float)ipy;
        int rx0 = 0, rx1 = WIN, ry0 = 0, ry1 = WIN;
        if (!(0 <= ipx && ipx < w - WIN && 0 <= ipy && ipy < h - WIN)) {
            rx0 = ipx >= 0 ? 0 : (-ipx < WIN ? -ipx : WIN);
            rx1 = ipx < w - WIN ? WIN : (w - ipx - 1 > 0 ? w - ipx - 1 : 0);
            ry0 = ipy >= 0 ? 0 : (-ipy < WIN ? -ipy : WIN);
            ry1 = ipy < h - WIN ? WIN : (h - ipy - 1 > 0 ? h - ipy - 1 : 0);
        }
        double acc[5] = {0, 0, 0, 0, 0};
        for (int e = lane; e < N * N; e += 64) {
            const int i = e / N, j = e - i * N;                              // gradient window element; patch element (i + 1, j + 1)
            auto sp = [&](int ii, int jj) { return sof_rect_sample(im, h, w, ipx, ipy, a, b, rx0, rx1, ry0, ry1, ii, jj); };
            const double tgx = (double)(sp(i + 1, j + 2) - sp(i + 1, j)), tgy = (double)(sp(i + 2, j + 1) - sp(i, j + 1));
            const int ai = i - HALF < 0 ? HALF - i : i - HALF, aj = j - HALF < 0 ? HALF - j : j - HALF;
            const double m = (double)(wt[ai] * wt[aj]);
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
            const double ppx = (double)(j - HALF), ppy = (double)(i - HALF);
            acc[0] += gxx; acc[1] += gxy; acc[2] += gyy; acc[3] += gxx * ppx + gxy * ppy; acc[4] += gxy * ppx + gyy * ppy;
        }
        for (int q = 0; q < 5; ++q) for (int m = 32; m > 0; m >>= 1) acc[q] += __shfl_xor(acc[q], m, 64);
        const double det = acc[0] * acc[2] - acc[1] * acc[1];
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double sc = 1.0 / det;
        const float c2x = (float)((double)cix + acc[2] * sc * acc[3] - acc[1] * sc * acc[4]);
        const float c2y = (float)((double)ciy - acc[1] * sc * acc[3] + acc[0] * sc * acc[4]);
        const double err = (double)((c2x - cix) * (c2x - cix) + (c2y - ciy) * (c2y - ciy));
        cix = c2x; ciy = c2y;
        if (cix < 0 || cix >= (float)w || ciy < 0 || ciy >= (float)h) break;
        if (!(++iter < 30 && err > 0.01 * 0.01)) break;
    }
    if (fabs((double)(cix - ctx)) > HALF || fabs((double)(ciy - cty)) > HALF) { cix = ctx; ciy = cty; }
    if (lane == 0) { p[0] = cix; p[1] = ciy; }
}

// The estimator's state machine (sof.py:59-77, 95-98, 122-129, 142-147): which keypoints the next frame tracks
__global__ void __launch_bounds__(SOF_THREADS) k_sof_finalize(float* __restrict__ prev_kps, const float* __restrict__ new_kps,
                                                              const float* __restrict__ valid_to, SofState* __restrict__ st_all) {
    const int s = blockIdx.x, tid = threadIdx.x;
    SofState* st = st_all + s;
    const int mode = st->mode, n_kps = st->n_kps, n_valid = st->n_valid;
    const bool keep_tracked = mode == 1 && n_kps < 4;                        // sof.py:123-125: fall back to the tracked points
    const float* src = (keep_tracked ? valid_to : new_kps) + (long)s * SOF_MAX_CORNERS * 2;
    const int n = keep_tracked ? n_valid : n_kps;
    for (int e = tid; e < 2 * n; e += SOF_THREADS) prev_kps[(long)s * SOF_MAX_CORNERS * 2 + e] = src[e];
    if (tid == 0) {
        st->n_prev = n;
        st->initialized = mode == 1 ? 1 : (n_kps >= 4 ? 1 : 0);
        st->n_cand = 0; st->max_key = 0;                                     // ready for the next frame's detection
    }
}

// the previous frame of the next call = this frame: pyramid and derivatives
__global__ void __launch_bounds__(256) k_sof_commit(const uint8_t* __restrict__ pyr_cur, uint8_t* __restrict__ pyr_prev, const short* __restrict__ der_cur,
                                                    short* __restrict__ der_prev, long total) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long g = (long)blockIdx.y * total + e;
    pyr_prev[g] = pyr_cur[g];
    der_prev[2 * g] = der_cur[2 * g]; der_prev[2 * g + 1] = der_cur[2 * g + 1];
}

struct SofBuffers {                     // device memory of a handle, every array [S][...]
    uint8_t *pyr_prev, *pyr_cur;        // [S][lv.total]
    short *der_prev, *der_cur;          // [S][lv.total][2]
    float* eig; uint8_t* mask; int* cand;                                    // [S][h w]
    float *prev_kps, *new_kps, *next_pts, *valid_to;                         // [S][1000][2]
    uint8_t* status;                    // [S][1000]
    SofState* st;                       // [S]
    double* warp;                       // [S][6]
};

// One frame of streams [s0, s0 + n): the whole kernel sequence.  `launch(kernel, grid_x, grid_y, threads, args...)` is
// hipLaunchKernelGGL on the handle's stream in the library and the CPU-thread launcher in tests/host_emu -- the same sequence.
// d_frames: device table of n frame pointers; d_dets: [n][max_dets][det_stride] fp32 (tlbr first), d_ndets [n] (both may be null).
template <class Launch>
void sof_frame(Launch& launch, const SofBuffers& B, const SofLevels& lv, int s0, int n, const uint8_t* const* d_frames, int rows, int cols,
               const float* d_dets, const int* d_ndets, int max_dets, int det_stride, const SofParams& prm) {
    const int h = lv.h[0], w = lv.w[0], P = h * w;
    const long T = lv.total, K = SOF_MAX_CORNERS;
    uint8_t* pc = B.pyr_cur + s0 * T; uint8_t* pp = B.pyr_prev + s0 * T;
    short* dc = B.der_cur + s0 * T * 2; short* dp = B.der_prev + s0 * T * 2;
    SofState* st = B.st + s0;
    float *pk = B.prev_kps + s0 * K * 2, *nk = B.new_kps + s0 * K * 2, *np_ = B.next_pts + s0 * K * 2, *vt = B.valid_to + s0 * K * 2;
    uint8_t* status = B.status + s0 * K;
    float* eig = B.eig + (long)s0 * P; uint8_t* mask = B.mask + (long)s0 * P; int* cand = B.cand + (long)s0 * P;
    const int pb = (P + 255) / 256;
    launch(k_sof_preprocess, pb, n, 256, d_frames, pc, T, rows, cols, h, w, 1.0 / prm.scale);
    for (int l = 1; l < lv.n; ++l) launch(k_sof_pyrdown, (lv.h[l] * lv.w[l] + 255) / 256, n, 256, pc, T, lv, l);
    for (int l = 0; l < lv.n; ++l) launch(k_sof_scharr, (lv.h[l] * lv.w[l] + 255) / 256, n, 256, (const uint8_t*)pc, T, dc, lv, l);
    launch(k_sof_lk, (SOF_MAX_CORNERS + 3) / 4, n, SOF_THREADS, (const uint8_t*)pp, (const uint8_t*)pc, T, (const short*)dp, lv, (const float*)pk, np_,
           status, (const SofState*)st, 30, 0.01 * 0.01, 1e-4);
    launch(k_sof_estimate, n, 1, SOF_THREADS, (const float*)pk, (const float*)np_, (const uint8_t*)status, vt, st, B.warp + s0 * 6, prm);
    launch(k_sof_eigen, pb, n, 256, (const uint8_t*)pc, T, eig, mask, d_dets, d_ndets, max_dets, det_stride, st, h, w, (float)prm.scale);
    launch(k_sof_candidates, pb, n, 256, (const float*)eig, (const uint8_t*)mask, cand, st, h, w);
    launch(k_sof_rank, pb, n, 256, (const float*)eig, (const int*)cand, nk, st, h, w);
    launch(k_sof_subpix, (SOF_MAX_CORNERS + 3) / 4, n, SOF_THREADS, (const uint8_t*)pc, T, nk, (const SofState*)st, h, w);
    launch(k_sof_finalize, n, 1, SOF_THREADS, pk, (const float*)nk, (const float*)vt, st);
    launch(k_sof_commit, (int)((T + 255) / 256), n, 256, (const uint8_t*)pc, pp, (const short*)dc, dp, T);
}

}  // namespace bm
