// Camera-motion estimation on the device: the ECC estimator the reference applies in StrongSORT on every frame and offers to
// the other trackers as cmc_method = "ecc" (boxmot/motion/cmc/ecc.py:45-96 with its defaults: MOTION_TRANSLATION, eps 1e-5,
// 100 iterations, scale 0.15, grayscale; BaseCMC.preprocess, base_cmc.py:30-61).
//
// The reference delegates the numerics to OpenCV (cv2.cvtColor, cv2.resize, cv2.findTransformECC).  What is built here is that
// algorithm (Evangelidis & Psarakis, PAMI 2008) in the structure of OpenCV's modules/video/src/ecc.cpp: fp32 images, central
// difference gradients with reflected borders, per iteration warpAffine(INTER_LINEAR | WARP_INVERSE_MAP) of the image and its
// gradients on warpAffine's 1/32-pixel coordinate grid plus a nearest-neighbour warp of the all-ones mask, masked mean / std in
// fp64, the zero-mean correlation rho, the 2 x 2 Gauss-Newton system of the translation Jacobian, the lambda of the illumination
// model, map[:, 2] += deltaP, stop when |rho - last_rho| < eps; OpenCV's two StsNoConv exits return the identity like
// ecc.py:67-76.  Every cast (fp32 storage of the Hessian / projections / update, fp64 sums) is where ecc.cpp has it.
//
//   k_ecc_preprocess   BGR frame -> BGR2GRAY (14-bit fixed point) -> INTER_LINEAR resize by `scale` (11-bit fixed point, the
//                      integer pipeline of the crop kernels) -> fp32 image; one thread per output pixel
//   k_ecc_gradients    [-0.5 0 0.5] central differences with reflected borders
//   k_ecc_solve        one workgroup (1024 threads) per stream, the whole Gauss-Newton iteration inside: per iteration three
//                      passes over the pixels (warp + moments | zero-mean correlation and projections | error projection),
//                      each ending in a workgroup reduction of fp64 partial sums; the 2 x 2 system is solved redundantly per
//                      thread.  Warped images stay in an L2-resident scratch (3 x pixels fp32 per stream).
// Data: per stream two image buffers (previous / current, swapped every frame), gradients and scratch, all [h][w] fp32 with
// (w, h) = (round(cols * scale), round(rows * scale)): 288 x 162 for 1080p.
#pragma once

#include <stdint.h>

#include "kernel_macros.hpp"
#include "reid_kernels_v1.hpp"       // ResizeAxis: cv2.resize's index / coefficient pair

namespace bm {

constexpr int ECC_THREADS = 1024;
constexpr int ECC_AB_BITS = 10, ECC_INTER_BITS = 5;

// cv2.resize(fx = fy = scale): the coordinate scale is 1 / fx whatever the rounded output size is (resize.cpp); otherwise the
// tables of resize_axis_x / resize_axis_y (reid_kernels_v1.hpp)
__device__ inline ResizeAxis ecc_axis(int d, int src_n, double inv_scale, bool clamp_coef) {
    float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, inv_scale), -0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    ResizeAxis r;
    if (clamp_coef) {                     // x axis: clamp the index and zero the fraction
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src_n - 1) { s = src_n - 1; f = 0.f; }
        r.s0 = s;
        r.s1 = s + 1 < src_n ? s + 1 : src_n - 1;
    } else {                              // y axis: keep the coefficients, clip the two row indices
        r.s0 = s < 0 ? 0 : (s > src_n - 1 ? src_n - 1 : s);
        r.s1 = s + 1 < 0 ? 0 : (s + 1 > src_n - 1 ? src_n - 1 : s + 1);
    }
    r.a0 = (int)rintf((1.f - f) * 2048.f);
    r.a1 = (int)rintf(f * 2048.f);
    return r;
}

__device__ inline int ecc_gray(const uint8_t* p) {      // COLOR_BGR2GRAY: (B 1868 + G 9617 + R 4899 + 2^13) >> 14
    return (p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14;
}

// frames: one pointer per stream ([rows][cols][3] uint8 BGR); out: fp32 image of stream s at out + s * out_stride
__global__ void __launch_bounds__(256) k_ecc_preprocess(const uint8_t* const* __restrict__ frames, float* __restrict__ out, long out_stride,
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
    out[(long)s * out_stride + e] = (float)v;
}

__global__ void __launch_bounds__(256) k_ecc_gradients(const float* __restrict__ img, long img_stride, float* __restrict__ gx,
                                                       float* __restrict__ gy, int h, int w) {
    const long base = (long)blockIdx.y * h * w;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= h * w) return;
    const int y = e / w, x = e - y * w;
    const float* im = img + (long)blockIdx.y * img_stride;
    const int xl = x > 0 ? x - 1 : (w > 1 ? 1 : 0), xr = x < w - 1 ? x + 1 : (w > 1 ? w - 2 : 0);     // BORDER_REFLECT_101
    const int yu = y > 0 ? y - 1 : (h > 1 ? 1 : 0), yd = y < h - 1 ? y + 1 : (h > 1 ? h - 2 : 0);
    gx[base + e] = 0.5f * im[y * w + xr] - 0.5f * im[y * w + xl];
    gy[base + e] = 0.5f * im[yd * w + x] - 0.5f * im[yu * w + x];
}

// workgroup sum of K fp64 values per thread; the result is returned to every thread.  red: LDS [K][16]
template <int K>
__device__ inline void ecc_reduce(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = v[k];
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) red[k * 16 + wave] = s;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double s = 0.0;
        for (int i = 0; i < nw; ++i) s += red[k * 16 + i];       // fixed order: every thread computes the same bits
        v[k] = s;
    }
    __syncthreads();
}


// warpAffine(INTER_LINEAR | WARP_INVERSE_MAP, BORDER_CONSTANT 0) sample of a translation: integer grid position (X, Y) in
// 1/32-pixel units
__device__ inline float ecc_bilinear(const float* __restrict__ im, int X, int Y, int w, int h) {
    const int ix = X >> ECC_INTER_BITS, iy = Y >> ECC_INTER_BITS;
    const float fx = (float)(X & 31) / 32.0f, fy = (float)(Y & 31) / 32.0f;
    auto tap = [&](int yy, int xx) { return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? im[yy * w + xx] : 0.0f; };
    const float w00 = (1.0f - fx) * (1.0f - fy), w01 = fx * (1.0f - fy), w10 = (1.0f - fx) * fy, w11 = fx * fy;
    float o = tap(iy, ix) * w00;
    o = o + tap(iy, ix + 1) * w01;
    o = o + tap(iy + 1, ix) * w10;
    o = o + tap(iy + 1, ix + 1) * w11;
    return o;
}

// tmpl / img: fp32 images of stream s at + s * img_stride (tmpl = previous frame, img = current); gx / gy: fp32 [S][h][w];
// scratch: fp32 [S][3][h][w];
// out_warp: fp64 [S][6] row-major 2 x 3 in FULL-RESOLUTION pixels (translation divided by `scale`, ecc.py:80-83);
// out_info: int [S][2] = (status: 1 estimated, 0 identity because the iteration hit one of OpenCV's StsNoConv exits; iterations)
__global__ void __launch_bounds__(ECC_THREADS) k_ecc_solve(const float* __restrict__ tmpl_all, const float* __restrict__ img_all, long img_stride,
                                                           const float* __restrict__ gx_all, const float* __restrict__ gy_all,
                                                           float* __restrict__ scratch_all, double* __restrict__ out_warp,
                                                           int* __restrict__ out_info, int h, int w, double eps, int max_iter,
                                                           float scale) {
    __shared__ double red[8 * 16];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int P = h * w;
    const float* tmpl = tmpl_all + (long)s * img_stride;
    const float* img = img_all + (long)s * img_stride;
    const float* gx = gx_all + (long)s * P;
    const float* gy = gy_all + (long)s * P;
    float* siw = scratch_all + (long)s * 3 * P;
    float* sgx = siw + P;
    float* sgy = sgx + P;
    float tx = 0.0f, ty = 0.0f;                      // the warp map is CV_32F
    double rho = -1.0, last_rho = -eps;
    int it = 0, status = 1;
    for (it = 1; it <= max_iter; ++it) {
        if (fabs(rho - last_rho) < eps) { --it; break; }
        // integer coordinate grids of warpAffine: X = (rint(tx 2^10) + round_delta + x 2^10) >> shift
        const long X0l = (long)rint((double)tx * 1024.0) + 16, Y0l = (long)rint((double)ty * 1024.0) + 16;
        const long X0n = (long)rint((double)tx * 1024.0) + 512, Y0n = (long)rint((double)ty * 1024.0) + 512;
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // n, sum iw, sum iw^2, sum t, sum t^2, H00, H01, H11
        for (int e = tid; e < P; e += ECC_THREADS) {
            const int y = e / w, x = e - y * w;
            const int X = (int)((X0l + ((long)x << 10)) >> 5), Y = (int)((Y0l + ((long)y << 10)) >> 5);
            const float iw = ecc_bilinear(img, X, Y, w, h), gxw = ecc_bilinear(gx, X, Y, w, h), gyw = ecc_bilinear(gy, X, Y, w, h);
            siw[e] = iw; sgx[e] = gxw; sgy[e] = gyw;
            const long Xn = (X0n + ((long)x << 10)) >> 10, Yn = (Y0n + ((long)y << 10)) >> 10;
            if (Xn >= 0 && Xn < w && Yn >= 0 && Yn < h) {
                const double di = (double)iw, dt = (double)tmpl[e];
                a[0] += 1.0; a[1] += di; a[2] += di * di; a[3] += dt; a[4] += dt * dt;
            }
            a[5] += (double)gxw * (double)gxw; a[6] += (double)gxw * (double)gyw; a[7] += (double)gyw * (double)gyw;
        }
        ecc_reduce<8>(a, red);
        const double n = a[0];
        const double im_mean = n > 0 ? a[1] / n : 0.0, tm_mean = n > 0 ? a[3] / n : 0.0;
        double im_var = n > 0 ? a[2] / n - im_mean * im_mean : 0.0, tm_var = n > 0 ? a[4] / n - tm_mean * tm_mean : 0.0;
        im_var = im_var > 0 ? im_var : 0.0; tm_var = tm_var > 0 ? tm_var : 0.0;
        const double im_std = sqrt(im_var), tm_std = sqrt(tm_var);
        const double tmp_norm = sqrt(n * tm_std * tm_std), img_norm = sqrt(n * im_std * im_std);
        const float H00 = (float)a[5], H01 = (float)a[6], H11 = (float)a[7];
        const double det = (double)H00 * (double)H11 - (double)H01 * (double)H01;
        float i00 = 0.f, i01 = 0.f, i10 = 0.f, i11 = 0.f;
        if (det != 0.0) { i00 = (float)((double)H11 / det); i01 = (float)(-(double)H01 / det); i10 = (float)(-(double)H01 / det); i11 = (float)((double)H00 / det); }
        // pass B: zero-mean images, correlation, projections onto the Jacobian [gxw | gyw]
        double b[5] = {0, 0, 0, 0, 0};               // corr, ip0, ip1, tp0, tp1
        for (int e = tid; e < P; e += ECC_THREADS) {
            const int y = e / w, x = e - y * w;
            const long Xn = (X0n + ((long)x << 10)) >> 10, Yn = (Y0n + ((long)y << 10)) >> 10;
            const bool m = Xn >= 0 && Xn < w && Yn >= 0 && Yn < h;
            const float iw = siw[e];
            const float iwz = m ? (float)((double)iw - im_mean) : iw;
            const float tz = m ? (float)((double)tmpl[e] - tm_mean) : 0.0f;
            const double gxw = (double)sgx[e], gyw = (double)sgy[e];
            b[0] += (double)tz * (double)iwz;
            b[1] += gxw * (double)iwz; b[2] += gyw * (double)iwz;
            b[3] += gxw * (double)tz; b[4] += gyw * (double)tz;
        }
        ecc_reduce<5>(b, red);
        const double corr = b[0];
        last_rho = rho;
        const double denom = img_norm * tmp_norm;
        rho = denom != 0.0 ? corr / denom : nan("");
        if (rho != rho) { status = 0; break; }                       // StsNoConv: NaN
        const float ip0 = (float)b[1], ip1 = (float)b[2], tp0 = (float)b[3], tp1 = (float)b[4];
        const float iph0 = (float)((double)i00 * (double)ip0 + (double)i01 * (double)ip1);
        const float iph1 = (float)((double)i10 * (double)ip0 + (double)i11 * (double)ip1);
        const double lambda_n = img_norm * img_norm - ((double)ip0 * (double)iph0 + (double)ip1 * (double)iph1);
        const double lambda_d = corr - ((double)tp0 * (double)iph0 + (double)tp1 * (double)iph1);
        if (lambda_d <= 0.0) { status = 0; break; }                  // StsNoConv: the correlation is going to be minimized
        const float lam = (float)(lambda_n / lambda_d);
        // pass C: error image projected onto the Jacobian
        double c[2] = {0, 0};
        for (int e = tid; e < P; e += ECC_THREADS) {
            const int y = e / w, x = e - y * w;
            const long Xn = (X0n + ((long)x << 10)) >> 10, Yn = (Y0n + ((long)y << 10)) >> 10;
            const bool m = Xn >= 0 && Xn < w && Yn >= 0 && Yn < h;
            const float iw = siw[e];
            const float iwz = m ? (float)((double)iw - im_mean) : iw;
            const float tz = m ? (float)((double)tmpl[e] - tm_mean) : 0.0f;
            const float err = lam * tz - iwz;
            c[0] += (double)sgx[e] * (double)err; c[1] += (double)sgy[e] * (double)err;
        }
        ecc_reduce<2>(c, red);
        const float ep0 = (float)c[0], ep1 = (float)c[1];
        const float dp0 = (float)((double)i00 * (double)ep0 + (double)i01 * (double)ep1);
        const float dp1 = (float)((double)i10 * (double)ep0 + (double)i11 * (double)ep1);
        tx = tx + dp0;
        ty = ty + dp1;
    }
    if (it > max_iter) it = max_iter;
    if (tid == 0) {
        double* o = out_warp + (long)s * 6;
        o[0] = 1.0; o[1] = 0.0; o[3] = 0.0; o[4] = 1.0;
        if (status) {
            const float fx = scale < 1.0f ? tx / scale : tx, fy = scale < 1.0f ? ty / scale : ty;     // fp32 division, ecc.py:80-83
            o[2] = (double)fx; o[5] = (double)fy;
        } else { o[2] = 0.0; o[5] = 0.0; }
        out_info[s * 2] = status; out_info[s * 2 + 1] = it;
    }
}

}  // namespace bm
