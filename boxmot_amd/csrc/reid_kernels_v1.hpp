// ReID path, generic per-layer HIP kernels (fp32 accumulate, NHWC activations).
//
// These are the straightforward device restatement of
//   BaseModelBackend.get_crops / get_features   boxmot/reid/backends/base_backend.py:148-207
//   OSNet.forward (eval)                         boxmot/reid/backbones/osnet.py:380-405
// one launch per layer.  They are the first correct device path and remain the
// on-device reference for the fused MFMA kernels (reid_fused.hpp).
#pragma once

#include <cstdint>

#include "kernel_macros.hpp"
#include "reid_layout.hpp"

namespace bm {

// ---------------------------------------------------------------------------
// crop -> cv2.resize(INTER_LINEAR, uint8 fixed point) -> BGR2RGB -> /255 -> (x-mean)/std
// One workgroup = one output row band of one crop; thread = output column.
// `lut` is the (3,256) fp32 normalisation table (exact by construction: the
// resized pixel is a uint8, so normalisation is a pure table lookup).
// ---------------------------------------------------------------------------
struct ResizeAxis { int s0, s1, a0, a1; };

__device__ inline ResizeAxis resize_axis_x(int d, int dst_n, int src_n) {
    // resize.cpp: fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx; clamp
    const double scale = (double)src_n / (double)dst_n;
    float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src_n - 1) { s = src_n - 1; f = 0.f; }
    ResizeAxis r;
    r.s0 = s;
    r.s1 = s + 1 < src_n ? s + 1 : src_n - 1;
    r.a0 = (int)rintf((1.f - f) * 2048.f);
    r.a1 = (int)rintf(f * 2048.f);
    return r;
}

__device__ inline ResizeAxis resize_axis_y(int d, int dst_n, int src_n) {
    // the y axis keeps its coefficients and clips the two row indices instead
    const double scale = (double)src_n / (double)dst_n;
    float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);
    const int s = (int)floorf(f);
    f = f - (float)s;
    ResizeAxis r;
    r.s0 = s < 0 ? 0 : (s > src_n - 1 ? src_n - 1 : s);
    r.s1 = s + 1 < 0 ? 0 : (s + 1 > src_n - 1 ? src_n - 1 : s + 1);
    r.a0 = (int)rintf((1.f - f) * 2048.f);
    r.a1 = (int)rintf(f * 2048.f);
    return r;
}

struct CropRect { int x1, y1, w, h; };   // w == 0 -> blank crop

__device__ inline CropRect crop_rect(const float* box, int W, int H) {
    // base_backend.py:172-179: round-half-even, clip, empty -> blank 256x128 zeros
    const int x1 = (int)rintf(box[0]), y1 = (int)rintf(box[1]);
    const int x2 = (int)rintf(box[2]), y2 = (int)rintf(box[3]);
    const int cx1 = x1 > 0 ? x1 : 0, cy1 = y1 > 0 ? y1 : 0;
    const int cx2 = x2 < W ? x2 : W, cy2 = y2 < H ? y2 : H;
    CropRect r;
    if (cx2 > cx1 && cy2 > cy1) { r.x1 = cx1; r.y1 = cy1; r.w = cx2 - cx1; r.h = cy2 - cy1; }
    else { r.x1 = r.y1 = 0; r.w = r.h = 0; }
    return r;
}

// One resized uint8 sample (BGR source channel c) at output (dy, dx).  `px(y, x, c)` fetches a pixel of the crop image (a slice of
// the frame for axis-aligned boxes; the rectified image of an oriented box is virtual, see obb_pixel).
template <class Fetch>
__device__ inline int resize_sample_f(const Fetch& px, const CropRect& r, const ResizeAxis& ax, const ResizeAxis& ay, int dy, int dx,
                                      int c, int out_w, int out_h) {
    if (r.w == 0) return 0;
    if (r.w == out_w && r.h == out_h) return px(dy, dx, c);
    if (r.w == 2 * out_w && r.h == 2 * out_h)     // exact 2x shrink -> INTER_AREA box filter
        return (px(2 * dy, 2 * dx, c) + px(2 * dy, 2 * dx + 1, c) + px(2 * dy + 1, 2 * dx, c) + px(2 * dy + 1, 2 * dx + 1, c) + 2) >> 2;
    const int S0 = px(ay.s0, ax.s0, c) * ax.a0 + px(ay.s0, ax.s1, c) * ax.a1;
    const int S1 = px(ay.s1, ax.s0, c) * ax.a0 + px(ay.s1, ax.s1, c) * ax.a1;
    return (((ay.a0 * (S0 >> 4)) >> 16) + ((ay.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
}

__device__ inline int resize_sample(const uint8_t* src, long row_stride, const CropRect& r,
                                    const ResizeAxis& ax, const ResizeAxis& ay, int dy, int dx, int c,
                                    int out_w, int out_h) {
    return resize_sample_f([=](int y, int x, int ch) { return (int)src[(long)y * row_stride + x * 3 + ch]; }, r, ax, ay, dy, dx, c, out_w, out_h);
}

// resize_pad (reid/core/preprocessing.py:21-45): aspect-preserving resize to (new_w, new_h) = int(dim * scale) with
// scale = min(tw / w, th / h), centred, constant ImageNet-mean border (BGR 104, 116, 124).
struct PadGeom { int new_w, new_h, top, left; };
__device__ inline PadGeom pad_geom(const CropRect& r, int out_w, int out_h) {
    const int w = r.w > 0 ? r.w : out_w, h = r.w > 0 ? r.h : out_h;      // an empty box is a blank out_h x out_w crop
    const double sx = (double)out_w / (double)w, sy = (double)out_h / (double)h;
    const double scale = sx < sy ? sx : sy;
    PadGeom g;
    g.new_w = (int)((double)w * scale);
    g.new_h = (int)((double)h * scale);
    g.top = (out_h - g.new_h) / 2;
    g.left = (out_w - g.new_w) / 2;
    return g;
}
// One uint8 sample (BGR source channel c) of the preprocessed crop at (dy, dx); pad == 0: plain resize.
__device__ inline int preprocess_sample(const uint8_t* src, long row_stride, const CropRect& r, const PadGeom& g, int pad,
                                        const ResizeAxis& ax_plain, int dy, int dx, int c, int out_w, int out_h) {
    if (!pad) {
        const ResizeAxis ay = resize_axis_y(dy, out_h, r.h > 0 ? r.h : 1);
        return resize_sample(src, row_stride, r, ax_plain, ay, dy, dx, c, out_w, out_h);
    }
    if (r.w == 0) return 0;
    const int yy = dy - g.top, xx = dx - g.left;
    if (yy < 0 || yy >= g.new_h || xx < 0 || xx >= g.new_w) return c == 0 ? 104 : (c == 1 ? 116 : 124);
    const ResizeAxis ax = resize_axis_x(xx, g.new_w, r.w), ay = resize_axis_y(yy, g.new_h, r.h);
    return resize_sample(src, row_stride, r, ax, ay, yy, xx, c, g.new_w, g.new_h);
}

// the same over a fetch functor (oriented boxes)
template <class Fetch>
__device__ inline int preprocess_sample_f(const Fetch& px, const CropRect& r, const PadGeom& g, int pad, const ResizeAxis& ax_plain, int dy,
                                          int dx, int c, int out_w, int out_h) {
    if (!pad) {
        const ResizeAxis ay = resize_axis_y(dy, out_h, r.h > 0 ? r.h : 1);
        return resize_sample_f(px, r, ax_plain, ay, dy, dx, c, out_w, out_h);
    }
    if (r.w == 0) return 0;
    const int yy = dy - g.top, xx = dx - g.left;
    if (yy < 0 || yy >= g.new_h || xx < 0 || xx >= g.new_w) return c == 0 ? 104 : (c == 1 ? 116 : 124);
    const ResizeAxis ax = resize_axis_x(xx, g.new_w, r.w), ay = resize_axis_y(yy, g.new_h, r.h);
    return resize_sample_f(px, r, ax, ay, yy, xx, c, g.new_w, g.new_h);
}

// ---------------------------------------------------------------------------
// Oriented boxes: BaseModelBackend._crop_obb (base_backend.py:91-117) = cv2.getRotationMatrix2D + cv2.warpAffine(INTER_LINEAR,
// BORDER_CONSTANT 0).  The rectified crop is never materialised: pixel (y, x) of it is computed on demand from the frame with
// warpAffine's arithmetic -- inverse 2x3 map `im` (inverted on the host in double precision, boxmot_hip.hip), fixed-point sampling
// position on the 1/32-pixel grid, integer bilinear weights (32 - fx)(32 - fy) * 32 ... summing to 2^15, taps outside the frame = 0
// -- and the ordinary resize / resize_pad runs on top of it.  geo = [out_w, out_h, im0 .. im5] (8 doubles per box).
// ---------------------------------------------------------------------------
__device__ inline int obb_pixel(const uint8_t* frame, int W, int H, const double* im, int y, int x, int c) {
    const long adelta = (long)rint(im[0] * (double)x * 1024.0), bdelta = (long)rint(im[3] * (double)x * 1024.0);
    const long X0 = (long)rint((im[1] * (double)y + im[2]) * 1024.0) + 16, Y0 = (long)rint((im[4] * (double)y + im[5]) * 1024.0) + 16;
    const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);            // saturate_cast<short>
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    auto tap = [&](long yy, long xx) { return (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (int)frame[(yy * W + xx) * 3 + c] : 0; };
    const int acc = tap(sy, sx) * ((32 - fx) * (32 - fy) * 32) + tap(sy, sx + 1) * (fx * (32 - fy) * 32) + tap(sy + 1, sx) * ((32 - fx) * fy * 32) +
                    tap(sy + 1, sx + 1) * (fx * fy * 32);
    const int v = (acc + (1 << 14)) >> 15;
    return v > 255 ? 255 : v;
}

// RGBX = false: out T [n][256][128][3] normalised (k_crop_resize's layout); RGBX = true: fp16 RGBX with a 3-pixel zero border,
// [n][262][136][4] (k_crop_resize_rgbx's layout: interior only, border and X channel stay zero from allocation)
template <typename T, bool RGBX>
__global__ void k_crop_resize_obb(const uint8_t* const* frames, const int* crop_stream, const double* geo, int W, int H, const float* lut,
                                  T* out, int rows_per_block, int pad) {
    const int i = blockIdx.x;
    const int dx = threadIdx.x;               // 0..127
    const uint8_t* frame = frames[crop_stream[i]];
    const double* gi = geo + (long)i * 8;
    CropRect r;
    r.x1 = 0; r.y1 = 0; r.w = (int)gi[0]; r.h = (int)gi[1];
    const double* im = gi + 2;
    const auto px = [=](int y, int x, int c) { return obb_pixel(frame, W, H, im, y, x, c); };
    const ResizeAxis ax = resize_axis_x(dx, REID_IN_W, r.w > 0 ? r.w : 1);
    const PadGeom g = pad_geom(r, REID_IN_W, REID_IN_H);
    const int y0 = blockIdx.y * rows_per_block;
    for (int dy = y0; dy < y0 + rows_per_block && dy < REID_IN_H; ++dy) {
        T v3[3];
        for (int c = 0; c < 3; ++c) {         // c = RGB output channel; source is BGR
            const int v = preprocess_sample_f(px, r, g, pad, ax, dy, dx, 2 - c, REID_IN_W, REID_IN_H);
            v3[c] = (T)lut[c * 256 + v];
        }
        if constexpr (RGBX) {
            T* o = out + (((long)i * 262 + dy + 3) * 136 + dx + 3) * 4;
            o[0] = v3[0]; o[1] = v3[1]; o[2] = v3[2]; o[3] = (T)0.f;
        } else {
            T* o = out + (((long)i * REID_IN_H + dy) * REID_IN_W + dx) * 3;
            o[0] = v3[0]; o[1] = v3[1]; o[2] = v3[2];
        }
    }
}

template <typename T>
__global__ void k_crop_resize(const uint8_t* const* frames, const int* crop_stream, const float* boxes,
                              int box_stride, int W, int H, const float* lut, T* out, int rows_per_block, int pad) {
    const int i = blockIdx.x;
    const int dx = threadIdx.x;               // 0..127
    const uint8_t* frame = frames[crop_stream[i]];
    const CropRect r = crop_rect(boxes + (long)i * box_stride, W, H);
    const long row_stride = (long)W * 3;
    const uint8_t* src = frame + (long)r.y1 * row_stride + r.x1 * 3;
    ResizeAxis ax = resize_axis_x(dx, REID_IN_W, r.w > 0 ? r.w : 1);
    const PadGeom g = pad_geom(r, REID_IN_W, REID_IN_H);
    const int y0 = blockIdx.y * rows_per_block;
    for (int dy = y0; dy < y0 + rows_per_block && dy < REID_IN_H; ++dy) {
        T* o = out + (((long)i * REID_IN_H + dy) * REID_IN_W + dx) * 3;
        for (int c = 0; c < 3; ++c) {         // c = RGB output channel; source is BGR
            const int v = preprocess_sample(src, row_stride, r, g, pad, ax, dy, dx, 2 - c, REID_IN_W, REID_IN_H);
            o[c] = (T)lut[c * 256 + v];
        }
    }
}

// ---------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3 (3 -> C0) + folded BN + ReLU   (osnet.py:294)
// in [N][256][128][3], w [C0][7][7][3], out [N][128][64][C0]
// ---------------------------------------------------------------------------
template <typename T, int C0>
__global__ void k_stem_conv(const T* in, const float* w, const float* b, T* out, long n_pix) {
    __shared__ float sw[C0 * 147 + C0];
    for (int e = threadIdx.x; e < C0 * 147; e += blockDim.x) sw[e] = w[e];
    for (int e = threadIdx.x; e < C0; e += blockDim.x) sw[C0 * 147 + e] = b[e];
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    const int ox = p % 64, oy = (p / 64) % 128;
    const long n = p / (64 * 128);
    float acc[C0];
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[c] = 0.f;
    const T* img = in + n * (REID_IN_H * REID_IN_W * 3);
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = oy * 2 - 3 + ky;
        if (iy < 0 || iy >= REID_IN_H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = ox * 2 - 3 + kx;
            if (ix < 0 || ix >= REID_IN_W) continue;
            const T* px = img + ((long)iy * REID_IN_W + ix) * 3;
            const float v0 = (float)px[0], v1 = (float)px[1], v2 = (float)px[2];
            const float* wk = sw + (ky * 7 + kx) * 3;
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                acc[c] += v0 * wk[c * 147] + v1 * wk[c * 147 + 1] + v2 * wk[c * 147 + 2];
            }
        }
    }
    T* o = out + p * C0;
#pragma unroll
    for (int c = 0; c < C0; ++c) {
        const float r = acc[c] + sw[C0 * 147 + c];
        o[c] = (T)(r > 0.f ? r : 0.f);
    }
}

// max pool 3x3 stride 2 pad 1 (osnet.py:295), NHWC
template <typename T>
__global__ void k_maxpool3x3s2(const T* in, T* out, int H, int W, int C, long total) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int OW = W / 2, OH = H / 2;
    const int c = e % C;
    const int ox = (e / C) % OW, oy = (e / ((long)C * OW)) % OH;
    const long n = e / ((long)C * OW * OH);
    float m = -3.0e38f;
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = oy * 2 + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = ox * 2 + dx;
            if (ix < 0 || ix >= W) continue;
            const float x = (float)in[((n * H + iy) * W + ix) * C + c];
            m = x > m ? x : m;
        }
    }
    out[e] = (T)m;
}

// 1x1 convolution (+ folded BN bias, optional residual add, optional ReLU), NHWC.
// thread = one pixel x CO_T output channels; blockIdx.y selects a chunk of
// `co_chunk` output channels whose weights are staged in LDS.
template <typename T, int CO_T>
__global__ void k_pointwise(const T* in, const float* w, const float* b, const T* res, T* out,
                            long n_pix, int cin, int cout, int relu, int co_chunk) {
    BM_DYNAMIC_LDS_T(float, s_w);             // [co_chunk][cin] + [co_chunk]
    const int co0 = blockIdx.y * co_chunk;
    const int co_n = (cout - co0) < co_chunk ? (cout - co0) : co_chunk;
    for (int e = threadIdx.x; e < co_n * cin; e += blockDim.x) s_w[e] = w[(long)co0 * cin + e];
    for (int e = threadIdx.x; e < co_n; e += blockDim.x) s_w[co_chunk * cin + e] = b ? b[co0 + e] : 0.f;
    __syncthreads();
    const int groups = co_n / CO_T;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long p = t / groups;
    const int g = t % groups;
    if (p >= n_pix) return;
    float acc[CO_T];
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;
    const T* x = in + p * cin;
    const float* wg = s_w + (long)g * CO_T * cin;
    for (int ci = 0; ci < cin; ++ci) {
        const float xv = (float)x[ci];
#pragma unroll
        for (int j = 0; j < CO_T; ++j) acc[j] += xv * wg[j * cin + ci];
    }
    T* o = out + p * cout + co0 + g * CO_T;
    const T* r = res ? res + p * cout + co0 + g * CO_T : nullptr;
#pragma unroll
    for (int j = 0; j < CO_T; ++j) {
        float v = acc[j] + s_w[co_chunk * cin + g * CO_T + j];
        if (r) v += (float)r[j];
        if (relu) v = v > 0.f ? v : 0.f;
        o[j] = (T)v;
    }
}

// depthwise 3x3 pad 1 + folded BN + ReLU (LightConv3x3 tail, osnet.py:141-155), NHWC
template <typename T>
__global__ void k_depthwise3x3(const T* in, const float* w, const float* b, T* out, int H, int W, int C, long total) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = e % C;
    const int x = (e / C) % W, y = (e / ((long)C * W)) % H;
    const long n = e / ((long)C * W * H);
    float acc = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = y + dy;
        if (iy < 0 || iy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = x + dx;
            if (ix < 0 || ix >= W) continue;
            acc += (float)in[((n * H + iy) * W + ix) * C + c] * w[c * 9 + (dy + 1) * 3 + (dx + 1)];
        }
    }
    acc += b[c];
    out[e] = (T)(acc > 0.f ? acc : 0.f);
}

// global average pool, NHWC: out[n][c] = mean_p in[n][p][c]; one workgroup per n
template <typename T>
__global__ void k_gap(const T* in, float* out, int P, int C) {
    __shared__ float s_part[1024];
    const long n = blockIdx.x;
    const int c = threadIdx.x % C;
    const int lanes_per_c = blockDim.x / C;
    const int q = threadIdx.x / C;
    float s = 0.f;
    if (q < lanes_per_c)
        for (int p = q; p < P; p += lanes_per_c) s += (float)in[(n * P + p) * C + c];
    s_part[threadIdx.x] = (q < lanes_per_c) ? s : 0.f;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < lanes_per_c; ++k) t += s_part[k * C + threadIdx.x];
        out[n * C + threadIdx.x] = t / (float)P;
    }
}

// ChannelGate (osnet.py:161-209) applied to one branch and accumulated:
//   acc[n][p][c] (+)= x[n][p][c] * sigmoid(fc2(relu(fc1(gap[n]))))[c]
template <typename T>
__global__ void k_gate_accumulate(const T* x, const float* gap, const float* fc1_w, const float* fc1_b,
                                  const float* fc2_w, const float* fc2_b, T* acc, int P, int C, int hid,
                                  int first, int pix_per_block) {
    __shared__ float s_h[16];
    __shared__ float s_g[512];
    const long n = blockIdx.x;
    if (threadIdx.x < hid) {
        float h = fc1_b[threadIdx.x];
        for (int c = 0; c < C; ++c) h += fc1_w[threadIdx.x * C + c] * gap[n * C + c];
        s_h[threadIdx.x] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float g = fc2_b[c];
        for (int k = 0; k < hid; ++k) g += fc2_w[c * hid + k] * s_h[k];
        s_g[c] = 1.f / (1.f + expf(-g));
    }
    __syncthreads();
    const long p0 = (long)blockIdx.y * pix_per_block;
    const long e0 = (n * P + p0) * C;
    const long cnt = (long)pix_per_block * C;
    for (long e = threadIdx.x; e < cnt; e += blockDim.x) {
        if (p0 + e / C >= P) break;
        const int c = (p0 * C + e) % C;
        const float v = (float)x[e0 + e] * s_g[c];
        acc[e0 + e] = (T)(first ? v : (float)acc[e0 + e] + v);
    }
}

// 2x2 average pool stride 2 (osnet.py:349), NHWC
template <typename T>
__global__ void k_avgpool2x2(const T* in, T* out, int H, int W, int C, long total) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int OW = W / 2, OH = H / 2;
    const int c = e % C;
    const int ox = (e / C) % OW, oy = (e / ((long)C * OW)) % OH;
    const long n = e / ((long)C * OW * OH);
    const T* p = in + ((n * H + oy * 2) * W + ox * 2) * C + c;
    const float s = (float)p[0] + (float)p[C] + (float)p[(long)W * C] + (float)p[(long)W * C + C];
    out[e] = (T)(s * 0.25f);
}

// head: GAP -> Linear + folded BatchNorm1d -> ReLU -> L2 normalise
// (osnet.py:393-396 + base_backend.py:206).  One workgroup (256 threads) per crop.
template <typename T>
__global__ void k_head(const T* in, const float* fc_w, const float* fc_b, float* out_base, const int* out_rows,
                       int P, int C, int F) {
    __shared__ float s_v[512];
    __shared__ float s_red[256];
    const long n = blockIdx.x;
    float* out = out_base + ((out_rows ? (long)out_rows[n] : n) - n) * F;   // row remap
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += (float)in[(n * P + p) * C + c];
        s_v[c] = s / (float)P;
    }
    __syncthreads();
    float sq = 0.f;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float a = fc_b[f];
        const float* wr = fc_w + (long)f * C;
        for (int c = 0; c < C; ++c) a += wr[c] * s_v[c];
        a = a > 0.f ? a : 0.f;
        out[n * F + f] = a;
        sq += a * a;
    }
    s_red[threadIdx.x] = sq;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    const float nrm = sqrtf(s_red[0]);
    for (int f = threadIdx.x; f < F; f += blockDim.x) out[n * F + f] = out[n * F + f] / nrm;
}

}  // namespace bm
