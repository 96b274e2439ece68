// Host-side ReID engine: owns the folded OSNet weights and activation buffers on
// the device and sequences the kernels for a batch of crops.
//   mode 0: per-layer fp32 kernels (reid_kernels_v1.hpp) -- first correct path
//   mode 1: fused fp16 MFMA kernels (reid_fused.hpp)
//   mode 2: fused fp32-grade kernels (reid_hp.hpp): the fused structure on fp16 (hi, lo) operand pairs, fp32 everywhere else;
//           for the wide OSNets (osnet_x1_0) the fp32-grade family of osnet_wide_hp.hpp (chain-fused LightConvs, (hi, lo) GEMMs)
// Reference path: BaseModelBackend.get_features, base_backend.py:197-207.
#pragma once
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "reid_layout.hpp"
#include "reid_kernels_v1.hpp"
#include "reid_fused.hpp"
#include "reid_hp.hpp"
#include "clip_engine.hpp"
#include "osnet_wide.hpp"
#include "osnet_wide_hp.hpp"

#ifndef BM_STAGE1_HANDOVER
#define BM_STAGE1_HANDOVER 0
#endif
#ifndef BM_HEAD_PER_CROP
#define BM_HEAD_PER_CROP 0          // 1: the round-1 head (one workgroup of two waves per crop) instead of k_head_batched
#endif

// default of BOXMOT_HIP_REID_PERSIST (the fp32-grade x0.25 block kernels as persistent workgroups; A/B: profiles/r5_hp_persist_ab.txt)
#ifndef BM_HP_PERSIST_DEFAULT
#define BM_HP_PERSIST_DEFAULT 0
#endif

namespace bm {

inline void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define BM_HIP(x) ::bm::hip_check((x), #x)

template <typename T>
inline T* dev_alloc(size_t n, std::vector<void*>& owned) {
    void* p = nullptr;
    BM_HIP(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
    owned.push_back(p);
    return static_cast<T*>(p);
}

inline std::vector<float> read_blob_file(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) throw std::runtime_error(std::string("cannot open ReID weight blob: ") + path);
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<float> v((size_t)bytes / 4);
    const size_t got = std::fread(v.data(), 4, v.size(), f);
    std::fclose(f);
    if (got != v.size()) throw std::runtime_error("short read on ReID weight blob");
    return v;
}

// crops processed per pass of the engine (bounds the activation buffers)
inline int reid_chunk_for(long max_total) { return (int)(max_total < 4096 ? max_total : 4096); }

class ReidEngine {
public:
    // max_crops: crops per pass of the per-layer path; fused_cap: crops the fused path can take in one pass
    ReidEngine(const float* blob, long n_floats, int max_crops, int fused_cap = 0)
        : max_crops_(max_crops), fused_cap_(fused_cap > max_crops ? fused_cap : max_crops) {
        if (n_floats < REID_HEADER_INTS) throw std::runtime_error("ReID blob too small");
        const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
        // normalisation table (x/255 - mean)/std in fp32, exactly as base_backend.py:189-193 evaluates it; "clip" models use
        // mean = std = 0.5 (base_backend.py:50-54)
        const bool is_clip = hdr[0] == CLIP_MAGIC;
        float lut[3 * 256];
        const float mean_i[3] = {0.485f, 0.456f, 0.406f}, std_i[3] = {0.229f, 0.224f, 0.225f};
        for (int c = 0; c < 3; ++c)
            for (int v = 0; v < 256; ++v) {
                volatile float a = (float)v / 255.0f;
                volatile float b = a - (is_clip ? 0.5f : mean_i[c]);
                lut[c * 256 + v] = b / (is_clip ? 0.5f : std_i[c]);
            }
        d_lut_ = dev_alloc<float>(3 * 256, owned_);
        BM_HIP(hipMemcpy(d_lut_, lut, sizeof(lut), hipMemcpyHostToDevice));
        if (is_clip) {
            // CLIP-ReID (ViT-B/16): one kernel family (fp16 GEMM operands, fp32 residual stream); crops through the standalone
            // crop kernel, then clip_engine.hpp
            clip_.reset(new ClipNet(blob, n_floats, max_crops, owned_));
            if (clip_->in_h != REID_IN_H || clip_->in_w != REID_IN_W)
                throw std::runtime_error("CLIP-ReID: the crop kernels produce 256 x 128 inputs only");
            crops_ = dev_alloc<float>((size_t)max_crops * REID_IN_H * REID_IN_W * 3, owned_);
            L_.feat = clip_->feature_dim();
        } else {
            if (hdr[0] != REID_MAGIC) throw std::runtime_error("ReID blob: bad magic (expected OSN1 or CLP1)");
            const int ch[4] = {hdr[1], hdr[2], hdr[3], hdr[4]};
            L_ = make_osnet_layout(ch, hdr[5]);
            if (hdr[6] != (int32_t)L_.total || n_floats != REID_HEADER_INTS + L_.total)
                throw std::runtime_error("ReID blob: size does not match the declared architecture");
            if (ch[0] != 16 && ch[0] != 32 && ch[0] != 48 && ch[0] != 64) throw std::runtime_error("ReID: unsupported stem width");
            d_w_ = dev_alloc<float>((size_t)L_.total, owned_);
            BM_HIP(hipMemcpy(d_w_, blob + REID_HEADER_INTS, (size_t)L_.total * 4, hipMemcpyHostToDevice));
            h_w_.assign(blob + REID_HEADER_INTS, blob + REID_HEADER_INTS + L_.total);
            alloc_buffers();
            if (ch[0] == 16 && ch[1] == 64 && ch[2] == 96 && ch[3] == 128 && L_.feat == 512) prepare_fused();
            else {
                // the matrix-pipe families run the network itself (osnet_x1_0) or its zero-padded copy (osnet_x0_5, osnet_x0_75:
                // middle widths 48 / 72, reid_layout.hpp: osnet_pad_weights); the per-layer fp32 kernels keep the original
                Lw_ = L_; hw_w_ = h_w_.data(); dw_w_ = d_w_;
                int cp[4];
                if (!WideOsnet::supports(L_) && osnet_padded_channels(L_, cp)) {
                    Lw_ = osnet_padded_layout(L_, cp);
                    h_wpad_ = osnet_pad_weights(h_w_.data(), L_, Lw_);
                    float* d = dev_alloc<float>((size_t)Lw_.total, owned_);
                    BM_HIP(hipMemcpy(d, h_wpad_.data(), (size_t)Lw_.total * 4, hipMemcpyHostToDevice));
                    hw_w_ = h_wpad_.data(); dw_w_ = d;
                }
                if (WideOsnet::supports(Lw_))       // layer-per-launch fp16 MFMA kernels (osnet_wide.hpp)
                    wide_.reset(new WideOsnet(hw_w_, Lw_, dw_w_, max_crops_ < 1024 ? max_crops_ : 1024, owned_));
            }
        }
        BM_HIP(hipEventCreate(&ev_[0]));
        BM_HIP(hipEventCreate(&ev_[1]));
        BM_HIP(hipEventCreate(&ev_[2]));
    }
    ~ReidEngine() {
        for (void* p : owned_) (void)hipFree(p);
        for (auto& e : ev_) (void)hipEventDestroy(e);
        for (auto& e : all_events_) (void)hipEventDestroy(e);
    }
    int feature_dim() const { return L_.feat; }
    int max_crops() const { return max_crops_; }
    void set_mode(int m) {
        if (m != 0 && m != 1 && m != 2)
            throw std::runtime_error("ReID mode must be 0 (per-layer fp32), 1 (fused fp16 MFMA) or 2 (fused fp32-grade)");
        if (clip_) return;                  // CLIP-ReID has one kernel family; the mode switch is OSNet's
        if (m == 1 && !fused_ready_ && !wide_)
            throw std::runtime_error("fp16 MFMA ReID kernels exist for OSNet-x0.25 (fused) and for widths that pad to multiples of 32 (osnet_x0_5, osnet_x0_75, osnet_x1_0)");
        if (m == 2) {
            if (fused_ready_) { if (!hp_ready_) prepare_hp(); }
            else if (hw_w_ && WideOsnetHP::supports(Lw_)) {
                if (!wide_hp_) wide_hp_.reset(new WideOsnetHP(hw_w_, Lw_, dw_w_, max_crops_ < 1024 ? max_crops_ : 1024, owned_));
            } else
                throw std::runtime_error("the fp32-grade ReID kernels (mode 2) exist for OSNet-x0.25 (fused) and for widths that pad to a stem of 32 / 64 and middle widths of 32 / 64 / 96 / 128 (osnet_x0_5, osnet_x0_75, osnet_x1_0)");
        }
        mode_ = m;
    }
    int mode() const { return mode_; }
    // the crop count may stay on the device (run_counted): only the fused x0.25 kernels take it
    bool counted_ok() const { return (mode_ == 1 || mode_ == 2) && fused_ready_; }
    void set_fuse_stem(bool on) { fuse_stem_ = on; }     // A/B switch: fused crop+stem kernel vs resize kernel + stem kernel
    // 0 = "resize" (default), 1 = "resize_pad" (reid/core/preprocessing.py:12-45); resize_pad runs the separate crop kernel
    void set_preprocess(int pad) { pad_ = pad; }
    // oriented boxes for the next preprocess / run: device array [n][8] doubles (out_w, out_h, inverse 2x3 map) or nullptr (axis-aligned).
    // In a chunked run the caller passes the chunk's slice (run() processes boxes from index 0 of what it is given).
    void set_obb_geometry(const double* d_geo) { obb_geo_ = d_geo; }
    int preprocess_mode() const { return pad_; }
    const OsnetLayout& layout() const { return L_; }
    float* crops_buffer() { return crops_; }

    // crops only (normalised NHWC fp32) for `n` boxes
    void preprocess(const uint8_t* const* d_frames, const int* d_crop_stream, const float* d_boxes,
                    int box_stride, int n, int W, int H, hipStream_t st) {
        const bool hp = mode_ == 2 && !force_fp32_crops_;
        if (hp) {       // (hi, lo) fp16 RGBX planes for k_stem_hp / k_wide_stem_hp
            const bool whp = !fused_ready_ && wide_hp_;
            if (n > (whp ? wide_hp_->max_crops() : fused_cap_)) throw std::runtime_error("ReID: crop batch exceeds the engine capacity");
            if (n == 0) return;
            if (obb_geo_) throw std::runtime_error("ReID mode 2 (fused fp32-grade kernels): oriented-box crops run in modes 0 / 1");
            hipLaunchKernelGGL(k_crop_resize_rgbx_hl, dim3(n, REID_IN_H / 16), dim3(REID_IN_W), 0, st, d_frames, d_crop_stream, d_boxes,
                               box_stride, W, H, d_lut_, whp ? wide_hp_->crops_h() : crops_h_, whp ? wide_hp_->crops_l() : crops_l_, 16,
                               whp ? static_cast<const int*>(nullptr) : d_count_, pad_);
            return;
        }
        const bool fused = mode_ == 1 && fused_ready_ && !force_fp32_crops_;
        const bool wide = mode_ == 1 && wide_ && !force_fp32_crops_;
        if (n > (fused ? fused_cap_ : (wide ? wide_->max_crops() : max_crops_))) throw std::runtime_error("ReID: crop batch exceeds the engine capacity");
        if (n == 0) return;
        const int rows_per_block = 16;
        if (obb_geo_) {         // oriented boxes: the rectified crop is sampled on demand (k_crop_resize_obb), same output layouts
            const dim3 grid(n, REID_IN_H / rows_per_block), block(REID_IN_W);
            if (wide) hipLaunchKernelGGL((k_crop_resize_obb<_Float16, true>), grid, block, 0, st, d_frames, d_crop_stream, obb_geo_, W, H, d_lut_, wide_->crops_buffer(), rows_per_block, pad_);
            else if (fused) hipLaunchKernelGGL((k_crop_resize_obb<_Float16, true>), grid, block, 0, st, d_frames, d_crop_stream, obb_geo_, W, H, d_lut_, crops_h_, rows_per_block, pad_);
            else hipLaunchKernelGGL((k_crop_resize_obb<float, false>), grid, block, 0, st, d_frames, d_crop_stream, obb_geo_, W, H, d_lut_, crops_, rows_per_block, pad_);
            return;
        }
        if (wide)
            hipLaunchKernelGGL(k_crop_resize_rgbx, dim3(n, REID_IN_H / rows_per_block), dim3(REID_IN_W), 0, st,
                               d_frames, d_crop_stream, d_boxes, box_stride, W, H, d_lut_, wide_->crops_buffer(), rows_per_block,
                               static_cast<const int*>(nullptr), pad_);
        else if (fused)
            hipLaunchKernelGGL(k_crop_resize_rgbx, dim3(n, REID_IN_H / rows_per_block), dim3(REID_IN_W), 0, st,
                               d_frames, d_crop_stream, d_boxes, box_stride, W, H, d_lut_, crops_h_, rows_per_block, d_count_, pad_);
        else
            hipLaunchKernelGGL(k_crop_resize<float>, dim3(n, REID_IN_H / rows_per_block), dim3(REID_IN_W), 0, st,
                               d_frames, d_crop_stream, d_boxes, box_stride, W, H, d_lut_, crops_, rows_per_block, pad_);
    }
    // the fp32 NHWC crop tensor is also the public "preprocess" result: force it regardless of the mode
    void preprocess_fp32(const uint8_t* const* d_frames, const int* d_crop_stream, const float* d_boxes,
                         int box_stride, int n, int W, int H, hipStream_t st) {
        force_fp32_crops_ = true;
        preprocess(d_frames, d_crop_stream, d_boxes, box_stride, n, W, H, st);
        force_fp32_crops_ = false;
    }

    // full path; out row of crop i is out_rows ? out_rows[i] : i, rows of `feat` floats.
    // Crops are processed in chunks of max_crops_ (activation buffers are sized for one chunk).
    void run(const uint8_t* const* d_frames, const int* d_crop_stream, const float* d_boxes, int box_stride,
             int n, int W, int H, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (n == 0) return;
        BM_HIP(hipEventRecord(ev_[0], st));
        const bool wide = mode_ == 1 && wide_;
        const bool wide_hp = mode_ == 2 && !fused_ready_ && wide_hp_;
        const int step = wide ? wide_->max_crops() : (wide_hp ? wide_hp_->max_crops() : (mode_ >= 1 ? fused_cap_ : max_crops_));
        const double* geo_all = obb_geo_;
        for (int i0 = 0; i0 < n; i0 += step) {
            const int m = (n - i0) < step ? (n - i0) : step;
            if (geo_all) obb_geo_ = geo_all + (long)i0 * 8;
            const bool fuse_stem = mode_ >= 1 && fused_ready_ && fuse_stem_ && !pad_ && !obb_geo_;
            if (!fuse_stem) preprocess(d_frames, d_crop_stream + i0, d_boxes + (long)i0 * box_stride, box_stride, m, W, H, st);
            if (i0 == 0) BM_HIP(hipEventRecord(ev_[1], st));
            hipEvent_t a = take_event(), b = take_event();
            BM_HIP(hipEventRecord(a, st));
            float* o = d_out_rows ? d_out : d_out + (long)i0 * L_.feat;
            const int* orow = d_out_rows ? d_out_rows + i0 : nullptr;
            if (clip_) clip_->forward(crops_, m, o, orow, st);
            else if (wide) wide_->forward(m, o, orow, st);
            else if (wide_hp) wide_hp_->forward(m, o, orow, st);
            else if (mode_ >= 1) {
                const FrameArgs fa{d_frames, d_crop_stream + i0, d_boxes + (long)i0 * box_stride, box_stride, W, H};
                if (mode_ == 2) forward_hp(m, fuse_stem ? &fa : nullptr, o, orow, st);
                else forward_fused(m, fuse_stem ? &fa : nullptr, o, orow, st);
            } else forward_v1(m, o, orow, st);
            BM_HIP(hipEventRecord(b, st));
            if (pending_.size() < 4096) pending_.emplace_back(a, b);
            else { free_events_.push_back(a); free_events_.push_back(b); }
        }
        obb_geo_ = geo_all;
        BM_HIP(hipEventRecord(ev_[2], st));
        timed_ = true;
    }
    // Fused path with the crop count resident on the device: launches cover n_max crops, workgroups
    // beyond *d_count exit immediately -- no host round trip between the crop list and the ReID kernels.
    void run_counted(const uint8_t* const* d_frames, const int* d_crop_stream, const float* d_boxes, int box_stride,
                     const int* d_count, int n_max, int W, int H, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (!counted_ok()) throw std::runtime_error("ReID: run_counted needs the fused x0.25 kernels (mode 1 or 2)");
        if (n_max > fused_cap_) throw std::runtime_error("ReID: crop batch exceeds the engine capacity");
        if (n_max == 0) return;
        d_count_ = d_count;
        BM_HIP(hipEventRecord(ev_[0], st));
        const bool fuse_stem = fuse_stem_ && !pad_;
        if (!fuse_stem) preprocess(d_frames, d_crop_stream, d_boxes, box_stride, n_max, W, H, st);
        BM_HIP(hipEventRecord(ev_[1], st));
        hipEvent_t a = take_event(), b = take_event();
        BM_HIP(hipEventRecord(a, st));
        const FrameArgs fa{d_frames, d_crop_stream, d_boxes, box_stride, W, H};
        if (mode_ == 2) forward_hp(n_max, fuse_stem ? &fa : nullptr, d_out, d_out_rows, st);
        else forward_fused(n_max, fuse_stem ? &fa : nullptr, d_out, d_out_rows, st);
        BM_HIP(hipEventRecord(b, st));
        if (pending_.size() < 4096) pending_.emplace_back(a, b);
        else { free_events_.push_back(a); free_events_.push_back(b); }
        BM_HIP(hipEventRecord(ev_[2], st));
        d_count_ = nullptr;
        timed_ = true;
    }
    // Accumulated device time of the forward region (events ev_[1]..ev_[2]) over the runs since the
    // last drain.  Every run() keeps its own event pair (ring of 64) so a timed loop needs no sync.
    void drain_kernel_timing(double& ms, int& launches) {
        ms = 0; launches = 0;
        for (auto& p : pending_) {
            float t = 0;
            if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) {
                ms += t; ++launches;
            }
            free_events_.push_back(p.first); free_events_.push_back(p.second);
        }
        pending_.clear();
    }
    // (pre, process) device milliseconds of the last run(); call after a stream sync
    void last_times(double& pre, double& proc) {
        pre = proc = 0.0;
        if (!timed_) return;
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, ev_[0], ev_[1]) == hipSuccess) pre = a;
        if (hipEventElapsedTime(&b, ev_[1], ev_[2]) == hipSuccess) proc = b;
    }

private:
    hipEvent_t take_event() {
        if (!free_events_.empty()) { hipEvent_t e = free_events_.back(); free_events_.pop_back(); return e; }
        hipEvent_t e;
        BM_HIP(hipEventCreate(&e));
        all_events_.push_back(e);
        return e;
    }
    template <int CO_T>
    void pointwise(const float* in, long w, long b, const float* res, float* out, long n_pix, int cin, int cout,
                   int relu, hipStream_t st) {
        int co_chunk = (8192 / cin) / CO_T * CO_T;          // <= 32 KB of weights per workgroup
        if (co_chunk > cout) co_chunk = cout;
        if (co_chunk < CO_T) co_chunk = CO_T;
        const int n_chunks = (cout + co_chunk - 1) / co_chunk;
        const long threads = n_pix * (co_chunk / CO_T);
        const int bs = 256;
        const size_t smem = ((size_t)co_chunk * cin + co_chunk) * sizeof(float);
        hipLaunchKernelGGL((k_pointwise<float, CO_T>), dim3((unsigned)((threads + bs - 1) / bs), n_chunks), dim3(bs),
                           smem, st, in, d_w_ + w, b >= 0 ? d_w_ + b : nullptr, res, out, n_pix, cin, cout, relu,
                           co_chunk);
    }
    void pw(const float* in, long w, long b, const float* res, float* out, long n_pix, int cin, int cout, int relu,
            hipStream_t st) {
        if (cout % 16 == 0) pointwise<16>(in, w, b, res, out, n_pix, cin, cout, relu, st);
        else pointwise<8>(in, w, b, res, out, n_pix, cin, cout, relu, st);
    }
    void osblock(const BlockW& B, const float* x, float* out, int n, int H, int W, hipStream_t st) {
        const int P = H * W;
        const long n_pix = (long)n * P;
        const long n_mid = n_pix * B.mid;
        const int bs = 256;
        pw(x, B.conv1_w, B.conv1_b, nullptr, x1_, n_pix, B.cin, B.mid, 1, st);
        int li = 0;
        for (int br = 0; br < 4; ++br) {
            const float* cur = x1_;
            for (int k = 0; k <= br; ++k, ++li) {
                float* dst = (k & 1) ? tb_ : ta_;
                // 1x1 linear (no bias) then depthwise 3x3 + BN + ReLU
                pw(cur, B.light[li].pw, -1, nullptr, tt_, n_pix, B.mid, B.mid, 0, st);
                hipLaunchKernelGGL(k_depthwise3x3<float>, dim3((unsigned)((n_mid + bs - 1) / bs)), dim3(bs), 0, st,
                                   tt_, d_w_ + B.light[li].dw, d_w_ + B.light[li].b, dst, H, W, B.mid, n_mid);
                cur = dst;
            }
            hipLaunchKernelGGL(k_gap<float>, dim3(n), dim3(768), 0, st, cur, gap_, P, B.mid);
            const int ppb = 64;
            hipLaunchKernelGGL(k_gate_accumulate<float>, dim3(n, (P + ppb - 1) / ppb), dim3(256), 0, st, cur, gap_,
                               d_w_ + B.fc1_w, d_w_ + B.fc1_b, d_w_ + B.fc2_w, d_w_ + B.fc2_b, acc_, P, B.mid, B.hid,
                               br == 0 ? 1 : 0, ppb);
        }
        const float* identity = x;
        if (B.down_w >= 0) {
            pw(x, B.down_w, B.down_b, nullptr, idn_, n_pix, B.cin, B.cout, 0, st);
            identity = idn_;
        }
        pw(acc_, B.conv3_w, B.conv3_b, identity, out, n_pix, B.mid, B.cout, 1, st);
    }
    void forward_v1(int n, float* d_out, const int* d_out_rows, hipStream_t st) {
        const int bs = 256;
        const int c0 = L_.c[0];
        const long stem_pix = (long)n * 128 * 64;
        if (c0 == 16)
            hipLaunchKernelGGL((k_stem_conv<float, 16>), dim3((unsigned)((stem_pix + bs - 1) / bs)), dim3(bs), 0, st,
                               crops_, d_w_ + L_.stem_w, d_w_ + L_.stem_b, big_a_, stem_pix);
        else if (c0 == 32)
            hipLaunchKernelGGL((k_stem_conv<float, 32>), dim3((unsigned)((stem_pix + 127) / 128)), dim3(128), 0, st,
                               crops_, d_w_ + L_.stem_w, d_w_ + L_.stem_b, big_a_, stem_pix);
        else if (c0 == 48)
            hipLaunchKernelGGL((k_stem_conv<float, 48>), dim3((unsigned)((stem_pix + 63) / 64)), dim3(64), 0, st,
                               crops_, d_w_ + L_.stem_w, d_w_ + L_.stem_b, big_a_, stem_pix);
        else
            hipLaunchKernelGGL((k_stem_conv<float, 64>), dim3((unsigned)((stem_pix + 63) / 64)), dim3(64), 0, st,
                               crops_, d_w_ + L_.stem_w, d_w_ + L_.stem_b, big_a_, stem_pix);
        long tot = (long)n * 64 * 32 * c0;
        hipLaunchKernelGGL(k_maxpool3x3s2<float>, dim3((unsigned)((tot + bs - 1) / bs)), dim3(bs), 0, st, big_a_, big_b_,
                           128, 64, c0, tot);
        float* cur = big_b_;
        float* other = big_a_;
        int H = 64, W = 32;
        for (int s = 0; s < 3; ++s) {
            for (int k = 0; k < 2; ++k) {
                osblock(L_.block[s * 2 + k], cur, other, n, H, W, st);
                std::swap(cur, other);
            }
            if (s < 2) {
                const int c = L_.c[s + 1];
                const long n_pix = (long)n * H * W;
                pw(cur, L_.trans_w[s], L_.trans_b[s], nullptr, other, n_pix, c, c, 1, st);
                tot = (long)n * (H / 2) * (W / 2) * c;
                hipLaunchKernelGGL(k_avgpool2x2<float>, dim3((unsigned)((tot + bs - 1) / bs)), dim3(bs), 0, st, other, cur,
                                   H, W, c, tot);
                H /= 2; W /= 2;
            }
        }
        const int c3 = L_.c[3];
        pw(cur, L_.conv5_w, L_.conv5_b, nullptr, other, (long)n * H * W, c3, c3, 1, st);
        hipLaunchKernelGGL(k_head<float>, dim3(n), dim3(256), 0, st, other, d_w_ + L_.fc_w, d_w_ + L_.fc_b, d_out,
                           d_out_rows, H * W, c3, L_.feat);
    }
    // ---- fused fp16 MFMA path (OSNet-x0.25) ----
    unsigned char* upload(const std::vector<uint8_t>& v) {
        unsigned char* d = dev_alloc<unsigned char>(v.size(), owned_);
        BM_HIP(hipMemcpy(d, v.data(), v.size(), hipMemcpyHostToDevice));
        return d;
    }
    template <class K>
    static void allow_lds(K kernel, int bytes) {
        BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    void prepare_fused() {
        const float* w = h_w_.data();
        std::vector<uint8_t> buf;
        pack_stem(w + L_.stem_w, w + L_.stem_b, buf);
        w_stem_ = upload(buf);
        static const int stage[6] = {0, 0, 1, 1, 2, 2}, cin[6] = {16, 64, 64, 96, 96, 128}, down[6] = {1, 0, 1, 0, 1, 0};
        for (int b = 0; b < 6; ++b) {
            bp_[b] = make_blk_pack(stage[b], cin[b], down[b]);
            pack_osblock(w, L_.block[b], bp_[b], buf);
            w_blk_[b] = upload(buf);
        }
        pack_pointwise(w + L_.trans_w[0], w + L_.trans_b[0], 64, 64, buf, 0.25f); w_tr_[0] = upload(buf);
        pack_pointwise(w + L_.trans_w[1], w + L_.trans_b[1], 96, 96, buf, 0.25f); w_tr_[1] = upload(buf);
        pack_pointwise(w + L_.conv5_w, w + L_.conv5_b, 128, 128, buf); w_c5_ = upload(buf);
        pack_fc(w + L_.fc_w, w + L_.fc_b, 512, 128, buf); w_fc_ = upload(buf);
        const size_t n = (size_t)fused_cap_;
        const size_t crop_halves = n * STEM_ROWS * STEM_COLS * 4;
        crops_h_ = dev_alloc<_Float16>(crop_halves, owned_);
        BM_HIP(hipMemset(crops_h_, 0, crop_halves * 2));          // the 3-pixel border and X channel stay zero
        // largest activation that reaches memory: stage-1 block output, 512 px x 96 channels (the 2048 x 64 tensor of stage 0 never does)
        act_a_ = dev_alloc<_Float16>(n * 2048 * 32, owned_);
        act_b_ = dev_alloc<_Float16>(n * 2048 * 32, owned_);
        x1s_ = dev_alloc<_Float16>(n * 2048 * 16, owned_);      // conv1 output of the second stage-0 block (k_osblock EMIT -> RECON)
        x2s_ = dev_alloc<_Float16>(n * 2048 * 16, owned_);      // branch sum of the first stage-0 block
        allow_lds(k_stem_resize_fused, STEM2_LDS);
        allow_lds(k_osblock<0, 16, true, false, true, false>, Geo<0>::LDS_BYTES);
        allow_lds(k_osblock<0, 64, false, true, false, true>, Geo<0>::LDS_BYTES);
#if BM_STAGE1_HANDOVER
        allow_lds(k_osblock<1, 64, true, false, true, false>, Geo<1>::LDS_BYTES);
        allow_lds(k_osblock<1, 96, false, true, false, true>, Geo<1>::LDS_BYTES);
#endif
        allow_lds(k_osblock<1, 64, true, false>, Geo<1>::LDS_BYTES);
        allow_lds(k_osblock<1, 96, false, true>, Geo<1>::LDS_BYTES);
        allow_lds(k_osblock<2, 96, true, false>, Geo<2>::LDS_BYTES);
        allow_lds(k_osblock<2, 128, false, false>, Geo<2>::LDS_BYTES);
        allow_lds(k_head_batched<128, 512>, HeadGeo<128>::LDS_BYTES);
        fused_ready_ = true;
    }
    struct FrameArgs { const uint8_t* const* frames; const int* crop_stream; const float* boxes; int box_stride, W, H; };
    void forward_fused(int n, const FrameArgs* fa, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (fa)     // crop + resize + normalise fused into the stem (the resized crop never reaches HBM)
            hipLaunchKernelGGL(k_stem_resize_fused, dim3(n), dim3(512), STEM2_LDS, st, fa->frames, fa->crop_stream, fa->boxes,
                               fa->box_stride, fa->W, fa->H, d_lut_, act_a_, w_stem_, d_count_);
        else
            hipLaunchKernelGGL(k_stem_fused, dim3(n), dim3(512), 0, st, crops_h_, act_a_, w_stem_, d_count_);
        // activations ping-pong between act_a_ and act_b_; the transitions are fused into the second block of a stage
        auto blk = [&](auto kernel, int nw, int lds, const _Float16* in, _Float16* out, int b, const unsigned char* wtr, BlkLink link) {
            hipLaunchKernelGGL(kernel, dim3(n), dim3(64 * nw), lds, st, in, out, w_blk_[b], bp_[b], d_count_, x1s_, wtr, link);
        };
        // stage 0: block 1 hands block 2 its conv1 result and its branch sum (2 x 16 channels) instead of its 64-channel
        // output; block 2 rebuilds that output per tile for the shortcut (k_osblock EMIT / RECON)
        blk(k_osblock<0, 16, true, false, true, false>, Geo<0>::NWAVES, Geo<0>::LDS_BYTES, act_a_, nullptr, 0, nullptr,
            BlkLink{w_blk_[1], bp_[1].conv1_a, bp_[1].conv1_b, 0, x2s_});
        blk(k_osblock<0, 64, false, true, false, true>, Geo<0>::NWAVES, Geo<0>::LDS_BYTES, act_a_, act_b_, 1, w_tr_[0],
            BlkLink{w_blk_[0], bp_[0].conv3_a, bp_[0].conv3_b, bp_[0].down_a, x2s_});
#if BM_STAGE1_HANDOVER
        // the same hand-over for stage 1 (its 512 x 96 block output is otherwise written once and read twice); validated in
        // emulation (tests/test_reid_emu.py), not yet measured on the device -- off by default
        blk(k_osblock<1, 64, true, false, true, false>, Geo<1>::NWAVES, Geo<1>::LDS_BYTES, act_b_, nullptr, 2, nullptr,
            BlkLink{w_blk_[3], bp_[3].conv1_a, bp_[3].conv1_b, 0, x2s_});
        blk(k_osblock<1, 96, false, true, false, true>, Geo<1>::NWAVES, Geo<1>::LDS_BYTES, act_b_, act_a_, 3, w_tr_[1],
            BlkLink{w_blk_[2], bp_[2].conv3_a, bp_[2].conv3_b, bp_[2].down_a, x2s_});
        _Float16 *s2_in = act_a_, *s2_mid = act_b_;
#else
        blk(k_osblock<1, 64, true, false>, Geo<1>::NWAVES, Geo<1>::LDS_BYTES, act_b_, act_a_, 2, nullptr, BlkLink{});
        blk(k_osblock<1, 96, false, true>, Geo<1>::NWAVES, Geo<1>::LDS_BYTES, act_a_, act_b_, 3, w_tr_[1], BlkLink{});
        _Float16 *s2_in = act_b_, *s2_mid = act_a_;
#endif
        blk(k_osblock<2, 96, true, false>, Geo<2>::NWAVES, Geo<2>::LDS_BYTES, s2_in, s2_mid, 4, nullptr, BlkLink{});
        blk(k_osblock<2, 128, false, false>, Geo<2>::NWAVES, Geo<2>::LDS_BYTES, s2_mid, s2_in, 5, nullptr, BlkLink{});
#if BM_HEAD_PER_CROP
        hipLaunchKernelGGL((k_head_fused<128, 512>), dim3(n), dim3(128), 0, st, s2_in, w_c5_, w_fc_, d_out, d_out_rows, d_count_);
#else
        hipLaunchKernelGGL((k_head_batched<128, 512>), dim3((n + HEAD_NB - 1) / HEAD_NB), dim3(256), HeadGeo<128>::LDS_BYTES, st, s2_in,
                           w_c5_, w_fc_, d_out, d_out_rows, d_count_, n);
#endif
    }
    // ---- fused fp32-grade path (OSNet-x0.25, mode 2): reid_hp.hpp ----
    void prepare_hp() {
        const float* w = h_w_.data();
        std::vector<uint8_t> buf;
        pack_stem_hp(w + L_.stem_w, w + L_.stem_b, buf);
        hw_stem_ = upload(buf);
        {
            const float mean_i[3] = {0.485f, 0.456f, 0.406f}, std_i[3] = {0.229f, 0.224f, 0.225f};     // base_backend.py:189-193
            pack_stem_hp_fused(w + L_.stem_w, w + L_.stem_b, mean_i, std_i, buf);
            hw_stem_fused_ = upload(buf);
        }
        allow_lds(k_stem_resize_fused_hp, STEM2_LDS_HP);
        static const int stage[6] = {0, 0, 1, 1, 2, 2}, cin[6] = {16, 64, 64, 96, 96, 128}, down[6] = {1, 0, 1, 0, 1, 0};
        for (int b = 0; b < 6; ++b) {
            hbp_[b] = make_blk_pack_hp(stage[b], cin[b], down[b]);
            pack_osblock_hp(w, L_.block[b], hbp_[b], buf);
            hw_blk_[b] = upload(buf);
        }
        pack_pointwise_hp(w + L_.trans_w[0], w + L_.trans_b[0], 64, 64, buf, 0.25f); hw_tr_[0] = upload(buf);
        pack_pointwise_hp(w + L_.trans_w[1], w + L_.trans_b[1], 96, 96, buf, 0.25f); hw_tr_[1] = upload(buf);
        pack_pointwise_hp(w + L_.conv5_w, w + L_.conv5_b, 128, 128, buf); hw_c5_ = upload(buf);
        pack_fc_hp(w + L_.fc_w, w + L_.fc_b, 512, 128, buf); hw_fc_ = upload(buf);
        const size_t n = (size_t)fused_cap_;
        const size_t crop_halves = n * STEM_ROWS * STEM_COLS * 4;
        crops_l_ = dev_alloc<_Float16>(crop_halves, owned_);
        BM_HIP(hipMemset(crops_l_, 0, crop_halves * 2));
        hact_al_ = dev_alloc<_Float16>(n * 2048 * 32, owned_);      // lo planes beside act_a_ / act_b_ (the hi planes)
        hact_bl_ = dev_alloc<_Float16>(n * 2048 * 32, owned_);
        hx1s_ = dev_alloc<float>(n * 2048 * 16, owned_);            // fp32 hand-over tensors of the stage-0 block pair
        hx2s_ = dev_alloc<float>(n * 2048 * 16, owned_);
        allow_lds(k_osblock_hp<0, 16, true, false, true, false>, GeoHP<0>::LDS_BYTES);
        allow_lds(k_osblock_hp<0, 64, false, true, false, true>, GeoHP<0>::LDS_BYTES);
        allow_lds(k_osblock_hp<1, 64, true, false>, GeoHP<1>::LDS_BYTES);
        allow_lds(k_osblock_hp<1, 96, false, true>, GeoHP<1>::LDS_BYTES);
        allow_lds(k_osblock_hp<2, 96, true, false>, GeoHP<2>::LDS_BYTES);
        allow_lds(k_osblock_hp<2, 128, false, false>, GeoHP<2>::LDS_BYTES);
        {
            const char* v = std::getenv("BOXMOT_HIP_REID_PERSIST");
            hp_persist_ = v && v[0] ? std::atoi(v) : BM_HP_PERSIST_DEFAULT;
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&hp_cus_, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || hp_cus_ <= 0) hp_cus_ = 256;
        }
        hp_ready_ = true;
    }
    void forward_hp(int n, const FrameArgs* fa, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (fa)     // crop + resize + normalise fused into the stem on raw pixel values (the resized crop never reaches HBM)
            hipLaunchKernelGGL(k_stem_resize_fused_hp, dim3(n), dim3(512), STEM2_LDS_HP, st, fa->frames, fa->crop_stream, fa->boxes,
                               fa->box_stride, fa->W, fa->H, act_a_, hact_al_, hw_stem_fused_, d_count_);
        else
            hipLaunchKernelGGL(k_stem_hp, dim3(n), dim3(512), 0, st, crops_h_, crops_l_, act_a_, hact_al_, hw_stem_, d_count_);
        auto blk = [&](auto kernel, int lds, const _Float16* ih, const _Float16* il, _Float16* oh, _Float16* ol, int b,
                       const unsigned char* wtr, BlkLinkHP link) {
            const int stg = b / 2, nthr = 64 * (stg == 0 ? GeoHP<0>::NWAVES : (stg == 1 ? GeoHP<1>::NWAVES : GeoHP<2>::NWAVES));
            int grid = n;
            if (BM_HP_PERSIST && hp_persist_ > 0) {      // BOXMOT_HIP_REID_PERSIST in a -DBM_HP_PERSIST=1 build: workgroups loop over crops (BlkLinkHP::n_crops)
                link.n_crops = n;
                const int slots = hp_cus_ * (stg == 2 ? 2 : 1) * hp_persist_;
                grid = n < slots ? n : slots;
            }
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(nthr), lds, st, ih, il, oh, ol, hw_blk_[b], hbp_[b], d_count_, hx1s_, wtr, link);
        };
        // stage 0: block 1 hands block 2 its conv1 result and its branch sum (fp32) instead of its 64-channel output
        blk(k_osblock_hp<0, 16, true, false, true, false>, GeoHP<0>::LDS_BYTES, act_a_, hact_al_, nullptr, nullptr, 0, nullptr,
            BlkLinkHP{hw_blk_[1], hbp_[1].conv1_a, hbp_[1].conv1_b, 0, hx2s_});
        blk(k_osblock_hp<0, 64, false, true, false, true>, GeoHP<0>::LDS_BYTES, act_a_, hact_al_, act_b_, hact_bl_, 1, hw_tr_[0],
            BlkLinkHP{hw_blk_[0], hbp_[0].conv3_a, hbp_[0].conv3_b, hbp_[0].down_a, hx2s_});
        blk(k_osblock_hp<1, 64, true, false>, GeoHP<1>::LDS_BYTES, act_b_, hact_bl_, act_a_, hact_al_, 2, nullptr, BlkLinkHP{});
        blk(k_osblock_hp<1, 96, false, true>, GeoHP<1>::LDS_BYTES, act_a_, hact_al_, act_b_, hact_bl_, 3, hw_tr_[1], BlkLinkHP{});
        blk(k_osblock_hp<2, 96, true, false>, GeoHP<2>::LDS_BYTES, act_b_, hact_bl_, act_a_, hact_al_, 4, nullptr, BlkLinkHP{});
        blk(k_osblock_hp<2, 128, false, false>, GeoHP<2>::LDS_BYTES, act_a_, hact_al_, act_b_, hact_bl_, 5, nullptr, BlkLinkHP{});
        hipLaunchKernelGGL((k_head_hp<128, 512>), dim3((n + HEAD_NB - 1) / HEAD_NB), dim3(256), 0, st, act_b_, hact_bl_, hw_c5_, hw_fc_,
                           d_out, d_out_rows, d_count_, n);
    }
    void alloc_buffers() {
        const size_t n = (size_t)max_crops_;
        crops_ = dev_alloc<float>(n * REID_IN_H * REID_IN_W * 3, owned_);
        const size_t big = n * 128 * 64 * (size_t)L_.c[0];            // stem output is the largest tensor
        const size_t s1 = n * 64 * 32 * (size_t)L_.c[1];
        big_a_ = dev_alloc<float>(big > s1 ? big : s1, owned_);
        big_b_ = dev_alloc<float>(big > s1 ? big : s1, owned_);
        idn_ = dev_alloc<float>(s1, owned_);
        const size_t mid_max = n * 64 * 32 * (size_t)(L_.c[1] / 4);   // mid tensors shrink with depth
        const size_t m2 = n * 32 * 16 * (size_t)(L_.c[2] / 4), m3 = n * 16 * 8 * (size_t)(L_.c[3] / 4);
        size_t mid = mid_max > m2 ? mid_max : m2;
        mid = mid > m3 ? mid : m3;
        x1_ = dev_alloc<float>(mid, owned_);
        ta_ = dev_alloc<float>(mid, owned_);
        tb_ = dev_alloc<float>(mid, owned_);
        tt_ = dev_alloc<float>(mid, owned_);
        acc_ = dev_alloc<float>(mid, owned_);
        gap_ = dev_alloc<float>(n * 512, owned_);
    }

    OsnetLayout L_;
    int max_crops_;
    int fused_cap_;
    const int* d_count_ = nullptr;
    int mode_ = 0;
    bool timed_ = false;
    std::vector<void*> owned_;
    std::vector<float> h_w_;
    OsnetLayout Lw_;                        // what the matrix-pipe families (wide_, wide_hp_) run: L_ or its zero-padded copy
    std::vector<float> h_wpad_;
    const float* hw_w_ = nullptr;           // host / device weights in Lw_'s layout
    const float* dw_w_ = nullptr;
    float* d_w_ = nullptr;
    float* d_lut_ = nullptr;
    float *crops_ = nullptr, *big_a_ = nullptr, *big_b_ = nullptr, *idn_ = nullptr;
    float *x1_ = nullptr, *ta_ = nullptr, *tb_ = nullptr, *tt_ = nullptr, *acc_ = nullptr, *gap_ = nullptr;
    // fused path
    std::unique_ptr<ClipNet> clip_;         // non-null: the weights are a CLP1 blob (CLIP-ReID ViT-B/16)
    std::unique_ptr<WideOsnet> wide_;       // non-null: OSNet widths the layer-per-launch fp16 MFMA kernels take (osnet_x1_0)
    std::unique_ptr<WideOsnetHP> wide_hp_;  // the fp32-grade family for the same widths (created when mode 2 is first selected)
    bool fused_ready_ = false, force_fp32_crops_ = false, fuse_stem_ = true;
    const double* obb_geo_ = nullptr;
    int pad_ = 0;
    BlkPack bp_[6];
    unsigned char* w_stem_ = nullptr;
    unsigned char* w_blk_[6] = {};
    unsigned char* w_tr_[2] = {};
    unsigned char *w_c5_ = nullptr, *w_fc_ = nullptr;
    _Float16 *crops_h_ = nullptr, *act_a_ = nullptr, *act_b_ = nullptr, *x1s_ = nullptr, *x2s_ = nullptr;
    // fused fp32-grade path (allocated when mode 2 is first selected)
    bool hp_ready_ = false;
    int hp_persist_ = 0, hp_cus_ = 256;          // persistent launch form of the fp32-grade block kernels (workgroups per CU slot; 0 = off)
    BlkPackHP hbp_[6];
    unsigned char *hw_stem_ = nullptr, *hw_stem_fused_ = nullptr;
    unsigned char* hw_blk_[6] = {};
    unsigned char* hw_tr_[2] = {};
    unsigned char *hw_c5_ = nullptr, *hw_fc_ = nullptr;
    _Float16 *crops_l_ = nullptr, *hact_al_ = nullptr, *hact_bl_ = nullptr;
    float *hx1s_ = nullptr, *hx2s_ = nullptr;
    hipEvent_t ev_[3];
    std::vector<hipEvent_t> all_events_, free_events_;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_;
};

}  // namespace bm
