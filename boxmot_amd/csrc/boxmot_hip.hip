// C-ABI shared library of the MI355X BoT-SORT update path (include/boxmot_hip.h).
// Host side: opaque handles, buffer staging, kernel sequencing on one HIP stream.
// Device side: botsort_step.hpp (tracker), reid_kernels_v1.hpp / reid_fused.hpp (ReID).
// There is no CPU implementation behind these entry points: without a usable
// HIP device every call fails with an error message.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/boxmot_hip.h"
#include "../../include/boxmot_compat.h"
#include "botsort_alloc.hpp"
#include "botsort_step.hpp"
#include "deepocsort_step.hpp"
#include "strongsort_step.hpp"
#include "reid_engine.hpp"
#include "cmc_ecc.hpp"
#include "cmc_sof.hpp"

namespace {

thread_local std::string g_last_error;

template <class F>
int guard(F&& f) {
    try {
        f();
        g_last_error.clear();
        return 1;
    } catch (const std::exception& e) {
        g_last_error = e.what();
    } catch (...) {
        g_last_error = "unknown failure in boxmot_hip";
    }
    return 0;
}

// Handle <-> device affinity (include/boxmot_hip.h, "Devices"): a handle lives on the HIP device that was current on the thread
// that created it (its tables, streams, events and engines are that device's); every entry point that takes a handle makes that
// device current for the duration of the call and restores the caller's device on exit, so one process can drive handles of
// several GPUs from any thread (SURVEY.md section 8(e)'s "one process, G HIP devices" layout) without a handle silently following
// the caller's current device.  The reference's threading contract (reid_capi.h:61-70): distinct handles on distinct threads.
struct DeviceBound {
    int device = -1;
    DeviceBound() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
};

class DeviceScope {
    int prev_ = -1;
    bool switched_ = false;
public:
    explicit DeviceScope(int device) {
        if (device < 0) return;
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != device) {
            if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("boxmot_hip: cannot make the handle's device current (hipSetDevice failed)");
            switched_ = true;
        }
    }
    ~DeviceScope() { if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

template <class H, class F>
int guard_on(H* h, F&& f) {
    return guard([&]() {
        DeviceScope scope(h ? h->device : -1);
        f();
    });
}

// destroy on the handle's own device (streams / events / allocations are released where they were made)
template <class H>
void destroy_on(H* h) {
    if (!h) return;
    try {
        DeviceScope scope(h->device);
        delete h;
    } catch (...) {
        delete h;
    }
}

constexpr int STEP_THREADS = 512;
constexpr int SS_STEP_THREADS_BIG = 1024;     // StrongSORT frame step with max_tracks >= 1024

template <int NTHR, bool DBG = false>
__global__ void __launch_bounds__(NTHR) botsort_step_kernel(bm::BotSortStepArgs args) {
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    __shared__ float sA[bm::COST_TILE][bm::COST_KC + 1];
    __shared__ float sB[bm::COST_TILE][bm::COST_KC + 1];
    BM_DYNAMIC_LDS_T(unsigned char, dyn_lds);       // assignment-solver state, sized by lap_lds_bytes(cap, max_dets)
    bm::botsort_step_stream<NTHR, DBG>(args, args.stream_base + blockIdx.x, s_int, s_dbl, sA, sB, dyn_lds);
}

// the same frame step for oriented detections (botsort_step.hpp: namespace bm::obb)
template <int NTHR>
__global__ void __launch_bounds__(NTHR) botsort_obb_step_kernel(bm::BotSortStepArgs args) {
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    __shared__ float sA[bm::COST_TILE][bm::COST_KC + 1];
    __shared__ float sB[bm::COST_TILE][bm::COST_KC + 1];
    BM_DYNAMIC_LDS_T(unsigned char, dyn_lds);
    bm::obb::botsort_step_stream<NTHR>(args, args.stream_base + blockIdx.x, s_int, s_dbl, sA, sB, dyn_lds);
}

// BotSort._obb_detections_to_cmc_boxes (botsort.py:126-132 over STrack.obb_to_xyxy, botsort_track.py:159-174): the enclosing axis-aligned
// box of every oriented detection of stream blockIdx.x + s0 -- what the reference hands its camera-motion estimator as the mask boxes --
// from the four corners as cv2.boxPoints lays them out (fp32; w, h floored at 1e-4; the angle in degrees as the fp32 product np.degrees is)
__global__ void obb_enclosing_boxes_kernel(const float* dets, const int* n_dets, int max_dets, float* boxes, int s0) {
    const int s = s0 + blockIdx.x;
    const int n = n_dets[s] > 0 ? n_dets[s] : 0;
    const float* d = dets + (long)s * max_dets * bm::obb::DET_COLS;
    float* o = boxes + (long)s * max_dets * 4;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float* r = d + j * bm::obb::DET_COLS;
        const double rect[5] = {(double)r[0], (double)r[1], (double)r[2] > 1e-4 ? (double)r[2] : 1e-4, (double)r[3] > 1e-4 ? (double)r[3] : 1e-4, (double)r[4]};
        double p[4][2];
        bm::obb::obb_corners_deg(rect, (double)(r[4] * (180.0f / 3.14159274f)), p);
        float x0 = (float)p[0][0], x1 = x0, y0 = (float)p[0][1], y1 = y0;
        for (int k = 1; k < 4; ++k) {
            const float x = (float)p[k][0], y = (float)p[k][1];
            x0 = x < x0 ? x : x0; x1 = x > x1 ? x : x1; y0 = y < y0 ? y : y0; y1 = y > y1 ? y : y1;
        }
        o[j * 4 + 0] = x0; o[j * 4 + 1] = y0; o[j * 4 + 2] = x1; o[j * 4 + 3] = y1;
    }
}

template <int NTHR>
__global__ void __launch_bounds__(NTHR) deepocsort_step_kernel(bm::DocsStepArgs args) {
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    BM_DYNAMIC_LDS_T(unsigned char, dyn_lds);       // assignment-solver state, docs_lap_lds_bytes(cap, max_dets)
    bm::docs_step_stream<NTHR>(args, args.stream_base + blockIdx.x, s_int, s_dbl, dyn_lds);
}

// the same frame step for oriented detections: OC-SORT with is_obb (deepocsort_step.hpp: namespace bm::obb)
template <int NTHR>
__global__ void __launch_bounds__(NTHR) deepocsort_obb_step_kernel(bm::DocsStepArgs args) {
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    BM_DYNAMIC_LDS_T(unsigned char, dyn_lds);
    bm::obb::docs_step_stream<NTHR>(args, args.stream_base + blockIdx.x, s_int, s_dbl, dyn_lds);
}

template <int NTHR>
__global__ void __launch_bounds__(NTHR) strongsort_detnorm_kernel(bm::SsStepArgs args) {
    bm::ss_det_norm_block<NTHR>(args, args.stream_base + blockIdx.x, blockIdx.y, gridDim.y);
}

template <int NTHR>
__global__ void __launch_bounds__(NTHR) strongsort_bank_kernel(bm::SsStepArgs args) {
    __shared__ float sA[bm::SS_KC][bm::SS_TILE + 1];
    __shared__ float sB[bm::SS_KC][bm::SS_TILE + 1];
    __shared__ float sMin[16][bm::SS_TILE];
    bm::ss_bank_distance_block<NTHR>(args, args.stream_base + blockIdx.y, blockIdx.x, sA, sB, sMin);
}

template <int NTHR>
__global__ void __launch_bounds__(NTHR) strongsort_bank_mfma_kernel(bm::SsStepArgs args) {
    __shared__ __attribute__((aligned(16))) float lds[bm::SS_MFMA_LDS_FLOATS(NTHR)];
    bm::ss_bank_distance_block_mfma<NTHR>(args, args.stream_base + blockIdx.y, blockIdx.x, lds);
}

template <int NTHR>
__global__ void __launch_bounds__(NTHR) strongsort_step_kernel(bm::SsStepArgs args) {
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    BM_DYNAMIC_LDS_T(unsigned char, dyn_lds);       // assignment-solver state, ss_lsa_lds_bytes(max(cap, max_dets))
    bm::ss_step_stream<NTHR>(args, args.stream_base + blockIdx.x, s_int, s_dbl, dyn_lds);
}

// Build the ReID crop list on the device: every detection with conf > track_high_thresh
// (botsort.py:191-192, :260).  One workgroup per stream; ranges reserved with one atomic.
__global__ void build_crop_list_kernel(const float* dets, const int* n_dets, int max_dets, double high,
                                       int* crop_count, int* crop_stream, float* crop_boxes, int* crop_row,
                                       int stream_base, int inclusive = 0) {
    __shared__ int s_base;
    __shared__ int s_cnt;
    const int s = stream_base + blockIdx.x;
    const int n = n_dets[s];
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const float* d = dets + (long)s * max_dets * bm::DET_COLS;
    int mine = 0;
    auto pass = [&](int j) { const double cf = (double)d[j * bm::DET_COLS + 4]; return inclusive ? cf >= high : cf > high; };
    for (int j = threadIdx.x; j < n; j += blockDim.x)
        if (pass(j)) ++mine;
    const int local = atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) s_base = atomicAdd(crop_count, s_cnt);
    __syncthreads();
    int k = s_base + local;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        if (pass(j)) {
            crop_stream[k] = s;
            crop_row[k] = s * max_dets + j;
            for (int q = 0; q < 4; ++q) crop_boxes[k * 4 + q] = d[j * bm::DET_COLS + q];
            ++k;
        }
    }
}

// A backbone family that sizes its launches on the host (the wide OSNets, CLIP-ReID) and a caller that declared an upper bound on the
// step's crops (set_crop_bound): the list is filled up to the bound with copies of its first entry -- the same crop computes the same
// embedding and writes it to the same row -- so the launches are sized by the bound and the count never travels to the host.  A
// count above the bound raises `flag` (reported by the next synchronize) and switches the frame off for the step.
// `ndets_out` (the detection counts the frame step of this call reads) = the caller's counts, or -1 for EVERY stream when the count
// exceeds the bound: the step kernels treat a negative count as "stream not stepped in this call", so a frame whose crops did not all
// get an embedding changes no tracker state and returns no rows (round-4 advisor finding: it used to step with stale embeddings).
__global__ void pad_crop_list_kernel(const int* crop_count, int bound, int* crop_stream, float* crop_boxes, int* crop_row, int* flag,
                                     const int* ndets_in, int* ndets_out, int n_streams) {
    const int n = *crop_count;
    if (blockIdx.x == 0)
        for (int s = threadIdx.x; s < n_streams; s += blockDim.x) ndets_out[s] = n > bound ? -1 : ndets_in[s];
    if (n > bound) { if (threadIdx.x == 0 && blockIdx.x == 0) *flag = 1; return; }
    const int s0 = n > 0 ? crop_stream[0] : 0, r0 = n > 0 ? crop_row[0] : 0;
    float b0[4];
    for (int q = 0; q < 4; ++q) b0[q] = n > 0 ? crop_boxes[q] : 0.0f;
    for (int i = n + blockIdx.x * blockDim.x + threadIdx.x; i < bound; i += gridDim.x * blockDim.x) {
        crop_stream[i] = s0; crop_row[i] = r0;
        for (int q = 0; q < 4; ++q) crop_boxes[i * 4 + q] = b0[q];
    }
}

}  // namespace

struct BoxMOTHipReID : DeviceBound {
    std::unique_ptr<bm::ReidEngine> engine;
    std::vector<void*> owned;
    hipStream_t stream = nullptr;
    uint8_t* d_frame = nullptr;
    size_t frame_bytes = 0;
    const uint8_t** d_frames = nullptr;
    int* d_crop_stream = nullptr;
    float* d_boxes = nullptr;
    float* d_feat = nullptr;
    double* d_obb = nullptr;                     // [max_crops][8]: out_w, out_h, inverse 2x3 map of oriented boxes (base_backend.py:91-117)
    bool obb = false;                            // the boxes of the last reid_stage were oriented (5 / 7 / 9 columns)
    int staged_n = -1, staged_rows = 0, staged_cols = 0;      // boxmot_reid_capi_preprocess -> process -> postprocess
    bool staged_done = false;
    ~BoxMOTHipReID() {
        engine.reset();
        for (void* p : owned) (void)hipFree(p);
        if (d_frame) (void)hipFree(d_frame);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// ECC camera-motion estimator (csrc/cmc_ecc.hpp): per stream two small grayscale images (previous / current), gradients, scratch
struct BoxMOTHipEcc : DeviceBound {
    int S = 0, rows = 0, cols = 0, h = 0, w = 0, max_iter = 100;
    double scale = 0.15, eps = 1e-5;
    hipStream_t stream = nullptr;
    std::vector<void*> owned;
    float *img = nullptr, *gx = nullptr, *gy = nullptr, *scratch = nullptr;      // img: [S][2][h * w]
    double* d_warp = nullptr; int* d_info = nullptr;
    const uint8_t** d_frames = nullptr;          // host-API staging: one device frame per stream
    std::vector<uint8_t*> frame_bufs;
    std::vector<int> cur;                        // per stream: which of its two image buffers holds the newest frame
    std::vector<char> has_prev;
    bool owns_stream = true;                     // false: launches on a tracker handle's stream (estimation inside update)
    ~BoxMOTHipEcc() {
        for (void* p : owned) (void)hipFree(p);
        for (auto p : frame_bufs) if (p) (void)hipFree(p);
        if (stream && owns_stream) (void)hipStreamDestroy(stream);
    }
};

// Sparse-optical-flow camera-motion estimator (csrc/cmc_sof.hpp): per stream the previous and the current frame's 8-bit pyramid and
// Scharr derivatives, the keypoints the next frame tracks, scratch of the corner detector
struct BoxMOTHipSof : DeviceBound {
    int S = 0, rows = 0, cols = 0, max_dets = 0;
    bm::SofLevels lv{};
    bm::SofParams prm{};
    bm::SofBuffers buf{};
    hipStream_t stream = nullptr;
    std::vector<void*> owned;
    float* d_dets = nullptr; int* d_ndets = nullptr;        // host-API staging: [S][max_dets][4], [S]
    const uint8_t** d_frames = nullptr;
    std::vector<uint8_t*> frame_bufs;
    bool owns_stream = true;
    ~BoxMOTHipSof() {
        for (void* p : owned) (void)hipFree(p);
        for (auto p : frame_bufs) if (p) (void)hipFree(p);
        if (stream && owns_stream) (void)hipStreamDestroy(stream);
    }
};

// Frame ingest ring: n_slots x n_streams page-locked host frames, their device twins, a copy stream and
// one "uploaded" + one "consumed" event per slot.  The caller decodes frame t + 1 straight into slot (t + 1) % n_slots while the
// kernels of frame t run; submit() queues the slot's H2D DMA on the copy stream, wait() makes the consuming stream wait for it,
// release() lets the next upload into the slot wait for the consumer -- no host synchronisation anywhere.
struct BoxMOTHipIngest : DeviceBound {
    int n_slots = 0, n_streams = 0, rows = 0, cols = 0;
    size_t frame_bytes = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<uint8_t*> h_slot, d_slot;           // [n_slots]: n_streams contiguous frames each
    std::vector<const uint8_t**> d_ptrs;            // [n_slots]: device table of n_streams frame pointers
    std::vector<hipEvent_t> uploaded, consumed;
    std::vector<char> has_consumer;
    ~BoxMOTHipIngest() {
        if (copy_stream) (void)hipStreamSynchronize(copy_stream);
        for (auto p : h_slot) if (p) (void)hipHostFree(p);
        for (auto p : d_slot) if (p) (void)hipFree(p);
        for (auto p : d_ptrs) if (p) (void)hipFree(p);
        for (auto e : uploaded) (void)hipEventDestroy(e);
        for (auto e : consumed) (void)hipEventDestroy(e);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

struct BoxMOTHipBotSort : DeviceBound {
    BoxMOTHipBotSortConfig cfg{};
    std::string reid_path;
    std::vector<float> reid_blob;            // host copy of the weight blob the engine was built from (path or set_reid_blob): growth rebuilds from it
    bm::BotSortStepArgs args{};
    std::vector<void*> owned;
    hipStream_t stream = nullptr;
    hipEvent_t ev[2]{};
    hipEvent_t timer_ev[2]{};
    int S = 1, cap = 0, nd = 0, dim = 0, n_lists = 1;
    // io staging
    float* d_dets = nullptr;
    int* d_ndets = nullptr;
    float* d_embs = nullptr;
    float* d_out = nullptr;
    int* d_out_n = nullptr;
    int* d_list_sel = nullptr;
    int* d_fc_set = nullptr;
    double* d_warp = nullptr;      // [S][6] camera-motion warps for the next update of each stream
    int* d_warp_flag = nullptr;
    std::vector<float> h_dets, h_out;
    std::vector<int> h_ndets, h_out_n, h_list_sel, h_fc_set, h_warp_flag;
    std::vector<double> h_warp;
    // frames owned by the handle (host API)
    std::vector<uint8_t*> frame_bufs;
    size_t frame_bytes = 0;
    int frame_rows = 0, frame_cols = 0;
    const uint8_t** d_frames = nullptr;
    // reid
    std::unique_ptr<bm::ReidEngine> reid;
    int reid_mode = 0, reid_pad = 0;
    bool is_obb = false;                         // oriented detections (7 columns in, 9 out, 10-state filter): config.is_obb
    float* d_cmc_boxes = nullptr;                // [S][nd][4] enclosing boxes of the oriented detections: the SOF estimator's mask boxes
    int det_cols() const { return is_obb ? bm::obb::DET_COLS : bm::DET_COLS; }
    int out_cols() const { return is_obb ? bm::obb::OUT_COLS : bm::OUT_COLS; }
    int kf_stride() const { return is_obb ? bm::obb::KF_STRIDE : bm::KF_STRIDE; }
    bool use_ecc = false;                        // cmc_method = "ecc": the estimator runs inside update on the uploaded frame
    std::unique_ptr<BoxMOTHipEcc> ecc;
    bool use_sof = false;                        // cmc_method = "sof" (configs/trackers/botsort.yaml): likewise, masked by the frame's detections
    bool cmc_with_fc_set = false;                // one update with a frame-counter preset that is NOT a per-class fan-out call (compat adapter's first
                                                 // real frame after empty ones): the estimator sees the frame as it does in the reference
    std::unique_ptr<BoxMOTHipSof> sof;
    int* d_crop_count = nullptr;
    int* d_crop_stream = nullptr;
    float* d_crop_boxes = nullptr;
    int* d_crop_row = nullptr;
    long long* d_phase_clock = nullptr;
    // parity debugging (boxmot_hip_botsort_debug_costs_enable): copies of the association cost matrices of the last step
    bool dbg_costs = false;
    double* d_dbg_cost = nullptr;
    int* d_dbg_shape = nullptr;
    int dbg_cap = 0, dbg_nd = 0;
    double last_track_ms = 0, last_reid_pre_ms = 0, last_reid_proc_ms = 0;
    // growth of the tables (grow_tables): what botsort_allocate handed out, in order; slots in use per stream after the last
    // host update (-1 = unknown: a device-resident step ran since)
    std::vector<std::pair<void*, size_t>> table_rec;
    std::vector<int> h_used, h_count_buf;
    int n_grows = 0;
    // step_device as a two-stage pipeline over consecutive asynchronous calls (see StreamIo below: the same scheme): the ReID pass of
    // frame t + 1 on `reid_stream` while the frame step of frame t runs on `stream`, alternating embedding tables.
    hipStream_t reid_stream = nullptr;
    float* d_embs_alt = nullptr;
    hipEvent_t ev_reid_done[2] = {}, ev_embs_free[2] = {}, ev_main = nullptr;
    int pipe_slot = 0;
    bool pipe = true;
    bool engine_on_main = false;
    bool stream_exposed = false;        // boxmot_hip_botsort_stream() was handed out (see StreamIo)

    ~BoxMOTHipBotSort() {
        if (reid_stream) (void)hipStreamSynchronize(reid_stream);
        if (stream) (void)hipStreamSynchronize(stream);
        for (hipEvent_t e : ev_reid_done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_embs_free) if (e) (void)hipEventDestroy(e);
        if (ev_main) (void)hipEventDestroy(ev_main);
        if (reid_stream) (void)hipStreamDestroy(reid_stream);
        reid.reset();
        for (void* p : owned) (void)hipFree(p);
        for (auto* p : frame_bufs) if (p) (void)hipFree(p);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : timer_ev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Host-side plumbing shared by the DeepOCSORT and StrongSORT handles: pinned-down staging of the per-stream inputs,
// the frames and ReID engine for "embeddings not supplied", pending camera-motion warps, result read-back.
struct StreamIo : DeviceBound {
    std::string reid_path;
    std::vector<float> reid_blob;            // host copy of the weight blob (see BoxMOTHipBotSort::reid_blob)
    std::vector<void*> owned;
    hipStream_t stream = nullptr;
    int S = 1, cap = 0, nd = 0, dim = 0;
    int det_cols = bm::DET_COLS, out_cols = bm::OUT_COLS;      // 7 / 9 on an oriented handle (OC-SORT with is_obb)
    float* d_dets = nullptr; int* d_ndets = nullptr; float* d_embs = nullptr; float* d_out = nullptr; int* d_out_n = nullptr;
    double* d_warp = nullptr; int* d_warp_flag = nullptr;
    std::vector<float> h_dets, h_out;
    std::vector<int> h_ndets, h_out_n, h_warp_flag;
    std::vector<double> h_warp;
    std::vector<uint8_t*> frame_bufs;
    size_t frame_bytes = 0;
    int frame_rows = 0, frame_cols = 0;
    const uint8_t** d_frames = nullptr;
    std::unique_ptr<bm::ReidEngine> reid;
    int reid_mode = -1;             // set_reid_mode's last value (-1: the engine's default), re-applied when the engine is re-made
    int* d_crop_count = nullptr; int* d_crop_stream = nullptr; float* d_crop_boxes = nullptr; int* d_crop_row = nullptr;
    // set_crop_bound: host-declared upper bound on the ReID crops of a device-resident step (-1: none, the count is read back);
    // d_crop_count[1] = "the count exceeded the bound"
    int crop_bound = -1;
    bool bounded_step_pending = false;      // a step sized by crop_bound was queued since the overflow flag was last read (io_check_crop_bound)
    // growth of the tables: what the tracker's allocate function handed out, in order; tracks per stream after the last host
    // update (-1 = unknown: a device-resident step ran since)
    std::vector<std::pair<void*, size_t>> table_rec;
    std::vector<int> h_used;
    int n_grows = 0;
    // device-resident steps upload pending warps from a small ring of staging slots (an event per slot): the host does not wait for
    // the stream -- i.e. for the previous step's ReID pass -- every time a warp is set
    static constexpr int WARP_SLOTS = 8;
    std::vector<double> warp_stage[WARP_SLOTS];
    std::vector<int> warp_flag_stage[WARP_SLOTS];
    hipEvent_t warp_ev[WARP_SLOTS] = {};
    unsigned warp_slot = 0;
    // step_device_frames as a two-stage pipeline over consecutive (asynchronous) calls: the ReID pass of frame t + 1 is enqueued on
    // `reid_stream` and runs while the frame step of frame t -- a handful of workgroups, one per stream, for milliseconds -- is
    // still on `stream`.  A frame's embeddings do not depend on the previous frame's step (the crops come from the detections), so
    // only the embedding table is double-buffered: ReID(t) writes table t & 1 after step(t - 2) has read it (ev_embs_free), step(t)
    // starts after ReID(t) (ev_reid_done).  Every ReID pass is followed by its step on `stream`, so waiting for `stream` waits for
    // both.  BOXMOT_HIP_PIPELINE=0 keeps everything on `stream` (A/B switch, profiles/r5_pipeline_ab.txt).
    hipStream_t reid_stream = nullptr;
    float* d_embs_alt = nullptr;
    hipEvent_t ev_reid_done[2] = {}, ev_embs_free[2] = {}, ev_main = nullptr;
    int pipe_slot = 0;
    bool pipe = true;
    bool engine_on_main = false;        // the engine / crop list were last used on `stream` (host-update paths)
    bool stream_exposed = false;        // *_stream() was handed out: the caller may order its inputs on `stream` -- every ReID pass waits for it
    int* d_ndets_step[2] = {};          // [S] each: the detection counts a bounded step reads (-1 everywhere after a bound overflow)
    const int* step_ndets = nullptr;    // what the frame step of the current step_device_frames call takes as n_dets
    ~StreamIo() {
        if (reid_stream) (void)hipStreamSynchronize(reid_stream);
        if (stream) (void)hipStreamSynchronize(stream);
        for (hipEvent_t e : warp_ev) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_reid_done) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_embs_free) if (e) (void)hipEventDestroy(e);
        if (ev_main) (void)hipEventDestroy(ev_main);
        if (reid_stream) (void)hipStreamDestroy(reid_stream);
        reid.reset();
        for (void* p : owned) (void)hipFree(p);
        for (auto* p : frame_bufs) if (p) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

struct BoxMOTHipDeepOcSort : StreamIo {
    BoxMOTHipDeepOcSortConfig cfg{};
    bm::DocsStepArgs args{};
    int dbg_cap = 0, dbg_nd = 0;    // sizes the debug cost planes (args.dbg_cost) were made at
    bool is_obb = false;            // oriented detections: config.is_obb (7 columns in, 9 out, 9-state filter)
};

struct BoxMOTHipStrongSort : StreamIo {
    BoxMOTHipStrongSortConfig cfg{};
    bm::SsStepArgs args{};
    int dbg_big = 0;                // leading dimension the debug cost planes (args.dbg_cost) were made at
    bool bank_valu = false;         // BOXMOT_HIP_SS_BANK=valu: the scalar-FMA bank-distance kernel (A/B timing, cross-check)
};

namespace {

using bm::dev_alloc;

void require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        throw std::runtime_error("boxmot_hip: no HIP device available (this library has no CPU fallback)");
}

template <typename T>
T* zalloc(size_t n, std::vector<void*>& owned) {
    T* p = dev_alloc<T>(n, owned);
    BM_HIP(hipMemset(p, 0, (n ? n : 1) * sizeof(T)));
    return p;
}

struct DevAlloc {
    std::vector<void*>* owned;
    template <typename T> T* get(size_t n) { return zalloc<T>(n, *owned); }
};

struct RecAlloc {         // DevAlloc that also notes every table (pointer, bytes) in allocation order
    std::vector<void*>* owned;
    std::vector<std::pair<void*, size_t>>* rec;
    template <typename T> T* get(size_t n) { T* p = zalloc<T>(n, *owned); rec->push_back({p, n * sizeof(T)}); return p; }
};

void release(std::vector<void*>& owned, void* p) {
    if (!p) return;
    for (size_t i = 0; i < owned.size(); ++i)
        if (owned[i] == p) { owned.erase(owned.begin() + (long)i); (void)hipFree(p); return; }
}

// per-frame buffers whose size follows max_dets (contents do not outlive an update)
void alloc_det_io(BoxMOTHipBotSort* h) {
    const size_t S = h->S, nd = h->nd, dim = h->dim;
    auto& o = h->owned;
    if (h->reid_stream) BM_HIP(hipStreamSynchronize(h->reid_stream));
    release(o, h->d_dets); release(o, h->d_embs); release(o, h->d_embs_alt); release(o, h->d_out);
    release(o, h->d_crop_stream); release(o, h->d_crop_boxes); release(o, h->d_crop_row);
    h->d_dets = zalloc<float>(S * nd * h->det_cols(), o);
    h->d_embs = zalloc<float>(S * nd * dim, o);
    h->d_embs_alt = zalloc<float>(S * nd * dim, o);          // step_device pipeline: frames alternate between the two tables
    h->d_out = zalloc<float>(S * nd * h->out_cols(), o);
    h->d_crop_stream = zalloc<int>(S * nd, o);
    h->d_crop_boxes = zalloc<float>(S * nd * 4, o);
    h->d_crop_row = zalloc<int>(S * nd, o);
    release(o, h->d_cmc_boxes);
    h->d_cmc_boxes = h->is_obb ? zalloc<float>(S * nd * 4, o) : nullptr;
    h->h_dets.assign(S * nd * h->det_cols(), 0.f);
    h->h_out.assign(S * nd * h->out_cols(), 0.f);
}

// A ReID engine for S x nd crops from the handle's blob copy (read from reid_path the first time).  Returned, not installed: the
// caller commits it together with whatever else changes size, so that a failure leaves the handle as it was.
std::unique_ptr<bm::ReidEngine> new_reid_engine(BoxMOTHipBotSort* h, int nd) {
    if (h->reid_blob.empty()) h->reid_blob = bm::read_blob_file(h->reid_path.c_str());
    const long crops = (long)h->S * nd;
    std::unique_ptr<bm::ReidEngine> e(new bm::ReidEngine(h->reid_blob.data(), (long)h->reid_blob.size(), bm::reid_chunk_for(crops), (int)crops));
    if (e->feature_dim() != h->dim) throw std::runtime_error("boxmot_hip: ReID feature dim != emb_dim");
    e->set_preprocess(h->reid_pad);
    if (h->reid_mode) e->set_mode(h->reid_mode);
    return e;
}
void make_reid_engine(BoxMOTHipBotSort* h) { h->reid = new_reid_engine(h, h->nd); }

void set_step_lds(BoxMOTHipBotSort* h) {
    const long lds = bm::lap_lds_bytes(h->cap, h->nd);
    BM_HIP(hipFuncSetAttribute(h->is_obb ? reinterpret_cast<const void*>(botsort_obb_step_kernel<STEP_THREADS>)
                                         : reinterpret_cast<const void*>(botsort_step_kernel<STEP_THREADS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

// The reference's lists have no capacity (botsort.py:177-250); the device tables do.  They are created at max_tracks / max_dets and
// GROW when a host update would not fit: a new set of tables at the larger sizes, every state table copied stream by stream
// (slot ids stay valid: a stream's slab only gets longer), scratch and per-frame buffers simply re-made, the ReID engine re-created
// for the larger crop count.  The limit is the LDS state of the assignment solver (lap_lds_bytes <= 120 KB), reported loudly.
void grow_tables(BoxMOTHipBotSort* h, int new_cap, int new_nd) {
    if (bm::lap_lds_bytes(new_cap, new_nd) > 120 * 1024)
        throw std::runtime_error("boxmot_hip: " + std::to_string(new_cap) + " tracks x " + std::to_string(new_nd) +
                                 " detections per stream is beyond what the assignment solver's LDS state can hold");
    BM_HIP(hipStreamSynchronize(h->stream));
    // the larger ReID engine first: if it cannot be built (weights, memory) the handle keeps its tables, engine and LDS attribute
    std::unique_ptr<bm::ReidEngine> new_reid;
    if (h->reid && new_nd != h->nd) new_reid = new_reid_engine(h, new_nd);
    bm::BotSortStepArgs na = h->args;
    std::vector<std::pair<void*, size_t>> rec;
    RecAlloc ra{&h->owned, &rec};
    bm::BotSortSizes z{h->S, new_cap, new_nd, h->dim, h->n_lists, h->args.st.removed_alloc, h->is_obb ? 1 : 0};
    bm::botsort_allocate(na, z, ra);
    if (rec.size() != h->table_rec.size()) throw std::runtime_error("boxmot_hip: table layout changed between allocations");
    for (size_t i = 0; i < rec.size() && rec[i].first != (void*)na.sc.det_xywh; ++i) {          // the state tables come first
        const auto& od = h->table_rec[i];
        const size_t rows = (size_t)h->S * (rec[i].first == (void*)na.st.active_list ? h->n_lists : 1);
        if (od.second == rec[i].second) BM_HIP(hipMemcpy(rec[i].first, od.first, od.second, hipMemcpyDeviceToDevice));
        else BM_HIP(hipMemcpy2D(rec[i].first, rec[i].second / rows, od.first, od.second / rows, od.second / rows, rows, hipMemcpyDeviceToDevice));
    }
    for (const auto& od : h->table_rec) release(h->owned, od.first);
    h->table_rec = rec;
    h->args = na;
    const bool more_dets = new_nd != h->nd;
    h->cap = new_cap; h->nd = new_nd;
    if (more_dets) {
        alloc_det_io(h);
        if (new_reid) h->reid = std::move(new_reid);
    }
    set_step_lds(h);
    ++h->n_grows;
}

int grown(int have, int need) {           // at least double, in steps of 64
    int v = have * 2 > need ? have * 2 : need;
    return (v + 63) / 64 * 64;
}
// Growth targets (cap, nd) for a frame that needs (need_cap, need_nd): doubled as above, but when the doubled pair exceeds what
// `fits(cap, nd)` accepts (the assignment solver's LDS state) the sizes fall back towards the need itself in steps of 64 -- a frame
// that fits is never refused because its DOUBLE would not.
template <class Fits>
void grown_pair(int have_cap, int need_cap, int have_nd, int need_nd, Fits fits, int& cap, int& nd) {
    cap = need_cap > have_cap ? grown(have_cap, need_cap) : have_cap;
    nd = need_nd > have_nd ? grown(have_nd, need_nd) : have_nd;
    const int min_cap = need_cap > have_cap ? (need_cap + 63) / 64 * 64 : have_cap, min_nd = need_nd > have_nd ? (need_nd + 63) / 64 * 64 : have_nd;
    while (!fits(cap, nd) && (cap > min_cap || nd > min_nd)) {
        if (cap > min_cap) cap -= 64;
        else nd -= 64;
    }
}

void zero_state(BoxMOTHipBotSort* h) {
    bm::BotSortState& st = h->args.st;
    const size_t S = h->S, cap = h->cap;
    BM_HIP(hipMemsetAsync(st.frame_count, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.id_count, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.n_active, 0, S * h->n_lists * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.n_lost, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.rm_head, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.rm_size, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.stamp, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.status, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.slot_used, 0, S * cap * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.mark, 0, S * cap * 4, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
}

void build(BoxMOTHipBotSort* h) {
    const BoxMOTHipBotSortConfig& c = h->cfg;
    h->use_ecc = c.cmc_method && std::strcmp(c.cmc_method, "ecc") == 0;
    h->use_sof = c.cmc_method && std::strcmp(c.cmc_method, "sof") == 0;
    if (c.cmc_method && c.cmc_method[0] && std::strcmp(c.cmc_method, "none") != 0 && !h->use_ecc && !h->use_sof)
        throw std::runtime_error(std::string("boxmot_hip: camera-motion estimator '") + c.cmc_method + "' is not implemented (have: sof, ecc, none); "
                                 "supply the warp per frame with boxmot_hip_botsort_set_warp");
    h->reid_pad = 0;
    if (c.reid_preprocess && c.reid_preprocess[0]) {
        if (std::strcmp(c.reid_preprocess, "resize_pad") == 0) h->reid_pad = 1;
        else if (std::strcmp(c.reid_preprocess, "resize") != 0)
            throw std::runtime_error("boxmot_hip: unknown ReID preprocess (have: resize, resize_pad)");
    }
    h->cfg.reid_preprocess = nullptr;
    if (c.n_streams < 1 || c.max_tracks < 8 || c.max_dets < 4 || c.emb_dim < 1 || c.n_class_lists < 1)
        throw std::runtime_error("boxmot_hip: invalid capacity configuration");
    if (c.removed_stracks_buffer < 0) throw std::runtime_error("boxmot_hip: removed_stracks_buffer must be >= 0");
    if (c.tracker_kind != 0 && c.tracker_kind != 1) throw std::runtime_error("boxmot_hip: tracker_kind must be 0 (BoT-SORT) or 1 (ByteTrack)");
    if (c.tracker_kind == 1) {
        h->cfg.with_reid = 0; h->cfg.fuse_first_associate = 1;
    }
    h->is_obb = c.is_obb != 0;
    if (h->is_obb) {
        // oriented detections (botsort.py:120-131, bytetrack.py:266-303).  Camera motion: a warp supplied with set_warp is applied to
        // the oriented tracks (STrack.multi_gmc_obb, botsort_track.py:197-230: kf_warp_wave of the oriented layout); the in-handle
        // estimators mask by axis-aligned detection boxes and are not wired to oriented tables (the reference estimates on the
        // enclosing boxes, botsort.py:147-158: the caller does that and supplies the warp).  Embeddings of oriented detections come
        // from the caller (the reference crops rotated rectangles with cv2.warpAffine, reid/backends/base_backend.py:92-118).
        // (cmc_method ecc / sof run on an oriented handle too: SOF is masked by the enclosing boxes, obb_enclosing_boxes_kernel)
        if ((c.reid_model_path && c.reid_model_path[0]) || !h->reid_path.empty())      // (create() moves the path into the handle)
            throw std::runtime_error("boxmot_hip: an oriented-box handle takes embeddings from the caller (embs), not from in-handle ReID weights");
    }
    h->S = c.n_streams; h->cap = c.max_tracks; h->nd = c.max_dets; h->dim = c.emb_dim; h->n_lists = c.n_class_lists;
    BM_HIP(hipStreamCreate(&h->stream));
    {
        // The frame step of this tracker is one 512-thread workgroup per stream at 256 registers: it owns a whole CU.  With fewer
        // streams than CUs the rest of the device is free for the next frame's ReID pass (the pipeline pays, as in configurations 3 / 5);
        // with a stream per CU (the headline's 256) nothing can run beside it -- measured flat, 12.04 vs 12.11 k frames/s,
        // profiles/r5_pipeline_ab.txt -- so the pipeline is on by default only below half the device (BOXMOT_HIP_PIPELINE=1 / 0 forces it).
        const char* v = std::getenv("BOXMOT_HIP_PIPELINE");
        int cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        h->pipe = (v && v[0]) ? v[0] != '0' : 2 * c.n_streams <= cus;
        if (h->pipe && !h->reid_stream) {
            BM_HIP(hipStreamCreate(&h->reid_stream));
            for (int k = 0; k < 2; ++k) {
                BM_HIP(hipEventCreateWithFlags(&h->ev_reid_done[k], hipEventDisableTiming));
                BM_HIP(hipEventCreateWithFlags(&h->ev_embs_free[k], hipEventDisableTiming));
            }
            BM_HIP(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
        }
    }
    BM_HIP(hipEventCreate(&h->ev[0]));
    BM_HIP(hipEventCreate(&h->ev[1]));
    BM_HIP(hipEventCreate(&h->timer_ev[0]));
    BM_HIP(hipEventCreate(&h->timer_ev[1]));
    const size_t S = h->S, cap = h->cap, nd = h->nd, dim = h->dim, nl = h->n_lists;
    auto& o = h->owned;
    h->args.cfg = bm::make_config_dev(c.track_high_thresh, c.track_low_thresh, c.new_track_thresh, c.match_thresh,
                                      c.proximity_thresh, c.appearance_thresh, c.second_match_thresh,
                                      c.unconfirmed_match_thresh, c.unconfirmed_emb_scale, c.fuse_first_associate,
                                      c.tracker_kind == 1 ? 0 : c.with_reid, c.frame_rate, c.track_buffer, c.removed_stracks_buffer,
                                      c.tracker_kind);
    bm::BotSortSizes z{h->S, h->cap, h->nd, h->dim, h->n_lists, c.removed_stracks_buffer > 0 ? c.removed_stracks_buffer : 1, h->is_obb ? 1 : 0};
    RecAlloc table_allocator{&o, &h->table_rec};
    bm::botsort_allocate(h->args, z, table_allocator);
    // io
    alloc_det_io(h);
    h->d_ndets = zalloc<int>(S, o);
    h->d_out_n = zalloc<int>(S, o);
    h->d_list_sel = zalloc<int>(S, o);
    h->d_fc_set = zalloc<int>(S, o);
    h->d_warp = zalloc<double>(S * 6, o);
    h->d_warp_flag = zalloc<int>(S, o);
    h->h_warp.assign(S * 6, 0.0); h->h_warp_flag.assign(S, 0);
    h->h_ndets.assign(S, 0); h->h_out_n.assign(S, 0); h->h_list_sel.assign(S, 0); h->h_fc_set.assign(S, 0);
    h->h_used.assign(S, 0); h->h_count_buf.assign(S * (nl + 1), 0);
    h->frame_bufs.assign(S, nullptr);
    h->d_frames = zalloc<const uint8_t*>(S, o);
    h->d_crop_count = zalloc<int>(1, o);
    h->d_phase_clock = zalloc<long long>(16, o);
    if (bm::lap_lds_bytes(h->cap, h->nd) > 120 * 1024)
        throw std::runtime_error("boxmot_hip: max_tracks/max_dets too large for the assignment solver's LDS state");
    set_step_lds(h);
    (void)cap; (void)nd; (void)dim;
    if (c.with_reid && !h->reid_path.empty()) make_reid_engine(h);
}

void launch_step(BoxMOTHipBotSort* h, int s0, int n_streams, const float* d_dets, const int* d_ndets, const float* d_embs,
                 const int* d_list_sel, const int* d_fc_set, float* d_out, int* d_out_n, bool with_warp = false) {
    bm::BotSortStepArgs a = h->args;
    a.dets = d_dets; a.n_dets = d_ndets; a.embs = d_embs; a.list_sel = d_list_sel; a.frame_count_set = d_fc_set;
    a.warp = with_warp ? h->d_warp : nullptr; a.warp_flag = with_warp ? h->d_warp_flag : nullptr;
    a.out = d_out; a.out_n = d_out_n; a.stream_base = s0; a.phase_clock = h->d_phase_clock;
    a.dbg_cost = nullptr; a.dbg_shape = nullptr;
    if (h->dbg_costs) {
        if (h->dbg_cap != h->cap || h->dbg_nd != h->nd) {           // (re-)made at the tables' current sizes (they may have grown)
            BM_HIP(hipStreamSynchronize(h->stream));
            if (h->d_dbg_cost) { release(h->owned, h->d_dbg_cost); release(h->owned, h->d_dbg_shape); }
            h->d_dbg_cost = zalloc<double>((size_t)h->S * bm::DBG_STAGES * bm::DBG_PLANES * h->nd * h->cap, h->owned);
            h->d_dbg_shape = zalloc<int>((size_t)h->S * bm::DBG_STAGES * 2, h->owned);
            h->dbg_cap = h->cap; h->dbg_nd = h->nd;
        }
        a.dbg_cost = h->d_dbg_cost; a.dbg_shape = h->d_dbg_shape;
    }
    if (h->is_obb)
        hipLaunchKernelGGL((botsort_obb_step_kernel<STEP_THREADS>), dim3(n_streams), dim3(STEP_THREADS),
                           (size_t)bm::lap_lds_bytes(h->cap, h->nd), h->stream, a);
    else if (a.dbg_cost) {          // the parity-debugging instantiation (same source, debug writes compiled in)
        BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(botsort_step_kernel<STEP_THREADS, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bm::lap_lds_bytes(h->cap, h->nd)));
        hipLaunchKernelGGL((botsort_step_kernel<STEP_THREADS, true>), dim3(n_streams), dim3(STEP_THREADS),
                           (size_t)bm::lap_lds_bytes(h->cap, h->nd), h->stream, a);
    } else
        hipLaunchKernelGGL((botsort_step_kernel<STEP_THREADS>), dim3(n_streams), dim3(STEP_THREADS),
                           (size_t)bm::lap_lds_bytes(h->cap, h->nd), h->stream, a);
    BM_HIP(hipGetLastError());
}

// ReID for every high-confidence detection of the first n_streams streams; writes d_embs rows.
void run_reid(BoxMOTHipBotSort* h, int s0, int n_streams, const float* d_dets, const int* d_ndets,
              const uint8_t* const* d_frames, int rows, int cols, float* d_embs, hipStream_t st = nullptr) {
    if (!h->reid) throw std::runtime_error("boxmot_hip: with_reid=1 and no embeddings supplied, but no ReID weights are loaded");
    if (!st) { st = h->stream; h->engine_on_main = true; }
    BM_HIP(hipMemsetAsync(h->d_crop_count, 0, 4, st));
    hipLaunchKernelGGL(build_crop_list_kernel, dim3(n_streams), dim3(256), 0, st, d_dets, d_ndets, h->nd,
                       h->cfg.track_high_thresh, h->d_crop_count, h->d_crop_stream, h->d_crop_boxes, h->d_crop_row, s0);
    if (h->reid->counted_ok()) {
        // crop count stays on the device: launches cover the capacity, surplus workgroups exit at once
        h->reid->run_counted(d_frames, h->d_crop_stream, h->d_crop_boxes, 4, h->d_crop_count, n_streams * h->nd, cols, rows,
                             d_embs, h->d_crop_row, st);
        return;
    }
    int n_crops = 0;
    BM_HIP(hipMemcpyAsync(&n_crops, h->d_crop_count, 4, hipMemcpyDeviceToHost, st));
    BM_HIP(hipStreamSynchronize(st));
    h->reid->run(d_frames, h->d_crop_stream, h->d_crop_boxes, 4, n_crops, cols, rows, d_embs, h->d_crop_row, st);
}

// Status words of streams [s0, s0 + n): a non-zero word is reported once and cleared, so that one overflow neither poisons the
// later updates of this stream nor the updates of the handle's other streams (the step has already run: the caller still
// receives the rows of this frame, see host_update).  Returns the message of the first failing stream ("" when all are ok).
std::string take_status(hipStream_t stream, int* d_status, int s0, int n, const char* tracker) {
    std::vector<int> st(n);
    BM_HIP(hipMemcpy(st.data(), d_status + s0, n * 4, hipMemcpyDeviceToHost));
    std::string msg;
    bool any = false;
    for (int k = 0; k < n; ++k) {
        if (st[k] == bm::STATUS_OK) continue;
        any = true;
        if (!msg.empty()) continue;
        const char* what = st[k] == bm::STATUS_TRACK_CAPACITY ? "track capacity (max_tracks) exceeded: the births of this frame were clamped"
                         : st[k] == bm::STATUS_CLASS_CAPACITY ? "more than 8 classes voted on one track"
                         : st[k] == bm::STATUS_LAP_STALL ? "assignment solver did not converge (non-finite costs?)"
                                                         : "innovation covariance is not positive definite";
        msg = std::string("boxmot_hip: ") + tracker + " stream " + std::to_string(s0 + k) + ": " + what;
    }
    if (any) {
        BM_HIP(hipMemsetAsync(d_status + s0, 0, n * 4, stream));
        BM_HIP(hipStreamSynchronize(stream));
    }
    return msg;
}

void upload_frame(BoxMOTHipBotSort* h, int s, const uint8_t* image, int rows, int cols, int channels) {
    if (channels != 3) throw std::runtime_error("boxmot_hip: ReID needs a 3-channel uint8 BGR image");
    const size_t bytes = (size_t)rows * cols * 3;
    if (h->frame_bufs[s] == nullptr || bytes != h->frame_bytes) {
        if (h->frame_bytes != 0 && bytes != h->frame_bytes)
            throw std::runtime_error("boxmot_hip: frame size changed between updates");
        void* p = nullptr;
        BM_HIP(hipMalloc(&p, bytes));
        h->frame_bufs[s] = static_cast<uint8_t*>(p);
        h->frame_bytes = bytes; h->frame_rows = rows; h->frame_cols = cols;
        BM_HIP(hipMemcpy(h->d_frames, h->frame_bufs.data(), h->S * sizeof(uint8_t*), hipMemcpyHostToDevice));
    }
    BM_HIP(hipMemcpyAsync(h->frame_bufs[s], image, bytes, hipMemcpyHostToDevice, h->stream));
}

// ---- ECC estimator plumbing (C ABI boxmot_hip_ecc_*; also owned by a BoT-SORT handle created with cmc_method = "ecc") ----
void ecc_init(BoxMOTHipEcc* h, int n_streams, int image_rows, int image_cols, double scale, double eps, int max_iter, hipStream_t external) {
    if (n_streams < 1 || image_rows < 8 || image_cols < 8) throw std::runtime_error("boxmot_hip: ECC needs >= 1 stream and a frame of at least 8 x 8");
    if (!(scale > 0.0) || scale > 1.0 || !(eps > 0.0) || max_iter < 1) throw std::runtime_error("boxmot_hip: ECC scale must be in (0, 1], eps > 0, max_iter >= 1");
    h->S = n_streams; h->rows = image_rows; h->cols = image_cols; h->scale = scale; h->eps = eps; h->max_iter = max_iter;
    h->w = (int)std::nearbyint(image_cols * scale); h->h = (int)std::nearbyint(image_rows * scale);     // saturate_cast<int>(ssize * fx)
    if (h->w < 4 || h->h < 4) throw std::runtime_error("boxmot_hip: ECC image too small after scaling");
    if (external) { h->stream = external; h->owns_stream = false; }
    else BM_HIP(hipStreamCreate(&h->stream));
    const size_t P = (size_t)h->h * h->w, S = n_streams;
    h->img = zalloc<float>(S * 2 * P, h->owned);
    h->gx = zalloc<float>(S * P, h->owned); h->gy = zalloc<float>(S * P, h->owned);
    h->scratch = zalloc<float>(S * 3 * P, h->owned);
    h->d_warp = zalloc<double>(S * 6, h->owned); h->d_info = zalloc<int>(S * 2, h->owned);
    h->d_frames = zalloc<const uint8_t*>(S, h->owned);
    h->frame_bufs.assign(S, nullptr); h->cur.assign(S, 0); h->has_prev.assign(S, 0);
}

// one stream, frames d_frames[0]: preprocess into the stream's other buffer; estimate against the previous one when there is one
void ecc_run_one(BoxMOTHipEcc* h, int s, const uint8_t* const* d_frame_ptr, double* out_warp6, int* out_iterations) {
    const long P = (long)h->h * h->w;
    const int nxt = 1 - h->cur[s];
    float* imgs = h->img + (size_t)s * 2 * P;
    const unsigned blocks = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(bm::k_ecc_preprocess, dim3(blocks, 1), dim3(256), 0, h->stream, d_frame_ptr, imgs + nxt * P, P, h->rows, h->cols, h->h,
                       h->w, 1.0 / h->scale);
    double warp[6] = {1, 0, 0, 0, 1, 0};
    int info[2] = {0, 0};
    if (h->has_prev[s]) {
        hipLaunchKernelGGL(bm::k_ecc_gradients, dim3(blocks, 1), dim3(256), 0, h->stream, imgs + nxt * P, P, h->gx + s * P, h->gy + s * P, h->h, h->w);
        hipLaunchKernelGGL(bm::k_ecc_solve, dim3(1), dim3(bm::ECC_THREADS), 0, h->stream, imgs + h->cur[s] * P, imgs + nxt * P, P, h->gx + s * P,
                           h->gy + s * P, h->scratch + (size_t)s * 3 * P, h->d_warp + s * 6, h->d_info + s * 2, h->h, h->w, h->eps, h->max_iter,
                           (float)h->scale);
        BM_HIP(hipMemcpyAsync(warp, h->d_warp + s * 6, sizeof(warp), hipMemcpyDeviceToHost, h->stream));
        BM_HIP(hipMemcpyAsync(info, h->d_info + s * 2, sizeof(info), hipMemcpyDeviceToHost, h->stream));
    }
    BM_HIP(hipStreamSynchronize(h->stream));
    BM_HIP(hipGetLastError());
    h->cur[s] = nxt; h->has_prev[s] = 1;
    for (int k = 0; k < 6; ++k) out_warp6[k] = warp[k];
    if (out_iterations) *out_iterations = info[1];
}

// ---- SOF estimator plumbing (C ABI boxmot_hip_sof_*; also owned by a BoT-SORT handle created with cmc_method = "sof") ----
struct HipLaunch {            // sof_frame's launcher: the kernel sequence of cmc_sof.hpp on the handle's stream
    hipStream_t stream;
    template <class K, class... A>
    void operator()(K kernel, int gx, int gy, int threads, A... args) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)gy), dim3((unsigned)threads), 0, stream, args...);
    }
};

void sof_alloc_dets(BoxMOTHipSof* h, int max_dets) {
    release(h->owned, h->d_dets);
    h->max_dets = max_dets;
    h->d_dets = zalloc<float>((size_t)h->S * max_dets * 4, h->owned);
}

void sof_init(BoxMOTHipSof* h, int n_streams, int image_rows, int image_cols, double scale, int min_inliers, double min_inlier_ratio,
              double thresh, hipStream_t external) {
    if (n_streams < 1) throw std::runtime_error("boxmot_hip: SOF needs >= 1 stream");
    if (!(scale > 0.0) || scale > 1.0 || min_inliers < 0 || !(thresh > 0.0)) throw std::runtime_error("boxmot_hip: SOF scale must be in (0, 1], ransac threshold > 0");
    h->S = n_streams; h->rows = image_rows; h->cols = image_cols;
    const int w = (int)std::nearbyint(image_cols * scale), hh = (int)std::nearbyint(image_rows * scale);     // saturate_cast<int>(ssize * fx)
    // the 21-pixel window borders are reflected: every pyramid level (level 0 included) must be larger than the window
    if (w <= bm::SOF_WIN + 1 || hh <= bm::SOF_WIN + 1) throw std::runtime_error("boxmot_hip: SOF image too small after scaling (needs > 22 x 22)");
    h->lv = bm::sof_levels(hh, w);
    h->prm = bm::SofParams{scale, min_inliers, min_inlier_ratio, thresh};
    if (external) { h->stream = external; h->owns_stream = false; }
    else BM_HIP(hipStreamCreate(&h->stream));
    const size_t T = (size_t)h->lv.total, P = (size_t)hh * w, S = n_streams, K = bm::SOF_MAX_CORNERS;
    auto& o = h->owned;
    bm::SofBuffers& b = h->buf;
    b.pyr_prev = zalloc<uint8_t>(S * T, o); b.pyr_cur = zalloc<uint8_t>(S * T, o);
    b.der_prev = zalloc<short>(S * T * 2, o); b.der_cur = zalloc<short>(S * T * 2, o);
    b.eig = zalloc<float>(S * P, o); b.mask = zalloc<uint8_t>(S * P, o); b.cand = zalloc<int>(S * P, o);
    b.prev_kps = zalloc<float>(S * K * 2, o); b.new_kps = zalloc<float>(S * K * 2, o);
    b.next_pts = zalloc<float>(S * K * 2, o); b.valid_to = zalloc<float>(S * K * 2, o);
    b.status = zalloc<uint8_t>(S * K, o);
    b.st = zalloc<bm::SofState>(S, o);
    b.warp = zalloc<double>(S * 6, o);
    h->d_ndets = zalloc<int>(S, o);
    h->d_frames = zalloc<const uint8_t*>(S, o);
    h->frame_bufs.assign(S, nullptr);
    sof_alloc_dets(h, 64);
}

// streams [s0, s0 + n): d_frame_ptrs = device table of n frame pointers; dets already on the device (d_dets [n][max_dets][stride],
// d_ndets [n], may be null); waits, copies the n warps (and the n state words, 8 ints each) out
void sof_run(BoxMOTHipSof* h, int s0, int n, const uint8_t* const* d_frame_ptrs, const float* d_dets, const int* d_ndets, int max_dets,
             int det_stride, double* out_warps, int* out_info8) {
    HipLaunch launch{h->stream};
    bm::sof_frame(launch, h->buf, h->lv, s0, n, d_frame_ptrs, h->rows, h->cols, d_dets, d_ndets, max_dets, det_stride, h->prm);
    BM_HIP(hipGetLastError());
    std::vector<bm::SofState> st((size_t)n);
    BM_HIP(hipMemcpyAsync(out_warps, h->buf.warp + (size_t)s0 * 6, (size_t)n * 6 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (out_info8) BM_HIP(hipMemcpyAsync(st.data(), h->buf.st + s0, (size_t)n * sizeof(bm::SofState), hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    if (out_info8)
        for (int k = 0; k < n; ++k) {
            const bm::SofState& t = st[(size_t)k];
            const int v[8] = {t.mode, t.n_prev, t.n_valid, t.n_inliers, t.ransac_iters, t.estimated, t.n_kps, t.initialized};
            for (int i = 0; i < 8; ++i) out_info8[k * 8 + i] = v[i];
        }
}

// host detections of one stream -> the handle's staging rows (first four columns: tlbr in frame pixels)
void sof_stage_dets(BoxMOTHipSof* h, int stream, const float* dets, int n_dets, int det_stride) {
    if (n_dets < 0 || (n_dets > 0 && (!dets || det_stride < 4))) throw std::runtime_error("boxmot_hip: SOF detections need >= 4 columns (tlbr)");
    if (n_dets > h->max_dets) {
        BM_HIP(hipStreamSynchronize(h->stream));
        sof_alloc_dets(h, grown(h->max_dets, n_dets));
    }
    std::vector<float> rows((size_t)(n_dets > 0 ? n_dets : 1) * 4, 0.f);
    for (int k = 0; k < n_dets; ++k) for (int c = 0; c < 4; ++c) rows[(size_t)k * 4 + c] = dets[(size_t)k * det_stride + c];
    if (n_dets > 0) BM_HIP(hipMemcpyAsync(h->d_dets + (size_t)stream * h->max_dets * 4, rows.data(), (size_t)n_dets * 16, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_ndets + stream, &n_dets, 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));            // rows / n_dets are stack and local storage
}

struct StreamIn {
    const float* dets; int det_rows;
    const float* embs;
    const uint8_t* image;
};

// Shared host path: stage inputs of streams [s0, s0+n), run ReID if needed, step, read back.
void host_update(BoxMOTHipBotSort* h, int s0, int n, const StreamIn* in, int det_cols, int emb_cols,
                 int image_rows, int image_cols, int image_channels, const int* list_sel, const int* fc_set,
                 float* const* out, int out_capacity_rows, int* out_rows, const uint8_t* const* d_frames_ext = nullptr) {
    {   // the reference has no capacities: make room before the step (every detection of a frame can become a track)
        int need_nd = h->nd, need_cap = h->cap;
        for (int k = 0; k < n; ++k) {
            const int rows = in[k].det_rows > 0 ? in[k].det_rows : 0;
            need_nd = rows > need_nd ? rows : need_nd;
            if (h->h_used[s0 + k] < 0) {                   // a device-resident step ran since the last count: ask the device
                const bm::BotSortState& st = h->args.st;
                std::vector<int> cnt(h->n_lists + 1);
                BM_HIP(hipStreamSynchronize(h->stream));
                BM_HIP(hipMemcpy(cnt.data(), st.n_active + (size_t)(s0 + k) * h->n_lists, h->n_lists * 4, hipMemcpyDeviceToHost));
                BM_HIP(hipMemcpy(cnt.data() + h->n_lists, st.n_lost + s0 + k, 4, hipMemcpyDeviceToHost));
                int used = 0;
                for (int v : cnt) used += v;
                h->h_used[s0 + k] = used;
            }
            need_cap = h->h_used[s0 + k] + rows > need_cap ? h->h_used[s0 + k] + rows : need_cap;
        }
        if (need_nd > h->nd || need_cap > h->cap) {
            int to_cap, to_nd;
            grown_pair(h->cap, need_cap, h->nd, need_nd, [](int c, int d) { return bm::lap_lds_bytes(c, d) <= 120 * 1024; }, to_cap, to_nd);
            grow_tables(h, to_cap, to_nd);
        }
    }
    const int nd = h->nd, dim = h->dim;
    const int DC = h->det_cols(), OC = h->out_cols();
    bool need_reid = false;
    for (int k = 0; k < n; ++k) {
        const int rows = in[k].det_rows;
        if (rows < -1) throw std::runtime_error("Negative matrix dimensions are not allowed.");
        if (rows > 0 && det_cols == 7 && !h->is_obb)
            throw std::runtime_error("boxmot_hip: oriented detections (7 columns) need a handle created with is_obb = 1");
        if (rows > 0 && det_cols == 6 && h->is_obb)
            throw std::runtime_error("boxmot_hip: this handle was created for oriented detections (7 columns)");
        if (rows > 0 && det_cols != DC) throw std::runtime_error("boxmot_hip live tracking supports AABB detections with 6 columns (7 with is_obb).");
        if (rows > 0 && in[k].dets == nullptr) throw std::runtime_error("Detection data pointer is null.");
        if (out_capacity_rows < rows) throw std::runtime_error("boxmot_hip: output buffer is too small for the current frame.");
        if (in[k].embs != nullptr && emb_cols != dim && rows > 0)
            throw std::runtime_error("boxmot_hip: embedding width does not match emb_dim");
        if (h->cfg.with_reid && in[k].embs == nullptr && rows > 0) need_reid = true;
    }
    if (need_reid && h->is_obb)
        throw std::runtime_error("boxmot_hip: an oriented-box handle with with_reid = 1 needs the embeddings of the frame's detections (embs)");
    if (need_reid)
        for (int k = 0; k < n; ++k)
            if (in[k].embs != nullptr && in[k].det_rows > 0)
                throw std::runtime_error("boxmot_hip: either every stream of a batch supplies embeddings or none does");
    // one staging buffer for the whole group
    float* hd = h->h_dets.data() + (size_t)s0 * nd * DC;
    for (int k = 0; k < n; ++k) {
        h->h_ndets[s0 + k] = in[k].det_rows;
        if (in[k].det_rows > 0)
            std::memcpy(hd + (size_t)k * nd * DC, in[k].dets, (size_t)in[k].det_rows * DC * 4);
        h->h_list_sel[s0 + k] = list_sel ? list_sel[k] : 0;
        h->h_fc_set[s0 + k] = fc_set ? fc_set[k] : 0;
    }
    float* d_dets = h->d_dets + (size_t)s0 * nd * DC;
    BM_HIP(hipMemcpyAsync(d_dets, hd, (size_t)n * nd * DC * 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_ndets + s0, h->h_ndets.data() + s0, n * 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_list_sel + s0, h->h_list_sel.data() + s0, n * 4, hipMemcpyHostToDevice, h->stream));
    if (fc_set) BM_HIP(hipMemcpyAsync(h->d_fc_set + s0, h->h_fc_set.data() + s0, n * 4, hipMemcpyHostToDevice, h->stream));
    const bool fanout = fc_set && !h->cmc_with_fc_set;          // per-class fan-out: the caller estimated once and supplied the warp
    const bool sof_here = h->use_sof && !fanout;
    if (sof_here) {
        // cmc_method = "sof" (botsort.py:116-117, :141-145): SOF.apply(img, dets) with the frame's whole detection table as the mask
        for (int k = 0; k < n; ++k) {
            if (in[k].det_rows < 0) continue;
            if (!in[k].image && !d_frames_ext) throw std::runtime_error("boxmot_hip: cmc_method=sof needs the frame (image pointer is null)");
            if (image_channels != 3) throw std::runtime_error("boxmot_hip: cmc_method=sof needs a 3-channel uint8 BGR image");
            if (!d_frames_ext) upload_frame(h, s0 + k, in[k].image, image_rows, image_cols, image_channels);
        }
        if (!h->sof) {
            h->sof.reset(new BoxMOTHipSof());
            sof_init(h->sof.get(), h->S, image_rows, image_cols, 0.15, 8, 0.2, 3.0, h->stream);
        }
        if (h->sof->rows != image_rows || h->sof->cols != image_cols) throw std::runtime_error("boxmot_hip: frame size changed between updates");
        for (int k = 0; k < n; ++k) {
            if (in[k].det_rows < 0) continue;
            const uint8_t* const* fp = (d_frames_ext ? d_frames_ext : h->d_frames) + (s0 + k);
            if (h->is_obb) {        // botsort.py:147-158: the estimator sees the enclosing boxes of the oriented detections
                hipLaunchKernelGGL(obb_enclosing_boxes_kernel, dim3(1), dim3(256), 0, h->stream, h->d_dets, h->d_ndets, nd, h->d_cmc_boxes, s0 + k);
                sof_run(h->sof.get(), s0 + k, 1, fp, h->d_cmc_boxes + (size_t)(s0 + k) * nd * 4, h->d_ndets + s0 + k, nd, 4,
                        h->h_warp.data() + (size_t)(s0 + k) * 6, nullptr);
            } else
            sof_run(h->sof.get(), s0 + k, 1, fp, d_dets + (size_t)k * nd * DC, h->d_ndets + s0 + k, nd, DC,
                    h->h_warp.data() + (size_t)(s0 + k) * 6, nullptr);
            h->h_warp_flag[s0 + k] = 1;
        }
    }
    const bool ecc_here = (h->use_ecc && !fanout) || sof_here;       // below: "the frame is already uploaded"
    if (h->use_ecc && !fanout) {
        // cmc_method = "ecc" (botsort.py:116-117, :141-145): the estimator sees every frame of the stream; its warp is applied to
        // the predicted pool by this update.  (Per-class fan-out calls rewind the frame counter, fc_set: the same frame is updated
        // once per class there -- the estimate is made once by the caller and supplied with set_warp.)
        for (int k = 0; k < n; ++k) {
            if (in[k].det_rows < 0) continue;
            if (!in[k].image) throw std::runtime_error("boxmot_hip: cmc_method=ecc needs the frame (image pointer is null)");
            if (image_channels != 3) throw std::runtime_error("boxmot_hip: cmc_method=ecc needs a 3-channel uint8 BGR image");
            if (!d_frames_ext) upload_frame(h, s0 + k, in[k].image, image_rows, image_cols, image_channels);
        }
        if (!h->ecc) {
            h->ecc.reset(new BoxMOTHipEcc());
            ecc_init(h->ecc.get(), h->S, image_rows, image_cols, 0.15, 1e-5, 100, h->stream);
        }
        if (h->ecc->rows != image_rows || h->ecc->cols != image_cols) throw std::runtime_error("boxmot_hip: frame size changed between updates");
        for (int k = 0; k < n; ++k) {
            if (in[k].det_rows < 0) continue;
            const uint8_t* const* fp = (d_frames_ext ? d_frames_ext : h->d_frames) + (s0 + k);
            ecc_run_one(h->ecc.get(), s0 + k, fp, h->h_warp.data() + (size_t)(s0 + k) * 6, nullptr);
            h->h_warp_flag[s0 + k] = 1;
        }
    }
    bool any_warp = false;
    for (int k = 0; k < n; ++k) any_warp = any_warp || h->h_warp_flag[s0 + k] != 0;
    if (any_warp) {     // warps set with boxmot_hip_botsort_set_warp are consumed by this update
        BM_HIP(hipMemcpyAsync(h->d_warp + (size_t)s0 * 6, h->h_warp.data() + (size_t)s0 * 6, (size_t)n * 6 * 8, hipMemcpyHostToDevice, h->stream));
        BM_HIP(hipMemcpyAsync(h->d_warp_flag + s0, h->h_warp_flag.data() + s0, n * 4, hipMemcpyHostToDevice, h->stream));
    }
    float* d_embs = h->d_embs + (size_t)s0 * nd * dim;
    for (int k = 0; k < n; ++k)
        if (in[k].embs && in[k].det_rows > 0)
            BM_HIP(hipMemcpyAsync(d_embs + (size_t)k * nd * dim, in[k].embs, (size_t)in[k].det_rows * dim * 4,
                                  hipMemcpyHostToDevice, h->stream));
    h->last_reid_pre_ms = h->last_reid_proc_ms = 0;
    if (need_reid && d_frames_ext) {       // frames already on the device (ingest ring slot): no upload
        if (s0 != 0) throw std::runtime_error("boxmot_hip: device frames are addressed from stream 0");
        run_reid(h, s0, n, h->d_dets, h->d_ndets, d_frames_ext, image_rows, image_cols, h->d_embs);
    } else if (need_reid) {
        for (int k = 0; k < n; ++k) {
            if (in[k].image) { if (!ecc_here) upload_frame(h, s0 + k, in[k].image, image_rows, image_cols, image_channels); }
            else if (h->frame_bufs[s0 + k] == nullptr) throw std::runtime_error("Image data pointer is null.");
        }
        run_reid(h, s0, n, h->d_dets, h->d_ndets, h->d_frames, h->frame_rows, h->frame_cols, h->d_embs);
    }
    BM_HIP(hipEventRecord(h->ev[0], h->stream));
    launch_step(h, s0, n, h->d_dets, h->d_ndets, h->cfg.with_reid ? h->d_embs : nullptr, h->d_list_sel,
                fc_set ? h->d_fc_set : nullptr, h->d_out, h->d_out_n, any_warp);
    BM_HIP(hipEventRecord(h->ev[1], h->stream));
    BM_HIP(hipMemcpyAsync(h->h_out_n.data(), h->d_out_n + s0, n * 4, hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipMemcpyAsync(h->h_out.data(), h->d_out + (size_t)s0 * nd * OC, (size_t)n * nd * OC * 4,
                          hipMemcpyDeviceToHost, h->stream));
    const int nl = h->n_lists;
    BM_HIP(hipMemcpyAsync(h->h_count_buf.data(), h->args.st.n_active + (size_t)s0 * nl, (size_t)n * nl * 4, hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipMemcpyAsync(h->h_count_buf.data() + (size_t)n * nl, h->args.st.n_lost + s0, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < n; ++k) {           // slots in use = tracked + lost tracks (removed ones give their slot back)
        int used = h->h_count_buf[(size_t)n * nl + k];
        for (int l = 0; l < nl; ++l) used += h->h_count_buf[(size_t)k * nl + l];
        h->h_used[s0 + k] = used;
    }
    for (int k = 0; k < n; ++k) h->h_warp_flag[s0 + k] = 0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->last_track_ms = ms;
    if (need_reid) h->reid->last_times(h->last_reid_pre_ms, h->last_reid_proc_ms);
    const std::string status_msg = take_status(h->stream, h->args.st.status, s0, n, h->cfg.tracker_kind == 1 ? "ByteTrack" : "BoT-SORT");
    for (int k = 0; k < n; ++k) {
        const int rows = h->h_out_n[k];
        if (rows > out_capacity_rows) throw std::runtime_error("boxmot_hip: output buffer is too small for the current frame.");
        const float* src = h->h_out.data() + (size_t)k * nd * OC;
        for (int r = 0; r < rows; ++r) {
            float* dst = out[k] + (size_t)r * 9;
            for (int q = 0; q < OC; ++q) dst[q] = src[r * OC + q];
            if (OC < 9) dst[8] = 0.0f;
        }
        out_rows[k] = rows;
    }
    if (!status_msg.empty()) throw std::runtime_error(status_msg);      // rows of this frame are in `out` all the same
}

// single stream at an arbitrary index: run it as a one-stream group by offsetting the args
void host_update_one(BoxMOTHipBotSort* h, int stream, int class_list, int frame_count, const StreamIn& in,
                     int det_cols, int emb_cols, int rows, int cols, int channels, float* out, int out_cap,
                     int* out_rows) {
    if (stream < 0 || stream >= h->S) throw std::runtime_error("boxmot_hip: stream index out of range");
    if (class_list < 0 || class_list >= h->n_lists) throw std::runtime_error("boxmot_hip: class list out of range");
    float* outs[1] = {out};
    const int sel[1] = {class_list};
    const int fc[1] = {frame_count};
    host_update(h, stream, 1, &in, det_cols, emb_cols, rows, cols, channels, sel, frame_count >= 0 ? fc : nullptr, outs,
                out_cap, out_rows);
}


// ---------------------------------------------------------------------------
// StreamIo: shared host plumbing of the DeepOCSORT / StrongSORT handles
// ---------------------------------------------------------------------------
std::unique_ptr<bm::ReidEngine> io_new_reid(StreamIo* h, int nd) {
    if (h->reid_blob.empty()) h->reid_blob = bm::read_blob_file(h->reid_path.c_str());
    const long crops = (long)h->S * nd;
    std::unique_ptr<bm::ReidEngine> e(new bm::ReidEngine(h->reid_blob.data(), (long)h->reid_blob.size(), bm::reid_chunk_for(crops), (int)crops));
    if (e->feature_dim() != h->dim) throw std::runtime_error("boxmot_hip: ReID feature dim != emb_dim");
    if (h->reid_mode >= 0) e->set_mode(h->reid_mode);
    return e;
}
void io_make_reid(StreamIo* h) { h->reid = io_new_reid(h, h->nd); }

// per-frame buffers whose size follows max_dets / max_tracks (contents do not outlive an update)
void io_alloc_sized(StreamIo* h) {
    auto& o = h->owned;
    const size_t s = h->S, c = h->cap, n = h->nd, d = h->dim;
    if (h->reid_stream) BM_HIP(hipStreamSynchronize(h->reid_stream));       // nothing in flight reads the tables being replaced
    release(o, h->d_dets); release(o, h->d_embs); release(o, h->d_embs_alt); release(o, h->d_out);
    release(o, h->d_crop_stream); release(o, h->d_crop_boxes); release(o, h->d_crop_row);
    h->d_dets = zalloc<float>(s * n * h->det_cols, o);
    h->d_embs = zalloc<float>(s * n * d, o);
    h->d_embs_alt = zalloc<float>(s * n * d, o);          // step_device_frames pipeline: frames alternate between the two tables
    h->d_out = zalloc<float>(s * c * h->out_cols, o);
    h->d_crop_stream = zalloc<int>(s * n, o);
    h->d_crop_boxes = zalloc<float>(s * n * 4, o);
    h->d_crop_row = zalloc<int>(s * n, o);
    h->h_dets.assign(s * n * h->det_cols, 0.f);
    h->h_out.assign(c * h->out_cols, 0.f);
}

void io_allocate(StreamIo* h, int S, int cap, int nd, int dim, bool with_reid) {
    h->S = S; h->cap = cap; h->nd = nd; h->dim = dim;
    BM_HIP(hipStreamCreate(&h->stream));
    {
        const char* v = std::getenv("BOXMOT_HIP_PIPELINE");
        h->pipe = !(v && v[0] == '0');
        if (h->pipe) {
            BM_HIP(hipStreamCreate(&h->reid_stream));
            for (int k = 0; k < 2; ++k) {
                BM_HIP(hipEventCreateWithFlags(&h->ev_reid_done[k], hipEventDisableTiming));
                BM_HIP(hipEventCreateWithFlags(&h->ev_embs_free[k], hipEventDisableTiming));
            }
            BM_HIP(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
        }
    }
    auto& o = h->owned;
    const size_t s = S;
    io_alloc_sized(h);
    h->d_ndets = zalloc<int>(s, o);
    h->d_out_n = zalloc<int>(s, o);
    h->d_warp = zalloc<double>(s * 6, o);
    h->d_warp_flag = zalloc<int>(s, o);
    h->h_ndets.assign(s, 0); h->h_out_n.assign(s, 0); h->h_used.assign(s, 0);
    h->h_warp.assign(s * 6, 0.0); h->h_warp_flag.assign(s, 0);
    h->frame_bufs.assign(s, nullptr);
    h->d_frames = zalloc<const uint8_t*>(s, o);
    h->d_crop_count = zalloc<int>(2, o);        // [0] the count, [1] the bound-overflow flag
    h->d_ndets_step[0] = zalloc<int>(s, o); h->d_ndets_step[1] = zalloc<int>(s, o);
    if (with_reid && !h->reid_path.empty()) io_make_reid(h);
}

// State tables of a tracker carried over into a larger set (see grow_tables): `rec` = the new tables in allocation order, the
// state tables first, up to `first_scratch`; every state table is [stream][slot][...], so a stream's slab only gets longer.
void io_migrate(StreamIo* h, const std::vector<std::pair<void*, size_t>>& rec, const void* first_scratch) {
    if (rec.size() != h->table_rec.size()) throw std::runtime_error("boxmot_hip: table layout changed between allocations");
    const size_t rows = (size_t)h->S;
    for (size_t i = 0; i < rec.size() && rec[i].first != first_scratch; ++i) {
        const auto& od = h->table_rec[i];
        if (od.second == rec[i].second) BM_HIP(hipMemcpy(rec[i].first, od.first, od.second, hipMemcpyDeviceToDevice));
        else BM_HIP(hipMemcpy2D(rec[i].first, rec[i].second / rows, od.first, od.second / rows, od.second / rows, rows, hipMemcpyDeviceToDevice));
    }
    for (const auto& od : h->table_rec) release(h->owned, od.first);
    h->table_rec = rec;
}

// Room for the frames of streams [s0, s0 + n) before they are staged: more detections than max_dets, or live tracks + this
// frame's detections > max_tracks, re-makes the tables through `grow(new_cap, new_nd)` (the reference's lists have no limit).
template <class Grow, class Fits>
void io_make_room(StreamIo* h, int s0, int n, const StreamIn* in, const int* d_n_tracks, Grow grow, Fits fits) {
    if (s0 < 0 || n < 1 || s0 + n > h->S) throw std::runtime_error("boxmot_hip: stream index out of range");
    int need_nd = h->nd, need_cap = h->cap;
    for (int k = 0; k < n; ++k) {
        const int rows = in[k].det_rows > 0 ? in[k].det_rows : 0;
        need_nd = rows > need_nd ? rows : need_nd;
        if (h->h_used[s0 + k] < 0) {
            BM_HIP(hipStreamSynchronize(h->stream));
            BM_HIP(hipMemcpy(&h->h_used[s0 + k], d_n_tracks + s0 + k, 4, hipMemcpyDeviceToHost));
        }
        need_cap = h->h_used[s0 + k] + rows > need_cap ? h->h_used[s0 + k] + rows : need_cap;
    }
    if (need_nd > h->nd || need_cap > h->cap) {
        BM_HIP(hipStreamSynchronize(h->stream));
        int to_cap, to_nd;
        grown_pair(h->cap, need_cap, h->nd, need_nd, fits, to_cap, to_nd);
        grow(to_cap, to_nd);
        ++h->n_grows;
    }
}

void io_set_warp(StreamIo* h, int stream, const double* warp_2x3) {
    if (stream < 0 || stream >= h->S) throw std::runtime_error("boxmot_hip: stream index out of range");
    if (warp_2x3 == nullptr) { h->h_warp_flag[stream] = 0; return; }
    for (int k = 0; k < 6; ++k) {
        if (!std::isfinite(warp_2x3[k])) throw std::runtime_error("boxmot_hip: camera-motion warp has non-finite entries");
        h->h_warp[(size_t)stream * 6 + k] = warp_2x3[k];
    }
    h->h_warp_flag[stream] = 1;
}

// Device-resident steps: the warps set with *_set_warp since the last step are consumed by this one (identity for the streams
// without a pending warp).  Returns whether any stream had one; the caller clears the flags after launching.
bool io_consume_warps(StreamIo* h) {
    bool any_warp = false;
    for (int s = 0; s < h->S; ++s) any_warp = any_warp || h->h_warp_flag[s] != 0;
    if (!any_warp) return false;
    for (int s = 0; s < h->S; ++s)
        if (!h->h_warp_flag[s]) { double* w = h->h_warp.data() + (size_t)s * 6; w[0] = 1; w[1] = 0; w[2] = 0; w[3] = 0; w[4] = 1; w[5] = 0; }
    // h_warp / h_warp_flag are rewritten by the next set_warp: the copies read a staging slot of their own, reused (after its event)
    // eight steps later -- no wait for the stream here
    const int k = (int)(h->warp_slot++ % StreamIo::WARP_SLOTS);
    if (h->warp_ev[k]) BM_HIP(hipEventSynchronize(h->warp_ev[k]));
    else BM_HIP(hipEventCreateWithFlags(&h->warp_ev[k], hipEventDisableTiming));
    h->warp_stage[k].assign(h->h_warp.begin(), h->h_warp.begin() + (size_t)h->S * 6);
    h->warp_flag_stage[k].assign(h->h_warp_flag.begin(), h->h_warp_flag.begin() + h->S);
    BM_HIP(hipMemcpyAsync(h->d_warp, h->warp_stage[k].data(), (size_t)h->S * 6 * 8, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_warp_flag, h->warp_flag_stage[k].data(), h->S * 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipEventRecord(h->warp_ev[k], h->stream));
    return true;
}
// after a device-resident step: pending warps are consumed, and the host no longer knows the track counts (io_make_room asks)
void io_clear_warps(StreamIo* h) { for (int s = 0; s < h->S; ++s) { h->h_warp_flag[s] = 0; h->h_used[s] = -1; } }

// Validate and upload the inputs of the first n streams; run the ReID engine on every detection passing the confidence
// test (`conf > thresh`, or `>=` when inclusive) when embeddings are wanted and not supplied.  Streams without a pending
// warp get the identity.  Returns whether any stream has a pending warp.
bool io_stage(StreamIo* h, int n, const StreamIn* in, int det_cols, int emb_cols, bool want_emb, int image_rows, int image_cols,
              int image_channels, int out_capacity_rows, double reid_thresh, int inclusive, int s0 = 0) {
    const int nd = h->nd, dim = h->dim;
    const size_t DC = h->det_cols;
    if (s0 < 0 || n < 1 || s0 + n > h->S) throw std::runtime_error("boxmot_hip: stream index out of range");
    bool need_reid = false;
    for (int k = 0; k < n; ++k) {
        const int rows = in[k].det_rows;
        if (rows < -1) throw std::runtime_error("Negative matrix dimensions are not allowed.");
        if (rows > 0 && det_cols != h->det_cols)
            throw std::runtime_error(h->det_cols == 7 ? "boxmot_hip: this handle was created for oriented detections (7 columns)"
                                     : det_cols == 7 ? "boxmot_hip: oriented detections (7 columns) need a handle created with is_obb = 1 (BoT-SORT, ByteTrack, OC-SORT)"
                                                     : "boxmot_hip live tracking supports AABB detections with 6 columns.");
        if (rows > nd) throw std::runtime_error("boxmot_hip: internal: detection tables were not grown before staging");
        if (rows > 0 && in[k].dets == nullptr) throw std::runtime_error("Detection data pointer is null.");
        if (out_capacity_rows < rows) throw std::runtime_error("boxmot_hip: output buffer is too small for the current frame.");
        if (want_emb && in[k].embs != nullptr && emb_cols != dim && rows > 0)
            throw std::runtime_error("boxmot_hip: embedding width does not match emb_dim");
        if (want_emb && in[k].embs == nullptr && rows > 0) need_reid = true;
    }
    if (need_reid)
        for (int k = 0; k < n; ++k)
            if (in[k].embs != nullptr && in[k].det_rows > 0)
                throw std::runtime_error("boxmot_hip: either every stream of a batch supplies embeddings or none does");
    bool any_warp = false;
    for (int k = 0; k < n; ++k) {
        const size_t sk = (size_t)(s0 + k);
        h->h_ndets[sk] = in[k].det_rows;
        if (in[k].det_rows > 0)
            std::memcpy(h->h_dets.data() + sk * nd * DC, in[k].dets, (size_t)in[k].det_rows * DC * 4);
        if (h->h_warp_flag[sk]) any_warp = true;
        else { double* w = h->h_warp.data() + sk * 6; w[0] = 1; w[1] = 0; w[2] = 0; w[3] = 0; w[4] = 1; w[5] = 0; }
    }
    const size_t o = (size_t)s0;
    BM_HIP(hipMemcpyAsync(h->d_dets + o * nd * DC, h->h_dets.data() + o * nd * DC, (size_t)n * nd * DC * 4,
                          hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_ndets + o, h->h_ndets.data() + o, n * 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_warp + o * 6, h->h_warp.data() + o * 6, (size_t)n * 6 * 8, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipMemcpyAsync(h->d_warp_flag + o, h->h_warp_flag.data() + o, n * 4, hipMemcpyHostToDevice, h->stream));
    if (want_emb)
        for (int k = 0; k < n; ++k)
            if (in[k].embs && in[k].det_rows > 0)
                BM_HIP(hipMemcpyAsync(h->d_embs + (size_t)(s0 + k) * nd * dim, in[k].embs, (size_t)in[k].det_rows * dim * 4,
                                      hipMemcpyHostToDevice, h->stream));
    if (!need_reid) return any_warp;
    if (!h->reid) throw std::runtime_error("boxmot_hip: embeddings are needed and none were supplied, but no ReID weights are loaded");
    if (image_channels != 3) throw std::runtime_error("boxmot_hip: ReID needs a 3-channel uint8 BGR image");
    const size_t bytes = (size_t)image_rows * image_cols * 3;
    for (int k = 0; k < n; ++k) {
        const int sk = s0 + k;
        if (!in[k].image) { if (!h->frame_bufs[sk]) throw std::runtime_error("Image data pointer is null."); continue; }
        if (h->frame_bufs[sk] == nullptr || bytes != h->frame_bytes) {
            if (h->frame_bytes != 0 && bytes != h->frame_bytes) throw std::runtime_error("boxmot_hip: frame size changed between updates");
            void* p = nullptr;
            BM_HIP(hipMalloc(&p, bytes));
            h->frame_bufs[sk] = static_cast<uint8_t*>(p);
            h->frame_bytes = bytes; h->frame_rows = image_rows; h->frame_cols = image_cols;
            BM_HIP(hipMemcpy(h->d_frames, h->frame_bufs.data(), h->S * sizeof(uint8_t*), hipMemcpyHostToDevice));
        }
        BM_HIP(hipMemcpyAsync(h->frame_bufs[sk], in[k].image, bytes, hipMemcpyHostToDevice, h->stream));
    }
    h->engine_on_main = true;           // (a later pipelined step_device_frames orders its ReID pass after this use)
    BM_HIP(hipMemsetAsync(h->d_crop_count, 0, 4, h->stream));
    hipLaunchKernelGGL(build_crop_list_kernel, dim3(n), dim3(256), 0, h->stream, h->d_dets, h->d_ndets, nd, reid_thresh,
                       h->d_crop_count, h->d_crop_stream, h->d_crop_boxes, h->d_crop_row, s0, inclusive);
    if (h->reid->counted_ok()) {
        h->reid->run_counted(h->d_frames, h->d_crop_stream, h->d_crop_boxes, 4, h->d_crop_count, n * nd, h->frame_cols,
                             h->frame_rows, h->d_embs, h->d_crop_row, h->stream);
    } else {
        int n_crops = 0;
        BM_HIP(hipMemcpyAsync(&n_crops, h->d_crop_count, 4, hipMemcpyDeviceToHost, h->stream));
        BM_HIP(hipStreamSynchronize(h->stream));
        h->reid->run(h->d_frames, h->d_crop_stream, h->d_crop_boxes, 4, n_crops, h->frame_cols, h->frame_rows, h->d_embs,
                     h->d_crop_row, h->stream);
    }
    return any_warp;
}

// Device-resident ReID for the DeepOCSORT / StrongSORT step_device_frames entry points: crop list from the caller's device
// detections, backbone over the caller's device frames, embeddings into h->d_embs (rows of skipped detections keep stale
// values; the step kernels never read them).
float* io_device_reid(StreamIo* h, const float* d_dets, const int* d_ndets, const uint8_t* const* d_frames, int image_rows,
                      int image_cols, double reid_thresh, int inclusive) {
    if (!h->reid) throw std::runtime_error("boxmot_hip: embeddings are needed and none were supplied, but no ReID weights are loaded");
    if (!d_frames || image_rows <= 0 || image_cols <= 0) throw std::runtime_error("boxmot_hip: step_device_frames needs device frames");
    // pipeline stage 1 (StreamIo): this frame's ReID on `reid_stream`, into the table the step before last has finished reading
    const bool pipe = h->pipe && h->reid_stream;
    hipStream_t rs = pipe ? h->reid_stream : h->stream;
    float* embs = (pipe && h->pipe_slot) ? h->d_embs_alt : h->d_embs;
    h->step_ndets = d_ndets;
    if (pipe) {
        // a host-update path used the engine / the crop list on `stream` since, or the caller holds `stream` (it may have queued the
        // producers of this step's detections / frames there, e.g. boxmot_hip_ingest_wait): the ReID pass is ordered after `stream` --
        // correct always; such a caller gives up the overlap, its own stream order already serialises the frames
        if (h->engine_on_main || h->stream_exposed) {
            BM_HIP(hipEventRecord(h->ev_main, h->stream));
            BM_HIP(hipStreamWaitEvent(rs, h->ev_main, 0));
            h->engine_on_main = false;
        }
        BM_HIP(hipStreamWaitEvent(rs, h->ev_embs_free[h->pipe_slot], 0));
    }
    BM_HIP(hipMemsetAsync(h->d_crop_count, 0, 4, rs));
    hipLaunchKernelGGL(build_crop_list_kernel, dim3(h->S), dim3(256), 0, rs, d_dets, d_ndets, h->nd, reid_thresh,
                       h->d_crop_count, h->d_crop_stream, h->d_crop_boxes, h->d_crop_row, 0, inclusive);
    if (h->reid->counted_ok()) {
        h->reid->run_counted(d_frames, h->d_crop_stream, h->d_crop_boxes, 4, h->d_crop_count, h->S * h->nd, image_cols, image_rows,
                             embs, h->d_crop_row, rs);
    } else if (h->crop_bound >= 0) {
        // launches sized by the caller's bound: no read-back, no stream synchronisation inside the step (the host keeps queueing)
        const int n = h->crop_bound < h->S * h->nd ? h->crop_bound : h->S * h->nd;
        int* nd_step = h->d_ndets_step[pipe ? h->pipe_slot : 0];        // (alternates with the embedding table: read by this call's step)
        // (a bound of 0 -- "no crops in this step" -- is checked like any other: one workgroup compares the count with it)
        hipLaunchKernelGGL(pad_crop_list_kernel, dim3(n > 0 ? (n + 255) / 256 : 1), dim3(256), 0, rs, (const int*)h->d_crop_count, n,
                           h->d_crop_stream, h->d_crop_boxes, h->d_crop_row, h->d_crop_count + 1, d_ndets, nd_step, h->S);
        h->step_ndets = nd_step;
        h->bounded_step_pending = true;
        if (n > 0) h->reid->run(d_frames, h->d_crop_stream, h->d_crop_boxes, 4, n, image_cols, image_rows, embs, h->d_crop_row, rs);
    } else {
        int n_crops = 0;
        BM_HIP(hipMemcpyAsync(&n_crops, h->d_crop_count, 4, hipMemcpyDeviceToHost, rs));
        BM_HIP(hipStreamSynchronize(rs));
        h->reid->run(d_frames, h->d_crop_stream, h->d_crop_boxes, 4, n_crops, image_cols, image_rows, embs, h->d_crop_row, rs);
    }
    if (pipe) {
        BM_HIP(hipEventRecord(h->ev_reid_done[h->pipe_slot], rs));
        BM_HIP(hipStreamWaitEvent(h->stream, h->ev_reid_done[h->pipe_slot], 0));
    }
    return embs;
}
// pipeline stage 2 enqueued (the frame step that reads this frame's embedding table is on `stream`): the table is free for the
// ReID pass after next once that step has run
void io_pipe_step_enqueued(StreamIo* h) {
    if (!(h->pipe && h->reid_stream)) return;
    BM_HIP(hipEventRecord(h->ev_embs_free[h->pipe_slot], h->stream));
    h->pipe_slot ^= 1;
}

void io_set_crop_bound(StreamIo* h, int max_total_crops) {
    if (max_total_crops < -1) throw std::runtime_error("boxmot_hip: crop bound must be >= 0, or -1 to read the count back");
    h->crop_bound = max_total_crops;
}
// after a stream synchronisation: a step whose crops exceeded the declared bound left detections without an embedding
// (latched on the host when a bounded step is queued, not read off the CURRENT bound: a caller that runs bounded steps and then
// sets the bound back to -1 before synchronising still sees the overflow of the earlier steps)
void io_check_crop_bound(StreamIo* h) {
    if (!h->bounded_step_pending || !h->d_crop_count) return;
    h->bounded_step_pending = false;
    int flag = 0;
    BM_HIP(hipMemcpy(&flag, h->d_crop_count + 1, 4, hipMemcpyDeviceToHost));
    if (flag) {
        BM_HIP(hipMemset(h->d_crop_count + 1, 0, 4));
        throw std::runtime_error("boxmot_hip: a device-resident step had more ReID crops than the bound declared with set_crop_bound");
    }
}

// After the step kernel: wait, clear the consumed warps, turn a non-zero status word into an exception, copy the rows out.
void io_read_back(StreamIo* h, int n, const int* d_status, const char* tracker, float* const* out, int out_capacity_rows, int* out_rows,
                  int s0 = 0, const int* d_n_tracks = nullptr) {
    BM_HIP(hipGetLastError());
    BM_HIP(hipMemcpyAsync(h->h_out_n.data(), h->d_out_n + s0, n * 4, hipMemcpyDeviceToHost, h->stream));
    if (d_n_tracks) BM_HIP(hipMemcpyAsync(h->h_used.data() + s0, d_n_tracks + s0, n * 4, hipMemcpyDeviceToHost, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    // an earlier device-resident step whose crops exceeded the declared bound (that frame was not stepped): reported AFTER this
    // update's read-back is complete -- its step has run on the device, so its rows are returned and its warps are cleared first
    std::string bound_msg;
    try { io_check_crop_bound(h); } catch (const std::exception& e) { bound_msg = e.what(); }
    for (int k = 0; k < n; ++k) h->h_warp_flag[s0 + k] = 0;
    const std::string status_msg = take_status(h->stream, const_cast<int*>(d_status), s0, n, tracker);
    for (int k = 0; k < n; ++k) {
        const int rows = h->h_out_n[k];
        if (rows > out_capacity_rows) throw std::runtime_error("boxmot_hip: output buffer is too small for the current frame.");
        const size_t OC = h->out_cols;
        if (rows) BM_HIP(hipMemcpy(h->h_out.data(), h->d_out + (size_t)(s0 + k) * h->cap * OC, (size_t)rows * OC * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < rows; ++r) {
            float* dst = out[k] + (size_t)r * 9;
            for (size_t q = 0; q < OC; ++q) dst[q] = h->h_out[(size_t)r * OC + q];
            if (OC < 9) dst[8] = 0.0f;
        }
        out_rows[k] = rows;
    }
    if (!status_msg.empty()) throw std::runtime_error(status_msg);      // rows of this frame are in `out` all the same
    if (!bound_msg.empty())         // (the "<tracker> stream <n>: " shape tells the host classes that this update's step ran: _lib.step_ran)
        throw std::runtime_error(std::string("boxmot_hip: ") + tracker + " stream " + std::to_string(s0) + ": an EARLIER step reported: " + bound_msg);
}

// ---------------------------------------------------------------------------
// DeepOCSORT host path
// ---------------------------------------------------------------------------
const void* docs_kernel(const BoxMOTHipDeepOcSort* h) {
    return h->is_obb ? reinterpret_cast<const void*>(deepocsort_obb_step_kernel<STEP_THREADS>)
                     : reinterpret_cast<const void*>(deepocsort_step_kernel<STEP_THREADS>);
}
void docs_set_lds(BoxMOTHipDeepOcSort* h, long lds) {
    BM_HIP(hipFuncSetAttribute(docs_kernel(h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}
void docs_launch(BoxMOTHipDeepOcSort* h, int n_streams, const bm::DocsStepArgs& a) {
    const size_t lds = (size_t)bm::docs_lap_lds_bytes(h->cap, h->nd);
    if (h->is_obb) hipLaunchKernelGGL((deepocsort_obb_step_kernel<STEP_THREADS>), dim3(n_streams), dim3(STEP_THREADS), lds, h->stream, a);
    else hipLaunchKernelGGL((deepocsort_step_kernel<STEP_THREADS>), dim3(n_streams), dim3(STEP_THREADS), lds, h->stream, a);
}

void docs_zero_state(BoxMOTHipDeepOcSort* h) {
    bm::DocsState& st = h->args.st;
    const size_t S = h->S, cap = h->cap;
    BM_HIP(hipMemsetAsync(st.frame_count, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.id_count, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.n_tracks, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.status, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.slot_used, 0, S * cap * 4, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
}

// centroid's norm_factor = np.sqrt(self.w**2 + self.h**2) (iou.py:266) of the frame size the tracker read off its first image
// (basetracker.py:175-180); 0 x 0 = not known yet
void docs_set_frame_size(BoxMOTHipDeepOcSort* h, int w, int hgt) {
    if (w < 0 || hgt < 0) throw std::runtime_error("boxmot_hip: frame size must not be negative");
    h->args.cfg.asso_diag = (w > 0 && hgt > 0) ? std::sqrt((double)((long)w * w + (long)hgt * hgt)) : 0.0;
}
// a step is about to run: `centroid` needs the frame size (host updates carry it with the image; device-resident steps do not)
void docs_need_frame_size(BoxMOTHipDeepOcSort* h, int image_rows, int image_cols) {
    if (h->args.cfg.asso_mode != BOXMOT_HIP_ASSO_CENTROID || h->args.cfg.asso_diag > 0.0) return;
    if (image_rows > 0 && image_cols > 0) { docs_set_frame_size(h, image_cols, image_rows); return; }
    throw std::runtime_error("boxmot_hip: asso_func centroid needs the frame size (config frame_w / frame_h, or a host update with an image first)");
}

void docs_build(BoxMOTHipDeepOcSort* h) {
    const BoxMOTHipDeepOcSortConfig& c = h->cfg;
    if (c.n_streams < 1 || c.max_tracks < 8 || c.max_dets < 4 || c.emb_dim < 1)
        throw std::runtime_error("boxmot_hip: invalid capacity configuration");
    if (c.max_age < 0 || c.max_age > 45)
        throw std::runtime_error("boxmot_hip: DeepOCSORT max_age must be in [0, 45] (the reference's 50-entry observation history, xysr.py:18)");
    if (c.delta_t < 1 || c.delta_t > 3) throw std::runtime_error("boxmot_hip: DeepOCSORT delta_t must be 1..3");
    if (!(c.aw_param < 1.0)) throw std::runtime_error("boxmot_hip: aw_param must be < 1");
    h->is_obb = c.is_obb != 0;
    if (h->is_obb) {
        // oriented detections: OC-SORT (ocsort.py:332 supports_obb; DeepOcSort does not): no appearance, no camera motion, and the
        // association function is the rotated IoU (detection_layout.py:25-26 turns "iou" into "iou_obb")
        if (!c.embedding_off) throw std::runtime_error("boxmot_hip: oriented detections run on OC-SORT (embedding_off = 1); DeepOCSORT takes axis-aligned boxes only");
        if (c.asso_func != BOXMOT_HIP_ASSO_IOU && c.asso_func != BOXMOT_HIP_ASSO_CENTROID)
            throw std::runtime_error("boxmot_hip: the oriented step has the rotated IoU and the centroid distance (asso_func must be BOXMOT_HIP_ASSO_IOU or _CENTROID)");
        h->det_cols = bm::obb::DOCS_DET_COLS; h->out_cols = bm::obb::DOCS_OUT_COLS;
    }
    io_allocate(h, c.n_streams, c.max_tracks, c.max_dets, c.embedding_off ? 1 : c.emb_dim, !c.embedding_off);
    bm::DocsConfigDev& d = h->args.cfg;
    d.det_thresh = c.det_thresh; d.det_thresh_f32 = (float)c.det_thresh; d.max_age = c.max_age; d.min_hits = c.min_hits;
    d.delta_t = c.delta_t; d.iou_threshold = c.iou_threshold; d.inertia = c.inertia; d.w_emb = c.w_association_emb;
    d.alpha_fixed = c.alpha_fixed_emb; d.aw_param = c.aw_param; d.q_xy = c.Q_xy_scaling; d.q_s = c.Q_s_scaling;
    d.embedding_off = c.embedding_off; d.aw_off = c.aw_off;
    if (c.use_byte && !c.embedding_off) throw std::runtime_error("boxmot_hip: use_byte is OC-SORT's option and needs embedding_off = 1");
    d.use_byte = c.use_byte ? 1 : 0; d.min_conf_f32 = (float)c.min_conf;
    if (c.asso_func < BOXMOT_HIP_ASSO_IOU || c.asso_func > BOXMOT_HIP_ASSO_CENTROID)
        throw std::runtime_error("boxmot_hip: asso_func must be one of BOXMOT_HIP_ASSO_* (iou, giou, diou, ciou, hmiou, centroid)");
    d.asso_mode = c.asso_func;
    docs_set_frame_size(h, c.frame_w, c.frame_h);
    RecAlloc table_allocator{&h->owned, &h->table_rec};
    bm::DocsSizes z{h->S, h->cap, h->nd, h->dim, h->is_obb ? 1 : 0};
    bm::docs_allocate(h->args, z, table_allocator);
    const long lds = bm::docs_lap_lds_bytes(h->cap, h->nd);
    if (lds > 120 * 1024) throw std::runtime_error("boxmot_hip: max_tracks/max_dets too large for the assignment solver's LDS state");
    docs_set_lds(h, lds);
}

// debug cost planes of the DeepOCSORT step (boxmot_hip_deepocsort_debug_costs_enable): made at the tables' current sizes
void docs_make_dbg(BoxMOTHipDeepOcSort* h) {
    if (h->args.dbg_cost) { release(h->owned, h->args.dbg_cost); release(h->owned, h->args.dbg_shape); }
    h->args.dbg_cost = zalloc<double>((size_t)h->S * bm::DOCS_DBG_PLANES * h->nd * h->cap, h->owned);
    h->args.dbg_shape = zalloc<int>((size_t)h->S * 4, h->owned);
    h->dbg_cap = h->cap; h->dbg_nd = h->nd;
}

void docs_grow(BoxMOTHipDeepOcSort* h, int new_cap, int new_nd) {
    const long lds = bm::docs_lap_lds_bytes(new_cap, new_nd);
    if (lds > 120 * 1024)
        throw std::runtime_error("boxmot_hip: " + std::to_string(new_cap) + " tracks x " + std::to_string(new_nd) +
                                 " detections per stream is beyond what the assignment solver's LDS state can hold");
    BM_HIP(hipStreamSynchronize(h->stream));
    std::unique_ptr<bm::ReidEngine> new_reid;           // first: a failure leaves the handle as it was
    if (h->reid && new_nd != h->nd) new_reid = io_new_reid(h, new_nd);
    bm::DocsStepArgs na = h->args;
    std::vector<std::pair<void*, size_t>> rec;
    RecAlloc ra{&h->owned, &rec};
    bm::docs_allocate(na, bm::DocsSizes{h->S, new_cap, new_nd, h->dim, h->is_obb ? 1 : 0}, ra);
    io_migrate(h, rec, na.sc.keep);
    h->args = na;
    const bool more_dets = new_nd != h->nd;
    h->cap = new_cap; h->nd = new_nd;
    io_alloc_sized(h);
    if (new_reid) h->reid = std::move(new_reid);
    docs_set_lds(h, lds);
    if (h->args.dbg_cost) docs_make_dbg(h);
}

// Stage the inputs of the first n streams (ReID on every detection above det_thresh when embeddings are not supplied,
// deepocsort.py:337-345), step, read back.
void docs_host_update(BoxMOTHipDeepOcSort* h, int n, const StreamIn* in, int det_cols, int emb_cols, int image_rows,
                      int image_cols, int image_channels, float* const* out, int out_capacity_rows, int* out_rows, int s0 = 0,
                      int frame_count = -1, int* id_count_inout = nullptr) {
    const bool want_emb = !h->cfg.embedding_off;
    io_make_room(h, s0, n, in, h->args.st.n_tracks, [&](int cap, int nd) { docs_grow(h, cap, nd); },
                 [](int cap, int nd) { return bm::docs_lap_lds_bytes(cap, nd) <= 120 * 1024; });
    const bool any_warp = io_stage(h, n, in, det_cols, emb_cols, want_emb, image_rows, image_cols, image_channels, out_capacity_rows,
                                   (double)(float)h->cfg.det_thresh, 0, s0);
    // per-class fan-out (basetracker.py:223-263): the frame counter is rewound for every class and the id counter
    // (KalmanBoxTracker.count, deepocsort.py:57,117-118) is shared by all of them
    if (frame_count >= 0) BM_HIP(hipMemcpyAsync(h->args.st.frame_count + s0, &frame_count, 4, hipMemcpyHostToDevice, h->stream));
    if (id_count_inout) BM_HIP(hipMemcpyAsync(h->args.st.id_count + s0, id_count_inout, 4, hipMemcpyHostToDevice, h->stream));
    bm::DocsStepArgs a = h->args;
    a.dets = h->d_dets; a.n_dets = h->d_ndets; a.embs = want_emb ? h->d_embs : nullptr;
    a.warp = any_warp ? h->d_warp : nullptr; a.warp_flag = any_warp ? h->d_warp_flag : nullptr;
    a.out = h->d_out; a.out_n = h->d_out_n; a.stream_base = s0;
    docs_launch(h, n, a);
    io_read_back(h, n, h->args.st.status, "DeepOCSORT", out, out_capacity_rows, out_rows, s0, h->args.st.n_tracks);
    if (id_count_inout) BM_HIP(hipMemcpy(id_count_inout, h->args.st.id_count + s0, 4, hipMemcpyDeviceToHost));
}

// ---------------------------------------------------------------------------
// StrongSORT host path
// ---------------------------------------------------------------------------
constexpr int SS_BANK_THREADS = 256;

void ss_zero_state(BoxMOTHipStrongSort* h) {
    bm::SsState& st = h->args.st;
    const size_t S = h->S, cap = h->cap;
    std::vector<int> ones(S, 1);
    BM_HIP(hipMemsetAsync(st.frame_count, 0, S * 4, h->stream));
    BM_HIP(hipMemcpyAsync(st.next_id, ones.data(), S * 4, hipMemcpyHostToDevice, h->stream));     // Tracker._next_id = 1, tracker.py:59
    BM_HIP(hipMemsetAsync(st.n_tracks, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.status, 0, S * 4, h->stream));
    BM_HIP(hipMemsetAsync(st.slot_used, 0, S * cap * 4, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
}

void ss_build(BoxMOTHipStrongSort* h) {
    const BoxMOTHipStrongSortConfig& c = h->cfg;
    if (c.n_streams < 1 || c.max_tracks < 8 || c.max_dets < 4 || c.emb_dim < 1)
        throw std::runtime_error("boxmot_hip: invalid capacity configuration");
    if (c.nn_budget < 1 || c.nn_budget > 1024) throw std::runtime_error("boxmot_hip: StrongSORT nn_budget must be in [1, 1024] (None is not supported)");
    if (c.n_init < 1 || c.max_age < 0) throw std::runtime_error("boxmot_hip: invalid n_init / max_age");
    { const char* v = std::getenv("BOXMOT_HIP_SS_BANK"); h->bank_valu = v && std::strcmp(v, "valu") == 0; }
    io_allocate(h, c.n_streams, c.max_tracks, c.max_dets, c.emb_dim, true);
    bm::SsConfigDev& d = h->args.cfg;
    d.min_conf = c.min_conf; d.max_cos_dist = c.max_cos_dist; d.max_iou_dist = c.max_iou_dist; d.mc_lambda = c.mc_lambda;
    d.ema_alpha_f32 = (float)c.ema_alpha; d.one_minus_alpha_f32 = (float)(1 - c.ema_alpha);
    d.max_age = c.max_age; d.n_init = c.n_init; d.budget = c.nn_budget;
    RecAlloc table_allocator{&h->owned, &h->table_rec};
    bm::SsSizes z{h->S, h->cap, h->nd, h->dim, c.nn_budget};
    bm::ss_allocate(h->args, z, table_allocator);
    const long lds = bm::ss_lsa_lds_bytes(h->cap > h->nd ? h->cap : h->nd);
    if (lds > 120 * 1024) throw std::runtime_error("boxmot_hip: max_tracks/max_dets too large for the assignment solver's LDS state");
    BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(strongsort_step_kernel<STEP_THREADS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(strongsort_step_kernel<SS_STEP_THREADS_BIG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ss_zero_state(h);
}

// debug cost planes of the StrongSORT step (boxmot_hip_strongsort_debug_costs_enable)
void ss_make_dbg(BoxMOTHipStrongSort* h) {
    if (h->args.dbg_cost) { release(h->owned, h->args.dbg_cost); release(h->owned, h->args.dbg_shape); }
    const size_t big = h->cap > h->nd ? h->cap : h->nd;
    h->args.dbg_cost = zalloc<double>((size_t)h->S * 4 * big * big, h->owned);
    h->args.dbg_shape = zalloc<int>((size_t)h->S * 4, h->owned);
    h->dbg_big = (int)big;
}

void ss_grow(BoxMOTHipStrongSort* h, int new_cap, int new_nd) {
    const long lds = bm::ss_lsa_lds_bytes(new_cap > new_nd ? new_cap : new_nd);
    if (lds > 120 * 1024)
        throw std::runtime_error("boxmot_hip: " + std::to_string(new_cap) + " tracks x " + std::to_string(new_nd) +
                                 " detections per stream is beyond what the assignment solver's LDS state can hold");
    BM_HIP(hipStreamSynchronize(h->stream));
    bm::SsStepArgs na = h->args;
    std::vector<std::pair<void*, size_t>> rec;
    RecAlloc ra{&h->owned, &rec};
    bm::ss_allocate(na, bm::SsSizes{h->S, new_cap, new_nd, h->dim, h->cfg.nn_budget}, ra);
    io_migrate(h, rec, na.sc.app);
    h->args = na;
    const bool more_dets = new_nd != h->nd;
    h->cap = new_cap; h->nd = new_nd;
    io_alloc_sized(h);
    if (more_dets && h->reid) io_make_reid(h);
    BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(strongsort_step_kernel<STEP_THREADS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    BM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(strongsort_step_kernel<SS_STEP_THREADS_BIG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (h->args.dbg_cost) ss_make_dbg(h);
}

#ifdef BM_SS_PROF
// tools only: phase clocks of strongsort_step_kernel (workgroup 0), cleared on read
extern "C" int boxmot_hip_debug_ss_prof(unsigned long long* out16) {
    unsigned long long zero[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(bm::g_ss_prof), sizeof(zero)) != hipSuccess) return 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(bm::g_ss_prof), zero, sizeof(zero)) == hipSuccess;
}
#endif

// detection norms -> sample-bank distances of every confirmed track (state BEFORE this frame's step) -> frame step
void ss_launch(BoxMOTHipStrongSort* h, const bm::SsStepArgs& a, int n) {
    hipLaunchKernelGGL((strongsort_detnorm_kernel<SS_BANK_THREADS>), dim3(n, h->nd >= 128 ? 16 : 1), dim3(SS_BANK_THREADS), 0, h->stream, a);
    // sample-bank distances: fp32 matrix pipe (budgets up to 112 samples; bit-identical sums), else / on request the scalar-FMA kernel
    if (h->cfg.nn_budget <= bm::SS_MT * 16 && !h->bank_valu)
        hipLaunchKernelGGL((strongsort_bank_mfma_kernel<SS_BANK_THREADS>), dim3(h->cap, n), dim3(SS_BANK_THREADS), 0, h->stream, a);
    else
        hipLaunchKernelGGL((strongsort_bank_kernel<SS_BANK_THREADS>), dim3(h->cap, n), dim3(SS_BANK_THREADS), 0, h->stream, a);
    // one workgroup per stream: 16 waves when the track table is large (the per-match / per-track phases run a wave per item)
    if (h->cap >= 1024)
        hipLaunchKernelGGL((strongsort_step_kernel<SS_STEP_THREADS_BIG>), dim3(n), dim3(SS_STEP_THREADS_BIG),
                           (size_t)bm::ss_lsa_lds_bytes(h->cap > h->nd ? h->cap : h->nd), h->stream, a);
    else
        hipLaunchKernelGGL((strongsort_step_kernel<STEP_THREADS>), dim3(n), dim3(STEP_THREADS),
                           (size_t)bm::ss_lsa_lds_bytes(h->cap > h->nd ? h->cap : h->nd), h->stream, a);
}

// Stage the inputs of the first n streams (ReID on every detection with conf >= min_conf when embeddings are not supplied,
// strongsort.py:74-91), step, read back.
void ss_host_update(BoxMOTHipStrongSort* h, int n, const StreamIn* in, int det_cols, int emb_cols, int image_rows,
                    int image_cols, int image_channels, float* const* out, int out_capacity_rows, int* out_rows) {
    io_make_room(h, 0, n, in, h->args.st.n_tracks, [&](int cap, int nd) { ss_grow(h, cap, nd); },
                 [](int cap, int nd) { return bm::ss_lsa_lds_bytes(cap > nd ? cap : nd) <= 120 * 1024; });
    io_stage(h, n, in, det_cols, emb_cols, true, image_rows, image_cols, image_channels, out_capacity_rows, h->cfg.min_conf, 1);
    bm::SsStepArgs a = h->args;
    a.dets = h->d_dets; a.n_dets = h->d_ndets; a.embs = h->d_embs; a.warp = h->d_warp;     // identity where no warp is pending
    a.out = h->d_out; a.out_n = h->d_out_n; a.stream_base = 0;
    ss_launch(h, a, n);
    io_read_back(h, n, h->args.st.status, "StrongSORT", out, out_capacity_rows, out_rows, 0, h->args.st.n_tracks);
}

}  // namespace

extern "C" {

const char* boxmot_hip_last_error(void) { return g_last_error.c_str(); }

int boxmot_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int boxmot_hip_botsort_device(BoxMOTHipBotSort* handle) { return handle ? handle->device : -1; }

void boxmot_hip_botsort_default_config(BoxMOTHipBotSortConfig* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->track_high_thresh = 0.5; c->track_low_thresh = 0.1; c->new_track_thresh = 0.6; c->track_buffer = 30;
    c->match_thresh = 0.8; c->proximity_thresh = 0.5; c->appearance_thresh = 0.25; c->cmc_method = nullptr;
    c->frame_rate = 30; c->fuse_first_associate = 0; c->with_reid = 1; c->max_obs = 50;
    c->reid_model_path = nullptr; c->reid_preprocess = nullptr;
    c->second_match_thresh = 0.5; c->unconfirmed_match_thresh = 0.7; c->unconfirmed_emb_scale = 2.0;
    c->removed_stracks_buffer = 100;
    c->n_streams = 1; c->max_tracks = 1024; c->max_dets = 256; c->emb_dim = 512; c->n_class_lists = 1;
}

void boxmot_hip_bytetrack_default_config(BoxMOTHipBotSortConfig* c) {
    if (!c) return;
    boxmot_hip_botsort_default_config(c);
    c->tracker_kind = 1;
    c->track_low_thresh = 0.1; c->track_high_thresh = 0.45; c->new_track_thresh = 0.45;     // min_conf, track_thresh, det_thresh
    c->match_thresh = 0.8; c->track_buffer = 25; c->frame_rate = 30;
    c->second_match_thresh = 0.5; c->unconfirmed_match_thresh = 0.7;                          // bytetrack.py:338, 358
    c->with_reid = 0; c->fuse_first_associate = 1; c->emb_dim = 1;
}

BoxMOTHipBotSort* boxmot_hip_botsort_create(const BoxMOTHipBotSortConfig* config) {
    BoxMOTHipBotSort* h = nullptr;
    const int ok = guard([&]() {
        if (config == nullptr) throw std::runtime_error("boxmot_hip BoT-SORT config is required.");
        require_device();
        h = new BoxMOTHipBotSort();
        h->cfg = *config;
        if (config->reid_model_path) h->reid_path = config->reid_model_path;
        h->cfg.reid_model_path = nullptr;
        build(h);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_botsort_destroy(BoxMOTHipBotSort* handle) { destroy_on(handle); }

int boxmot_hip_botsort_reset(BoxMOTHipBotSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is null.");
        zero_state(handle);
        handle->h_used.assign(handle->S, 0);
    });
}

int boxmot_hip_botsort_reserve(BoxMOTHipBotSort* handle, int max_tracks, int max_dets) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is null.");
        const int cap = max_tracks > handle->cap ? (max_tracks + 63) / 64 * 64 : handle->cap;
        const int nd = max_dets > handle->nd ? (max_dets + 63) / 64 * 64 : handle->nd;
        if (cap != handle->cap || nd != handle->nd) grow_tables(handle, cap, nd);
    });
}

int boxmot_hip_botsort_capacity(BoxMOTHipBotSort* handle, int* max_tracks, int* max_dets, int* n_grows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is null.");
        if (max_tracks) *max_tracks = handle->cap;
        if (max_dets) *max_dets = handle->nd;
        if (n_grows) *n_grows = handle->n_grows;
    });
}

int boxmot_hip_botsort_update(BoxMOTHipBotSort* handle, const float* dets, int det_rows, int det_cols,
                              const float* embs, int emb_rows, int emb_cols, const uint8_t* image, int image_rows,
                              int image_cols, int image_channels, float* out_tracks, int out_capacity_rows,
                              int out_cols, int* out_rows, int* out_is_obb) {
    return boxmot_hip_botsort_update_stream(handle, 0, 0, -1, dets, det_rows, det_cols, embs, emb_rows, emb_cols, image,
                                            image_rows, image_cols, image_channels, out_tracks, out_capacity_rows,
                                            out_cols, out_rows, out_is_obb);
}

int boxmot_hip_botsort_update_stream(BoxMOTHipBotSort* handle, int stream, int class_list, int frame_count,
                                     const float* dets, int det_rows, int det_cols, const float* embs, int emb_rows,
                                     int emb_cols, const uint8_t* image, int image_rows, int image_cols,
                                     int image_channels, float* out_tracks, int out_capacity_rows, int out_cols,
                                     int* out_rows, int* out_is_obb) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (!out_rows || !out_is_obb) throw std::runtime_error("Output pointers are null.");
        if (out_cols != 9) throw std::runtime_error("boxmot_hip live tracking expects an output buffer with 9 columns.");
        if (embs != nullptr && emb_rows != det_rows) throw std::runtime_error("Detection and embedding row counts must match.");
        if (image_rows <= 0 || image_cols <= 0) throw std::runtime_error("Image dimensions must be positive.");
        StreamIn in{dets, det_rows, (embs && emb_cols > 0) ? embs : nullptr, image};
        host_update_one(handle, stream, class_list, frame_count, in, det_cols, emb_cols, image_rows, image_cols,
                        image_channels, out_tracks, out_capacity_rows, out_rows);
        *out_is_obb = handle->is_obb ? 1 : 0;
    });
}

int boxmot_hip_botsort_update_batch(BoxMOTHipBotSort* handle, int n_streams, const float* const* dets,
                                    const int* det_rows, const float* const* embs, int emb_cols,
                                    const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
                                    float* const* out_tracks, int out_capacity_rows, int* out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (n_streams < 1 || n_streams > handle->S) throw std::runtime_error("boxmot_hip: n_streams out of range");
        if (!dets || !det_rows || !out_tracks || !out_rows) throw std::runtime_error("boxmot_hip: null batch pointers");
        std::vector<StreamIn> in(n_streams);
        for (int s = 0; s < n_streams; ++s)
            in[s] = StreamIn{dets[s], det_rows[s], (embs && emb_cols > 0) ? embs[s] : nullptr, images ? images[s] : nullptr};
        host_update(handle, 0, n_streams, in.data(), handle->det_cols(), emb_cols, image_rows, image_cols, image_channels, nullptr, nullptr,
                    out_tracks, out_capacity_rows, out_rows);
    });
}

int boxmot_hip_botsort_update_batch_frames(BoxMOTHipBotSort* handle, int n_streams, const float* const* dets,
                                           const int* det_rows, const float* const* embs, int emb_cols,
                                           const uint8_t* const* d_frames, int image_rows, int image_cols,
                                           float* const* out_tracks, int out_capacity_rows, int* out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (n_streams < 1 || n_streams > handle->S) throw std::runtime_error("boxmot_hip: n_streams out of range");
        if (!dets || !det_rows || !out_tracks || !out_rows) throw std::runtime_error("boxmot_hip: null batch pointers");
        if (!d_frames || image_rows < 1 || image_cols < 1) throw std::runtime_error("boxmot_hip: update_batch_frames needs device frames");
        std::vector<StreamIn> in(n_streams);
        for (int s = 0; s < n_streams; ++s) in[s] = StreamIn{dets[s], det_rows[s], (embs && emb_cols > 0) ? embs[s] : nullptr, nullptr};
        host_update(handle, 0, n_streams, in.data(), handle->det_cols(), emb_cols, image_rows, image_cols, 3, nullptr, nullptr, out_tracks,
                    out_capacity_rows, out_rows, d_frames);
    });
}

int boxmot_hip_botsort_step_device(BoxMOTHipBotSort* handle, const float* d_dets, const int* d_det_rows,
                                   const float* d_embs, const uint8_t* const* d_frames, int image_rows, int image_cols,
                                   float* d_out, int* d_out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (!d_dets || !d_det_rows || !d_out || !d_out_rows) throw std::runtime_error("boxmot_hip: null device pointers");
        const float* embs = d_embs;
        bool piped = false;
        if (handle->cfg.with_reid && d_embs == nullptr) {
            if (handle->is_obb) throw std::runtime_error("boxmot_hip: an oriented-box handle with with_reid = 1 needs d_embs");
            if (!d_frames) throw std::runtime_error("boxmot_hip: with_reid needs d_embs or d_frames");
            // pipeline stage 1: this frame's ReID on `reid_stream`, into the table the step before last has finished reading
            const bool pipe = handle->pipe && handle->reid_stream;
            float* tab = (pipe && handle->pipe_slot) ? handle->d_embs_alt : handle->d_embs;
            if (pipe) {
                hipStream_t rs = handle->reid_stream;
                if (handle->engine_on_main || handle->stream_exposed) {       // (as io_device_reid: engine used on `stream`, or the caller holds `stream`)
                    BM_HIP(hipEventRecord(handle->ev_main, handle->stream));
                    BM_HIP(hipStreamWaitEvent(rs, handle->ev_main, 0));
                    handle->engine_on_main = false;
                }
                BM_HIP(hipStreamWaitEvent(rs, handle->ev_embs_free[handle->pipe_slot], 0));
                run_reid(handle, 0, handle->S, d_dets, d_det_rows, d_frames, image_rows, image_cols, tab, rs);
                BM_HIP(hipEventRecord(handle->ev_reid_done[handle->pipe_slot], rs));
                BM_HIP(hipStreamWaitEvent(handle->stream, handle->ev_reid_done[handle->pipe_slot], 0));
                piped = true;
            } else {
                run_reid(handle, 0, handle->S, d_dets, d_det_rows, d_frames, image_rows, image_cols, tab);
            }
            embs = tab;
        }
        // cmc_method = "sof" / "ecc": the estimator runs inside the step here too, on the device-resident frames, for every stream
        // that was not given a warp with set_warp (the estimate comes back to the host: one synchronisation per step; callers that
        // want none estimate ahead with boxmot_hip_sof_* / boxmot_hip_ecc_* and set_warp, or create the handle with cmc_method "none")
        if (handle->use_sof || handle->use_ecc) {
            BoxMOTHipBotSort* h = handle;
            bool missing = false;
            for (int s = 0; s < h->S; ++s) missing = missing || h->h_warp_flag[s] == 0;
            if (missing) {
                if (!d_frames || image_rows < 1 || image_cols < 1)
                    throw std::runtime_error("boxmot_hip: cmc_method=sof/ecc in a device-resident step needs d_frames (or a warp per stream from set_warp)");
                if (h->use_sof) {
                    if (!h->sof) { h->sof.reset(new BoxMOTHipSof()); sof_init(h->sof.get(), h->S, image_rows, image_cols, 0.15, 8, 0.2, 3.0, h->stream); }
                    if (h->sof->rows != image_rows || h->sof->cols != image_cols) throw std::runtime_error("boxmot_hip: frame size changed between updates");
                } else {
                    if (!h->ecc) { h->ecc.reset(new BoxMOTHipEcc()); ecc_init(h->ecc.get(), h->S, image_rows, image_cols, 0.15, 1e-5, 100, h->stream); }
                    if (h->ecc->rows != image_rows || h->ecc->cols != image_cols) throw std::runtime_error("boxmot_hip: frame size changed between updates");
                }
                std::vector<double> w(6);
                for (int s = 0; s < h->S; ++s) {
                    if (h->h_warp_flag[s]) continue;
                    if (h->use_sof && h->is_obb) {
                        hipLaunchKernelGGL(obb_enclosing_boxes_kernel, dim3(1), dim3(256), 0, h->stream, d_dets, d_det_rows, h->nd, h->d_cmc_boxes, s);
                        sof_run(h->sof.get(), s, 1, d_frames + s, h->d_cmc_boxes + (size_t)s * h->nd * 4, d_det_rows + s, h->nd, 4, w.data(), nullptr);
                    } else if (h->use_sof) sof_run(h->sof.get(), s, 1, d_frames + s, d_dets + (size_t)s * h->nd * bm::DET_COLS, d_det_rows + s, h->nd, bm::DET_COLS, w.data(), nullptr);
                    else ecc_run_one(h->ecc.get(), s, d_frames + s, w.data(), nullptr);
                    for (int k = 0; k < 6; ++k) h->h_warp[(size_t)s * 6 + k] = w[k];
                    h->h_warp_flag[s] = 1;
                }
            }
        }
        // warps set with boxmot_hip_botsort_set_warp since the last step are consumed by this one
        bool any_warp = false;
        for (int s = 0; s < handle->S; ++s) any_warp = any_warp || handle->h_warp_flag[s] != 0;
        if (any_warp) {
            BM_HIP(hipMemcpyAsync(handle->d_warp, handle->h_warp.data(), (size_t)handle->S * 6 * 8, hipMemcpyHostToDevice, handle->stream));
            BM_HIP(hipMemcpyAsync(handle->d_warp_flag, handle->h_warp_flag.data(), handle->S * 4, hipMemcpyHostToDevice, handle->stream));
            // the staging vectors are reused by the next set_warp: the copies must have left the host before returning
            BM_HIP(hipStreamSynchronize(handle->stream));
        }
        launch_step(handle, 0, handle->S, d_dets, d_det_rows, handle->cfg.with_reid ? embs : nullptr, nullptr, nullptr,
                    d_out, d_out_rows, any_warp);
        if (piped) {        // pipeline stage 2 enqueued: the table is free for the ReID pass after next once this step has run
            BM_HIP(hipEventRecord(handle->ev_embs_free[handle->pipe_slot], handle->stream));
            handle->pipe_slot ^= 1;
        }
        for (int s = 0; s < handle->S; ++s) { handle->h_warp_flag[s] = 0; handle->h_used[s] = -1; }      // slot counts: unknown to the host now
    });
}

int boxmot_hip_botsort_set_warp(BoxMOTHipBotSort* handle, int stream, const double* warp_2x3) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is null.");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (warp_2x3 == nullptr) { handle->h_warp_flag[stream] = 0; return; }
        for (int k = 0; k < 6; ++k) {
            if (!std::isfinite(warp_2x3[k])) throw std::runtime_error("boxmot_hip: camera-motion warp has non-finite entries");
            handle->h_warp[(size_t)stream * 6 + k] = warp_2x3[k];
        }
        handle->h_warp_flag[stream] = 1;
    });
}

int boxmot_hip_botsort_synchronize(BoxMOTHipBotSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        BM_HIP(hipStreamSynchronize(handle->stream));
    });
}

int boxmot_hip_botsort_timer_start(BoxMOTHipBotSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        BM_HIP(hipEventRecord(handle->timer_ev[0], handle->stream));
    });
}

int boxmot_hip_botsort_timer_stop_ms(BoxMOTHipBotSort* handle, double* out_ms) {
    return guard_on(handle, [&]() {
        if (!handle || !out_ms) throw std::runtime_error("boxmot_hip: null argument");
        BM_HIP(hipEventRecord(handle->timer_ev[1], handle->stream));
        BM_HIP(hipEventSynchronize(handle->timer_ev[1]));
        float ms = 0;
        BM_HIP(hipEventElapsedTime(&ms, handle->timer_ev[0], handle->timer_ev[1]));
        *out_ms = ms;
    });
}

int boxmot_hip_botsort_reid_kernel_ms(BoxMOTHipBotSort* handle, double* out_ms, int* out_launches) {
    return guard_on(handle, [&]() {
        if (!handle || !out_ms || !out_launches) throw std::runtime_error("boxmot_hip: null argument");
        *out_ms = 0; *out_launches = 0;
        if (handle->reid) handle->reid->drain_kernel_timing(*out_ms, *out_launches);
    });
}

int boxmot_hip_botsort_phase_clocks(BoxMOTHipBotSort* handle, long long* out16) {
    return guard_on(handle, [&]() {
        if (!handle || !out16) throw std::runtime_error("boxmot_hip: null argument");
        BM_HIP(hipStreamSynchronize(handle->stream));
        BM_HIP(hipMemcpy(out16, handle->d_phase_clock, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    });
}

void* boxmot_hip_botsort_stream(BoxMOTHipBotSort* handle) { if (handle) handle->stream_exposed = true; return handle ? (void*)handle->stream : nullptr; }

int boxmot_hip_botsort_status(BoxMOTHipBotSort* handle, int* out_status, int capacity) {
    return guard_on(handle, [&]() {
        if (!handle || !out_status) throw std::runtime_error("boxmot_hip: null argument");
        const int n = capacity < handle->S ? capacity : handle->S;
        BM_HIP(hipStreamSynchronize(handle->stream));
        BM_HIP(hipMemcpy(out_status, handle->args.st.status, n * 4, hipMemcpyDeviceToHost));
    });
}

int boxmot_hip_botsort_set_reid_blob(BoxMOTHipBotSort* handle, const float* blob, long n_floats) {
    return guard_on(handle, [&]() {
        if (!handle || !blob || n_floats <= 0) throw std::runtime_error("boxmot_hip: null argument");
        // the handle keeps a host copy: growing max_dets (reserve, or an update with more detections) rebuilds the engine from it
        std::vector<float> keep(blob, blob + n_floats), old;
        old.swap(handle->reid_blob);
        handle->reid_blob.swap(keep);
        try {
            handle->reid = new_reid_engine(handle, handle->nd);
        } catch (...) {
            handle->reid_blob.swap(old);
            throw;
        }
    });
}

int boxmot_hip_botsort_set_reid_mode(BoxMOTHipBotSort* handle, int mode) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null argument");
        if (handle->reid) handle->reid->set_mode(mode);
        handle->reid_mode = mode;
    });
}

static int get_ms(BoxMOTHipBotSort* h, double* out, double BoxMOTHipBotSort::*field, bool sum_reid) {
    return guard_on(h, [&]() {
        if (!h) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (!out) throw std::runtime_error("Output timing pointer is null.");
        *out = sum_reid ? h->last_reid_pre_ms + h->last_reid_proc_ms : h->*field;
    });
}
int boxmot_hip_botsort_last_reid_time_ms(BoxMOTHipBotSort* h, double* o) { return get_ms(h, o, &BoxMOTHipBotSort::last_reid_pre_ms, true); }
int boxmot_hip_botsort_last_reid_preprocess_time_ms(BoxMOTHipBotSort* h, double* o) { return get_ms(h, o, &BoxMOTHipBotSort::last_reid_pre_ms, false); }
int boxmot_hip_botsort_last_reid_process_time_ms(BoxMOTHipBotSort* h, double* o) { return get_ms(h, o, &BoxMOTHipBotSort::last_reid_proc_ms, false); }
int boxmot_hip_botsort_last_reid_postprocess_time_ms(BoxMOTHipBotSort* h, double* o) {
    return guard_on(h, [&]() {
        if (!h) throw std::runtime_error("boxmot_hip BoT-SORT handle is not initialized.");
        if (!o) throw std::runtime_error("Output timing pointer is null.");
        *o = 0.0;   // L2 normalisation is fused into the head kernel
    });
}
int boxmot_hip_botsort_last_track_time_ms(BoxMOTHipBotSort* h, double* o) { return get_ms(h, o, &BoxMOTHipBotSort::last_track_ms, false); }

int boxmot_hip_botsort_state_dump(BoxMOTHipBotSort* handle, int stream, int which, int class_list, int* ints,
                                  double* kf, float* smooth, float* misc, int* out_rows, int* out_frame_count,
                                  int* out_id_count) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        BM_HIP(hipStreamSynchronize(handle->stream));
        const bm::BotSortState& st = handle->args.st;
        const size_t cap = handle->cap, dim = handle->dim, s = stream;
        int n = 0;
        std::vector<int> list(cap);
        if (which == 0) {
            BM_HIP(hipMemcpy(&n, st.n_active + s * handle->n_lists + class_list, 4, hipMemcpyDeviceToHost));
            BM_HIP(hipMemcpy(list.data(), st.active_list + (s * handle->n_lists + class_list) * cap, cap * 4, hipMemcpyDeviceToHost));
        } else {
            BM_HIP(hipMemcpy(&n, st.n_lost + s, 4, hipMemcpyDeviceToHost));
            BM_HIP(hipMemcpy(list.data(), st.lost_list + s * cap, cap * 4, hipMemcpyDeviceToHost));
        }
        auto pull_i = [&](const int* src) { std::vector<int> v(cap); BM_HIP(hipMemcpy(v.data(), src + s * cap, cap * 4, hipMemcpyDeviceToHost)); return v; };
        auto pull_f = [&](const float* src) { std::vector<float> v(cap); BM_HIP(hipMemcpy(v.data(), src + s * cap, cap * 4, hipMemcpyDeviceToHost)); return v; };
        const auto id = pull_i(st.id), state = pull_i(st.state), act = pull_i(st.is_activated), fid = pull_i(st.frame_id),
                   sf = pull_i(st.start_frame), tl = pull_i(st.tracklet_len);
        const auto conf = pull_f(st.conf), cls = pull_f(st.cls), di = pull_f(st.det_ind);
        const size_t KS = handle->kf_stride();        // 72 doubles per track (mean 8 + cov 64), 110 for oriented boxes (10 + 100)
        std::vector<double> kfall(cap * KS);
        BM_HIP(hipMemcpy(kfall.data(), st.kf + s * cap * KS, kfall.size() * 8, hipMemcpyDeviceToHost));
        std::vector<float> sm;
        if (smooth) { sm.resize(cap * dim); BM_HIP(hipMemcpy(sm.data(), st.smooth + s * cap * dim, sm.size() * 4, hipMemcpyDeviceToHost)); }
        for (int r = 0; r < n; ++r) {
            const int sl = list[r];
            if (ints) { int* o = ints + r * 6; o[0] = id[sl]; o[1] = state[sl]; o[2] = act[sl]; o[3] = fid[sl]; o[4] = sf[sl]; o[5] = tl[sl]; }
            if (kf) std::memcpy(kf + (size_t)r * KS, kfall.data() + (size_t)sl * KS, KS * 8);
            if (smooth) std::memcpy(smooth + (size_t)r * dim, sm.data() + (size_t)sl * dim, dim * 4);
            if (misc) { misc[r * 3] = conf[sl]; misc[r * 3 + 1] = cls[sl]; misc[r * 3 + 2] = di[sl]; }
        }
        *out_rows = n;
        if (out_frame_count) BM_HIP(hipMemcpy(out_frame_count, st.frame_count + s, 4, hipMemcpyDeviceToHost));
        if (out_id_count) BM_HIP(hipMemcpy(out_id_count, st.id_count + s, 4, hipMemcpyDeviceToHost));
    });
}

int boxmot_hip_botsort_debug_costs_enable(BoxMOTHipBotSort* handle, int on) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null argument");
        if (on && handle->is_obb) throw std::runtime_error("boxmot_hip: debug_costs is built for the axis-aligned frame step only");
        BM_HIP(hipStreamSynchronize(handle->stream));
        handle->dbg_costs = on != 0;
    });
}

int boxmot_hip_botsort_debug_costs(BoxMOTHipBotSort* handle, int stream, int stage, int plane, double* out, long out_capacity,
                                   int* out_rows, int* out_cols) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows || !out_cols) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (stage < 0 || stage >= bm::DBG_STAGES || plane < 0 || plane >= bm::DBG_PLANES)
            throw std::runtime_error("boxmot_hip: debug_costs stage must be 0..2 and plane 0..2");
        if (!handle->dbg_costs || !handle->d_dbg_cost)
            throw std::runtime_error("boxmot_hip: debug_costs needs boxmot_hip_botsort_debug_costs_enable(handle, 1) before the update");
        BM_HIP(hipStreamSynchronize(handle->stream));
        int shape[2] = {0, 0};
        BM_HIP(hipMemcpy(shape, handle->d_dbg_shape + ((size_t)stream * bm::DBG_STAGES + stage) * 2, 8, hipMemcpyDeviceToHost));
        const int R = shape[0], C = shape[1];
        *out_rows = R; *out_cols = C;
        if (R == 0 || C == 0) return;
        if (!out || out_capacity < (long)R * C) throw std::runtime_error("boxmot_hip: debug_costs output capacity is smaller than rows x cols");
        const size_t cap = handle->dbg_cap, nd = handle->dbg_nd;
        std::vector<double> m(nd * cap);
        BM_HIP(hipMemcpy(m.data(), handle->d_dbg_cost + (((size_t)stream * bm::DBG_STAGES + stage) * bm::DBG_PLANES + plane) * nd * cap,
                         m.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r)                       // the device stores detection-major; the caller gets the reference's (tracks, dets)
            for (int c = 0; c < C; ++c) out[(size_t)r * C + c] = m[(size_t)c * cap + r];
    });
}

// ---- ReID C ABI ----
BoxMOTHipReID* boxmot_hip_reid_create(const char* model_path, const float* blob, long n_floats, int max_crops) {
    BoxMOTHipReID* h = nullptr;
    const int ok = guard([&]() {
        require_device();
        if (max_crops < 1) throw std::runtime_error("boxmot_hip: max_crops must be >= 1");
        h = new BoxMOTHipReID();
        BM_HIP(hipStreamCreate(&h->stream));
        if (blob) h->engine.reset(new bm::ReidEngine(blob, n_floats, max_crops));
        else if (model_path) {
            const std::vector<float> v = bm::read_blob_file(model_path);
            h->engine.reset(new bm::ReidEngine(v.data(), (long)v.size(), max_crops));
        } else throw std::runtime_error("boxmot_hip: ReID needs a weight blob or a model path");
        h->d_frames = dev_alloc<const uint8_t*>(1, h->owned);
        h->d_crop_stream = zalloc<int>(max_crops, h->owned);
        h->d_boxes = dev_alloc<float>((size_t)max_crops * 4, h->owned);
        h->d_feat = dev_alloc<float>((size_t)max_crops * h->engine->feature_dim(), h->owned);
        h->d_obb = dev_alloc<double>((size_t)max_crops * 8, h->owned);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_reid_destroy(BoxMOTHipReID* handle) { destroy_on(handle); }

int boxmot_hip_reid_feature_dim(BoxMOTHipReID* handle) { return handle && handle->engine ? handle->engine->feature_dim() : 0; }

int boxmot_hip_reid_set_preprocess(BoxMOTHipReID* handle, const char* name) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null ReID handle");
        if (name == nullptr || std::strcmp(name, "resize") == 0) handle->engine->set_preprocess(0);
        else if (std::strcmp(name, "resize_pad") == 0) handle->engine->set_preprocess(1);
        else throw std::runtime_error("boxmot_hip: unknown ReID preprocess (have: resize, resize_pad)");
    });
}

int boxmot_hip_reid_set_mode(BoxMOTHipReID* handle, int mode) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null ReID handle");
        handle->engine->set_mode(mode);
    });
}

static void reid_stage(BoxMOTHipReID* h, const uint8_t* image, int rows, int cols, int channels, const float* boxes,
                       int n, int box_cols) {
    if (!h) throw std::runtime_error("boxmot_hip: null ReID handle");
    if (!image) throw std::runtime_error("Image data pointer is null.");
    if (channels != 3) throw std::runtime_error("boxmot_hip: ReID needs a 3-channel uint8 BGR image");
    if (rows <= 0 || cols <= 0) throw std::runtime_error("Image dimensions must be positive.");
    if (n < 0 || box_cols < 4) throw std::runtime_error("boxmot_hip: boxes must have at least 4 columns");
    if (n > h->engine->max_crops()) throw std::runtime_error("boxmot_hip: more boxes than max_crops");
    const size_t bytes = (size_t)rows * cols * 3;
    if (bytes > h->frame_bytes) {
        if (h->d_frame) BM_HIP(hipFree(h->d_frame));
        void* p = nullptr;
        BM_HIP(hipMalloc(&p, bytes));
        h->d_frame = static_cast<uint8_t*>(p);
        h->frame_bytes = bytes;
        BM_HIP(hipMemcpy(h->d_frames, &h->d_frame, sizeof(uint8_t*), hipMemcpyHostToDevice));
    }
    BM_HIP(hipMemcpyAsync(h->d_frame, image, bytes, hipMemcpyHostToDevice, h->stream));
    // base_backend.py:119-122, 157: rows of 5 / 7 / 9 values are oriented boxes [cx, cy, w, h, angle, ...]
    h->obb = n > 0 && (box_cols == 5 || box_cols == 7 || box_cols == 9);
    std::vector<float> b((size_t)n * 4);
    std::vector<double> geo;
    if (h->obb) {
        // _crop_obb (base_backend.py:91-117): getRotationMatrix2D about the box centre, shifted so that the centre lands on the middle
        // of the (round(w), round(h)) output; cv2.warpAffine inverts the matrix in double precision before it samples
        geo.resize((size_t)n * 8);
        const float rad2deg = 180.0f / 3.14159265358979323846f;         // np.degrees on a float32 scalar
        for (int i = 0; i < n; ++i) {
            const float* bx = boxes + (size_t)i * box_cols;
            const double cx = bx[0], cy = bx[1];
            const double bw = bx[2] > 1.0f ? (double)bx[2] : 1.0, bh = bx[3] > 1.0f ? (double)bx[3] : 1.0;
            const int ow = std::max((int)std::nearbyint(bw), 1), oh = std::max((int)std::nearbyint(bh), 1);
            const double a = (double)(bx[4] * rad2deg) * (3.14159265358979323846 / 180.0);
            const double alpha = std::cos(a), beta = std::sin(a);
            double m00 = alpha, m01 = beta, m02 = (1 - alpha) * cx - beta * cy, m10 = -beta, m11 = alpha, m12 = beta * cx + (1 - alpha) * cy;
            m02 += ow / 2.0 - cx;
            m12 += oh / 2.0 - cy;
            double D = m00 * m11 - m01 * m10;
            D = D != 0 ? 1.0 / D : 0.0;
            const double A11 = m11 * D, A22 = m00 * D;
            double* g = geo.data() + (size_t)i * 8;
            g[0] = ow; g[1] = oh;
            g[2] = A11; g[3] = m01 * (-D); g[5] = m10 * (-D); g[6] = A22;
            g[4] = -g[2] * m02 - g[3] * m12;
            g[7] = -g[5] * m02 - g[6] * m12;
            for (int q = 0; q < 4; ++q) b[i * 4 + q] = 0.f;
        }
        BM_HIP(hipMemcpyAsync(h->d_obb, geo.data(), geo.size() * 8, hipMemcpyHostToDevice, h->stream));
    } else {
        for (int i = 0; i < n; ++i)
            for (int q = 0; q < 4; ++q) b[i * 4 + q] = boxes[(size_t)i * box_cols + q];
    }
    if (n) BM_HIP(hipMemcpyAsync(h->d_boxes, b.data(), b.size() * 4, hipMemcpyHostToDevice, h->stream));
    BM_HIP(hipStreamSynchronize(h->stream));
    h->engine->set_obb_geometry(h->obb ? h->d_obb : nullptr);
}

// oriented-box geometry is valid for one staged batch only
struct ObbScope {
    BoxMOTHipReID* h;
    ~ObbScope() { if (h && h->engine) h->engine->set_obb_geometry(nullptr); }
};

int boxmot_hip_reid_compute_features(BoxMOTHipReID* handle, const uint8_t* image, int image_rows, int image_cols,
                                     int image_channels, const float* boxes, int n_boxes, int box_cols,
                                     float* out_features, int out_capacity_rows) {
    return guard_on(handle, [&]() {
        ObbScope scope{handle};
        reid_stage(handle, image, image_rows, image_cols, image_channels, boxes, n_boxes, box_cols);
        if (out_capacity_rows < n_boxes) throw std::runtime_error("boxmot_hip: feature buffer too small");
        if (n_boxes == 0) return;
        handle->engine->run(handle->d_frames, handle->d_crop_stream, handle->d_boxes, 4, n_boxes, image_cols, image_rows,
                            handle->d_feat, nullptr, handle->stream);
        BM_HIP(hipMemcpyAsync(out_features, handle->d_feat, (size_t)n_boxes * handle->engine->feature_dim() * 4,
                              hipMemcpyDeviceToHost, handle->stream));
        BM_HIP(hipStreamSynchronize(handle->stream));
    });
}

int boxmot_hip_reid_last_time_ms(BoxMOTHipReID* handle, double* out_preprocess_ms, double* out_process_ms) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null ReID handle");
        double pre = 0, proc = 0;
        handle->engine->last_times(pre, proc);
        if (out_preprocess_ms) *out_preprocess_ms = pre;
        if (out_process_ms) *out_process_ms = proc;
    });
}

int boxmot_hip_reid_preprocess(BoxMOTHipReID* handle, const uint8_t* image, int image_rows, int image_cols,
                               int image_channels, const float* boxes, int n_boxes, int box_cols, float* out_crops) {
    return guard_on(handle, [&]() {
        ObbScope scope{handle};
        reid_stage(handle, image, image_rows, image_cols, image_channels, boxes, n_boxes, box_cols);
        if (n_boxes == 0) return;
        handle->engine->preprocess_fp32(handle->d_frames, handle->d_crop_stream, handle->d_boxes, 4, n_boxes, image_cols,
                                        image_rows, handle->stream);
        BM_HIP(hipMemcpyAsync(out_crops, handle->engine->crops_buffer(),
                              (size_t)n_boxes * bm::REID_IN_H * bm::REID_IN_W * 3 * 4, hipMemcpyDeviceToHost, handle->stream));
        BM_HIP(hipStreamSynchronize(handle->stream));
    });
}

// ---- ECC camera-motion estimation ----
BoxMOTHipEcc* boxmot_hip_ecc_create(int n_streams, int image_rows, int image_cols, double scale, double eps, int max_iter) {
    BoxMOTHipEcc* h = nullptr;
    const int ok = guard([&]() {
        require_device();
        h = new BoxMOTHipEcc();
        ecc_init(h, n_streams, image_rows, image_cols, scale, eps, max_iter, nullptr);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_ecc_destroy(BoxMOTHipEcc* handle) { destroy_on(handle); }

int boxmot_hip_ecc_reset(BoxMOTHipEcc* handle, int stream) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null ECC handle");
        if (stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        for (int s = 0; s < handle->S; ++s) if (stream < 0 || s == stream) handle->has_prev[s] = 0;
    });
}

int boxmot_hip_ecc_apply(BoxMOTHipEcc* handle, int stream, const uint8_t* image, int image_rows, int image_cols, int image_channels,
                         double* out_warp_2x3, int* out_iterations) {
    return guard_on(handle, [&]() {
        if (!handle || !out_warp_2x3) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (!image) throw std::runtime_error("Image data pointer is null.");
        if (image_channels != 3 || image_rows != handle->rows || image_cols != handle->cols)
            throw std::runtime_error("boxmot_hip: ECC was created for a different frame size (3-channel BGR uint8)");
        const size_t bytes = (size_t)image_rows * image_cols * 3;
        if (!handle->frame_bufs[stream]) {
            void* p = nullptr;
            BM_HIP(hipMalloc(&p, bytes));
            handle->frame_bufs[stream] = static_cast<uint8_t*>(p);
            BM_HIP(hipMemcpy(handle->d_frames, handle->frame_bufs.data(), handle->S * sizeof(uint8_t*), hipMemcpyHostToDevice));
        }
        BM_HIP(hipMemcpyAsync(handle->frame_bufs[stream], image, bytes, hipMemcpyHostToDevice, handle->stream));
        ecc_run_one(handle, stream, handle->d_frames + stream, out_warp_2x3, out_iterations);
    });
}

int boxmot_hip_ecc_apply_device(BoxMOTHipEcc* handle, int stream, const uint8_t* d_frame, double* out_warp_2x3, int* out_iterations) {
    return guard_on(handle, [&]() {
        if (!handle || !out_warp_2x3 || !d_frame) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        // the pointer table lives on the device: one slot per stream, written on the handle's stream before the kernels read it
        BM_HIP(hipMemcpyAsync(handle->d_frames + stream, &d_frame, sizeof(uint8_t*), hipMemcpyHostToDevice, handle->stream));
        BM_HIP(hipStreamSynchronize(handle->stream));       // &d_frame is a stack address
        ecc_run_one(handle, stream, handle->d_frames + stream, out_warp_2x3, out_iterations);
    });
}

// ---- SOF camera-motion estimation ----
BoxMOTHipSof* boxmot_hip_sof_create(int n_streams, int image_rows, int image_cols, double scale, int min_inliers, double min_inlier_ratio,
                                    double ransac_reproj_threshold) {
    BoxMOTHipSof* h = nullptr;
    const int ok = guard([&]() {
        require_device();
        h = new BoxMOTHipSof();
        sof_init(h, n_streams, image_rows, image_cols, scale, min_inliers, min_inlier_ratio, ransac_reproj_threshold, nullptr);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_sof_destroy(BoxMOTHipSof* handle) { destroy_on(handle); }

int boxmot_hip_sof_reset(BoxMOTHipSof* handle, int stream) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null SOF handle");
        if (stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        BM_HIP(hipStreamSynchronize(handle->stream));
        if (stream < 0) BM_HIP(hipMemset(handle->buf.st, 0, (size_t)handle->S * sizeof(bm::SofState)));
        else BM_HIP(hipMemset(handle->buf.st + stream, 0, sizeof(bm::SofState)));
    });
}

int boxmot_hip_sof_apply(BoxMOTHipSof* handle, int stream, const uint8_t* image, int image_rows, int image_cols, int image_channels,
                         const float* dets, int n_dets, int det_stride, double* out_warp_2x3, int* out_info8) {
    return guard_on(handle, [&]() {
        if (!handle || !out_warp_2x3) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (!image) throw std::runtime_error("Image data pointer is null.");
        if (image_channels != 3 || image_rows != handle->rows || image_cols != handle->cols)
            throw std::runtime_error("boxmot_hip: SOF was created for a different frame size (3-channel BGR uint8)");
        const size_t bytes = (size_t)image_rows * image_cols * 3;
        if (!handle->frame_bufs[stream]) {
            void* p = nullptr;
            BM_HIP(hipMalloc(&p, bytes));
            handle->frame_bufs[stream] = static_cast<uint8_t*>(p);
            BM_HIP(hipMemcpy(handle->d_frames, handle->frame_bufs.data(), handle->S * sizeof(uint8_t*), hipMemcpyHostToDevice));
        }
        BM_HIP(hipMemcpyAsync(handle->frame_bufs[stream], image, bytes, hipMemcpyHostToDevice, handle->stream));
        sof_stage_dets(handle, stream, dets, n_dets, det_stride);
        sof_run(handle, stream, 1, handle->d_frames + stream, handle->d_dets + (size_t)stream * handle->max_dets * 4, handle->d_ndets + stream,
                handle->max_dets, 4, out_warp_2x3, out_info8);
    });
}

int boxmot_hip_sof_apply_device(BoxMOTHipSof* handle, const uint8_t* const* d_frames, const float* d_dets, const int* d_ndets, int max_dets,
                                int det_stride, double* out_warps, int* out_info8) {
    return guard_on(handle, [&]() {
        if (!handle || !out_warps || !d_frames) throw std::runtime_error("boxmot_hip: null argument");
        if (d_dets && (!d_ndets || max_dets < 1 || det_stride < 4)) throw std::runtime_error("boxmot_hip: SOF device detections need d_ndets, max_dets >= 1 and >= 4 columns");
        sof_run(handle, 0, handle->S, d_frames, d_dets, d_dets ? d_ndets : nullptr, max_dets, det_stride, out_warps, out_info8);
    });
}

int boxmot_hip_sof_keypoints(BoxMOTHipSof* handle, int stream, float* out_xy, int capacity, int* out_n) {
    return guard_on(handle, [&]() {
        if (!handle || !out_n) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        bm::SofState st{};
        BM_HIP(hipStreamSynchronize(handle->stream));
        BM_HIP(hipMemcpy(&st, handle->buf.st + stream, sizeof(st), hipMemcpyDeviceToHost));
        *out_n = st.n_prev;
        const int n = st.n_prev < capacity ? st.n_prev : capacity;
        if (n > 0 && out_xy) BM_HIP(hipMemcpy(out_xy, handle->buf.prev_kps + (size_t)stream * bm::SOF_MAX_CORNERS * 2, (size_t)n * 8, hipMemcpyDeviceToHost));
    });
}

// test access to the detector's intermediate images of the last frame: which = 0 minimum-eigenvalue map (fp32 [h][w]), 1 detection mask
// (uint8 [h][w]), 2 the scaled grayscale frame (uint8 [h][w])
int boxmot_hip_sof_debug_map(BoxMOTHipSof* handle, int stream, int which, void* out, int capacity_bytes, int* out_h, int* out_w) {
    return guard_on(handle, [&]() {
        if (!handle || !out) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        const size_t P = (size_t)handle->lv.h[0] * handle->lv.w[0];
        const size_t bytes = which == 0 ? P * 4 : P;
        if ((size_t)capacity_bytes < bytes) throw std::runtime_error("boxmot_hip: output buffer is too small");
        BM_HIP(hipStreamSynchronize(handle->stream));
        const void* src = which == 0 ? (const void*)(handle->buf.eig + stream * P)
                        : which == 1 ? (const void*)(handle->buf.mask + stream * P)
                                     : (const void*)(handle->buf.pyr_prev + (size_t)stream * handle->lv.total);
        BM_HIP(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
        if (out_h) *out_h = handle->lv.h[0];
        if (out_w) *out_w = handle->lv.w[0];
    });
}

// ---- frame ingest ring ----
BoxMOTHipIngest* boxmot_hip_ingest_create(int n_slots, int n_streams, int image_rows, int image_cols) {
    BoxMOTHipIngest* h = nullptr;
    const int ok = guard([&]() {
        require_device();
        if (n_slots < 2 || n_streams < 1 || image_rows < 1 || image_cols < 1)
            throw std::runtime_error("boxmot_hip: ingest ring needs >= 2 slots, >= 1 stream and positive frame dimensions");
        h = new BoxMOTHipIngest();
        h->n_slots = n_slots; h->n_streams = n_streams; h->rows = image_rows; h->cols = image_cols;
        h->frame_bytes = (size_t)image_rows * image_cols * 3;
        BM_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        const size_t slot_bytes = h->frame_bytes * n_streams;
        for (int k = 0; k < n_slots; ++k) {
            void *hp = nullptr, *dp = nullptr, *tp = nullptr;
            BM_HIP(hipHostMalloc(&hp, slot_bytes, hipHostMallocDefault));
            h->h_slot.push_back(static_cast<uint8_t*>(hp));
            BM_HIP(hipMalloc(&dp, slot_bytes));
            h->d_slot.push_back(static_cast<uint8_t*>(dp));
            BM_HIP(hipMalloc(&tp, n_streams * sizeof(uint8_t*)));
            h->d_ptrs.push_back(static_cast<const uint8_t**>(tp));
            std::vector<const uint8_t*> table(n_streams);
            for (int s = 0; s < n_streams; ++s) table[s] = h->d_slot[k] + (size_t)s * h->frame_bytes;
            BM_HIP(hipMemcpy(tp, table.data(), n_streams * sizeof(uint8_t*), hipMemcpyHostToDevice));
            hipEvent_t a, b;
            BM_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            BM_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            h->uploaded.push_back(a); h->consumed.push_back(b);
        }
        h->has_consumer.assign(n_slots, 0);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_ingest_destroy(BoxMOTHipIngest* handle) { destroy_on(handle); }

static void ingest_slot(BoxMOTHipIngest* h, int slot) {
    if (!h) throw std::runtime_error("boxmot_hip: null ingest handle");
    if (slot < 0 || slot >= h->n_slots) throw std::runtime_error("boxmot_hip: ingest slot out of range");
}

uint8_t* boxmot_hip_ingest_host_ptr(BoxMOTHipIngest* handle, int slot, int stream) {
    uint8_t* p = nullptr;
    guard_on(handle, [&]() {
        ingest_slot(handle, slot);
        if (stream < 0 || stream >= handle->n_streams) throw std::runtime_error("boxmot_hip: stream index out of range");
        p = handle->h_slot[slot] + (size_t)stream * handle->frame_bytes;
    });
    return p;
}

const uint8_t* const* boxmot_hip_ingest_device_frames(BoxMOTHipIngest* handle, int slot) {
    const uint8_t* const* p = nullptr;
    guard_on(handle, [&]() { ingest_slot(handle, slot); p = handle->d_ptrs[slot]; });
    return p;
}

int boxmot_hip_ingest_submit(BoxMOTHipIngest* handle, int slot, int n_streams) {
    return guard_on(handle, [&]() {
        ingest_slot(handle, slot);
        if (n_streams < 1 || n_streams > handle->n_streams) throw std::runtime_error("boxmot_hip: stream count out of range");
        // the previous consumer of this slot must be done with the device frames before they are overwritten
        if (handle->has_consumer[slot]) BM_HIP(hipStreamWaitEvent(handle->copy_stream, handle->consumed[slot], 0));
        BM_HIP(hipMemcpyAsync(handle->d_slot[slot], handle->h_slot[slot], handle->frame_bytes * n_streams, hipMemcpyHostToDevice,
                              handle->copy_stream));
        BM_HIP(hipEventRecord(handle->uploaded[slot], handle->copy_stream));
    });
}

int boxmot_hip_ingest_wait(BoxMOTHipIngest* handle, int slot, void* consumer_stream) {
    return guard_on(handle, [&]() {
        ingest_slot(handle, slot);
        BM_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(consumer_stream), handle->uploaded[slot], 0));
    });
}

int boxmot_hip_ingest_release(BoxMOTHipIngest* handle, int slot, void* consumer_stream) {
    return guard_on(handle, [&]() {
        ingest_slot(handle, slot);
        BM_HIP(hipEventRecord(handle->consumed[slot], static_cast<hipStream_t>(consumer_stream)));
        handle->has_consumer[slot] = 1;
    });
}

int boxmot_hip_ingest_host_done(BoxMOTHipIngest* handle, int slot) {
    return guard_on(handle, [&]() {
        ingest_slot(handle, slot);
        BM_HIP(hipEventSynchronize(handle->uploaded[slot]));       // the host buffer of the slot may be refilled
    });
}

// ---- DeepOCSORT ----
void boxmot_hip_deepocsort_default_config(BoxMOTHipDeepOcSortConfig* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->det_thresh = 0.3; c->max_age = 30; c->max_obs = 50; c->min_hits = 3; c->iou_threshold = 0.3;
    c->delta_t = 3; c->inertia = 0.2; c->w_association_emb = 0.5; c->alpha_fixed_emb = 0.95; c->aw_param = 0.5;
    c->embedding_off = 0; c->cmc_off = 0; c->aw_off = 0; c->Q_xy_scaling = 0.01; c->Q_s_scaling = 0.0001;
    c->reid_model_path = nullptr;
    c->n_streams = 1; c->max_tracks = 1024; c->max_dets = 256; c->emb_dim = 512;
    c->use_byte = 0; c->min_conf = 0.1;
    c->asso_func = BOXMOT_HIP_ASSO_IOU; c->frame_w = 0; c->frame_h = 0;
}

BoxMOTHipDeepOcSort* boxmot_hip_deepocsort_create(const BoxMOTHipDeepOcSortConfig* config) {
    BoxMOTHipDeepOcSort* h = nullptr;
    const int ok = guard([&]() {
        if (config == nullptr) throw std::runtime_error("boxmot_hip DeepOCSORT config is required.");
        require_device();
        h = new BoxMOTHipDeepOcSort();
        h->cfg = *config;
        if (config->reid_model_path) h->reid_path = config->reid_model_path;
        h->cfg.reid_model_path = nullptr;
        docs_build(h);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_deepocsort_destroy(BoxMOTHipDeepOcSort* handle) { destroy_on(handle); }

int boxmot_hip_deepocsort_reset(BoxMOTHipDeepOcSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is null.");
        docs_zero_state(handle);
        handle->h_used.assign(handle->S, 0);
    });
}

int boxmot_hip_deepocsort_reserve(BoxMOTHipDeepOcSort* handle, int max_tracks, int max_dets) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is null.");
        const int cap = max_tracks > handle->cap ? (max_tracks + 63) / 64 * 64 : handle->cap;
        const int nd = max_dets > handle->nd ? (max_dets + 63) / 64 * 64 : handle->nd;
        if (cap != handle->cap || nd != handle->nd) { docs_grow(handle, cap, nd); ++handle->n_grows; }
    });
}

int boxmot_hip_deepocsort_capacity(BoxMOTHipDeepOcSort* handle, int* max_tracks, int* max_dets, int* n_grows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is null.");
        if (max_tracks) *max_tracks = handle->cap;
        if (max_dets) *max_dets = handle->nd;
        if (n_grows) *n_grows = handle->n_grows;
    });
}

int boxmot_hip_deepocsort_set_warp(BoxMOTHipDeepOcSort* handle, int stream, const double* warp_2x3) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is null.");
        if (handle->is_obb && warp_2x3) throw std::runtime_error("boxmot_hip: camera-motion warps are not applied to oriented detections");
        io_set_warp(handle, stream, warp_2x3);
    });
}

int boxmot_hip_deepocsort_update_batch(BoxMOTHipDeepOcSort* handle, int n_streams, const float* const* dets,
                                       const int* det_rows, const float* const* embs, int emb_cols,
                                       const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
                                       float* const* out_tracks, int out_capacity_rows, int* out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        if (n_streams < 1 || n_streams > handle->S) throw std::runtime_error("boxmot_hip: n_streams out of range");
        if (!dets || !det_rows || !out_tracks || !out_rows) throw std::runtime_error("boxmot_hip: null batch pointers");
        std::vector<StreamIn> in(n_streams);
        for (int s = 0; s < n_streams; ++s)
            in[s] = StreamIn{dets[s], det_rows[s], (embs && emb_cols > 0) ? embs[s] : nullptr, images ? images[s] : nullptr};
        docs_need_frame_size(handle, image_rows, image_cols);
        docs_host_update(handle, n_streams, in.data(), handle->det_cols, emb_cols, image_rows, image_cols, image_channels, out_tracks,
                         out_capacity_rows, out_rows);
    });
}

int boxmot_hip_deepocsort_update(BoxMOTHipDeepOcSort* handle, const float* dets, int det_rows, int det_cols,
                                 const float* embs, int emb_rows, int emb_cols, const uint8_t* image, int image_rows,
                                 int image_cols, int image_channels, float* out_tracks, int out_capacity_rows,
                                 int out_cols, int* out_rows, int* out_is_obb) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        if (!out_rows || !out_is_obb) throw std::runtime_error("Output pointers are null.");
        if (out_cols != 9) throw std::runtime_error("boxmot_hip live tracking expects an output buffer with 9 columns.");
        if (embs != nullptr && emb_rows != det_rows) throw std::runtime_error("Detection and embedding row counts must match.");
        if (image_rows <= 0 || image_cols <= 0) throw std::runtime_error("Image dimensions must be positive.");
        StreamIn in{dets, det_rows, (embs && emb_cols > 0) ? embs : nullptr, image};
        float* outs[1] = {out_tracks};
        docs_need_frame_size(handle, image_rows, image_cols);
        docs_host_update(handle, 1, &in, det_cols, emb_cols, image_rows, image_cols, image_channels, outs, out_capacity_rows, out_rows);
        *out_is_obb = handle->is_obb ? 1 : 0;
    });
}

int boxmot_hip_deepocsort_update_stream(BoxMOTHipDeepOcSort* handle, int stream, int frame_count, int* id_count_inout,
                                        const float* dets, int det_rows, int det_cols, const float* embs, int emb_rows,
                                        int emb_cols, const uint8_t* image, int image_rows, int image_cols, int image_channels,
                                        float* out_tracks, int out_capacity_rows, int out_cols, int* out_rows, int* out_is_obb) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        if (!out_rows || !out_is_obb) throw std::runtime_error("Output pointers are null.");
        if (out_cols != 9) throw std::runtime_error("boxmot_hip live tracking expects an output buffer with 9 columns.");
        if (embs != nullptr && emb_rows != det_rows) throw std::runtime_error("Detection and embedding row counts must match.");
        if (image_rows <= 0 || image_cols <= 0) throw std::runtime_error("Image dimensions must be positive.");
        StreamIn in{dets, det_rows, (embs && emb_cols > 0) ? embs : nullptr, image};
        float* outs[1] = {out_tracks};
        docs_need_frame_size(handle, image_rows, image_cols);
        docs_host_update(handle, 1, &in, det_cols, emb_cols, image_rows, image_cols, image_channels, outs, out_capacity_rows, out_rows,
                         stream, frame_count, id_count_inout);
        *out_is_obb = handle->is_obb ? 1 : 0;
    });
}

int boxmot_hip_deepocsort_step_device(BoxMOTHipDeepOcSort* handle, const float* d_dets, const int* d_det_rows,
                                       const float* d_embs, float* d_out, int* d_out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        if (!d_dets || !d_det_rows || !d_out || !d_out_rows) throw std::runtime_error("boxmot_hip: null device pointers");
        if (!handle->cfg.embedding_off && !d_embs) throw std::runtime_error("boxmot_hip: step_device needs d_embs unless embedding_off");
        docs_need_frame_size(handle, 0, 0);
        bm::DocsStepArgs a = handle->args;
        a.dets = d_dets; a.n_dets = d_det_rows; a.embs = handle->cfg.embedding_off ? nullptr : d_embs;
        const bool any_warp = io_consume_warps(handle);         // boxmot_hip_deepocsort_set_warp since the last step
        a.warp = any_warp ? handle->d_warp : nullptr; a.warp_flag = any_warp ? handle->d_warp_flag : nullptr;
        a.out = d_out; a.out_n = d_out_rows; a.stream_base = 0;
        docs_launch(handle, handle->S, a);
        BM_HIP(hipGetLastError());
        io_clear_warps(handle);
    });
}

int boxmot_hip_deepocsort_step_device_frames(BoxMOTHipDeepOcSort* handle, const float* d_dets, const int* d_det_rows,
                                              const uint8_t* const* d_frames, int image_rows, int image_cols, float* d_out,
                                              int* d_out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        if (!d_dets || !d_det_rows || !d_out || !d_out_rows) throw std::runtime_error("boxmot_hip: null device pointers");
        const float* embs = nullptr;
        if (!handle->cfg.embedding_off)      // deepocsort.py:337-345: every detection above det_thresh is embedded
            embs = io_device_reid(handle, d_dets, d_det_rows, d_frames, image_rows, image_cols, (double)(float)handle->cfg.det_thresh, 0);
        docs_need_frame_size(handle, image_rows, image_cols);
        bm::DocsStepArgs a = handle->args;
        a.dets = d_dets; a.n_dets = embs ? handle->step_ndets : d_det_rows; a.embs = embs;
        const bool any_warp = io_consume_warps(handle);
        a.warp = any_warp ? handle->d_warp : nullptr; a.warp_flag = any_warp ? handle->d_warp_flag : nullptr;
        a.out = d_out; a.out_n = d_out_rows; a.stream_base = 0;
        docs_launch(handle, handle->S, a);
        BM_HIP(hipGetLastError());
        if (embs) io_pipe_step_enqueued(handle);
        io_clear_warps(handle);
    });
}

int boxmot_hip_deepocsort_set_reid_mode(BoxMOTHipDeepOcSort* handle, int mode) {
    return guard_on(handle, [&]() {
        if (!handle || !handle->reid) throw std::runtime_error("boxmot_hip: no ReID weights are loaded in this handle");
        handle->reid->set_mode(mode);
        handle->reid_mode = mode;
    });
}

int boxmot_hip_deepocsort_reid_kernel_ms(BoxMOTHipDeepOcSort* handle, double* out_ms, int* out_launches) {
    return guard_on(handle, [&]() {
        if (!handle || !out_ms || !out_launches) throw std::runtime_error("boxmot_hip: null argument");
        *out_ms = 0; *out_launches = 0;
        if (handle->reid) handle->reid->drain_kernel_timing(*out_ms, *out_launches);
    });
}

void* boxmot_hip_deepocsort_stream(BoxMOTHipDeepOcSort* handle) { if (handle) handle->stream_exposed = true; return handle ? (void*)handle->stream : nullptr; }

int boxmot_hip_deepocsort_synchronize(BoxMOTHipDeepOcSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        BM_HIP(hipStreamSynchronize(handle->stream));
        io_check_crop_bound(handle);
    });
}

int boxmot_hip_deepocsort_set_crop_bound(BoxMOTHipDeepOcSort* handle, int max_total_crops) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip DeepOCSORT handle is not initialized.");
        io_set_crop_bound(handle, max_total_crops);
    });
}

int boxmot_hip_deepocsort_state_dump(BoxMOTHipDeepOcSort* handle, int stream, int* ints5, double* kf72, double* emb,
                                     int* out_rows, int* out_frame_count, int* out_id_count) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        BM_HIP(hipStreamSynchronize(handle->stream));
        const bm::DocsState& st = handle->args.st;
        const size_t cap = handle->cap, dim = handle->dim, off = (size_t)stream * cap;
        int n = 0, fc = 0, ic = 0;
        BM_HIP(hipMemcpy(&n, st.n_tracks + stream, 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(&fc, st.frame_count + stream, 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(&ic, st.id_count + stream, 4, hipMemcpyDeviceToHost));
        std::vector<int> list(cap), id(cap), age(cap), tsu(cap), hs(cap), obs(cap);
        BM_HIP(hipMemcpy(list.data(), st.list + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(id.data(), st.id + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(age.data(), st.age + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(tsu.data(), st.tsu + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(hs.data(), st.hit_streak + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(obs.data(), st.observed + off, cap * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < n; ++r) {
            const int sl = list[r];
            if (ints5) { int* o = ints5 + r * 5; o[0] = id[sl]; o[1] = age[sl]; o[2] = tsu[sl]; o[3] = hs[sl]; o[4] = obs[sl]; }
            const size_t KS = bm::docs_kf_stride(handle->is_obb);      // 72 doubles per track, 90 (x[9] ++ P[9][9]) on an oriented handle
            if (kf72) BM_HIP(hipMemcpy(kf72 + (size_t)r * KS, st.kf + (off + sl) * KS, KS * 8, hipMemcpyDeviceToHost));
            if (emb) BM_HIP(hipMemcpy(emb + (size_t)r * dim, st.emb + (off + sl) * dim, dim * 8, hipMemcpyDeviceToHost));
        }
        *out_rows = n;
        if (out_frame_count) *out_frame_count = fc;
        if (out_id_count) *out_id_count = ic;
    });
}

// ---- StrongSORT ----
void boxmot_hip_strongsort_default_config(BoxMOTHipStrongSortConfig* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->max_age = 30; c->min_conf = 0.1; c->max_cos_dist = 0.2; c->max_iou_dist = 0.7; c->n_init = 3; c->nn_budget = 100;
    c->mc_lambda = 0.98; c->ema_alpha = 0.9; c->reid_model_path = nullptr;
    c->n_streams = 1; c->max_tracks = 1024; c->max_dets = 256; c->emb_dim = 512;
}

BoxMOTHipStrongSort* boxmot_hip_strongsort_create(const BoxMOTHipStrongSortConfig* config) {
    BoxMOTHipStrongSort* h = nullptr;
    const int ok = guard([&]() {
        if (config == nullptr) throw std::runtime_error("boxmot_hip StrongSORT config is required.");
        require_device();
        h = new BoxMOTHipStrongSort();
        h->cfg = *config;
        if (config->reid_model_path) h->reid_path = config->reid_model_path;
        h->cfg.reid_model_path = nullptr;
        ss_build(h);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}

void boxmot_hip_strongsort_destroy(BoxMOTHipStrongSort* handle) { destroy_on(handle); }

int boxmot_hip_strongsort_reset(BoxMOTHipStrongSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is null.");
        ss_zero_state(handle);
        handle->h_used.assign(handle->S, 0);
    });
}

int boxmot_hip_strongsort_reserve(BoxMOTHipStrongSort* handle, int max_tracks, int max_dets) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is null.");
        const int cap = max_tracks > handle->cap ? (max_tracks + 63) / 64 * 64 : handle->cap;
        const int nd = max_dets > handle->nd ? (max_dets + 63) / 64 * 64 : handle->nd;
        if (cap != handle->cap || nd != handle->nd) { ss_grow(handle, cap, nd); ++handle->n_grows; }
    });
}

int boxmot_hip_strongsort_capacity(BoxMOTHipStrongSort* handle, int* max_tracks, int* max_dets, int* n_grows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is null.");
        if (max_tracks) *max_tracks = handle->cap;
        if (max_dets) *max_dets = handle->nd;
        if (n_grows) *n_grows = handle->n_grows;
    });
}

int boxmot_hip_strongsort_set_warp(BoxMOTHipStrongSort* handle, int stream, const double* warp_2x3) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is null.");
        io_set_warp(handle, stream, warp_2x3);
    });
}

int boxmot_hip_strongsort_update_batch(BoxMOTHipStrongSort* handle, int n_streams, const float* const* dets,
                                       const int* det_rows, const float* const* embs, int emb_cols,
                                       const uint8_t* const* images, int image_rows, int image_cols, int image_channels,
                                       float* const* out_tracks, int out_capacity_rows, int* out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        if (n_streams < 1 || n_streams > handle->S) throw std::runtime_error("boxmot_hip: n_streams out of range");
        if (!dets || !det_rows || !out_tracks || !out_rows) throw std::runtime_error("boxmot_hip: null batch pointers");
        std::vector<StreamIn> in(n_streams);
        for (int s = 0; s < n_streams; ++s)
            in[s] = StreamIn{dets[s], det_rows[s], (embs && emb_cols > 0) ? embs[s] : nullptr, images ? images[s] : nullptr};
        ss_host_update(handle, n_streams, in.data(), 6, emb_cols, image_rows, image_cols, image_channels, out_tracks,
                       out_capacity_rows, out_rows);
    });
}

int boxmot_hip_strongsort_update(BoxMOTHipStrongSort* handle, const float* dets, int det_rows, int det_cols,
                                 const float* embs, int emb_rows, int emb_cols, const uint8_t* image, int image_rows,
                                 int image_cols, int image_channels, float* out_tracks, int out_capacity_rows,
                                 int out_cols, int* out_rows, int* out_is_obb) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        if (!out_rows || !out_is_obb) throw std::runtime_error("Output pointers are null.");
        if (out_cols != 9) throw std::runtime_error("boxmot_hip live tracking expects an output buffer with 9 columns.");
        if (embs != nullptr && emb_rows != det_rows) throw std::runtime_error("Detection and embedding row counts must match.");
        if (image_rows <= 0 || image_cols <= 0) throw std::runtime_error("Image dimensions must be positive.");
        StreamIn in{dets, det_rows, (embs && emb_cols > 0) ? embs : nullptr, image};
        float* outs[1] = {out_tracks};
        ss_host_update(handle, 1, &in, det_cols, emb_cols, image_rows, image_cols, image_channels, outs, out_capacity_rows, out_rows);
        *out_is_obb = 0;
    });
}

int boxmot_hip_strongsort_step_device(BoxMOTHipStrongSort* handle, const float* d_dets, const int* d_det_rows,
                                       const float* d_embs, float* d_out, int* d_out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        if (!d_dets || !d_det_rows || !d_embs || !d_out || !d_out_rows) throw std::runtime_error("boxmot_hip: null device pointers");
        bm::SsStepArgs a = handle->args;
        a.dets = d_dets; a.n_dets = d_det_rows; a.embs = d_embs;
        a.warp = io_consume_warps(handle) ? handle->d_warp : nullptr;       // boxmot_hip_strongsort_set_warp since the last step
        a.out = d_out; a.out_n = d_out_rows; a.stream_base = 0;
        ss_launch(handle, a, handle->S);
        BM_HIP(hipGetLastError());
        io_clear_warps(handle);
    });
}

int boxmot_hip_strongsort_step_device_frames(BoxMOTHipStrongSort* handle, const float* d_dets, const int* d_det_rows,
                                              const uint8_t* const* d_frames, int image_rows, int image_cols, float* d_out,
                                              int* d_out_rows) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        if (!d_dets || !d_det_rows || !d_out || !d_out_rows) throw std::runtime_error("boxmot_hip: null device pointers");
        const float* embs = io_device_reid(handle, d_dets, d_det_rows, d_frames, image_rows, image_cols, handle->cfg.min_conf, 1);    // strongsort.py:74-91
        bm::SsStepArgs a = handle->args;
        a.dets = d_dets; a.n_dets = handle->step_ndets; a.embs = embs;
        a.warp = io_consume_warps(handle) ? handle->d_warp : nullptr;
        a.out = d_out; a.out_n = d_out_rows; a.stream_base = 0;
        ss_launch(handle, a, handle->S);
        BM_HIP(hipGetLastError());
        io_pipe_step_enqueued(handle);
        io_clear_warps(handle);
    });
}

int boxmot_hip_strongsort_set_reid_mode(BoxMOTHipStrongSort* handle, int mode) {
    return guard_on(handle, [&]() {
        if (!handle || !handle->reid) throw std::runtime_error("boxmot_hip: no ReID weights are loaded in this handle");
        handle->reid->set_mode(mode);
        handle->reid_mode = mode;
    });
}

int boxmot_hip_strongsort_reid_kernel_ms(BoxMOTHipStrongSort* handle, double* out_ms, int* out_launches) {
    return guard_on(handle, [&]() {
        if (!handle || !out_ms || !out_launches) throw std::runtime_error("boxmot_hip: null argument");
        *out_ms = 0; *out_launches = 0;
        if (handle->reid) handle->reid->drain_kernel_timing(*out_ms, *out_launches);
    });
}

int boxmot_hip_strongsort_track_count(BoxMOTHipStrongSort* handle, int stream, int* out_tracks) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is null.");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (!out_tracks) throw std::runtime_error("Output pointers are null.");
        BM_HIP(hipStreamSynchronize(handle->stream));
        BM_HIP(hipMemcpy(out_tracks, handle->args.st.n_tracks + stream, 4, hipMemcpyDeviceToHost));
    });
}

void* boxmot_hip_strongsort_stream(BoxMOTHipStrongSort* handle) { if (handle) handle->stream_exposed = true; return handle ? (void*)handle->stream : nullptr; }

int boxmot_hip_strongsort_synchronize(BoxMOTHipStrongSort* handle) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        BM_HIP(hipStreamSynchronize(handle->stream));
        io_check_crop_bound(handle);
    });
}

int boxmot_hip_strongsort_set_crop_bound(BoxMOTHipStrongSort* handle, int max_total_crops) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip StrongSORT handle is not initialized.");
        io_set_crop_bound(handle, max_total_crops);
    });
}

int boxmot_hip_deepocsort_debug_costs_enable(BoxMOTHipDeepOcSort* handle, int on) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null argument");
        BM_HIP(hipStreamSynchronize(handle->stream));
        if (on) docs_make_dbg(handle);
        else if (handle->args.dbg_cost) {
            release(handle->owned, handle->args.dbg_cost); release(handle->owned, handle->args.dbg_shape);
            handle->args.dbg_cost = nullptr; handle->args.dbg_shape = nullptr;
        }
    });
}

int boxmot_hip_deepocsort_debug_costs(BoxMOTHipDeepOcSort* handle, int stream, int plane, double* out, long out_capacity,
                                      int* out_rows, int* out_cols, int* out_branch) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows || !out_cols) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (plane < 0 || plane >= bm::DOCS_DBG_PLANES) throw std::runtime_error("boxmot_hip: debug_costs plane must be 0..2");
        if (!handle->args.dbg_cost)
            throw std::runtime_error("boxmot_hip: debug_costs needs boxmot_hip_deepocsort_debug_costs_enable(handle, 1) before the update");
        BM_HIP(hipStreamSynchronize(handle->stream));
        int shape[4] = {0, 0, 0, 0};
        BM_HIP(hipMemcpy(shape, handle->args.dbg_shape + (size_t)stream * 4, 16, hipMemcpyDeviceToHost));
        const int R = shape[0], C = shape[1];             // (detections, tracks): association.py's iou_matrix / final_cost orientation
        *out_rows = R; *out_cols = C;
        if (out_branch) *out_branch = shape[2];
        if (R == 0 || C == 0) return;
        if (!out || out_capacity < (long)R * C) throw std::runtime_error("boxmot_hip: debug_costs output capacity is smaller than rows x cols");
        const size_t cap = handle->dbg_cap, nd = handle->dbg_nd;
        std::vector<double> m(nd * cap);
        BM_HIP(hipMemcpy(m.data(), handle->args.dbg_cost + ((size_t)stream * bm::DOCS_DBG_PLANES + plane) * nd * cap, m.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r) std::memcpy(out + (size_t)r * C, m.data() + (size_t)r * cap, (size_t)C * 8);
    });
}

int boxmot_hip_strongsort_debug_costs_enable(BoxMOTHipStrongSort* handle, int on) {
    return guard_on(handle, [&]() {
        if (!handle) throw std::runtime_error("boxmot_hip: null argument");
        BM_HIP(hipStreamSynchronize(handle->stream));
        if (on) ss_make_dbg(handle);
        else if (handle->args.dbg_cost) {
            release(handle->owned, handle->args.dbg_cost); release(handle->owned, handle->args.dbg_shape);
            handle->args.dbg_cost = nullptr; handle->args.dbg_shape = nullptr;
        }
    });
}

int boxmot_hip_strongsort_debug_costs(BoxMOTHipStrongSort* handle, int stream, int stage, int plane, double* out, long out_capacity,
                                      int* out_rows, int* out_cols) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows || !out_cols) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        if (stage < 0 || stage > 1 || plane < 0 || plane > 1) throw std::runtime_error("boxmot_hip: debug_costs stage and plane must be 0 or 1");
        if (!handle->args.dbg_cost)
            throw std::runtime_error("boxmot_hip: debug_costs needs boxmot_hip_strongsort_debug_costs_enable(handle, 1) before the update");
        BM_HIP(hipStreamSynchronize(handle->stream));
        int shape[4] = {0, 0, 0, 0};
        BM_HIP(hipMemcpy(shape, handle->args.dbg_shape + (size_t)stream * 4, 16, hipMemcpyDeviceToHost));
        const int R = shape[stage * 2], C = shape[stage * 2 + 1];        // (tracks, detections)
        *out_rows = R; *out_cols = C;
        if (R == 0 || C == 0) return;
        if (!out || out_capacity < (long)R * C) throw std::runtime_error("boxmot_hip: debug_costs output capacity is smaller than rows x cols");
        const size_t big = handle->dbg_big;
        std::vector<double> m(big * big);
        BM_HIP(hipMemcpy(m.data(), handle->args.dbg_cost + ((size_t)stream * 4 + stage * 2 + plane) * big * big, m.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 0; r < R; ++r) std::memcpy(out + (size_t)r * C, m.data() + (size_t)r * big, (size_t)C * 8);
    });
}

int boxmot_hip_strongsort_state_dump(BoxMOTHipStrongSort* handle, int stream, int* ints6, double* kf72, float* feat,
                                     int* out_rows, int* out_frame_count, int* out_next_id) {
    return guard_on(handle, [&]() {
        if (!handle || !out_rows) throw std::runtime_error("boxmot_hip: null argument");
        if (stream < 0 || stream >= handle->S) throw std::runtime_error("boxmot_hip: stream index out of range");
        BM_HIP(hipStreamSynchronize(handle->stream));
        const bm::SsState& st = handle->args.st;
        const size_t cap = handle->cap, dim = handle->dim, off = (size_t)stream * cap;
        int n = 0, fc = 0, ni = 0;
        BM_HIP(hipMemcpy(&n, st.n_tracks + stream, 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(&fc, st.frame_count + stream, 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(&ni, st.next_id + stream, 4, hipMemcpyDeviceToHost));
        std::vector<int> list(cap), id(cap), state(cap), hits(cap), age(cap), tsu(cap), bank(cap);
        BM_HIP(hipMemcpy(list.data(), st.list + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(id.data(), st.id + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(state.data(), st.state + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(hits.data(), st.hits + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(age.data(), st.age + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(tsu.data(), st.tsu + off, cap * 4, hipMemcpyDeviceToHost));
        BM_HIP(hipMemcpy(bank.data(), st.bank_n + off, cap * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < n; ++r) {
            const int sl = list[r];
            if (ints6) {
                int* o = ints6 + r * 6;
                o[0] = id[sl]; o[1] = state[sl]; o[2] = hits[sl]; o[3] = age[sl]; o[4] = tsu[sl];
                o[5] = bank[sl] < st.budget ? bank[sl] : st.budget;
            }
            if (kf72) BM_HIP(hipMemcpy(kf72 + (size_t)r * bm::KF_STRIDE, st.kf + (off + sl) * bm::KF_STRIDE, bm::KF_STRIDE * 8, hipMemcpyDeviceToHost));
            if (feat) BM_HIP(hipMemcpy(feat + (size_t)r * dim, st.feat + (off + sl) * dim, dim * 4, hipMemcpyDeviceToHost));
        }
        *out_rows = n;
        if (out_frame_count) *out_frame_count = fc;
        if (out_next_id) *out_next_id = ni;
    });
}


// ---------------------------------------------------------------------------------------------------------------------
// Reference-named exports (include/boxmot_compat.h): the exact symbol names, struct layouts and signatures of the
// reference's native FFI, so that its own ctypes bindings load this library unchanged:
//   boxmot_botsort_*    boxmot/native/cpp/trackers/botsort/include/botsort/c_api.hpp:17-61   (native/trackers/botsort.py:94-146)
//   boxmot_bytetrack_*  boxmot/native/cpp/trackers/bytetrack/include/bytetrack/c_api.hpp:16-46
//   boxmot_ocsort_*     boxmot/native/cpp/trackers/ocsort/include/ocsort/c_api.hpp:16-52
//   boxmot_reid_capi_*  boxmot/native/cpp/trackers/base/include/boxmot/trackers/base/reid_capi.h:36-94
// They are thin adapters over the boxmot_hip_* entry points above (float thresholds widened to double; the knobs the C++
// reference hard-codes -- second/unconfirmed thresholds 0.5 / 0.7, embedding scale 2.0, tracker.cpp:435,465,476 -- at
// those values; capacities from BOXMOT_HIP_MAX_TRACKS / BOXMOT_HIP_MAX_DETS, default 1024 / 512).
// ---------------------------------------------------------------------------------------------------------------------
// The embedding width is a create-time capacity of the device state; the reference learns it from the first embeddings it
// sees.  The adapter therefore builds the inner handle at create when the width is known (ReID weights given: their feature
// width; with_reid = 0: none needed) and otherwise at the first update that brings embeddings.
// The reference's native trackers take a table of 6 (AABB) or 7 (OBB) columns per call (live_c_api.hpp:22-60: `detection.is_obb =
// det_cols == 7`, out_is_obb = det_cols == 7, :147-149) and its Python wrappers fix the layout with the first table ("cannot switch
// between AABB and OBB inputs", native/trackers/botsort.py).  A device handle is sized for one layout, so the adapters keep the
// configuration, decide the layout with the first non-empty table and re-make the (still unused) inner handle when it is oriented.
struct CompatLayout : DeviceBound {          // the device current at create: a lazily built inner handle lands there too
    int layout = -1;                // -1 undecided, 0 axis-aligned, 1 oriented
    bool stepped = false;           // an update has run on the inner handle: its layout is final
    int empty_frames = 0;           // updates that only counted: before the inner handle exists (BoT-SORT without an embedding width yet,
                                    // botsort.py:183) or 0 x 0 tables before the layout is known; replayed into the frame counter
};
struct BoxMOTBotSortHandle : CompatLayout {
    BoxMOTHipBotSortConfig cfg{};
    std::string reid_path, reid_pre, cmc;
    BoxMOTHipBotSort* inner = nullptr;
    ~BoxMOTBotSortHandle() { delete inner; }
};
struct BoxMOTByteTrackHandle : CompatLayout { BoxMOTHipBotSortConfig cfg{}; BoxMOTHipBotSort* inner = nullptr; ~BoxMOTByteTrackHandle() { delete inner; } };
struct BoxMOTOCSORTHandle : CompatLayout { BoxMOTHipDeepOcSortConfig cfg{}; BoxMOTHipDeepOcSort* inner = nullptr; ~BoxMOTOCSORTHandle() { delete inner; } };

namespace {
int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    const int x = std::atoi(v);
    return x > 0 ? x : dflt;
}
// Layout of this call's table against the handle's: returns true when the inner handle has to be (re-)made with `is_obb`.
// ValidateLiveDetectionShape (live_c_api.hpp:22-33): an empty 0 x 0 table is accepted and keeps the layout.
bool compat_layout(CompatLayout* h, int det_rows, int det_cols, int& is_obb, const char* tracker) {
    is_obb = h->layout == 1 ? 1 : 0;
    if (det_rows < 0 || det_cols < 0) throw std::runtime_error("Negative matrix dimensions are not allowed.");
    if (det_cols == 0 && det_rows == 0) return false;
    if (det_cols != 6 && det_cols != 7)
        throw std::runtime_error(std::string(tracker) + " live tracking expects detections with 6 (AABB) or 7 (OBB) columns.");
    const int want = det_cols == 7 ? 1 : 0;
    if (h->layout == want) return false;
    if (h->layout >= 0 && h->stepped)
        throw std::runtime_error(std::string(tracker) + ": cannot switch between AABB and OBB inputs");
    h->layout = want;
    is_obb = want;
    return true;
}

void compat_build_inner(BoxMOTBotSortHandle* h, int emb_dim) {
    h->cfg.emb_dim = emb_dim;
    h->cfg.reid_model_path = h->reid_path.empty() ? nullptr : h->reid_path.c_str();
    h->cfg.reid_preprocess = h->reid_pre.empty() ? nullptr : h->reid_pre.c_str();
    h->cfg.cmc_method = h->cmc.empty() ? nullptr : h->cmc.c_str();
    BoxMOTHipBotSort* inner = boxmot_hip_botsort_create(&h->cfg);
    if (!inner) throw std::runtime_error(g_last_error);
    delete h->inner;
    h->inner = inner;
}
// feature width declared in the header of an OSN1 weight blob (reid_layout.hpp)
int blob_feature_dim(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open ReID weight blob: " + path);
    int32_t hdr[bm::REID_HEADER_INTS] = {0};
    const size_t got = std::fread(hdr, 4, bm::REID_HEADER_INTS, f);
    std::fclose(f);
    if (got == (size_t)bm::REID_HEADER_INTS && hdr[0] == bm::CLIP_MAGIC) return hdr[1] + hdr[7];       // CLIP-ReID: width + projection
    if (got != (size_t)bm::REID_HEADER_INTS || hdr[0] != bm::REID_MAGIC)
        throw std::runtime_error("ReID weights must be an OSN1 (OSNet) or CLP1 (CLIP-ReID) blob (boxmot_amd.reid_weights / clip_weights): " + path);
    return hdr[5];
}
}  // namespace

BoxMOTBotSortHandle* boxmot_botsort_create(const BoxMOTBotSortConfig* c) {
    BoxMOTBotSortHandle* h = nullptr;
    const int ok = guard([&]() {
        if (c == nullptr) throw std::runtime_error("BoTSORT config is required.");
        require_device();
        h = new BoxMOTBotSortHandle();
        BoxMOTHipBotSortConfig& k = h->cfg;
        boxmot_hip_botsort_default_config(&k);          // second 0.5, unconfirmed 0.7, scale 2.0, removed buffer 100
        k.track_high_thresh = (double)c->track_high_thresh; k.track_low_thresh = (double)c->track_low_thresh;
        k.new_track_thresh = (double)c->new_track_thresh; k.track_buffer = c->track_buffer;
        k.match_thresh = (double)c->match_thresh; k.proximity_thresh = (double)c->proximity_thresh;
        k.appearance_thresh = (double)c->appearance_thresh;
        k.frame_rate = c->frame_rate; k.fuse_first_associate = c->fuse_first_associate; k.with_reid = c->with_reid ? 1 : 0;
        k.max_obs = c->max_obs;
        if (c->cmc_method) h->cmc = c->cmc_method;
        if (!h->cmc.empty() && h->cmc != "none" && h->cmc != "ecc" && h->cmc != "sof")     // checked here: the inner handle may only be built at the first update
            throw std::runtime_error("boxmot_hip: camera-motion estimator '" + h->cmc + "' is not implemented (have: sof, ecc, none); "
                                     "supply the warp per frame with boxmot_hip_botsort_set_warp");
        if (c->reid_model_path) h->reid_path = c->reid_model_path;
        if (c->reid_preprocess) h->reid_pre = c->reid_preprocess;
        k.n_streams = 1; k.n_class_lists = 1; k.tracker_kind = 0;
        k.max_tracks = env_int("BOXMOT_HIP_MAX_TRACKS", 1024);
        k.max_dets = env_int("BOXMOT_HIP_MAX_DETS", 512);
        if (!k.with_reid) compat_build_inner(h, 1);
        else if (!h->reid_path.empty()) compat_build_inner(h, blob_feature_dim(h->reid_path));
        // else: embeddings will be supplied by the caller; their width is known at the first update
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}
void boxmot_botsort_destroy(BoxMOTBotSortHandle* h) { destroy_on(h); }
int boxmot_botsort_reset(BoxMOTBotSortHandle* h) {
    if (h) { h->empty_frames = 0; h->stepped = false; h->layout = -1; }          // the next table decides the layout again
    if (h && !h->inner) return guard([]() {});
    return boxmot_hip_botsort_reset(h ? h->inner : nullptr);
}
int boxmot_botsort_update(BoxMOTBotSortHandle* h, const float* dets, int det_rows, int det_cols, const float* embs,
                          int emb_rows, int emb_cols, const uint8_t* image, int image_rows, int image_cols,
                          int image_channels, float* out_tracks, int out_capacity_rows, int out_cols, int* out_rows,
                          int* out_is_obb) {
    if (!h) return boxmot_hip_botsort_update(nullptr, dets, det_rows, det_cols, embs, emb_rows, emb_cols, image, image_rows, image_cols,
                                             image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    const int ok = guard_on(h, [&]() {
        int is_obb = 0;
        if (compat_layout(h, det_rows, det_cols, is_obb, "BoTSORT") && is_obb != h->cfg.is_obb) {
            h->cfg.is_obb = is_obb;
            if (h->inner) compat_build_inner(h, h->cfg.emb_dim);           // not stepped yet: the tables of the other layout
        }
        if (!h->inner && embs != nullptr && emb_cols > 0) compat_build_inner(h, emb_cols);
        if (!h->inner && det_rows > 0)
            throw std::runtime_error("BoTSORT: with_reid is set, no ReID weights were given and no embeddings were supplied.");
    });
    if (!ok) return 0;
    if (det_rows == 0 && det_cols == 0) det_cols = h->layout == 1 ? 7 : 6;
    if (!h->inner || (h->layout < 0 && det_rows == 0)) {        // nothing to track yet (no width / no layout known): the frame only counts
        h->empty_frames += 1;
        if (out_rows) *out_rows = 0;
        if (out_is_obb) *out_is_obb = h->layout == 1 ? 1 : 0;
        return 1;
    }
    int rc;
    if (h->empty_frames > 0) {      // the frames that went by count (a first detection on frame > 1 is not activated at once)
        const int fc = h->empty_frames;
        h->empty_frames = 0;
        // the reference's estimator saw the empty frames (botsort.py:141-145 runs cmc.apply on every frame); there were no tracks to
        // warp, so all that matters is that it sees THIS frame as its newest: run it although the frame counter is preset
        h->inner->cmc_with_fc_set = true;
        rc = boxmot_hip_botsort_update_stream(h->inner, 0, 0, fc, dets, det_rows, det_cols, embs, emb_rows, emb_cols, image,
                                              image_rows, image_cols, image_channels, out_tracks, out_capacity_rows, out_cols,
                                              out_rows, out_is_obb);
        h->inner->cmc_with_fc_set = false;
    } else {
        rc = boxmot_hip_botsort_update(h->inner, dets, det_rows, det_cols, embs, emb_rows, emb_cols, image, image_rows, image_cols,
                                       image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    }
    if (rc) h->stepped = true;
    return rc;
}
#define BM_COMPAT_TIME(name, fn)                                                         \
    int name(BoxMOTBotSortHandle* h, double* out) {                                      \
        if (h && !h->inner) { if (out) *out = 0.0; return 1; }                           \
        return fn(h ? h->inner : nullptr, out);                                          \
    }
BM_COMPAT_TIME(boxmot_botsort_last_reid_time_ms, boxmot_hip_botsort_last_reid_time_ms)
BM_COMPAT_TIME(boxmot_botsort_last_reid_preprocess_time_ms, boxmot_hip_botsort_last_reid_preprocess_time_ms)
BM_COMPAT_TIME(boxmot_botsort_last_reid_process_time_ms, boxmot_hip_botsort_last_reid_process_time_ms)
BM_COMPAT_TIME(boxmot_botsort_last_reid_postprocess_time_ms, boxmot_hip_botsort_last_reid_postprocess_time_ms)
#undef BM_COMPAT_TIME
const char* boxmot_botsort_last_error() { return g_last_error.c_str(); }

BoxMOTByteTrackHandle* boxmot_bytetrack_create(const BoxMOTByteTrackConfig* c) {
    BoxMOTByteTrackHandle* h = nullptr;
    const int ok = guard([&]() {
        if (c == nullptr) throw std::runtime_error("ByteTrack config is required.");
        BoxMOTHipBotSortConfig k;
        boxmot_hip_bytetrack_default_config(&k);
        k.track_low_thresh = (double)c->min_conf; k.track_high_thresh = (double)c->track_thresh;
        k.new_track_thresh = (double)c->track_thresh;       // det_thresh = track_thresh, bytetrack.py:250
        k.match_thresh = (double)c->match_thresh; k.track_buffer = c->track_buffer; k.frame_rate = c->frame_rate;
        k.max_obs = c->max_obs;
        k.max_tracks = env_int("BOXMOT_HIP_MAX_TRACKS", 1024);
        k.max_dets = env_int("BOXMOT_HIP_MAX_DETS", 512);
        h = new BoxMOTByteTrackHandle();
        h->cfg = k;
        h->inner = boxmot_hip_botsort_create(&k);
        if (!h->inner) throw std::runtime_error(g_last_error);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}
void boxmot_bytetrack_destroy(BoxMOTByteTrackHandle* h) { destroy_on(h); }
int boxmot_bytetrack_reset(BoxMOTByteTrackHandle* h) {
    if (h) { h->empty_frames = 0; h->stepped = false; h->layout = -1; }
    return boxmot_hip_botsort_reset(h ? h->inner : nullptr);
}
int boxmot_bytetrack_update(BoxMOTByteTrackHandle* h, const float* dets, int det_rows, int det_cols, const uint8_t* image,
                            int image_rows, int image_cols, int image_channels, float* out_tracks, int out_capacity_rows,
                            int out_cols, int* out_rows, int* out_is_obb) {
    if (!h) return boxmot_hip_botsort_update(nullptr, dets, det_rows, det_cols, nullptr, 0, 0, image, image_rows, image_cols,
                                             image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    const int ok = guard_on(h, [&]() {
        int is_obb = 0;
        if (compat_layout(h, det_rows, det_cols, is_obb, "ByteTrack") && is_obb != h->cfg.is_obb) {
            h->cfg.is_obb = is_obb;
            BoxMOTHipBotSort* inner = boxmot_hip_botsort_create(&h->cfg);          // not stepped yet: the tables of the other layout
            if (!inner) throw std::runtime_error(g_last_error);
            delete h->inner;
            h->inner = inner;
        }
    });
    if (!ok) return 0;
    if (det_rows == 0 && det_cols == 0) det_cols = h->layout == 1 ? 7 : 6;
    if (h->layout < 0 && det_rows == 0) {          // a 0 x 0 table before the layout is known: the frame only counts
        h->empty_frames += 1;
        if (out_rows) *out_rows = 0;
        if (out_is_obb) *out_is_obb = 0;
        return 1;
    }
    const int fc = h->empty_frames > 0 ? h->empty_frames : -1;
    h->empty_frames = 0;
    const int rc = boxmot_hip_botsort_update_stream(h->inner, 0, 0, fc, dets, det_rows, det_cols, nullptr, 0, 0, image, image_rows,
                                                    image_cols, image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    if (rc) h->stepped = true;
    return rc;
}
const char* boxmot_bytetrack_last_error() { return g_last_error.c_str(); }

BoxMOTOCSORTHandle* boxmot_ocsort_create(const BoxMOTOCSORTConfig* c) {
    BoxMOTOCSORTHandle* h = nullptr;
    const int ok = guard([&]() {
        if (c == nullptr) throw std::runtime_error("OCSORT config is required.");
        BoxMOTHipDeepOcSortConfig k;
        boxmot_hip_deepocsort_default_config(&k);
        k.embedding_off = 1; k.cmc_off = 1; k.aw_off = 1;          // OC-SORT = this step without appearance / camera terms
        k.min_conf = (double)c->min_conf; k.det_thresh = (double)c->det_thresh; k.iou_threshold = (double)c->iou_threshold;
        k.max_age = c->max_age; k.min_hits = c->min_hits; k.delta_t = c->delta_t; k.use_byte = c->use_byte ? 1 : 0;
        k.inertia = (double)c->inertia; k.Q_xy_scaling = (double)c->q_xy_scaling; k.Q_s_scaling = (double)c->q_s_scaling;
        k.max_obs = c->max_obs;
        k.max_tracks = env_int("BOXMOT_HIP_MAX_TRACKS", 1024);
        k.max_dets = env_int("BOXMOT_HIP_MAX_DETS", 512);
        h = new BoxMOTOCSORTHandle();
        h->cfg = k;
        h->inner = boxmot_hip_deepocsort_create(&k);
        if (!h->inner) throw std::runtime_error(g_last_error);
    });
    if (!ok) { delete h; return nullptr; }
    return h;
}
void boxmot_ocsort_destroy(BoxMOTOCSORTHandle* h) { destroy_on(h); }
int boxmot_ocsort_reset(BoxMOTOCSORTHandle* h) {
    if (h) { h->empty_frames = 0; h->stepped = false; h->layout = -1; }
    return boxmot_hip_deepocsort_reset(h ? h->inner : nullptr);
}
int boxmot_ocsort_update(BoxMOTOCSORTHandle* h, const float* dets, int det_rows, int det_cols, const uint8_t* image,
                         int image_rows, int image_cols, int image_channels, float* out_tracks, int out_capacity_rows,
                         int out_cols, int* out_rows, int* out_is_obb) {
    if (!h) return boxmot_hip_deepocsort_update(nullptr, dets, det_rows, det_cols, nullptr, 0, 0, image, image_rows, image_cols,
                                                image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    const int ok = guard_on(h, [&]() {
        int is_obb = 0;
        if (compat_layout(h, det_rows, det_cols, is_obb, "OCSORT") && is_obb != h->cfg.is_obb) {
            h->cfg.is_obb = is_obb;
            BoxMOTHipDeepOcSort* inner = boxmot_hip_deepocsort_create(&h->cfg);    // not stepped yet: the tables of the other layout
            if (!inner) throw std::runtime_error(g_last_error);
            delete h->inner;
            h->inner = inner;
        }
    });
    if (!ok) return 0;
    if (det_rows == 0 && det_cols == 0) det_cols = h->layout == 1 ? 7 : 6;
    if (h->layout < 0 && det_rows == 0) {          // a 0 x 0 table before the layout is known: the frame only counts
        h->empty_frames += 1;
        if (out_rows) *out_rows = 0;
        if (out_is_obb) *out_is_obb = 0;
        return 1;
    }
    const int fc = h->empty_frames > 0 ? h->empty_frames : -1;
    h->empty_frames = 0;
    const int rc = boxmot_hip_deepocsort_update_stream(h->inner, 0, fc, nullptr, dets, det_rows, det_cols, nullptr, 0, 0, image, image_rows,
                                                       image_cols, image_channels, out_tracks, out_capacity_rows, out_cols, out_rows, out_is_obb);
    if (rc) h->stepped = true;
    return rc;
}
const char* boxmot_ocsort_last_error() { return g_last_error.c_str(); }

// ---- boxmot_reid_capi_* (reid_capi.h:36-94) ----
// model_path: an OSN1 weight blob (the reference takes an ONNX file; this library reads the folded OSNet weights written by
// boxmot_amd.reid_weights.save_blob).  preprocess NULL -> "resize_pad", the reference's default (reid_capi.h:33-34).
// BOXMOT_HIP_REID_MODE selects the kernel family (0 per-layer fp32 = default, 1 fused fp16 MFMA), BOXMOT_HIP_REID_MAX_CROPS
// the per-call capacity (default 1024).
int boxmot_reid_capi_create(const char* model_path, const char* preprocess, void** out_handle) {
    return guard([&]() {
        if (out_handle == nullptr) throw std::runtime_error("out_handle is null.");
        *out_handle = nullptr;
        if (model_path == nullptr || !*model_path) throw std::runtime_error("ReID model path is required.");
        BoxMOTHipReID* h = boxmot_hip_reid_create(model_path, nullptr, 0, env_int("BOXMOT_HIP_REID_MAX_CROPS", 1024));
        if (!h) throw std::runtime_error(g_last_error);
        const std::string err_keep;
        if (!boxmot_hip_reid_set_preprocess(h, preprocess ? preprocess : "resize_pad")) {
            const std::string e = g_last_error; delete h; throw std::runtime_error(e);
        }
        const char* m = std::getenv("BOXMOT_HIP_REID_MODE");
        if (m && *m && !boxmot_hip_reid_set_mode(h, std::atoi(m))) { const std::string e = g_last_error; delete h; throw std::runtime_error(e); }
        *out_handle = h;
    });
}
void boxmot_reid_capi_destroy(void* handle) { destroy_on(static_cast<BoxMOTHipReID*>(handle)); }
int boxmot_reid_capi_feature_dim(void* handle, int* out_feature_dim) {
    return guard([&]() {
        if (!handle) throw std::runtime_error("ReID handle is null.");
        if (!out_feature_dim) throw std::runtime_error("out_feature_dim is null.");
        *out_feature_dim = boxmot_hip_reid_feature_dim(static_cast<BoxMOTHipReID*>(handle));
    });
}
int boxmot_reid_capi_compute_features(void* handle, const float* boxes_xyxy, int n_boxes, const uint8_t* image_data,
                                      int image_rows, int image_cols, int image_channels, float* out_features,
                                      int out_capacity_floats) {
    BoxMOTHipReID* h = static_cast<BoxMOTHipReID*>(handle);
    const int ok = guard([&]() {
        if (!h) throw std::runtime_error("ReID handle is null.");
        if (n_boxes > 0 && (long)out_capacity_floats < (long)n_boxes * h->engine->feature_dim())
            throw std::runtime_error("Output feature buffer is too small.");
    });
    if (!ok) return 0;
    // the reference takes any number of boxes per call: walk them in chunks of the engine's capacity
    const int step = h->engine->max_crops(), dim = h->engine->feature_dim();
    int i0 = 0;
    do {
        const int m = n_boxes - i0 < step ? n_boxes - i0 : step;
        if (!boxmot_hip_reid_compute_features(h, image_data, image_rows, image_cols, image_channels, boxes_xyxy + (size_t)i0 * 4, m, 4,
                                              out_features + (size_t)i0 * dim, m)) return 0;
        i0 += m;
    } while (i0 < n_boxes);
    return 1;
}
// Staged calls: preprocess stages frame + boxes on the device and (per-layer path) writes the normalised crop blob into the
// engine; process runs the forward pass into the handle's feature buffer (the head kernel already L2-normalises -- the
// norm is idempotent); postprocess copies the rows out.  preprocess -> process -> postprocess == compute_features, bit for bit.
int boxmot_reid_capi_preprocess(void* handle, const float* boxes_xyxy, int n_boxes, const uint8_t* image_data, int image_rows,
                                int image_cols, int image_channels) {
    BoxMOTHipReID* h = static_cast<BoxMOTHipReID*>(handle);
    return guard_on(h, [&]() {
        if (!h) throw std::runtime_error("ReID handle is null.");
        h->staged_n = -1;
        reid_stage(h, image_data, image_rows, image_cols, image_channels, boxes_xyxy, n_boxes, 4);
        h->staged_n = n_boxes; h->staged_rows = image_rows; h->staged_cols = image_cols; h->staged_done = false;
    });
}
int boxmot_reid_capi_process(void* handle) {
    BoxMOTHipReID* h = static_cast<BoxMOTHipReID*>(handle);
    return guard_on(h, [&]() {
        if (!h) throw std::runtime_error("ReID handle is null.");
        if (h->staged_n < 0) throw std::runtime_error("boxmot_reid_capi_process called before boxmot_reid_capi_preprocess.");
        if (h->staged_n > 0) {
            h->engine->run(h->d_frames, h->d_crop_stream, h->d_boxes, 4, h->staged_n, h->staged_cols, h->staged_rows, h->d_feat,
                           nullptr, h->stream);
            BM_HIP(hipStreamSynchronize(h->stream));
        }
        h->staged_done = true;
    });
}
int boxmot_reid_capi_postprocess(void* handle, float* out_features, int out_capacity_floats) {
    BoxMOTHipReID* h = static_cast<BoxMOTHipReID*>(handle);
    return guard_on(h, [&]() {
        if (!h) throw std::runtime_error("ReID handle is null.");
        if (h->staged_n < 0 || !h->staged_done) throw std::runtime_error("boxmot_reid_capi_postprocess called before boxmot_reid_capi_process.");
        const long need = (long)h->staged_n * h->engine->feature_dim();
        if ((long)out_capacity_floats < need) throw std::runtime_error("Output feature buffer is too small.");
        if (need) BM_HIP(hipMemcpy(out_features, h->d_feat, (size_t)need * 4, hipMemcpyDeviceToHost));
        h->staged_n = -1; h->staged_done = false;
    });
}
const char* boxmot_reid_capi_last_error(void) { return g_last_error.c_str(); }

}  // extern "C"


