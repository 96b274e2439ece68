// TEST-ONLY harness: runs the ECC camera-motion kernels (boxmot_amd/csrc/cmc_ecc.hpp, the device source unchanged) on CPU threads
// for one stream: preprocess of two BGR frames, gradients of the second, the Gauss-Newton solve.
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/cmc_ecc.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuDim3 gridDim;
EmuBlock* g_emu_block = nullptr;
unsigned char* g_emu_dynamic_lds = nullptr;

namespace {
struct TA { const std::function<void()>* fn; int tid, bx, by; };
void* tmain(void* p) {
    TA* a = static_cast<TA*>(p);
    threadIdx.x = a->tid; blockIdx.x = a->bx; blockIdx.y = a->by;
    (*a->fn)();
    return nullptr;
}
void launch(int gx, int gy, int nthr, const std::function<void()>& fn) {
    static EmuBlock block;
    g_emu_block = &block;
    blockDim.x = nthr; gridDim.x = gx; gridDim.y = gy;
    block.block_barrier.init(nthr);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            std::vector<TA> ta(nthr);
            for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, bx, by};
            emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 19);
        }
}
}  // namespace

// prev / curr: BGR uint8 [rows][cols][3]; out_small (optional): the two preprocessed images fp32 [2][h][w]
extern "C" int emu_ecc(const uint8_t* prev, const uint8_t* curr, int rows, int cols, double scale, double eps, int max_iter,
                       double* out_warp6, int* out_info2, float* out_small) {
    using namespace bm;
    const int w = (int)std::nearbyint(cols * scale), h = (int)std::nearbyint(rows * scale), P = h * w;
    std::vector<float> a(P), b(P), gx(P), gy(P), scratch(3 * (size_t)P);
    const double inv = 1.0 / scale;
    { const uint8_t* fr[1] = {prev}; const uint8_t* const* f = fr; float* o = a.data();
      launch((P + 255) / 256, 1, 256, [=]() { k_ecc_preprocess(f, o, (long)P, rows, cols, h, w, inv); }); }
    { const uint8_t* fr[1] = {curr}; const uint8_t* const* f = fr; float* o = b.data();
      launch((P + 255) / 256, 1, 256, [=]() { k_ecc_preprocess(f, o, (long)P, rows, cols, h, w, inv); }); }
    { const float* i = b.data(); float* x = gx.data(); float* y = gy.data();
      launch((P + 255) / 256, 1, 256, [=]() { k_ecc_gradients(i, (long)P, x, y, h, w); }); }
    { const float *t = a.data(), *i = b.data(), *x = gx.data(), *y = gy.data(); float* sc = scratch.data();
      launch(1, 1, ECC_THREADS, [=]() { k_ecc_solve(t, i, (long)P, x, y, sc, out_warp6, out_info2, h, w, eps, max_iter, (float)scale); }); }
    if (out_small) { std::memcpy(out_small, a.data(), P * 4); std::memcpy(out_small + P, b.data(), P * 4); }
    return 0;
}
