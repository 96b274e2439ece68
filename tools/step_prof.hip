// Standalone frame-step timer over the C ABI (include/boxmot_hip.h): the device-resident step of one tracker (embeddings supplied:
// tracker math only, mode M1) on inputs written by tools/make_step_inputs.py.  Development tool: an A/B of a step-kernel or solver
// variant costs ~1 GPU-minute instead of a Python session (tools/tracker_bench.py measures the same loop and adds the oracle gate).
//   python tools/make_step_inputs.py --tracker deepocsort --config c3          # on the build container, no GPU
//   hipcc -O2 -std=c++17 -I include tools/step_prof.hip -o tools/_build/step_prof -L boxmot_amd -lboxmot_hip -Wl,-rpath,'$ORIGIN/../../boxmot_amd'
//   tools/_build/step_prof tools/_build/steps_deepocsort_c3.bin [--parse-only]
// Prints ms per step (one step = one frame of every stream), stream-frames per second and a checksum of the output rows of all
// timed steps (equal across library variants = same rows on the device).  --parse-only reads and summarises the file without
// touching the GPU (runs in the build container).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "boxmot_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ABI(x) do { if (!(x)) { fprintf(stderr, "%s failed: %s\n", #x, boxmot_hip_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: step_prof <steps_*.bin> [--parse-only]\n"); return 2; }
    const bool parse_only = argc > 2 && !strcmp(argv[2], "--parse-only");
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    char magic[8];
    int32_t hdr[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BMSTEP01", 8) || fread(hdr, 4, 8, f) != 8) { fprintf(stderr, "not a BMSTEP01 file\n"); return 1; }
    const int kind = hdr[0], S = hdr[1], T = hdr[2], warmup = hdr[3], nd = hdr[4], dim = hdr[5], cap = hdr[6], cfg_len = hdr[7];
    std::vector<unsigned char> cfg(cfg_len);
    if (fread(cfg.data(), 1, cfg_len, f) != (size_t)cfg_len) { fprintf(stderr, "truncated file (config)\n"); return 1; }
    const size_t want_cfg = kind == 0 ? sizeof(BoxMOTHipBotSortConfig) : (kind == 1 ? sizeof(BoxMOTHipDeepOcSortConfig) : sizeof(BoxMOTHipStrongSortConfig));
    if ((size_t)cfg_len != want_cfg) { fprintf(stderr, "configuration struct is %d bytes in the file, %zu in this header: regenerate the inputs\n", cfg_len, want_cfg); return 1; }
    std::vector<int32_t> cnt((size_t)T * S);
    if (fread(cnt.data(), 4, cnt.size(), f) != cnt.size()) { fprintf(stderr, "truncated file (counts)\n"); return 1; }
    std::vector<float> dets((size_t)T * S * nd * 6, 0.f), embs((size_t)T * S * nd * dim, 0.f);
    long rows = 0;
    for (int t = 0; t < T; ++t)
        for (int s = 0; s < S; ++s) {
            const int n = cnt[(size_t)t * S + s];
            if (n < 0 || n > nd) { fprintf(stderr, "bad row count %d at frame %d stream %d\n", n, t, s); return 1; }
            float* d = dets.data() + ((size_t)t * S + s) * nd * 6;
            float* e = embs.data() + ((size_t)t * S + s) * nd * dim;
            if (fread(d, 4, (size_t)n * 6, f) != (size_t)n * 6 || fread(e, 4, (size_t)n * dim, f) != (size_t)n * dim) { fprintf(stderr, "truncated file (rows)\n"); return 1; }
            rows += n;
        }
    fclose(f);
    static const char* NAMES[3] = {"botsort", "deepocsort", "strongsort"};
    printf("%s: %d streams x %d frames (%d warm-up), %d-row detection slots, %d-d embeddings, %d track rows, %ld detections in all\n",
           NAMES[kind], S, T, warmup, nd, dim, cap, rows);
    if (parse_only) { printf("first detection: %.2f %.2f %.2f %.2f conf %.3f cls %.0f\n", dets[0], dets[1], dets[2], dets[3], dets[4], dets[5]); return 0; }

    float *d_dets, *d_embs, *d_out;
    int32_t *d_cnt, *d_out_n;
    const int out_rows = kind == 0 ? nd : cap;          // BoT-SORT emits at most one row per detection, the others per track
    CK(hipMalloc(&d_dets, dets.size() * 4)); CK(hipMalloc(&d_embs, embs.size() * 4)); CK(hipMalloc(&d_cnt, cnt.size() * 4));
    CK(hipMalloc(&d_out, (size_t)T * S * out_rows * 8 * 4)); CK(hipMalloc(&d_out_n, (size_t)T * S * 4));
    CK(hipMemcpy(d_dets, dets.data(), dets.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_embs, embs.data(), embs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0, (size_t)T * S * out_rows * 8 * 4)); CK(hipMemset(d_out_n, 0, (size_t)T * S * 4));
    BoxMOTHipBotSort* hb = nullptr; BoxMOTHipDeepOcSort* hd = nullptr; BoxMOTHipStrongSort* hs = nullptr;
    if (kind == 0) { BoxMOTHipBotSortConfig c; memcpy(&c, cfg.data(), sizeof(c)); ABI(hb = boxmot_hip_botsort_create(&c)); }
    else if (kind == 1) { BoxMOTHipDeepOcSortConfig c; memcpy(&c, cfg.data(), sizeof(c)); ABI(hd = boxmot_hip_deepocsort_create(&c)); }
    else { BoxMOTHipStrongSortConfig c; memcpy(&c, cfg.data(), sizeof(c)); ABI(hs = boxmot_hip_strongsort_create(&c)); }
    auto step = [&](int t) {
        const float* dd = d_dets + (size_t)t * S * nd * 6;
        const float* de = d_embs + (size_t)t * S * nd * dim;
        const int32_t* dc = d_cnt + (size_t)t * S;
        float* o = d_out + (size_t)t * S * out_rows * 8;
        int32_t* on = d_out_n + (size_t)t * S;
        if (kind == 0) ABI(boxmot_hip_botsort_step_device(hb, dd, dc, de, nullptr, 1080, 1920, o, on));
        else if (kind == 1) ABI(boxmot_hip_deepocsort_step_device(hd, dd, dc, de, o, on));
        else ABI(boxmot_hip_strongsort_step_device(hs, dd, dc, de, o, on));
    };
    auto sync = [&]() {
        if (kind == 0) ABI(boxmot_hip_botsort_synchronize(hb));
        else if (kind == 1) ABI(boxmot_hip_deepocsort_synchronize(hd));
        else ABI(boxmot_hip_strongsort_synchronize(hs));
    };
    for (int t = 0; t < warmup; ++t) step(t);
    sync();
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = warmup; t < T; ++t) step(t);
    sync();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int K = T - warmup;
    std::vector<float> out((size_t)T * S * out_rows * 8);
    std::vector<int32_t> out_n((size_t)T * S);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out_n.data(), d_out_n, out_n.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    long out_total = 0;
    for (int t = warmup; t < T; ++t)
        for (int s = 0; s < S; ++s) {
            const int n = out_n[(size_t)t * S + s];
            out_total += n;
            const float* r = out.data() + ((size_t)t * S + s) * out_rows * 8;
            for (int k = 0; k < n * 8; ++k) { unsigned u; memcpy(&u, r + k, 4); sum += (unsigned long long)u * ((k + 31 * t + 7 * s) % 8191 + 1); }
        }
    printf("%s step: %.3f ms per step of %d stream-frames = %.1f stream-frames/s; %ld output rows, checksum %016llx\n", NAMES[kind], 1e3 * dt / K, S,
           S * K / dt, out_total, sum);
    if (hb) boxmot_hip_botsort_destroy(hb);
    if (hd) boxmot_hip_deepocsort_destroy(hd);
    if (hs) boxmot_hip_strongsort_destroy(hs);
    return 0;
}
