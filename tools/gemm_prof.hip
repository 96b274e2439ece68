// GEMM kernels of gemm_f16.hpp on CLIP-ReID's shapes: correctness against a CPU product on sampled outputs, time, TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I boxmot_amd/csrc tools/gemm_prof.hip -o tools/_build/gemm_prof && tools/_build/gemm_prof [crops]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_f16.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }

template <class Launch>
static float time_ms(Launch launch, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms / iters < best ? ms / iters : best;
    }
    return best;
}

int main(int argc, char** argv) {
    const int crops = argc > 1 ? atoi(argv[1]) : 256;
    const long M = (long)crops * 129;
    const int shapes[4][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};       // N, K: qkv, proj, fc1, fc2
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_glds<0, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::gemm_glds_lds_bytes<64>()));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_256<0>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::GEMM256_LDS_BYTES));
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<_Float16> hx((size_t)M * K), hw((size_t)N * K);
        std::vector<float> hb(N);
        unsigned s = 7;
        for (auto& v : hx) v = (_Float16)frand(s);
        for (auto& v : hw) v = (_Float16)(frand(s) * 0.1f);
        for (auto& v : hb) v = frand(s);
        _Float16 *dx, *dw, *dc0, *dc1; float* db;
        CK(hipMalloc(&dx, hx.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&db, N * 4));
        CK(hipMalloc(&dc0, (size_t)M * N * 2)); CK(hipMalloc(&dc1, (size_t)M * N * 2));
        CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemset(dc0, 0, (size_t)M * N * 2)); CK(hipMemset(dc1, 0, (size_t)M * N * 2));
        const unsigned g128 = (unsigned)(((M + 127) / 128) * (N / 128)), g256 = (unsigned)(((M + 255) / 256) * (N / 256));
        auto old_k = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_glds<0, 64>), dim3(g128), dim3(256), bm::gemm_glds_lds_bytes<64>(), 0, dx, dw, db, dc0, (const _Float16*)nullptr, (int)M, N, K, 0, bm::GemmExt{}); };
        auto new_k = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_256<0>), dim3(g256), dim3(512), bm::GEMM256_LDS_BYTES, 0, dx, dw, db, dc1, (const _Float16*)nullptr, (int)M, N, K, 0); };
        const float t0 = time_ms(old_k, 10), t1 = time_ms(new_k, 10);
        CK(hipGetLastError());
        std::vector<_Float16> c0((size_t)M * N), c1((size_t)M * N);
        CK(hipMemcpy(c0.data(), dc0, c0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), dc1, c1.size() * 2, hipMemcpyDeviceToHost));
        // every element against the old kernel (same arithmetic: fp32 accumulation, k order differs only inside MFMA groups), samples against the CPU
        double max_pair = 0, max_cpu = 0;
        for (size_t i = 0; i < c0.size(); ++i) { const double d = std::fabs((double)(float)c0[i] - (double)(float)c1[i]); max_pair = d > max_pair ? d : max_pair; }
        unsigned q = 99;
        for (int k = 0; k < 4000; ++k) {
            q = q * 1664525u + 1013904223u; const long m = (q >> 4) % M;
            q = q * 1664525u + 1013904223u; const int n = (q >> 4) % N;
            double acc = hb[n];
            for (int kk = 0; kk < K; ++kk) acc += (double)(float)hx[m * K + kk] * (double)(float)hw[(size_t)n * K + kk];
            const double d = std::fabs(acc - (double)(float)c1[m * N + n]);
            max_cpu = d > max_cpu ? d : max_cpu;
        }
        {   // the two other epilogues the ViT uses: QuickGELU (fp16 out) and fp32 accumulate into the residual stream
            float* df; CK(hipMalloc(&df, (size_t)M * N * 4)); CK(hipMemset(df, 0, (size_t)M * N * 4));
            static bool once = false;
            if (!once) {
                once = true;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_glds<1, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::gemm_glds_lds_bytes<64>()));
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_glds<2, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::gemm_glds_lds_bytes<64>()));
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_256<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::GEMM256_LDS_BYTES));
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_gemm_f16_256<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::GEMM256_LDS_BYTES));
            }
            auto o1 = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_glds<1, 64>), dim3(g128), dim3(256), bm::gemm_glds_lds_bytes<64>(), 0, dx, dw, db, dc0, (const _Float16*)nullptr, (int)M, N, K, 0, bm::GemmExt{}); };
            auto n1 = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_256<1>), dim3(g256), dim3(512), bm::GEMM256_LDS_BYTES, 0, dx, dw, db, dc1, (const _Float16*)nullptr, (int)M, N, K, 0); };
            auto o2 = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_glds<2, 64>), dim3(g128), dim3(256), bm::gemm_glds_lds_bytes<64>(), 0, dx, dw, db, df, (const _Float16*)nullptr, (int)M, N, K, 0, bm::GemmExt{}); };
            auto n2 = [&]() { hipLaunchKernelGGL((bm::k_gemm_f16_256<2>), dim3(g256), dim3(512), bm::GEMM256_LDS_BYTES, 0, dx, dw, db, df, (const _Float16*)nullptr, (int)M, N, K, 0); };
            const float a1 = time_ms(o1, 10), b1 = time_ms(n1, 10);
            std::vector<_Float16> e0((size_t)M * N), e1((size_t)M * N);
            CK(hipMemcpy(e0.data(), dc0, e0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(e1.data(), dc1, e1.size() * 2, hipMemcpyDeviceToHost));
            double dg = 0;
            for (size_t i = 0; i < e0.size(); ++i) { const double d = std::fabs((double)(float)e0[i] - (double)(float)e1[i]); dg = d > dg ? d : dg; }
            const float a2 = time_ms(o2, 10), b2 = time_ms(n2, 10);
            // accumulate check: zero, one launch of each kernel, compare
            std::vector<float> f0((size_t)M * N), f1((size_t)M * N);
            CK(hipMemset(df, 0, (size_t)M * N * 4)); o2(); CK(hipMemcpy(f0.data(), df, f0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemset(df, 0, (size_t)M * N * 4)); n2(); n2(); CK(hipMemcpy(f1.data(), df, f1.size() * 4, hipMemcpyDeviceToHost));
            double da = 0;
            for (size_t i = 0; i < f0.size(); ++i) { const double d = std::fabs(2.0 * (double)f0[i] - (double)f1[i]); da = d > da ? d : da; }
            printf("    QuickGELU epilogue: %.3f -> %.3f ms (max|new-old| %.5f)   fp32 accumulate epilogue: %.3f -> %.3f ms (max|2 old - new twice| %.6f)\n", a1, b1, dg, a2, b2, da);
            hipFree(df);
        }
        const double fl = 2.0 * M * N * K;
        printf("M=%ld N=%d K=%d: 128x128 glds %.3f ms %.0f TF | 256x256 phased %.3f ms %.0f TF | max|new-old| %.4f max|new-cpu| %.4f\n", M, N, K, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, max_pair, max_cpu);
        hipFree(dx); hipFree(dw); hipFree(db); hipFree(dc0); hipFree(dc1);
    }
    return 0;
}
