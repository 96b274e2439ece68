// Micro-benchmark (development tool): do MFMAs overlap with vector arithmetic on one SIMD?  One 512-thread workgroup (two waves per SIMD),
// per iteration M MFMAs on independent accumulators (v_mfma_f32_16x16x32_f16 or v_mfma_f32_16x16x4_f32) interleaved with F v_pk_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap_bench.hip -o tools/_build/mfma_valu_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// KIND 0: f16 16x16x32, 1: f32 16x16x4.  M MFMAs (chains of DEP dependent ones), F pk_fma per MFMA interleaved after each MFMA
template <int KIND, int M, int DEP, int F>
__global__ void __launch_bounds__(512, 2) k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc[M / DEP > 0 ? M / DEP : 1];
    for (auto& a : acc) a = f4{0, 0, 0, 0};
    h8 a16, b16;
    for (int j = 0; j < 8; ++j) { a16[j] = (_Float16)(0.01f * (lane + j)); b16[j] = (_Float16)(0.02f * j); }
    float a32 = 0.01f * lane, b32 = 0.5f;
    f4 v[4] = {f4{1, 2, 3, 4}, f4{2, 3, 4, 5}, f4{3, 4, 5, 6}, f4{4, 5, 6, 7}};
    const f4 w = {1.0001f, 0.9999f, 1.0002f, 0.9998f};
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if constexpr (KIND == 0) acc[m / DEP] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc[m / DEP], 0, 0, 0);
            else acc[m / DEP] = __builtin_amdgcn_mfma_f32_16x16x4f32(a32, b32, acc[m / DEP], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F / 2; ++f) v[f & 3] = __builtin_elementwise_fma(w, v[(f + 1) & 3], v[f & 3]);
        }
        if (M == 0) {
#pragma unroll
            for (int f = 0; f < F / 2; ++f) v[f & 3] = __builtin_elementwise_fma(w, v[(f + 1) & 3], v[f & 3]);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 s = v[0] + v[1] + v[2] + v[3];
    for (auto& a : acc) s += a;
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cyc[wave] = t1 - t0;
}
template <class K>
static int run(const char* name, K kern, int threads) {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 512 * 4)); CK(hipMalloc(&cyc, 8 * 8));
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, iters); CK(hipDeviceSynchronize()); }
    std::vector<long long> h(8);
    CK(hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost));
    long long mx = 0; for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-70s waves/SIMD %d  cycles per iteration %.1f\n", name, threads / 256, (double)mx / iters);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}
#define R2(name, ...) for (int t : {256, 512}) if (run(name, k<__VA_ARGS__>, t)) return 1;
int main() {
    R2("f16 16x16x32: 8 independent MFMAs, no arithmetic", 0, 8, 1, 0)
    R2("f16 16x16x32: 8 MFMAs in 4 chains of 2, no arithmetic", 0, 8, 2, 0)
    R2("arithmetic only: 32 v_pk_fma_f32 (M=0 -> F per iteration)", 0, 0, 1, 32)
    R2("f16: 8 independent MFMAs + 4 v_pk_fma_f32 after each (32)", 0, 8, 1, 4)
    R2("f16: 8 MFMAs (chains of 2) + 4 v_pk_fma_f32 after each (32)", 0, 8, 2, 4)
    R2("f16: 8 independent MFMAs + 8 v_pk_fma_f32 after each (64)", 0, 8, 1, 8)
    R2("arithmetic only: 64 v_pk_fma_f32", 0, 0, 1, 64)
    R2("f32 16x16x4: 8 independent MFMAs, no arithmetic", 1, 8, 1, 0)
    R2("f32 16x16x4: 8 MFMAs in 2 chains of 4, no arithmetic", 1, 8, 4, 0)
    R2("f32: 8 independent MFMAs + 4 v_pk_fma_f32 after each (32)", 1, 8, 1, 4)
    R2("f32: 8 independent MFMAs + 8 v_pk_fma_f32 after each (64)", 1, 8, 1, 8)
    R2("f32: 8 MFMAs (chains of 4) + 8 v_pk_fma_f32 after each (64)", 1, 8, 4, 8)
    return 0;
}
