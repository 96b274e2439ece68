// Micro-benchmark (development tool): which vector instructions run BESIDE the matrix pipe / the LDS return path on one SIMD?
// One 512-thread workgroup (two waves per SIMD).  Per iteration a wave issues a block B (8 x v_mfma_f32_16x16x32_f16 on independent
// accumulators, or 6 x ds_read_b128 one iteration ahead, or nothing) interleaved with 32 instructions of one vector class.
// Reported: cycles per iteration alone and together; "overlap" = 1 - (together - max) / min  (1 = fully hidden, 0 = serial).
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_bench.hip -o tools/_build/coissue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CLS>
__device__ inline void vop(float (&s)[8], f2 (&p)[4], int i) {
    if constexpr (CLS == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
    if constexpr (CLS == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
    if constexpr (CLS == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(s[i & 7]) : "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
    if constexpr (CLS == 3) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(s[i & 7]) : "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
    if constexpr (CLS == 4) asm volatile("v_max_i32 %0, 0, %1" : "=v"(s[i & 7]) : "v"(s[(i + 1) & 7]));
    if constexpr (CLS == 5) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(s[i & 7]) : "v"(s[(i + 3) & 7]));
    if constexpr (CLS == 6) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
    if constexpr (CLS == 7) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s[i & 7]) : "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
    if constexpr (CLS == 8) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(s[i & 7]) : "v"(s[(i + 1) & 7]), "v"(s[(i + 2) & 7]));
}

// BLK 0 none, 1 MFMA f16 x8, 2 ds_read_b128 x6 (pipelined), 3 ds_write_b128 x4
template <int BLK, int CLS, int NV>
__global__ void __launch_bounds__(512, 2) k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) reinterpret_cast<f4*>(lds)[e] = f4{1.f, 2.f, 3.f, (float)e};
    __syncthreads();
    const f4* base = reinterpret_cast<const f4*>(lds) + lane + wave * 64;
    f4* wbase = reinterpret_cast<f4*>(lds) + lane + wave * 64;
    float s[8]; f2 p[4];
    for (int i = 0; i < 8; ++i) s[i] = 0.001f * (lane + i);
    for (int i = 0; i < 4; ++i) p[i] = f2{0.5f + i, 0.25f * lane};
    f4 acc[8];
    for (auto& a : acc) a = f4{0, 0, 0, 0};
    h8 a16, b16;
    for (int j = 0; j < 8; ++j) { a16[j] = (_Float16)(0.01f * (lane + j)); b16[j] = (_Float16)(0.02f * j); }
    f4 ld[6];
    for (auto& v : ld) v = f4{0, 0, 0, 0};
    f4 keep = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (BLK == 2) {
#pragma unroll
            for (int r = 0; r < 6; ++r) { keep += ld[r]; ld[r] = base[((it + r) & 15) * 512]; }      // (6 pk_add of the previous values: the use)
        }
        if constexpr (BLK == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) wbase[((it + r) & 15) * 512] = acc[r];
        }
        constexpr int NB = BLK == 1 ? 8 : 1;
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            if constexpr (BLK == 1) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16, b16, acc[m], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NV / NB; ++f) vop<CLS>(s, p, m * (NV / NB) + f);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 r = keep;
    for (auto& a : acc) r += a;
    for (auto& v : ld) r += v;
    float q = r[0] + r[1] + r[2] + r[3];
    for (int i = 0; i < 8; ++i) q += s[i];
    for (int i = 0; i < 4; ++i) q += p[i][0] + p[i][1];
    out[threadIdx.x] = q;
    if (lane == 0) cyc[wave] = t1 - t0;
}
template <class K>
static double run(K kern) {
    float* out; long long* cyc;
    if (hipMalloc(&out, 512 * 4) != hipSuccess || hipMalloc(&cyc, 64) != hipSuccess) return -1;
    const int iters = 4000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(512), 160 * 1024, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
    std::vector<long long> h(8);
    (void)hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
    (void)hipFree(out); (void)hipFree(cyc);
    return (double)mx / iters;
}
template <int CLS>
static void row(const char* name) {
    const double v = run(k<0, CLS, 32>), m = run(k<1, CLS, 0>), vm = run(k<1, CLS, 32>), l = run(k<2, CLS, 0>), vl = run(k<2, CLS, 32>),
                 w = run(k<3, CLS, 0>), vw = run(k<3, CLS, 32>);
    auto ov = [](double a, double b, double ab) { const double mx = a > b ? a : b, mn = a > b ? b : a; return 1.0 - (ab - mx) / mn; };
    printf("%-18s alone %6.1f | 8 MFMA %6.1f together %6.1f overlap %5.2f | 6 ds_read_b128 %6.1f together %6.1f overlap %5.2f | 4 ds_write_b128 %6.1f together %6.1f overlap %5.2f\n",
           name, v, m, vm, ov(v, m, vm), l, vl, ov(v, l, vl), w, vw, ov(v, w, vw));
}
int main() {
    printf("# 32 instructions of one class per wave and iteration, two waves per SIMD; cycles per iteration (slowest wave)\n");
    row<0>("v_pk_fma_f32"); row<1>("v_fma_f32"); row<2>("v_cvt_pk_f16_f32"); row<3>("v_fma_mix_f32"); row<4>("v_max_i32");
    row<5>("v_mov_b32_dpp"); row<6>("v_pk_add_f32"); row<7>("v_mul_f32"); row<8>("v_pk_fma_f16");
    return 0;
}
