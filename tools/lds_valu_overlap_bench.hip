// Micro-benchmark (development tool): do LDS reads / writes overlap with vector arithmetic of the SAME workgroup on one CU?
// One 512-thread workgroup (two waves per SIMD, like the stage-0 / stage-1 block kernels), per iteration R ds_read_b128 and F v_pk_fma_f32
// per wave, in four arrangements: arithmetic only, reads only (pipelined), both with the reads of iteration i + 1 issued before the
// arithmetic of iteration i, both with every read waited for before its arithmetic.  Also: W ds_write_b128 + arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_valu_overlap_bench.hip -o tools/_build/lds_valu_overlap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0 arithmetic only, 1 reads only, 2 pipelined, 3 read-wait-compute, 4 writes + arithmetic, 5 writes only
template <int MODE, int R, int F>
__global__ void __launch_bounds__(512, 2) k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 8192; e += blockDim.x) reinterpret_cast<f4*>(lds)[e] = f4{1.f, 2.f, 3.f, (float)e};
    __syncthreads();
    const f4* base = reinterpret_cast<const f4*>(lds) + lane + wave * 64;
    f4* wbase = reinterpret_cast<f4*>(lds) + lane + wave * 64;
    f4 acc[4] = {f4{0, 0, 0, 0}, f4{1, 1, 1, 1}, f4{2, 2, 2, 2}, f4{3, 3, 3, 3}};
    const f4 w = {1.0001f, 0.9999f, 1.0002f, 0.9998f};
    f4 cur[R], nxt[R];
    for (int r = 0; r < R; ++r) cur[r] = base[r * 512];
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) nxt[r] = base[((it + r) & 15) * 512];
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE == 3) {
#pragma unroll
            for (int r = 0; r < R; ++r) cur[r] = base[((it + r) & 15) * 512];
        }
        if constexpr (MODE == 4 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < R; ++r) wbase[((it + r) & 15) * 512] = acc[r & 3];
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE != 1 && MODE != 5) {
#pragma unroll
            for (int f = 0; f < F / 2; ++f) acc[f & 3] = __builtin_elementwise_fma(w, cur[f % R], acc[f & 3]);      // one f4 fma = two v_pk_fma_f32
        }
        if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) { if (MODE == 1) asm volatile("" :: "v"(nxt[r][0]), "v"(nxt[r][3])); cur[r] = nxt[r]; }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int r = 0; r < R; ++r) s += cur[r];
    out[threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cyc[wave] = t1 - t0;
}

template <class K>
static int run(const char* name, K kern) {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 512 * 4)); CK(hipMalloc(&cyc, 8 * 8));
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(1), dim3(512), 160 * 1024, 0, out, cyc, iters); CK(hipDeviceSynchronize()); }
    std::vector<long long> h(8);
    CK(hipMemcpy(h.data(), cyc, 64, hipMemcpyDeviceToHost));
    long long mx = 0; for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-62s cycles per iteration (slowest wave) %.1f\n", name, (double)mx / iters);
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}
#define SET(R, F) \
    if (run("R=" #R " F=" #F "  arithmetic only", k<0, R, F>)) return 1; \
    if (run("R=" #R " F=" #F "  reads only (pipelined)", k<1, R, F>)) return 1; \
    if (run("R=" #R " F=" #F "  reads one iteration ahead + arithmetic", k<2, R, F>)) return 1; \
    if (run("R=" #R " F=" #F "  read, wait, arithmetic", k<3, R, F>)) return 1; \
    if (run("R=" #R " F=" #F "  writes + arithmetic", k<4, R, F>)) return 1; \
    if (run("R=" #R " F=" #F "  writes only", k<5, R, F>)) return 1;
int main() {
    for (auto* kern : {(const void*)k<0, 3, 24>, (const void*)k<1, 3, 24>, (const void*)k<2, 3, 24>, (const void*)k<3, 3, 24>, (const void*)k<4, 3, 24>, (const void*)k<5, 3, 24>,
                       (const void*)k<0, 6, 48>, (const void*)k<1, 6, 48>, (const void*)k<2, 6, 48>, (const void*)k<3, 6, 48>, (const void*)k<4, 6, 48>, (const void*)k<5, 6, 48>,
                       (const void*)k<0, 3, 48>, (const void*)k<1, 3, 48>, (const void*)k<2, 3, 48>, (const void*)k<3, 3, 48>, (const void*)k<4, 3, 48>, (const void*)k<5, 3, 48>})
        CK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SET(3, 24)
    SET(6, 48)
    SET(3, 48)
    return 0;
}
