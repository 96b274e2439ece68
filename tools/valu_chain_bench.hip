// Micro-benchmark (development tool): issue / dependent-chain cost of the vector instructions the fp32-grade depthwise pass is made of,
// on one CU with 1, 2 or 4 waves per SIMD.  Cycles per instruction per wave from s_memtime around an unrolled block.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_chain_bench.hip -o tools/_build/valu_chain_bench && tools/_build/valu_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int CHAINS>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    f2 a[CHAINS], w = {1.0001f, 0.9999f}, v = {0.5f + lane, 0.25f};
    float s[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { a[c] = f2{(float)c, (float)lane}; s[c] = c + lane; }
    f4* lp = reinterpret_cast<f4*>(lds) + threadIdx.x;
    *lp = f4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(w), "v"(v));
                if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[c]) : "v"(w[0]), "v"(v[0]));
                if constexpr (MODE == 2) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[c]));
                if constexpr (MODE == 3) asm volatile("v_max_i32 %0, 0, %0" : "+v"(s[c]));
                if constexpr (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[c]) : "v"(w));
                if constexpr (MODE == 5) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[c]) : "v"(v[0]), "v"(w[0]));
                if constexpr (MODE == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[c]) : "v"(w));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int c = 0; c < CHAINS; ++c) r += a[c][0] + a[c][1] + s[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS reads: `depth` ds_read_b128 in flight per wait
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_lds(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) reinterpret_cast<f4*>(lds)[e] = f4{1.f, 2.f, 3.f, (float)e};
    __syncthreads();
    f4 acc = {0, 0, 0, 0};
    const f4* base = reinterpret_cast<const f4*>(lds) + lane + wave * 64;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = base[((it * DEPTH + d) & 7) * 1024];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}
template <int DEPTH>
__global__ void __launch_bounds__(1024) k_ldsw(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 val = {1.f, 2.f, 3.f, (float)lane};
    f4* base = reinterpret_cast<f4*>(lds) + lane + wave * 64;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { base[((it * DEPTH + d) & 7) * 1024] = val; val[0] += 1.f; }
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = reinterpret_cast<float*>(lds)[threadIdx.x];
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <class K>
static int run(const char* name, K kern, int waves, int n_inst_per_iter, size_t lds_bytes) {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1024 * 4)); CK(hipMalloc(&cyc, 16 * 8));
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), lds_bytes, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(1), dim3(64 * waves), lds_bytes, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(16);
    CK(hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost));
    long long mx = 0; for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    // s_memtime / readcyclecounter ticks at 100 MHz on this part? print raw and per-instruction
    printf("%-44s waves/CU %2d  ticks %8lld  ticks per instr per wave %.3f\n", name, waves, mx, (double)mx / ((double)iters * n_inst_per_iter));
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

#define RUNV(MODE, CH, NAME) for (int w : {4, 8, 16}) if (run(NAME, k<MODE, CH>, w, 16 * CH, 65536)) return 1;
int main() {
    RUNV(0, 1, "v_pk_fma_f32 1 dependent chain");
    RUNV(0, 2, "v_pk_fma_f32 2 chains");
    RUNV(0, 4, "v_pk_fma_f32 4 chains");
    RUNV(0, 8, "v_pk_fma_f32 8 chains");
    RUNV(1, 1, "v_fma_f32 1 dependent chain");
    RUNV(1, 2, "v_fma_f32 2 chains");
    RUNV(1, 4, "v_fma_f32 4 chains");
    RUNV(1, 8, "v_fma_f32 8 chains");
    RUNV(2, 1, "v_mov_b32_dpp 1 chain");
    RUNV(2, 4, "v_mov_b32_dpp 4 chains");
    RUNV(2, 8, "v_mov_b32_dpp 8 chains");
    RUNV(3, 8, "v_max_i32 8 chains");
    RUNV(4, 8, "v_pk_add_f32 8 chains");
    RUNV(6, 8, "v_pk_mul_f32 8 chains");
    RUNV(5, 4, "v_fmac_f32_dpp 4 chains");
    RUNV(5, 8, "v_fmac_f32_dpp 8 chains");
    for (int w : {4, 8, 16}) { if (run("ds_read_b128 depth 3", k_lds<3>, w, 3, 160 * 1024)) return 1; }
    for (int w : {4, 8, 16}) { if (run("ds_read_b128 depth 6", k_lds<6>, w, 6, 160 * 1024)) return 1; }
    for (int w : {4, 8, 16}) { if (run("ds_read_b128 depth 12", k_lds<12>, w, 12, 160 * 1024)) return 1; }
    for (int w : {4, 8, 16}) { if (run("ds_write_b128 x4", k_ldsw<4>, w, 4, 160 * 1024)) return 1; }
    for (int w : {4, 8, 16}) { if (run("ds_write_b128 x16", k_ldsw<16>, w, 16, 160 * 1024)) return 1; }
    return 0;
}
