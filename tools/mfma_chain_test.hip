// Known-answer test of dependent MFMA chains on gfx950 (development tool).  Finding (ROCm 7.2, MI355X): a chain on ONE accumulator
// that mixes v_mfma_f32_16x16x16_f16 and v_mfma_f32_16x16x32_f16 (modes 4, 9, 10: three or more links) returns wrong sums with the
// compiler-generated code; uniform chains of any length (modes 2, 3, 5-8) and two-link mixed chains (modes 0, 1) are exact.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I boxmot_amd/csrc tools/mfma_chain_test.hip -o tools/_build/mfma_chain_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "kernel_macros.hpp"
#include "reid_fused.hpp"
using bm::h4; using bm::h8; using bm::f4;
// mode 0: K32 then K16 on the same accumulator; 1: K16 then K32; 2: K32,K32; 3: K16,K16; 4: K32,K16,K32,K16,K32,K16 (the layer loop's chain)
template <int MODE>
__global__ void k(const _Float16* a8p, const _Float16* b8p, const _Float16* a4p, const _Float16* b4p, const float* cp, float* out) {
    const int lane = threadIdx.x;
    const h8 a8 = *reinterpret_cast<const h8*>(a8p + lane * 8), b8 = *reinterpret_cast<const h8*>(b8p + lane * 8);
    const h4 a4 = *reinterpret_cast<const h4*>(a4p + lane * 4), b4 = *reinterpret_cast<const h4*>(b4p + lane * 4);
    f4 acc = *reinterpret_cast<const f4*>(cp + lane * 4);
    if (MODE == 0) { acc = BM_MFMA_F16_K32(a8, b8, acc); acc = BM_MFMA_F16_K16(a4, b4, acc); }
    if (MODE == 1) { acc = BM_MFMA_F16_K16(a4, b4, acc); acc = BM_MFMA_F16_K32(a8, b8, acc); }
    if (MODE == 2) { acc = BM_MFMA_F16_K32(a8, b8, acc); acc = BM_MFMA_F16_K32(a8, b8, acc); }
    if (MODE == 3) { acc = BM_MFMA_F16_K16(a4, b4, acc); acc = BM_MFMA_F16_K16(a4, b4, acc); }
    if (MODE == 4) for (int q = 0; q < 3; ++q) { acc = BM_MFMA_F16_K32(a8, b8, acc); acc = BM_MFMA_F16_K16(a4, b4, acc); }
    if (MODE == 5) for (int q = 0; q < 6; ++q) acc = BM_MFMA_F16_K32(a8, b8, acc);
    if (MODE == 6) for (int q = 0; q < 32; ++q) acc = BM_MFMA_F16_K16(a4, b4, acc);
    if (MODE == 7) for (int q = 0; q < 9; ++q) acc = BM_MFMA_F16_K16(a4, b4, acc);
    if (MODE == 8) { f4 acc2 = acc; for (int q = 0; q < 6; ++q) { acc = BM_MFMA_F16_K32(a8, b8, acc); acc2 = BM_MFMA_F16_K32(b8, a8, acc2); } acc = acc + acc2 * 0.f; }
    if (MODE == 9) { acc = BM_MFMA_F16_K32(a8, b8, acc); acc = BM_MFMA_F16_K16(a4, b4, acc); acc = BM_MFMA_F16_K32(a8, b8, acc); }
    if (MODE == 10) { acc = BM_MFMA_F16_K16(a4, b4, acc); acc = BM_MFMA_F16_K32(a8, b8, acc); acc = BM_MFMA_F16_K16(a4, b4, acc); }
    // modes 11-13: the failing chain of mode 4 with explicit wait states between the links (2, 8, 16): if one of them is
    // exact, the mixed-shape dense 3x3 of DESIGN.md section 8 becomes usable with hand-placed s_nops
#define BM_GAP(n) asm volatile("s_nop " #n : "+v"(acc))      /* tied to the accumulator so it stays between the two MFMAs */
    if (MODE == 11) for (int q = 0; q < 3; ++q) { acc = BM_MFMA_F16_K32(a8, b8, acc); BM_GAP(1); acc = BM_MFMA_F16_K16(a4, b4, acc); BM_GAP(1); }
    if (MODE == 12) for (int q = 0; q < 3; ++q) { acc = BM_MFMA_F16_K32(a8, b8, acc); BM_GAP(7); acc = BM_MFMA_F16_K16(a4, b4, acc); BM_GAP(7); }
    if (MODE == 13) for (int q = 0; q < 3; ++q) { acc = BM_MFMA_F16_K32(a8, b8, acc); BM_GAP(15); acc = BM_MFMA_F16_K16(a4, b4, acc); BM_GAP(15); }
    *reinterpret_cast<f4*>(out + lane * 4) = acc;
}
static float A8[16][32], B8[32][16], A4[16][16], B4[16][16], C[16][16];
int main() {
    _Float16 ha8[512], hb8[512], ha4[256], hb4[256]; float hc[256], ho[256];
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 9) & 0xff) / 64.0f - 2.0f; };   // multiples of 1/64: exact sums
    for (int lane = 0; lane < 64; ++lane) {
        const int rc = lane & 15, g = lane >> 4;
        for (int j = 0; j < 8; ++j) { A8[rc][8 * g + j] = rnd(); ha8[lane * 8 + j] = (_Float16)A8[rc][8 * g + j]; B8[8 * g + j][rc] = rnd(); hb8[lane * 8 + j] = (_Float16)B8[8 * g + j][rc]; }
        for (int j = 0; j < 4; ++j) { A4[rc][4 * g + j] = rnd(); ha4[lane * 4 + j] = (_Float16)A4[rc][4 * g + j]; B4[4 * g + j][rc] = rnd(); hb4[lane * 4 + j] = (_Float16)B4[4 * g + j][rc]; }
        for (int r = 0; r < 4; ++r) { C[4 * g + r][rc] = rnd(); hc[lane * 4 + r] = C[4 * g + r][rc]; }
    }
    _Float16 *da8, *db8, *da4, *db4; float *dc, *dout;
    (void)hipMalloc(&da8, 1024); (void)hipMalloc(&db8, 1024); (void)hipMalloc(&da4, 512); (void)hipMalloc(&db4, 512); (void)hipMalloc(&dc, 1024); (void)hipMalloc(&dout, 1024);
    (void)hipMemcpy(da8, ha8, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(db8, hb8, 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(da4, ha4, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db4, hb4, 512, hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc, 1024, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 14; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 11) hipLaunchKernelGGL(k<11>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 12) hipLaunchKernelGGL(k<12>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 13) hipLaunchKernelGGL(k<13>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, da8, db8, da4, db4, dc, dout);
        (void)hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        static const int N32[14] = {1, 1, 2, 0, 3, 6, 0, 0, 6, 2, 1, 3, 3, 3}, N16[14] = {1, 1, 0, 2, 3, 0, 32, 9, 0, 1, 2, 3, 3, 3};
        const int n32 = N32[mode], n16 = N16[mode];
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;
            double p32 = 0, p16 = 0;
            for (int kk = 0; kk < 32; ++kk) p32 += (double)A8[row][kk] * B8[kk][col];
            for (int kk = 0; kk < 16; ++kk) p16 += (double)A4[row][kk] * B4[kk][col];
            const double want = C[row][col] + n32 * p32 + n16 * p16;
            if (std::fabs(ho[lane * 4 + r] - want) > 2e-3 * (1 + std::fabs(want))) { if (bad < 4) printf("  mode %d lane %d r %d got %f want %f\n", mode, lane, r, ho[lane * 4 + r], want); ++bad; }
        }
        printf("mode %d: %d mismatches\n", mode, bad);
    }
    return 0;
}
