// TEST-ONLY: minimal host emulation of the HIP constructs used by the tracker
// kernels (boxmot_amd/csrc/botsort_step.hpp) so the *same device source* can be
// executed by g++ on CPU threads -- one pthread per GPU thread, pthread barriers
// for __syncthreads(), a per-wave exchange buffer for the 64-lane shuffles and
// ballots.  It exists to run the kernel logic under ASAN/UBSAN and to debug
// index/ordering bugs without spending GPU time.  It is NOT a product path:
// nothing in boxmot_amd/ includes this header, and the shipped library has no
// CPU implementation.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
extern thread_local EmuDim3 threadIdx;
extern thread_local EmuDim3 blockIdx;
extern EmuDim3 blockDim;
extern EmuDim3 gridDim;

#define BM_CLOCK() 0LL
#define BM_SLEEP_8K() ((void)0)
constexpr int EMU_WAVE = 64;
constexpr int EMU_MAX_WAVES = 16;

// sense-reversing barrier that yields instead of sleeping: with more emulated
// threads than cores a futex barrier costs ~100 us, yielding costs ~10 us
struct EmuBarrier {
    std::atomic<int> count{0};
    std::atomic<int> generation{0};
    int parties = 1;
    void init(int n) { parties = n; count = 0; generation = 0; }
    void wait() {
        const int gen = generation.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == parties) {
            count.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
        } else {
            while (generation.load(std::memory_order_acquire) == gen) sched_yield();
        }
    }
};

struct EmuBlock {
    EmuBarrier block_barrier;
    EmuBarrier wave_barrier[EMU_MAX_WAVES];
    uint64_t xbuf[EMU_MAX_WAVES][EMU_WAVE];
};
extern EmuBlock* g_emu_block;

inline void __syncthreads() { g_emu_block->block_barrier.wait(); }

template <class T>
inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    const int lane = threadIdx.x % EMU_WAVE, wave = threadIdx.x / EMU_WAVE;
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    g_emu_block->xbuf[wave][lane] = raw;
    g_emu_block->wave_barrier[wave].wait();
    raw = g_emu_block->xbuf[wave][src_lane & (EMU_WAVE - 1)];
    g_emu_block->wave_barrier[wave].wait();
    T out;
    std::memcpy(&out, &raw, sizeof(T));
    return out;
}

template <class T> inline T __shfl_xor(T v, int mask, int = EMU_WAVE) { return emu_exchange(v, (int)(threadIdx.x % EMU_WAVE) ^ mask); }
template <class T> inline T __shfl(T v, int src, int = EMU_WAVE) { return emu_exchange(v, src); }

inline unsigned long long __ballot(int pred) {
    const int wave = threadIdx.x / EMU_WAVE, lane = threadIdx.x % EMU_WAVE;
    g_emu_block->xbuf[wave][lane] = pred ? 1 : 0;
    g_emu_block->wave_barrier[wave].wait();
    unsigned long long m = 0;
    for (int l = 0; l < EMU_WAVE; ++l) m |= (unsigned long long)(g_emu_block->xbuf[wave][l] & 1) << l;
    g_emu_block->wave_barrier[wave].wait();
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---- additions for the fused ReID kernels (compiled with a host clang that knows _Float16) ----
#define __restrict__
#define BM_DYNAMIC_LDS_T(type, name) type* name = reinterpret_cast<type*>(g_emu_dynamic_lds)
#define __shared__ static
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dmul_rn(double a, double b) { return a * b; }
#define BM_EXPF(x) expf(x)
#define BM_SCHED_FENCE() ((void)0)
#define BM_SETPRIO(n) ((void)0)
#define BM_WAIT_VM0() ((void)0)
#define BM_RESID_F16(hp, hi, v, out) do { unsigned short b_ = (unsigned short)((hp) >> (16 * (hi))); _Float16 h_; std::memcpy(&h_, &b_, 2); (out) = (v) - (float)h_; } while (0)
#define BM_RCPF(x) (1.0f / (x))
#define BM_OPAQUE_U32(x) ((void)0)
inline float emu_row_shift(float v, int d, bool rotate) {
    const int lane = threadIdx.x % EMU_WAVE, src16 = (lane & 15) + d;
    const float r = emu_exchange(v, (lane & 48) | (src16 & 15));
    return (rotate || (src16 >= 0 && src16 < 16)) ? r : 0.f;
}
// DPP row modes used by the kernels: 0x10n row_shl:n (lane+n), 0x11n row_shr:n (lane-n), 0x12n row_ror:n
inline unsigned emu_dpp_u32(unsigned old, unsigned v, int ctrl, bool zero_fill) {
    const int lane = threadIdx.x % EMU_WAVE, l16 = lane & 15, n = ctrl & 15, mode = ctrl & 0x1f0;
    const int src16 = mode == 0x100 ? l16 + n : (mode == 0x110 ? l16 - n : ((l16 - n) & 15));
    const unsigned r = emu_exchange(v, (lane & 48) | (src16 & 15));
    return (src16 >= 0 && src16 < 16) ? r : (zero_fill ? 0u : old);
}
#define BM_DPP_U32(old, v, ctrl, zero_fill) emu_dpp_u32(old, v, ctrl, zero_fill)
#define BM_READLANE_U32(v, l) ((unsigned)__shfl((int)(v), l, 64))
#define BM_QUAD_SWAP1_F32(v) __shfl_xor((float)(v), 1, 64)
#define BM_UNIFORM_I32(x) ((int)(x))
#define BM_MUL24(a, b) ((unsigned)(((unsigned)(a) & 0xffffffu) * ((unsigned)(b) & 0xffffffu)))
#define BM_MULHI24(a, b) ((unsigned)(((unsigned long long)((unsigned)(a) & 0xffffffu) * ((unsigned)(b) & 0xffffffu)) >> 32))
#define BM_ROW_SHL1_F32(v) emu_row_shift(v, 1, false)
#define BM_ROW_SHR1_F32(v) emu_row_shift(v, -1, false)
#define BM_ROW_ROR1_F32(v) emu_row_shift(v, -1, true)
extern unsigned char* g_emu_dynamic_lds;
inline float sqrtf_emu(float x) { return std::sqrt(x); }

// MFMA emulation: every lane deposits its A/B fragment, then computes its 4 outputs of
// D = A.B + C with the CDNA layouts (A: row = lane&15, k = H*(lane>>4)+j; B: col = lane&15,
// same k; D: col = lane&15, rows 4*(lane>>4)+r).  H = 4 (16x16x16) or 8 (16x16x32).
struct EmuMfmaBuf { float a[EMU_MAX_WAVES][EMU_WAVE][8]; float b[EMU_MAX_WAVES][EMU_WAVE][8]; };
extern EmuMfmaBuf* g_emu_mfma;

template <int H, class HA, class F>
inline F emu_mfma(HA a, HA b, F c) {
    const int wave = threadIdx.x / EMU_WAVE, lane = threadIdx.x % EMU_WAVE;
    for (int j = 0; j < H; ++j) { g_emu_mfma->a[wave][lane][j] = (float)a[j]; g_emu_mfma->b[wave][lane][j] = (float)b[j]; }
    g_emu_block->wave_barrier[wave].wait();
    const int col = lane & 15, g = lane >> 4;
    F d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4 * H; ++k) {
            const int kg = k / H, kj = k % H;
            acc += g_emu_mfma->a[wave][row + 16 * kg][kj] * g_emu_mfma->b[wave][col + 16 * kg][kj];
        }
        d[r] = acc;
    }
    g_emu_block->wave_barrier[wave].wait();
    return d;
}
#define BM_MFMA_F16_K16(a, b, c) emu_mfma<4>(a, b, c)
#define BM_MFMA_F16_K32(a, b, c) emu_mfma<8>(a, b, c)

// fp32 16x16x4 MFMA: the k-ordered fmaf chain the hardware computes (one rounding per product-accumulate)
#if defined(__clang__)
typedef float emu_f4 __attribute__((ext_vector_type(4)));
#else
typedef float emu_f4 __attribute__((vector_size(16)));
#endif
inline emu_f4 emu_mfma_f32_k4(float a, float b, emu_f4 c) {
    const int wave = threadIdx.x / EMU_WAVE, lane = threadIdx.x % EMU_WAVE;
    g_emu_mfma->a[wave][lane][0] = a; g_emu_mfma->b[wave][lane][0] = b;
    g_emu_block->wave_barrier[wave].wait();
    const int col = lane & 15, g = lane >> 4;
    emu_f4 d = c;
    for (int r = 0; r < 4; ++r) {
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(g_emu_mfma->a[wave][4 * g + r + 16 * k][0], g_emu_mfma->b[wave][col + 16 * k][0], acc);
        d[r] = acc;
    }
    g_emu_block->wave_barrier[wave].wait();
    return d;
}
#define BM_MFMA_F32_K4(a, b, c) emu_mfma_f32_k4(a, b, c)
#define BM_WAVE_LDS_SYNC() g_emu_block->wave_barrier[threadIdx.x / EMU_WAVE].wait()
#define BM_GLDS16(gptr, lds_wave_base, lane) std::memcpy(reinterpret_cast<unsigned char*>(lds_wave_base) + 16 * (lane), (gptr), 16)
