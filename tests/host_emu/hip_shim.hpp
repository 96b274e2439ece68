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

// ---------------------------------------------------------------------------------------------------------------------
// Emulated threads.  Default: FIBERS -- every emulated GPU thread of a workgroup is a user-level context on ONE OS thread;
// a barrier wait is a stack switch to the next lane (tens of nanoseconds), not a trip through the kernel scheduler.  A
// wave-scope wait (shuffles, ballots, emulated MFMAs) cycles through the 64 lanes of its own wave only, a workgroup barrier
// through all threads.  With EMU_PTHREADS (and always under the sanitizers, which do not follow hand-made stack switches)
// every emulated thread is a pthread and a wait is a sched_yield loop -- the original scheme, 10-50x slower.
// ---------------------------------------------------------------------------------------------------------------------
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
#define EMU_PTHREADS 1
#endif
#if defined(__has_feature)
#if __has_feature(address_sanitizer) || __has_feature(thread_sanitizer)
#ifndef EMU_PTHREADS
#define EMU_PTHREADS 1
#endif
#endif
#endif
#if !defined(__x86_64__) && !defined(EMU_PTHREADS)
#define EMU_PTHREADS 1
#endif

#if defined(EMU_DEFER_GLDS) && !defined(EMU_PTHREADS)
#include <vector>
struct EmuPendingCopy { unsigned char* dst; unsigned char src[16]; int bytes; unsigned tid; };
inline std::vector<EmuPendingCopy>& emu_pending() { static thread_local std::vector<EmuPendingCopy> v; return v; }
inline void emu_glds(const void* g, void* l, int bytes) {
    EmuPendingCopy c;
    c.dst = static_cast<unsigned char*>(l); std::memcpy(c.src, g, bytes); c.bytes = bytes; c.tid = threadIdx.x;
    emu_pending().push_back(c);
}
inline void emu_complete_copies(bool all) {
    auto& v = emu_pending();
    size_t keep = 0;
    for (size_t k = 0; k < v.size(); ++k) {
        if (all || v[k].tid == threadIdx.x) std::memcpy(v[k].dst, v[k].src, v[k].bytes);
        else v[keep++] = v[k];
    }
    v.resize(keep);
}
#endif

#ifndef EMU_PTHREADS
#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
// callee-saved registers of the SysV x86-64 ABI on the stack, stack pointers exchanged
asm(".text\n"
    ".p2align 4\n"
    ".globl emu_ctx_switch\n"
    ".type emu_ctx_switch,@function\n"
    "emu_ctx_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size emu_ctx_switch,.-emu_ctx_switch\n");

struct EmuFiber { void* sp = nullptr; bool done = false; void* (*fn)(void*) = nullptr; void* arg = nullptr; };
struct EmuSched {
    std::vector<EmuFiber> f;
    int cur = 0, alive = 0;
    void* main_sp = nullptr;
};
inline EmuSched*& emu_sched() { static thread_local EmuSched* s = nullptr; return s; }

// hand the OS thread to fiber `nx` (the caller continues when somebody switches back to it)
inline void emu_switch_to(EmuSched* s, int nx) {
    const int me = s->cur;
    if (nx == me) return;
    const EmuDim3 t = threadIdx;            // threadIdx lives in thread-local storage shared by the fibers: every fiber carries its own
    s->cur = nx;
    emu_ctx_switch(&s->f[me].sp, s->f[nx].sp);
    threadIdx = t;
}
// next unfinished fiber after `me` inside [base, base + cnt); `me` itself if there is none
inline int emu_next_in(EmuSched* s, int me, int base, int cnt) {
    for (int k = 1; k <= cnt; ++k) {
        const int c = base + (me - base + k) % cnt;
        if (!s->f[c].done) return c;
    }
    return me;
}
inline void emu_yield_block() {
    EmuSched* s = emu_sched();
    emu_switch_to(s, emu_next_in(s, s->cur, 0, (int)s->f.size()));
}
inline void emu_yield_wave() {
    EmuSched* s = emu_sched();
    const int me = s->cur, base = me / 64 * 64, n = (int)s->f.size();
    const int nx = emu_next_in(s, me, base, n - base < 64 ? n - base : 64);
    if (nx != me) emu_switch_to(s, nx);
    else emu_yield_block();                 // (the rest of the wave has exited: let the other waves run)
}
extern "C" inline void emu_fiber_main() {
    EmuSched* s = emu_sched();
    {
        EmuFiber& me = s->f[s->cur];
        me.fn(me.arg);
        me.done = true;
        s->alive -= 1;
    }
    void* dead = nullptr;
    if (s->alive == 0) emu_ctx_switch(&dead, s->main_sp);
    const int nx = emu_next_in(s, s->cur, 0, (int)s->f.size());
    s->cur = nx;
    emu_ctx_switch(&dead, s->f[nx].sp);
    std::abort();                           // a finished fiber is never resumed
}
// run fn(args + t * stride) for t = 0 .. nthr - 1 as one workgroup; returns when every emulated thread has returned
inline void emu_run_threads(int nthr, void* (*fn)(void*), void* args, size_t stride, size_t stack_bytes = 1 << 20) {
    EmuSched sched;
    sched.f.resize(nthr);
    const size_t slot = (stack_bytes + 4095) / 4096 * 4096 + 4096;         // + a guard page below every stack
    unsigned char* mem = static_cast<unsigned char*>(mmap(nullptr, slot * nthr, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (mem == MAP_FAILED) { std::perror("emu_run_threads: mmap"); std::abort(); }
    for (int t = 0; t < nthr; ++t) {
        mprotect(mem + slot * t, 4096, PROT_NONE);
        uintptr_t top = reinterpret_cast<uintptr_t>(mem + slot * (t + 1)) & ~uintptr_t(15);
        void** sp = reinterpret_cast<void**>(top);
        *--sp = nullptr;                                        // (return address of emu_fiber_main: never used)
        *--sp = reinterpret_cast<void*>(&emu_fiber_main);       // popped by the first switch's `ret`: rsp = top - 8 at entry, as after a call
        for (int r = 0; r < 6; ++r) *--sp = nullptr;            // rbp rbx r12 r13 r14 r15
        sched.f[t].sp = sp;
        sched.f[t].fn = fn;
        sched.f[t].arg = static_cast<unsigned char*>(args) + stride * t;
    }
    sched.alive = nthr;
    sched.cur = 0;
    EmuSched* outer = emu_sched();
    emu_sched() = &sched;
    const EmuDim3 t0 = threadIdx;
    emu_ctx_switch(&sched.main_sp, sched.f[0].sp);
    threadIdx = t0;
    emu_sched() = outer;
#ifdef EMU_DEFER_GLDS
    emu_complete_copies(true);              // copies nobody waited for land at the end of the workgroup
#endif
    munmap(mem, slot * nthr);
}
#else
#include <vector>
inline void emu_yield_block() { sched_yield(); }
inline void emu_yield_wave() { sched_yield(); }
inline void emu_run_threads(int nthr, void* (*fn)(void*), void* args, size_t stride, size_t stack_bytes = 1 << 20) {
    std::vector<pthread_t> th(nthr);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, stack_bytes);
    for (int t = 0; t < nthr; ++t) pthread_create(&th[t], &attr, fn, static_cast<unsigned char*>(args) + stride * t);
    for (int t = 0; t < nthr; ++t) pthread_join(th[t], nullptr);
    pthread_attr_destroy(&attr);
}
#endif

// sense-reversing barrier; a waiter hands the processor on (fibers: to the next lane; pthreads: sched_yield -- with more
// emulated threads than cores a futex barrier costs ~100 us, yielding ~10 us)
struct EmuBarrier {
    std::atomic<int> count{0};
    std::atomic<int> generation{0};
    int parties = 1;
    bool wave_scope = false;
    void init(int n) { parties = n; count = 0; generation = 0; }
    void wait() {
        const int gen = generation.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == parties) {
            count.store(0, std::memory_order_relaxed);
            generation.store(gen + 1, std::memory_order_release);
        } else if (wave_scope) {
            while (generation.load(std::memory_order_acquire) == gen) emu_yield_wave();
        } else {
            while (generation.load(std::memory_order_acquire) == gen) emu_yield_block();
        }
    }
};

struct EmuBlock {
    EmuBarrier block_barrier;
    EmuBarrier wave_barrier[EMU_MAX_WAVES];
    uint64_t xbuf[EMU_MAX_WAVES][EMU_WAVE];
    EmuBlock() { for (auto& w : wave_barrier) w.wave_scope = true; }
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
// wave-to-wave LDS progress flags (kernel_macros.hpp): the waiter hands the processor on between polls (fibers: the flag's writer
// is another fiber of this OS thread -- a poll loop that did not yield would never see it change)
// (the lanes of an emulated wave do not run in lockstep: the flag may only be set once EVERY lane of the wave has issued the stores
// that precede it -- on the device that is the in-order execution of the wave's DS instructions)
#define BM_LDS_FLAG_SET(ptr, val) do { g_emu_block->wave_barrier[threadIdx.x / EMU_WAVE].wait(); *(volatile int*)(ptr) = (val); } while (0)
inline bool emu_lds_flag_wait(const volatile int* p, int v) {
    for (long k = 0; k < (1L << 24); ++k) {
        if (*p >= v) return true;
        emu_yield_block();
    }
    return false;
}
#ifdef EMU_NO_FLAG_WAIT            // negative control: the waits do nothing -- a kernel that needs them must come out wrong
#define BM_LDS_FLAG_WAIT(ptr, val) (true)
#else
#define BM_LDS_FLAG_WAIT(ptr, val) emu_lds_flag_wait((const volatile int*)(ptr), (val))
#endif
#define BM_FMA_F32(a, b, c, out) ((out) = __builtin_fmaf((a), (b), (c)))
#define BM_ADD_F32(a, b, out) ((out) = (a) + (b))
#define BM_SCHED_FENCE() ((void)0)
#define BM_SETPRIO(n) ((void)0)
#define BM_RESID_F16(hp, hi, v, out) do { unsigned short b_ = (unsigned short)((hp) >> (16 * (hi))); _Float16 h_; std::memcpy(&h_, &b_, 2); (out) = (v) - (float)h_; } while (0)
#define BM_RESID_PK_F16(hp, v0, v1, out) do { unsigned short b0_ = (unsigned short)(hp), b1_ = (unsigned short)((hp) >> 16); _Float16 h0_, h1_; \
    std::memcpy(&h0_, &b0_, 2); std::memcpy(&h1_, &b1_, 2); const _Float16 r0_ = (_Float16)((v0) - (float)h0_), r1_ = (_Float16)((v1) - (float)h1_); \
    unsigned short q0_, q1_; std::memcpy(&q0_, &r0_, 2); std::memcpy(&q1_, &r1_, 2); (out) = (unsigned)q0_ | ((unsigned)q1_ << 16); } while (0)
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
inline float emu_wave_sum_f32(float v) {       // the butterfly whose value the device's DPP sequence reproduces (kernel_macros.hpp)
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}
#define BM_WAVE_SUM_F32(v) emu_wave_sum_f32(v)
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
// ds_read_b64_tr_b16: the 16 lanes of a lane group name the sixteen 8-byte chunks (4 halves) of a [4 rows][16 columns] block in chunk order
// (lane i: row i / 4, columns 4 (i % 4) ..); lane c of the group receives column c: the halves (row 0..3, column c)
// (a template over the half type: harnesses built with a compiler that has no _Float16 include this header too and never instantiate it)
template <class H>
inline auto emu_ds_read_tr16_b64(const H* p) {
    typedef H h4_tr __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x % EMU_WAVE, wave = threadIdx.x / EMU_WAVE, g = lane >> 4, c = lane & 15;
    g_emu_block->xbuf[wave][lane] = (uint64_t)reinterpret_cast<uintptr_t>(p);
    g_emu_block->wave_barrier[wave].wait();
    h4_tr out;
    for (int j = 0; j < 4; ++j) {
        const H* src = reinterpret_cast<const H*>((uintptr_t)g_emu_block->xbuf[wave][16 * g + 4 * j + (c >> 2)]);
        out[j] = src[c & 3];
    }
    g_emu_block->wave_barrier[wave].wait();
    return out;
}
#define BM_DS_READ_TR16_B64(lds_ptr) emu_ds_read_tr16_b64(lds_ptr)

// Asynchronous global -> LDS copies (BM_GLDS16 / BM_GLDS4) and the wait that completes them (BM_WAIT_VM0).
// Default: the copy happens at once.  EMU_DEFER_GLDS (fiber mode): the copy is only QUEUED and lands when the issuing thread
// executes BM_WAIT_VM0 -- the latest moment the hardware allows -- so a kernel that publishes such data through a barrier without
// waiting first reads the launcher's LDS poison here too (a workgroup barrier does not complete anybody's copies).
// EMU_NO_VM_WAIT turns the wait into nothing: the negative control of that check.
#if defined(EMU_DEFER_GLDS) && !defined(EMU_PTHREADS)
#define BM_GLDS16(gptr, lds_wave_base, lane) emu_glds((gptr), reinterpret_cast<unsigned char*>(lds_wave_base) + 16 * (lane), 16)
#define BM_GLDS4(gptr, lds_wave_base, lane) emu_glds((gptr), reinterpret_cast<unsigned char*>(lds_wave_base) + 4 * (lane), 4)
#ifdef EMU_NO_VM_WAIT
#define BM_WAIT_VM0() ((void)0)
#else
#define BM_WAIT_VM0() emu_complete_copies(false)
#endif
#else
#define BM_GLDS16(gptr, lds_wave_base, lane) std::memcpy(reinterpret_cast<unsigned char*>(lds_wave_base) + 16 * (lane), (gptr), 16)
#define BM_GLDS4(gptr, lds_wave_base, lane) std::memcpy(reinterpret_cast<unsigned char*>(lds_wave_base) + 4 * (lane), (gptr), 4)
#define BM_WAIT_VM0() ((void)0)
#endif
