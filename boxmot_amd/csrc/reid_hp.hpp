// fp32-grade fused ReID kernels for OSNet-x0.25 (gfx950, wave64): ReID "mode 2".
//
// Reference computation: OSNet.forward, boxmot/reid/backbones/osnet.py:380-405 (OSBlock :212-260, LightConv3x3 :127-155,
// ChannelGate :161-209) in fp32 on the CPU (reid/backends/base_backend.py:197-217) -- north_star's comparator, tolerance 1e-3.
// The fp16-operand family (reid_fused.hpp, mode 1) meets that tolerance on the reference's own initialisation only; on
// networks with non-trivial BatchNorm statistics fp16 activations alone cost 5e-3 (profiles/r2_reid_error_budget.txt).  This
// family keeps the SAME structure -- one workgroup owns one crop for a whole OSBlock, every 1x1 convolution is an MFMA whose
// accumulator layout is the next B operand, the depthwise 3x3 goes through an LDS image, gates meet in LDS -- with fp32-grade
// arithmetic everywhere:
//   * 1x1 convolutions: both operands as fp16 (hi, lo) pairs, three v_mfma_f32_16x16x32_f16 per K = 32 product tile, two for a
//     K = 16 layer (reid_hp_pack.hpp); fp32 accumulation; weights split on the host, activations split in registers;
//   * activations between MFMAs, the depthwise 3x3, the gates and the shortcut sums: fp32 registers / fp32 LDS image;
//   * activations between kernels: two fp16 planes (hi, lo) in the lane-group-major NHWC layout (the B fragments of the
//     consumer are plain 16-byte loads; same bytes as fp32).
// LDS image: channel-group planes [ct][g][row][x] of 16-byte pixels (4 channels fp32), plane stride a multiple of 256 bytes:
// the 16 lanes of a lane group touch 16 consecutive 16-byte slots and the four lane groups of a ds_read_b128 / ds_write_b128
// service group land on disjoint slots -- conflict-free for the centre and the x +- 1 taps alike.
// The depthwise 3x3 streams down a column strip: every input row is read once (3 reads: x - 1, x, x + 1) and feeds the three
// output rows it touches (three live accumulators instead of a 3 x 3 window of registers).
#pragma once

#include "reid_fused.hpp"
#include "reid_hp_pack.hpp"

// stage-2 blocks: 4 = two workgroups per CU at <= 128 registers (spills ~35), 2 = one workgroup per CU at <= 256 (A/B switch)
#ifndef BM_HP_S2_WPS
#define BM_HP_S2_WPS 4
#endif
// waves per workgroup in stages 0 / 1 (A/B switch): 8 = two waves per SIMD at <= 256 registers, 16 = four at <= 128
#ifndef BM_HP_NW0
#define BM_HP_NW0 8
#endif
#ifndef BM_HP_NW1
#define BM_HP_NW1 8
#endif
#ifndef BM_HP_STAGGER
#define BM_HP_STAGGER 0
#endif
// epilogue: tiles per group (the A fragments of an output-channel tile are read from LDS once per GROUP and held in registers
// while the group's tiles run through them: 1 / TG of the fragment reads, TG independent MFMA chains)
#ifndef BM_HP_EPI_TG0E
#define BM_HP_EPI_TG0E 2            // stage 0, first block (EMIT)
#endif
#ifndef BM_HP_EPI_TG0R
#define BM_HP_EPI_TG0R 2            // stage 0, second block (RECON + fused transition): 4 would spill
#endif
#ifndef BM_HP_EPI_TG1
#define BM_HP_EPI_TG1 2             // stage 1 (2 and 4 measured equal; 2 keeps 12 registers free)
#endif
// weights into LDS by asynchronous global -> LDS copies (no register round trip, no latency per loop trip) issued ahead of the
// phase that needs them; 0 = load / store loops at the point of use (A/B switch)
// fused transition: pooled output stored as 16-byte channel-tile pairs (1) or 8-byte tiles (0; A/B switch)
#ifndef BM_HP_TRANS_ST16
#define BM_HP_TRANS_ST16 1
#endif
// ChannelGate's crop-wide sums: wave reduction on the DPP path (1) or through the LDS crossbar (__shfl_xor; 0)
#ifndef BM_HP_GATE_DPP
#define BM_HP_GATE_DPP 1
#endif
#ifndef BM_HP_ASYNC_STAGE
#define BM_HP_ASYNC_STAGE 1
#endif
// streaming accesses (hand-over tensors and block outputs written once, operands read once) marked non-temporal so that the
// tensor a kernel re-reads per branch (128 KiB per crop, one L2 share) is not evicted by them
// LightConv layer loop of stages 0 / 1: neighbour synchronisation (1) instead of two workgroup barriers per layer (0).  A wave's
// depthwise pass reads its own rows and ONE row of each neighbouring wave, so wave w only has to know that waves w - 1 and w + 1 have
// written the layer (flag `wr`) and, before it overwrites its first / last row, that they have read the previous one (flag `rd`):
// 20 barriers per crop fewer, and waves may run up to a layer apart (the two waves of a SIMD are four waves apart: their LDS-heavy
// depthwise phases and their MFMA / memory phases stop coinciding).  The gates stay workgroup barriers (crop-wide sums).
#ifndef BM_HP_NBR_SYNC
#define BM_HP_NBR_SYNC 0
#endif
// persistent workgroups (BlkLinkHP::n_crops): compiled in only on request -- measured flat (profiles/r5_hp_persist_ab.txt)
#ifndef BM_HP_PERSIST
#define BM_HP_PERSIST 0
#endif
#ifndef BM_HP_S0_RECOMP
#define BM_HP_S0_RECOMP 1
#endif
#ifndef BM_HP_DW_STATIC_MORE
#define BM_HP_DW_STATIC_MORE 0
#endif
// depthwise 3x3: reads of the next input row issued right after the last use of the current one (1) or at the top of its own
// iteration (0: the compiler schedules them a few instructions before their first use); A/B switch, profiles/r5_hp_s0_ab.txt
#ifndef BM_HP_DW_PREFETCH
#define BM_HP_DW_PREFETCH 0
#endif
// Depthwise 3x3 of stages 0 / 1 ON REGISTERS (1) instead of through an LDS image (0; A/B switch, profiles/r6_hp_dwreg_ab.txt).
// A wave owns whole image rows (8 rows x 32 pixels in stage 0, 4 x 16 in stage 1), so the vertical neighbours of a pixel are other
// registers of the SAME lane and the horizontal ones are the adjacent lanes of its 16-lane row: x +- 1 come from DPP row shifts
// (v_mov_b32_dpp row_shr:1 / row_shl:1, zero fill = the zero padding at the image edge; in stage 0 the seam between the two 16-pixel
// halves of a row is closed with a row_ror of the other half's register as the `old` operand).  Only the rows at the edge of a wave's
// strip go through LDS -- a double-buffered halo exchange of 4 KiB per wave and layer, ONE workgroup barrier per layer -- instead of
// the whole 128 / 48 KiB tensor written and read three times with two barriers: the un-instrumented ablations (BM_HP_ABL,
// profiles/r6_hp_ablation.txt) price the image writes at 28 % and the image reads at 30 % of a stage-0 block kernel.  Same taps in the
// same order on the same fp32 values: bit-identical embeddings (tests/test_reid_emu.py).
#ifndef BM_HP_DW_REG
#define BM_HP_DW_REG 0
#endif
#ifndef BM_HP_IMGW_PIPE
#define BM_HP_IMGW_PIPE 1
#endif
// Image writes of stages 0 / 1 issued UNDER the next layer's 1x1 (1): tile i - LAG goes to the image right after tile i's (hi, lo) split and
// MFMAs, so the CU's one LDS store path (13.5 cycles per ds_write_b128, 16 per wave and layer) works while the vector unit converts --
// instead of a burst of all eight waves with nothing to run under.  The layer's second barrier moves from behind the 1x1 to in front of it
// (every wave is past its depthwise reads before the first tile is overwritten); still two barriers per layer.  Needs BM_HP_PW_BATCH; takes
// the place of BM_HP_IMGW_PIPE.  A/B: profiles/r6_hp_variants_ab.txt
#ifndef BM_HP_IMGW_PW
#define BM_HP_IMGW_PW 1
#endif
#ifndef BM_HP_IMGW_LAG
#define BM_HP_IMGW_LAG 1            // 1: 1.063 / 1.140 / 0.824 / 0.758 ms, 2: 1.079 / 1.154 / 0.824 / 0.777, 4: 1.083 / 1.173 / 0.842 / 0.798 (base 1.125 / 1.190 / 0.856 / 0.792)
#endif
#ifndef BM_HP_PW_BATCH
#define BM_HP_PW_BATCH 1            // measured -2 % / -5 % on the two stage-0 block kernels (profiles/r6_hp_variants_ab.txt)
#endif
#ifndef BM_HP_DW_PIPE
#define BM_HP_DW_PIPE 0             // LDS reads of the depthwise pass issued this many input rows ahead of their taps (0: at their use)
#endif
#ifndef BM_HP_DWREG_FENCE
#define BM_HP_DWREG_FENCE 1
#endif
#ifndef BM_HP_NT
#define BM_HP_NT 0
#endif
#ifndef BM_HP_ABL
#define BM_HP_ABL 0         // timing-only ablation mask (tools/hp_prof), described below
#endif
#if (BM_HP_ABL & 64)
#define BM_NT_STORE(ptr, val) hp_keep(val)
#define BM_NT_LOAD(ptr) (*(ptr))
#elif BM_HP_NT
#define BM_NT_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#define BM_NT_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define BM_NT_STORE(ptr, val) (*(ptr) = (val))
#define BM_NT_LOAD(ptr) (*(ptr))
#endif

// TIMING-ONLY ablations (tools/hp_prof built with -DBM_HP_ABL=<mask>; never set in the library): each bit compiles ONE phase's work out
// of k_osblock_hp while everything else stays, so the un-instrumented cost of that phase in situ is time(base) - time(ablated) -- the
// phase clocks of BM_OSBLOCK_PROF serialise the wave at every stamp (2.3 x slower) and say where a wave's dependent work is, not what
// the overlapped kernel pays for it.  The results of an ablated build are garbage.  profiles/r6_hp_ablation.txt
//   1 depthwise taps (LDS reads kept)        2 depthwise LDS reads (taps on registers)   4 LightConv 1x1 (split + MFMAs)
//   8 image writes                          16 branch input (conv1 recompute / x1 reload)  32 epilogue MFMAs + output split
//  64 epilogue global stores               128 epilogue operand loads                    256 ChannelGate arithmetic
// 512 the two barriers of a LightConv layer 1024 conv1 (loads + MFMAs; not the per-branch recompute)  2048 epilogue fragment reads from LDS
#ifndef BM_HP_ABL
#define BM_HP_ABL 0
#endif

#include <type_traits>

namespace bm {

constexpr bool hp_abl(int bit) { return (BM_HP_ABL & bit) != 0; }
#ifndef BM_OPAQUE_F4
#define BM_OPAQUE_F4(v) asm volatile("" : "+v"((v)[0]), "+v"((v)[1]), "+v"((v)[2]), "+v"((v)[3]))
#endif
__device__ inline void hp_keep(f4 v) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
__device__ inline void hp_keep(h8 v) { const f4 f = __builtin_bit_cast(f4, v); hp_keep(f); }
__device__ inline void hp_keep(h4 v) { typedef float f2v __attribute__((ext_vector_type(2))); const f2v f = __builtin_bit_cast(f2v, v); asm volatile("" ::"v"(f[0]), "v"(f[1])); }

template <int STAGE>
struct GeoHP {
    static constexpr int H = 64 >> STAGE, W = 32 >> STAGE, P = H * W;
    static constexpr int MID = STAGE == 0 ? 16 : (STAGE == 1 ? 24 : 32);
    static constexpr int KT = STAGE == 0 ? 1 : 2;
    static constexpr int MIDP = 16 * KT;
    static constexpr int HID = MID / 16;
    static constexpr int COUT = STAGE == 0 ? 64 : (STAGE == 1 ? 96 : 128);
    static constexpr int NCT = COUT / 16;
    static constexpr int NWAVES = STAGE == 0 ? BM_HP_NW0 : (STAGE == 1 ? BM_HP_NW1 : 8);
    static constexpr int NT = P / 16 / NWAVES;                       // 16, 4, 1 pixel tiles per wave (8 waves)
    // stages 0 / 1: one workgroup per CU (the fp32 image is 141 / 78 KiB): two waves per SIMD, 256 registers each;
    // stage 2: two workgroups per CU
    static constexpr int WAVES_PER_SIMD = STAGE == 2 ? BM_HP_S2_WPS : NWAVES / 4;
    // stage 2 (8-pixel rows): compact pitch (10 pixels): 2-way conflicts between the two rows of a tile, but two workgroups per CU
    static constexpr int ROWP = STAGE == 0 ? 34 * 16 : (STAGE == 1 ? 18 * 16 : 160);
    static constexpr int PLANE = STAGE == 0 ? 141 * 256 : (STAGE == 1 ? 39 * 256 : 12 * 256);
    // BM_HP_DW_REG: no image -- the halo exchange [2 buffers][NWAVES][top row, bottom row][KT][NH] tiles of 1 KiB (lane-linear f4)
    static constexpr bool DWREG = BM_HP_DW_REG && STAGE < 2;
    static constexpr int NH = STAGE == 0 ? 2 : 1;                     // 16-pixel tiles per image row
    static constexpr int HALO = 2 * NWAVES * 2 * KT * NH * 1024;
    static constexpr int IMG = DWREG ? HALO : 4 * KT * PLANE;
    // LightConv weights of the whole block, staged once into LDS behind the image: no global load (and no exposed L2 / fabric
    // round trip) inside the 10-layer loop.  Stages 1 / 2: everything (1x1 fragment pairs, depthwise taps, biases = the packed
    // blob's light records); stage 0 has 19 KiB left beside its 141 KiB image: depthwise taps + biases only, the 1x1 fragments
    // are requested a layer ahead into registers.
    static constexpr bool W_ALL = STAGE != 0;
    static constexpr int LIGHT_BYTES = KT * 2048 + MIDP * 9 * 4 + MIDP * 4;         // = BlkPackHP::light_bytes
    static constexpr int WREC = W_ALL ? LIGHT_BYTES : MIDP * 9 * 4 + MIDP * 4;      // bytes staged per LightConv
    static constexpr int GATE_BYTES = ((HID * MIDP * 4 + 15) / 16 + (HID * 4 + 15) / 16 + (MIDP * HID * 4 + 15) / 16 + (MIDP * 4 + 15) / 16) * 16;   // fc1_w, fc1_b, fc2_w, fc2_b
    static constexpr int WBYTES = 10 * WREC + GATE_BYTES;          // + the ChannelGate's weights (no global load in the gate either)
    // the epilogue stages its A fragments over the (then dead) image and weights; stage 2's downsample block needs 66048 bytes
    static constexpr int TBUF = IMG + WBYTES;
    static constexpr int FLAGS = TBUF + 4 * NWAVES * HID * 4;       // [2][NWAVES] int: layer written / layer read (BM_HP_NBR_SYNC)
    static constexpr int LDS_BYTES = FLAGS + (BM_HP_NBR_SYNC ? 2 * NWAVES * 4 : 0);
    static_assert(PLANE >= (H + 2) * ROWP && PLANE % 256 == 0, "plane holds the haloed image; stride keeps the lane groups on disjoint slots");
    static_assert(!DWREG || (NWAVES == 8 && NT % NH == 0 && !BM_HP_NBR_SYNC), "register-resident depthwise: a wave owns whole rows");
    static_assert(LDS_BYTES <= 163840 / (STAGE == 2 ? 2 : 1), "LDS budget");
};

// fp32 multiply-adds / adds of four channels: packed (v_pk_fma_f32 / v_pk_add_f32: half the instructions) or four scalar
// instructions (BM_HP_SCALAR_F32 = 1: packed f32 VALU operations are priced well above two scalar ones beside MFMAs on this part,
// kernel_macros.hpp BM_FMA_F32; identical results; A/B switch, profiles/r5_hp_scalar_f32_ab.txt)
// (CAUTION, found in round 6: the scalar forms are inline assembly, and the compiler's hazard recogniser does not protect an inline-asm
// vector instruction that reads an MFMA result -- the epilogue's shortcut sum under this switch returned run-to-run different checksums
// on the device (profiles/r6_hp_variants_ab.txt).  Timing switch only; never the library default.)
#ifndef BM_HP_SCALAR_F32
#define BM_HP_SCALAR_F32 0
#endif
__device__ inline f4 fma_f4(f4 a, f4 b, f4 c) {
#if BM_HP_SCALAR_F32
    f4 r;
    BM_FMA_F32(a[0], b[0], c[0], r[0]); BM_FMA_F32(a[1], b[1], c[1], r[1]); BM_FMA_F32(a[2], b[2], c[2], r[2]); BM_FMA_F32(a[3], b[3], c[3], r[3]);
    return r;
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ inline f4 add_f4(f4 a, f4 b) {
#if BM_HP_SCALAR_F32
    f4 r;
    BM_ADD_F32(a[0], b[0], r[0]); BM_ADD_F32(a[1], b[1], r[1]); BM_ADD_F32(a[2], b[2], r[2]); BM_ADD_F32(a[3], b[3], r[3]);
    return r;
#else
    return a + b;
#endif
}
// (hi, lo) fp16 parts of four fp32 values: v = hi + lo up to 2^-22 relative
#ifndef BM_HP_SPLIT_MIX
#define BM_HP_SPLIT_MIX 1           // residuals by v_fma_mix_f32 (8 instructions per split4 instead of 12-13); 0: convert + subtract
#endif
__device__ inline void split4(f4 v, h4& h, h4& l) {
    h = to_h4(v);
#if BM_HP_SPLIT_MIX == 2
    // residual and its conversion in ONE instruction per value (v_fma_mixlo_f16 / v_fma_mixhi_f16): 6 instructions per split4
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v hp = __builtin_bit_cast(u2v, h);
    u2v lp;
    BM_RESID_PK_F16(hp[0], v[0], v[1], lp[0]);
    BM_RESID_PK_F16(hp[1], v[2], v[3], lp[1]);
    l = __builtin_bit_cast(h4, lp);
#elif BM_HP_SPLIT_MIX
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const u2v hp = __builtin_bit_cast(u2v, h);
    f4 r;
    BM_RESID_F16(hp[0], 0, v[0], r[0]); BM_RESID_F16(hp[0], 1, v[1], r[1]);
    BM_RESID_F16(hp[1], 0, v[2], r[2]); BM_RESID_F16(hp[1], 1, v[3], r[3]);
    l = to_h4(r);
#else
    l = to_h4(f4{v[0] - (float)h[0], v[1] - (float)h[1], v[2] - (float)h[2], v[3] - (float)h[3]});
#endif
}
// K = 32 product tile on (hi, lo) operands: the fragment pair at `a` (hi at +0, lo at +1024; 16 bytes per lane)
__device__ inline f4 mm3(const unsigned char* a, int lane, h8 bh, h8 bl, f4 acc) {
    const h8 ah = *reinterpret_cast<const h8*>(a + lane * 16), al = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
    acc = BM_MFMA_F16_K32(ah, bh, acc);
    acc = BM_MFMA_F16_K32(ah, bl, acc);
    acc = BM_MFMA_F16_K32(al, bh, acc);
    return acc;
}
// K = 16 layer in the duplicated form: b = [xh | xl] against [Wh | Wh] and [Wl | Wl]
__device__ inline f4 mm2(const unsigned char* a, int lane, h8 b, f4 acc) {
    const h8 ah = *reinterpret_cast<const h8*>(a + lane * 16), al = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
    acc = BM_MFMA_F16_K32(ah, b, acc);
    acc = BM_MFMA_F16_K32(al, b, acc);
    return acc;
}

__device__ inline f4 mm3r(h8 ah, h8 al, h8 bh, h8 bl, f4 acc) {
    acc = BM_MFMA_F16_K32(ah, bh, acc);
    acc = BM_MFMA_F16_K32(ah, bl, acc);
    acc = BM_MFMA_F16_K32(al, bh, acc);
    return acc;
}
__device__ inline f4 mm2r(h8 ah, h8 al, h8 b, f4 acc) {
    acc = BM_MFMA_F16_K32(ah, b, acc);
    acc = BM_MFMA_F16_K32(al, b, acc);
    return acc;
}

struct BlkLinkHP {
    const unsigned char* w = nullptr;
    long a0 = 0, a1 = 0, a2 = 0;
    float* x2s = nullptr;
    // BM_HP_PERSIST builds only; > 0: PERSISTENT launch -- the grid is a fixed number of workgroups (one or two per CU) and workgroup b runs crops b, b + gridDim.x,
    // ... < n_crops in a loop, instead of one workgroup launch per crop (64 launches per CU and kernel at the headline's 16384 crops:
    // dispatch, wave start-up, LDS allocation and the argument loads once per workgroup instead of once per crop).  0: one crop per
    // workgroup.  A/B: profiles/r5_hp_persist_ab.txt
    int n_crops = 0;
};

// ---------------------------------------------------------------------------
// OSBlock: in (hi, lo) [n][P][CIN] -> out (hi, lo) [n][P or P/4][COUT]
// TRANS / EMIT / RECON as in k_osblock (reid_fused.hpp): fused transition; first block of a pair hands the second its
// conv1 result (`x1s`) and its gated branch sum (`link.x2s`), both fp32, instead of its 64-channel output.
// ---------------------------------------------------------------------------
template <int STAGE, int CIN, bool DOWN, bool TRANS, bool EMIT = false, bool RECON = false>
__global__ void __launch_bounds__(64 * GeoHP<STAGE>::NWAVES, GeoHP<STAGE>::WAVES_PER_SIMD)
k_osblock_hp(const _Float16* __restrict__ in_h, const _Float16* __restrict__ in_l, _Float16* __restrict__ out_h, _Float16* __restrict__ out_l,
             const unsigned char* __restrict__ wts, BlkPackHP bp, const int* __restrict__ count, float* __restrict__ x1s,
             const unsigned char* __restrict__ wtr, BlkLinkHP link) {
    static_assert(!EMIT || (STAGE <= 1 && CIN == (STAGE == 0 ? 16 : 64) && DOWN && !TRANS), "EMIT: first block of stage 0 or 1");
    static_assert(!RECON || (STAGE <= 1 && CIN == GeoHP<STAGE>::COUT && !DOWN && TRANS), "RECON: second block of stage 0 or 1");
    using G = GeoHP<STAGE>;
#if BM_HP_PERSIST        // (compile-time: the loop's three live scalars cost the second stage-0 block 3 spilled registers in the library build)
    const long crop_end = link.n_crops > 0 ? (long)link.n_crops : (long)blockIdx.x + 1;
    const long crop_step = link.n_crops > 0 ? (long)gridDim.x : 1;
#pragma unroll 1
    for (long crop = blockIdx.x; crop < crop_end; crop += crop_step) {
    if (count && crop >= *count) break;
#else
    if (count && (int)blockIdx.x >= *count) return;
    const long crop = blockIdx.x;
    {
#endif
    // Phase stagger: every workgroup of a launch does the same phases for the same time, so without it all 256 CUs read
    // (branch inputs, epilogue operands) and write at the same moments and the fabric is idle in between.  The workgroups of the
    // FIRST wave of the grid start a quarter period apart in four groups (neighbouring CUs of an XCD in different groups); the
    // offset persists for the whole launch because a CU takes its next crop when it finishes the last.
    if (BM_HP_STAGGER && blockIdx.x < 256 * (STAGE == 2 ? 2 : 1)) {
        const int grp = (blockIdx.x >> 3) & 3;
        constexpr int QUARTER_8K = STAGE == 0 ? 6 : (STAGE == 1 ? 4 : 2);       // ~quarter of the per-crop time in units of 8128 cycles
        for (int k = 0; k < grp * QUARTER_8K; ++k) BM_SLEEP_8K();
    }
    constexpr int KT = G::KT, NT = G::NT, MIDP = G::MIDP, NCT = G::NCT, COUT = G::COUT, P = G::P;
    constexpr int KIN = CIN == 16 ? 1 : CIN / 32;
    constexpr int PREV_CIN = STAGE == 0 ? 16 : 64, KINP = PREV_CIN == 16 ? 1 : PREV_CIN / 32;
    // first block of stage 0: conv1 recomputed per branch from the (hi, lo) input planes (two 8-byte loads + two MFMAs per tile), or --
    // BM_HP_S0_RECOMP = 0 -- computed once, parked in the fp32 hand-over scratch (which this block only overwrites in its epilogue, when
    // the branches are done) and read back per branch with ONE 16-byte load per tile: half the vector-memory instructions of a branch
    // start (phase clocks, profiles/r5_hp_phases.txt: the branch-input phase is 33 % of this kernel and 2.7 x the second block's, which
    // moves the same bytes with 16-byte loads).  A/B switch.
    constexpr bool RECOMP = STAGE == 0 && CIN == 16 && BM_HP_S0_RECOMP;
    constexpr bool X1_MEM = (STAGE == 0 && !RECOMP) || RECON;     // conv1 result in the fp32 scratch `x1s` (L2), read per branch
    constexpr bool X1_REG = !RECOMP && !X1_MEM;
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    unsigned char* tbuf = lds;
    unsigned char* wl = lds + G::IMG;           // staged LightConv weights, record l at wl + l * WREC
    float* gap_part = reinterpret_cast<float*>(lds + G::TBUF);      // [4 branches][NWAVES][HID]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const _Float16* xh = in_h + crop * P * (RECON ? PREV_CIN : CIN);       // RECON: the previous block's input
    const _Float16* xl = in_l + crop * P * (RECON ? PREV_CIN : CIN);
    const long out_px = TRANS ? P / 4 : P;

    BM_PROF_DECL();

    // this lane's pixel of tile 0 inside plane (ct = 0, g); tile i and channel tile ct add compile-time constants
    const int y_l = STAGE == 0 ? wave * (NT / 2) : (STAGE == 1 ? wave * NT : 2 * wave + (l16 >> 3));
    const int x_l = STAGE == 2 ? (l16 & 7) : l16;
    const int pix0 = g * G::PLANE + (y_l + 1) * G::ROWP + (x_l + 1) * 16;
    auto tile_off = [](int i) constexpr { return STAGE == 0 ? (i >> 1) * G::ROWP + (i & 1) * 256 : (STAGE == 1 ? i * G::ROWP : 0); };
    float* x1w = nullptr;       // this lane's slots in the scratch tensors: [(tile, ct)][lane] f4 = 1 KiB per wave access
    if constexpr (X1_MEM || EMIT) x1w = x1s + crop * (P * MIDP) + (long)(wave * NT * KT * 64 + lane) * 4;
    float* x2w = nullptr;
    if constexpr (EMIT || RECON) x2w = link.x2s + crop * (P * MIDP) + (long)(wave * NT * KT * 64 + lane) * 4;

    // workgroup copy global -> LDS of `bytes` (a multiple of 16; src and dst 16-byte aligned).  Asynchronous form: one
    // global_load_lds per KiB and wave, no registers, completion at the next __syncthreads(); `gather` maps a destination offset
    // to its source offset (identity for contiguous records).
    auto stage_copy = [&](const unsigned char* src, unsigned char* dst, int bytes, auto gather) {
        if constexpr (BM_HP_ASYNC_STAGE) {
            for (int c = wave * 1024; c < bytes; c += G::NWAVES * 1024)
                if (c + lane * 16 < bytes) BM_GLDS16(src + gather(c + lane * 16), dst + c, lane);
        } else {
            for (int e = tid * 16; e < bytes; e += 64 * G::NWAVES * 16)
                *reinterpret_cast<f4*>(dst + e) = *reinterpret_cast<const f4*>(src + gather(e));
        }
    };
    auto same = [](int e) { return (long)e; };
    // LightConv weights (stage 0: depthwise taps + biases only) and the gate's weights -> `wl`
    auto stage_light = [&]() {
        if constexpr (G::W_ALL) {           // light records and gate weights are contiguous in the packed blob
            stage_copy(wts + bp.light0, wl, G::WBYTES, same);
        } else {
            const long lb = bp.light_bytes, ld = bp.light_dw;
            stage_copy(wts + bp.light0, wl, 10 * G::WREC, [&](int e) { const int l = e / G::WREC; return l * lb + ld + (e - l * G::WREC); });
            stage_copy(wts + bp.fc1_w, wl + 10 * G::WREC, G::GATE_BYTES, same);
        }
    };
    if constexpr (BM_HP_ASYNC_STAGE) stage_light();          // in flight under conv1 / the image clear

    // ---- conv1: 1x1 CIN -> MID, + bias, ReLU (osnet.py:248) ----
    auto conv1_into = [&](f4 (&x1)[NT][KT]) {
        if constexpr (hp_abl(1024) && !(STAGE == 0 && CIN == 16)) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) x1[i][ct] = f4{0.25f, 0.5f, 0.125f, 1.f};
            return;
        }
        unsigned xo = 0;
        if constexpr (RECOMP) BM_OPAQUE_U32(xo);       // re-read per branch, do not hoist 16 tiles of input out of the branch loop
        f4 bias[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) bias[ct] = *reinterpret_cast<const f4*>(wts + bp.conv1_b + (16 * ct + 4 * g) * 4);
        if constexpr (CIN == 16) {
            const h8 ah = *reinterpret_cast<const h8*>(wts + bp.conv1_a + lane * 16);
            const h8 al = *reinterpret_cast<const h8*>(wts + bp.conv1_a + 1024 + lane * 16);
            h8 b[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const unsigned o = xo + (unsigned)(((wave * NT + i) * 16 + l16) * 16 + g * 4);
                b[i] = cat8(*reinterpret_cast<const h4*>(xh + o), *reinterpret_cast<const h4*>(xl + o));
            }
            BM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                f4 acc = BM_MFMA_F16_K32(ah, b[i], bias[0]);
                acc = BM_MFMA_F16_K32(al, b[i], acc);
                x1[i][0] = relu4(acc);
            }
        } else {
            // fragment pairs [ks][ct] staged into LDS (the image area is not in use yet); the B operands of tile i + 1 are
            // requested before tile i computes
            stage_copy(wts + bp.conv1_a, tbuf, KIN * KT * (int)HP_FRAG_PAIR, same);
            h8 bh[2][KIN], bl[2][KIN];
            auto loads = [&](int i, h8 (&h)[KIN], h8 (&l)[KIN]) {
                const unsigned p = (wave * NT + i) * 16 + l16;
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks) {
                    const unsigned o = p * CIN + g * (CIN / 4) + 8 * ks;
                    h[ks] = *reinterpret_cast<const h8*>(xh + o); l[ks] = *reinterpret_cast<const h8*>(xl + o);
                }
            };
            loads(0, bh[0], bl[0]);
            if constexpr (BM_HP_ASYNC_STAGE) BM_WAIT_VM0();         // the fragment copies (and the LightConv weights issued before them)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (i + 1 < NT) loads(i + 1, bh[(i + 1) & 1], bl[(i + 1) & 1]);
                f4 acc[KT];
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) acc[ct] = bias[ct];
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks)
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) acc[ct] = mm3(tbuf + (long)(ks * KT + ct) * HP_FRAG_PAIR, lane, bh[i & 1][ks], bl[i & 1][ks], acc[ct]);
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) x1[i][ct] = relu4(acc[ct]);
            }
            __syncthreads();            // every wave is done with the staged fragments before the image is zeroed over them
        }
    };
    f4 x1[X1_REG ? NT : 1][KT];
    if constexpr (X1_REG) conv1_into(x1);
    if constexpr (X1_MEM && !RECON) {       // conv1 once, parked in the scratch
        f4 t[NT][KT];
        conv1_into(t);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(x1w + (i * KT + ct) * 256) = t[i][ct];
    }

    // 1x1 (linear, mid -> mid) of a LightConv on the fp32 tile set: weight fragments held in registers for the layer, operands
    // split in registers
    // the next layer's 1x1 of a tile right after the tile's depthwise row (1: its MFMAs run "under" the next rows' taps) or for all tiles after
    // the pass (0).  BM_HP_PW_BATCH = 1 un-fuses the mid-width-16 stage as well: a v_pk_fma_f32 issued beside an MFMA costs more than both
    // apart (tools/coissue_bench.hip, profiles/r6_coissue_microbench.txt: 8 MFMAs + 32 packed FMAs take 743 cycles together, 594 apart), and the
    // fused form alternates 18 packed taps and 2 MFMAs per tile.
    constexpr bool PW_FUSED = KT == 1 && !BM_HP_PW_BATCH;
    constexpr bool PW32 = BM_HP_PW32 && STAGE == 0;       // LightConv 1x1 on the fp32 matrix pipe (reid_hp_pack.hpp)
    auto load_pw = [&](int l, h8 (&A)[KT][2]) {         // 1x1 fragment pairs of LightConv l (0..9)
        const unsigned char* src = G::W_ALL ? wl + l * G::WREC : wts + bp.light0 + (long)l * bp.light_bytes + bp.light_pw;
        if constexpr (PW32) {                           // one fp32 fragment (16 bytes per lane); kept in the first half of A[0][0]'s slot
            const f4 a = *reinterpret_cast<const f4*>(src + lane * 16);
            A[0][0] = __builtin_bit_cast(h8, a);
            return;
        }
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) {
            A[ct][0] = *reinterpret_cast<const h8*>(src + (long)ct * HP_FRAG_PAIR + lane * 16);
            A[ct][1] = *reinterpret_cast<const h8*>(src + (long)ct * HP_FRAG_PAIR + 1024 + lane * 16);
        }
    };
    auto pointwise = [&](const h8 (&A)[KT][2], f4 (&c)[KT]) {
        if constexpr (hp_abl(4)) return;
        if constexpr (PW32) {
            const f4 a = __builtin_bit_cast(f4, A[0][0]);
            const f4 x = c[0];
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = BM_MFMA_F32_K4(a[kk], x[kk], acc);
            c[0] = acc;
            return;
        }
        h4 hh[KT], ll[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) split4(c[ct], hh[ct], ll[ct]);
        const f4 z = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (KT == 1) {
            c[0] = mm2r(A[0][0], A[0][1], cat8(hh[0], ll[0]), z);
        } else {
            const h8 bh = cat8(hh[0], hh[KT - 1]), bl = cat8(ll[0], ll[KT - 1]);
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) c[ct] = mm3r(A[ct][0], A[ct][1], bh, bl, z);
        }
    };

    // the 1x1 of a layer on every tile of the wave; BM_HP_IMGW_PW: each tile's result goes to the image BM_HP_IMGW_LAG tiles later, under the
    // following tiles' splits (the caller has passed the barrier that ends the previous layer's reads of the image)
    constexpr bool IMGW_PW = BM_HP_IMGW_PW && BM_HP_PW_BATCH && STAGE < 2 && !G::DWREG && !(BM_HP_NBR_SYNC);
    auto img_write = [&](int i, const f4 (&c)[KT]) {
        if constexpr (!hp_abl(8)) {
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(i)) = c[ct];
        }
    };
    auto pw_all = [&](const h8 (&A)[KT][2], f4 (&c)[NT][KT]) {
        if constexpr (IMGW_PW) {
            constexpr int LAG = BM_HP_IMGW_LAG < NT ? BM_HP_IMGW_LAG : NT - 1;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                pointwise(A, c[i]);
                if (i >= LAG) img_write(i - LAG, c[i - LAG]);
                BM_SCHED_FENCE();
            }
#pragma unroll
            for (int i = NT - LAG; i < NT; ++i) img_write(i, c[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) pointwise(A, c[i]);
        }
    };

    if constexpr (!G::DWREG)
    for (int e = tid * 16; e < G::IMG; e += 64 * G::NWAVES * 16) *reinterpret_cast<f4*>(tbuf + e) = f4{0.f, 0.f, 0.f, 0.f};   // halo = zero padding
    constexpr bool NBR = BM_HP_NBR_SYNC && STAGE < 2;
    volatile int* wr_flag = reinterpret_cast<volatile int*>(lds + G::FLAGS);        // wr_flag[w]: last layer (1-based) wave w has written
    volatile int* rd_flag = wr_flag + G::NWAVES;                                     // rd_flag[w]: last layer wave w has finished reading
    if constexpr (NBR) { if (tid < 2 * G::NWAVES) wr_flag[tid] = 0; }
    if constexpr (!BM_HP_ASYNC_STAGE) stage_light();
    const unsigned char* wgate = wl + 10 * G::WREC;                 // gate weights: fc1_w at +0, then fc1_b, fc2_w, fc2_b as in the blob
    const int g_fc1b = (int)(bp.fc1_b - bp.fc1_w), g_fc2w = (int)(bp.fc2_w - bp.fc1_w), g_fc2b = (int)(bp.fc2_b - bp.fc1_w);
    f4 x2[NT][KT];          // gated sum of the four branches
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) x2[i][ct] = f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (BM_HP_ASYNC_STAGE) BM_WAIT_VM0();                 // this wave's share of the staged weights has landed
    __syncthreads();
    BM_PROF(0);

    // The A fragments of the epilogue are staged ONCE per workgroup into LDS (the image is dead after the last layer): eight waves
    // re-reading 20-60 KB of fragments per tile through the vector L1 (64 B/clk) was a large share of the block time; LDS
    // delivers them at 256 B/clk (lane-linear 1 KiB reads, conflict-free).
    //   [0, E_OWN)  this block's conv3 pairs, bias, downsample pairs (contiguous in the packed blob)
    //   then: EMIT the next block's conv1 pairs | RECON the previous block's conv3 pairs .. downsample pairs | TRANS pairs + bias
    // Stages 0 / 1: the operands fit the image area alone, so the (asynchronous) copies are issued as soon as the last depthwise
    // pass has read the image -- BEFORE the last branch's gate, whose weights live behind the image -- and land under the gate.
    constexpr int KS3 = COUT / 32, KSN = COUT / 32;
    constexpr int E_OWN_C = NCT * (int)HP_FRAG_PAIR + COUT * 4 + (DOWN ? NCT * KIN * (int)HP_FRAG_PAIR : 0);
    constexpr int E_PREV_C = RECON ? NCT * (int)HP_FRAG_PAIR + COUT * 4 + NCT * KINP * (int)HP_FRAG_PAIR : 0;
    constexpr int E_LINK = EMIT ? KSN * KT * (int)HP_FRAG_PAIR : 0;
    constexpr int E_TR = TRANS ? NCT * KS3 * (int)HP_FRAG_PAIR + COUT * 4 : 0;
    static_assert(E_OWN_C + E_LINK + E_PREV_C + E_TR <= G::TBUF, "epilogue operands fit the (dead) image and weight area");
    constexpr bool EPI_EARLY = BM_HP_ASYNC_STAGE && STAGE < 2 && E_OWN_C + E_LINK + E_PREV_C + E_TR <= G::IMG;
    const int e_own = (int)(bp.total - bp.conv3_a);         // (host side: prepare_hp checks e_own + the other regions <= TBUF)
    const int e_prev = RECON ? (int)(link.a2 - link.a0) + NCT * KINP * (int)HP_FRAG_PAIR : 0;
    auto stage_epilogue = [&]() {
        stage_copy(wts + bp.conv3_a, tbuf, e_own, same);
        if constexpr (EMIT) stage_copy(link.w + link.a0, tbuf + e_own, E_LINK, same);
        if constexpr (RECON) stage_copy(link.w + link.a0, tbuf + e_own, e_prev, same);
        if constexpr (TRANS) stage_copy(wtr, tbuf + e_own + e_prev, E_TR, same);
    };

    int li = 0;
#pragma unroll 1
    for (int br = 0; br < 4; ++br) {
        f4 cur[NT][KT];
        if constexpr (hp_abl(16)) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) cur[i][ct] = x2[i][ct] + f4{0.5f, 0.25f, 0.125f, 1.f};
        } else if constexpr (X1_MEM) {
            unsigned xo = 0;
            BM_OPAQUE_U32(xo);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) cur[i][ct] = *reinterpret_cast<const f4*>(x1w + (xo + (unsigned)((i * KT + ct) * 256)));
        } else if constexpr (RECOMP) conv1_into(cur);
        else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) cur[i][ct] = x1[i][ct];
        }
        // `cur` holds, in turn: the branch input, the 1x1 output of the layer about to run its depthwise pass, the layer output
        h8 An[KT][2];           // fragments of the NEXT layer's 1x1 (stage 0: requested a layer ahead of their use)
        {
            h8 A0[KT][2];
            load_pw(li, A0);
            if (br > 0) load_pw(li + 1, An);
            pw_all(A0, cur);
        }
        BM_PROF(1);
#pragma unroll 1
        for (int k = 0; k <= br; ++k, ++li) {
            const bool more = k < br;
            // depthwise taps and bias of this layer: from the staged weights (LDS)
            const unsigned char* wdl = wl + li * G::WREC + (G::W_ALL ? KT * 2048 : 0);
#ifndef BM_HP_DWREG_NWD1
#define BM_HP_DWREG_NWD1 1
#endif
            constexpr int NWD = (STAGE == 2 || (G::DWREG && STAGE == 1 && BM_HP_DWREG_NWD1)) ? 1 : KT;          // depthwise tap sets in flight (stage 2 has 128 registers)
            f4 wdv[NWD][9], dbias[NWD];
            auto load_dw = [&](int ct, f4 (&wd)[9], f4& bias) {
                const f4* wsrc = reinterpret_cast<const f4*>(wdl) + (ct * 4 + g) * 9;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wd[tap] = wsrc[tap];
                bias = *reinterpret_cast<const f4*>(wdl + MIDP * 9 * 4 + (16 * ct + 4 * g) * 4);
            };
#pragma unroll
            for (int c = 0; c < NWD; ++c) load_dw(c, wdv[c], dbias[c]);
            constexpr int NH = G::NH, ROWS_W = NT / NH;
            // Image writes spread over the depthwise pass (1) instead of one burst of all eight waves before the barrier (0): a ds_write_b128
            // costs 13.5 cycles on the CU's one store path but runs under OTHER instructions' issue (tools/coissue_bench.hip: 4 writes + 32 vector
            // instructions per wave take what the longer of the two takes alone) -- in a burst there is nothing to run under.
            constexpr bool IMGW_PIPE = BM_HP_IMGW_PIPE && !IMGW_PW && STAGE < 2 && !G::DWREG && !(BM_HP_NBR_SYNC) && ROWS_W > 2;
            auto IMGW_ROW = [](int i) constexpr { return i / NH; };        // strip row of tile i (stage 0: tiles 2 r, 2 r + 1; stage 1: tile r)
            // halo exchange slot of (buffer, wave, edge: 0 = the strip's first row, 1 = its last, channel tile, row half)
            auto halo_at = [&](int buf, int w, int edge, int ct, int h) {
                return tbuf + (long)((((buf * G::NWAVES + w) * 2 + edge) * KT + ct) * NH + h) * 1024 + lane * 16;
            };
            if constexpr (G::DWREG) {
                const int hb = li & 1;
                if constexpr (!hp_abl(8)) {
#pragma unroll
                for (int ct = 0; ct < KT; ++ct)
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        *reinterpret_cast<f4*>(halo_at(hb, wave, 0, ct, h)) = cur[h][ct];
                        *reinterpret_cast<f4*>(halo_at(hb, wave, 1, ct, h)) = cur[(ROWS_W - 1) * NH + h][ct];
                    }
                }
                BM_PROF(2);
                if constexpr (!hp_abl(512)) __syncthreads();      // the only barrier of the layer: the buffers alternate, so the next layer's edge rows
                BM_PROF(3);                                       // are written while a slow neighbour may still be reading this layer's
            } else if constexpr (NBR) {
                // rows of this wave's strip: tiles (2 r, 2 r + 1) in stage 0, tile r in stage 1; the first row is read by wave - 1, the last
                // by wave + 1 -- those two are overwritten only after that neighbour has finished the previous layer's reads
                constexpr int ROWS_W = STAGE == 0 ? NT / 2 : NT;
                auto row_of = [](int i) constexpr { return STAGE == 0 ? i >> 1 : i; };
                const int seq = li + 1;
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (row_of(i) != 0 && row_of(i) != ROWS_W - 1) {
#pragma unroll
                        for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(i)) = cur[i][ct];
                    }
                if (wave > 0) (void)BM_LDS_FLAG_WAIT(rd_flag + wave - 1, seq - 1);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (row_of(i) == 0) {
#pragma unroll
                        for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(i)) = cur[i][ct];
                    }
                if (wave < G::NWAVES - 1) (void)BM_LDS_FLAG_WAIT(rd_flag + wave + 1, seq - 1);
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (row_of(i) == ROWS_W - 1 && ROWS_W > 1) {
#pragma unroll
                        for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(i)) = cur[i][ct];
                    }
                BM_LDS_FLAG_SET(wr_flag + wave, seq);
                BM_PROF(2);
                if (wave > 0) (void)BM_LDS_FLAG_WAIT(wr_flag + wave - 1, seq);
                if (wave < G::NWAVES - 1) (void)BM_LDS_FLAG_WAIT(wr_flag + wave + 1, seq);
                BM_PROF(3);
            } else {
                if constexpr (!hp_abl(8) && !IMGW_PW) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    // BM_HP_IMGW_PIPE: only the rows a NEIGHBOURING wave reads (the first and the last of the strip) are written ahead of
                    // the barrier; a wave's inner rows are read by nobody else, and its own LDS operations execute in order, so they are
                    // written inside the depthwise pass, one row ahead of their first read (IMGW_ROW below)
                    if constexpr (IMGW_PIPE) { if (IMGW_ROW(i) != 0 && IMGW_ROW(i) != ROWS_W - 1) continue; }
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(i)) = cur[i][ct];
                }
                }
                BM_PROF(2);
                if constexpr (!hp_abl(512)) __syncthreads();
                BM_PROF(3);
            }
            // (the pass is a generic lambda so that "another layer follows" can be a compile-time fact: BM_HP_DW_STATIC_MORE = 1 instantiates
            // it twice -- with the runtime flag every output row ends in a branch around the fused 1x1, i.e. in a basic-block boundary the
            // instruction scheduler cannot move work across; without the branch a whole strip is one block and row r's split / MFMAs can
            // be interleaved with row r + 1's taps.  A/B switch, profiles/r5_hp_static_more_ab.txt)
            auto dw_pass = [&](auto more_c) {
                bool do_pw;
                if constexpr (std::is_same_v<decltype(more_c), bool>) do_pw = more_c; else do_pw = decltype(more_c)::value;
            // depthwise 3x3 (pad 1) + bias + ReLU
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                if constexpr (NWD < KT) { if (ct > 0) load_dw(ct, wdv[0], dbias[0]); }
                const f4 (&wd)[9] = wdv[NWD < KT ? 0 : ct];
                const f4 bias = dbias[NWD < KT ? 0 : ct];
                const unsigned char* cbase = tbuf + pix0 + ct * 4 * G::PLANE;
                if constexpr (G::DWREG) {
                    const int hb = li & 1;
                    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
                    f4 acc[NH][3];
#pragma unroll
                    for (int rr = 0; rr < ROWS_W + 2; ++rr) {          // input row (first row of the strip) - 1 + rr
                        f4 C[NH], Lv[NH], Rv[NH];
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            if (rr == 0) C[h] = (wave > 0 && !hp_abl(2)) ? *reinterpret_cast<const f4*>(halo_at(hb, wave - 1, 1, ct, h)) : zero4;
                            else if (rr == ROWS_W + 1) C[h] = (wave < G::NWAVES - 1 && !hp_abl(2)) ? *reinterpret_cast<const f4*>(halo_at(hb, wave + 1, 0, ct, h)) : zero4;
                            else C[h] = cur[(rr - 1) * NH + h][ct];
                        }
#if BM_HP_DWREG_FENCE
                        // (the row's values made opaque here: the shifts below are pure moves the instruction selector would otherwise
                        // hoist to the top of the unrolled pass -- every row's x +- 1 copies live at once)
#pragma unroll
                        for (int h = 0; h < NH; ++h) BM_OPAQUE_F4(C[h]);
#endif
                        // x - 1 / x + 1 of every pixel of the row: DPP shifts inside the 16-lane rows, zero at the image edge
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            auto u = [](float v) { return __builtin_bit_cast(unsigned, v); };
                            auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
                            if constexpr (NH == 1) {
                                Lv[0][r] = f(BM_DPP_U32(0u, u(C[0][r]), 0x111, true));          // row_shr:1 -> lane - 1
                                Rv[0][r] = f(BM_DPP_U32(0u, u(C[0][r]), 0x101, true));          // row_shl:1 -> lane + 1
                            } else {
                                Lv[0][r] = f(BM_DPP_U32(0u, u(C[0][r]), 0x111, true));
                                const unsigned seam_l = BM_DPP_U32(0u, u(C[0][r]), 0x121, true);       // row_ror:1: lane 0 <- the left half's lane 15
                                Lv[1][r] = f(BM_DPP_U32(seam_l, u(C[1][r]), 0x111, false));          // lanes 1..15 <- lane - 1, lane 0 keeps the seam value
                                const unsigned seam_r = BM_DPP_U32(0u, u(C[1][r]), 0x12F, true);       // row_ror:15: lane 15 <- the right half's lane 0
                                Rv[0][r] = f(BM_DPP_U32(seam_r, u(C[0][r]), 0x101, false));
                                Rv[1][r] = f(BM_DPP_U32(0u, u(C[1][r]), 0x101, true));
                            }
                        }
#pragma unroll
                        for (int h = 0; h < NH; ++h) {
                            const f4 v0 = Lv[h], v1 = C[h], v2 = Rv[h];
                            if constexpr (hp_abl(1)) {
                                hp_keep(v0); hp_keep(v2);
                                if (rr >= 2) { const int i = (rr - 2) * NH + h; cur[i][ct] = relu4(v1); if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); } }
                                continue;
                            }
                            if (rr >= 2) {                              // completes output row rr - 2 (its input role ended a step ago: in place)
                                f4 a = acc[h][(rr - 2) % 3];
                                a = fma_f4(wd[6], v0, a); a = fma_f4(wd[7], v1, a); a = fma_f4(wd[8], v2, a);
                                const int i = (rr - 2) * NH + h;
                                cur[i][ct] = relu4(a);
                                if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); }
                            }
                            if (rr >= 1 && rr <= ROWS_W) {
                                f4 a = acc[h][(rr - 1) % 3];
                                a = fma_f4(wd[3], v0, a); a = fma_f4(wd[4], v1, a); a = fma_f4(wd[5], v2, a);
                                acc[h][(rr - 1) % 3] = a;
                            }
                            if (rr <= ROWS_W - 1) {
                                f4 a = fma_f4(wd[0], v0, bias);
                                a = fma_f4(wd[1], v1, a); a = fma_f4(wd[2], v2, a);
                                acc[h][rr % 3] = a;
                            }
                        }
#if BM_HP_DWREG_FENCE
                        BM_SCHED_FENCE();       // one input row's shifts and taps at a time (the scheduler would hoist every row's DPP moves)
#endif
                    }
                } else if constexpr (STAGE == 2) {
                    f4 o = bias;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        o = fma_f4(wd[tap], *reinterpret_cast<const f4*>(cbase + (tap / 3 - 1) * G::ROWP + (tap % 3 - 1) * 16), o);
                    cur[0][ct] = relu4(o);
                } else {
                    constexpr int NSEQ = STAGE == 0 ? 2 : 1, L = NT / NSEQ;
#if BM_HP_DW_PREFETCH
                    // The same taps in the same order, but the three reads of input row rr + 1 are issued as soon as row rr's values have
                    // had their last use -- BEFORE the ReLU / (hi, lo) split / 1x1 MFMAs of the output row that row rr completed -- so
                    // the LDS round trip runs under ~20 instructions of independent work instead of being waited for at once.
#pragma unroll
                    for (int sq = 0; sq < NSEQ; ++sq) {
                        f4 acc[3];
                        const unsigned char* rp0 = cbase + sq * 256 - G::ROWP;
                        f4 v0 = *reinterpret_cast<const f4*>(rp0 - 16), v1 = *reinterpret_cast<const f4*>(rp0), v2 = *reinterpret_cast<const f4*>(rp0 + 16);
#pragma unroll
                        for (int rr = 0; rr < L + 2; ++rr) {
                            f4 done = f4{0.f, 0.f, 0.f, 0.f};
                            if (rr >= 2) {
                                f4 a = acc[(rr - 2) % 3];
                                a = fma_f4(wd[6], v0, a); a = fma_f4(wd[7], v1, a); a = fma_f4(wd[8], v2, a);
                                done = a;
                            }
                            if (rr >= 1 && rr <= L) {
                                f4 a = acc[(rr - 1) % 3];
                                a = fma_f4(wd[3], v0, a); a = fma_f4(wd[4], v1, a); a = fma_f4(wd[5], v2, a);
                                acc[(rr - 1) % 3] = a;
                            }
                            if (rr <= L - 1) {
                                f4 a = fma_f4(wd[0], v0, bias);
                                a = fma_f4(wd[1], v1, a); a = fma_f4(wd[2], v2, a);
                                acc[rr % 3] = a;
                            }
                            if (rr + 1 < L + 2) {
                                const unsigned char* rp = cbase + sq * 256 + rr * G::ROWP;
                                BM_SCHED_FENCE();
                                v0 = *reinterpret_cast<const f4*>(rp - 16); v1 = *reinterpret_cast<const f4*>(rp); v2 = *reinterpret_cast<const f4*>(rp + 16);
                                BM_SCHED_FENCE();
                            }
                            if (rr >= 2) {
                                const int i = STAGE == 0 ? 2 * (rr - 2) + sq : rr - 2;
                                cur[i][ct] = relu4(done);
                                if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); }
                            }
                        }
                    }
#elif BM_HP_DW_PIPE > 0
                    // The same taps in the same order with the LDS reads SOFTWARE-PIPELINED: the three reads of the input row BM_HP_DW_PIPE
                    // steps ahead are issued (into their own registers: a ring of BM_HP_DW_PIPE + 1 row sets) BEFORE this step's taps, so a
                    // wave computes on one row while its next rows are in flight -- with three reads per wait a ds_read_b128 costs a wave
                    // ~59 cycles at 8 waves per CU, with twelve in flight ~33 (tools/valu_chain_bench.hip, profiles/r6_valu_lds_microbench.txt);
                    // the un-pipelined pass waits for every row before it issues the row's 25 vector instructions.  The steps of both
                    // column halves form ONE sequence, so the pipeline does not drain between them.
                    {
                        constexpr int D = BM_HP_DW_PIPE, NS = NSEQ * (L + 2);
                        f4 ring[D + 1][3];
                        auto issue = [&](int s, f4 (&v)[3]) {
                            const int sq = s / (L + 2), rr = s % (L + 2);
                            const unsigned char* rp = cbase + sq * 256 + (rr - 1) * G::ROWP;
                            v[0] = *reinterpret_cast<const f4*>(rp - 16); v[1] = *reinterpret_cast<const f4*>(rp); v[2] = *reinterpret_cast<const f4*>(rp + 16);
                        };
#pragma unroll
                        for (int s = 0; s < D && s < NS; ++s) issue(s, ring[s % (D + 1)]);
                        f4 acc[3];
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            const int sq = s / (L + 2), rr = s % (L + 2);
                            if (s + D < NS) issue(s + D, ring[(s + D) % (D + 1)]);
                            BM_SCHED_FENCE();           // the requests above stay above this step's arithmetic
                            const f4 v0 = ring[s % (D + 1)][0], v1 = ring[s % (D + 1)][1], v2 = ring[s % (D + 1)][2];
                            if (rr >= 2) {                              // completes output row rr - 2
                                f4 a = acc[(rr - 2) % 3];
                                a = fma_f4(wd[6], v0, a); a = fma_f4(wd[7], v1, a); a = fma_f4(wd[8], v2, a);
                                const int i = STAGE == 0 ? 2 * (rr - 2) + sq : rr - 2;
                                cur[i][ct] = relu4(a);
                                if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); }
                            }
                            if (rr >= 1 && rr <= L) {
                                f4 a = acc[(rr - 1) % 3];
                                a = fma_f4(wd[3], v0, a); a = fma_f4(wd[4], v1, a); a = fma_f4(wd[5], v2, a);
                                acc[(rr - 1) % 3] = a;
                            }
                            if (rr <= L - 1) {
                                f4 a = fma_f4(wd[0], v0, bias);
                                a = fma_f4(wd[1], v1, a); a = fma_f4(wd[2], v2, a);
                                acc[rr % 3] = a;
                            }
                        }
                    }
#else
#pragma unroll
                    for (int sq = 0; sq < NSEQ; ++sq) {
                        f4 acc[3];
#pragma unroll
                        for (int rr = 0; rr < L + 2; ++rr) {          // input row (first row of the strip) - 1 + rr
                            const unsigned char* rp = cbase + sq * 256 + (rr - 1) * G::ROWP;
                            if constexpr (IMGW_PIPE && !hp_abl(8)) {
                                // strip row rr (read at step rr + 1) goes to the image now: every half of it, from the registers that still hold the
                                // layer's input (outputs reach row rr at step rr + 2), during the first column half's pass only
                                if (sq == 0 && rr >= 1 && rr <= ROWS_W - 2) {
#pragma unroll
                                    for (int h = 0; h < NH; ++h)
                                        *reinterpret_cast<f4*>(tbuf + pix0 + ct * 4 * G::PLANE + tile_off(rr * NH + h)) = cur[rr * NH + h][ct];
                                }
                            }
                            f4 v0, v1, v2;
                            if constexpr (hp_abl(2)) { v0 = bias; v1 = wd[4]; v2 = wd[0]; }
                            else { v0 = *reinterpret_cast<const f4*>(rp - 16); v1 = *reinterpret_cast<const f4*>(rp); v2 = *reinterpret_cast<const f4*>(rp + 16); }
                            if constexpr (hp_abl(1)) {           // reads kept alive, no taps
                                hp_keep(v0); hp_keep(v1); hp_keep(v2);
                                if (rr >= 2) {
                                    const int i = STAGE == 0 ? 2 * (rr - 2) + sq : rr - 2;
                                    cur[i][ct] = relu4(v1);
                                    if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); }
                                }
                                continue;
                            }
                            if (rr >= 2) {                              // completes output row rr - 2
                                f4 a = acc[(rr - 2) % 3];
                                a = fma_f4(wd[6], v0, a); a = fma_f4(wd[7], v1, a); a = fma_f4(wd[8], v2, a);
                                const int i = STAGE == 0 ? 2 * (rr - 2) + sq : rr - 2;
                                cur[i][ct] = relu4(a);
                                if constexpr (PW_FUSED) { if (do_pw) pointwise(An, cur[i]); }     // next layer's 1x1 under the next rows' taps
                            }
                            if (rr >= 1 && rr <= L) {
                                f4 a = acc[(rr - 1) % 3];
                                a = fma_f4(wd[3], v0, a); a = fma_f4(wd[4], v1, a); a = fma_f4(wd[5], v2, a);
                                acc[(rr - 1) % 3] = a;
                            }
                            if (rr <= L - 1) {
                                f4 a = fma_f4(wd[0], v0, bias);
                                a = fma_f4(wd[1], v1, a); a = fma_f4(wd[2], v2, a);
                                acc[rr % 3] = a;
                            }
                        }
                    }
#endif
                }
            }
            };
#if BM_HP_DW_STATIC_MORE
            if (more) dw_pass(std::true_type{}); else dw_pass(std::false_type{});
#else
            dw_pass(more);
#endif
            if constexpr (IMGW_PW) {
                // every wave is past its reads of this layer's image before the next layer's tiles are written over it
                if constexpr (!hp_abl(512)) __syncthreads();
                if (more) pw_all(An, cur);
                if (k + 1 < br) load_pw(li + 2, An);
                BM_PROF(4);
                BM_PROF(5);
            } else {
            if constexpr (!PW_FUSED) {
                if (more) pw_all(An, cur);
            }
            if (k + 1 < br) load_pw(li + 2, An);        // next layer's fused 1x1 uses these; in flight across the barrier and the image write
            BM_PROF(4);
            if constexpr (G::DWREG) { /* no second barrier: the halo buffers alternate */ }
            else if constexpr (NBR) BM_LDS_FLAG_SET(rd_flag + wave, li + 1);
            else if constexpr (!hp_abl(512)) __syncthreads();
            BM_PROF(5);
            }
        }
        if constexpr (EPI_EARLY && !G::DWREG) {
            if (br == 3) {
                if constexpr (NBR) __syncthreads();      // (neighbour flags only order neighbours: the copies below overwrite the whole image)
                stage_epilogue();                        // every wave is past the last read of the image (barrier above)
            }
        }
        // ChannelGate (osnet.py:194-209): crop-wide average -> fc1 -> ReLU -> fc2 -> sigmoid -> scale
        float* part = gap_part + br * (G::NWAVES * G::HID);
        if constexpr (hp_abl(256)) {
            __syncthreads();
            if constexpr (EPI_EARLY && G::DWREG) { if (br == 3) stage_epilogue(); }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) x2[i][ct] = add_f4(x2[i][ct], cur[i][ct]);
            continue;
        }
        {
            float ph[G::HID];
#pragma unroll
            for (int h = 0; h < G::HID; ++h) ph[h] = 0.f;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                f4 s = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NT; ++i) s = add_f4(s, cur[i][ct]);
#pragma unroll
                for (int h = 0; h < G::HID; ++h) {
                    const f4 w1 = *reinterpret_cast<const f4*>(wgate + 4 * (h * MIDP + 16 * ct + 4 * g));
#pragma unroll
                    for (int r = 0; r < 4; ++r) ph[h] = __builtin_fmaf(w1[r], s[r], ph[h]);
                }
            }
#pragma unroll
            for (int h = 0; h < G::HID; ++h) {
                float v = ph[h];
#if BM_HP_GATE_DPP
                v = BM_WAVE_SUM_F32(v);
#else
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
#endif
                if (lane == 0) part[wave * G::HID + h] = v;
            }
        }
        __syncthreads();
        if constexpr (EPI_EARLY && G::DWREG) {
            if (br == 3) stage_epilogue();               // every wave is past its last halo read (it has delivered its gate sums): the copies land under the gate
        }
        float hidv[G::HID];
#pragma unroll
        for (int h = 0; h < G::HID; ++h) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < G::NWAVES; ++w) sum += part[w * G::HID + h];
            const float z = *reinterpret_cast<const float*>(wgate + g_fc1b + 4 * h) + sum * (1.0f / P);
            hidv[h] = z > 0.f ? z : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) {
            const f4 zb = *reinterpret_cast<const f4*>(wgate + g_fc2b + 4 * (16 * ct + 4 * g));
            f4 gate;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + 4 * g + r;
                float z = zb[r];
#pragma unroll
                for (int h = 0; h < G::HID; ++h) z = __builtin_fmaf(*reinterpret_cast<const float*>(wgate + g_fc2w + 4 * (c * G::HID + h)), hidv[h], z);
                gate[r] = 1.f / (1.f + BM_EXPF(-z));
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) x2[i][ct] = fma_f4(gate, cur[i][ct], x2[i][ct]);
        }
        BM_PROF(6);
    }

    // ---- conv3 (1x1 MID -> COUT, linear) + downsample(x) or identity, ReLU (osnet.py:254-260) ----
    if constexpr (!EPI_EARLY) stage_epilogue();
    const unsigned char* w3 = tbuf;
    const unsigned char* b3 = tbuf + (bp.conv3_b - bp.conv3_a);
    const unsigned char* wdn = tbuf + (bp.down_a - bp.conv3_a);
    const unsigned char* wln = tbuf + e_own;                            // EMIT: next block's conv1 pairs [ks][ct]
    const unsigned char* pv3 = tbuf + e_own;                            // RECON: previous block's conv3 pairs, bias, downsample pairs
    const unsigned char* pvb = pv3 + (link.a1 - link.a0);
    const unsigned char* pvd = pv3 + (link.a2 - link.a0);
    const unsigned char* wtl = tbuf + e_own + e_prev;                   // TRANS: pairs [ct][ks], then fp32 bias
    h8 eye;                 // [I | I]: row l16, k-slots j <-> channel 4 g + (j & 3): adds the hi and the lo plane of the shortcut
#pragma unroll
    for (int j = 0; j < 8; ++j) eye[j] = (_Float16)(l16 == 4 * g + (j & 3) ? 1.f : 0.f);
    // operands a tile takes from memory, requested one tile ahead of their use
    struct TileOps {
        h8 dxh[DOWN ? KIN : 1], dxl[DOWN ? KIN : 1];            // DOWN: the block input
        f4 x2p[RECON ? KT : 1];                                  // RECON: previous block's branch sum ...
        h8 rxh[RECON ? KINP : 1], rxl[RECON ? KINP : 1];         // ... and input
        h4 idh[(!DOWN && !RECON) ? NCT : 1], idl[(!DOWN && !RECON) ? NCT : 1];     // identity shortcut
    };
    auto tile_loads = [&](int i, TileOps& t) {
        if constexpr (hp_abl(128)) { t = TileOps{}; return; }
        unsigned p = (wave * NT + i) * 16 + l16;
        if constexpr (STAGE == 0) BM_OPAQUE_U32(p);         // addresses are formed at the use, not kept for 16 tiles
        if constexpr (DOWN) {
            if constexpr (CIN == 16) {
                const unsigned o = p * 16 + g * 4;
                t.dxh[0] = cat8(*reinterpret_cast<const h4*>(xh + o), *reinterpret_cast<const h4*>(xl + o));
            } else {
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks) {
                    const unsigned o = p * CIN + g * (CIN / 4) + 8 * ks;
                    t.dxh[ks] = *reinterpret_cast<const h8*>(xh + o); t.dxl[ks] = *reinterpret_cast<const h8*>(xl + o);
                }
            }
        } else if constexpr (RECON) {
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) t.x2p[ct] = BM_NT_LOAD(reinterpret_cast<const f4*>(x2w + (i * KT + ct) * 256));
            if constexpr (PREV_CIN == 16) {
                const unsigned o = p * 16 + g * 4;
                t.rxh[0] = cat8(*reinterpret_cast<const h4*>(xh + o), *reinterpret_cast<const h4*>(xl + o));
            } else {
#pragma unroll
                for (int ks = 0; ks < KINP; ++ks) {
                    const unsigned o = p * PREV_CIN + g * (PREV_CIN / 4) + 8 * ks;
                    t.rxh[ks] = *reinterpret_cast<const h8*>(xh + o); t.rxl[ks] = *reinterpret_cast<const h8*>(xl + o);
                }
            }
        } else {
#pragma unroll
            for (int co = 0; co < NCT; ++co) {
                const unsigned o = p * CIN + g * (CIN / 4) + 4 * co;
                t.idh[co] = *reinterpret_cast<const h4*>(xh + o); t.idl[co] = *reinterpret_cast<const h4*>(xl + o);
            }
        }
    };
    // The epilogue visits the tiles in GROUPS of TG: the A fragments of an output-channel tile (conv3, downsample / the previous
    // block's conv3 + downsample, then the transition / the next block's conv1) are read from LDS once per group and held in
    // registers while the group's TG tiles run through them -- 1 / TG of the fragment reads of a tile-by-tile epilogue and TG
    // independent accumulator chains on the matrix pipe.  Every (tile, output tile) accumulates in the same order as before.
    constexpr int TG = STAGE == 0 ? (RECON ? BM_HP_EPI_TG0R : BM_HP_EPI_TG0E) : (STAGE == 1 ? BM_HP_EPI_TG1 : 1);
    static_assert(NT % TG == 0 && (!TRANS || TG % 2 == 0), "groups tile the wave's strip; a fused transition pools tile pairs");
    // order in which the epilogue visits the tiles (TRANS: vertically adjacent pairs)
    auto seq_tile = [](int k) constexpr {
        if (!TRANS) return k;
        const int pr = k >> 1, i0 = STAGE == 0 ? (pr >> 1) * 4 + (pr & 1) : 2 * pr;
        return i0 + (k & 1) * (STAGE == 0 ? 2 : 1);
    };
    // Operand prefetch: the tensors these come from were written by an earlier launch (HBM), and one CU draws ~10 B/clk: a
    // group's operands are requested one group ahead (ring of 2 TG sets, static indices under full unrolling).
    constexpr int NR = NT > TG ? 2 * TG : TG;
    TileOps ops[NR];
    // conv3 + shortcut of the TG tiles at sequence positions k0 .. k0 + TG - 1 -> block output as (hi, lo) halves
    auto block_group = [&](int k0, h4 (&yh)[TG][NCT], h4 (&yl)[TG][NCT]) {
        if constexpr (hp_abl(32)) {          // operands kept alive, no MFMAs, no output split
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const TileOps& o = ops[(k0 + t) % NR];
                const int i = seq_tile(k0 + t);
                if constexpr (DOWN) { for (int ks = 0; ks < (CIN == 16 ? 1 : KIN); ++ks) { hp_keep(o.dxh[ks]); if (CIN != 16) hp_keep(o.dxl[ks]); } }
                else if constexpr (RECON) { for (int ct = 0; ct < KT; ++ct) hp_keep(o.x2p[ct]); for (int ks = 0; ks < KINP; ++ks) { hp_keep(o.rxh[ks]); if (PREV_CIN != 16) hp_keep(o.rxl[ks]); } }
                else { for (int co = 0; co < NCT; ++co) { hp_keep(o.idh[co]); hp_keep(o.idl[co]); } }
#pragma unroll
                for (int co = 0; co < NCT; ++co) { yh[t][co] = to_h4(x2[i][0]); yl[t][co] = to_h4(x2[i][KT - 1]); }
            }
            return;
        }
        h8 b2h[TG], b2l[KT == 1 ? 1 : TG], rb2h[RECON ? TG : 1], rb2l[(RECON && KT > 1) ? TG : 1];
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int i = seq_tile(k0 + t);
            h4 xh4[KT], xl4[KT];
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) split4(x2[i][ct], xh4[ct], xl4[ct]);
            if constexpr (KT == 1) b2h[t] = cat8(xh4[0], xl4[0]);
            else { b2h[t] = cat8(xh4[0], xh4[KT - 1]); b2l[t] = cat8(xl4[0], xl4[KT - 1]); }
            if constexpr (RECON) {
                h4 ph4[KT], pl4[KT];
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) split4(ops[(k0 + t) % NR].x2p[ct], ph4[ct], pl4[ct]);
                if constexpr (KT == 1) rb2h[t] = cat8(ph4[0], pl4[0]);
                else { rb2h[t] = cat8(ph4[0], ph4[KT - 1]); rb2l[t] = cat8(pl4[0], pl4[KT - 1]); }
            }
        }
        auto pair_at = [&](const unsigned char* a, h8& hi, h8& lo) {
            if constexpr (hp_abl(2048)) { hi = eye; lo = eye; return; }
            hi = *reinterpret_cast<const h8*>(a + lane * 16); lo = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
        };
#pragma unroll
        for (int co = 0; co < NCT; ++co) {
            const f4 bias3 = *reinterpret_cast<const f4*>(b3 + (16 * co + 4 * g) * 4);
            h8 a3h, a3l;
            pair_at(w3 + (long)co * HP_FRAG_PAIR, a3h, a3l);
            h8 adh[DOWN ? KIN : 1], adl[DOWN ? KIN : 1];
            if constexpr (DOWN) {
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks) pair_at(wdn + (long)(co * KIN + ks) * HP_FRAG_PAIR, adh[ks], adl[ks]);
            }
            h8 p3h, p3l, pdh[RECON ? KINP : 1], pdl[RECON ? KINP : 1];
            f4 biasp;
            if constexpr (RECON) {
                biasp = *reinterpret_cast<const f4*>(pvb + (16 * co + 4 * g) * 4);
                pair_at(pv3 + (long)co * HP_FRAG_PAIR, p3h, p3l);
#pragma unroll
                for (int ks = 0; ks < KINP; ++ks) pair_at(pvd + (long)(co * KINP + ks) * HP_FRAG_PAIR, pdh[ks], pdl[ks]);
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const TileOps& o = ops[(k0 + t) % NR];
                f4 acc = bias3;
                if constexpr (KT == 1) acc = mm2r(a3h, a3l, b2h[t], acc);
                else acc = mm3r(a3h, a3l, b2h[t], b2l[t], acc);
                if constexpr (DOWN) {
                    if constexpr (CIN == 16) acc = mm2r(adh[0], adl[0], o.dxh[0], acc);
                    else {
#pragma unroll
                        for (int ks = 0; ks < KIN; ++ks) acc = mm3r(adh[ks], adl[ks], o.dxh[ks], o.dxl[ks], acc);
                    }
                } else if constexpr (RECON) {
                    // block input = ReLU(conv3_prev . x2_prev + down_prev . x_prev + bias), rebuilt in fp32 and added as it is
                    f4 ap = biasp;
                    if constexpr (KT == 1) ap = mm2r(p3h, p3l, rb2h[t], ap);
                    else ap = mm3r(p3h, p3l, rb2h[t], rb2l[t], ap);
                    if constexpr (PREV_CIN == 16) ap = mm2r(pdh[0], pdl[0], o.rxh[0], ap);
                    else {
#pragma unroll
                        for (int ks = 0; ks < KINP; ++ks) ap = mm3r(pdh[ks], pdl[ks], o.rxh[ks], o.rxl[ks], ap);
                    }
                    acc = add_f4(acc, relu4(ap));
                } else {
                    acc = BM_MFMA_F16_K32(eye, cat8(o.idh[co], o.idl[co]), acc);
                }
                split4(relu4(acc), yh[t][co], yl[t][co]);
            }
        }
    };
    if constexpr (BM_HP_ASYNC_STAGE) BM_WAIT_VM0();         // this wave's operand copies have landed (issued a gate ago in stages 0 / 1)
#pragma unroll
    for (int k = 0; k < NR && k < NT; ++k) tile_loads(seq_tile(k), ops[k]);       // the first sets fly under the staging barrier
    __syncthreads();
    f4 bn[EMIT ? KT : 1];   // EMIT: the next block's conv1 bias, loaded once (a global load inside the loop drains the operand prefetch)
    if constexpr (EMIT) {
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) bn[ct] = *reinterpret_cast<const f4*>(link.w + link.a1 + (16 * ct + 4 * g) * 4);
    }
    _Float16* yh_out = EMIT ? nullptr : out_h + crop * out_px * COUT;         // (EMIT hands over fp32 scratch instead of its output)
    _Float16* yl_out = EMIT ? nullptr : out_l + crop * out_px * COUT;
#pragma unroll
    for (int k0 = 0; k0 < NT; k0 += TG) {
        h4 yh[TG][NCT], yl[TG][NCT];
        block_group(k0, yh, yl);
#pragma unroll
        for (int t = 0; t < TG; ++t)
            if (k0 + t + NR < NT) tile_loads(seq_tile(k0 + t + NR), ops[(k0 + t) % NR]);          // the sets just consumed are free again
        if constexpr (EMIT) {
            // next block's conv1 (COUT -> MID, + bias, ReLU) on the in-register block output
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                f4 an[TG];
#pragma unroll
                for (int t = 0; t < TG; ++t) an[t] = bn[ct];
#pragma unroll
                for (int ks = 0; ks < (hp_abl(32) ? 0 : KSN); ++ks) {
                    const unsigned char* a = wln + (long)(ks * KT + ct) * HP_FRAG_PAIR;
                    const h8 ah = *reinterpret_cast<const h8*>(a + lane * 16), al = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
#pragma unroll
                    for (int t = 0; t < TG; ++t)
                        an[t] = mm3r(ah, al, cat8(yh[t][2 * ks], yh[t][2 * ks + 1]), cat8(yl[t][2 * ks], yl[t][2 * ks + 1]), an[t]);
                }
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    const int i = seq_tile(k0 + t);
                    BM_NT_STORE(reinterpret_cast<f4*>(x1w + (i * KT + ct) * 256), relu4(an[t]));
                    BM_NT_STORE(reinterpret_cast<f4*>(x2w + (i * KT + ct) * 256), x2[i][ct]);
                }
            }
        } else if constexpr (!TRANS) {
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const unsigned p = (wave * NT + seq_tile(k0 + t)) * 16 + l16;
#pragma unroll
                for (int co = 0; co < NCT; ++co) {
                    const unsigned o = p * COUT + g * (COUT / 4) + 4 * co;
                    BM_NT_STORE(reinterpret_cast<h4*>(yh_out + o), yh[t][co]);
                    BM_NT_STORE(reinterpret_cast<h4*>(yl_out + o), yl[t][co]);
                }
            }
        } else {
            // pairs of vertically adjacent tiles: transition conv on the in-register block output, ReLU, 2x2 average (vertical =
            // the two tiles, horizontal = lane ^ 1; the 1/4 is folded into `wtr`), even lanes store the pooled pixel
            static_assert(!TRANS || (STAGE < 2 && COUT % 32 == 0), "fused transition: stages 0 and 1");
            constexpr int WP = G::W / 2;
            const unsigned char* tbias = wtl + (long)NCT * KS3 * HP_FRAG_PAIR;
            static_assert(NCT % 2 == 0, "pooled channel tiles leave in pairs");
            h4 ph[TG / 2], pl[TG / 2];          // the even tile of a pair, held for one iteration
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f4 bv = *reinterpret_cast<const f4*>(tbias + (16 * ct + 4 * g) * 4);
                f4 a[TG];
#pragma unroll
                for (int t = 0; t < TG; ++t) a[t] = bv;
#pragma unroll
                for (int ks = 0; ks < (hp_abl(32) ? 0 : KS3); ++ks) {
                    const unsigned char* af = wtl + (long)(ct * KS3 + ks) * HP_FRAG_PAIR;
                    const h8 ah = *reinterpret_cast<const h8*>(af + lane * 16), al = *reinterpret_cast<const h8*>(af + 1024 + lane * 16);
#pragma unroll
                    for (int t = 0; t < TG; ++t)
                        a[t] = mm3r(ah, al, cat8(yh[t][2 * ks], yh[t][2 * ks + 1]), cat8(yl[t][2 * ks], yl[t][2 * ks + 1]), a[t]);
                }
#pragma unroll
                for (int pr = 0; pr < TG / 2; ++pr) {
                    const int i0 = seq_tile(k0 + 2 * pr);
                    const int row = STAGE == 0 ? wave * (NT / 2) + (i0 >> 1) : wave * NT + i0;     // even image row of tile i0
                    const int po = (row >> 1) * WP + (STAGE == 0 ? (i0 & 1) * 8 : 0) + (l16 >> 1);
                    const f4 a0 = relu4(a[2 * pr]), a1 = relu4(a[2 * pr + 1]);
                    f4 sp;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#if BM_HP_SCALAR_F32
                        float v, w;
                        BM_ADD_F32(a0[r], a1[r], v);
                        BM_ADD_F32(v, BM_QUAD_SWAP1_F32(v), w);
                        sp[r] = w;
#else
                        const float v = a0[r] + a1[r];
                        sp[r] = v + BM_QUAD_SWAP1_F32(v);
#endif
                    }
                    // a lane's channel tiles are adjacent in the lane-group-major layout: tiles (ct - 1, ct) leave as ONE 16-byte
                    // store per plane (8-byte pieces at a 32-byte stride fill a sector in four partial writes)
                    h4 hh, ll;
                    split4(sp, hh, ll);
                    if constexpr (BM_HP_TRANS_ST16) {
                        if (ct & 1) {
                            if ((l16 & 1) == 0) {
                                const unsigned o = po * COUT + g * (COUT / 4) + 4 * (ct - 1);
                                BM_NT_STORE(reinterpret_cast<h8*>(yh_out + o), cat8(ph[pr], hh));
                                BM_NT_STORE(reinterpret_cast<h8*>(yl_out + o), cat8(pl[pr], ll));
                            }
                        } else { ph[pr] = hh; pl[pr] = ll; }
                    } else if ((l16 & 1) == 0) {
                        const unsigned o = po * COUT + g * (COUT / 4) + 4 * ct;
                        BM_NT_STORE(reinterpret_cast<h4*>(yh_out + o), hh);
                        BM_NT_STORE(reinterpret_cast<h4*>(yl_out + o), ll);
                    }
                }
            }
        }
        BM_SCHED_FENCE();
    }
    BM_PROF(7);
    BM_PROF_FLUSH();
#if BM_HP_PERSIST
    if (link.n_crops > 0) __syncthreads();      // persistent: every wave is done with the staged operands before the next crop's copies land
#endif
    }   // crops of this workgroup
}

// ---------------------------------------------------------------------------
// head: conv5 (1x1 C->C + ReLU) -> GAP -> FC(C->F) + BN1d + ReLU -> L2 norm (osnet.py:310-315, 393-396; base_backend.py:206)
// in (hi, lo) [n][128 px][C]; HEAD_NB crops per workgroup of 4 waves: conv5 on (hi, lo) operands (wave w owns output-channel
// tiles 2w, 2w+1, fragments in registers), bias + ReLU + average in fp32; the FC for the 16 crops as one MFMA tile per 16
// features with (hi, lo) parts of both the pooled vectors and the weights.
// ---------------------------------------------------------------------------
#ifndef BM_HP_HEAD_UNROLL
#define BM_HP_HEAD_UNROLL 4         // pixel tiles whose operand loads are in flight together (2: 0.181, 4: 0.164, 8: 0.176 ms per 4096 crops)
#endif
template <int C, int F>
__global__ void __launch_bounds__(256, 2) k_head_hp(const _Float16* __restrict__ in_h, const _Float16* __restrict__ in_l,
                                                    const unsigned char* __restrict__ wts5, const unsigned char* __restrict__ wfc,
                                                    float* __restrict__ out_base, const int* __restrict__ out_rows,
                                                    const int* __restrict__ count, int n_total) {
    static_assert(C == 128 && F % 64 == 0, "head: 128 channels in, a multiple of 64 features out");
    constexpr int NCT = C / 16, KS = C / 32, P = 128, NB = HEAD_NB, VS = C + 4;
    constexpr int FT_PER_WAVE = F / 16 / 4;
    constexpr int HEAD_UNROLL = BM_HP_HEAD_UNROLL;          // (a constant expression: a macro inside a pragma does not survive -save-temps)
    __shared__ __attribute__((aligned(16))) float vbuf[NB * VS];          // pooled vectors (L-layout channel order)
    __shared__ float red[4 * NB];
    int n_eff = n_total;
    if (count) { const int c = *count; n_eff = c < n_total ? c : n_total; }
    const long crop0 = (long)blockIdx.x * NB;
    if (crop0 >= n_eff) return;
    const int nb = n_eff - crop0 < NB ? (int)(n_eff - crop0) : NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const unsigned char* bias5 = wts5 + (long)NCT * KS * HP_FRAG_PAIR;
    h8 ah[2][KS], al[2][KS];
    f4 bias[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int ct = 2 * wave + c2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ah[c2][ks] = *reinterpret_cast<const h8*>(wts5 + (long)(ct * KS + ks) * HP_FRAG_PAIR + lane * 16);
            al[c2][ks] = *reinterpret_cast<const h8*>(wts5 + (long)(ct * KS + ks) * HP_FRAG_PAIR + 1024 + lane * 16);
        }
        bias[c2] = *reinterpret_cast<const f4*>(bias5 + (16 * ct + 4 * g) * 4);
    }
    for (int e = tid; e < NB * VS; e += 256) vbuf[e] = 0.f;                 // rows of absent crops stay zero
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < nb; ++k) {
        const _Float16* xh = in_h + (crop0 + k) * (long)(P * C);
        const _Float16* xl = in_l + (crop0 + k) * (long)(P * C);
        f4 sum[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll HEAD_UNROLL
        for (int i = 0; i < P / 16; ++i) {
            const int p = i * 16 + l16;
            h8 bh[KS], bl[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bh[ks] = *reinterpret_cast<const h8*>(xh + p * C + g * (C / 4) + 8 * ks);
                bl[ks] = *reinterpret_cast<const h8*>(xl + p * C + g * (C / 4) + 8 * ks);
            }
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                f4 acc = bias[c2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    acc = BM_MFMA_F16_K32(ah[c2][ks], bh[ks], acc);
                    acc = BM_MFMA_F16_K32(ah[c2][ks], bl[ks], acc);
                    acc = BM_MFMA_F16_K32(al[c2][ks], bh[ks], acc);
                }
                sum[c2] += relu4(acc);
            }
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sum[c2][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (l16 == 0) vbuf[k * VS + g * (C / 4) + 4 * (2 * wave + c2) + r] = v * (1.0f / P);
            }
    }
    __syncthreads();
    // ---- FC + BN1d (folded) + ReLU for the NB crops: D[f][crop] = sum_c W[f][c] * v[crop][c], (hi, lo) on both sides ----
    const _Float16* fch = reinterpret_cast<const _Float16*>(wfc);
    const _Float16* fcl = fch + (long)F * C;
    const float* fcb = reinterpret_cast<const float*>(wfc + (long)F * C * 4);
    h8 vh[KS], vl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const f4 v0 = *reinterpret_cast<const f4*>(vbuf + l16 * VS + 32 * ks + 8 * g);
        const f4 v1 = *reinterpret_cast<const f4*>(vbuf + l16 * VS + 32 * ks + 8 * g + 4);
        h4 h0, l0, h1, l1;
        split4(v0, h0, l0); split4(v1, h1, l1);
        vh[ks] = cat8(h0, h1); vl[ks] = cat8(l0, l1);
    }
    f4 vals[FT_PER_WAVE];
    float sq = 0.f;
#pragma unroll
    for (int fi = 0; fi < FT_PER_WAVE; ++fi) {
        const int ft = wave * FT_PER_WAVE + fi;
        f4 acc = *reinterpret_cast<const f4*>(fcb + 16 * ft + 4 * g);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const h8 wh = *reinterpret_cast<const h8*>(fch + (long)(16 * ft + l16) * C + 32 * ks + 8 * g);
            const h8 wl = *reinterpret_cast<const h8*>(fcl + (long)(16 * ft + l16) * C + 32 * ks + 8 * g);
            acc = BM_MFMA_F16_K32(wh, vh[ks], acc);
            acc = BM_MFMA_F16_K32(wh, vl[ks], acc);
            acc = BM_MFMA_F16_K32(wl, vh[ks], acc);
        }
        acc = relu4(acc);
        vals[fi] = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) sq = __builtin_fmaf(acc[r], acc[r], sq);
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < 16) red[wave * NB + l16] = sq;
    __syncthreads();
    const float nrm = sqrtf(red[l16] + red[NB + l16] + red[2 * NB + l16] + red[3 * NB + l16]);
    if (l16 < nb) {
        float* out = out_base + (out_rows ? (long)out_rows[crop0 + l16] : crop0 + l16) * F;
#pragma unroll
        for (int fi = 0; fi < FT_PER_WAVE; ++fi) {
            const int ft = wave * FT_PER_WAVE + fi;
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = vals[fi][r] / nrm;
            *reinterpret_cast<f4*>(out + 16 * ft + 4 * g) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3 (3 -> 16) + BN + ReLU + maxpool 3x3 stride 2 pad 1 (osnet.py:294-295) on (hi, lo) fp16 RGBX
// crops with a 3-pixel zero border, [n][262][136][4] per plane (k_crop_resize_rgbx with a lo plane); output (hi, lo)
// [n][64*32][16].  One workgroup (8 waves) per crop; wave w produces pooled rows 8w..8w+7; the 7x7x3 window is 7 k-steps (one
// per kernel row) of three MFMAs.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_stem_hp(const _Float16* __restrict__ crops_h, const _Float16* __restrict__ crops_l,
                                                 _Float16* __restrict__ out_h, _Float16* __restrict__ out_l,
                                                 const unsigned char* __restrict__ wts, const int* __restrict__ count) {
    if (count && (int)blockIdx.x >= *count) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x;
    const _Float16* imh = crops_h + crop * (STEM_ROWS * STEM_COLS * 4);
    const _Float16* iml = crops_l + crop * (STEM_ROWS * STEM_COLS * 4);
    _Float16* yh = out_h + crop * (64 * 32) * 16;
    _Float16* yl = out_l + crop * (64 * 32) * 16;
    h8 ah[7], al[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        ah[ky] = *reinterpret_cast<const h8*>(wts + ky * HP_FRAG_PAIR + lane * 16);
        al[ky] = *reinterpret_cast<const h8*>(wts + ky * HP_FRAG_PAIR + 1024 + lane * 16);
    }
    const f4 bias = *reinterpret_cast<const f4*>(wts + 7 * HP_FRAG_PAIR + 4 * g * 4);
    auto conv_row = [&](int cy, f4 (&row)[4]) {
        if (cy < 0 || cy > 127) {
#pragma unroll
            for (int t = 0; t < 4; ++t) row[t] = f4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 acc = bias;
            const int cx = t * 16 + l16;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const long o = ((long)(2 * cy + ky) * STEM_COLS + 2 * cx + 2 * g) * 4;
                const h8 bh = *reinterpret_cast<const h8*>(imh + o), bl = *reinterpret_cast<const h8*>(iml + o);
                acc = BM_MFMA_F16_K32(ah[ky], bh, acc);
                acc = BM_MFMA_F16_K32(ah[ky], bl, acc);
                acc = BM_MFMA_F16_K32(al[ky], bh, acc);
            }
            row[t] = relu4(acc);
        }
    };
    f4 prev[4], mid[4], next[4];
    const int oy0 = wave * 8;
    conv_row(2 * oy0 - 1, prev);
#pragma unroll 1
    for (int oy = oy0; oy < oy0 + 8; ++oy) {
        conv_row(2 * oy, mid);
        conv_row(2 * oy + 1, next);
        f4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = max4(max4(prev[t], mid[t]), next[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 m = v[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // all values are >= 0 (post-ReLU), so a missing neighbour reads as 0
                const float right = BM_ROW_SHL1_F32(v[t][r]);
                float left = BM_ROW_SHR1_F32(v[t][r]);
                const float left_prev_tile = BM_ROW_ROR1_F32(v[t > 0 ? t - 1 : 0][r]);
                if (l16 == 0) left = t > 0 ? left_prev_tile : 0.f;
                const float mm = m[r] > right ? m[r] : right;
                m[r] = mm > left ? mm : left;
            }
            if ((l16 & 1) == 0) {
                const int p = oy * 32 + t * 8 + (l16 >> 1);
                h4 hh, ll;
                split4(m, hh, ll);
                *reinterpret_cast<h4*>(yh + (long)p * 16 + g * 4) = hh;
                *reinterpret_cast<h4*>(yl + (long)p * 16 + g * 4) = ll;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) prev[t] = next[t];
    }
}

// crop -> resize -> normalise into the stem's (hi, lo) fp16 RGBX planes (interior only; border and X channel stay zero)
__global__ void k_crop_resize_rgbx_hl(const uint8_t* const* frames, const int* crop_stream, const float* boxes, int box_stride, int W,
                                      int H, const float* lut, _Float16* out_hi, _Float16* out_lo, int rows_per_block, const int* count,
                                      int pad) {
    if (count && (int)blockIdx.x >= *count) return;
    const int i = blockIdx.x;
    const int dx = threadIdx.x;
    const uint8_t* frame = frames[crop_stream[i]];
    const CropRect r = crop_rect(boxes + (long)i * box_stride, W, H);
    const long row_stride = (long)W * 3;
    const uint8_t* src = frame + (long)r.y1 * row_stride + r.x1 * 3;
    const ResizeAxis ax = resize_axis_x(dx, REID_IN_W, r.w > 0 ? r.w : 1);
    const PadGeom pg = pad_geom(r, REID_IN_W, REID_IN_H);
    const int y0 = blockIdx.y * rows_per_block;
    for (int dy = y0; dy < y0 + rows_per_block && dy < REID_IN_H; ++dy) {
        h4 ph, pl;
        for (int c = 0; c < 3; ++c) {
            const int v = preprocess_sample(src, row_stride, r, pg, pad, ax, dy, dx, 2 - c, REID_IN_W, REID_IN_H);
            const float f = lut[c * 256 + v];
            ph[c] = (_Float16)f;
            pl[c] = (_Float16)(f - (float)ph[c]);
        }
        ph[3] = (_Float16)0.f; pl[3] = (_Float16)0.f;
        const long o = (((long)i * STEM_ROWS + dy + 3) * STEM_COLS + dx + 3) * 4;
        *reinterpret_cast<h4*>(out_hi + o) = ph;
        *reinterpret_cast<h4*>(out_lo + o) = pl;
    }
}

}  // namespace bm
