// Table-driven implicit-GEMM Conv3d for gfx950 (MI355X): v_mfma_f32_32x32x16_{bf16,f16}.
//
// One workgroup = one box of output rows (host planner: video_distillation_amd/plan.py).
//   * the input patch of the box is staged ONCE per 8-channel chunk into LDS by LDS-DMA
//     (global_load_lds_dwordx4: 64 x 16-byte slots per wave-instruction, per-lane source address
//     from a host-built gather table at dword granularity, zero fill from a 16-byte zero slot), so
//     every source byte is fetched once per box and re-used by all taps (147 for the 3x7x7
//     kernels) out of LDS;
//   * waves split the output channels (NT tiles of 32) and, if N < 128, the rows (MW);
//     each wave keeps MTW 32x32 fp32 accumulator tiles in registers;
//   * per K-step (2 taps x 8 channels) a wave issues ONE 1-KiB coalesced load of its
//     pre-packed B fragment (prefetched DB steps ahead, counted vmcnt) and MTW ds_read_b128 A
//     fragments (one step ahead; bank-conflict-free row maps are chosen by the planner), then
//     MTW (x1) or 3*MTW (x3 split precision) MFMAs;
//   * epilogue: bias + ReLU + 2x2x2 / 1x2x2 max-pool with arg-max entirely in-lane (the
//     planner puts the 8 rows of a pool window into one lane's registers); channels-last outputs
//     are staged through LDS and stored as whole 16-byte slots; or plain fp32 rows (dgrad);
//   * workgroups are mapped to boxes XCD-contiguously; for the first layer (short boxes) each
//     workgroup walks several consecutive boxes.
// Diagnostics: VdConvParams.dbg bits 0-2 ablate epilogue / K loop / DMA, bit 3 records s_memtime
// stamps of the workgroup phases (tools/ablate.py, tools/stamps.py); 0 in production.
//
// Replaces nn.Conv3d / nn.ReLU / nn.MaxPool3d of ConvNet3D.features (reference
// networks.py:757, 768-770, 799) and the input-gradient half of their autograd backward.
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#include "../../include/vd_hip.h"

// tuning switches (compile-time A/B builds; defaults are the shipped configuration)
#ifndef VD_DB_X1
#define VD_DB_X1 3
#endif
#ifdef VD_NO_SCHED_BARRIER
#define VD_SCHED_BARRIER()
#else
#define VD_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef VD_NO_SETPRIO_LOOP      // (round 5: the single-pass K loop of a channel chunk runs at priority 2 -- above a partner workgroup's patch DMA /
#define VD_PRIO_LOOP(x) __builtin_amdgcn_s_setprio(x)      //  epilogue phases on the same SIMD: level 1 -0.8 % alone, the DM step -0.1 ms in
#else                           //  four of four same-box pairs, profiles/r05_prioloop_ab.txt)
#define VD_PRIO_LOOP(x)
#endif
#ifdef VD_SETPRIO
#define VD_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define VD_PRIO(x)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// VD_PREC_F16C8: power-of-two scalings of the fp8 operands of the correction products (E8M0 codes = 127 + log2 of what the stored
// byte must be multiplied by): low parts of the activations are stored x 2^9 (a_lo <= 2^-11 a), the fp8 image of a high fragment
// is a / 4 (v_cvt_scalef32_pk_fp8_f16 divides by its scale operand; beyond 464 it returns NaN, hence the clamp at 1792)
#define VD_C8_ALO_SHIFT 9
#define VD_C8_SA_LO (127 - VD_C8_ALO_SHIFT)
#define VD_C8_SA_HI (127 + 2)
__device__ __forceinline__ uint32_t vd_c8_hi_byte(uint16_t h16) {      // e4m3 byte of (fp16 value) / 4
    // (the PRODUCER of the plane clamps its outputs to 1792 -- emit_lo = 2 -- so a / 4 stays inside e4m3's finite range; the
    //  instruction returns NaN beyond 464)
    const _Float16 h = __builtin_bit_cast(_Float16, h16);
    const f16x2 hh = {h, h};
    i16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, hh, 4.0f, false);
    return (uint32_t)__builtin_bit_cast(int, r) & 0xffu;
}
__device__ __forceinline__ void vd_c8_image(const uint4& a, int& w0, int& w1, const float div) {      // 8 f16 -> 8 e4m3 bytes of a / div
    i16x2 r0, r1;                     // (every byte is written below: no initialisation)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
    r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, __builtin_bit_cast(f16x2, a.x), div, false);
    r0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r0, __builtin_bit_cast(f16x2, a.y), div, true);
    r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, __builtin_bit_cast(f16x2, a.z), div, false);
    r1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r1, __builtin_bit_cast(f16x2, a.w), div, true);
#pragma clang diagnostic pop
    w0 = __builtin_bit_cast(int, r0); w1 = __builtin_bit_cast(int, r1);
}
__device__ __forceinline__ uint32_t vd_c8_lo_byte(float v) {            // e4m3 byte of (v - rn16(v)) * 2^9, clamped to the finite range
    float r = (v - (float)(_Float16)v) * (float)(1 << VD_C8_ALO_SHIFT);
    r = fminf(fmaxf(r, -448.f), 448.f);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(r, r, 0, false) & 0xffu;
}

// Per-kernel, per-DEVICE launch preparation (160 KB dynamic LDS attribute, CU count): hipFuncSetAttribute applies to the
// current device only, and several host threads / devices may launch the same instantiation.
struct VdDevCache {
    std::mutex mu;
    bool done[64] = {};
    int ncu[64] = {};
};
static int vd_dev_prepare(const void* kern, VdDevCache& c, int& ncu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return -4;
    std::lock_guard<std::mutex> lk(c.mu);
    if (!c.done[dev]) {
        e = hipDeviceGetAttribute(&c.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || c.ncu[dev] <= 0) return e != hipSuccess ? (int)e : -4;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        c.done[dev] = true;
    }
    ncu = c.ncu[dev];
    return 0;
}

template <int PREC>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, f32x16 c) {
    if constexpr (PREC == VD_PREC_BF16 || PREC == VD_PREC_BF16X3) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PREC>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const u32x4& b, f32x16 c) {
    if constexpr (PREC == VD_PREC_BF16 || PREC == VD_PREC_BF16X3) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
}

template <int PREC>
__device__ __forceinline__ void split16(float v, uint16_t& hi, uint16_t& lo) {
    if constexpr (PREC == VD_PREC_BF16 || PREC == VD_PREC_BF16X3) {
        __bf16 h = (__bf16)v;
        __bf16 l = (__bf16)(v - (float)h);
        hi = __builtin_bit_cast(uint16_t, h);
        lo = __builtin_bit_cast(uint16_t, l);
    } else {
        _Float16 h = (_Float16)v;
        _Float16 l = (_Float16)(v - (float)h);
        hi = __builtin_bit_cast(uint16_t, h);
        lo = __builtin_bit_cast(uint16_t, l);
    }
}

// SO = second-order variant: split sources + select epilogue (gradient matching / MTT); a separate
// instantiation so that the extra parameters cost the hot forward / dgrad programs no scalar registers.
// minimum waves per SIMD the register allocation must allow: the short first-layer programs (<= 4 accumulator
// tiles, single-pass formats) hide their per-box latencies with more resident workgroups
// Ablation / timing hooks (VdConvParams.dbg) are compiled in only with -DVD_DBG_HOOKS=1 (tools/ablate.py, stamps.py
// load that build): in the production library the conditions are the constant 0 and cost the hot loops nothing.
#ifndef VD_DBG_HOOKS
#define VD_DBG_HOOKS 0
#endif
#if VD_DBG_HOOKS
#define VD_DBG(p) ((p).dbg)
#else
#define VD_DBG(p) 0
#endif
#ifndef VD_OCC_SMALL
#define VD_OCC_SMALL 2
#endif
// SQ (hi+lo formats): PLANE-SEQUENTIAL K loop -- only ONE operand plane of a channel chunk's patch is resident at a time: the chunk
// runs as two passes, (A_hi x B_lo + A_hi x B_hi) over the high plane, then A_lo x B_hi over the low plane staged into the same
// LDS, so the program needs the LDS of a single-pass one and TWO workgroups share a CU where one did.  Same products; the
// low-plane products of a chunk are added after its high-plane ones instead of interleaved with them (fp32 summation order).
// It pays where a box has ONE chunk and a patch that is expensive to stage -- the first level (2 x 55 KB of dword-aligned kw-slots
// per box: 5.44 -> 4.81 ms per 512 clips, same box) -- and costs where the chunks are many and short (16 chunks of the last
// level: twice the chunk boundaries, 0.66 -> 0.77 ms; level 1: 8.29 -> 8.40), so only the first-level programs are dispatched to it.
#define VD_OCC(PREC, MTW, NTW, BAL) \
    ((((MTW) * (NTW) + (BAL)) <= 4 && ((PREC) == VD_PREC_BF16 || (PREC) == VD_PREC_F16)) ? VD_OCC_SMALL : 2)
// NTW = N tiles (of 32 output channels) per wave: with 2, an A fragment read from LDS feeds two MFMAs,
// which halves the LDS read traffic per MFMA (the co-critical resource of the NTW = 1 layout).
// BAL = 1 (with NTW = 2, MTW = 3): boxes of 7 M tiles on 2 x 2 waves -- every wave owns 3 M tiles x 2 N
// tiles plus ONE N tile of the seventh M tile: 7 MFMAs per K step for 4 A-fragment reads (instead of 7).
template <int I> struct VdIC { static constexpr int v = I; };
template <class F, int... Is>
__device__ __forceinline__ void vd_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(VdIC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void vd_static_for(F&& f) { vd_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// (the kernel body is a device function so that two entry points share it: one program per launch -- p by value in the kernel
//  arguments -- and up to VD_MULTI_MAX programs of the same instantiation in ONE launch, conv_mfma_multi_kernel below)
template <int PREC, int MTW, bool SO = false, int NTW = 1, int BAL = 0, bool SQ = false>
__device__ __forceinline__ void conv_mfma_body(const VdConvParams& p, const int boxes_per_wg, const int total_boxes, const int block_id, const int nblocks) {
    constexpr bool C8 = (PREC == VD_PREC_F16C8);        // fp16 main product + fp8 corrections (two operand planes, like the hi+lo formats)
    constexpr bool X3 = (PREC == VD_PREC_BF16X3 || PREC == VD_PREC_F16X3 || C8);
    static_assert(!C8 || (MTW == 4 && NTW == 1 && BAL == 0 && !SO && !SQ), "VD_PREC_F16C8: the last level's 4 x 1-tile layout only");
    constexpr bool EXT = SO || MTW == 5;
    constexpr bool SEQ = X3 && SQ;                      // hi+lo formats: one operand plane resident at a time (see SQ above)
    constexpr int LPL = (X3 && !SEQ) ? 2 : 1;           // patch planes resident in LDS
    // ALT (hi+lo formats, round 6): the matrix instruction rounds with a DIRECTED bias -- 600 chained v_mfma_f32_32x32x16_f16 end 1.4 ulp
    // BELOW the exact sum whatever the sign of the products (tools/micro/mfma_round_probe, profiles/r06_mfma_rounding.txt) -- which is
    // 1e-7 of an output and invisible per element, but coherent: sums over 1e7 - 1e8 outputs of the gradient programs (bias gradients,
    // the hallucinator's parameter gradients) do not average it out the way a CPU's round-to-nearest does.  So the accumulation's SIGN
    // alternates per channel chunk: at every chunk boundary the accumulators are negated (16 x TILES v_xor per wave and chunk) and the
    // B fragments of odd chunks get their sign bits flipped as they are loaded -- the same products, accumulated into -sum in every
    // other chunk, so that the hardware's pull toward minus infinity pulls the result up as often as down.  One chunk: nothing changes.
    constexpr bool ALT = X3 && !C8;
    constexpr int TILES = MTW * NTW + BAL;              // accumulator tiles per wave
    constexpr int MA = MTW + BAL;                       // M tiles (A fragments per K step) a wave touches
    static_assert(BAL == 0 || (NTW == 2 && !X3 && !SO), "balanced layout: single-pass formats, two N tiles per wave");
    constexpr int H0 = (MTW + 1) / 2;   // tiles whose A fragments are fetched one half-step ahead
    constexpr int H1 = MTW - H0;
    constexpr int AD = 1;                               // x1: A-fragment prefetch distance (K-steps)
    constexpr int DB = C8 ? 3 : X3 ? (TILES <= 2 ? 5 : ((SEQ && NTW == 1) ? 3 : 2)) : VD_DB_X1;     // (C8: a ring of 4 = one correction group)   // B-fragment prefetch distance (K steps); x1: (DB+1) % (AD+1) == 0
    //   (plane-sequential: the low-plane pass has one MFMA per tile and step instead of three, so its steps are short)
    static_assert(X3 || (DB + 1) % (AD + 1) == 0, "ring sizes must divide the unroll factor");
    constexpr int LU = C8 ? 6 : (TILES <= 4) ? 14 : 17;   // DMA groups per wave whose gather entries stay in registers (4 waves x LU x 64 slots >= the plan's patch;
    //                                                       the fp8-corrected program: its position-tile patches are 23 groups, larger ones re-read the table)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncols = p.NT / NTW;                      // wave columns; p.NT counts all N tiles
    const int wn = wave % ncols;
    const int wm = wave / ncols;
    // NTW = 2: boxes of 7 M tiles run on 2 wave rows x 4 tiles; the wave row that owns the padding tile
    // skips its LDS reads and MFMAs (the MFMA pipe of its SIMD is then free for the co-resident workgroup)
    bool short_row = false;
    if constexpr (NTW == 2) short_row = p.mt_valid > 0 && (wm + 1) * MTW > p.mt_valid;
    const int half = lane >> 5;

    // XCD-aware block order (T1): workgroups are dealt round-robin to the 8 XCDs, so give each XCD a
    // CONTIGUOUS range of boxes -- neighbouring boxes of a clip share halo rows, which then hit
    // in that XCD's private L2 instead of being fetched once per XCD.  Bijective for any grid.
    int wgid;
    {
        const int nwg = nblocks, q = nwg >> 3, r = nwg & 7;
        const int xcd = block_id & 7, k = block_id >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // Each workgroup walks `boxes_per_wg` consecutive boxes (launching one workgroup per box is
    // dispatch-rate bound for the first layer: ~100 000 boxes of ~10 us each per launch).
    unsigned long long t_stamp[8];
    auto stamp = [&](int k) { if (VD_DBG(p) & 8) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); t_stamp[k] = t; } };
    auto finish = [&]() {
        if ((VD_DBG(p) & 8) && tid == 0 && p.stamps != nullptr) {
            stamp(7);
            unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stamps) + (size_t)block_id * 8;
            for (int k = 0; k < 8; ++k) o[k] = t_stamp[k];
        }
    };
    const int ngroups = (int)(p.gather_stride >> 6);   // 64-slot DMA groups per patch
    const int nwaves = nthreads >> 6;
    const int plane_bytes = p.lds_plane_bytes;
    int* lds_tap = reinterpret_cast<int*>(smem + LPL * plane_bytes);
    int* lds_otab = lds_tap + 2 * p.S;
    int* lds_skip = lds_otab + (p.MW * MTW + BAL) * 4;      // VD_PREC_F16C8: S / 4 tile skip masks behind the type's tap offsets
    const bool one_type = (p.ntypes == 1);
    const uint32_t* zslot = reinterpret_cast<const uint32_t*>(p.zero_slot);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.src);   // dword addressing: a slot may start at any dword
    const uint4* wbase = reinterpret_cast<const uint4*>(p.wpk);
    const int64_t w_lo = p.w_plane_stride >> 3;  // uint4 units
    const int wstep = p.NT * 64;                 // uint4 per K-step
    const int S = p.S;
    int a_off[MA];
    const int mt_tot = p.MW * MTW + BAL;               // M tiles of the workgroup
    auto tile_of = [&](int i) { return (BAL && i == MTW) ? p.MW * MTW : wm * MTW + i; };
    if (one_type) {   // box-type tables are launch constants: load them once per workgroup
        const int32_t* a_tab0 = p.tables + p.tab_ofs[0];
        const int32_t* o_tab0 = p.tables + p.tab_ofs[1];
        const int32_t* t_tab0 = p.tables + p.tab_ofs[2];
        for (int k = tid; k < 2 * p.S; k += nthreads) lds_tap[k] = t_tab0[k];
        if constexpr (C8)
            for (int k = tid; k < (p.S >> 2); k += nthreads) lds_skip[k] = t_tab0[2 * p.S + k];
        if (p.epi != VD_EPI_ROWS)
            for (int k = tid; k < mt_tot * 4; k += nthreads) lds_otab[k] = o_tab0[k];
#pragma unroll
        for (int i = 0; i < MA; ++i) a_off[i] = a_tab0[tile_of(i) * 32 + (lane & 31)];
    }

  constexpr bool LOOPED = (TILES <= 4);   // only the first layer's instantiations have boxes short enough to need it
  const int nbit = LOOPED ? boxes_per_wg : 1;
  for (int bit = 0; bit < nbit; ++bit) {
    const int bid = wgid * nbit + bit;
    if (bid >= total_boxes) break;
    stamp(0);
    int grp = bid / p.nbox;
    int bi = bid - grp * p.nbox;
    if constexpr (EXT) {
        // weight-gradient programs: every box has its own B operand (packed dy, MBs), shared by all clip
        // groups (= input channels).  Box-major order makes the workgroups resident on an XCD stream the
        // SAME B concurrently, so it is fetched into that L2 once instead of once per channel.
        if (p.w_box_stride != 0) {
            const int ngrp = total_boxes / p.nbox;
            bi = bid / ngrp;
            grp = bid - bi * ngrp;
        }
    }
    if constexpr (C8) {
        // position-tile programs, persist bit 19: WINDOW-major box order -- with the XCD-contiguous ranges above an XCD then works on
        // one pool window, i.e. streams ONE of the program's packed B operand sets through its L2 instead of all of them
        if (p.persist & 0x80000) {
            const int ngrp = total_boxes / p.nbox;
            bi = bid / ngrp;
            grp = bid - bi * ngrp;
        }
    }
    const int clip0 = grp * p.ncl;

    // (1) the gather entries of this wave's DMA groups: the longest dependent chain of the prologue,
    //     so they are requested first; they are the same for every channel chunk and stay in registers.
    const int32_t* gtab = p.gather + (int64_t)bi * p.gather_stride;
    constexpr bool HOIST = (TILES < 8);   // MTW = 8 has no registers to spare: it re-reads the table per chunk
#ifndef VD_C8_HOIST
#define VD_C8_HOIST 1
#endif
    const bool hoist = HOIST && (!C8 || (VD_C8_HOIST && ngroups <= nwaves * LU));
    uint32_t goff[HOIST ? LU : 1];
#pragma unroll
    for (int u = 0; u < (HOIST ? LU : 1); ++u) {
        const int gi = wave + u * nwaves;
        const int e = gtab[((gi < ngroups) ? gi : ngroups - 1) * 64 + lane];
        const int ci = e >> 24;
        const bool ok = (e >= 0) && (clip0 + ci < p.nclips);
        goff[u] = ok ? ((uint32_t)ci * (uint32_t)p.src_clip_stride4 + (uint32_t)(e & 0xFFFFFF)) : 0xFFFFFFFFu;
    }
    // (2) one 32-byte row per box: table offsets of its type + output origin (scalar loads)
    //     (with a single box type the table offsets are launch constants, so nothing in the prologue
    //      depends on this load: it is only needed for the output origin in the epilogue)
    const int32_t* box = p.boxes + bi * 8;
    const int32_t* o_tab = p.tables + (one_type ? p.tab_ofs[1] : box[1]);
    const int out_rel = box[3];
    if (!one_type) {
        const int32_t* a_tab = p.tables + box[0];
        const int32_t* t_tab = p.tables + box[2];
        __syncthreads();   // the previous box's epilogue is done with the LDS tables
        for (int k = tid; k < 2 * p.S; k += nthreads) lds_tap[k] = t_tab[k];
        if constexpr (C8)
            for (int k = tid; k < (p.S >> 2); k += nthreads) lds_skip[k] = t_tab[2 * p.S + k];
        if (p.epi != VD_EPI_ROWS)
            for (int k = tid; k < mt_tot * 4; k += nthreads) lds_otab[k] = o_tab[k];
#pragma unroll
        for (int i = 0; i < MA; ++i) a_off[i] = a_tab[tile_of(i) * 32 + (lane & 31)];
    }
#pragma unroll
    for (int u = 0; u < (HOIST ? LU : 1); ++u) asm volatile("" : "+v"(goff[u]));   // consumed before any DMA is in flight

    f32x16 acc[TILES];      // [j * MTW + i]: N tile j of this wave, M tile i
#pragma unroll
    for (int i = 0; i < TILES; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;

    for (int cc = 0; cc < p.CC; ++cc) {
      // (plane-sequential hi+lo: the chunk body runs twice, ph = 0 with the high plane of the patch resident, ph = 1 with the low
      //  plane; ph is a compile-time constant of each copy)
      vd_static_for<SEQ ? 2 : 1>([&](auto PHC) __attribute__((always_inline)) {
        constexpr int ph = decltype(PHC)::v;
        __syncthreads();  // previous chunk's fragment reads are done
        // ---- stage the patch of this channel chunk: LU independent 16-byte loads in flight ----
        // first B fragments of this chunk: issued before the patch DMA so both latencies overlap
        // (per-box B operands and atomic accumulation exist only in the weight-gradient (MTW 5) and
        //  second-order instantiations: two scalar registers the hot programs do not pay for)
        // (single-pass programs: w_set_clips > 0 selects one of several operand sets, w_plane_stride apart, by the box's first
        //  clip -- the dithered weights of the real side, distill.HipBackend.embed_pool)
        // (the sign flip is applied where a fragment is USED, a step or two after its load was issued: flipped at the load it would have to
        //  land at once -- the B prefetch ring would be gone, measured -7 .. -10 % on the hi+lo programs.  persist bit 20: alternation off, A/B)
        const bool alt_on = ALT && !(p.persist & 0x100000);
        if constexpr (ALT) {
            if (alt_on && ph == 0 && cc > 0) {
#pragma unroll
                for (int i = 0; i < TILES; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][k] = -acc[i][k];
            }
        }
        const uint32_t bsgn = (alt_on && (cc & 1)) ? 0x80008000u : 0u;      // sign bits of the two 16-bit values of a dword
        const int64_t wset = (!X3 && !EXT && p.w_set_clips > 0) ? (int64_t)(clip0 / p.w_set_clips) * w_lo : (int64_t)0;
        const uint4* wp = wbase + wset + ((EXT || C8) ? (((int64_t)bi * p.w_box_stride) >> 3) : (int64_t)0) + ((int64_t)cc * S * p.NT + wn * NTW) * 64 + lane;
        auto load_b = [&](int s, uint4* bh, uint4* bl) {
            const int sc = (VD_DBG(p) & 16) ? 0 : ((s < S) ? s : S - 1);   // dbg 16: always the same (cached) B fragment
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                bh[j] = wp[(int64_t)sc * wstep + j * 64];
                if constexpr (X3 && !C8 && (!SEQ || ph == 0)) bl[j] = wp[(int64_t)sc * wstep + j * 64 + w_lo];   // (C8: plane 1 holds fp8 pieces, loaded by the K loop itself)
            }
        };
        uint4 bqh[DB + 1][NTW], bql[DB + 1][NTW];
#pragma unroll
        for (int u = 0; u <= DB; ++u)
#pragma unroll
            for (int j = 0; j < NTW; ++j) bql[u][j] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < DB; ++u) load_b(u, bqh[u], bql[u]);
        // channel chunks >= src_split_cc live in a second tensor of the same shape (K-concatenated operands
        // of the second-order passes): its distance from the first one is src_split_off4 dwords
        const uint32_t* csrc = src + (int64_t)clip0 * p.src_clip_stride4 + (int64_t)cc * p.src_chunk_stride4;
        if constexpr (TILES <= 4) {
            // first-layer programs (one clip per box) may read their clip through an index: the real pool is kept
            // resident as 16-bit pixel rows and a batch is just a list of pool indices (get_images, distill_baseline.py:84-90)
            if (p.clip_index != nullptr) csrc = src + p.clip_index[clip0] * (int64_t)p.src_clip_stride4;
        }
        if constexpr (SO) {
            if (p.src_split_cc > 0 && cc >= p.src_split_cc)
                csrc = src + (int64_t)clip0 * p.src_clip_stride4 + p.src_split_off4 +
                       (int64_t)(cc - p.src_split_cc) * p.src_chunk_stride4;
        }
        if (cc == 0 && ph == 0) stamp(1);
        // LDS-DMA: each wave-instruction moves 64 slots (1 KiB) straight into LDS; the per-lane
        // SOURCE address comes from the gather table, the destination is lane-linear.  Zero fill
        // (conv padding, pitch padding, clips beyond the batch) reads a 16-byte zero slot.
        // (all table entries are consumed BEFORE the first DMA is issued: with an LDS-DMA in flight
        //  hipcc waits vmcnt(0) at the next use of an ordinary load, which would serialise the DMAs)
        if (hoist && !(VD_DBG(p) & 4)) {
            int gi = wave;
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                asm volatile("" : "+s"(gi));   // keep the per-group scalars (bound test, LDS address) out of SGPR-hungry hoisting
                if (gi < ngroups) {     // wave-uniform
                    const bool ok = goff[HOIST ? u : 0] != 0xFFFFFFFFu;
                    const uint32_t* gp = ok ? csrc + goff[HOIST ? u : 0] + (ph ? p.src_plane_stride4 : (int64_t)0) : zslot;
                    char* dst = smem + gi * 1024;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                    if constexpr (X3 && !SEQ) {
                        const uint32_t* gl = ok ? csrc + goff[HOIST ? u : 0] + p.src_plane_stride4 : zslot;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gl,
                                                         (__attribute__((address_space(3))) void*)(dst + plane_bytes), 16, 0, 0);
                    }
                }
                gi += nwaves;
            }
        }
        if (!hoist && !(VD_DBG(p) & 4)) {
            constexpr int LB = 8;
            for (int g0 = wave; g0 < ngroups; g0 += nwaves * LB) {
                uint32_t off[LB];
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    const int gi = g0 + u * nwaves;
                    const int e = gtab[((gi < ngroups) ? gi : ngroups - 1) * 64 + lane];
                    const int ci = e >> 24;
                    const bool ok = (e >= 0) && (clip0 + ci < p.nclips);
                    off[u] = ok ? ((uint32_t)ci * (uint32_t)p.src_clip_stride4 + (uint32_t)(e & 0xFFFFFF)) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < LB; ++u) asm volatile("" : "+v"(off[u]));   // all consumed before the first DMA
                int gi = g0;
#pragma unroll
                for (int u = 0; u < LB; ++u) {
                    asm volatile("" : "+s"(gi));
                    if (gi < ngroups) {
                        const bool ok = off[u] != 0xFFFFFFFFu;
                        const uint32_t* gp = ok ? csrc + off[u] + (ph ? p.src_plane_stride4 : (int64_t)0) : zslot;
                        char* dst = smem + gi * 1024;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                        if constexpr (X3 && !SEQ) {
                            const uint32_t* gl = ok ? csrc + off[u] + p.src_plane_stride4 : zslot;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gl,
                                                             (__attribute__((address_space(3))) void*)(dst + plane_bytes), 16, 0, 0);
                        }
                    }
                    gi += nwaves;
                }
            }
        }
        if (cc == 0 && ph == 0) stamp(2);
        __syncthreads();
        if (cc == 0 && ph == 0) stamp(3);
        // ---- K loop over tap pairs ---------------------------------------------------------
        if (VD_DBG(p) & 2) return;
        // B fragments are fetched DB steps ahead (counted vmcnt), A fragments one full step ahead
        // (x1: two register sets) or half a step ahead (x3: registers are short).
        if constexpr (!X3) {
            // tap offsets live in LDS and are read one step before the A reads that need them
            auto tap_of = [&](int st) {
                const int sc = (VD_DBG(p) & 32) ? 0 : ((st < S) ? st : S - 1);   // dbg 32: every step reads the same LDS rows
                return lds_tap[2 * sc + half];
            };
            // A fragments: AD + 1 = 2 register sets.  The set an M tile's MFMAs have just consumed is refilled AT ONCE with the
            // fragment of TWO steps ahead, tile by tile, so an LDS read has ~1.75 K steps of matrix work to return instead
            // of one (same registers, same K order per output, bitwise the same results): measured on the first layer, whose
            // waves are often alone on their SIMD (partner workgroup in its DMA / epilogue phase), -4.7 %.
            static_assert(AD == 1, "the refill scheme below is written for two register sets");
            uint4 A[AD + 1][MA];
#pragma unroll
            for (int d = 0; d <= AD; ++d) {
                const int tp = tap_of(d);
#pragma unroll
                for (int i = 0; i < MA; ++i) A[d][i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tp);
            }
            int tp = tap_of(AD + 1);          // tap offset of step s + 2
            VD_PRIO_LOOP(2);                  // (A/B switch: the whole K loop of a chunk above a partner workgroup's DMA / epilogue phases)
            auto k_step = [&](const int u, const int s_abs) {
                load_b(s_abs + DB, bqh[(u + DB) % (DB + 1)], bql[(u + DB) % (DB + 1)]);
                const int tp_next = tap_of(s_abs + AD + 2);
                VD_SCHED_BARRIER();
                VD_PRIO(1);
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    if (NTW == 2 && BAL == 0 && i == MTW - 1 && short_row) continue;
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[j * MTW + i] = mfma16<PREC>(A[u % (AD + 1)][i], bqh[u][j], acc[j * MTW + i]);
                    VD_SCHED_BARRIER();
                    if (i > 0 && (VD_DBG(p) & 64)) A[u % (AD + 1)][i] = A[u % (AD + 1)][0];                 // dbg 64: one LDS read per step
                    else A[u % (AD + 1)][i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tp);
                    VD_SCHED_BARRIER();
                }
                if constexpr (BAL) {   // seventh M tile: this wave's single N tile of it (wave row picks which)
                    uint4 bx = bqh[u][0];
                    if (wm) bx = bqh[u][1];
                    acc[MTW * NTW] = mfma16<PREC>(A[u % (AD + 1)][MTW], bx, acc[MTW * NTW]);
                    VD_SCHED_BARRIER();
                    A[u % (AD + 1)][MTW] = *reinterpret_cast<const uint4*>(smem + a_off[MTW] + tp);
                }
                VD_PRIO(0);
                VD_SCHED_BARRIER();
                tp = tp_next;
            };
            int s = 0;
            for (; s + DB < S; s += DB + 1) {   // full groups of DB+1 steps: no bound checks in the hot loop
#pragma unroll
                for (int u = 0; u <= DB; ++u) k_step(u, s + u);
            }
            if (s < S) {                      // remaining 1..DB steps
#pragma unroll
                for (int u = 0; u <= DB; ++u) {
                    if (s + u >= S) break;
                    k_step(u, s + u);
                }
            }
            VD_PRIO_LOOP(0);
        } else if constexpr (C8) {
            // fp16 main product every K step; the two correction products once per FOUR steps (a GROUP) on the fp8 instruction (K = 64 =
            // 4 steps x 2 taps x 8 channels).  Element j = 8 q + e of a lane's 32-byte fp8 fragment is (step 4 g + q, this lane's tap
            // half, channel e) in BOTH operands (vd_pack_weights_c8 packs B in that order; the instruction pairs element j of the A lane
            // (row, half) with element j of the B lane (column, half): tools/micro/mfma_f8_probe.hip).  Plane 1 of the patch holds, per
            // 16-byte slot, the 8 low parts (x 2^9) AND the 8-byte fp8 image of the high parts (a / 4), both written by the producing
            // program (emit_lo = 2): one ds_read_b128 per step of the group fetches both, and the loop converts nothing.
            //
            // TILE SKIP MASKS (position-tile programs, plan_forward_pos): bit i of a group's word = M tile i of this wave takes no part
            // in the group's four K steps -- none of its rows has those taps inside the input grid.  The planner orders a window's
            // tiles so that the masks it emits are 0 (all four tiles), 0b1100 (tiles 0, 1), 0b1010 (tiles 0, 2), 0b1110 (tile 0)
            // wherever the geometry allows (7 x 7: always); each has its own compile-time body, any other mask runs a generic one.
            // In the sparse bodies the A registers of the idle tiles hold the fragments of LATER steps of the busy ones, so that a
            // step never waits for the LDS read the previous step issued (one or two MFMAs do not cover that latency).
            const int* sc8 = reinterpret_cast<const int*>(p.out_scale);
            const int sb_hi = sc8[0], sb_lo = sc8[1];
            auto skip_of = [&](int st) { return __builtin_amdgcn_readfirstlane(lds_skip[((st < S) ? st : S - 4) >> 2]); };
            const int* tapv = lds_tap + half;                   // this lane half's tap offsets: tapv[2 * step]
            auto taps_of = [&](int st, int* t4) {                // the four tap offsets of the group that starts at step st (clamped)
                const int sc = (st < S) ? st : S - 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = tapv[2 * (sc + q)];
            };
            auto rdA = [&](int i, int tap) { return *reinterpret_cast<const uint4*>(smem + a_off[i] + tap); };
            const uint4* wq = wp;                               // B operands of the current group's first step
            auto piece = [&](int st) { return wq[(int64_t)st * wstep + w_lo]; };      // (reads up to 6 steps past the chunk: the buffer carries slack)
            int skip = skip_of(0);
            int tq[4];
            taps_of(0, tq);
            uint4 Ah[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                Ah[i] = make_uint4(0, 0, 0, 0);
                if (!((skip >> i) & 1)) Ah[i] = rdA(i, tq[0]);
            }
            // (the fp8 image of W_hi is not loaded: it is converted from each step's fp16 B fragment in registers -- same lane, same
            //  element order -- which takes a quarter off the bytes a wave pulls through its CU's vector-memory path, the busiest unit
            //  of this program: 69 % against the matrix pipes' 56 % with the image loaded)
            const float w_div = p.out_scale[3];          // 1 / s of vd_pack_weights_c8 (a power of two): image = W_hi * s
            uint4 b8_2, b8_3;
            i32x8 B8hi;
            auto img_b = [&](auto UC) __attribute__((always_inline)) {
                constexpr int u = decltype(UC)::v;
                int w0, w1;
                vd_c8_image(bqh[u][0], w0, w1, w_div);
                B8hi[2 * u] = w0; B8hi[2 * u + 1] = w1;
            };
            i32x8 a8lo[2], a8hi[2];
            auto corr_lo = [&](auto IC, auto RC) __attribute__((always_inline)) {
                constexpr int i = decltype(IC)::v, r = decltype(RC)::v;
                if (VD_DBG(p) & 0x800) return;          // dbg 0x800: no correction products
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8lo[r], B8hi, acc[i], 0, 0, 0, VD_C8_SA_LO, 0, sb_hi);
            };
            auto corr_hi = [&](auto IC, auto RC, const i32x8& B8lo) __attribute__((always_inline)) {
                constexpr int i = decltype(IC)::v, r = decltype(RC)::v;
                if (VD_DBG(p) & 0x800) return;
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8hi[r], B8lo, acc[i], 0, 0, 0, VD_C8_SA_HI, 0, sb_lo);
            };
            // plane 1 of tile i at the group's four taps -> ring entries rl (low parts) and rh (fp8 image of the high parts)
            auto rd_planes = [&](auto IC, auto RL, auto RH) __attribute__((always_inline)) {
                constexpr int i = decltype(IC)::v, rl = decltype(RL)::v, rh = decltype(RH)::v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint4 x = *reinterpret_cast<const uint4*>(smem + plane_bytes + a_off[i] + tq[q]);
                    a8lo[rl][2 * q] = (int)x.x; a8lo[rl][2 * q + 1] = (int)x.y;
                    a8hi[rh][2 * q] = (int)x.z; a8hi[rh][2 * q + 1] = (int)x.w;
                }
            };
            // one group = four K steps.  MK >= 0: compile-time mask; MK < 0: the run-time mask `skip`.
            auto group = [&](auto MC, const int skip_n, const int tqn0) __attribute__((always_inline)) {
                constexpr int MK = decltype(MC)::v;
                constexpr bool P4 = (MK == 0), P2A = (MK == 0xC), P2B = (MK == 0xA), P1 = (MK == 0xE);
                // ---- steps 0 .. 2 (and the main products of step 3) ----
                if constexpr (P1) {
                    // tile 0 only: its fragments of steps 1, 2, 3 go to the idle tiles' registers at once
                    Ah[1] = rdA(0, tq[1]); Ah[2] = rdA(0, tq[2]); Ah[3] = rdA(0, tq[3]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bqh[(u + DB) % (DB + 1)][0] = wq[(int64_t)(u + DB) * wstep];
                        if (u == 0) b8_2 = piece(2);
                        if (u == 1) b8_3 = piece(3);
                        if (u == 0) img_b(VdIC<0>{});
                        if (u == 1) img_b(VdIC<1>{});
                        if (u == 2) img_b(VdIC<2>{});
                        if (u == 3) img_b(VdIC<3>{});
                        __builtin_amdgcn_sched_barrier(0);
                        acc[0] = mfma16<PREC>(Ah[u], bqh[u][0], acc[0]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (P2A || P2B) {
                    // two tiles (0 and T1): the idle pair of registers is the second fragment set, reads run two steps ahead
                    constexpr int T1 = P2A ? 1 : 2, J0 = P2A ? 2 : 1, J1 = 3;
                    Ah[J0] = rdA(0, tq[1]); Ah[J1] = rdA(T1, tq[1]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bqh[(u + DB) % (DB + 1)][0] = wq[(int64_t)(u + DB) * wstep];
                        if (u == 0) b8_2 = piece(2);
                        if (u == 1) b8_3 = piece(3);
                        if (u == 0) img_b(VdIC<0>{});
                        if (u == 1) img_b(VdIC<1>{});
                        if (u == 2) img_b(VdIC<2>{});
                        if (u == 3) img_b(VdIC<3>{});
                        __builtin_amdgcn_sched_barrier(0);
                        const int ra = (u & 1) ? J0 : 0, rb = (u & 1) ? J1 : T1;
                        acc[0] = mfma16<PREC>(Ah[ra], bqh[u][0], acc[0]);
                        if (u < 2) Ah[ra] = rdA(0, tq[u + 2]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[T1] = mfma16<PREC>(Ah[rb], bqh[u][0], acc[T1]);
                        if (u < 2) Ah[rb] = rdA(T1, tq[u + 2]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bqh[(u + DB) % (DB + 1)][0] = wq[(int64_t)(u + DB) * wstep];
                        if (u == 0) b8_2 = piece(2);
                        if (u == 1) b8_3 = piece(3);
                        if (u == 0) img_b(VdIC<0>{});
                        if (u == 1) img_b(VdIC<1>{});
                        if (u == 2) img_b(VdIC<2>{});
                        if (u == 3) img_b(VdIC<3>{});
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < MTW; ++i) {
                            if (P4 || !((skip >> i) & 1)) {
                                acc[i] = mfma16<PREC>(Ah[i], bqh[u][0], acc[i]);
                                __builtin_amdgcn_sched_barrier(0);
                                if (u < 3) Ah[i] = rdA(i, tq[u + 1]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
                // ---- corrections of the group + the next group's first fragments ----
                const i32x8 B8lo = {(int)b8_2.x, (int)b8_2.y, (int)b8_2.z, (int)b8_2.w, (int)b8_3.x, (int)b8_3.y, (int)b8_3.z, (int)b8_3.w};
                auto prefetch_next = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
                        if (!((skip_n >> i) & 1)) Ah[i] = rdA(i, tqn0);
                };
                if constexpr (P4) {
                    // main products done; a tile's plane-1 slots are read one tile ahead of its first correction, the two corrections of an
                    // accumulator are a tile apart, so that consecutive writes of one accumulator are separated by other matrix instructions
                    rd_planes(VdIC<0>{}, VdIC<0>{}, VdIC<0>{});
                    rd_planes(VdIC<1>{}, VdIC<1>{}, VdIC<1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    corr_lo(VdIC<0>{}, VdIC<0>{});
                    prefetch_next();
                    __builtin_amdgcn_sched_barrier(0);
                    corr_lo(VdIC<1>{}, VdIC<1>{});
                    corr_hi(VdIC<0>{}, VdIC<0>{}, B8lo);
                    __builtin_amdgcn_sched_barrier(0);
                    rd_planes(VdIC<2>{}, VdIC<0>{}, VdIC<0>{});
                    __builtin_amdgcn_sched_barrier(0);
                    corr_hi(VdIC<1>{}, VdIC<1>{}, B8lo);
                    __builtin_amdgcn_sched_barrier(0);
                    rd_planes(VdIC<3>{}, VdIC<1>{}, VdIC<1>{});
                    __builtin_amdgcn_sched_barrier(0);
                    corr_lo(VdIC<2>{}, VdIC<0>{});
                    corr_lo(VdIC<3>{}, VdIC<1>{});
                    corr_hi(VdIC<2>{}, VdIC<0>{}, B8lo);
                    corr_hi(VdIC<3>{}, VdIC<1>{}, B8lo);
                } else if constexpr (P2A || P2B) {
                    constexpr int T1 = P2A ? 1 : 2;
                    rd_planes(VdIC<0>{}, VdIC<0>{}, VdIC<0>{});
                    rd_planes(VdIC<T1>{}, VdIC<1>{}, VdIC<1>{});
                    prefetch_next();
                    __builtin_amdgcn_sched_barrier(0);
                    corr_lo(VdIC<0>{}, VdIC<0>{});
                    corr_lo(VdIC<T1>{}, VdIC<1>{});
                    corr_hi(VdIC<0>{}, VdIC<0>{}, B8lo);
                    corr_hi(VdIC<T1>{}, VdIC<1>{}, B8lo);
                } else if constexpr (P1) {
                    rd_planes(VdIC<0>{}, VdIC<0>{}, VdIC<0>{});
                    prefetch_next();
                    __builtin_amdgcn_sched_barrier(0);
                    corr_lo(VdIC<0>{}, VdIC<0>{});
                    corr_hi(VdIC<0>{}, VdIC<0>{}, B8lo);
                } else {
                    vd_static_for<MTW>([&](auto IC) __attribute__((always_inline)) {
                        constexpr int i = decltype(IC)::v;
                        if (!((skip >> i) & 1)) {
                            rd_planes(IC, VdIC<i & 1>{}, VdIC<i & 1>{});
                            corr_lo(IC, VdIC<i & 1>{});
                            corr_hi(IC, VdIC<i & 1>{}, B8lo);
                        }
                    });
                    prefetch_next();
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // groups of one mask are consecutive (the planner sorts them): one loop per specialised body
            int s = 0;
            auto run = [&](auto MC) __attribute__((always_inline)) {
                constexpr int MK = decltype(MC)::v;
                const int skip_n = skip_of(s + 4);           // the NEXT group's mask and first tap: this group's last step prefetches for it
                const int tqn0 = tapv[2 * ((s + 4 < S) ? s + 4 : S - 4)];
                if (MK >= 0 || skip != 0xF) group(MC, skip_n, tqn0);
                else {                                      // nothing to do in this group (padding of a window with fewer groups): keep the rings going
#pragma unroll
                    for (int u = 0; u < 4; ++u) bqh[(u + DB) % (DB + 1)][0] = wq[(int64_t)(u + DB) * wstep];
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
                        if (!((skip_n >> i) & 1)) Ah[i] = rdA(i, tqn0);
                }
                wq += (int64_t)4 * wstep;
                skip = skip_n;
                s += 4;
                taps_of(s, tq);
            };
            while (s < S) {
                while (s < S && skip == 0) run(VdIC<0>{});
                while (s < S && skip == 0xA) run(VdIC<0xA>{});
                while (s < S && skip == 0xC) run(VdIC<0xC>{});
                while (s < S && skip == 0xE) run(VdIC<0xE>{});
                if (s < S && skip != 0 && skip != 0xA && skip != 0xC && skip != 0xE) run(VdIC<-1>{});
            }
        } else if constexpr (SEQ) {
            // hi+lo formats, one plane resident: the fragment a tile's MFMAs have just consumed is refilled at once with the
            // next K step's (same idea as below).  Pass 0: A_hi x B_lo and A_hi x B_hi, pass 1: A_lo x B_hi.
            {
                constexpr int PH = ph;
                int tap_next = lds_tap[2 * ((1 < S) ? 1 : 0) + half];
                uint4 Ax[MTW];
                {
                    const int tap0 = lds_tap[half];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) Ax[i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tap0);
                }
                for (int s = 0; s < S; s += DB + 1) {
#pragma unroll
                    for (int u = 0; u <= DB; ++u) {
                        if (s + u >= S) break;
                        load_b(s + u + DB, bqh[(u + DB) % (DB + 1)], bql[(u + DB) % (DB + 1)]);
                        uint4 bh[NTW], bl[NTW];
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            bh[j] = bqh[u][j]; bl[j] = bql[u][j];
                            bh[j].x ^= bsgn; bh[j].y ^= bsgn; bh[j].z ^= bsgn; bh[j].w ^= bsgn;
                            if constexpr (PH == 0) { bl[j].x ^= bsgn; bl[j].y ^= bsgn; bl[j].z ^= bsgn; bl[j].w ^= bsgn; }
                        }
                        const int sn2 = (s + u + 2 < S) ? s + u + 2 : S - 1;
                        const int tap_next2 = lds_tap[2 * sn2 + half];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < MTW; ++i) {
                            if (NTW == 2 && BAL == 0 && i == MTW - 1 && short_row) continue;
#pragma unroll
                            for (int j = 0; j < NTW; ++j) {
                                if constexpr (PH == 0) {
                                    acc[j * MTW + i] = mfma16<PREC>(Ax[i], bl[j], acc[j * MTW + i]);
                                    acc[j * MTW + i] = mfma16<PREC>(Ax[i], bh[j], acc[j * MTW + i]);
                                } else {
                                    acc[j * MTW + i] = mfma16<PREC>(Ax[i], bh[j], acc[j * MTW + i]);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            Ax[i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tap_next);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        tap_next = tap_next2;
                    }
                }
            }
        } else {
            // hi+lo formats: one register set of A fragments (hi and lo plane); the pair a tile's three MFMAs have just
            // consumed is refilled at once with the next K step's, so a read has the other tiles' MFMAs (almost a whole step)
            // to return -- the same idea as in the single-pass loop, same K order per output
            int tap_next = lds_tap[2 * ((1 < S) ? 1 : 0) + half];
            uint4 Ah[MTW], Al[MTW];
            {
                const int tap0 = lds_tap[half];
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    Ah[i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tap0);
                    Al[i] = *reinterpret_cast<const uint4*>(smem + plane_bytes + a_off[i] + tap0);
                }
            }
            for (int s = 0; s < S; s += DB + 1) {
#pragma unroll
                for (int u = 0; u <= DB; ++u) {
                    if (s + u >= S) break;
                    load_b(s + u + DB, bqh[(u + DB) % (DB + 1)], bql[(u + DB) % (DB + 1)]);
                    uint4 bh[NTW], bl[NTW];
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        bh[j] = bqh[u][j]; bl[j] = bql[u][j];
                        bh[j].x ^= bsgn; bh[j].y ^= bsgn; bh[j].z ^= bsgn; bh[j].w ^= bsgn;
                        bl[j].x ^= bsgn; bl[j].y ^= bsgn; bl[j].z ^= bsgn; bl[j].w ^= bsgn;
                    }
                    const int sn2 = (s + u + 2 < S) ? s + u + 2 : S - 1;
                    const int tap_next2 = lds_tap[2 * sn2 + half];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MTW; ++i) {
                        if (NTW == 2 && BAL == 0 && i == MTW - 1 && short_row) continue;
#pragma unroll
                        for (int j = 0; j < NTW; ++j) {
                            acc[j * MTW + i] = mfma16<PREC>(Al[i], bh[j], acc[j * MTW + i]);
                            acc[j * MTW + i] = mfma16<PREC>(Ah[i], bl[j], acc[j * MTW + i]);
                            acc[j * MTW + i] = mfma16<PREC>(Ah[i], bh[j], acc[j * MTW + i]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        Ah[i] = *reinterpret_cast<const uint4*>(smem + a_off[i] + tap_next);
                        Al[i] = *reinterpret_cast<const uint4*>(smem + plane_bytes + a_off[i] + tap_next);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    tap_next = tap_next2;
                }
            }
        }
      });
    }

    if constexpr (ALT) {            // an even number of chunks leaves -sum in the accumulators
        if (!(p.persist & 0x100000) && !(p.CC & 1)) {
#pragma unroll
            for (int i = 0; i < TILES; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[i][k] = -acc[i][k];
        }
    }
    // ---- epilogue ---------------------------------------------------------------------
    stamp(4);
    if (VD_DBG(p) & 1) { if (acc[0][0] == 123.456f) reinterpret_cast<float*>(p.dst)[0] = 1.f; continue; }
    const int64_t out_total = (int64_t)p.nclips * p.out_clip_stride;
    const int64_t out_base = (int64_t)clip0 * p.out_clip_stride + out_rel;
    // fp16 hi+lo programs over packed WEIGHTS: vd_pack_weights* stored W x 2^VD_F16X3_WSHIFT (vd_hip.h), undone here (exact)
    const float asc = (PREC == VD_PREC_F16X3 && p.w_box_stride == 0) ? 1.f / (float)(1 << VD_F16X3_WSHIFT) : 1.f;

    if (p.epi == VD_EPI_ROWS) {
        float* dst = reinterpret_cast<float*>(p.dst);
        if constexpr (EXT) { if (p.atomic) dst += (int64_t)box[5] * p.replica_stride; }   // this box's copy of the accumulation target
        const float osc = ((p.out_scale != nullptr) ? p.out_scale[0] : 1.f) * asc;
#pragma unroll
      for (int j = 0; j < NTW + BAL; ++j) {
        const bool ex = BAL && j == NTW;                     // the extra (seventh-tile) accumulator of a balanced wave
        const int n = (wn * NTW + (ex ? wm : j)) * 32 + (lane & 31);
        const bool n_ok = n < p.n_out;
        const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
        const int64_t coff = (p.col_off != nullptr) ? (int64_t)p.col_off[n & 31] : (int64_t)n * p.n_stride;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            if (ex && i > 0) continue;
            const int gi = ex ? p.MW * MTW : wm * MTW + i;
            const f32x16& at = acc[ex ? MTW * NTW : j * MTW + i];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int row = (k & 3) + 8 * (k >> 2) + 4 * half;
                const int o = o_tab[gi * 32 + row];
                const int64_t idx = out_base + o;
                float v = at[k] * osc + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (o >= 0 && n_ok && idx < out_total) {
                    if (EXT && p.atomic) atomicAdd(&dst[idx + coff], v);   // several boxes add into the same rows (wgrad)
                    else dst[idx + coff] = v;
                }
            }
        }
      }
        finish();
        continue;
    }

    if constexpr (C8) {
        if (p.epi == VD_EPI_POS_FEAT) {
            // position tiles: the wave's four accumulator tiles are the four positions of ONE pool window, rows 2 e and 2 e + 1 (two
            // registers of one lane) a frame pair of one clip: the 2 x 2 x 2 pool is a max over 8 registers, no lane traffic
            float* dstf = reinterpret_cast<float*>(p.dst);
            const int n = wn * 32 + (lane & 31);
            const bool n_ok = n < p.n_out;
            const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                float m = fmaxf(acc[0][k], acc[0][k + 1]);
#pragma unroll
                for (int i = 1; i < MTW; ++i) m = fmaxf(m, fmaxf(acc[i][k], acc[i][k + 1]));
                m += bias;
                if (p.relu) m = fmaxf(m, 0.f);
                const int row = (k & 3) + 8 * (k >> 2) + 4 * half;
                const int o = lds_otab[row >> 1];
                const int64_t idx = out_base + o + (int64_t)n * p.n_stride;
                if (o >= 0 && n_ok && idx < out_total) dstf[idx] = m;
            }
            finish();
            continue;
        }
    }

    // pooled epilogues: registers 8*qh .. 8*qh+7 of this lane are one 2x2x2 row group.
    // max first (bias is per lane and ReLU monotone: relu(max_j(a_j) + b) == max_j relu(a_j + b)),
    // 32-bit per-lane offsets on top of a 64-bit per-workgroup base.
    const int64_t lim64 = out_total - out_base;
    const int lim = (int)(lim64 > 0x7fffffff ? 0x7fffffff : (lim64 < 0 ? 0 : lim64));
    const bool feat = (p.epi == VD_EPI_POOL_FEAT);
    float* dstf = reinterpret_cast<float*>(p.dst) + out_base;
    uint16_t* dst16 = reinterpret_cast<uint16_t*>(p.dst) + out_base * 8;
    uint8_t* amx = p.argmax ? p.argmax + (feat ? out_base : out_base * 8) : nullptr;
    const int nsets = (p.pool_t == 2) ? 1 : 2;
    // Channels-last outputs without arg-max (the real-clip path): stage the pooled tile through LDS
    // and write whole 16-byte slots.  (Per-lane 2-byte stores cost one memory instruction per value
    // and made the epilogue as expensive as the K loop of the first layer.)
    const bool staged = !feat && amx == nullptr;
    // (single-pass programs may be asked for the low plane of their pooled outputs as well: VdConvParams.emit_lo)
    const bool lo_out = X3 || (!EXT && p.emit_lo != 0);
    const int NCH = p.NT * 32;
    const int Q = mt_tot * 4 * nsets;
    uint16_t* stg = reinterpret_cast<uint16_t*>(smem);
    if (staged) __syncthreads();          // every wave is done reading the patch
    stamp(5);
    float rs_amax = 0.f;
    int rs_sat = 0;
    // All register indices below are compile-time constants (a run-time loop bound over acc[]
    // would turn into select chains): both 4-row halves are reduced unconditionally, pool_t only
    // decides whether they are merged.
#pragma unroll
  for (int j = 0; j < NTW + BAL; ++j) {
    const bool ex = BAL && j == NTW;
    const int n = (wn * NTW + (ex ? wm : j)) * 32 + (lane & 31);
    const bool n_ok = n < p.n_out;
    const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
    const uint32_t chan = feat ? (uint32_t)n * (uint32_t)p.n_stride
                               : (uint32_t)(n >> 3) * (uint32_t)p.out_chunk_stride * 8u + (uint32_t)(n & 7);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        if (ex && i > 0) continue;
        const int gi = ex ? p.MW * MTW : wm * MTW + i;
        const f32x16& at = acc[ex ? MTW * NTW : j * MTW + i];
#pragma unroll
        for (int qh = 0; qh < 2; ++qh) {
            const int r0 = 8 * qh;
            if (staged) {
                float m0 = fmaxf(fmaxf(at[r0], at[r0 + 1]), fmaxf(at[r0 + 2], at[r0 + 3]));
                float m1 = fmaxf(fmaxf(at[r0 + 4], at[r0 + 5]), fmaxf(at[r0 + 6], at[r0 + 7]));
                if (p.pool_t == 2) m0 = fmaxf(m0, m1);
                if constexpr (PREC == VD_PREC_F16X3) { m0 *= asc; m1 *= asc; }
                m0 += bias; m1 += bias;
                if (p.relu) { m0 = fmaxf(m0, 0.f); m1 = fmaxf(m1, 0.f); }
                if (!X3 && !EXT && p.emit_lo == 2) {      // the fp8 image of these values (x 1/4) must stay finite for the consumer: clamp at 1792
                    const float a0 = fabsf(m0), a1 = (p.pool_t == 2) ? 0.f : fabsf(m1);      // (range monitor: VdConvParams.range_stats)
                    rs_amax = fmaxf(rs_amax, fmaxf(a0, a1));
                    rs_sat += (!(a0 <= 1792.f) ? 1 : 0) + (!(a1 <= 1792.f) ? 1 : 0);      // (NaN counts as saturated: fmaxf drops it from the maximum)
                    m0 = fminf(fmaxf(m0, -1792.f), 1792.f); m1 = fminf(fmaxf(m1, -1792.f), 1792.f);
                }
                const int q = (gi * 4 + half + 2 * qh) * nsets;
                uint16_t hi, lo;
                split16<PREC>(m0, hi, lo);
                stg[q * NCH + n] = hi;
                // (emit_lo = 2: plane 1 for a VD_PREC_F16C8 consumer -- per 16-byte slot the 8 low parts as e4m3 bytes, x 2^9, then the e4m3 image of the 8 high parts, / 4)
                const bool lo8 = !X3 && !EXT && p.emit_lo == 2;
                uint8_t* stg8 = reinterpret_cast<uint8_t*>(stg);
                const int b8 = (n >> 3) * 16 + (n & 7);
                if (lo8) { stg8[(Q + q) * NCH * 2 + b8] = (uint8_t)vd_c8_lo_byte(m0); stg8[(Q + q) * NCH * 2 + b8 + 8] = (uint8_t)vd_c8_hi_byte(hi); }
                else if (lo_out) stg[(Q + q) * NCH + n] = lo;
                if (p.pool_t != 2) {
                    split16<PREC>(m1, hi, lo);
                    stg[(q + 1) * NCH + n] = hi;
                    if (lo8) { stg8[(Q + q + 1) * NCH * 2 + b8] = (uint8_t)vd_c8_lo_byte(m1); stg8[(Q + q + 1) * NCH * 2 + b8 + 8] = (uint8_t)vd_c8_hi_byte(hi); }
                    else if (lo_out) stg[(Q + q + 1) * NCH + n] = lo;
                }
                continue;
            }
            const int o = lds_otab[gi * 4 + half + 2 * qh];
            if (o < 0 || !n_ok) continue;
            if (SO && p.select) {
                const float osc_sel = ((p.out_scale != nullptr) ? p.out_scale[0] : 1.f) * asc;
                // the arg-max bytes are an INPUT: emit the accumulator row a previous forward selected
                // (0 where that forward's ReLU was dead) -- the adjoint of vd_unpool_relu_bwd
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    if (st == 1 && p.pool_t == 2) continue;
                    const int base = o + st * (((p.pair_flip >> (half + 2 * qh)) & 1) ? -p.out_t_stride : p.out_t_stride);
                    if (base >= lim) continue;
                    const uint32_t idx = feat ? (uint32_t)base + chan : (uint32_t)base * 8u + chan;
                    const int ab = amx[idx];
                    const int jsel = (p.pool_t == 2) ? (ab & 7) : (4 * st + (ab & 3));
                    float v = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v = (j == jsel) ? at[r0 + j] : v;
                    v = (ab & 0x80) ? 0.f : v * osc_sel + bias;
                    if (feat) {
                        dstf[idx] = v;
                    } else if (p.select == 2) {      // fp32 in the slot order: the caller measures the range, then splits (vd_split_scaled)
                        reinterpret_cast<float*>(p.dst)[out_base * 8 + idx] = v;
                    } else {
                        uint16_t hi, lo;
                        split16<PREC>(v, hi, lo);
                        dst16[idx] = hi;
                        if constexpr (X3) dst16[idx + p.dst_plane_stride * 8] = lo;
                    }
                }
                continue;
            }
            float mv[2]; int av[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float mx = at[r0 + 4 * st];
                int am = 0;
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    const float v = at[r0 + 4 * st + j];
                    if (v > mx) { mx = v; am = j; }
                }
                mv[st] = mx; av[st] = am;
            }
            if (p.pool_t == 2 && mv[1] > mv[0]) { mv[0] = mv[1]; av[0] = 4 + av[1]; }   // first maximum wins ties
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                if (st == 1 && p.pool_t == 2) continue;
                float mx = mv[st];
                if constexpr (PREC == VD_PREC_F16X3) mx *= asc;
                mx += bias;
                if (p.relu) mx = fmaxf(mx, 0.f);
                const int base = o + st * (((p.pair_flip >> (half + 2 * qh)) & 1) ? -p.out_t_stride : p.out_t_stride);
                if (base >= lim) continue;
                const uint8_t ab = (uint8_t)(av[st] | (mx > 0.f ? 0 : 0x80));
                if (feat) {
                    const uint32_t idx = (uint32_t)base + chan;
                    dstf[idx] = mx;
                    if (amx) amx[idx] = ab;
                } else {
                    const uint32_t idx = (uint32_t)base * 8u + chan;
                    uint16_t hi, lo;
                    split16<PREC>(mx, hi, lo);
                    dst16[idx] = hi;
                    if constexpr (X3) dst16[idx + p.dst_plane_stride * 8] = lo;
                    if (amx) amx[idx] = ab;
                }
            }
        }
    }
  }   // N tiles of this wave
    if constexpr (!X3 && !EXT) {
        if (staged && p.emit_lo == 2 && p.range_stats != nullptr) {      // one wave's share of the launch's output range (rare atomics:
#pragma unroll                                                            //  the maximum converges after a few boxes, saturation is an error)
            for (int o = 32; o > 0; o >>= 1) {
                rs_amax = fmaxf(rs_amax, __shfl_xor(rs_amax, o));
                rs_sat += __shfl_xor(rs_sat, o);
            }
            if (lane == 0) {
                const uint32_t bits = __float_as_uint(rs_amax);
                if (bits > __builtin_nontemporal_load(p.range_stats + 1)) atomicMax(p.range_stats + 1, bits);
                if (rs_sat) atomicAdd(p.range_stats, (uint32_t)rs_sat);
            }
        }
    }
    if (staged) {
        __syncthreads();
        stamp(6);
        const int cpq_sh = __builtin_ctz(NCH >> 3);        // 16-byte slots per pooled position (4, 8 or 16)
        const int ns_sh = nsets - 1;
        uint4* dslots = reinterpret_cast<uint4*>(p.dst) + out_base;
        for (int item = tid; item < ((Q << cpq_sh)); item += nthreads) {
            const int q = item >> cpq_sh, cc = item & ((1 << cpq_sh) - 1);
            const int g = q >> ns_sh, st = q & ns_sh;
            const int o = lds_otab[g];
            const int base = o + st * (((p.pair_flip >> (g & 3)) & 1) ? -p.out_t_stride : p.out_t_stride);       // (frame-tile programs: vd_hip.h)
            if (o < 0 || base >= lim) continue;
            const uint32_t slot = (uint32_t)base + (uint32_t)cc * (uint32_t)p.out_chunk_stride;
            dslots[slot] = *reinterpret_cast<const uint4*>(stg + q * NCH + cc * 8);
            if (lo_out) dslots[slot + p.dst_plane_stride] = *reinterpret_cast<const uint4*>(stg + (Q + q) * NCH + cc * 8);
        }
    }
    finish();
  }   // box loop
}

template <int PREC, int MTW, bool SO = false, int NTW = 1, int BAL = 0, bool SQ = false>
__global__ __launch_bounds__(256, VD_OCC(PREC, MTW, NTW, BAL)) void conv_mfma_kernel(const VdConvParams p, const int boxes_per_wg, const int total_boxes) {
    conv_mfma_body<PREC, MTW, SO, NTW, BAL, SQ>(p, boxes_per_wg, total_boxes, (int)blockIdx.x, (int)gridDim.x);
}

// Several tile programs of ONE instantiation in one launch: the parity classes of an input-gradient pass (4 programs per level,
// each too small to fill the chip at small batches: 64 workgroups for a 256-clip batch of 64x64x8 clips at the last level) go out
// together; workgroup b runs box b - first[k] of program k.  The parameter blocks travel in the kernel arguments (scalar loads
// with a wave-uniform index), results are bitwise those of the separate launches.
#define VD_MULTI_MAX 4
struct VdConvMulti {
    VdConvParams p[VD_MULTI_MAX];
    int first[VD_MULTI_MAX + 1];       // first[k] .. first[k+1]: the workgroups of program k
    int total[VD_MULTI_MAX];           // boxes of program k
};
template <int PREC, int MTW, bool SO = false, int NTW = 1, int BAL = 0>
__global__ __launch_bounds__(256, VD_OCC(PREC, MTW, NTW, BAL)) void conv_mfma_multi_kernel(const VdConvMulti m) {
    const int b = (int)blockIdx.x;
    int k = 0;
#pragma unroll
    for (int i = 1; i < VD_MULTI_MAX; ++i) k += (b >= m.first[i]) ? 1 : 0;
    conv_mfma_body<PREC, MTW, SO, NTW, BAL>(m.p[k], 1, m.total[k], b - m.first[k], m.first[k + 1] - m.first[k]);
}

// tuning switches of the first-level kernel (A/B builds: tools/l0_variants.sh)
#ifndef VD_L0_D
#define VD_L0_D 6          // A fragments in flight in the frame-sharing K loop
#endif
#ifndef VD_L0_PRIO
#define VD_L0_PRIO 1       // s_setprio 3 around the K loop: the partner wave on the SIMD is in its vector-ALU phase (same box: 8.53 -> 8.33 ms)
#endif
#ifndef VD_L0_TOUCH
#define VD_L0_TOUCH 0      // 1: touch the next box's rows from inside the K loop (one dword per row half into a register nobody reads, so that
                           // the next phase's row loads hit the caches).  Measured SLOWER on the same box, 8.53 vs 7.90 ms per 3200 clips
                           // (profiles/r05_l0_variants.txt): with the box scalars prefetched and the stores moved behind the expansion the
                           // other phase is the shorter one, and everything added to the K phase lengthens the critical one
#endif
#define VD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")      // (no vmcnt wait: global loads stay in flight)

// The first level's single-pass kernel (3 -> 64 channels, K = 512: the layer's B fragments live in 128 registers per wave for a whole
// box walk).  ONE workgroup of eight waves per CU, two groups of four that alternate roles box by box -- while group 0 runs
// the K loop of its box (matrix pipes), group 1 finishes its previous box and prepares its next one (pool, stage, kw-slot
// expansion, output slots: vector ALU + LDS), then they swap; one s_barrier per phase, none inside a phase.
// (Rounds 1-3 ran two independent four-wave workgroups per CU, which drift into lockstep -- both in their K loops, then both out
// of them: 3.5 k cycles per box with idle matrix pipes; docs/history.md.)  A wave of group 0 and one of group 1 share each SIMD.
// Same tile program, same K order per output as the generic kernel: bitwise its results.
//
// FS (round 5, frame-tile programs: VdConvParams.pair_flip != 0, LDS pitches 9 / 189 slots -- the only ones a 4 x 8 x 8 box gets):
//  * the four M tiles of a wave are the SAME 32 positions in four consecutive frames, and K step 3 j + kt multiplies tap pair j of
//    kernel plane kt.  The A fragment of (frame f, pair j) is then the operand of tile f at kt = 0, tile f - 1 at kt = 1 and tile
//    f - 2 at kt = 2: it is read from LDS ONCE and feeds up to three MFMAs -- 60 + 8 reads per box and wave instead of 128 (with
//    one ds_read_b128 per MFMA the LDS is exactly as busy as the matrix pipes: 512 KB per box at 128 bytes per clock = the box's
//    4 096 matrix cycles).  Steps 30 / 31 (the three left-over taps of the kt planes and the zero tap) are read per tile;
//  * every tap offset is a compile-time constant that rides in the ds_read's offset field (two address registers per wave: the
//    lane halves' taps are 144 bytes apart except in the pair that straddles two channels);
//  * the rows of a group's next box are TOUCHED in front of its K loop (one dword per row half into a register nobody reads) and
//    travel from HBM into the caches under it; the row loads of the next phase hit there -- their HBM latency (1.5 - 2.5 k cycles,
//    tools/stamps_l0.py) used to sit inside that phase -- and the box scalars they need (table row, clip index) are s_loads of
//    the phase before.
template <int PREC, bool FS>
__global__ __launch_bounds__(512, 1) void conv0_breg_kernel(const VdConvParams p, const int boxes_per_wg) {
    constexpr int MTW = 4, S = 32, NI = 3;
    constexpr int PH = 9, PF = 189;                       // FS: LDS pitches in slots (checked by the launcher)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, tg = tid & 255;          // role group, thread index within the group
    const int wn = wave & 1, wm = (wave >> 1) & 1;
    const int half = lane >> 5;
    const int32_t* a_tab = p.tables + p.tab_ofs[0];
    const int32_t* o_tab = p.tables + p.tab_ofs[1];
    const int32_t* t_tab = p.tables + p.tab_ofs[2];
    const int plane_bytes = p.lds_plane_bytes;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.src);
    char* patch = smem + grp * plane_bytes;              // this group's patch
    int* lds_tap = reinterpret_cast<int*>(smem + 2 * plane_bytes);
    uint16_t* stg16 = reinterpret_cast<uint16_t*>(smem + 2 * plane_bytes + 512 + wave * 2048);
    for (int k = tid; k < 2 * S; k += 512) lds_tap[k] = t_tab[k];
    int a_off[FS ? 1 : MTW];
#pragma unroll
    for (int i = 0; i < (FS ? 1 : MTW); ++i) a_off[i] = a_tab[(wm * MTW + i) * 32 + (lane & 31)];
    int o_reg[2];          // output slot of this lane's two staged rows (item = lane + 64 k: pooled row item >> 2), -1 = none
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ql = (lane + 64 * k) >> 2, g = ql >> 1;
        const int o = o_tab[wm * 16 + g];
        const int step = ((p.pair_flip >> (g & 3)) & 1) ? -p.out_t_stride : p.out_t_stride;
        o_reg[k] = o < 0 ? -1 : o + (ql & 1) * step;
    }
    const int ph = p.type_desc[1], pitch_h = FS ? PH : p.type_desc[3], pitch_f = FS ? PF : p.type_desc[4], nitems = 2 * p.type_desc[0] * ph;
    const int ph_magic = (65536 + ph - 1) / ph;
    const int row4 = (int)(p.src_chunk_stride4 / ((int64_t)p.src_planes * p.src_rows));
    u32x4 breg[S];
    // (inline assembly, landed where they are issued: loads the compiler tracks get their s_waitcnt vmcnt(0) at the first use -- inside
    //  the K loop, on every pass, although the operand set changes once in hundreds of boxes -- and there it would also wait for the
    //  row-touch loads issued in front of the loop)
    // (round 6: loads AND their wait are one asm block per 16 fragments, outputs early-clobber -- a value the block defines has
    //  landed when the block ends.  With the wait in a separate, operand-less asm the compiler was free to copy or spill a
    //  fragment between its load and the wait, i.e. to read registers whose load was still in flight; nothing but the bitwise test
    //  against the generic kernel stood against that.  The second block issues after the first one's wait: one more memory
    //  latency per operand-set change, once in hundreds of boxes.)
    auto load_breg = [&](int set) {
        const char* wbase = reinterpret_cast<const char*>(reinterpret_cast<const uint4*>(p.wpk) + (int64_t)set * (p.w_plane_stride >> 3) + (int64_t)wn * 64);
        uint32_t voff = (uint32_t)lane * 16u;
#define VD_L0_LD(k) "global_load_dwordx4 %" #k ", %16, %17\n\tv_add_u32 %16, 0x800, %16\n\t"
#define VD_L0_LD16 VD_L0_LD(0) VD_L0_LD(1) VD_L0_LD(2) VD_L0_LD(3) VD_L0_LD(4) VD_L0_LD(5) VD_L0_LD(6) VD_L0_LD(7) \
                   VD_L0_LD(8) VD_L0_LD(9) VD_L0_LD(10) VD_L0_LD(11) VD_L0_LD(12) VD_L0_LD(13) VD_L0_LD(14) VD_L0_LD(15)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4* b = breg + 16 * h;
            asm volatile(VD_L0_LD16 "s_waitcnt vmcnt(0)"
                         : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3]), "=&v"(b[4]), "=&v"(b[5]), "=&v"(b[6]), "=&v"(b[7]),
                           "=&v"(b[8]), "=&v"(b[9]), "=&v"(b[10]), "=&v"(b[11]), "=&v"(b[12]), "=&v"(b[13]), "=&v"(b[14]), "=&v"(b[15]),
                           "+v"(voff)
                         : "s"(wbase)
                         : "memory");
        }
#undef VD_L0_LD16
#undef VD_L0_LD
        __builtin_amdgcn_sched_barrier(0);
    };
    int cur_set = -1;
    const int n = wn * 32 + (lane & 31);
    const float bias = (p.bias != nullptr) ? p.bias[n] : 0.f;
    const int64_t out_total = (int64_t)p.nclips * p.out_clip_stride;
    const int total = p.nclips * p.nbox;            // ncl == 1
    int wgid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7;
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int b_lo = wgid * boxes_per_wg;
    const int b_hi = (b_lo + boxes_per_wg < total) ? b_lo + boxes_per_wg : total;
    if (b_lo >= b_hi) return;
    const int nmine = (b_hi - b_lo + 1 - grp) >> 1;      // boxes b_lo + grp, b_lo + grp + 2, ... of this group
    uint4 raw[NI][2];
    // Box scalars, prepared a phase ahead of their use (round 5): the table row of the box, its clip's entry of the index list and the
    // output origin are loads whose latencies -- two dependent ones -- used to sit in front of the row loads of every box.
    // Wave-uniform (SGPRs); read through the CONSTANT address space -- tables the kernel never writes: s_load, no vector-memory
    // counter involved.
    const char* rs_base = nullptr;      // source address of the patch origin of the box whose rows are loaded next
    int rs_f0 = 0, rs_h0 = 0, rs_w0 = 0;
    int64_t ep_base = 0;                // output slot origin of the box that is finished next
    typedef const int32_t __attribute__((address_space(4))) * cint_p;
    typedef const int64_t __attribute__((address_space(4))) * clong_p;
    const cint_p c_boxes = (cint_p)(uintptr_t)p.boxes;
    const clong_p c_index = (clong_p)(uintptr_t)p.clip_index;
    auto prep_rows = [&](int b) {
        const int clip0 = b / p.nbox, bi = b - clip0 * p.nbox;
        const cint_p box = c_boxes + bi * 8;
        const int b6 = box[6], b7 = box[7];
        const int64_t ci = p.clip_index != nullptr ? c_index[clip0] : (int64_t)clip0;
        rs_f0 = b6 >> 16; rs_h0 = (int)(int16_t)(b6 & 0xFFFF); rs_w0 = b7;      // plane / row / dword of the patch origin
        rs_base = reinterpret_cast<const char*>(src + ci * p.src_clip_stride4 + ((int64_t)rs_f0 * p.src_rows + rs_h0) * row4 + rs_w0);
    };
    auto prep_out = [&](int b) {
        const int clip0 = b / p.nbox, bi = b - clip0 * p.nbox;
        ep_base = (int64_t)clip0 * p.out_clip_stride + c_boxes[bi * 8 + 3];
    };
    auto load_raw = [&]() {            // the aligned chunks of the prepared box's patch rows -> registers (zeros outside the clip); branch free:
        const int f0 = rs_f0, h0 = rs_h0, w0 = rs_w0;             // rows outside the clip load a valid row of the box and are zeroed
        const char* base = rs_base;
        int tv = tg;
        asm volatile("" : "+v"(tv));       // (opaque per call: otherwise the row coordinates below are hoisted out of the box loop into
                                           //  registers the K loop does not have)
        const uint32_t off_valid = (uint32_t)((((f0 < 0 ? -f0 : 0) * p.src_rows + (h0 < 0 ? -h0 : 0)) * row4) * 4);   // first row of the box inside the clip
        const uint32_t hf16 = (uint32_t)(tv & 1) * 16u;
        const bool in_row = (w0 + 4 * (tv & 1) + 8 <= row4);                         // the second chunk stays inside the pixel row
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int it = tv + 256 * k, r = it >> 1;
            const int pl = (r * ph_magic) >> 16, h = r - pl * ph;
            const bool ok = it < nitems && (unsigned)(f0 + pl) < (unsigned)p.src_planes && (unsigned)(h0 + h) < (unsigned)p.src_rows;
            const uint32_t o1 = (ok ? (uint32_t)((pl * p.src_rows + h) * row4 * 4) : off_valid) + hf16;
            const uint32_t o2 = (ok && in_row) ? o1 + 16u : o1;
            // (masked with AND, not selected: a select lets the compiler sink the load under a branch, and a load that MAY have been
            //  issued makes every later s_waitcnt vmcnt conservative -- 0 -- which then also waits for the output stores)
            const uint4 v0 = *reinterpret_cast<const uint4*>(base + o1);
            const uint4 v1 = *reinterpret_cast<const uint4*>(base + o2);
            const uint32_t m0 = ok ? 0xffffffffu : 0u, m1 = (ok && in_row) ? 0xffffffffu : 0u;
            raw[k][0] = make_uint4(v0.x & m0, v0.y & m0, v0.z & m0, v0.w & m0);
            raw[k][1] = make_uint4(v1.x & m1, v1.y & m1, v1.z & m1, v1.w & m1);
        }
    };
    // FS: the same rows TOUCHED a phase earlier -- one dword per row half into a register nobody reads, issued in front of the group's
    // K loop: the lines travel from HBM to the L2 / vector L1 under the K loop (1.5 - 2.5 k cycles, tools/stamps_l0.py), and the loads
    // of the next phase find them there.  (Holding the rows themselves across the K loop takes 21 registers the loop does not have:
    // the compiler spills B fragments into the loop.)  Inline assembly: the compiler must not wait for these loads anywhere; they are
    // older than every load it tracks, so its own in-order s_waitcnt counts stay sufficient.
    uint32_t sink = 0;
    auto touch_rows = [&]() {
        const int f0 = rs_f0, h0 = rs_h0;
        const char* base = rs_base;
        int tv = tg;
        asm volatile("" : "+v"(tv));
        const uint32_t off_valid = (uint32_t)((((f0 < 0 ? -f0 : 0) * p.src_rows + (h0 < 0 ? -h0 : 0)) * row4) * 4);
        const uint32_t hf16 = (uint32_t)(tv & 1) * 16u;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int it = tv + 256 * k, r = it >> 1;
            const int pl = (r * ph_magic) >> 16, h = r - pl * ph;
            const bool ok = it < nitems && (unsigned)(f0 + pl) < (unsigned)p.src_planes && (unsigned)(h0 + h) < (unsigned)p.src_rows;
            const uint32_t o1 = (ok ? (uint32_t)((pl * p.src_rows + h) * row4 * 4) : off_valid) + hf16;
            asm volatile("global_load_dword %0, %1, %2" : "+v"(sink) : "v"(o1), "s"(base) : "memory");
        }
    };
    auto expand = [&]() {
        int tv = tg;
        asm volatile("" : "+v"(tv));
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int it = tv + 256 * k, r = it >> 1;
            if (it >= nitems) continue;
            const int pl = (r * ph_magic) >> 16, h = r - pl * ph;
            const uint4 a = raw[k][0], c = raw[k][1];
            uint4* dst = reinterpret_cast<uint4*>(patch + (pl * pitch_f + h * pitch_h + 4 * (tv & 1)) * 16);
            dst[0] = a;
            dst[1] = make_uint4(a.y, a.z, a.w, c.x);
            dst[2] = make_uint4(a.z, a.w, c.x, c.y);
            dst[3] = make_uint4(a.w, c.x, c.y, c.z);
        }
    };
    f32x16 acc[MTW];
    // dbg bit 3 (library built with VD_DBG_HOOKS): s_memtime sums per role phase of this wave, written by wave 0 of each group to
    // stamps[(workgroup * 2 + group) * 8 ..]: K phase up to the K loop, K loop, barrier after it, pool + stage, expand (FS: its rows
    // are in registers; else: row-load issue + wait + expand), slots out + scalars + barrier, boxes, whole walk (tools/stamps_l0.py;
    // every stamp is a scalar-memory round trip of its own: a few hundred cycles)
    unsigned long long t_sum[6] = {0, 0, 0, 0, 0, 0}, t_last = 0, t_first = 0;
    auto tick = [&](int k) {
        if (VD_DBG(p) & 8) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            if (k >= 0) t_sum[k] += t - t_last;
            else if (k == -2) t_first = t;
            t_last = t;
        }
    };
    // group 0 prepares its first box before the first phase; group 1 does so in phase 0, under group 0's first K loop
    const int b_first = (b_lo + grp < b_hi) ? b_lo + grp : b_hi - 1;
    prep_rows(b_first);
    load_raw();
    if (grp == 0) {
        expand();
        if (FS) prep_rows(b_first + 2 < b_hi ? b_first + 2 : b_first);      // (its first K phase loads the rows of its second box)
    }
    {
        const int set = (p.w_set_clips > 0) ? (b_first / p.nbox) / p.w_set_clips : 0;
        cur_set = set; load_breg(set);
    }
    __syncthreads();                     // tables published, group 0's first patch in place
    tick(-2);
    // phases 0 .. 2 * n0: in phase q group g runs a K loop when (q + g) is even, the other half of its work when odd
    const int n0 = (b_hi - b_lo + 1) >> 1;
    for (int q = 0; q <= 2 * n0; ++q) {
        const int j = (q - grp) >> 1;                    // index into this group's boxes (K loop: box j; other phase: finish j-1... see below)
        const bool k_phase = ((q + grp) & 1) == 0;
        if (k_phase) {
            // ---- K loop of this group's box j ----
            if (j >= 0 && j < nmine) {
                const int b = b_lo + grp + 2 * j;
                if (p.w_set_clips > 0) {
                    const int set = (b / p.nbox) / p.w_set_clips;
                    if (set != cur_set) { cur_set = set; load_breg(set); }
                }
                prep_out(b);                             // output origin of this box, for the next phase
                if constexpr (!FS) prep_rows(j + 1 < nmine ? b + 2 : b);
                tick(0);
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
                if constexpr (FS) {
                    // reads in the order r = 6 j + f (pair j = 0..9, frame f = 0..5), then the 8 per-tile reads of steps 30 / 31;
                    // a ring of D fragments is kept in flight.  Tap t of the K order (plan.plan_forward_pix) = (kt, c, kh) with
                    // q = 7 c + kh: pair j holds q = 2 j (lanes 0-31) and 2 j + 1 (lanes 32-63) -- 144 bytes apart (pa), except
                    // q = 6 / 7 which straddle two channels (pb); left-over taps: q = 20 of kt = 0 / 1 (step 30), of kt = 2 and the
                    // zero tap (step 31: the upper lanes read the lower lanes' address, their B operand is zero)
                    constexpr int D = VD_L0_D, NR = 68;
                    constexpr int FSTRIDE = 3 * PF * 16;
                    uint4 R[D];
                    const char* pa = patch + a_off[0] + half * (PH * 16);
                    const char* pb = patch + a_off[0] + half * ((PF - 6 * PH) * 16);
                    const char* pl0 = patch + a_off[0] + half * FSTRIDE;          // step 30: kt = 0 below, kt = 1 above
                    const char* pl1 = patch + a_off[0];                           // step 31
                    auto tapq = [](int qq) constexpr { return ((qq / 7) * PF + (qq % 7) * PH) * 16; };
                    auto rd = [&](auto rc) -> uint4 {
                        constexpr int r = decltype(rc)::v;
                        if constexpr (r < 60) {
                            constexpr int jj = r / 6, f = r % 6;
                            return *reinterpret_cast<const uint4*>((jj == 3 ? pb : pa) + f * FSTRIDE + tapq(2 * jj));
                        } else {
                            constexpr int i = (r - 60) & 3, st = (r - 60) >> 2;
                            return *reinterpret_cast<const uint4*>((st ? pl1 : pl0) + (i + 2 * st) * FSTRIDE + tapq(20));
                        }
                    };
                    vd_static_for<D>([&](auto rc) { R[decltype(rc)::v] = rd(rc); });
                    if (VD_L0_PRIO) __builtin_amdgcn_s_setprio(3);      // (the partner wave on this SIMD is in its vector-ALU phase)
                    vd_static_for<NR>([&](auto rc) {
                        constexpr int r = decltype(rc)::v;
                        if constexpr (r < 60) {
                            constexpr int jj = r / 6, f = r % 6;
                            vd_static_for<3>([&](auto kc) {
                                constexpr int kt = decltype(kc)::v, i = f - kt;
                                if constexpr (i >= 0 && i < MTW) {
                                    acc[i] = mfma16<PREC>(R[r % D], breg[3 * jj + kt], acc[i]);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            });
                        } else {
                            acc[(r - 60) & 3] = mfma16<PREC>(R[r % D], breg[30 + ((r - 60) >> 2)], acc[(r - 60) & 3]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (r + D < NR) {
                            R[r % D] = rd(VdIC<r + D>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // the rows of this group's NEXT box (scalars: prep_rows of the phase before) start their way into the caches; the
                        // next phase loads them.  HERE, a few MFMAs into the loop: the compiler guards the first writes of the ring
                        // registers with s_waitcnt vmcnt (they were targets of the last phase's loads and sources of its stores), and
                        // those waits must not see the touch loads
                        if constexpr (r == 5 && VD_L0_TOUCH) { touch_rows(); __builtin_amdgcn_sched_barrier(0); }
                    });
                    if (VD_L0_PRIO) __builtin_amdgcn_s_setprio(0);
                } else {
                    uint4 A[2][MTW];
                    const int tp0 = lds_tap[half], tp1 = lds_tap[2 + half];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) A[0][i] = *reinterpret_cast<const uint4*>(patch + a_off[i] + tp0);
#pragma unroll
                    for (int i = 0; i < MTW; ++i) A[1][i] = *reinterpret_cast<const uint4*>(patch + a_off[i] + tp1);
                    int tp2 = lds_tap[4 + half];
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int tp3 = lds_tap[2 * ((s + 3 < S) ? s + 3 : S - 1) + half];
#pragma unroll
                        for (int i = 0; i < MTW; ++i) {
                            acc[i] = mfma16<PREC>(A[s & 1][i], breg[s], acc[i]);
                            __builtin_amdgcn_sched_barrier(0);
                            if (s + 2 < S) A[s & 1][i] = *reinterpret_cast<const uint4*>(patch + a_off[i] + tp2);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        tp2 = tp3;
                    }
                }
                tick(1);
            }
        } else {
            // ---- the other half: finish box jn - 1 (its accumulators are in this wave's registers since the previous phase) and build
            //      the kw-slots of box jn = (q + 1 - grp) / 2 ----
            const int jn = (q + 1 - grp) >> 1;
            const bool have_prev = jn - 1 >= 0 && jn - 1 < nmine;
            load_raw();                      // (unconditional: keeps `raw` out of the K loop's live ranges; box scalars: prep_rows)
            __builtin_amdgcn_sched_barrier(0);
            if (have_prev) {
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
#pragma unroll
                    for (int qh = 0; qh < 2; ++qh) {
                        const int r0 = 8 * qh;
                        float m0 = fmaxf(fmaxf(acc[i][r0], acc[i][r0 + 1]), fmaxf(acc[i][r0 + 2], acc[i][r0 + 3]));
                        float m1 = fmaxf(fmaxf(acc[i][r0 + 4], acc[i][r0 + 5]), fmaxf(acc[i][r0 + 6], acc[i][r0 + 7]));
                        m0 = fmaxf(m0 + bias, 0.f); m1 = fmaxf(m1 + bias, 0.f);
                        const int ql = (i * 4 + half + 2 * qh) * 2;
                        uint16_t hi, lo;
                        split16<PREC>(m0, hi, lo);
                        stg16[ql * 32 + (lane & 31)] = hi;
                        split16<PREC>(m1, hi, lo);
                        stg16[(ql + 1) * 32 + (lane & 31)] = hi;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            tick(3);
            expand();                        // this group's patch is free: its K loop ended a phase ago.  (Unconditional -- past the group's
                                             //  last box it rebuilds that box's patch: a row load left pending on some path would make the
                                             //  compiler wait for vector memory in front of the K loop's first LDS reads into those registers)
            __builtin_amdgcn_sched_barrier(0);
            tick(4);
            // output slots LAST: the stores are the only vector-memory operations in flight when the phase ends, and nothing waits for
            // them (in front of the expansion they sat between the row loads and their s_waitcnt, which then waited for the stores' acks)
            if (have_prev) {
                const int64_t out_base = ep_base;
                const int64_t lim64 = out_total - out_base;
                const int lim = (int)(lim64 > 0x7fffffff ? 0x7fffffff : (lim64 < 0 ? 0 : lim64));
                uint4* dslots = reinterpret_cast<uint4*>(p.dst) + out_base;
                const uint4* stg4 = reinterpret_cast<const uint4*>(stg16);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int item = lane + 64 * k;
                    const int ch = item & 3;
                    const uint4 v = stg4[item];
                    const int base = o_reg[k];
                    if (base >= 0 && base < lim)
                        dslots[(uint32_t)base + (uint32_t)(wn * 4 + ch) * (uint32_t)p.out_chunk_stride] = v;
                }
            }
            if constexpr (FS) {              // scalars of the rows this group's next K phase loads: its box jn + 1 (past the end: jn again)
                const int bn = b_lo + grp + 2 * (jn + 1 < nmine ? jn + 1 : (jn < nmine ? jn : (nmine > 0 ? nmine - 1 : 0)));
                prep_rows(bn < b_hi ? bn : b_hi - 1);
            }
        }
        tick(k_phase ? -1 : 5);
        VD_LDS_BARRIER();                    // the one barrier of a phase (all eight waves)
        tick(k_phase ? 2 : 5);
    }
    if ((VD_DBG(p) & 8) && p.stamps != nullptr && (tid & 255) == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stamps) + ((size_t)wgid * 2 + grp) * 8;
        for (int k = 0; k < 6; ++k) o[k] = t_sum[k];
        o[6] = (unsigned long long)nmine;
        o[7] = t_last - t_first;
    }
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(sink) : "memory");      // (the touch loads' register stays reserved until they have landed)
}

template <int PREC, bool FS>
static int launch_conv0_breg(const VdConvParams& p, hipStream_t st) {
    const int64_t total = (int64_t)p.nclips * p.nbox;
    if (total <= 0) return 0;
    auto kern = conv0_breg_kernel<PREC, FS>;
    static VdDevCache cache;
    int ncu = 0;
    if (int rc = vd_dev_prepare(reinterpret_cast<const void*>(kern), cache, ncu)) return rc;
    const size_t lds = (size_t)2 * p.lds_plane_bytes + 512 + 8 * 2048;
    if (lds > 160 * 1024) return -3;
    const int64_t slots = (int64_t)ncu;
    const int gens = p.persist > 0 ? p.persist : 4;
    int per = (int)((total + slots * gens - 1) / (slots * gens));
    if (per < 2) per = 2;
    per += per & 1;                          // an even number of boxes per workgroup: both role groups get the same share
    const int64_t grid = (total + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, st, p, per);
    return (int)hipGetLastError();
}

extern "C" int vd_conv0_breg(const VdConvParams* pp, void* stream) {
    if (pp == nullptr) return -1;
    const VdConvParams& p = *pp;
    if (p.NT != 2 || p.MW != 2 || p.MTW != 4 || (p.NTW != 0 && p.NTW != 1) || p.S != 32 || p.CC != 1 || p.ncl != 1 ||
        p.ntypes != 1 || p.epi != VD_EPI_POOL_CL || p.pool_t != 1 || p.argmax != nullptr || !p.relu ||
        8 * 1024 > p.lds_plane_bytes || p.src_planes <= 0 || p.src_rows <= 0 || p.type_desc == nullptr)
        return -2;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    VdConvParams q = p;
    q.persist = p.persist & 0xFFFF;
    // frame-tile programs (pair_flip != 0; plan.plan_forward_pix): the K loop that reads every A fragment once for all the tiles
    // it serves, unless persist bit 19 asks for the plain one (A/B measurements, bitwise the same results)
    // (the frame-sharing loop carries the LDS pitches of the 4 x 8 x 8 box as instruction offsets: pair_flip bits 8..15 / 16..31)
    const bool fs = (p.pair_flip & 0xF) != 0 && (p.pair_flip >> 8) == (9 | (189 << 8)) && !(p.persist & 0x80000);
    if (p.prec == VD_PREC_F16) return fs ? launch_conv0_breg<VD_PREC_F16, true>(q, st) : launch_conv0_breg<VD_PREC_F16, false>(q, st);
    if (p.prec == VD_PREC_BF16) return fs ? launch_conv0_breg<VD_PREC_BF16, true>(q, st) : launch_conv0_breg<VD_PREC_BF16, false>(q, st);
    return -2;
}

template <int PREC, int MTW, bool SO = false, int NTW = 1, int BAL = 0, bool SQ = false>
static int launch(const VdConvParams& p, hipStream_t st) {
    constexpr bool X3 = (PREC == VD_PREC_BF16X3 || PREC == VD_PREC_F16X3 || PREC == VD_PREC_F16C8);
    const int groups = (p.nclips + p.ncl - 1) / p.ncl;
    const int64_t total = (int64_t)groups * p.nbox;
    if (total <= 0) return 0;
    size_t lds = (size_t)((X3 && !SQ) ? 2 : 1) * p.lds_plane_bytes +
                 (size_t)(2 * p.S + (p.MW * MTW + BAL) * 4 + (PREC == VD_PREC_F16C8 ? p.S / 4 : 0)) * sizeof(int) + 16;
    if (lds > 160 * 1024) return -3;
    if ((p.dbg & 0x100) && lds < 100 * 1024) lds = 100 * 1024;   // diagnostic (tools/stamps.py --alone): one workgroup per CU
    auto kern = conv_mfma_kernel<PREC, MTW, SO, NTW, BAL, SQ>;
    if (p.NT % NTW != 0) return -2;
    const int ncols = p.NT / NTW;
    static VdDevCache cache;
    int ncu = 0;
    if (int rc = vd_dev_prepare(reinterpret_cast<const void*>(kern), cache, ncu)) return rc;
    // resident workgroups per CU: LDS and the register budget of this instantiation (MTW 4: 3 waves
    // per SIMD, otherwise 2; x3 variants of MTW 4 use more registers -> 2)
    int occ = (int)((160 * 1024) / lds);
    const int waves_per_simd = VD_OCC(PREC, MTW, NTW, BAL);
    const int wg_waves = ncols * p.MW;
    const int by_regs = (waves_per_simd * 4) / wg_waves;
    if (occ > by_regs) occ = by_regs;
    if (occ < 1) occ = 1;
    // a few workgroup "generations" keep the tail short while still amortising the dispatch
    const int64_t slots = (int64_t)ncu * occ;
    const int gens = p.persist & 0xFFFF;          // (the bits above are options of single programs)
    int per = (gens > 0 && MTW * NTW <= 4) ? (int)((total + slots * gens - 1) / (slots * gens)) : 1;
    if (per < 1) per = 1;
    const int64_t grid = (total + per - 1) / per;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * ncols * p.MW), lds, st, p, per, (int)total);
    return (int)hipGetLastError();
}

static bool vd_first_level_seq_enabled() {       // VD_X3_SEQ=0: both planes resident everywhere (A/B measurements)
    static const bool on = [] { const char* e = getenv("VD_X3_SEQ"); return !(e && e[0] == '0'); }();
    return on;
}

extern "C" int vd_conv_mfma(const VdConvParams* pp, void* stream) {
    if (pp == nullptr) return -1;
    const VdConvParams& p = *pp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ntw = p.NTW > 0 ? p.NTW : 1;
    if (p.NT % ntw != 0) return -2;
    const int wgw = (p.NT / ntw) * p.MW;
    if (wgw < 1 || wgw > 4 || p.lds_plane_bytes % 16 != 0) return -2;
    if (p.ncl * 65536 <= 0 || p.S <= 0 || p.CC <= 0) return -2;
    if (p.MTW * ntw < 8 && (p.gather_stride >> 6) > (int64_t)wgw * (p.MTW * ntw <= 4 ? 14 : 17)) return -2;   // patch larger than the DMA register budget
    if (p.clip_index != nullptr && (p.ncl != 1 || p.CC != 1 || p.MTW * ntw > 4)) return -2;
    if (p.prec == VD_PREC_F16C8) {      // fp16 + fp8 corrections: the last level's forward only (see include/vd_hip.h)
        if (ntw != 1 || p.MTW != 4 || p.S % 4 != 0 || (p.epi != VD_EPI_POOL_FEAT && p.epi != VD_EPI_POS_FEAT) || p.out_scale == nullptr ||
            p.select || p.atomic || p.src_split_cc > 0 || p.emit_lo != 0)
            return -2;
        // (position-tile programs: one box type and one packed B operand set per pool window, pool_t = 2, one wave row, no arg-max)
        if (p.epi == VD_EPI_POS_FEAT && (p.MW != 1 || p.NT != 4 || p.pool_t != 2 || p.argmax != nullptr || p.w_box_stride <= 0)) return -2;
        if (p.epi != VD_EPI_POS_FEAT && p.w_box_stride != 0) return -2;
        return launch<VD_PREC_F16C8, 4>(p, st);
    }
    if (p.emit_lo != 0) {   // low plane of a single-pass program's pooled outputs: staged channels-last epilogue only, and the staging tile
                            // (two planes of pooled rows) must fit the one patch plane it aliases
        const int mt_tot = p.MW * p.MTW + ((ntw == 2 && p.MTW == 3) ? 1 : 0);
        if (p.prec > VD_PREC_F16 || p.epi != VD_EPI_POOL_CL || p.argmax != nullptr || p.dst_plane_stride <= 0 ||
            (int64_t)2 * mt_tot * 4 * (p.pool_t == 2 ? 1 : 2) * p.NT * 32 * 2 > p.lds_plane_bytes)
            return -2;
    }
    if (ntw == 2 && p.MTW == 3) {   // balanced 7-tile layout (2 x 2 waves), single-pass formats
        if (p.MW != 2 || p.NT != 4 || p.select || p.src_split_cc > 0 || p.atomic || p.w_box_stride != 0) return -2;
        if ((p.gather_stride >> 6) > (int64_t)4 * 17) return -2;
        if (p.prec == VD_PREC_F16) return launch<VD_PREC_F16, 3, false, 2, 1>(p, st);
        if (p.prec == VD_PREC_BF16) return launch<VD_PREC_BF16, 3, false, 2, 1>(p, st);
        return -2;
    }
    if (ntw == 2) {     // two N tiles per wave: forward / dgrad programs with 4 M tiles per wave (x1 and x3 operand formats)
        if ((p.MTW != 4 && p.MTW != 2) || p.select || p.src_split_cc > 0 || p.atomic || p.w_box_stride != 0) return -2;
        switch (p.prec) {
            case VD_PREC_BF16: return p.MTW == 4 ? launch<VD_PREC_BF16, 4, false, 2>(p, st) : launch<VD_PREC_BF16, 2, false, 2>(p, st);
            case VD_PREC_F16: return p.MTW == 4 ? launch<VD_PREC_F16, 4, false, 2>(p, st) : launch<VD_PREC_F16, 2, false, 2>(p, st);
            case VD_PREC_BF16X3: return p.MTW == 4 ? launch<VD_PREC_BF16X3, 4, false, 2>(p, st) : launch<VD_PREC_BF16X3, 2, false, 2>(p, st);
            case VD_PREC_F16X3: return p.MTW == 4 ? launch<VD_PREC_F16X3, 4, false, 2>(p, st) : launch<VD_PREC_F16X3, 2, false, 2>(p, st);
            default: return -2;
        }
    }
    if (ntw != 1) return -2;
    if (p.MTW != 5 && (p.atomic || p.w_box_stride != 0 || p.select || p.src_split_cc > 0)) {
        // second-order programs (and accumulating dgrad launches): hi+lo pairs (train.GradMatchEngine; fp16 pairs with the
        // power-of-two scales of its sweeps, bf16 pairs unscaled)
        if (p.select == 2 && (p.epi != VD_EPI_POOL_CL || p.argmax == nullptr)) return -2;
        if (p.prec == VD_PREC_F16X3) {
            if (p.MTW == 2) return launch<VD_PREC_F16X3, 2, true>(p, st);
            if (p.MTW == 4) return launch<VD_PREC_F16X3, 4, true>(p, st);
            if (p.MTW == 7) return launch<VD_PREC_F16X3, 7, true>(p, st);
            if (p.MTW == 8) return launch<VD_PREC_F16X3, 8, true>(p, st);
            return -2;
        }
        if (p.prec != VD_PREC_BF16X3) return -2;
        if (p.MTW == 2) return launch<VD_PREC_BF16X3, 2, true>(p, st);
        if (p.MTW == 4) return launch<VD_PREC_BF16X3, 4, true>(p, st);
        if (p.MTW == 7) return launch<VD_PREC_BF16X3, 7, true>(p, st);
        if (p.MTW == 8) return launch<VD_PREC_BF16X3, 8, true>(p, st);
        return -2;
    }
    // first-level hi+lo programs (one channel chunk per box, 4 M tiles per wave): the plane-sequential K loop (SQ above)
    // (its LDS holds ONE patch plane; a staged channels-last epilogue stages hi+lo tiles -- 2 * MW * MTW * 4 * sets * NT * 32 * 2
    //  bytes -- in that plane: a program whose plane is smaller than its staging falls through to the two-plane kernel)
    const int64_t sq_stage = (p.epi == VD_EPI_POOL_CL && p.argmax == nullptr)
        ? (int64_t)2 * p.MW * p.MTW * 4 * (p.pool_t == 2 ? 1 : 2) * p.NT * 32 * 2 : 0;
    if (p.MTW == 4 && p.CC == 1 && p.ncl == 1 && sq_stage <= p.lds_plane_bytes && vd_first_level_seq_enabled()) {
        if (p.prec == VD_PREC_BF16X3) return launch<VD_PREC_BF16X3, 4, false, 1, 0, true>(p, st);
        if (p.prec == VD_PREC_F16X3) return launch<VD_PREC_F16X3, 4, false, 1, 0, true>(p, st);
    }
#define VD_DISPATCH(PR)                                                   \
    case PR:                                                              \
        if (p.MTW == 2) return launch<PR, 2>(p, st);                      \
        if (p.MTW == 4) return launch<PR, 4>(p, st);                      \
        if (p.MTW == 5) return launch<PR, 5>(p, st);                      \
        if (p.MTW == 7) return launch<PR, 7>(p, st);                      \
        if (p.MTW == 8) return launch<PR, 8>(p, st);                      \
        return -2;
    switch (p.prec) {
        VD_DISPATCH(VD_PREC_BF16)
        VD_DISPATCH(VD_PREC_F16)
        VD_DISPATCH(VD_PREC_BF16X3)
        VD_DISPATCH(VD_PREC_F16X3)
        default: return -2;
    }
#undef VD_DISPATCH
}

// ---- several programs of one instantiation in one launch ------------------------------------------------------------------
template <int PREC, int MTW, bool SO = false>
static int launch_multi(const VdConvParams* const* pp, int n, hipStream_t st) {
    constexpr bool X3 = (PREC == VD_PREC_BF16X3 || PREC == VD_PREC_F16X3);
    VdConvMulti m;
    ::memset(&m, 0, sizeof(m));
    size_t lds = 0;
    int64_t grid = 0;
    int used = 0;
    const int threads = 64 * pp[0]->NT * pp[0]->MW;
    for (int k = 0; k < n; ++k) {
        const VdConvParams& p = *pp[k];
        const int groups = (p.nclips + p.ncl - 1) / p.ncl;
        const int64_t total = (int64_t)groups * p.nbox;
        if (total <= 0) continue;
        if (total > 0x3fffffff) return -2;
        const size_t l = (size_t)(X3 ? 2 : 1) * p.lds_plane_bytes + (size_t)(2 * p.S + p.MW * MTW * 4) * sizeof(int) + 16;
        if (l > lds) lds = l;
        m.p[used] = p;
        m.first[used] = (int)grid;
        m.total[used] = (int)total;
        grid += (total + 7) & ~(int64_t)7;          // every program starts at a multiple of 8 workgroups: the XCD-contiguous box order
        ++used;                                     // of the body reads the XCD off the low bits of its block index
    }
    if (used == 0) return 0;
    if (lds > 160 * 1024 || grid > 0x7fffffff) return -3;
    for (int k = used; k <= VD_MULTI_MAX; ++k) m.first[k] = (int)grid;
    auto kern = conv_mfma_multi_kernel<PREC, MTW, SO>;
    static VdDevCache cache;
    int ncu = 0;
    if (int rc = vd_dev_prepare(reinterpret_cast<const void*>(kern), cache, ncu)) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), lds, st, m);
    return (int)hipGetLastError();
}

extern "C" int vd_conv_mfma_multi(const VdConvParams* const* pp, int n, void* stream) {
    if (pp == nullptr || n < 1 || n > VD_MULTI_MAX) return -1;
    for (int k = 0; k < n; ++k)
        if (pp[k] == nullptr) return -1;
    if (n == 1) return vd_conv_mfma(pp[0], stream);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const VdConvParams& a = *pp[0];
    bool so = false;
    const bool so0 = a.atomic || a.select || a.src_split_cc > 0;
    for (int k = 0; k < n; ++k) {
        const VdConvParams& p = *pp[k];
        // all plain or all accumulating / second-order (include/vd_hip.h): a plain program must not run under the SO instantiation
        if ((p.atomic || p.select || p.src_split_cc > 0) != so0) return -2;
        // one instantiation, one block shape; the plain layouts only (one N tile per wave, no first-level index, no low-plane output)
        if (p.prec != a.prec || p.MTW != a.MTW || p.NT != a.NT || p.MW != a.MW || (p.NTW != 0 && p.NTW != 1)) return -2;
        const int wgw = p.NT * p.MW;
        if (wgw < 1 || wgw > 4 || p.lds_plane_bytes % 16 != 0 || p.ncl * 65536 <= 0 || p.S <= 0 || p.CC <= 0) return -2;
        if (p.MTW < 8 && (p.gather_stride >> 6) > (int64_t)wgw * (p.MTW <= 4 ? 14 : 17)) return -2;
        if (p.clip_index != nullptr || p.emit_lo != 0 || p.w_box_stride != 0 || p.MTW == 5) return -2;
        so = so || p.atomic || p.select || p.src_split_cc > 0;
    }
    if (so) {       // accumulating / second-order programs: hi+lo pairs, as in vd_conv_mfma
        if (a.prec == VD_PREC_F16X3) {
            if (a.MTW == 2) return launch_multi<VD_PREC_F16X3, 2, true>(pp, n, st);
            if (a.MTW == 4) return launch_multi<VD_PREC_F16X3, 4, true>(pp, n, st);
            if (a.MTW == 7) return launch_multi<VD_PREC_F16X3, 7, true>(pp, n, st);
            if (a.MTW == 8) return launch_multi<VD_PREC_F16X3, 8, true>(pp, n, st);
            return -2;
        }
        if (a.prec != VD_PREC_BF16X3) return -2;
        if (a.MTW == 2) return launch_multi<VD_PREC_BF16X3, 2, true>(pp, n, st);
        if (a.MTW == 4) return launch_multi<VD_PREC_BF16X3, 4, true>(pp, n, st);
        if (a.MTW == 7) return launch_multi<VD_PREC_BF16X3, 7, true>(pp, n, st);
        if (a.MTW == 8) return launch_multi<VD_PREC_BF16X3, 8, true>(pp, n, st);
        return -2;
    }
#define VD_DISPATCH_MULTI(PR)                                                \
    case PR:                                                                 \
        if (a.MTW == 2) return launch_multi<PR, 2>(pp, n, st);               \
        if (a.MTW == 4) return launch_multi<PR, 4>(pp, n, st);               \
        if (a.MTW == 7) return launch_multi<PR, 7>(pp, n, st);               \
        if (a.MTW == 8) return launch_multi<PR, 8>(pp, n, st);               \
        return -2;
    switch (a.prec) {
        VD_DISPATCH_MULTI(VD_PREC_BF16)
        VD_DISPATCH_MULTI(VD_PREC_F16)
        VD_DISPATCH_MULTI(VD_PREC_BF16X3)
        VD_DISPATCH_MULTI(VD_PREC_F16X3)
        default: return -2;
    }
#undef VD_DISPATCH_MULTI
}
