// pnr_split.hip -- the fused per-point network at fp32-class accuracy on the f16 matrix cores (gfx950).
//
// fp32 MFMA peaks at 157 TFLOP/s on MI355X, 1/16 of the f16 rate; an fp32 product, however, is recovered from
// three f16 products when both operands are split into a rounded-to-f16 head and an f16 tail,
//       w = wh + wl,  x = xh + xl :   w x ~= wh xh + wh xl + wl xh        (the dropped wl xl is <= 2^-22 |w x|),
// each product exact in the fp32 accumulator (11 x 11 significand bits).  The representation itself is ~19 bits for the
// weights, not 22: the tail of a typical |w| ~ 0.05 weight (|w - wh| <= 1.5e-5 < 6.1e-5) lies in the fp16 SUBNORMAL range and
// carries ~8 bits.  Measured per-point error against the reference's fp32 outputs: 1.2-1.9e-6 (bar 2e-5).  This kernel runs the SAME fused chain
// as pnr_mlp.hip -- residual stream in the accumulators, weights streamed L2 -> VGPR, activations through LDS,
// lin_z folded into per-texel tables (fp32 tables here) -- with every operand carried as such a pair:
//   weights      two packed streams (head, tail) with the layout of the 16-bit folded stream   (pnr_pack_mlp_split)
//   activations  two LDS images (head, tail); relu(v) -> h = f16(v), l = f16(v - h)
//   tables       fp32 rows, bilinear blend in fp32, added to the accumulators as they are       (pnr_fold_latent_f32)
// 3 MFMAs per (weight fragment, activation fragment) instead of 1, 2x the operand bytes: ~1/3 of the f16 kernel's
// rate, several times the fp32-MFMA ceiling, at the accuracy class of the reference's own fp32 arithmetic
// (tests/test_hip_split.py holds it to the bars of tests/test_hip_f32.py: per-point |rgb| <= 2e-5).
// Folded form; 64-point tiles (several source views: the view sum parked in a per-workgroup scratch).  The unfused fp32-MFMA
// path (pnr_f32.hip) remains the implementation-independent yardstick.
//
// Round 4: every 512-wide linear starts with the products against the wave's OWN 64 features, taken from its accumulators (they
// are B fragments as they are), while the same fragments go to the operand images for the other waves: stage_own / gemm_split_rot.
//
// Kernels of this file:
//   eval_split_kernel<RAYS, MV, TIMING, TRAIN, GUARD>   the fused network.  GUARD = the fp16-range guard (pnr_saturation_guard): the
//                                                same bits, plus one flag bit per layer whose operand image received a value >= 65504.
//                                                TRAIN = the fp32-class TRAINING forward: the same launch also
//                                                copies every wide linear's (head, tail) operand image out of LDS (relu(x) /
//                                                relu(net): the operands of the weight gradients), writes 1-bit relu masks and
//                                                the stream in front of lin_out; the inference instantiations compile to the
//                                                code they had before the flag existed.
//   bwd_split_kernel<MV>                         the fused data-gradient chain behind it: all 15 transposed products of a network
//                                                (transposed head / tail streams, gradient of the stream in the accumulators,
//                                                gradient images copied out the same way, d z_lat / d code as fp32 rows).
//   pack_weights_bwd_split_kernel                its weight streams, from the raw parameters.
// The weight gradients over those images are dw_split_kernel (pnr_bwd.hip); host side: pnr_f32.hip (pnr_*_split_train).
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_device.h"
#include "pnr_internal.h"
#include "pnr_layout.h"

namespace pnr {

typedef Prec<PNR_PREC_F16> PH;
typedef PH::T8 h8;

// The split kernels are generic in the wave count (stage_own, gemm_split_rot, the geometry loop, the own-K packing: IT = 4, BODIES = 2
// at NW = 4), but only the 8-wave form is parity-tested; the 4-wave build exists for timing (profiles/r05_split_gemm_ubench.txt: -13 %)
// and has to ask for itself.
#if !defined(PNR_VARIANT)
static_assert(NW == 8, "pnr_split.hip: the product build is 8 waves per workgroup (other wave counts: variant builds, timing only)");
#endif

// LDS map: two activation images (head / tail), two lin_in operand images, corner metadata, lin_out partials.
// The fp32 table rows (2 KiB + 16 B pad per point) are looked up into the space of the two activation images
// while those are free (tile start, and after fc_1 of blocks 0-1 has finished reading).
template <int MT_> struct SplitTileT {
    static_assert(MT_ == 64 || MT_ == 32, "64-point tiles (the 32-point form, view sum in registers, is kept for A/B: -DPNR_SPLIT_MV32)");
    static constexpr int MT = MT_, JT = MT_ / 32;
    static constexpr int A_HI = 0;
    static constexpr int A_LO = MT * ROW_ACT;             // 66,560 (64)
    static constexpr int LDS_IN = 2 * MT * ROW_ACT;        // 133,120 (head image of the lin_in operand)
    static constexpr int IN_LO_DELTA = MT * ROW_IN;        // tail image right behind it
    static constexpr int LDS_META = LDS_IN + 2 * MT * ROW_IN;
    static constexpr int LDS_OUT = LDS_META + MT * 32;
    static constexpr int LDS_TOTAL = LDS_OUT + NW * MT * 16;  // 161,792 B (64) / 80,896 B (32)
    static constexpr int ROW_TAB = D_HID * 4 + 16;          // fp32 table row: 2064 B (129 16-B slots: conflict-free)
    static constexpr int LDS_Z = 0;                         // (geometry_item only needs LDS_IN / LDS_META)
    static_assert(MT * ROW_TAB <= LDS_IN, "table image must fit in the space of the two activation images");
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
};

struct SplitRing {
    h8 h[4][IT], l[4][IT];
    const char *base_h, *base_l;  // this wave's head / tail stream + lane*16
    int pf_rs, pf_view;
};

// prefetch cursor: the per-view segment [0, RS_VIEW_END_F) is consumed NS times, then the pooled tail, then wrap
__device__ __forceinline__ void ring_advance(SplitRing &R, int NS) {
    int rs = R.pf_rs + 4, v = R.pf_view;
    if (rs == RS_VIEW_END_F && v + 1 < NS) { v += 1; rs = 0; }
    else if (rs == RS_TOTAL_F) { rs = 0; v = 0; }
    R.pf_rs = rs; R.pf_view = v;
}

// the fp32-class BACKWARD chain (bwd_split_kernel below) walks a transposed stream in the layout of pnr_layout.h's backward
// stream: pooled head [0, BRS_HEAD_END) once, then the per-view segment [BRS_HEAD_END, BSRS_TOTAL) NS times, then wrap
// (lin_out^T, fc_1[4]^T fc_0[4]^T fc_1[3]^T fc_0[3]^T | fc_1[2]^T ... fc_0[0]^T, lin_z[2]^T lin_z[1]^T lin_z[0]^T, lin_in^T)
constexpr int BSRS_TOTAL = BRS_TOTAL;  // 424
static_assert(BSRS_TOTAL % 4 == 0, "ring depth 4 needs aligned segments");
constexpr size_t BSPACKED_BYTES = (size_t)BSRS_TOTAL * IT * 1024 * NW;  // one blob (head or tail): 5,308,416 B
struct SplitAdvFwd {
    static __device__ __forceinline__ void step(SplitRing &R, int NS) { ring_advance(R, NS); }
};
struct SplitAdvBwd {
    static __device__ __forceinline__ void step(SplitRing &R, int NS) {
        int rs = R.pf_rs + 4, v = R.pf_view;
        if (rs == BSRS_TOTAL) {
            if (v + 1 < NS) { v += 1; rs = BRS_HEAD_END; }
            else { rs = 0; v = 0; }
        }
        R.pf_rs = rs; R.pf_view = v;
    }
};

__device__ __forceinline__ f32x16 mf(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// acc[it][jt] += (Wh + Wl)(Xh + Xl) without the tail-tail term; B rows at bhi0 + jt*jstride (+ lo_delta for the tails)
template <int JT, typename ADV = SplitAdvFwd>
__device__ __forceinline__ void gemm_split(f32x16 (&acc)[IT][JT], const char *smem, uint32_t bhi0, uint32_t jstride,
                                           uint32_t lo_delta, int nbody, SplitRing &R, int NS) {
    h8 bh[2][JT], bl[2][JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        bh[0][jt] = lds8<PH>(smem, bhi0 + jt * jstride);
        bl[0][jt] = lds8<PH>(smem, bhi0 + jt * jstride + lo_delta);
    }
#pragma unroll 1
    for (int body = 0; body < nbody; ++body) {
        const size_t pf = (size_t)R.pf_rs * (IT * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                bh[cur ^ 1][jt] = lds8<PH>(smem, bhi0 + jt * jstride + (j + 1) * 32);
                bl[cur ^ 1][jt] = lds8<PH>(smem, bhi0 + jt * jstride + lo_delta + (j + 1) * 32);
            }
            h8 ah[IT], al[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) { ah[it] = R.h[j][it]; al[it] = R.l[j][it]; }
            // the three products of one accumulator are spread over the step so that consecutive MFMAs are independent
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bh[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bl[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(al[it], bh[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                R.h[j][it] = gload8<PH>(R.base_h + pf + j * (IT * 1024) + it * 1024);
                R.l[j][it] = gload8<PH>(R.base_l + pf + j * (IT * 1024) + it * 1024);
            }
            // Issue order of one k-step, pinned (round 3): hipcc batches the 16 refills of a loop body behind its last MFMA and
            // clusters the LDS reads; here every LDS read of the next step's B fragments follows ONE MFMA and every weight
            // refill follows TWO -- (M L) x 2JT, (M M G) x 2IT.  Same-box A/B on sn64 / srn_car / DTU: +2.7 ... +6 % in four
            // runs (profiles/r03_split_kernel_ab.txt, which also lists the eleven other orders tried: -6 ... +4 %; the order
            // matters here, unlike in the f16 kernel where the same kind of pinning measured -1.9 %).  Results unchanged.
#pragma unroll
            for (int i = 0; i < 2 * JT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read
            }
#pragma unroll
            for (int i = 0; i < 2 * IT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
            }
            if constexpr (3 * IT * JT - 2 * JT - 4 * IT > 0) __builtin_amdgcn_sched_group_barrier(0x008, 3 * IT * JT - 2 * JT - 4 * IT, 0);
        }
        bhi0 += 128;
        ADV::step(R, NS);
    }
}

// head / tail of 8 fp32 values (optionally through relu); MODE.FP16_OVFL is set: heads saturate at 65504.
// tail = f16(v - head): one v_fma_mix{lo,hi}_f16 per value (fp32 FMA of the f16 head taken straight out of its packed
// register, times -1, plus v; result rounded once to f16) -- the same bits as convert-back + subtract + convert
// (v - head is exact in fp32), 5 VALU operations per pair instead of 8.
// GUARD: `amax` (a packed f16 pair) follows the largest HEAD produced, one v_pk_maximum3_f16 per FOUR values (gfx950; the
// IEEE-2019 maximum: a NaN head sticks): the fp16-range guard of pnr_saturation_guard().  Half the VALU cost of following the
// fp32 inputs with v_max3_f32 (round 5), which is what lets the guard run on every inference call (round 6).  Heads are >= 0
// behind the relu, so no magnitudes are needed; a head of 65504 means v >= 65488 (round to nearest): the guard fires 16 below
// the exact saturation point.
template <bool RELU, bool GUARD = false>
__device__ __forceinline__ void split8(const float (&v)[8], h8 &hi, h8 &lo, [[maybe_unused]] uint32_t *amax = nullptr) {
    static_assert(RELU || !GUARD, "the guard follows relu'd heads");
    u32x4 uh, ul;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 p = {v[2 * k], v[2 * k + 1]};
        if (RELU) {
            // one v_med3_f32 per value: median(v, 0, FLT_MAX) = max(v, 0) for every finite v (the head saturates at 65504 anyway;
            // with +inf as the upper bound LLVM folds the median back into the two-instruction maxnum).  NOT an inline-asm
            // v_max_f32: the accumulators come straight out of the MFMA pipe and hipcc does not place the MFMA -> VALU wait
            // states in front of inline asm -- that form read stale registers in the multi-view instantiations.
            p[0] = __builtin_amdgcn_fmed3f(p[0], 0.f, 3.402823466e38f); p[1] = __builtin_amdgcn_fmed3f(p[1], 0.f, 3.402823466e38f);
        }
        const f16x2 h = __builtin_convertvector(p, f16x2);
        uh[k] = __builtin_bit_cast(uint32_t, h);
        uint32_t l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(uh[k]), "v"(p[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(uh[k]), "v"(p[1]));
        ul[k] = l;
    }
    if constexpr (GUARD) {  // (the operands are v_cvt_pk results, never straight MFMA outputs: inline asm is safe here)
        uint32_t m = *amax;
        asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(m) : "v"(uh[0]), "v"(uh[1]));
        asm("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(m) : "v"(uh[2]), "v"(uh[3]));
        *amax = m;
    }
    hi = __builtin_bit_cast(h8, uh);
    lo = __builtin_bit_cast(h8, ul);
}

// [relu](acc) -> head / tail images (storage order, like write_act)
template <typename ST, bool RELU = true, int JT>
__device__ __forceinline__ void write_split(const f32x16 (&acc)[IT][JT], char *smem, uint32_t waddr) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const f32x16 &a = acc[it][jt];
            const uint32_t ad = waddr + jt * 32 * ROW_ACT + it * 64;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = a[8 * half + e];
                h8 hi, lo;
                split8<RELU>(v, hi, lo);
                *reinterpret_cast<h8 *>(smem + ST::A_HI + ad + 16 * half) = hi;
                *reinterpret_cast<h8 *>(smem + ST::A_LO + ad + 16 * half) = lo;
            }
        }
}

// ---- "own K block first" form of the 512 x 512 linears (round 4) -----------------------------------------------------------
// A wave's accumulators ARE B fragments of the next linear: lane (point pl, half h) holds, in registers 8 rr .. 8 rr + 7 of
// feature tile `it`, exactly the 8 K-elements an MFMA B operand wants from that lane for the 16 features
// feat_of(wv IT + it, {0,1}, 8 rr + {0..7}) (the property lin_out has always used).  So the next GEMM's products against THIS
// wave's 64 features need no LDS round trip and no barrier: relu + (head, tail) split of 16 accumulator values yields the B
// fragments of one k-step, which are multiplied right away (12 MFMAs) AND stored into the operand images for the other seven
// waves.  The split epilogue (VALU-bound: ~170 operations per wave and GEMM, 8 % of a tile with the matrix pipe idle) thereby
// runs in the shadow of 48 MFMAs per wave that had to be issued anyway, and the barrier that publishes the images is crossed
// with an eighth of the GEMM already done.  The packed stream carries each wave's own K block first, in register order, then
// the blocks of waves wv+1 .. wv+7 (mod 8) in the image's storage order (pnr_pack.hip, OWNK).  Per output element the 512
// products are summed in a wave-dependent block order -- fixed per (feature, wave), identical for every point and tile size:
// chunked = whole and sharded = unsharded stay bit for bit.
struct NoMark { __device__ __forceinline__ void operator()(int) const {} };
// MARK: the TIMING instantiation's clock (sub-phases PH_OWN_PROLOGUE / PH_OWN_KSTEPS); a no-op everywhere else
template <typename ST, int JT, bool GUARD, typename MARK = NoMark>
__device__ __forceinline__ void stage_own(f32x16 (&acc)[IT][JT], const f32x16 (&src)[IT][JT], char *smem, uint32_t waddr,
                                          SplitRing &R, int NS, [[maybe_unused]] uint32_t *amax, MARK mark = MARK()) {
    // k-step jj of the own block = (feature tile jj >> 1, register half jj & 1); 2 IT k-steps = IT / 2 ring bodies of 4
    static_assert(IT % 2 == 0, "whole ring bodies");
    h8 bh[2][JT], bl[2][JT];
    auto make = [&](int jj, int buf) {
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[jj >> 1][jt][8 * (jj & 1) + e];
            split8<true, GUARD>(v, bh[buf][jt], bl[buf][jt], amax);
        }
    };
    make(0, 0);
    mark(PH_OWN_PROLOGUE);  // accumulator drain + the first k-step's split
#pragma unroll
    for (int body = 0; body < IT / 2; ++body) {
        const size_t pf = (size_t)R.pf_rs * (IT * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int jj = body * 4 + j;
            const int cur = jj & 1;
            if (jj + 1 < 2 * IT) make(jj + 1, cur ^ 1);
            h8 ah[IT], al[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) { ah[it] = R.h[j][it]; al[it] = R.l[j][it]; }
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bh[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bl[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(al[it], bh[cur][jt], acc[it][jt]);
            // the same fragments, for the other waves: storage position of (feature tile jj >> 1, register half jj & 1), as write_split
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const uint32_t ad = waddr + jt * 32 * ROW_ACT + (jj >> 1) * 64 + 16 * (jj & 1);
                *reinterpret_cast<h8 *>(smem + ST::A_HI + ad) = bh[cur][jt];
                *reinterpret_cast<h8 *>(smem + ST::A_LO + ad) = bl[cur][jt];
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                R.h[j][it] = gload8<PH>(R.base_h + pf + j * (IT * 1024) + it * 1024);
                R.l[j][it] = gload8<PH>(R.base_l + pf + j * (IT * 1024) + it * 1024);
            }
            // issue order of one k-step, pinned: the split of the NEXT k-step's 16 values (~40 VALU) is spread over this step's
            // MFMAs, the image stores and the ring refills of THIS step sit behind its later MFMAs -- left to itself hipcc puts
            // all refills behind the stage's last MFMA and runs the last MFMAs back to back with the VALU work in front of them
            if constexpr (IT == 2) {
#pragma unroll
                for (int i = 0; i < 3 * IT * JT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                          // up to 4 VALU
                    if (i >= 4 && i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 LDS write  (i = 4, 6, 8, 10)
                    if (i >= 5 && i % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read  (i = 5, 7, 9, 11)
                }
            } else {
                // 3 IT JT MFMAs, 2 JT image stores, 2 IT refills, ~20 JT VALU of the next k-step's split
                constexpr int NM = 3 * IT * JT;
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (i >= 4 && i < 4 + 2 * JT) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (i >= NM - 2 * IT - 2 && i < NM - 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
        ring_advance(R, NS);
    }
    mark(PH_OWN_KSTEPS);
}

// the other seven K blocks of the same linear, from the operand images: blocks (wv + 1) .. (wv + 7) mod 8, the order the stream
// carries them in.  Loop body and pinned issue order of gemm_split.
template <int JT>
__device__ __forceinline__ void gemm_split_rot(f32x16 (&acc)[IT][JT], const char *smem, uint32_t a_rd0, uint32_t jstride,
                                               uint32_t lo_delta, int wv, SplitRing &R, int NS) {
    constexpr int BLK = SL * 2;         // bytes of one wave's K block in an image row (128 at 8 waves)
    constexpr int BODIES = SL / 64;     // ring bodies (4 k-steps of 16) per block
    h8 bh[2][JT], bl[2][JT];
    uint32_t bhi0 = a_rd0 + (uint32_t)((wv + 1) & (NW - 1)) * BLK;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        bh[0][jt] = lds8<PH>(smem, bhi0 + jt * jstride);
        bl[0][jt] = lds8<PH>(smem, bhi0 + jt * jstride + lo_delta);
    }
#pragma unroll 1
    for (int mb = 2 * BODIES; mb < (NW + 1) * BODIES; ++mb) {   // body index counted in ring bodies: block m = mb / BODIES
        const int m = mb / BODIES, sub = mb % BODIES;
        // after this body's last k-step: the next 64 K of the same block, or the start of the next block
        // (after the last block: this wave's own, read and dropped)
        const uint32_t nxt = sub + 1 < BODIES ? bhi0 + 128 : a_rd0 + (uint32_t)((wv + m) & (NW - 1)) * BLK;
        const size_t pf = (size_t)R.pf_rs * (IT * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            const uint32_t rd = j < 3 ? bhi0 + (j + 1) * 32 : nxt;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                bh[cur ^ 1][jt] = lds8<PH>(smem, rd + jt * jstride);
                bl[cur ^ 1][jt] = lds8<PH>(smem, rd + jt * jstride + lo_delta);
            }
            h8 ah[IT], al[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) { ah[it] = R.h[j][it]; al[it] = R.l[j][it]; }
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bh[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(ah[it], bl[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc[it][jt] = mf(al[it], bh[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                R.h[j][it] = gload8<PH>(R.base_h + pf + j * (IT * 1024) + it * 1024);
                R.l[j][it] = gload8<PH>(R.base_l + pf + j * (IT * 1024) + it * 1024);
            }
            constexpr int NM = 3 * IT * JT, PER = (NM - 2 * JT) / (2 * IT);
#pragma unroll
            for (int i = 0; i < 2 * JT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 LDS read
            }
#pragma unroll
            for (int i = 0; i < 2 * IT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // PER MFMAs (2 at IT = JT = 2)
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // 1 VMEM read
            }
            if constexpr (NM - 2 * JT - PER * 2 * IT > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - 2 * JT - PER * 2 * IT, 0);
        }
        bhi0 = nxt;
        ring_advance(R, NS);
    }
}

// fp32 bilinear lookup of table b: wave handles points wave*8..+7; lane handles storage slots 4*lane..+3 and 256 + 4*lane..+3
// (two fully coalesced 1 KiB loads per corner, and two 16-byte LDS writes per point at a 16-byte lane stride -- a 32-byte
// lane stride put lanes l and l+8 on the same banks: 5.9 % of the LDS cycles of the round-2 kernel were conflicts);
// rows land in the table image (ROW_TAB stride) at offset 0
template <int GB, typename ST>
__device__ __forceinline__ void gather_table_f32(const EvalParams &q, char *smem, int wv, int lane, int b) {
    static_assert(GB == 2, "pairs of consecutive points");
    const float *tab = reinterpret_cast<const float *>(q.tables) + (size_t)b * q.table_stride + lane * 4;
    // A wave's 8 points are consecutive samples of ONE ray (tiles never straddle rays: K is a multiple of the tile), and consecutive
    // samples mostly fall into the same cell of the feature grid (sn64: 4-12 samples per texel): a point whose four corner offsets
    // equal its predecessor's re-uses the predecessor's rows from registers instead of loading 8 KiB again.  The comparison is
    // wave-uniform (every lane reads the same META words), the arithmetic per point is unchanged (same rows, same order).
    // Same-box A/B against loading every point's rows (profiles/r04_split_kernel_ab.txt, session 9): lookups 30.2 k -> 19.7 k
    // cycles per tile; sn64 +1.1 %, srn_car +3.4 %, DTU +2.0 %.
    // (Wave-per-point, lanes along the channels: whole 1 KiB pieces of a row per instruction.  The LDS-free alternative -- every lane
    // reading the 64 bytes of its own point's row that hold its own 16 features, blended straight into the accumulators, no barrier
    // pair -- measured -2.4 % on sn64 / srn_car and -11.7 % on DTU: eight waves then fetch eight slices of every row at eight
    // different times.  profiles/r05_split_kernel_ab.txt.)
    // (The rows are ordinary, L2-allocating loads on purpose: neighbouring points and tiles hit the same texels.  Non-temporal loads
    // measured -3 % on sn64, -5 % on srn_car, -11 % on DTU, same box: profiles/r05_split_kernel_ab.txt.)
    f32x4 v[GB][4][2];
    uint32_t last[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // offsets of the rows held in v[1]
#pragma unroll 1
    for (int i = 0; i < ST::MT / NW; i += GB) {
        f32x4 w[GB];
        uint32_t off[GB][4];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (ST::MT / NW) + i + u;
            const u32x4 o = *reinterpret_cast<const u32x4 *>(smem + ST::LDS_META + p * 32);
            w[u] = *reinterpret_cast<const f32x4 *>(smem + ST::LDS_META + p * 32 + 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) off[u][c] = __builtin_amdgcn_readfirstlane(o[c]);
        }
        const bool keep0 = off[0][0] == last[0] && off[0][1] == last[1] && off[0][2] == last[2] && off[0][3] == last[3];
        const bool keep1 = off[1][0] == off[0][0] && off[1][1] == off[0][1] && off[1][2] == off[0][2] && off[1][3] == off[0][3];
        if (keep0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[0][c][0] = v[1][c][0]; v[0][c][1] = v[1][c][1]; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[0][c][0] = *reinterpret_cast<const f32x4 *>(tab + off[0][c]);
                v[0][c][1] = *reinterpret_cast<const f32x4 *>(tab + off[0][c] + D_HID / 2);
            }
        }
        if (keep1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[1][c][0] = v[0][c][0]; v[1][c][1] = v[0][c][1]; }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[1][c][0] = *reinterpret_cast<const f32x4 *>(tab + off[1][c]);
                v[1][c][1] = *reinterpret_cast<const f32x4 *>(tab + off[1][c] + D_HID / 2);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) last[c] = off[1][c];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (ST::MT / NW) + i + u;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                f32x4 r = v[u][0][hh] * w[u][0];
                r += v[u][1][hh] * w[u][1];
                r += v[u][2][hh] * w[u][2];
                r += v[u][3][hh] * w[u][3];
                *reinterpret_cast<f32x4 *>(smem + p * ST::ROW_TAB + hh * (D_HID * 2) + lane * 16) = r;
            }
        }
    }
}

// x += this lane's slots of the fp32 table rows
template <typename ST, int JT>
__device__ __forceinline__ void add_from_table(f32x16 (&x)[IT][JT], const char *smem, int pl, int h, int wv) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const char *row = smem + (jt * 32 + pl) * ST::ROW_TAB + ((wv * IT + it) * 32 + h * 16) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(row + 16 * k);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[it][jt][4 * k + e] += t[e];
            }
        }
}

// MV: views go through blocks 0-2 one after the other on the same 64 points; the running view sum (util.combine_interleaved,
// util.py:461-466) does not fit the registers next to x + net + the 64-register head/tail ring, so it is PARKED in a
// per-workgroup scratch (q.mv_ws, [slot][thread] layout: every lane re-reads only what it wrote itself, no synchronisation;
// 128 KiB per workgroup, L2-resident) and read-modify-written once per view and tile -- 256 KiB of traffic against the 6.6 MB
// of weights the same view streams.  (Round 2 ran multi-view scenes on 32-point tiles with the sum in registers: twice the
// weight stream per point, 105-147 k rays/s; -DPNR_SPLIT_MV32 rebuilds that form.)  Fixed summation order
// (view 0 + view 1) + ...: bit-identical to the in-register form.
constexpr int SPLIT_MV_TILE = 64;
template <bool RAYS, bool MV, bool TIMING = false, bool TRAIN = false, bool GUARD = false>
__global__ void __launch_bounds__(NTHREADS, NW / 4) eval_split_kernel(const EvalParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef SplitTileT<MV ? SPLIT_MV_TILE : 64> ST;
    constexpr int JT = ST::JT, MT = ST::MT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int NS = MV ? q.NS : 1;

    const uint32_t a_rd0 = ST::A_HI + pl * ROW_ACT + h * 16;
    const uint32_t in_rd0 = ST::LDS_IN + pl * ROW_IN + h * 16;
    const uint32_t a_wr = pl * ROW_ACT + (wv * IT) * 64 + h * 32;
    const float *bias_lane = q.bias + wv * BIAS_FLOATS_PER_WAVE + h * 16;

    f16_ovfl_mode<PH>();
    // static priority for the second-dispatched half of the workgroup (the arbitration loser of every segment when both waves of
    // a SIMD run at priority 0; MI355X_MICROARCH.md, "two waves per SIMD", item 4): one s_setprio before the main loop, no flips.
    // Same-box A/B: +0.9 % on sn64 / srn_car / DTU (profiles/r03_split_kernel_ab.txt).  Does not touch results.
    if (NW > 4 && wv >= NW / 2) __builtin_amdgcn_s_setprio(1);
    SplitRing R;
    R.base_h = q.wstream + (size_t)wv * (RS_TOTAL_F * IT * 1024) + lane * 16;
    R.base_l = R.base_h + PACKED_BYTES;  // the tail blob follows the head blob (pnr_pack_mlp_split)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            R.h[j][it] = gload8<PH>(R.base_h + j * (IT * 1024) + it * 1024);
            R.l[j][it] = gload8<PH>(R.base_l + j * (IT * 1024) + it * 1024);
        }
    R.pf_rs = 4;
    R.pf_view = 0;
    [[maybe_unused]] unsigned long long *tim = q.tim;
    [[maybe_unused]] unsigned long long tlast = TIMING ? __builtin_readcyclecounter() : 0ull;

    // TRAIN: an accumulator tile set leaves as fp32 rows in natural feature order (D register r of tile (it, jt) = feature
    // 64 wv + 32 it + (r&3) + 8 (r>>2) + 4 h of point 32 jt + pl): 16-byte pieces, `rows` = first row of the tile in dst
    [[maybe_unused]] auto dump_rows = [&](const f32x16 (&a)[IT][JT], float *dst, long long rows, long long rows_left) {
        float *d = dst + ((size_t)rows + pl) * D_HID + (wv * IT) * 32 + 4 * h;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if (jt * 32 + pl >= rows_left) continue;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = {a[it][jt][4 * k], a[it][jt][4 * k + 1], a[it][jt][4 * k + 2], a[it][jt][4 * k + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(d + (size_t)jt * 32 * D_HID + it * 32 + 8 * k));  // streamed past the L2 (dump_image)
                }
        }
    };
    // TRAIN: where the current tile's rows live (set by the tile / view loops; the inference instantiations never touch them)
    [[maybe_unused]] long long tr_rows_view = 0, tr_rows_pooled = 0, tr_rows_left = 0;
    // TRAIN: relu masks for the fused backward chain, one 64-bit word per thread and layer in the layout of the 16-bit training
    // kernels (pnr_device.h: bit (it JT + jt) 16 + r = "register r of accumulator tile (it, jt) is positive"; layer 2b = x
    // entering block b, 2b + 1 = its fc_0 output, 10 = the stream in front of lin_out; [layer][view][tile][thread])
    [[maybe_unused]] size_t tr_mask_view = 0, tr_mask_pooled = 0;
    [[maybe_unused]] const size_t tr_mask_layer = (size_t)NS * (size_t)q.ntiles * NTHREADS;
    [[maybe_unused]] auto put_mask = [&](const f32x16 (&a)[IT][JT], int layer, size_t word) {
        unsigned long long m = 0ull;
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                uint32_t m16 = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) m16 |= (a[it][jt][r] > 0.f ? 1u : 0u) << r;
                m |= (unsigned long long)m16 << ((it * JT + jt) * 16);
            }
        __builtin_nontemporal_store(m, q.d_mask + (size_t)layer * tr_mask_layer + word);
    };
    // TRAIN: the operand images just published (head, tail) -> their 16-bit row sets, whole 1 KiB rows per wave instruction
    [[maybe_unused]] auto dump_pair = [&](char *head, int b) {
        const long long rows = b < COMBINE_LAYER ? tr_rows_view : tr_rows_pooled;
        const size_t total = (size_t)(b < COMBINE_LAYER ? (long long)NS * q.P : q.P) * (D_HID * 2);
        dump_image<MT>(smem, ST::A_HI, head + (size_t)rows * (D_HID * 2), tr_rows_left, wv, lane);
        dump_image<MT>(smem, ST::A_LO, head + total + (size_t)rows * (D_HID * 2), tr_rows_left, wv, lane);
    };
    // fp16-range guard (GUARD instantiation, pnr_saturation_guard): bit l of sat_bits = "a head of 65504 (the largest fp16: heads
    // saturate there; i.e. a value >= 65488) went into the operand image of layer l" (2b: relu(x) entering blocks[b].fc_0, 2b+1: relu(net) entering
    // fc_1, 10: the stream in front of lin_out), bit 11 = a non-finite network output
    [[maybe_unused]] uint32_t sat_bits = 0;
    [[maybe_unused]] uint32_t amax = 0u;  // packed f16 pair: the running maximum of the heads (split8)
    [[maybe_unused]] auto sat_note = [&](int layer) {
        if constexpr (GUARD) {  // either half >= 0x7BFF: 65504, +inf or a NaN
            if ((amax & 0x7FFFu) >= 0x7BFFu || (amax & 0x7FFF0000u) >= 0x7BFF0000u) sat_bits |= 1u << layer;
            amax = 0u;
        }
    };
    [[maybe_unused]] auto own_mark = [&](int ph) { PNR_T(ph); };
    // one residual block on x (resnetfc.py:66-88); lookup: lin_z[b+1] via table b+1 behind it.
    // Every 512-wide linear = stage_own (split epilogue + this wave's own K block, from registers) | barrier | gemm_split_rot
    auto block = [&](f32x16 (&x)[IT][JT], int b, bool lookup) {
        if constexpr (TRAIN) put_mask(x, 2 * b, b < COMBINE_LAYER ? tr_mask_view : tr_mask_pooled);
        __syncthreads();  // table rows / previous operand images are no longer read
        PNR_T(PH_BAR1);
        {
            f32x16 net[IT][JT];
            add_bias<true>(net, bias_lane, 1 + 2 * b);
            if constexpr (TIMING) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the bias has ARRIVED when the stage's clock starts (diagnostic: also drains the ring)
            own_mark(PH_OWN_BIAS);
            stage_own<ST, JT, GUARD>(net, x, smem, a_wr, R, NS, &amax, own_mark);                  // fc_0, own block
            sat_note(2 * b);
            PNR_T(PH_WRITE_X);
            __syncthreads();
            PNR_T(PH_BAR2);
            if constexpr (TRAIN) dump_pair(q.s_a[b], b);
            gemm_split_rot<JT>(net, smem, a_rd0, 32 * ROW_ACT, ST::A_LO - ST::A_HI, wv, R, NS);   // fc_0, blocks of the other waves
            PNR_T(PH_GEMM_FC0);
            if constexpr (TRAIN) put_mask(net, 2 * b + 1, b < COMBINE_LAYER ? tr_mask_view : tr_mask_pooled);
            __syncthreads();
            PNR_T(PH_BAR3);
            add_bias<false>(x, bias_lane, 2 + 2 * b);
            own_mark(PH_OWN_BIAS);
            stage_own<ST, JT, GUARD>(x, net, smem, a_wr, R, NS, &amax, own_mark);                  // fc_1, own block
            sat_note(2 * b + 1);
            PNR_T(PH_WRITE_NET);
        }
        __syncthreads();
        PNR_T(PH_BAR4);
        if constexpr (TRAIN) dump_pair(q.s_n[b], b);
        gemm_split_rot<JT>(x, smem, a_rd0, 32 * ROW_ACT, ST::A_LO - ST::A_HI, wv, R, NS);         // fc_1
        PNR_T(PH_GEMM_FC1_Z);
        if (lookup) {
            __syncthreads();
            gather_table_f32<2, ST>(q, smem, wv, lane, b + 1);
            __syncthreads();
            add_from_table<ST>(x, smem, pl, h, wv);
            PNR_T(PH_TABLE);
        }
    };

    // XCD-aware tile order (pnr_device.h tile_range; which tile a workgroup runs does not touch any result)
    const TileRange order = tile_range(q.ntiles, q.n_xcd);
    for (int tile = order.begin; tile < order.end; tile += order.step) {
        f32x16 x[IT][JT];
#pragma unroll 1
        for (int view = 0; view < NS; ++view) {
            __syncthreads();  // previous tile / view: every reader of the images / IN / META is done
            PNR_T(PH_SYNC_TOP);
            // 8 work items per point (projection, six frequency bands, padding): MT * 8 items over the workgroup's threads
            if constexpr (MT * 8 <= NTHREADS) {
                if (MT * 8 == NTHREADS || tid < MT * 8) geometry_item<PH, RAYS, ST>(q, smem, tile, view, tid % MT, tid / MT);
            } else {
#pragma unroll 1
                for (int wi = tid; wi < MT * 8; wi += NTHREADS) geometry_item<PH, RAYS, ST>(q, smem, tile, view, wi % MT, wi / MT);
            }
            __syncthreads();
            PNR_T(PH_GEOMETRY);
            gather_table_f32<2, ST>(q, smem, wv, lane, 0);
            PNR_T(PH_GATHER);
            add_bias<true>(x, bias_lane, B_IN_Z0);
            gemm_split(x, smem, in_rd0, 32 * ROW_IN, ST::IN_LO_DELTA, KS_IN / 4, R, NS);  // lin_in   resnetfc.py:147
            __syncthreads();  // table rows of every wave are in place
            add_from_table<ST>(x, smem, pl, h, wv);                                           // lin_z[0] via table 0
            PNR_T(PH_GEMM_IN_Z0);
            if constexpr (TRAIN) {
                tr_rows_pooled = (long long)tile * MT;
                tr_rows_view = (long long)view * q.P + tr_rows_pooled;
                tr_rows_left = q.P - tr_rows_pooled;
                tr_mask_pooled = (size_t)tile * NTHREADS + tid;
                tr_mask_view = tr_mask_pooled + (size_t)view * (size_t)q.ntiles * NTHREADS;
            }
#pragma unroll 1
            for (int b = 0; b < COMBINE_LAYER; ++b) block(x, b, b + 1 < COMBINE_LAYER);
            if constexpr (MV) {
                f32x4 *ws = reinterpret_cast<f32x4 *>(q.mv_ws) + (size_t)blockIdx.x * (IT * JT * 4 * NTHREADS) + tid;
                const float inv = 1.f / (float)NS;
                const bool first = view == 0, last = view + 1 == NS;
                // util.combine_interleaved (util.py:461-471): mean, or -- network flag -- the maximum over the source views
                const bool cmax = (reinterpret_cast<const int *>(q.bout)[BOUT_FLAGS_INDEX] & 1) != 0;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        __builtin_amdgcn_sched_barrier(0);  // one accumulator tile (4 loads) in flight at a time: no register spike
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = (it * JT + jt) * 4 + k;
                            f32x4 v = {x[it][jt][4 * k], x[it][jt][4 * k + 1], x[it][jt][4 * k + 2], x[it][jt][4 * k + 3]};
                            if (!first) {  // 1 KiB per wave-instruction
                                const f32x4 prev = ws[i * NTHREADS];  // (ordinary, L2-resident accesses: non-temporal ones measured -1 ... -2.5 %)
                                if (cmax) { v[0] = fmaxf(prev[0], v[0]); v[1] = fmaxf(prev[1], v[1]); v[2] = fmaxf(prev[2], v[2]); v[3] = fmaxf(prev[3], v[3]); }
                                else v = prev + v;
                            }
                            if (!last) ws[i * NTHREADS] = v;
                            else if (!cmax) v *= inv;
                            x[it][jt][4 * k] = v[0]; x[it][jt][4 * k + 1] = v[1]; x[it][jt][4 * k + 2] = v[2]; x[it][jt][4 * k + 3] = v[3];
                        }
                    }
            }
        }
#pragma unroll 1
        for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b) block(x, b, false);
        if constexpr (TRAIN) {
            dump_rows(x, q.f_x5, tr_rows_pooled, tr_rows_left);
            put_mask(x, 10, tr_mask_pooled);
        }

        // lin_out(relu(x)): each wave contracts its own 64 features (the wave's accumulators are the B operand)
        {
            f32x16 o[JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[jt][r] = 0.f;
            const size_t pf = (size_t)R.pf_rs * (IT * 1024);
#pragma unroll
            for (int qk = 0; qk < 2 * IT; ++qk) {
                const int xit = qk >> 1, rr = qk & 1;
                const h8 ah = R.h[qk / IT][qk % IT], al = R.l[qk / IT][qk % IT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x[xit][jt][8 * rr + e];
                    h8 bh, bl;
                    split8<true, GUARD>(v, bh, bl, &amax);
                    o[jt] = mf(ah, bh, o[jt]);
                    o[jt] = mf(ah, bl, o[jt]);
                    o[jt] = mf(al, bh, o[jt]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int it = 0; it < IT; ++it) {
                    R.h[j][it] = gload8<PH>(R.base_h + pf + j * (IT * 1024) + it * 1024);
                    R.l[j][it] = gload8<PH>(R.base_l + pf + j * (IT * 1024) + it * 1024);
                }
            ring_advance(R, NS);
            sat_note(10);
            if (h == 0) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    f32x4 t = {o[jt][0], o[jt][1], o[jt][2], o[jt][3]};
                    *reinterpret_cast<f32x4 *>(smem + ST::LDS_OUT + (wv * MT + jt * 32 + pl) * 16) = t;
                }
            }
        }
        PNR_T(PH_LIN_OUT);
        __syncthreads();
        PNR_T(PH_BAR_OUT);
        if (tid < MT) {
            const long long g = (long long)tile * MT + tid;
            f32x4 s = *reinterpret_cast<const f32x4 *>(q.bout);
#pragma unroll
            for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x4 *>(smem + ST::LDS_OUT + (w * MT + tid) * 16);
            // models.py:260-265: rgb = sigmoid(out[:3]), sigma = relu(out[3])
            f32x4 res = {1.f / (1.f + expf(-s[0])), 1.f / (1.f + expf(-s[1])), 1.f / (1.f + expf(-s[2])), fmaxf(s[3], 0.f)};
            if (g < q.P) {
                *reinterpret_cast<f32x4 *>(q.out + g * 4) = res;
                if constexpr (GUARD) {
                    const float t = res[0] + res[1] + res[2] + res[3];
                    if (!(fabsf(t) <= 3.0e38f)) sat_bits |= 1u << 11;  // NaN / inf output
                }
            }
        }
        PNR_T(PH_FINAL);
    }
    if constexpr (GUARD) {
        if (sat_bits && q.sat_flag) atomicOr(q.sat_flag, sat_bits);
    }
}

// ================================================================ fp32-class backward chain (training), round 3
// The data-gradient chain of one ResnetFC at split-operand precision, fused like pnr_bwd.hip's bwd_kernel: same tile geometry as
// the forward (persistent 512-thread workgroup, 64-point tiles, wave w owns features 64w..64w+63), the gradient G of the residual
// stream resident in fp32 accumulators, a TRANSPOSED (head, tail) weight stream
//     lin_out^T | fc_1[4]^T fc_0[4]^T fc_1[3]^T fc_0[3]^T | per source view: fc_1[2]^T fc_0[2]^T ... fc_0[0]^T,
// gradient images as (head, tail) f16 pairs in LDS, relu masks = the TRAIN forward's 1-bit words.  Every layer's output gradient
// dY leaves as fp32 rows (natural feature order) at the chain's power-of-two scale: the operands of the weight-gradient GEMMs
// and of the lin_z^T / lin_in^T products, which stay on the split-operand GEMM kernel of pnr_f32.hip.
// (reference: torch autograd through src/model/resnetfc.py:132-184, resnetfc.py:55-62 per block.)

// one thread = one lane's 8-element fragment slice of the head AND the tail stream (16-byte stores)
__global__ void pack_weights_bwd_split_kernel(PnrMlpWeights p, _Float16 *__restrict__ out_h, _Float16 *__restrict__ out_l) {
    const size_t idx8 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx8 >= (size_t)BSRS_TOTAL * IT * 64 * NW) return;
    const int lane = idx8 & 63;
    const int it = (idx8 >> 6) % IT;
    const size_t rest = idx8 / (64 * IT);
    const int rs = rest % BSRS_TOTAL;
    const int wv = rest / BSRS_TOTAL;
    int g = 0;
    while (g + 1 < NBGEMM && rs >= bgemm_offset(g + 1)) ++g;
    const int st = rs - bgemm_offset(g);
    const int i = lane & 31, h = lane >> 5;
    const int f_row = wv * SL + it * 32 + i;  // A-operand row = output row of the transposed GEMM = INPUT feature of the layer
    __attribute__((aligned(16))) _Float16 oh[8], ol[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        if (g == BG_Z2 || g == BG_Z1 || g == BG_Z0) {
            // d z_lat[c] += sum_f dY_b[f] W_z[b][f][c]: rows = latent channel c (natural order, wave w owns 64w..64w+63),
            // K = hidden feature f in the storage order of the gradient image
            const int b = g == BG_Z2 ? 2 : (g == BG_Z1 ? 1 : 0);
            const int f_o = feat_of(st >> 1, st & 1, 8 * h + e);
            v = p.lin_z_w[b][f_o * C_LAT + f_row];
        } else if (g == BG_IN) {
            // d(code | viewdir)[k] = sum_f dY[f] W_in[f][k]: every wave holds the FULL 64 (42 real) output rows and contracts
            // only its own 64 hidden features = storage elements 64w + 16s + 8h + e (K-split, reduced across waves in LDS)
            const int k_row = it * 32 + i;
            const int f_o = feat_of(wv * IT + (st >> 1), st & 1, 8 * h + e);
            if (k_row < D_IN) v = p.lin_in_w[f_o * D_IN + k_row];
        } else if (g == BG_OUT) {
            const int k = st * 16 + h * 8 + e;  // natural order of the 4 network outputs, zero padded
            if (k < D_OUT) v = p.lin_out_w[k * D_HID + f_row];
        } else {
            const int b = 4 - (g - 1) / 2;       // BG_FC1_4, BG_FC0_4, BG_FC1_3, ... -> block index
            const float *w = ((g - 1) & 1) == 0 ? p.fc1_w[b] : p.fc0_w[b];
            const int f_o = feat_of(st >> 1, st & 1, 8 * h + e);  // K index = OUTPUT feature, storage order of the gradient image
            v = w[f_o * D_HID + f_row];
        }
        const _Float16 hh = (_Float16)v;
        oh[e] = hh;
        ol[e] = (_Float16)(v - (float)hh);
    }
    *reinterpret_cast<uint4 *>(out_h + idx8 * 8) = *reinterpret_cast<const uint4 *>(oh);
    *reinterpret_cast<uint4 *>(out_l + idx8 * 8) = *reinterpret_cast<const uint4 *>(ol);
}

struct BwdSplitParams {
    const char *wstream;                 // head blob; the tail blob follows at + BSPACKED_BYTES
    const unsigned long long *d_mask;    // relu bit masks of the forward, [layer][view][tile][thread]
    const float *g_out;                  // (P,4) dL/d(lin_out output), unscaled
    const float *scale_dev;              // device [s, 1/s]: the chain runs at s (pnr_grad_scale)
    long long P;
    int NS, ntiles;
    char *g_fc1[5], *g_fc0[5], *g_x0;    // dY at scale s as (head | tail) 16-bit rows in storage order (copies of the gradient images):
                                         // b < 3 (NS*P,512) [view][point], else (P,512); g_x0 (NS*P,512); tail array behind the head array
    float *d_zlat;                       // (NS*P, 512) fp32, natural channel order, UNSCALED: d(interpolated latent) = sum_b dY_b W_z[b]
    float *d_in;                         // (NS*P, 42) fp32, unscaled: d(positional code | view direction)   (nullable: not written)
    float *mv_ws;                        // several views: per-workgroup scratch for the pooled gradient every view starts from
};

template <bool MV>
__global__ void __launch_bounds__(NTHREADS, NW / 4) bwd_split_kernel(const BwdSplitParams q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef SplitTileT<64> ST;
    constexpr int JT = ST::JT, MT = ST::MT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int NS = MV ? q.NS : 1;
    const uint32_t a_rd0 = ST::A_HI + pl * ROW_ACT + h * 16;
    const uint32_t in_rd0 = ST::LDS_IN + pl * ROW_IN + h * 16;
    const uint32_t a_wr = pl * ROW_ACT + (wv * IT) * 64 + h * 32;
    f16_ovfl_mode<PH>();
    SplitRing R;
    R.base_h = q.wstream + (size_t)wv * (BSRS_TOTAL * IT * 1024) + lane * 16;
    R.base_l = R.base_h + BSPACKED_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            R.h[j][it] = gload8<PH>(R.base_h + j * (IT * 1024) + it * 1024);
            R.l[j][it] = gload8<PH>(R.base_l + j * (IT * 1024) + it * 1024);
        }
    R.pf_rs = 4;
    R.pf_view = 0;
    const float scale = q.scale_dev[0], inv_scale = q.scale_dev[1];
    const size_t mask_layer = (size_t)NS * (size_t)q.ntiles * NTHREADS;

    auto zero = [&](f32x16 (&a)[IT][JT]) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) a[it][jt][r] = 0.f;
    };
    // acc = bit ? acc : 0   /   G += bit ? t : 0
    auto apply_mask = [&](f32x16 (&a)[IT][JT], unsigned long long mk) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const uint32_t m16 = (uint32_t)(mk >> ((it * JT + jt) * 16)) & 0xffffu;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[it][jt][r] = (m16 >> r) & 1u ? a[it][jt][r] : 0.f;
            }
    };
    auto masked_add = [&](f32x16 (&G)[IT][JT], const f32x16 (&t)[IT][JT], unsigned long long mk) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                const uint32_t m16 = (uint32_t)(mk >> ((it * JT + jt) * 16)) & 0xffffu;
#pragma unroll
                for (int r = 0; r < 16; ++r) G[it][jt][r] += (m16 >> r) & 1u ? t[it][jt][r] : 0.f;
            }
    };
    long long rows_left = 0;
    // the gradient image just published (head, tail) -> its 16-bit row sets (operands of the weight-gradient GEMM), whole rows
    auto dump_pair = [&](char *head, long long rows, bool per_view) {
        const size_t total = (size_t)(per_view ? (long long)NS * q.P : q.P) * (D_HID * 2);
        dump_image<MT>(smem, ST::A_HI, head + (size_t)rows * (D_HID * 2), rows_left, wv, lane);
        dump_image<MT>(smem, ST::A_LO, head + total + (size_t)rows * (D_HID * 2), rows_left, wv, lane);
    };
    // the GEMM on the image just published, with the image's copy-out in front of it
    auto gemm_dump = [&](f32x16 (&a)[IT][JT], char *head, long long rows, bool per_view) {
        dump_pair(head, rows, per_view);
        gemm_split<JT, SplitAdvBwd>(a, smem, a_rd0, 32 * ROW_ACT, ST::A_LO - ST::A_HI, KS_BIG / 4, R, NS);
    };
    // reverse of one residual block (resnetfc.py:55-62):  given G = dL/d(x + fc_1(relu(fc_0(relu(x))))),
    //   dY(fc_1) = G ;  d net = (fc_1^T G) . [net > 0] = dY(fc_0) ;  G += (fc_0^T d net) . [x > 0]
    auto bwd_block = [&](f32x16 (&G)[IT][JT], int b, long long rows, size_t mask_off) {
        __syncthreads();  // every wave is done reading the gradient image (previous GEMM)
        write_split<ST, false>(G, smem, a_wr);
        __syncthreads();
        f32x16 t[IT][JT];
        const unsigned long long mk_n = (q.d_mask + (size_t)(2 * b + 1) * mask_layer)[mask_off];
        zero(t);
        gemm_dump(t, q.g_fc1[b], rows, b < COMBINE_LAYER);
        apply_mask(t, mk_n);
        __syncthreads();
        write_split<ST, false>(t, smem, a_wr);
        __syncthreads();
        const unsigned long long mk_a = (q.d_mask + (size_t)(2 * b) * mask_layer)[mask_off];
        zero(t);
        gemm_dump(t, q.g_fc0[b], rows, b < COMBINE_LAYER);
        masked_add(G, t, mk_a);
    };

    for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        rows_left = q.P - (long long)tile * MT;
        const long long rows_pooled = (long long)tile * MT;
        const size_t mask_pooled = (size_t)tile * NTHREADS + tid;
        __syncthreads();  // previous tile: readers of the staging rows / gradient image are done
        {   // stage s * g_out as (head, tail) operand rows [point][64] (4 real values, zero padded)
            const int row = tid >> 3, chunk = tid & 7;
            const long long g = rows_pooled + row;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (chunk == 0 && g < q.P) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(q.g_out + g * 4) * scale;
                v[0] = gv[0]; v[1] = gv[1]; v[2] = gv[2]; v[3] = gv[3];
            }
            h8 hi, lo;
            split8<false>(v, hi, lo);
            *reinterpret_cast<h8 *>(smem + ST::LDS_IN + row * ROW_IN + chunk * 16) = hi;
            *reinterpret_cast<h8 *>(smem + ST::LDS_IN + ST::IN_LO_DELTA + row * ROW_IN + chunk * 16) = lo;
        }
        __syncthreads();
        f32x16 G[IT][JT];
        {
            const unsigned long long mk = (q.d_mask + (size_t)10 * mask_layer)[mask_pooled];
            zero(G);
            gemm_split<JT, SplitAdvBwd>(G, smem, in_rd0, 32 * ROW_IN, ST::IN_LO_DELTA, KS_IN / 4, R, NS);  // lin_out^T g_out
            apply_mask(G, mk);                                                                             // . [x5 > 0]
        }
#pragma unroll 1
        for (int b = N_BLOCKS - 1; b >= COMBINE_LAYER; --b) bwd_block(G, b, rows_pooled, mask_pooled);
        // backward of the view mean (util.py:461-466): every view starts from G / NS, parked in the per-workgroup scratch
        [[maybe_unused]] f32x4 *gws = MV ? reinterpret_cast<f32x4 *>(q.mv_ws) + (size_t)blockIdx.x * (IT * JT * 4 * NTHREADS) + tid : nullptr;
        if constexpr (MV) {
            const float inv = 1.f / (float)NS;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4 v = {G[it][jt][4 * k], G[it][jt][4 * k + 1], G[it][jt][4 * k + 2], G[it][jt][4 * k + 3]};
                        gws[((it * JT + jt) * 4 + k) * NTHREADS] = v * inv;
                    }
        }
#pragma unroll 1
        for (int view = 0; view < NS; ++view) {
            if constexpr (MV) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = gws[((it * JT + jt) * 4 + k) * NTHREADS];
                            G[it][jt][4 * k] = v[0]; G[it][jt][4 * k + 1] = v[1]; G[it][jt][4 * k + 2] = v[2]; G[it][jt][4 * k + 3] = v[3];
                        }
            }
            const long long rows_view = (long long)view * q.P + rows_pooled;
            const size_t mask_view = mask_pooled + (size_t)view * (size_t)q.ntiles * NTHREADS;
#pragma unroll 1
            for (int b = COMBINE_LAYER - 1; b >= 0; --b) bwd_block(G, b, rows_view, mask_view);
            // ---- d z_lat = sum_b dY_b W_z[b] and d(code) = dY_0 W_in (resnetfc.py:147,175-180 backward): four more transposed-
            // stream GEMMs on gradient images this tile has just produced.  dY_2, dY_1 (= g_fc1[1], g_fc1[0]) come back from the
            // rows this workgroup copied out a moment ago (L2-resident); dY_0 = G is still in registers.
            f32x16 Z[IT][JT];
            zero(Z);
#pragma unroll 1
            for (int b = COMBINE_LAYER - 1; b >= 1; --b) {
                __threadfence_block();
                __syncthreads();  // the image's readers are done; the rows this tile dumped are visible
                {
                    const char *src = q.g_fc1[b - 1] + (size_t)rows_view * (D_HID * 2);
                    const size_t total = (size_t)NS * (size_t)q.P * (D_HID * 2);
#pragma unroll
                    for (int u = 0; u < MT / NW; ++u) {
                        const int row = wv * (MT / NW) + u;
                        u32x4 vh = {0, 0, 0, 0}, vl = vh;
                        if (row < rows_left) {
                            vh = *reinterpret_cast<const u32x4 *>(src + (size_t)row * (D_HID * 2) + lane * 16);
                            vl = *reinterpret_cast<const u32x4 *>(src + total + (size_t)row * (D_HID * 2) + lane * 16);
                        }
                        *reinterpret_cast<u32x4 *>(smem + ST::A_HI + row * ROW_ACT + lane * 16) = vh;
                        *reinterpret_cast<u32x4 *>(smem + ST::A_LO + row * ROW_ACT + lane * 16) = vl;
                    }
                }
                __syncthreads();
                gemm_split<JT, SplitAdvBwd>(Z, smem, a_rd0, 32 * ROW_ACT, ST::A_LO - ST::A_HI, KS_BIG / 4, R, NS);  // lin_z[b]^T dY_b
            }
            __syncthreads();
            write_split<ST, false>(G, smem, a_wr);  // dY of lin_in and lin_z[0]
            __syncthreads();
            gemm_dump(Z, q.g_x0, rows_view, true);                                                                   // lin_z[0]^T dY_0
            {   // accumulator (channel 64 wv + 32 it + (r&3) + 8(r>>2) + 4h, point) -> fp32 rows, 16-byte pieces, out of the scaled domain
                float *dst = q.d_zlat + ((size_t)rows_view + pl) * C_LAT + (wv * IT) * 32 + 4 * h;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    if (jt * 32 + pl >= rows_left) continue;
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = {Z[it][jt][4 * k], Z[it][jt][4 * k + 1], Z[it][jt][4 * k + 2], Z[it][jt][4 * k + 3]};
                            __builtin_nontemporal_store(v * inv_scale, reinterpret_cast<f32x4 *>(dst + (size_t)jt * 32 * C_LAT + it * 32 + 8 * k));
                        }
                }
            }
            // lin_in^T, K-split: this wave's 64 hidden features = bytes [128 wv, 128 wv + 128) of every image row
            zero(Z);
            gemm_split<JT, SplitAdvBwd>(Z, smem, a_rd0 + wv * 128, 32 * ROW_ACT, ST::A_LO - ST::A_HI, KS_IN / 4, R, NS);
            __syncthreads();  // every wave is done with the images: their space takes the partials
            {
                // partial[w][point][k], 66-float rows: lanes of a half-wave write consecutive points -> distinct banks
                float *part = reinterpret_cast<float *>(smem) + (size_t)wv * (MT * 66);
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            part[(jt * 32 + pl) * 66 + it * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = Z[it][jt][r];
            }
            __syncthreads();
            if (q.d_in) {
                const int pnt = tid >> 3, k0 = (tid & 7) * 8;
                if (pnt < rows_left) {
#pragma unroll
                    for (int k = k0; k < k0 + 8; ++k) {
                        if (k >= D_IN) break;
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) sum += reinterpret_cast<const float *>(smem)[(size_t)w * (MT * 66) + pnt * 66 + k];
                        q.d_in[((size_t)rows_view + pnt) * D_IN + k] = sum * inv_scale;
                    }
                }
            }
            __syncthreads();  // the partials are consumed before the next view / tile reuses the space
        }
    }
}

static int bwd_split_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, c = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c = prop.multiProcessorCount;
        n = c;
    }
    return n;
}


static int split_launch(const PnrScene *s, const void *packed, const void *tables, EvalParams &q, bool rays, hipStream_t st) {
    if (!s || !packed || !tables || !q.out) return pnr_fail(PNR_E_INVALID, "pnr_eval_split: null argument");
    if (s->SB <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_eval_split: bad scene shape");
    if (s->NS <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_split: NS must be positive");
    if (!(s->n_focal == 1 || s->n_focal == s->SB) || !(s->n_c == 1 || s->n_c == s->SB))
        return pnr_fail(PNR_E_INVALID, "pnr_eval_split: focal / c must have 1 or SB rows");
    if (q.P == 0) return PNR_OK;
    if (q.P > 0x7fffff80LL) return pnr_fail(PNR_E_INVALID, "pnr_eval_split: too many points (P must stay below 2^31)");
    if ((long long)s->SB * s->NS * s->Hl * s->Wl * C_LAT > 0xffffffffLL)
        return pnr_fail(PNR_E_INVALID, "pnr_eval_split: feature grid too large (SB*NS*Hl*Wl*512 must stay below 2^32 elements)");
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.wstream = (const char *)packed;
    q.bias = (const float *)((const char *)packed + BIAS_OFFSET_BYTES);
    q.bout = (const float *)((const char *)packed + BOUT_OFFSET_BYTES);
    q.tables = (const char *)tables;
    q.table_stride = (long long)s->SB * s->NS * s->Hl * s->Wl * C_LAT;
    const bool mv = s->NS > 1;
    int dev = 0, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        ncu = prop.multiProcessorCount;
    // tile: 64 points (the two full operand images fill the LDS; the 96-point K-half-staged form of round 3 measured 15 % slower,
    // profiles/r03_split_kernel_ab.txt, and was removed with the other experiment code in round 4)
    const int MT = mv ? SplitTileT<SPLIT_MV_TILE>::MT : SplitTileT<64>::MT;
    const int lds = mv ? SplitTileT<SPLIT_MV_TILE>::LDS_TOTAL : SplitTileT<64>::LDS_TOTAL;
    const long long nt = (q.P + MT - 1) / MT;
    q.ntiles = (int)nt;
    const int grid = (int)(nt < ncu ? nt : ncu);
    if (mv && SPLIT_MV_TILE == 64) {
        q.mv_ws = mv_scratch(st, (size_t)ncu * 64 * D_HID * sizeof(float));
        if (!q.mv_ws) return pnr_fail(PNR_E_HIP, "pnr_eval_split: cannot allocate the multi-view pooling scratch (32 MiB)");
    }
    auto k = mv ? (rays ? eval_split_kernel<true, true> : eval_split_kernel<false, true>)
                : (rays ? eval_split_kernel<true, false> : eval_split_kernel<false, false>);
    // fp16-range guard armed on this host thread (pnr_saturation_guard): the instantiations that follow the largest value
    // entering every operand image; word `slot` of the caller's flag array (0: a coarse-network launch, 1: a fine-network one)
    unsigned int *guard = saturation_guard_word();
    q.sat_flag = guard;
    if (guard)
        k = mv ? (rays ? eval_split_kernel<true, true, false, false, true> : eval_split_kernel<false, true, false, false, true>)
               : (rays ? eval_split_kernel<true, false, false, false, true> : eval_split_kernel<false, false, false, false, true>);
    if (q.f_x5) {  // training forward: the same kernel + fp32 rows of what the backward keeps
        if (!rays) return pnr_fail(PNR_E_INVALID, "pnr_eval_split: the training instantiation takes ray samples on 64-point tiles");
        k = mv ? (guard ? eval_split_kernel<true, true, false, true, true> : eval_split_kernel<true, true, false, true>)
               : (guard ? eval_split_kernel<true, false, false, true, true> : eval_split_kernel<true, false, false, true>);
    }
    if (q.tim) {  // diagnostic instantiation (pnr_debug_phase_timing_split): single view, rays
        if (mv || !rays) return pnr_fail(PNR_E_INVALID, "phase timing: single-view ray launches only");
        k = eval_split_kernel<true, false, true>;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(eval_split_kernel)");
    {
        q.n_xcd = device_xcd_count();
        ProfileScope prof(st);  // HIP events around the launch on ITS stream when pnr_profile_enable(1) is active
        hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, q);
    }
    return pnr_check_launch("eval_split_kernel");
}

size_t bwd_split_packed_bytes() { return 2 * BSPACKED_BYTES; }

int pack_bwd_split(const PnrMlpWeights *w, void *packed, hipStream_t st) {
    if (!w || !packed) return pnr_fail(PNR_E_INVALID, "pack_bwd_split: null argument");
    const size_t n8 = (size_t)BSRS_TOTAL * IT * 64 * NW;
    hipLaunchKernelGGL(pack_weights_bwd_split_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, *w, (_Float16 *)packed,
                       (_Float16 *)((char *)packed + BSPACKED_BYTES));
    return pnr_check_launch("pack_weights_bwd_split_kernel");
}

int mlp_backward_split_chain(const void *packed_bwd_split, const unsigned long long *masks, const float *g_out, const float *scale_dev,
                             long long P, int NS, void *const *g_fc1, void *const *g_fc0, void *g_x0, float *d_zlat, float *d_in,
                             hipStream_t st) {
    if (!packed_bwd_split || !masks || !g_out || !scale_dev || !g_fc1 || !g_fc0 || !g_x0 || !d_zlat || P <= 0 || NS <= 0)
        return pnr_fail(PNR_E_INVALID, "mlp_backward_split_chain: bad argument");
    BwdSplitParams q = {};
    q.wstream = (const char *)packed_bwd_split; q.d_mask = masks; q.g_out = g_out; q.scale_dev = scale_dev; q.P = P; q.NS = NS;
    const long long nt = (P + 63) / 64;
    if (nt * NTHREADS * NS > 0xffffffffLL) return pnr_fail(PNR_E_INVALID, "mlp_backward_split_chain: too many points");
    q.ntiles = (int)nt;
    for (int b = 0; b < 5; ++b) {
        if (!g_fc1[b] || !g_fc0[b]) return pnr_fail(PNR_E_INVALID, "mlp_backward_split_chain: null gradient buffer");
        q.g_fc1[b] = (char *)g_fc1[b]; q.g_fc0[b] = (char *)g_fc0[b];
    }
    q.g_x0 = (char *)g_x0; q.d_zlat = d_zlat; q.d_in = d_in;
    const int cus = bwd_split_cus();
    const int grid = (int)(nt < cus ? nt : cus);
    const bool mv = NS > 1;
    if (mv) {
        q.mv_ws = mv_scratch(st, (size_t)cus * 64 * D_HID * sizeof(float));
        if (!q.mv_ws) return pnr_fail(PNR_E_HIP, "mlp_backward_split_chain: cannot allocate the multi-view scratch");
    }
    auto k = mv ? bwd_split_kernel<true> : bwd_split_kernel<false>;
    const int lds = SplitTileT<64>::LDS_TOTAL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(bwd_split_kernel)");
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, q);
    return pnr_check_launch("bwd_split_kernel");
}

}  // namespace pnr

extern "C" size_t pnr_packed_mlp_split_bytes(void) { return 2 * pnr::PACKED_BYTES; }

int pnr::eval_samples_split_src(const PnrScene *scene, const void *packed_split, const void *tables_f32, const RaySrc &src,
                                const float *z, int R, int rays_per_obj, int K, float *rgbsigma, hipStream_t stream) {
    if (R < 0 || K <= 0 || rays_per_obj <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split: bad sizes");
    if (R > 0 && ((!src.rays && !src.poses) || !z)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split: null rays/z");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split: R != SB * rays_per_obj");
    pnr::EvalParams q = {};
    q.rays = src.rays; q.cam = src; q.cam.rays = nullptr;
    q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma;
    return pnr::split_launch(scene, packed_split, tables_f32, q, true, stream);
}

// training forward of the fp32-class path (pnr_f32.hip, pnr_eval_ray_samples_split_train): outputs + what the backward keeps
int pnr::eval_samples_split_train(const PnrScene *scene, const void *packed_split, const void *tables_f32, const float *rays,
                                  const float *z, int R, int rays_per_obj, int K, float *rgbsigma, void *const *img_a, void *const *img_n,
                                  float *x5, void *masks, hipStream_t stream) {
    if (!rays || !z || !img_a || !img_n || !x5 || !masks) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: null argument");
    if (SPLIT_MV_TILE != 64) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: built with 32-point multi-view tiles");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma;
    for (int b = 0; b < 5; ++b) {
        if (!img_a[b] || !img_n[b]) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: null operand buffer");
        q.s_a[b] = (char *)img_a[b]; q.s_n[b] = (char *)img_n[b];
    }
    q.f_x5 = x5;
    q.d_mask = (unsigned long long *)masks;
    return pnr::split_launch(scene, packed_split, tables_f32, q, true, stream);
}

extern "C" int pnr_eval_ray_samples_split(const PnrScene *scene, const void *packed_split, const void *tables_f32,
                                          const float *rays, const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                          void *stream) {
    pnr::RaySrc src = {};
    src.rays = rays;
    return pnr::eval_samples_split_src(scene, packed_split, tables_f32, src, z, R, rays_per_obj, K, rgbsigma, (hipStream_t)stream);
}

// test/diagnostic hook (not in the public header): per-phase cycle totals of every wave of workgroup 0, one single-view launch
extern "C" int pnr_debug_phase_timing_split(const PnrScene *scene, const void *packed_split, const void *tables_f32, const float *rays,
                                            const float *z, int R, int rays_per_obj, int K, float *rgbsigma, unsigned long long *tim,
                                            void *stream) {
    if (!tim || !rays || !z) return pnr_fail(PNR_E_INVALID, "pnr_debug_phase_timing_split: null argument");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma; q.tim = tim;
    return pnr::split_launch(scene, packed_split, tables_f32, q, true, (hipStream_t)stream);
}

extern "C" int pnr_eval_points_split(const PnrScene *scene, const void *packed_split, const void *tables_f32, const float *xyz,
                                     const float *viewdirs, int B, float *rgbsigma, void *stream) {
    if (B < 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_points_split: bad sizes");
    if (B > 0 && (!xyz || !viewdirs)) return pnr_fail(PNR_E_INVALID, "pnr_eval_points_split: null xyz/viewdirs");
    pnr::EvalParams q = {};
    q.xyz = xyz; q.viewdirs = viewdirs; q.K = 1; q.per_obj = B > 0 ? B : 1;
    q.P = scene ? (long long)scene->SB * B : 0; q.out = rgbsigma;
    return pnr::split_launch(scene, packed_split, tables_f32, q, false, (hipStream_t)stream);
}
