// pnr_device.h -- device-side building blocks shared by the forward (pnr_mlp.hip) and backward
// (pnr_bwd.hip) fused network kernels: MFMA traits, 16-bit packing, the weight prefetch ring,
// the tile GEMM, activation-image writes.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_layout.h"
#include "pnr_raysrc.h"

namespace pnr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PREC> struct Prec;
template <> struct Prec<PNR_PREC_F16> {
    typedef _Float16 T;
    typedef f16x8 T8;
    typedef f16x2 T2;
    static constexpr float kMaxFinite = 65504.f;
    static constexpr bool kIsF16 = true;
    static __device__ __forceinline__ f32x16 mfma(T8 a, T8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Prec<PNR_PREC_BF16> {
    typedef __bf16 T;
    typedef bf16x8 T8;
    typedef bf16x2 T2;
    static constexpr float kMaxFinite = 3.3895313892515355e38f;  // largest finite bf16
    static constexpr bool kIsF16 = false;
    static __device__ __forceinline__ f32x16 mfma(T8 a, T8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

struct EvalParams {
    // scene (PnrScene)
    const float *latent, *poses, *focal, *c;
    int SB, NS, Hl, Wl, n_focal, n_c;
    float img_w, img_h;
    // packed network
    const char *wstream;
    const float *bias, *bout;
    // folded inference: 3 per-texel tables (16-bit, NHWC like the grid, hidden features in storage order)
    const char *tables;
    long long table_stride;  // elements per table
    // points: variant A (rays + z) or B (xyz + viewdirs)
    const float *rays, *z, *xyz, *viewdirs;
    RaySrc cam;        // variant A with rays == NULL: rays are regenerated from the camera (cam.rays stays NULL)
    int K;             // samples per ray (A)
    int per_obj;       // rays per object (A) or points per object (B)
    long long P;       // total points
    int ntiles;
    int n_xcd;         // XCDs workgroups are dealt to round-robin (device_xcd_count(); 0 / 1: plain grid-stride tile order)
    float *out;        // (P,4)
    float *mv_ws;      // multi-view: per-workgroup scratch for the parked view sum (MT x 512 fp32 each)
    float *dbg;        // optional debug dump of the final residual stream x (P,512), may be null
    unsigned long long *tim;  // phase-timing accumulators (TIMING instantiation only)
    // TRAIN instantiation: 16-bit row-major dumps of every linear layer's input operand
    // (rows = view*P + point for the per-view layers, point for the pooled ones)
    char *d_in;    // (NS*P, 64)   positional code + view direction (natural order, zero padded)
    unsigned long long *d_mask;  // relu bit masks of the 11 dumped activations, [layer][view][tile][thread] (see relu_bits)
    char *d_z;     // (NS*P, 512)  interpolated latent (natural channel order)
    char *d_a[5];  // relu(x) in front of blocks[b].fc_0, storage order; b<3: (NS*P,512), else (P,512)
    char *d_n[5];  // relu(net) in front of blocks[b].fc_1, same shapes
    char *d_x5;    // (P,512) relu(x) in front of lin_out
    // TRAIN instantiation of the split-operand kernel (pnr_split.hip): the (head, tail) operand images of every linear, copied
    // out of LDS as 16-bit rows in storage order (s_a[b]: relu(x) entering blocks[b].fc_0, s_n[b]: relu(net) entering fc_1;
    // b < 3: NS*P rows [view][point], else P rows; the tail array follows the head array), the stream in front of lin_out as
    // fp32 rows in natural feature order (f_x5), and the relu masks in d_mask
    char *s_a[5], *s_n[5];
    float *f_x5;
    // GUARD instantiation of the split-operand kernel: one word that collects "layer l saw a value beyond the fp16 range" bits
    unsigned int *sat_flag;
};

// Tile order of the fused evaluation kernels (eval_kernel, eval_split_kernel).  Workgroup b runs on XCD b % n_xcd (the
// dispatcher deals workgroups to the XCDs round-robin -- relied on for speed only, never for a result): every XCD takes ONE
// contiguous n_xcd-th of the tiles and its CUs walk it side by side, so the CUs that share an L2 look up neighbouring rays'
// texels at the same time.  n_xcd comes from the host (device_xcd_count(): hipDeviceAttributeNumberOfXccs of the current
// partition mode, PIXELNERF_XCD_COUNT overrides); 0 / 1, or a grid that is not a multiple of it, give the plain grid-stride
// order.  Every tile is visited exactly once either way (tests/test_hip_split.py runs both orders).
// Same-box A/B against grid-stride (profiles/r05_split_kernel_ab.txt): sn64 +0.2 %, srn_car +0.4 %, DTU +2.4 % (f16x3).
struct TileRange { int begin, end, step; };
__device__ __forceinline__ TileRange tile_range(int ntiles, int n_xcd) {
    TileRange r = {(int)blockIdx.x, ntiles, (int)gridDim.x};
    if (n_xcd > 1 && gridDim.x % (unsigned)n_xcd == 0) {
        const int chunk = (ntiles + n_xcd - 1) / n_xcd, xcd = blockIdx.x % (unsigned)n_xcd;
        r.begin = xcd * chunk + blockIdx.x / (unsigned)n_xcd;
        r.end = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
        r.step = gridDim.x / (unsigned)n_xcd;
    }
    return r;
}

// phase ids of the TIMING instantiation (wave 0 of workgroup 0, s_memtime ticks)
enum Phase { PH_SYNC_TOP = 0, PH_GEOMETRY, PH_GATHER, PH_GEMM_IN_Z0, PH_BAR1, PH_WRITE_X, PH_BAR2, PH_GEMM_FC0, PH_BAR3,
             PH_WRITE_NET, PH_BAR4, PH_GEMM_FC1_Z, PH_LIN_OUT, PH_BAR_OUT, PH_FINAL, PH_TABLE,
             // sub-phases of the fp32-class kernel's own-K stages (round 6; they split PH_WRITE_X / PH_WRITE_NET further: the
             // stage's remaining time -- waiting for the image stores, loop exit -- stays under those two)
             PH_OWN_BIAS, PH_OWN_PROLOGUE, PH_OWN_KSTEPS, NPHASE };
#define PNR_T(ph)                                                         \
    do {                                                                  \
        if constexpr (TIMING) {                                           \
            if ((tid & 63) == 0 && blockIdx.x == 0) {                     \
                const unsigned long long t_ = __builtin_readcyclecounter(); \
                atomicAdd(&tim[(tid >> 6) * NPHASE + ph], t_ - tlast);    \
                tlast = t_;                                               \
            }                                                             \
        }                                                                 \
    } while (0)

__device__ __forceinline__ uint32_t pack2(float a, float b, _Float16) {
    f32x2 v = {a, b};
    f16x2 h = __builtin_convertvector(v, f16x2);
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint32_t pack2(float a, float b, __bf16) {
    f32x2 v = {a, b};
    bf16x2 h = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(uint32_t, h);
}

// MODE.FP16_OVFL = 1 for the rest of the wave's life (hwreg(HW_REG_MODE, 23, 1)): f16 conversions saturate
template <typename P> __device__ __forceinline__ void f16_ovfl_mode() {
    if constexpr (P::kIsF16) __builtin_amdgcn_s_setreg(1473, 1);
}

// 8 fp32 -> 8 x 16-bit, optional relu
template <typename P, bool RELU>
__device__ __forceinline__ typename P::T8 pack8(float v0, float v1, float v2, float v3, float v4, float v5,
                                                float v6, float v7) {
    // f16: the fused kernels run with MODE.FP16_OVFL set (f16_ovfl_mode()), so v_cvt_pk_f16_f32 clamps an
    // overflowing result to +-65504 instead of producing inf: relu + saturation = convert, then one v_pk_max_f16
    // per pair (bit-identical to clamping in fp32 first; 2 VALU per pair instead of 3)
    if constexpr (RELU && P::kIsF16) {
        const f16x2 z = {(_Float16)0, (_Float16)0};
        const f16x2 h0 = __builtin_elementwise_max(__builtin_convertvector((f32x2){v0, v1}, f16x2), z);
        const f16x2 h1 = __builtin_elementwise_max(__builtin_convertvector((f32x2){v2, v3}, f16x2), z);
        const f16x2 h2 = __builtin_elementwise_max(__builtin_convertvector((f32x2){v4, v5}, f16x2), z);
        const f16x2 h3 = __builtin_elementwise_max(__builtin_convertvector((f32x2){v6, v7}, f16x2), z);
        const u32x4 u = {__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1), __builtin_bit_cast(uint32_t, h2),
                         __builtin_bit_cast(uint32_t, h3)};
        return __builtin_bit_cast(typename P::T8, u);
    }
    if (RELU) {
        // relu fused with saturation to the operand type's largest finite value: one v_med3_f32 per
        // element, and an fp16 activation can never become inf (65504 for f16; bf16 has fp32's range)
        const float hi = P::kMaxFinite;
        v0 = __builtin_amdgcn_fmed3f(v0, 0.f, hi); v1 = __builtin_amdgcn_fmed3f(v1, 0.f, hi);
        v2 = __builtin_amdgcn_fmed3f(v2, 0.f, hi); v3 = __builtin_amdgcn_fmed3f(v3, 0.f, hi);
        v4 = __builtin_amdgcn_fmed3f(v4, 0.f, hi); v5 = __builtin_amdgcn_fmed3f(v5, 0.f, hi);
        v6 = __builtin_amdgcn_fmed3f(v6, 0.f, hi); v7 = __builtin_amdgcn_fmed3f(v7, 0.f, hi);
    }
    typename P::T t = (typename P::T)0;
    u32x4 u = {pack2(v0, v1, t), pack2(v2, v3, t), pack2(v4, v5, t), pack2(v6, v7, t)};
    return __builtin_bit_cast(typename P::T8, u);
}

// ---------------------------------------------------------------- weight prefetch ring
// ring slot j holds the IT fragments of stream position (consumed position + j); every consumed
// slot is immediately refilled with position +4.  The prefetch cursor follows the consumption
// order [per-view segment] x NS, [tail segment], and wraps to the start for the next tile.
template <typename P> struct Ring {
    typename P::T8 r[4][IT];
    const char *wave_base;  // this wave's stream + lane*16
    int pf_rs;              // ring step the next refill (slot 0) will fetch
    int pf_view;
};

template <typename P> __device__ __forceinline__ typename P::T8 gload8(const char *p) {
    return *reinterpret_cast<const typename P::T8 *>(p);
}
template <typename P> __device__ __forceinline__ typename P::T8 lds8(const char *smem, uint32_t a) {
    return *reinterpret_cast<const typename P::T8 *>(smem + a);
}

// Prefetch-cursor policy: the stream segment [LOOP_LO, LOOP_HI) is consumed once per source view
// (NS times), everything else once per tile; at TOTAL the cursor wraps to 0 for the next tile.
//   forward : [0, RS_VIEW_END) x NS, then the pooled tail          -> Advance<0, RS_VIEW_END, RS_TOTAL>
//   backward: pooled head once, then [BRS_HEAD_END, BRS_TOTAL) x NS -> Advance<BRS_HEAD_END, BRS_TOTAL, BRS_TOTAL>
template <int LOOP_LO, int LOOP_HI, int TOTAL> struct Advance {
    template <typename P> static __device__ __forceinline__ void step4(Ring<P> &R, int NS) {
        int rs = R.pf_rs + 4, v = R.pf_view;
        if (rs == LOOP_HI && v + 1 < NS) {
            v += 1; rs = LOOP_LO;
        } else if (rs == TOTAL) {
            rs = 0; v = 0;
        }
        R.pf_rs = rs; R.pf_view = v;
    }
};
typedef Advance<0, RS_VIEW_END, RS_TOTAL> AdvanceFwd;

// Training dumps riding inside a GEMM (round 3): the copy of the just-published activation / gradient image to its (rows,512)
// dump -- one 1 KiB row per wave and loop body, 8 bodies = the wave's 8 rows of a 64-point tile -- is issued from INSIDE the GEMM
// loop, the LDS read of row b in body b and its global store in body b+1, so both sit in the shadow of that body's MFMAs
// instead of in front of the GEMM (dump_image: 8 LDS reads + 8 stores per wave before the first MFMA of every dumped GEMM).
struct DumpJob {
    const char *src;      // LDS image base (row stride ROW_ACT)
    char *dst;            // first row of this tile in the dump
    long long rows_left;  // rows of the dump from there on
    int wv, lane;
};

// acc[it][jt] += W-fragments (ring) x B-fragments (LDS rows baddr0/baddr1, 32 B per k-step),
// nbody*4 k-steps.
template <typename P, typename ADV = AdvanceFwd, bool DUMP = false, int JT_>
__device__ __forceinline__ void gemm(f32x16 (&acc)[IT][JT_], const char *smem, uint32_t baddr0, uint32_t baddr1,
                                     int nbody, Ring<P> &R, int NS, const DumpJob *dj = nullptr) {
    [[maybe_unused]] u32x4 dump_v = {0, 0, 0, 0};
    // column tile jt reads rows baddr0 + jt * (baddr1 - baddr0): 32 point rows further down the image
    const uint32_t jstride = baddr1 - baddr0;
    typename P::T8 b[2][JT_];
#pragma unroll
    for (int jt = 0; jt < JT_; ++jt) b[0][jt] = lds8<P>(smem, baddr0 + jt * jstride);
#pragma unroll 1
    for (int body = 0; body < nbody; ++body) {
        if constexpr (DUMP) {  // nbody == MT / NW == 8 rows per wave (64-point tiles)
            const int row = dj->wv * 8 + body;
            if (body > 0 && row - 1 < dj->rows_left)
                *reinterpret_cast<u32x4 *>(dj->dst + (size_t)(row - 1) * (D_HID * 2) + dj->lane * 16) = dump_v;
            dump_v = *reinterpret_cast<const u32x4 *>(dj->src + row * ROW_ACT + dj->lane * 16);
        }

        const char *pf = R.wave_base + (size_t)R.pf_rs * (IT * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            // intended step order: LDS reads for step+1 | IT*JT MFMAs of this step | refill this ring
            // slot (step+4).  hipcc re-orders this (it batches the refills behind the last MFMA
            // of the body); pinning the order with sched_barrier (-DPNR_PIN_SCHEDULE) gives the
            // textbook stream but measured 2-3 % SLOWER (profiles/r01_gemm_experiments.md), so
            // the compiler's schedule is the default.
#pragma unroll
            for (int jt = 0; jt < JT_; ++jt) {
                b[cur ^ 1][jt] = lds8<P>(smem, baddr0 + jt * jstride + (j + 1) * 32);
            }
            typename P::T8 a[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) a[it] = R.r[j][it];
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT_; ++jt) acc[it][jt] = P::mfma(a[it], b[cur][jt], acc[it][jt]);
#pragma unroll
            for (int it = 0; it < IT; ++it) R.r[j][it] = gload8<P>(pf + j * (IT * 1024) + it * 1024);
        }
        baddr0 += 128;
        ADV::step4(R, NS);
    }
    if constexpr (DUMP) {
        const int row = dj->wv * 8 + nbody - 1;
        if (row < dj->rows_left) *reinterpret_cast<u32x4 *>(dj->dst + (size_t)row * (D_HID * 2) + dj->lane * 16) = dump_v;
    }
}

// [relu](acc) -> 16-bit -> activation image.  Lane (p,h) writes registers 0..15 of feature tile
// T = wave*IT+it as 32 contiguous bytes at element offset 32T + 16h of its point row ("storage
// order").  DUMP additionally stores the same 32 bytes to a row-major (rows,512) 16-bit array in
// HBM (training: operands of the weight-gradient GEMMs and relu masks of the backward chain);
// dump_lane = array + ((first row of the tile + p)*512 + 32*wave*IT + 16h) elements, valid[jt]
// guards rows beyond the last point.
// relu bit masks for the backward chain: bit (it*JT + jt)*16 + r of a thread's 64-bit word = "register r of its
// accumulator tile (it, jt) was dumped as a non-zero 16-bit value" (the raw bits, like the dump itself would be tested).
// 64 bytes per point and layer instead of re-reading the 1 KiB dump row: the backward kernel spent a third of its time
// waiting for those rows.  v = the 8 packed 16-bit values of registers 8*half .. 8*half+7.
template <typename T8>
__device__ __forceinline__ uint32_t nonzero_bits8(const T8 &v) {
    const u32x4 w = __builtin_bit_cast(u32x4, v);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m |= ((w[k] & 0xffffu) ? 1u : 0u) << (2 * k);
        m |= ((w[k] >> 16) ? 1u : 0u) << (2 * k + 1);
    }
    return m;
}

template <typename P, bool RELU = true, bool DUMP = false, int JT_>
__device__ __forceinline__ void write_act(const f32x16 (&acc)[IT][JT_], char *smem, uint32_t waddr,
                                          char *dump_lane = nullptr, const bool *valid = nullptr,
                                          unsigned long long *mask_slot = nullptr) {
    unsigned long long mbits = 0ull;
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT_; ++jt) {
            const f32x16 &a = acc[it][jt];
            const uint32_t ad = waddr + jt * 32 * ROW_ACT + it * 64;
            const typename P::T8 lo = pack8<P, RELU>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
            const typename P::T8 hi = pack8<P, RELU>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
            *reinterpret_cast<typename P::T8 *>(smem + ad) = lo;
            *reinterpret_cast<typename P::T8 *>(smem + ad + 16) = hi;
            if (DUMP) {
                if (dump_lane && valid[jt]) {  // per-lane fragment stores (dump_lane == null: the caller copies the image, dump_image)
                    char *d = dump_lane + (size_t)jt * 32 * (D_HID * 2) + it * 64;
                    // plain stores: streaming (nontemporal) stores for the dumps + loads for the masks measured 5 % SLOWER
                    // on the training step -- the dumps are served to the backward kernels from the Infinity Cache
                    *reinterpret_cast<typename P::T8 *>(d) = lo;
                    *reinterpret_cast<typename P::T8 *>(d + 16) = hi;
                }
                if (RELU && mask_slot && IT * JT_ * 16 <= 64)
                    mbits |= (unsigned long long)(nonzero_bits8(lo) | (nonzero_bits8(hi) << 8)) << ((it * JT_ + jt) * 16);
            }
        }
    if (DUMP && RELU && mask_slot) *mask_slot = mbits;
}

// Cooperative copy of an activation / gradient image (MT rows of 1 KiB, row stride ROW_ACT) to its (rows,512) 16-bit dump:
// every wave instruction moves one whole row, 1 KiB contiguous in LDS and in HBM (8 full lines), instead of the 64 scattered
// 16-byte pieces per instruction of the per-lane fragment stores.  Call after the barrier that publishes the image.
// dst_tile = first row of this tile in the dump; rows_left = rows of the dump from there on.
template <int MT_>
__device__ __forceinline__ void dump_image(const char *smem, uint32_t image, char *dst_tile, long long rows_left, int wv, int lane) {
    constexpr int RPW = MT_ / NW;  // rows per wave
    static_assert(RPW % 4 == 0, "copied in batches of 4 rows");
#pragma unroll
    for (int u0 = 0; u0 < RPW; u0 += 4) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = *reinterpret_cast<const u32x4 *>(smem + image + (wv * RPW + u0 + u) * ROW_ACT + lane * 16);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wv * RPW + u0 + u;
            // NON-TEMPORAL store (round 5): the operand / gradient rows are written once and read by a LATER kernel, 1.3 MB per tile
            // against the 4 MB L2 the 32 CUs of an XCD share for the weight stream.  As ordinary stores they evicted the weight
            // fragments the neighbouring CUs were about to read; streamed past the L2 the training forward runs 746 -> 617 us and
            // the fp32-class backward chain 818 -> 694 us per launch (same-box A/B, profiles/r05_train_step_notes.md).
            if (row < rows_left) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4 *>(dst_tile + (size_t)row * (D_HID * 2) + lane * 16));
        }
    }
}

template <bool INIT, int JT_>
__device__ __forceinline__ void add_bias(f32x16 (&acc)[IT][JT_], const float *bias_lane, int slot) {
    // bias_lane = bias + wave*BIAS_FLOATS_PER_WAVE + h*16 ; slot stride NW*BIAS_FLOATS_PER_WAVE
    const float *b = bias_lane + (size_t)slot * (NW * BIAS_FLOATS_PER_WAVE);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        f32x4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const f32x4 *>(b + it * 32 + i * 4);
#pragma unroll
        for (int jt = 0; jt < JT_; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (INIT) acc[it][jt][r] = q[r >> 2][r & 3];
                else acc[it][jt][r] += q[r >> 2][r & 3];
            }
    }
}


// ---------------------------------------------------------------- projection + bilinear setup
// Camera-space point and pinhole projection (models.py:165,206-212), SpatialEncoder.index
// scaling (encoder.py:96-99,161-163) and grid_sample(bilinear, border, align_corners=True)
// corner offsets / weights, in the reference's fp32 op order (no FMA contraction).
// xr = R x (rotated point); off[] are element offsets into the NHWC grid.
struct Proj {
    uint32_t off[4];  // nw, ne, sw, se
    float w[4];
};
#pragma clang fp contract(off)
__device__ __forceinline__ Proj project_point(const EvalParams &q, const float *pose, int obj, int view, float xr0,
                                              float xr1, float xr2, bool valid) {
    const float xc0 = xr0 + pose[3], xc1 = xr1 + pose[7], xc2 = xr2 + pose[11];
    const float *fo = q.focal + (q.n_focal > 1 ? obj * 2 : 0);
    const float *cc = q.c + (q.n_c > 1 ? obj * 2 : 0);
    float u = -xc0 / xc2; u = u * fo[0]; u = u + cc[0];
    float v = -xc1 / xc2; v = v * fo[1]; v = v + cc[1];
    const float Wl = (float)q.Wl, Hl = (float)q.Hl;
    const float lsx = Wl / (Wl - 1.f) * 2.f, lsy = Hl / (Hl - 1.f) * 2.f;
    const float gx = u * (lsx / q.img_w) - 1.f, gy = v * (lsy / q.img_h) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * (Wl - 1.f), iy = ((gy + 1.f) / 2.f) * (Hl - 1.f);
    ix = fminf(Wl - 1.f, fmaxf(ix, 0.f));
    iy = fminf(Hl - 1.f, fmaxf(iy, 0.f));
    if (!(ix == ix) || !valid) ix = 0.f;  // NaN (point on the camera plane): keep reads in bounds
    if (!(iy == iy) || !valid) iy = 0.f;
    const float ix0 = floorf(ix), iy0 = floorf(iy);
    const float ix1 = ix0 + 1.f, iy1 = iy0 + 1.f;
    float w_nw = (ix1 - ix) * (iy1 - iy), w_ne = (ix - ix0) * (iy1 - iy);
    float w_sw = (ix1 - ix) * (iy - iy0), w_se = (ix - ix0) * (iy - iy0);
    const int x0 = (int)ix0, y0 = (int)iy0;
    const int x1 = min(x0 + 1, q.Wl - 1), y1 = min(y0 + 1, q.Hl - 1);  // out-of-range corner has weight 0
    if (x0 + 1 > q.Wl - 1) { w_ne = 0.f; w_se = 0.f; }
    if (y0 + 1 > q.Hl - 1) { w_sw = 0.f; w_se = 0.f; }
    if (!valid) { w_nw = w_ne = w_sw = w_se = 0.f; }
    const uint32_t rowbase = (uint32_t)(obj * q.NS + view) * (uint32_t)(q.Hl * q.Wl);
    Proj pr;
    pr.off[0] = (rowbase + y0 * q.Wl + x0) * C_LAT; pr.off[1] = (rowbase + y0 * q.Wl + x1) * C_LAT;
    pr.off[2] = (rowbase + y1 * q.Wl + x0) * C_LAT; pr.off[3] = (rowbase + y1 * q.Wl + x1) * C_LAT;
    pr.w[0] = w_nw; pr.w[1] = w_ne; pr.w[2] = w_sw; pr.w[3] = w_se;
    return pr;
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------- feature phase (geometry)
// Thread (p = tid&63, sub = tid>>6).  Follows the reference op order without FMA contraction
// so that fp32 intermediates round like the PyTorch eager path.
#pragma clang fp contract(off)
template <typename P, bool RAYS, typename TL>
__device__ __forceinline__ void geometry_item(const EvalParams &q, char *smem, int tile, int view, int p, int sub) {
    typedef typename P::T T;
    constexpr int MT = TL::MT, LDS_IN = TL::LDS_IN, LDS_META = TL::LDS_META;
    const int g = tile * MT + p;  // P < 2^31 (checked on the host)
    const bool valid = g < (int)q.P;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    int obj = 0;
    float X = 0, Y = 0, Z = 0;
    if (valid) {
        if (RAYS) {
            const int r = g / q.K;
            if (q.rays) {
                const float *ray = q.rays + (size_t)r * 8;
                ox = ray[0]; oy = ray[1]; oz = ray[2]; dx = ray[3]; dy = ray[4]; dz = ray[5];
            } else {  // util.gen_rays evaluated in place (same operation order as gen_rays_kernel: same bits)
                const Ray8 ry = load_ray(q.cam, r);
                ox = ry.ox; oy = ry.oy; oz = ry.oz; dx = ry.dx; dy = ry.dy; dz = ry.dz;
            }
            const float zz = q.z[g];
            X = ox + zz * dx; Y = oy + zz * dy; Z = oz + zz * dz;  // nerf.py:185
            obj = r / q.per_obj;
        } else {
            X = q.xyz[g * 3 + 0]; Y = q.xyz[g * 3 + 1]; Z = q.xyz[g * 3 + 2];
            dx = q.viewdirs[g * 3 + 0]; dy = q.viewdirs[g * 3 + 1]; dz = q.viewdirs[g * 3 + 2];
            obj = g / q.per_obj;
        }
    }
    const float *pose = q.poses + (size_t)(obj * q.NS + view) * 12;  // row = obj*NS + view
    // xyz_rot = R x (models.py:162-164)
    const float xr0 = pose[0] * X + pose[1] * Y + pose[2] * Z;
    const float xr1 = pose[4] * X + pose[5] * Y + pose[6] * Z;
    const float xr2 = pose[8] * X + pose[9] * Y + pose[10] * Z;
    T *in_row = reinterpret_cast<T *>(smem + LDS_IN + p * ROW_IN);
    // split-operand form (pnr_split.hip): value = hi + lo, two 16-bit images IN_LO_DELTA bytes apart
    auto put = [&](int i, float v) {
        const T hi = (T)v;
        in_row[i] = hi;
        if constexpr (TL::IN_LO_DELTA != 0) reinterpret_cast<T *>(reinterpret_cast<char *>(in_row) + TL::IN_LO_DELTA)[i] = (T)(v - (float)hi);
    };
    // two neighbouring elements (i even) as ONE 32-bit LDS store per image
    auto put2 = [&](int i, float v0, float v1) {
        struct alignas(4) Pair { T a, b; };
        const Pair hi = {(T)v0, (T)v1};
        *reinterpret_cast<Pair *>(in_row + i) = hi;
        if constexpr (TL::IN_LO_DELTA != 0) {
            const Pair lo = {(T)(v0 - (float)hi.a), (T)(v1 - (float)hi.b)};
            *reinterpret_cast<Pair *>(reinterpret_cast<char *>(in_row + i) + TL::IN_LO_DELTA) = lo;
        }
    };
    if (sub == 0) {
        // identity part of the code, rotated view direction (models.py:188-196), zero pad
        const float dv0 = pose[0] * dx + pose[1] * dy + pose[2] * dz;
        const float dv1 = pose[4] * dx + pose[5] * dy + pose[6] * dz;
        const float dv2 = pose[8] * dx + pose[9] * dy + pose[10] * dz;
        put2(0, valid ? xr0 : 0.f, valid ? xr1 : 0.f); put(2, valid ? xr2 : 0.f);
        put(39, dv0); put2(40, dv1, dv2);
        // camera-space point, pinhole projection, bilinear corner setup
        const Proj pr = project_point(q, pose, obj, view, xr0, xr1, xr2, valid);
        uint32_t *mo = reinterpret_cast<uint32_t *>(smem + LDS_META + p * 32);
        float *mw = reinterpret_cast<float *>(smem + LDS_META + p * 32 + 16);
        mo[0] = pr.off[0]; mo[1] = pr.off[1]; mo[2] = pr.off[2]; mo[3] = pr.off[3];
        mw[0] = pr.w[0]; mw[1] = pr.w[1]; mw[2] = pr.w[2]; mw[3] = pr.w[3];
    } else if (sub <= 6) {
        // frequency k = sub-1: sin(f x), sin(f x + pi/2), f = 1.5 * 2^k (code.py:15,37-41)
        const float f = 1.5f * (float)(1 << (sub - 1));
        const float HALF_PI = 1.57079637050628662109375f;  // fp32(pi/2), code.py:26
        const float a0 = xr0 * f, a1 = xr1 * f, a2 = xr2 * f;
        const int o = 3 + 6 * (sub - 1);
        // 16-bit operand kernels: the code is rounded to 11 (f16) / 8 (bf16) significand bits on its way into LDS, so the
        // hardware sine (v_sin_f32 on the argument in revolutions, |error| < 1e-5 for |a| < 150) is exact enough by two
        // orders of magnitude and ~10x cheaper than libm's range-reduced sinf; the split-operand (fp32-class) kernel
        // keeps the precise one.
        auto sn = [](float a) { return TL::IN_LO_DELTA != 0 ? sinf(a) : __sinf(a); };
        const float s0 = valid ? sn(a0) : 0.f, s1 = valid ? sn(a1) : 0.f, s2 = valid ? sn(a2) : 0.f;
        // the phase-shifted terms: ATen's addcmul(phases, x, freqs) (code.py:39) is ONE fused multiply-add per element -- x f
        // + pi/2 rounded once -- so the argument is formed with fmaf here as well (separately rounded it differs by up to an ulp
        // of the argument, 1.5e-5 at f = 48: found by tests/test_hip_features.py against the reference's own output)
        const float c0 = valid ? sn(__builtin_fmaf(xr0, f, HALF_PI)) : 0.f, c1 = valid ? sn(__builtin_fmaf(xr1, f, HALF_PI)) : 0.f;
        const float c2 = valid ? sn(__builtin_fmaf(xr2, f, HALF_PI)) : 0.f;
        put(o, s0); put2(o + 1, s1, s2); put2(o + 3, c0, c1); put(o + 5, c2);  // o is odd: the even-indexed pairs are 4-byte aligned
    } else {
        // zero the K padding 42..63 (+ the 8-element row pad read by the last B prefetch)
        for (int i = D_IN; i < D_IN_PAD + 8; i += 2) put2(i, 0.f, 0.f);
    }
}
#pragma clang fp contract(fast)

// work items (point, sub): sub 0 = identity code + view direction + projection, 1..6 = one frequency band
// (6 precise sinf), 7 = zero padding.  64-point tile: one item per thread; 96 points: 768 items over 512 threads.
template <typename P, bool RAYS, typename TL>
__device__ __forceinline__ void geometry(const EvalParams &q, char *smem, int tile, int view, int tid) {
    if constexpr (TL::MT == 64) {
        geometry_item<P, RAYS, TL>(q, smem, tile, view, tid & 63, tid >> 6);
    } else {
        // the six frequency bands first; the padding and projection items share the second, half-empty pass
#pragma unroll 1
        for (int wi = tid; wi < TL::MT * 8; wi += NTHREADS)
            geometry_item<P, RAYS, TL>(q, smem, tile, view, wi % TL::MT, (wi / TL::MT + 1) & 7);
    }
}


}  // namespace pnr
