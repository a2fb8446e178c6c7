// pnr_mlp.hip -- the fused per-point pixelNeRF network for gfx950 (MI355X).
//
// One persistent 512-thread workgroup per CU walks 64-point tiles.  Per tile and source view:
//   feature phase : world->camera transform, positional code, pinhole projection
//                   (models.py:161-212, code.py:30-42) -> LDS_IN / LDS_META;
//                   bilinear latent lookup from the NHWC grid (encoder.py:80-109) -> LDS_Z
//   network phase : ResnetFC (resnetfc.py:132-184) as a chain of MFMA GEMMs.  The residual
//                   stream x lives in the fp32 accumulators of the 8 waves (wave w owns hidden
//                   features 64w..64w+63 for all 64 points); weights stream from L2 straight
//                   into VGPRs through a 4-deep prefetch ring; the only activation traffic is
//                   relu(x)/relu(net) as 16-bit operands through one 64 KiB LDS buffer.
//   output        : lin_out as a K-split MFMA on the wave's own accumulators, 8-way reduce in
//                   LDS, sigmoid/relu (models.py:260-265), 16 B per point to HBM.
// Multi-view: views are processed sequentially through blocks 0-2, summed in registers and
// averaged before block 3 (util.combine_interleaved, util.py:461-471).
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "pnr_common.h"
#include "pnr_device.h"
#include "pnr_internal.h"
#include "pnr_layout.h"

// table-lookup batch sizes of the 96-point tile (points in flight per wave; 12 points per wave and table)
#ifndef PNR_GB0
#define PNR_GB0 6   // tile start: no accumulator is live
#endif
#ifndef PNR_GB12
#define PNR_GB12 4  // inside blocks 0-1: the residual stream is live
#endif

namespace pnr {

// bilinear lookup: wave handles points wave*8..+7; lane handles channels 8*lane..+7
template <typename P, bool TRAIN, int GB, typename TL>
__device__ __forceinline__ void gather(const EvalParams &q, char *smem, int wv, int lane, int tile, int view) {
    constexpr int MT = TL::MT, LDS_META = TL::LDS_META, LDS_Z = TL::LDS_Z;
    const float *lat = q.latent + lane * 8;
    // GB = points per batch (8 x 16-byte loads in flight per point); 4 when registers allow
    static_assert((MT / NW) % GB == 0, "gather batch");
#pragma unroll 1
    for (int i = 0; i < MT / NW; i += GB) {
        f32x4 v[GB][4][2];
        f32x4 w[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (MT / NW) + i + u;
            const u32x4 off = *reinterpret_cast<const u32x4 *>(smem + LDS_META + p * 32);
            w[u] = *reinterpret_cast<const f32x4 *>(smem + LDS_META + p * 32 + 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float *src = lat + off[c];
                v[u][c][0] = *reinterpret_cast<const f32x4 *>(src);
                v[u][c][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (MT / NW) + i + u;
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int hh = e >> 2, ee = e & 3;
                float a = v[u][0][hh][ee] * w[u][0];
                a += v[u][1][hh][ee] * w[u][1];
                a += v[u][2][hh][ee] * w[u][2];
                a += v[u][3][hh][ee] * w[u][3];
                r[e] = a;
            }
            const typename P::T8 zz = pack8<P, false>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
            *reinterpret_cast<typename P::T8 *>(smem + LDS_Z + p * ROW_ACT + lane * 16) = zz;
            if (TRAIN) {
                const long long g = (long long)tile * MT + p;
                if (g < q.P)
                    *reinterpret_cast<typename P::T8 *>(q.d_z + (((long long)view * q.P + g) * C_LAT + lane * 8) * 2) = zz;
            }
        }
    }
}


// Folded form (inference): the same bilinear combination over a per-texel TABLE T_b = W_z[b] . grid + b_z[b]
// (16-bit, hidden features in storage order, pnr_fold_latent) gives lin_z[b](z) directly; the rows go to the
// LDS_Z image and every lane adds its own slots to the residual stream (add_from_z).
template <typename P, int GB, typename TL>
__device__ __forceinline__ void gather_table(const EvalParams &q, char *smem, int wv, int lane, int b) {
    constexpr int MT = TL::MT, LDS_META = TL::LDS_META, LDS_Z = TL::LDS_Z;
    const typename P::T *tab = reinterpret_cast<const typename P::T *>(q.tables) + (size_t)b * q.table_stride + lane * 8;
    static_assert((MT / NW) % GB == 0, "gather batch");
#pragma unroll 1
    for (int i = 0; i < MT / NW; i += GB) {
        typename P::T8 v[GB][4];
        f32x4 w[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (MT / NW) + i + u;
            const u32x4 off = *reinterpret_cast<const u32x4 *>(smem + LDS_META + p * 32);
            w[u] = *reinterpret_cast<const f32x4 *>(smem + LDS_META + p * 32 + 16);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = *reinterpret_cast<const typename P::T8 *>(tab + off[c]);
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            const int p = wv * (MT / NW) + i + u;
            float r[8];
            if constexpr (P::kIsF16) {
                // v_fma_mix_f32: fp32 FMA that converts its f16 operand on the fly (op_sel picks the half of the packed
                // register) -- the same fused multiply-adds in the same order as the generic form below, without the 32
                // separate v_cvt_f32_f16 per point the compiler otherwise emits (the blend is VALU-bound)
                const u32x4 c0 = __builtin_bit_cast(u32x4, v[u][0]), c1 = __builtin_bit_cast(u32x4, v[u][1]);
                const u32x4 c2 = __builtin_bit_cast(u32x4, v[u][2]), c3 = __builtin_bit_cast(u32x4, v[u][3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float lo, hi;
                    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(c0[k]), "v"(w[u][0]));
                    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(c0[k]), "v"(w[u][0]));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(c1[k]), "v"(w[u][1]), "v"(lo));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(c1[k]), "v"(w[u][1]), "v"(hi));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(c2[k]), "v"(w[u][2]), "v"(lo));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(c2[k]), "v"(w[u][2]), "v"(hi));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(c3[k]), "v"(w[u][3]), "v"(lo));
                    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(c3[k]), "v"(w[u][3]), "v"(hi));
                    r[2 * k] = lo; r[2 * k + 1] = hi;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float a = (float)v[u][0][e] * w[u][0];
                    a += (float)v[u][1][e] * w[u][1];
                    a += (float)v[u][2][e] * w[u][2];
                    a += (float)v[u][3][e] * w[u][3];
                    r[e] = a;
                }
            }
            *reinterpret_cast<typename P::T8 *>(smem + LDS_Z + p * ROW_ACT + lane * 16) =
                pack8<P, false>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        }
    }
}

// x += this lane's slots of the 16-bit rows in the LDS_Z image (storage order = the accumulator map)
template <typename P, int JT_>
__device__ __forceinline__ void add_from_z(f32x16 (&x)[IT][JT_], const char *smem, uint32_t z_slot) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT_; ++jt) {
            const uint32_t ad = z_slot + jt * 32 * ROW_ACT + it * 64;
            const typename P::T8 lo = *reinterpret_cast<const typename P::T8 *>(smem + ad);
            const typename P::T8 hi = *reinterpret_cast<const typename P::T8 *>(smem + ad + 16);
            if constexpr (P::kIsF16) {  // x += f16 -> one v_fma_mix_f32 (x = h * 1.0 + x) instead of convert + add
                const u32x4 ul = __builtin_bit_cast(u32x4, lo), uh = __builtin_bit_cast(u32x4, hi);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x[it][jt][2 * k]) : "v"(ul[k]));
                    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[it][jt][2 * k + 1]) : "v"(ul[k]));
                    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(x[it][jt][8 + 2 * k]) : "v"(uh[k]));
                    asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[it][jt][8 + 2 * k + 1]) : "v"(uh[k]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    x[it][jt][r] += (float)lo[r];
                    x[it][jt][8 + r] += (float)hi[r];
                }
            }
        }
}

// one residual block (+ the lin_z of the next block when with_z):
//   net = fc_0(relu(x)); x += fc_1(relu(net)) [+ lin_z[b+1](z)]       resnetfc.py:55-62,174-182
template <typename P, bool TIMING, bool TRAIN, bool FOLD, typename TL, bool MV_PARK = false>
__device__ __forceinline__ void res_block(f32x16 (&x)[IT][TL::JT], char *smem, int b, bool with_z, Ring<P> &R,
                                          int NS, const float *bias_lane, uint32_t a_rd0, uint32_t a_rd1,
                                          uint32_t z_rd0, uint32_t z_rd1, uint32_t a_wr, int tid,
                                          unsigned long long *tim, unsigned long long &tlast,
                                          const EvalParams &q, size_t dump_off, const bool *valid, int wv, int lane,
                                          uint32_t mask_off = 0, size_t mask_layer = 0, long long rows_left = 0, f32x4 *park = nullptr, bool park_first = true, bool park_last = true, float park_inv = 1.f, bool park_max = false) {
    typedef Advance<0, FOLD ? RS_VIEW_END_F : RS_VIEW_END, FOLD ? RS_TOTAL_F : RS_TOTAL> ADV;
    constexpr int JT = TL::JT;
    // dump_off: byte offset of this lane's 32-byte slot in a (rows,512) 16-bit dump array
    // mask_off / mask_layer: this thread's word within a layer of q.d_mask / words per layer (layer 2b: x, 2b+1: net)
    // park (multi-view, last per-view block only): this thread's slots of the parked running view sum (see eval_kernel)
    __syncthreads();  // every wave is done reading LDS_A (previous fc_1)
    PNR_T(PH_BAR1);
    // training dumps: relu bit masks from the registers, the rows themselves copied out of the image behind the barrier
    [[maybe_unused]] const size_t dump_tile = dump_off - (size_t)(((lane & 31) * D_HID + (wv * IT) * 32 + (lane >> 5) * 16) * 2);
    write_act<P, true, TRAIN>(x, smem, a_wr, nullptr, valid, TRAIN ? q.d_mask + (size_t)(2 * b) * mask_layer + mask_off : nullptr);
    PNR_T(PH_WRITE_X);
    __syncthreads();
    PNR_T(PH_BAR2);
    if constexpr (TRAIN) dump_image<TL::MT>(smem, TL::LDS_A, q.d_a[b] + dump_tile, rows_left, wv, lane);
    {
        f32x16 net[IT][JT];
        add_bias<true>(net, bias_lane, 1 + 2 * b);
        gemm<P, ADV>(net, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
        PNR_T(PH_GEMM_FC0);
        __syncthreads();  // every wave is done reading relu(x)
        PNR_T(PH_BAR3);
        write_act<P, true, TRAIN>(net, smem, a_wr, nullptr, valid, TRAIN ? q.d_mask + (size_t)(2 * b + 1) * mask_layer + mask_off : nullptr);
        PNR_T(PH_WRITE_NET);
    }
    __syncthreads();
    PNR_T(PH_BAR4);
    if constexpr (TRAIN) dump_image<TL::MT>(smem, TL::LDS_A, q.d_n[b] + dump_tile, rows_left, wv, lane);
    add_bias<false>(x, bias_lane, 2 + 2 * b);
    // multi-view pooling: the running sum of the previous views comes back from its scratch UNDER this GEMM (`net` is dead,
    // its registers hold the loads in flight), so the view boundary costs no exposed memory round trip
    f32x4 parked[(MV_PARK ? IT * JT * 4 : 1)];
    if constexpr (MV_PARK) {
        if (park && !park_first) {
#pragma unroll
            for (int i = 0; i < IT * JT * 4; ++i) parked[i] = park[i * NTHREADS];  // [slot][thread]: 1 KiB per wave-instruction
        }
    }
    gemm<P, ADV>(x, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
    if constexpr (MV_PARK) {
        if (park) {
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = (it * JT + jt) * 4 + k;
                        f32x4 v = {x[it][jt][4 * k], x[it][jt][4 * k + 1], x[it][jt][4 * k + 2], x[it][jt][4 * k + 3]};
                        if (!park_first) {
                            if (park_max) { v[0] = fmaxf(parked[i][0], v[0]); v[1] = fmaxf(parked[i][1], v[1]); v[2] = fmaxf(parked[i][2], v[2]); v[3] = fmaxf(parked[i][3], v[3]); }
                            else v += parked[i];
                        }
                        if (!park_last) park[i * NTHREADS] = v;
                        else if (!park_max) v *= park_inv;
                        x[it][jt][4 * k] = v[0]; x[it][jt][4 * k + 1] = v[1]; x[it][jt][4 * k + 2] = v[2]; x[it][jt][4 * k + 3] = v[3];
                    }
        }
    }
    if (with_z) {
        if constexpr (FOLD) {  // lin_z[b+1](z) = bilinear lookup in table b+1 (LDS_Z is free: its last readers ran before fc_0)
            PNR_T(PH_GEMM_FC1_Z);
            if constexpr (TL::SINGLE_IMAGE) __syncthreads();  // the rows land in the image fc_1 has just been reading
            gather_table<P, TL::SINGLE_IMAGE ? PNR_GB12 : 2, TL>(q, smem, wv, lane, b + 1);  // the residual stream is live here: smaller batches
            __syncthreads();
            add_from_z<P>(x, smem, a_wr - TL::LDS_A + TL::LDS_Z);
            PNR_T(PH_TABLE);
        } else {
            gemm<P, ADV>(x, smem, z_rd0, z_rd1, KS_BIG / 4, R, NS);
        }
    }
    PNR_T(PH_GEMM_FC1_Z);
}

template <int PREC, bool RAYS, bool MV, bool TIMING = false, bool TRAIN = false, bool FOLD = false, int MT_ = 64>
__global__ void __launch_bounds__(NTHREADS, NW / 4) eval_kernel(const EvalParams q) {
    typedef Prec<PREC> P;
    typedef Tile<MT_> TL;
    typedef Advance<0, FOLD ? RS_VIEW_END_F : RS_VIEW_END, FOLD ? RS_TOTAL_F : RS_TOTAL> ADV;
    constexpr int MT = TL::MT, JT = TL::JT;
    constexpr int LDS_Z = TL::LDS_Z, LDS_A = TL::LDS_A, LDS_IN = TL::LDS_IN, LDS_OUT = TL::LDS_OUT;
    static_assert(!(FOLD && TRAIN), "the training instantiation keeps the lin_z GEMMs (their operands are dumped)");
    static_assert(!TL::SINGLE_IMAGE || FOLD, "the one-image tile is the folded form (lin_z as table lookups)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int NS = MV ? q.NS : 1;

    // per-lane LDS addresses
    const uint32_t a_rd0 = LDS_A + pl * ROW_ACT + h * 16, a_rd1 = a_rd0 + 32 * ROW_ACT;
    const uint32_t z_rd0 = LDS_Z + pl * ROW_ACT + h * 16, z_rd1 = z_rd0 + 32 * ROW_ACT;
    const uint32_t in_rd0 = LDS_IN + pl * ROW_IN + h * 16, in_rd1 = in_rd0 + 32 * ROW_IN;
    const uint32_t a_wr = LDS_A + pl * ROW_ACT + (wv * IT) * 64 + h * 32;
    const float *bias_lane = q.bias + wv * BIAS_FLOATS_PER_WAVE + h * 16;
    if constexpr (TL::BIAS_IN_LDS) {
        static_assert(NBIAS == 11, "LDS_BIAS size");
        for (int i = tid; i < NBIAS * NW * BIAS_FLOATS_PER_WAVE / 4; i += NTHREADS)
            reinterpret_cast<f32x4 *>(smem + TL::LDS_BIAS)[i] = reinterpret_cast<const f32x4 *>(q.bias)[i];
        bias_lane = reinterpret_cast<const float *>(smem + TL::LDS_BIAS) + wv * BIAS_FLOATS_PER_WAVE + h * 16;
        // (published by the first tile's top-of-tile barrier)
    }

    f16_ovfl_mode<P>();
    Ring<P> R;
    R.wave_base = q.wstream + (size_t)wv * ((FOLD ? RS_TOTAL_F : RS_TOTAL) * IT * 1024) + lane * 16;
    R.pf_rs = 0;
    R.pf_view = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int it = 0; it < IT; ++it) R.r[j][it] = gload8<P>(R.wave_base + j * (IT * 1024) + it * 1024);
    R.pf_rs = 4;
    unsigned long long *tim = q.tim;
    unsigned long long tlast = TIMING ? __builtin_readcyclecounter() : 0ull;

    // XCD-aware tile order (pnr_device.h tile_range; same-box A/B against plain grid-stride: sn64 +0.7 %, srn_car +0.4 %, DTU +1.4 %)
    const TileRange order = tile_range(q.ntiles, q.n_xcd);
    for (int tile = order.begin; tile < order.end; tile += order.step) {
        f32x16 x[IT][JT];
        // training dumps: this lane's 32-byte slot in a (rows,512) array, row = [view*P +] point
        bool valid[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) valid[jt] = (long long)tile * MT + jt * 32 + pl < q.P;
        // training + multi-view: the running view sum is parked (simple read-modify-write at the view boundary, as in
        // pnr_split.hip) -- with the dump bookkeeping live as well, the in-register form spilled 76 registers
        [[maybe_unused]] constexpr bool PARK_SUM = MV && TRAIN;
        // pooling over the source views: util.combine_interleaved's mean, or (network flag word, pnr_layout.h) its "max"
        [[maybe_unused]] const bool cmax = MV && (reinterpret_cast<const int *>(q.bout)[BOUT_FLAGS_INDEX] & 1) != 0;
        // multi-view: running view sum in 64 live registers.  The instantiation then sits at the 256-register limit and spills
        // 20-40 registers to scratch OUTSIDE the GEMM loops; the spill-free alternatives (-DPNR_MV_PARK: sum parked in an
        // L2-resident scratch, simple or prefetched under the last fc_1) measured 2.7-4 % and 4-8 % SLOWER on the same box
        // (profiles/r02_mv_pooling_ab.txt), so the registers stay.
        [[maybe_unused]] f32x16 xsum[(MV && !PARK_SUM) ? IT : 1][(MV && !PARK_SUM) ? JT : 1];
        const size_t dump_pooled = (((size_t)tile * MT + pl) * D_HID + (wv * IT) * 32 + h * 16) * 2;
        // relu bit masks (training): [layer][view][tile][thread] words; pooled layers use view slot 0
        const size_t mask_layer = (size_t)NS * (size_t)q.ntiles * NTHREADS;
        const long long rows_left = q.P - (long long)tile * MT;
        const uint32_t mask_pooled = (uint32_t)tile * NTHREADS + tid;  // 32 bits: NS * tiles * 512 < 2^32 (host check)
#pragma unroll 1
        for (int view = 0; view < NS; ++view) {
            const size_t dump_view = dump_pooled + (size_t)view * (size_t)q.P * (D_HID * 2);
            const uint32_t mask_view = mask_pooled + (uint32_t)view * (uint32_t)q.ntiles * NTHREADS;
            __syncthreads();  // previous users of LDS_IN / LDS_META / LDS_Z are done
            PNR_T(PH_SYNC_TOP);
            geometry<P, RAYS, TL>(q, smem, tile, view, tid);
            __syncthreads();
            PNR_T(PH_GEOMETRY);
            if (TRAIN) {  // dump the lin_in operand rows (64 x 128 B) of this tile
                const int row = tid >> 3, chunk = tid & 7;
                const long long g = (long long)tile * MT + row;
                if (tid < MT * 8 && g < q.P)
                    *reinterpret_cast<u32x4 *>(q.d_in + (((long long)view * q.P + g) * D_IN_PAD + chunk * 8) * 2) =
                        *reinterpret_cast<const u32x4 *>(smem + LDS_IN + row * ROW_IN + chunk * 16);
            }
            if constexpr (FOLD) gather_table<P, TL::SINGLE_IMAGE ? PNR_GB0 : (MV ? 2 : 4), TL>(q, smem, wv, lane, 0);
            else gather<P, TRAIN, MV ? 2 : 4, TL>(q, smem, wv, lane, tile, view);  // multi-view also holds the view sum
            __syncthreads();
            PNR_T(PH_GATHER);
            add_bias<true>(x, bias_lane, B_IN_Z0);
            gemm<P, ADV>(x, smem, in_rd0, in_rd1, KS_IN / 4, R, NS);      // lin_in     resnetfc.py:147
            if constexpr (FOLD) add_from_z<P>(x, smem, a_wr - LDS_A + LDS_Z);  // lin_z[0] via table 0
            else gemm<P, ADV>(x, smem, z_rd0, z_rd1, KS_BIG / 4, R, NS);  // lin_z[0]   resnetfc.py:175-180
            PNR_T(PH_GEMM_IN_Z0);
#pragma unroll 1
            for (int b = 0; b < COMBINE_LAYER; ++b)
                res_block<P, TIMING, TRAIN, FOLD, TL>(x, smem, b, b + 1 < COMBINE_LAYER, R, NS, bias_lane, a_rd0, a_rd1, z_rd0, z_rd1,
                                                  a_wr, tid, tim, tlast, q, dump_view, valid, wv, lane, mask_view, mask_layer, rows_left);
            if constexpr (MV && PARK_SUM) {  // mean over source views, the sum parked in the per-workgroup scratch ([slot][thread])
                f32x4 *ws = reinterpret_cast<f32x4 *>(q.mv_ws) + (size_t)blockIdx.x * (IT * JT * 4 * NTHREADS) + tid;
                const float inv = 1.f / (float)NS;
                const bool first = view == 0, last = view + 1 == NS;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int i = (it * JT + jt) * 4 + k;
                            f32x4 v = {x[it][jt][4 * k], x[it][jt][4 * k + 1], x[it][jt][4 * k + 2], x[it][jt][4 * k + 3]};
                            if (!first) {  // same association as the in-register form: (v0 + v1) + ...
                                const f32x4 prev = ws[i * NTHREADS];
                                if (cmax) { v[0] = fmaxf(prev[0], v[0]); v[1] = fmaxf(prev[1], v[1]); v[2] = fmaxf(prev[2], v[2]); v[3] = fmaxf(prev[3], v[3]); }
                                else v = prev + v;
                            }
                            if (!last) ws[i * NTHREADS] = v;
                            else if (!cmax) v *= inv;
                            x[it][jt][4 * k] = v[0]; x[it][jt][4 * k + 1] = v[1]; x[it][jt][4 * k + 2] = v[2]; x[it][jt][4 * k + 3] = v[3];
                        }
                    }
            } else if constexpr (MV) {  // mean over source views (util.combine_interleaved, util.py:461-466)
                const float inv = cmax ? 1.f : 1.f / (float)NS;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if (view == 0) xsum[it][jt] = x[it][jt];
                        else if (cmax) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) xsum[it][jt][r] = fmaxf(xsum[it][jt][r], x[it][jt][r]);
                        } else xsum[it][jt] += x[it][jt];
                        if (view + 1 == NS) x[it][jt] = xsum[it][jt] * inv;
                    }
            }
        }
#pragma unroll 1
        for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b)
            res_block<P, TIMING, TRAIN, FOLD, TL>(x, smem, b, false, R, NS, bias_lane, a_rd0, a_rd1, z_rd0, z_rd1, a_wr, tid, tim, tlast,
                                              q, dump_pooled, valid, wv, lane, mask_pooled, mask_layer, rows_left);

        if (q.dbg) {
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const long long g = (long long)tile * MT + jt * 32 + pl;
                    if (g < q.P)
                        for (int r = 0; r < 16; ++r) q.dbg[g * D_HID + feat_of(wv * IT + it, h, r)] = x[it][jt][r];
                }
        }

        // lin_out(relu(x)) (resnetfc.py:183): each wave contracts its own 64 features
        {
            f32x16 o[JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[jt][r] = 0.f;
            const char *pf = R.wave_base + (size_t)R.pf_rs * (IT * 1024);
            [[maybe_unused]] unsigned long long x5_bits = 0ull;
#pragma unroll
            for (int qk = 0; qk < 2 * IT; ++qk) {
                const int xit = qk >> 1, rr = qk & 1;
                const typename P::T8 a = R.r[qk / IT][qk % IT];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const f32x16 &v = x[xit][jt];
                    const typename P::T8 bq =
                        rr == 0 ? pack8<P, true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
                                : pack8<P, true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
                    if (TRAIN) {
                        if (valid[jt])
                            *reinterpret_cast<typename P::T8 *>(q.d_x5 + dump_pooled + (size_t)jt * 32 * (D_HID * 2) +
                                                                xit * 64 + rr * 16) = bq;
                        if (IT * JT * 16 <= 64) x5_bits |= (unsigned long long)nonzero_bits8(bq) << ((xit * JT + jt) * 16 + 8 * rr);
                    }
                    o[jt] = P::mfma(a, bq, o[jt]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int it = 0; it < IT; ++it) R.r[j][it] = gload8<P>(pf + j * (IT * 1024) + it * 1024);
            ADV::step4(R, NS);
            if (TRAIN) {
                (q.d_mask + (size_t)10 * mask_layer)[mask_pooled] = x5_bits;
            }
            if (h == 0) {
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    f32x4 t = {o[jt][0], o[jt][1], o[jt][2], o[jt][3]};
                    *reinterpret_cast<f32x4 *>(smem + LDS_OUT + (wv * MT + jt * 32 + pl) * 16) = t;
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
            for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x4 *>(smem + LDS_OUT + (w * MT + tid) * 16);
            // models.py:260-265: rgb = sigmoid(out[:3]), sigma = relu(out[3])
            f32x4 res = {1.f / (1.f + expf(-s[0])), 1.f / (1.f + expf(-s[1])), 1.f / (1.f + expf(-s[2])),
                         fmaxf(s[3], 0.f)};
            if (g < q.P) *reinterpret_cast<f32x4 *>(q.out + g * 4) = res;
        }
        PNR_T(PH_FINAL);
    }
}

// ---------------------------------------------------------------- host side
static bool g_profile = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events;

static int num_cus();

ProfileScope::ProfileScope(hipStream_t s) : st(s) {
    if (!g_profile) return;
    if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
    hipEventRecord(e0, st);
}
ProfileScope::~ProfileScope() {
    if (!e0) return;
    hipEvent_t e1 = nullptr;
    if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return; }
    hipEventRecord(e1, st);
    g_events.emplace_back(e0, e1);
}

// tile size by instantiation: the folded single-view inference form runs 96-point tiles (one LDS image), everything
// else (unfolded, multi-view, training dumps) the 64-point two-image tile.  Small launches are a whole number of rounds
// of one tile per CU: a 96-point tile costs 1.37 x a 64-point one (per point it is 9-10 % cheaper), so the 64-point form
// is taken when it needs fewer rounds-times-cost (32 768 points on 256 CUs: 2 rounds either way -> 64; 49 152: 2 rounds
// of 96 against 3 of 64 -> 96).  Both forms give the same bits per point.
static inline bool use_tile96(const EvalParams &q, bool mv) {
    if (!(q.tables && !mv && !q.d_z)) return false;  // multi-view: the 64-point tile
    const long long ncu = num_cus();
    const long long r96 = ((q.P + 95) / 96 + ncu - 1) / ncu, r64 = ((q.P + 63) / 64 + ncu - 1) / ncu;
    return r96 * 137 <= r64 * 100;
}

// per (device, stream) scratch of the multi-view instantiations (the split-operand kernel, and -DPNR_MV_PARK builds of this one): the parked view sum, one tile
// of fp32 accumulators per workgroup (256 x 192 KiB = 48 MiB), allocated at the first multi-view launch on a stream.
float *mv_scratch(hipStream_t st, size_t bytes) {
    struct Slot { int dev; hipStream_t st; float *p; size_t bytes; };
    static std::vector<Slot> slots;
    static std::mutex mu;  // single-process multi-GPU training runs one autograd thread per device: the table is shared
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    // growing the table allocates: illegal inside a HIP-graph capture.  A capture runs on a stream of its own (torch.cuda.graph), which
    // no warm-up call has ever seen: the launch being captured then BORROWS the largest scratch another stream of this device
    // already owns (the warm-up's).  That is ordered correctly as long as the replayed graph and multi-view launches on that
    // other stream do not run at the same time -- the one restriction of capturing a multi-view call.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    for (auto &sl : slots)
        if (sl.dev == dev && sl.st == st) {
            if (sl.bytes >= bytes) return sl.p;
            if (capturing) break;
            (void)hipFree(sl.p);
            sl.p = nullptr; sl.bytes = 0;
            if (hipMalloc(&sl.p, bytes) != hipSuccess) return nullptr;
            sl.bytes = bytes;
            return sl.p;
        }
    if (capturing) {
        float *best = nullptr;
        size_t best_bytes = 0;
        for (auto &sl : slots)
            if (sl.dev == dev && sl.bytes >= bytes && sl.bytes > best_bytes) { best = sl.p; best_bytes = sl.bytes; }
        return best;  // nullptr: no multi-view launch has run on this device outside of a capture yet
    }
    Slot sl = {dev, st, nullptr, bytes};
    if (hipMalloc(&sl.p, bytes) != hipSuccess) return nullptr;
    slots.push_back(sl);
    return sl.p;
}

template <int PREC, bool RAYS>
static int launch(EvalParams &q, bool mv, hipStream_t st) {
    hipError_t e;
    auto k = mv ? eval_kernel<PREC, RAYS, true> : eval_kernel<PREC, RAYS, false>;
    int mt = 64, lds = Tile<64>::LDS_TOTAL;
    if (RAYS && q.d_z) k = mv ? eval_kernel<PREC, true, true, false, true> : eval_kernel<PREC, true, false, false, true>;
    else if (use_tile96(q, mv)) {
        k = eval_kernel<PREC, RAYS, false, false, false, true, 96>;
        mt = 96; lds = Tile<96>::LDS_TOTAL;
    }
    else if (q.tables) k = mv ? eval_kernel<PREC, RAYS, true, false, false, true> : eval_kernel<PREC, RAYS, false, false, false, true>;
    const long long nt = (q.P + mt - 1) / mt;
    q.ntiles = (int)nt;
    const int grid = (int)(nt < num_cus() ? nt : num_cus());
    if (mv && RAYS && q.d_z)  // the training instantiation parks its view sum
    {
        q.mv_ws = mv_scratch(st, (size_t)num_cus() * 96 * D_HID * sizeof(float));
        if (!q.mv_ws) return pnr_fail(PNR_E_HIP, "pnr_eval: cannot allocate the multi-view pooling scratch (48 MiB)");
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(eval_kernel)");
    {
        q.n_xcd = device_xcd_count();
        ProfileScope prof(st);
        hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, st, q);
    }
    return pnr_check_launch("eval_kernel");
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static int eval_common(const PnrScene *s, const void *packed, int precision, EvalParams &q, bool rays, hipStream_t st,
                       const void *tables = nullptr) {
    if (!s || !packed || !q.out) return pnr_fail(PNR_E_INVALID, "pnr_eval: null argument");
    q.tables = (const char *)tables;  // non-null: `packed` is a folded stream (pnr_pack_mlp_folded)
    if (tables && q.d_z) return pnr_fail(PNR_E_INVALID, "pnr_eval: the training instantiation is not folded");
    q.table_stride = (long long)s->SB * s->NS * s->Hl * s->Wl * C_LAT;
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_eval: bad scene shape");
    if (!(s->n_focal == 1 || s->n_focal == s->SB) || !(s->n_c == 1 || s->n_c == s->SB))
        return pnr_fail(PNR_E_INVALID, "pnr_eval: focal / c must have 1 or SB rows");
    if (q.P == 0) return PNR_OK;
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.wstream = (const char *)packed;
    q.bias = (const float *)((const char *)packed + BIAS_OFFSET_BYTES);
    q.bout = (const float *)((const char *)packed + BOUT_OFFSET_BYTES);
    if (q.P > 0x7fffff80LL) return pnr_fail(PNR_E_INVALID, "pnr_eval: too many points (P must stay below 2^31)");
    // texel offsets are 32-bit element indices into the grid / the tables (project_point)
    if ((long long)s->SB * s->NS * s->Hl * s->Wl * C_LAT > 0xffffffffLL)
        return pnr_fail(PNR_E_INVALID, "pnr_eval: feature grid too large (SB*NS*Hl*Wl*512 must stay below 2^32 elements)");
    const bool mv = s->NS > 1;
    if (precision == PNR_PREC_F16) return rays ? launch<PNR_PREC_F16, true>(q, mv, st) : launch<PNR_PREC_F16, false>(q, mv, st);
    if (precision == PNR_PREC_BF16) return rays ? launch<PNR_PREC_BF16, true>(q, mv, st) : launch<PNR_PREC_BF16, false>(q, mv, st);
    return pnr_fail(PNR_E_INVALID, "pnr_eval: unknown precision");
}

}  // namespace pnr

// test/diagnostic hook (not in the public header): per-phase s_memtime totals of wave 0 of
// workgroup 0 for one f16 single-view launch.  tim: NW*NPHASE device counters, zeroed by the caller.
extern "C" int pnr_debug_phase_timing(const PnrScene *s, const void *packed, const void *tables, const float *rays,
                                      const float *z, int R, int rays_per_obj, int K, unsigned long long *tim,
                                      void *stream) {
    using namespace pnr;
    if (!s || !packed || !rays || !z || !tim || s->NS != 1) return pnr_fail(PNR_E_INVALID, "pnr_debug_phase_timing: bad argument");
    EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.tim = tim;
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = 1; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.wstream = (const char *)packed;
    q.bias = (const float *)((const char *)packed + BIAS_OFFSET_BYTES);
    q.bout = (const float *)((const char *)packed + BOUT_OFFSET_BYTES);
    const int mt = tables ? 96 : 64;
    q.ntiles = (int)((q.P + mt - 1) / mt);
    q.tables = (const char *)tables;  // non-null: folded stream
    q.table_stride = (long long)s->SB * s->NS * s->Hl * s->Wl * C_LAT;
    static float *scratch_out = nullptr;
    static long long scratch_n = 0;
    if (scratch_n < q.P) {
        if (scratch_out) (void)hipFree(scratch_out);
        if (hipMalloc(&scratch_out, (size_t)q.P * 16) != hipSuccess) return pnr_fail(PNR_E_HIP, "hipMalloc");
        scratch_n = q.P;
    }
    q.out = scratch_out;
    auto k = tables ? eval_kernel<PNR_PREC_F16, true, false, true, false, true, 96> : eval_kernel<PNR_PREC_F16, true, false, true>;
    const int lds = tables ? Tile<96>::LDS_TOTAL : Tile<64>::LDS_TOTAL;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute");
    const int grid = q.ntiles < num_cus() ? q.ntiles : num_cus();
    q.n_xcd = device_xcd_count();
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), lds, (hipStream_t)stream, q);
    return pnr_check_launch("eval_kernel<timing>");
}

static float *g_dbg_ptr = nullptr;
// test hook (not part of the public header): dump the final residual stream of the next launches
extern "C" int pnr_debug_set_x_dump(float *ptr) { g_dbg_ptr = ptr; return PNR_OK; }

int pnr::eval_samples_src(const PnrScene *scene, const void *packed, const void *tables, int precision, const RaySrc &src,
                          const float *z, int R, int rays_per_obj, int K, float *rgbsigma, hipStream_t stream) {
    if (precision == PNR_PREC_F16X3) return eval_samples_split_src(scene, packed, tables, src, z, R, rays_per_obj, K, rgbsigma, stream);
    if (R < 0 || K <= 0 || rays_per_obj <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: bad sizes");
    if (R > 0 && ((!src.rays && !src.poses) || !z)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: null rays/z");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: R != SB * rays_per_obj");
    pnr::EvalParams q = {};
    q.rays = src.rays; q.cam = src; q.cam.rays = nullptr;
    q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma; q.dbg = g_dbg_ptr;
    return pnr::eval_common(scene, packed, precision, q, true, stream, tables);
}

static int eval_ray_samples_impl(const PnrScene *scene, const void *packed, const void *tables, int precision,
                                 const float *rays, const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                 void *stream) {
    pnr::RaySrc src = {};
    src.rays = rays;
    return pnr::eval_samples_src(scene, packed, tables, precision, src, z, R, rays_per_obj, K, rgbsigma, (hipStream_t)stream);
}

extern "C" int pnr_eval_ray_samples(const PnrScene *scene, const void *packed, int precision, const float *rays,
                                    const float *z, int R, int rays_per_obj, int K, float *rgbsigma, void *stream) {
    return eval_ray_samples_impl(scene, packed, nullptr, precision, rays, z, R, rays_per_obj, K, rgbsigma, stream);
}

extern "C" int pnr_eval_ray_samples_folded(const PnrScene *scene, const void *packed_folded, const void *tables,
                                           int precision, const float *rays, const float *z, int R, int rays_per_obj,
                                           int K, float *rgbsigma, void *stream) {
    if (!tables) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_folded: null tables");
    return eval_ray_samples_impl(scene, packed_folded, tables, precision, rays, z, R, rays_per_obj, K, rgbsigma, stream);
}

extern "C" int pnr_eval_ray_samples_train(const PnrScene *scene, const void *packed, int precision, const float *rays,
                                          const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                          const PnrTrainDumps *dumps, void *stream) {
    if (R <= 0 || K <= 0 || rays_per_obj <= 0 || !rays || !z || !dumps)
        return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_train: bad argument");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_train: R != SB * rays_per_obj");
    if (!dumps->d_in || !dumps->d_z || !dumps->d_x5) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_train: null dump buffer");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma;
    q.d_in = (char *)dumps->d_in; q.d_z = (char *)dumps->d_z; q.d_x5 = (char *)dumps->d_x5;
    q.d_mask = (unsigned long long *)dumps->d_mask;
    if (!q.d_mask) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_train: null relu-mask buffer (PnrTrainDumps.d_mask)");
    for (int b = 0; b < 5; ++b) {
        if (!dumps->d_a[b] || !dumps->d_n[b]) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_train: null dump buffer");
        q.d_a[b] = (char *)dumps->d_a[b]; q.d_n[b] = (char *)dumps->d_n[b];
    }
    return pnr::eval_common(scene, packed, precision, q, true, (hipStream_t)stream);
}

static int eval_points_impl(const PnrScene *scene, const void *packed, const void *tables, int precision, const float *xyz,
                            const float *viewdirs, int B, float *rgbsigma, void *stream) {
    if (B < 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_points: bad sizes");
    if (B > 0 && (!xyz || !viewdirs)) return pnr_fail(PNR_E_INVALID, "pnr_eval_points: null xyz/viewdirs");
    pnr::EvalParams q = {};
    q.xyz = xyz; q.viewdirs = viewdirs; q.K = 1; q.per_obj = B > 0 ? B : 1;
    q.P = scene ? (long long)scene->SB * B : 0; q.out = rgbsigma; q.dbg = g_dbg_ptr;
    return pnr::eval_common(scene, packed, precision, q, false, (hipStream_t)stream, tables);
}

extern "C" int pnr_eval_points(const PnrScene *scene, const void *packed, int precision, const float *xyz,
                               const float *viewdirs, int B, float *rgbsigma, void *stream) {
    return eval_points_impl(scene, packed, nullptr, precision, xyz, viewdirs, B, rgbsigma, stream);
}

extern "C" int pnr_eval_points_folded(const PnrScene *scene, const void *packed_folded, const void *tables, int precision,
                                      const float *xyz, const float *viewdirs, int B, float *rgbsigma, void *stream) {
    if (!tables) return pnr_fail(PNR_E_INVALID, "pnr_eval_points_folded: null tables");
    return eval_points_impl(scene, packed_folded, tables, precision, xyz, viewdirs, B, rgbsigma, stream);
}

extern "C" int pnr_profile_enable(int on) {
    pnr::g_profile = on != 0;
    for (auto &p : pnr::g_events) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    pnr::g_events.clear();
    return PNR_OK;
}

extern "C" int pnr_profile_read(double *mlp_kernel_ms, int *mlp_launches) {
    double tot = 0;
    for (auto &p : pnr::g_events) {
        float ms = 0.f;
        hipError_t e = hipEventElapsedTime(&ms, p.first, p.second);
        if (e != hipSuccess) return pnr_check_hip(e, "pnr_profile_read (stream not synchronised?)");
        tot += ms;
    }
    if (mlp_kernel_ms) *mlp_kernel_ms = tot;
    if (mlp_launches) *mlp_launches = (int)pnr::g_events.size();
    return PNR_OK;
}
