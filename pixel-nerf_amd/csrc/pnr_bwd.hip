// pnr_bwd.hip -- backward of the hot path for training (BASELINE config 5), gfx950.
//
//   bwd_kernel            fused data-gradient chain of one ResnetFC: same tile geometry as the
//                         forward kernel (persistent 512-thread workgroup, 64-point tiles, wave w
//                         owns features 64w..64w+63, gradient of the residual stream resident in
//                         fp32 accumulators), driven by a TRANSPOSED weight stream
//                         (lin_out^T, fc_1[b]^T, fc_0[b]^T, b = 4..0, then lin_z[2..0]^T and lin_in^T).
//                         relu masks = the forward's 1-bit-per-element words; every layer's output
//                         gradient dY leaves as whole 16-bit rows copied out of the LDS image
//                         (operands of the weight-gradient GEMMs dW = dY^T X).
//   dw_kernel             dW = dY^T X for all linears of a network in one launch (MFMA from the dumps
//                         through transposing LDS reads), dw_reduce_kernel sums the row slices.
//   dw_split_kernel       the same at split-operand (fp32-class) precision: (head | tail) f16 operand rows, three
//                         MFMAs per product -- the weight gradients of the fused fp32-class training path.
//   composite_bwd_kernel  backward of the alpha compositing (nerf.py:223-249), wavefront per ray.
//                         also emits dL/dz through the deltas and depth = sum w z.
//   latent_scatter_*      d(interpolated latent) -> d(feature grid): bilinear scatter-add (small grids: an fp64
//                         slab in LDS per (image, channel slice) fed per ray segment; large: global atomics).
//   position_bwd_kernel   dL/dz through the network inputs (positional code, projection, bilinear
//                         coordinates): the reference's position gradient through the n_fine_depth
//                         samples (nerf.py:292), for all points or for the depth samples only.
#include <hip/hip_runtime.h>

#include <cstring>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "pnr_common.h"
#include "pnr_device.h"
#include "pnr_internal.h"
#include "pnr_layout.h"

namespace pnr {

typedef Advance<BRS_HEAD_END, BRS_TOTAL, BRS_TOTAL> AdvanceBwd;

// ---------------------------------------------------------------- transposed weight stream
// one thread = one lane's 8-element fragment slice (16-byte store)
template <typename T>
__global__ void pack_weights_bwd_kernel(PnrMlpWeights p, T *__restrict__ out) {
    const size_t idx8 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx8 >= BWSTREAM_ELEMS_PER_WAVE / 8 * NW) return;
    const int lane = idx8 & 63;
    const int it = (idx8 >> 6) % IT;
    const size_t rest = idx8 / ((FRAG_ELEMS / 8) * IT);
    const int rs = rest % BRS_TOTAL;
    const int wv = rest / BRS_TOTAL;
    int g = 0;
    while (g + 1 < NBGEMM && rs >= bgemm_offset(g + 1)) ++g;
    const int s = rs - bgemm_offset(g);
    const int i = lane & 31, h = lane >> 5;
    // A-operand row = output row of the transposed GEMM = INPUT feature of the layer
    const int f_row = wv * SL + it * 32 + i;
    __attribute__((aligned(16))) T o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = 0.f;
        if (g == BG_Z2 || g == BG_Z1 || g == BG_Z0) {
            // d z_lat[c] += sum_f dY_b[f] W_z[b][f][c]: rows = latent channel c (natural order, wave w owns 64w..64w+63),
            // K = hidden feature f in the storage order of the gradient image
            const int b = g == BG_Z2 ? 2 : (g == BG_Z1 ? 1 : 0);
            const int f_o = feat_of(s >> 1, s & 1, 8 * h + e);
            v = p.lin_z_w[b][f_o * C_LAT + f_row];
        } else if (g == BG_IN) {
            // d(code | viewdir)[k] = sum_f dY[f] W_in[f][k]: every wave holds the FULL 64 (42 real) output rows and
            // contracts only its own 64 hidden features = storage elements 64w + 16s + 8h + e (K-split, reduced across
            // waves in LDS)
            const int k_row = it * 32 + i;
            const int f_o = feat_of(wv * IT + (s >> 1), s & 1, 8 * h + e);
            if (k_row < D_IN) v = p.lin_in_w[f_o * D_IN + k_row];
        } else if (g == BG_OUT) {
            const int k = s * 16 + h * 8 + e;  // natural order of the 4 network outputs, zero padded
            if (k < D_OUT) v = p.lin_out_w[k * D_HID + f_row];
        } else {
            const int b = 4 - (g - 1) / 2;       // BG_FC1_4, BG_FC0_4, BG_FC1_3, ... -> block index
            const bool fc1 = ((g - 1) & 1) == 0;
            const float *w = fc1 ? p.fc1_w[b] : p.fc0_w[b];
            // K index = OUTPUT feature of the layer, in the storage order of the gradient image
            const int f_o = feat_of(s >> 1, s & 1, 8 * h + e);
            v = w[f_o * D_HID + f_row];
        }
        o[e] = (T)v;
    }
    *reinterpret_cast<uint4 *>(out + idx8 * 8) = *reinterpret_cast<const uint4 *>(o);
}

// ---------------------------------------------------------------- fused data-gradient chain
struct BwdParams {
    const char *wstream;
    const unsigned long long *d_mask;    // relu bit masks of the forward (pnr_device.h: nonzero_bits8), [layer][view][tile][thread]
    const float *g_out;                  // (P,4) dL/d(lin_out output)
    float scale;
    const float *scale_dev;  // when set, the chain runs at *scale_dev instead (scale picked on the device)
    long long P;
    int NS, ntiles;
    char *g_fc1[5], *g_fc0[5], *g_x0;
    float *d_zlat;  // (NS*P, 512) fp32, natural channel order: d(interpolated latent)   (nullable: skip)
    float *d_in;    // (NS*P, 42)  fp32: d(positional code | view direction)             (nullable: skip)
    float *mv_ws;   // multi-view: per-workgroup scratch for the pooled gradient every view starts from ([slot][thread])
};

// relu masks: one 64-bit word per thread and layer from the forward kernel (bit (it*JT + jt)*16 + r = register r of the
// thread's accumulator tile (it, jt) was positive); fetched BEFORE the GEMM whose result it gates.  (Reading the masks
// from the 1 KiB dump rows instead cost the chain a third of its time: 592 -> 405 us for 49 152 points without them.)
static_assert(IT * JT * 16 == 64, "one 64-bit mask word per thread");

// acc = bit ? acc : 0
__device__ __forceinline__ void apply_mask(f32x16 (&acc)[IT][JT], unsigned long long mk) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const uint32_t m16 = (uint32_t)(mk >> ((it * JT + jt) * 16)) & 0xffffu;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[it][jt][r] = (m16 >> r) & 1u ? acc[it][jt][r] : 0.f;
        }
}

// G += bit ? t : 0
__device__ __forceinline__ void masked_add(f32x16 (&G)[IT][JT], const f32x16 (&t)[IT][JT], unsigned long long mk) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const uint32_t m16 = (uint32_t)(mk >> ((it * JT + jt) * 16)) & 0xffffu;
#pragma unroll
            for (int r = 0; r < 16; ++r) G[it][jt][r] += (m16 >> r) & 1u ? t[it][jt][r] : 0.f;
        }
}

template <typename P>
__device__ __forceinline__ void dump_only(const f32x16 (&acc)[IT][JT], char *dump_lane, const bool *valid) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            if (!valid[jt]) continue;
            const f32x16 &a = acc[it][jt];
            char *d = dump_lane + (size_t)jt * 32 * (D_HID * 2) + it * 64;
            *reinterpret_cast<typename P::T8 *>(d) = pack8<P, false>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
            *reinterpret_cast<typename P::T8 *>(d + 16) =
                pack8<P, false>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
        }
}

__device__ __forceinline__ void zero_acc(f32x16 (&a)[IT][JT]) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[it][jt][r] = 0.f;
}

// reverse of one residual block (resnetfc.py:55-62):  given G = dL/d(x + fc_1(relu(fc_0(relu(x))))),
//   dY(fc_1) = G ;  d net = (fc_1^T G) . [net > 0] = dY(fc_0) ;  G += (fc_0^T d net) . [x > 0]
template <typename P>
__device__ __forceinline__ void bwd_block(f32x16 (&G)[IT][JT], char *smem, int b, Ring<P> &R, int NS, const BwdParams &q,
                                          size_t off, const bool *valid, uint32_t a_rd0, uint32_t a_rd1, uint32_t a_wr,
                                          uint32_t mask_off, size_t mask_layer, long long rows_left, int wv, int lane) {
    constexpr bool DUMP = true;
    // gradient dumps (operands of the weight-gradient GEMMs): copied out of the image behind the barrier, whole rows
    const size_t off_tile = off - (size_t)(((lane & 31) * D_HID + (wv * IT) * 32 + (lane >> 5) * 16) * 2);
    __syncthreads();  // every wave is done reading the gradient image (previous GEMM)
    write_act<P, false, false>(G, smem, a_wr);
    __syncthreads();
    if (DUMP) dump_image<MT>(smem, LDS_A, q.g_fc1[b] + off_tile, rows_left, wv, lane);
    f32x16 t[IT][JT];
    // mask_off / mask_layer: this thread's word within a layer of q.d_mask / words per layer (layer 2b: x, 2b+1: net)
    const unsigned long long mk_n = (q.d_mask + (size_t)(2 * b + 1) * mask_layer)[mask_off];
    zero_acc(t);
    gemm<P, AdvanceBwd>(t, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
    apply_mask(t, mk_n);
    __syncthreads();
    write_act<P, false, false>(t, smem, a_wr);
    __syncthreads();
    if (DUMP) dump_image<MT>(smem, LDS_A, q.g_fc0[b] + off_tile, rows_left, wv, lane);
    const unsigned long long mk_a = (q.d_mask + (size_t)(2 * b) * mask_layer)[mask_off];
    zero_acc(t);
    gemm<P, AdvanceBwd>(t, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
    masked_add(G, t, mk_a);
}

template <int PREC, bool MV>
__global__ void __launch_bounds__(NTHREADS, NW / 4) bwd_kernel(const BwdParams q) {
    typedef Prec<PREC> P;
    typedef typename P::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    const int NS = MV ? q.NS : 1;
    const uint32_t a_rd0 = LDS_A + pl * ROW_ACT + h * 16, a_rd1 = a_rd0 + 32 * ROW_ACT;
    const uint32_t in_rd0 = LDS_IN + pl * ROW_IN + h * 16, in_rd1 = in_rd0 + 32 * ROW_IN;
    const uint32_t a_wr = LDS_A + pl * ROW_ACT + (wv * IT) * 64 + h * 32;

    Ring<P> R;
    R.wave_base = q.wstream + (size_t)wv * (BRS_TOTAL * IT * 1024) + lane * 16;
    R.pf_view = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int it = 0; it < IT; ++it) R.r[j][it] = gload8<P>(R.wave_base + j * (IT * 1024) + it * 1024);
    R.pf_rs = 4;

    for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        bool valid[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) valid[jt] = (long long)tile * MT + jt * 32 + pl < q.P;
        const size_t off_pooled = (((size_t)tile * MT + pl) * D_HID + (wv * IT) * 32 + h * 16) * 2;

        __syncthreads();  // previous tile: readers of the staging rows / gradient image are done
        {   // stage scale * g_out as 16-bit operand rows [point][64] (4 real values, zero padded)
            const int row = tid >> 3, chunk = tid & 7;
            const long long g = (long long)tile * MT + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (chunk == 0 && g < q.P) v = *reinterpret_cast<const f32x4 *>(q.g_out + g * 4) * (q.scale_dev ? *q.scale_dev : q.scale);
            if (tid < MT * 8)
                *reinterpret_cast<typename P::T8 *>(smem + LDS_IN + row * ROW_IN + chunk * 16) =
                    pack8<P, false>(v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        f32x16 G[IT][JT];
        const size_t mask_layer = (size_t)NS * (size_t)q.ntiles * NTHREADS;
        const long long rows_left = q.P - (long long)tile * MT;
        const uint32_t mask_pooled = (uint32_t)tile * NTHREADS + tid;  // 32 bits: NS * tiles * 512 < 2^32 (host check)
        {
            const unsigned long long mk = (q.d_mask + (size_t)10 * mask_layer)[mask_pooled];
            zero_acc(G);
            gemm<P, AdvanceBwd>(G, smem, in_rd0, in_rd1, KS_IN / 4, R, NS);  // lin_out^T g_out
            apply_mask(G, mk);                                                // . [x5 > 0]
        }
#pragma unroll 1
        for (int b = N_BLOCKS - 1; b >= COMBINE_LAYER; --b)
            bwd_block<P>(G, smem, b, R, NS, q, off_pooled, valid, a_rd0, a_rd1, a_wr, mask_pooled, mask_layer, rows_left, wv, lane);
        // backward of the view mean (util.py:461-466): every view starts from G / NS.  That gradient is PARKED in a
        // per-workgroup scratch ([slot][thread]: every lane re-reads what it wrote, L2-resident) instead of 64 registers held
        // across the whole per-view loop -- the in-register form spilled 106 registers.
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
            const size_t off_view = off_pooled + (size_t)view * (size_t)q.P * (D_HID * 2);
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
#pragma unroll 1
            for (int b = COMBINE_LAYER - 1; b >= 0; --b)
                bwd_block<P>(G, smem, b, R, NS, q, off_view, valid, a_rd0, a_rd1, a_wr,
                             mask_pooled + (uint32_t)view * (uint32_t)q.ntiles * NTHREADS, mask_layer, rows_left, wv, lane);
            // ---- d z_lat = sum_b dY_b W_z[b] and d(code) = dY_0 W_in (resnetfc.py:147,175-180 backward): four more
            // transposed-stream GEMMs on gradient images this tile has just produced.  dY_2, dY_1 (= g_fc1[1], g_fc1[0])
            // come back from their dumps (written by this workgroup a moment ago, L2-resident), dY_0 = G is in registers.
            const float inv_scale = 1.f / (q.scale_dev ? *q.scale_dev : q.scale);
            f32x16 Z[IT][JT];
            zero_acc(Z);
#pragma unroll 1
            for (int b = COMBINE_LAYER - 1; b >= 1; --b) {
                __threadfence_block();
                __syncthreads();  // the image's readers are done; the dump rows of this tile are visible
                {
                    const char *src = q.g_fc1[b - 1] + (size_t)view * (size_t)q.P * (D_HID * 2);
#pragma unroll
                    for (int u = 0; u < MT / NW; ++u) {
                        const int row = wv * (MT / NW) + u;
                        const long long g = (long long)tile * MT + row;
                        u32x4 v = {0, 0, 0, 0};
                        if (g < q.P) v = *reinterpret_cast<const u32x4 *>(src + (size_t)g * (D_HID * 2) + lane * 16);
                        *reinterpret_cast<u32x4 *>(smem + LDS_A + row * ROW_ACT + lane * 16) = v;
                    }
                }
                __syncthreads();
                gemm<P, AdvanceBwd>(Z, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);  // lin_z[b]^T dY_b
            }
            __syncthreads();
            write_act<P, false, false>(G, smem, a_wr);  // dY of lin_in and lin_z[0]
            __syncthreads();
            dump_image<MT>(smem, LDS_A, q.g_x0 + (size_t)view * (size_t)q.P * (D_HID * 2) + (size_t)tile * MT * (D_HID * 2), rows_left,
                           wv, lane);
            gemm<P, AdvanceBwd>(Z, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);      // lin_z[0]^T dY_0
            {   // accumulator (channel 32T + (r&3) + 8(r>>2) + 4h, point) -> fp32 rows, 16-byte pieces
                float *dst = q.d_zlat + ((size_t)view * (size_t)q.P + (size_t)tile * MT + pl) * C_LAT + (wv * IT) * 32 + 4 * h;
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if (!valid[jt]) continue;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x4 v = {Z[it][jt][4 * k], Z[it][jt][4 * k + 1], Z[it][jt][4 * k + 2], Z[it][jt][4 * k + 3]};
                            __builtin_nontemporal_store(v * inv_scale, reinterpret_cast<f32x4 *>(dst + (size_t)jt * 32 * C_LAT + it * 32 + 8 * k));  // streamed past the L2 (dump_image)
                        }
                    }
            }
            // lin_in^T, K-split: this wave's 64 hidden features = bytes [128 wv, 128 wv + 128) of every image row
            zero_acc(Z);
            gemm<P, AdvanceBwd>(Z, smem, a_rd0 + wv * 128, a_rd1 + wv * 128, KS_IN / 4, R, NS);
            __syncthreads();  // every wave is done with the image: its space (and LDS_Z below it) takes the partials
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
                const long long g = (long long)tile * MT + pnt;
                if (g < q.P) {
#pragma unroll
                    for (int k = k0; k < k0 + 8; ++k) {
                        if (k >= D_IN) break;
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) sum += reinterpret_cast<const float *>(smem)[(size_t)w * (MT * 66) + pnt * 66 + k];
                        q.d_in[((size_t)view * (size_t)q.P + (size_t)g) * D_IN + k] = sum * inv_scale;
                    }
                }
            }
            __syncthreads();  // the partials are consumed before the next view / tile reuses the space
        }
    }
}

// ---------------------------------------------------------------- weight-gradient GEMM
// dW[o][k] = sum_r dY[r][o] X[r][k]   over 16-bit row-major dumps (rows,512) / (rows,512), fp32 out.
// The reduction index r is the slow dimension of both operands.  Slabs of 32 rows are staged in LDS AS
// THEY LIE IN MEMORY (row-major, 16-byte global loads -> 16-byte LDS writes) and the MFMA fragments
// (8 consecutive rows of one column) come out of gfx950's transposing LDS read ds_read_b64_tr_b16:
// per 16-lane group a [4 rows][16 columns] block, lane c receives column c.  Row stride 576 B puts the
// 4 rows of a block and the two blocks of a 32-lane half on disjoint banks.
// Block = 8 waves = one 256x256 tile of dW (wave = 64x128 = 2x4 MFMA tiles: every operand element is
// fetched from L2 twice, not four times as with 128x128 tiles); the 1-D grid enumerates (tile, slice of the
// rows = split-K, job): all linears of a network go in ONE launch.  Every slice writes its own
// partial (part[job][z][512][512], bpart[job][z][512]) with plain stores and dw_reduce_kernel sums them in
// a fixed order -> bit-reproducible, no atomics.  bpart = bias gradient sum_r dY[r][o].
constexpr int DW_MAX_JOBS = 16;
struct DwJobs {
    const void *dY[DW_MAX_JOBS], *X[DW_MAX_JOBS];
    float *dW[DW_MAX_JOBS], *db[DW_MAX_JOBS];
    long long rows[DW_MAX_JOBS];
    short nx[DW_MAX_JOBS];   // columns of X (= its row stride): 512, or 64 for the lin_in operand
    short ncw[DW_MAX_JOBS];  // columns of dW written (= its row stride): 512, or 42 for lin_in
    unsigned char rows_st[DW_MAX_JOBS], cols_st[DW_MAX_JOBS];
    int nsplit;
};

template <typename T8, int LDB>
__device__ __forceinline__ T8 tr_frag(const char *smem_row_col) {
    // two transposing reads: rows +0..3 and +4..7 (row stride LDB bytes) of this lane's column
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s4 *lds_s4;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(smem_row_col));
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(smem_row_col + 4 * LDB));
    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(T8, v);
}

template <int PREC>
__global__ void __launch_bounds__(512)
dw_kernel(const DwJobs jobs, float *__restrict__ part, float *__restrict__ bpart) {
    typedef Prec<PREC> P;
    typedef typename P::T T;
    constexpr int SR = 32, LDB = 576;  // rows per slab, LDS row stride in bytes (256 columns + 64 B pad: 144 dwords = 16 mod 64)
    __shared__ __attribute__((aligned(16))) char sYb[2][SR * LDB];  // double-buffered: slab s+1 is written while s is multiplied
    __shared__ __attribute__((aligned(16))) char sXb[2][SR * LDB];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // XCD-aware placement.  The four 256x256 tiles of one (job, row slice) read the SAME dY / X rows; workgroups are dealt
    // to the 8 XCDs round-robin by linear id, so a (tile, slice, job) grid puts the four on four different L2s and every
    // operand byte crosses the fabric twice.  Here the linear id is re-read as (XCD, k): four consecutive k of one XCD
    // are the four tiles of a group, so the group's operands are fetched into ONE L2 once.
    const int lid = blockIdx.x, ngroups = gridDim.x >> 2, full = (ngroups >> 3) * 32;
    int grp, tile4;
    if (lid < full) { const int k = lid >> 3; grp = (k >> 2) * 8 + (lid & 7); tile4 = k & 3; }
    else { const int rem = lid - full; grp = (full >> 2) + (rem >> 2); tile4 = rem & 3; }  // last partial row of groups
    const int job = grp / jobs.nsplit, slice = grp - job * jobs.nsplit;
    const T *dY = reinterpret_cast<const T *>(jobs.dY[job]);
    const T *X = reinterpret_cast<const T *>(jobs.X[job]);
    const long long rows = jobs.rows[job];
    const int nx = jobs.nx[job];
    long long per = (rows + jobs.nsplit - 1) / jobs.nsplit;
    per = (per + SR - 1) / SR * SR;
    const int o0 = (tile4 >> 1) * 256, k0 = (tile4 & 1) * 256;
    if (k0 >= nx) return;  // narrow X (lin_in): only the first column tile exists
    const long long r_begin = (long long)slice * per;
    const long long r_end = r_begin + per < rows ? r_begin + per : rows;
    const int wo = (w >> 1) * 64, wk = (w & 1) * 128;  // wave tile: 64 (o) x 128 (k) = 2 x 4 MFMA tiles
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc[2][4];
    float bsum[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // register-staged software pipeline: the global loads of slab s+1 are in flight while slab s is
    // being multiplied out of LDS.  slab = 32 rows x 32 chunks of 8 columns per operand; 2 chunks per thread
    u32x4 vy[2], vx[2];
    auto load_slab = [&](long long r0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int chunk = t + u * 512;  // consecutive lanes -> consecutive 16-byte chunks of a row
            const int srow = chunk >> 5, scol = (chunk & 31) * 8;
            vy[u] = u32x4{0, 0, 0, 0};
            vx[u] = u32x4{0, 0, 0, 0};
            if (r0 + srow < r_end) {
                vy[u] = *reinterpret_cast<const u32x4 *>(dY + (r0 + srow) * D_HID + o0 + scol);
                if (k0 + scol < nx) vx[u] = *reinterpret_cast<const u32x4 *>(X + (r0 + srow) * nx + k0 + scol);
            }
        }
    };
    // this lane's corner of the [4][16] transpose blocks: row 8kh + (c16>>2), column 16*((lane>>4)&1) + 4*(c16&3)
    const int c16 = lane & 15;
    const int frag_off = (8 * kh + (c16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (c16 & 3)) * 2;
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int chunk = t + u * 512;
            const int srow = chunk >> 5, scol = (chunk & 31) * 8;
            *reinterpret_cast<u32x4 *>(sYb[buf] + srow * LDB + scol * 2) = vy[u];
            *reinterpret_cast<u32x4 *>(sXb[buf] + srow * LDB + scol * 2) = vx[u];
        }
    };
    if (r_begin < r_end) {
        load_slab(r_begin);
        store_slab(0);
    }
    __syncthreads();
    int cur = 0;
    for (long long r0 = r_begin; r0 < r_end; r0 += SR, cur ^= 1) {
        const bool more = r0 + SR < r_end;
        if (more) load_slab(r0 + SR);  // in flight under this slab's MFMAs
        const char *sY = sYb[cur], *sX = sXb[cur];
#pragma unroll
        for (int ks = 0; ks < SR / 16; ++ks) {
            typename P::T8 af[2], bf[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag<typename P::T8, LDB>(sY + ks * 16 * LDB + (wo + a * 32) * 2 + frag_off);
#pragma unroll
            for (int b = 0; b < 4; ++b) bf[b] = tr_frag<typename P::T8, LDB>(sX + ks * 16 * LDB + (wk + b * 32) * 2 + frag_off);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(af[a], bf[b], acc[a][b]);
            if ((tile4 & 1) == 0 && (w & 1) == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[a] += (float)af[a][e];
            }
        }
        if (more) store_slab(cur ^ 1);  // the other buffer: its readers passed the barrier of the previous slab
        __syncthreads();
    }
    // D layout: column j = lane&31 -> k, row (r&3)+8(r>>2)+4kh -> o
    float *pz = part + ((size_t)job * jobs.nsplit + slice) * (D_HID * D_HID);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = o0 + wo + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                pz[(size_t)orow * D_HID + k0 + wk + b * 32 + i] = acc[a][b][r];
            }
    if ((tile4 & 1) == 0 && (w & 1) == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);  // both row halves of every k-step
            if (kh == 0) bpart[((size_t)job * jobs.nsplit + slice) * D_HID + o0 + wo + a * 32 + i] = v;
        }
    }
}

// The same weight-gradient GEMM at split-operand (fp32-class) precision: both operands arrive as (head, tail) f16 row sets -- the
// activation images of the fused split forward and the gradient images of the fused split chain, copied out of LDS as they lie
// there; the tail array of an operand follows its head array -- and every fragment pair costs three MFMAs (head x head,
// head x tail, tail x head; fp32 accumulate), the arithmetic of PNR_PREC_F16X3.  Same tiling, placement, slice reduction and
// storage -> feature order mapping as dw_kernel; four operand slabs (dY / X, head / tail), double-buffered: 147 KiB of LDS, one
// workgroup of 8 waves per CU -- the kernel is MFMA-bound (3 MFMAs per 2 fragment reads), which 2 waves per SIMD cover.
__global__ void __launch_bounds__(512)
dw_split_kernel(const DwJobs jobs, float *__restrict__ part, float *__restrict__ bpart) {
    typedef Prec<PNR_PREC_F16> P;
    typedef _Float16 T;
    constexpr int SR = 32, LDB = 576;
    extern __shared__ __attribute__((aligned(16))) char dws[];  // [buf][dY head, dY tail, X head, X tail][SR * LDB]
    constexpr int SLAB = SR * LDB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lid = blockIdx.x, ngroups = gridDim.x >> 2, full = (ngroups >> 3) * 32;
    int grp, tile4;
    if (lid < full) { const int k = lid >> 3; grp = (k >> 2) * 8 + (lid & 7); tile4 = k & 3; }
    else { const int rem = lid - full; grp = (full >> 2) + (rem >> 2); tile4 = rem & 3; }
    const int job = grp / jobs.nsplit, slice = grp - job * jobs.nsplit;
    const long long rows = jobs.rows[job];
    const int nx = jobs.nx[job];
    const T *dYh = reinterpret_cast<const T *>(jobs.dY[job]), *dYl = dYh + (size_t)rows * D_HID;
    const T *Xh = reinterpret_cast<const T *>(jobs.X[job]), *Xl = Xh + (size_t)rows * nx;
    long long per = (rows + jobs.nsplit - 1) / jobs.nsplit;
    per = (per + SR - 1) / SR * SR;
    const int o0 = (tile4 >> 1) * 256, k0 = (tile4 & 1) * 256;
    if (k0 >= nx) return;  // narrow X (lin_in): only the first column tile exists
    const long long r_begin = (long long)slice * per;
    const long long r_end = r_begin + per < rows ? r_begin + per : rows;
    const int wo = (w >> 1) * 64, wk = (w & 1) * 128;  // wave tile: 64 (o) x 128 (k) = 2 x 4 MFMA tiles
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc[2][4];
    float bsum[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    u32x4 vyh[2], vyl[2], vxh[2], vxl[2];
    auto load_slab = [&](long long r0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int chunk = t + u * 512;  // consecutive lanes -> consecutive 16-byte chunks of a row
            const int srow = chunk >> 5, scol = (chunk & 31) * 8;
            vyh[u] = vyl[u] = vxh[u] = vxl[u] = u32x4{0, 0, 0, 0};
            if (r0 + srow < r_end) {
                const size_t oy = (size_t)(r0 + srow) * D_HID + o0 + scol;
                vyh[u] = *reinterpret_cast<const u32x4 *>(dYh + oy);
                vyl[u] = *reinterpret_cast<const u32x4 *>(dYl + oy);
                if (k0 + scol < nx) {
                    const size_t ox = (size_t)(r0 + srow) * nx + k0 + scol;
                    vxh[u] = *reinterpret_cast<const u32x4 *>(Xh + ox);
                    vxl[u] = *reinterpret_cast<const u32x4 *>(Xl + ox);
                }
            }
        }
    };
    const int c16 = lane & 15;
    const int frag_off = (8 * kh + (c16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (c16 & 3)) * 2;
    auto store_slab = [&](int buf) {
        char *base = dws + buf * (4 * SLAB);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int chunk = t + u * 512;
            const int off = (chunk >> 5) * LDB + (chunk & 31) * 16;
            *reinterpret_cast<u32x4 *>(base + off) = vyh[u];
            *reinterpret_cast<u32x4 *>(base + SLAB + off) = vyl[u];
            *reinterpret_cast<u32x4 *>(base + 2 * SLAB + off) = vxh[u];
            *reinterpret_cast<u32x4 *>(base + 3 * SLAB + off) = vxl[u];
        }
    };
    if (r_begin < r_end) {
        load_slab(r_begin);
        store_slab(0);
    }
    __syncthreads();
    int cur = 0;
    for (long long r0 = r_begin; r0 < r_end; r0 += SR, cur ^= 1) {
        const bool more = r0 + SR < r_end;
        if (more) load_slab(r0 + SR);  // in flight under this slab's MFMAs
        const char *sYh = dws + cur * (4 * SLAB), *sYl = sYh + SLAB, *sXh = sYh + 2 * SLAB, *sXl = sYh + 3 * SLAB;
#pragma unroll
        for (int ks = 0; ks < SR / 16; ++ks) {
            P::T8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = tr_frag<P::T8, LDB>(sYh + ks * 16 * LDB + (wo + a * 32) * 2 + frag_off);
                al[a] = tr_frag<P::T8, LDB>(sYl + ks * 16 * LDB + (wo + a * 32) * 2 + frag_off);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                bh[b] = tr_frag<P::T8, LDB>(sXh + ks * 16 * LDB + (wk + b * 32) * 2 + frag_off);
                bl[b] = tr_frag<P::T8, LDB>(sXl + ks * 16 * LDB + (wk + b * 32) * 2 + frag_off);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(ah[a], bh[b], acc[a][b]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(ah[a], bl[b], acc[a][b]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(al[a], bh[b], acc[a][b]);
            if ((tile4 & 1) == 0 && (w & 1) == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[a] += (float)ah[a][e] + (float)al[a][e];
            }
            // the next slab's rows (requested at the top of this slab, one k-step = 48 MFMAs per SIMD ago) go into the OTHER
            // buffer -- its readers passed the barrier of the previous slab -- between the two k-steps: the 8 LDS stores per
            // thread ride under the second k-step's MFMAs instead of standing between the last MFMA and the barrier
            // (same-box A/B against storing after the last MFMA: 907.4 / 908.0 vs 918.8 / 914.4 us per launch, -1 %;
            // profiles/r05_train_step_notes.md)
            if (ks == 0 && more) store_slab(cur ^ 1);
        }
        __syncthreads();
    }
    float *pz = part + ((size_t)job * jobs.nsplit + slice) * (D_HID * D_HID);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = o0 + wo + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                pz[(size_t)orow * D_HID + k0 + wk + b * 32 + i] = acc[a][b][r];
            }
    if ((tile4 & 1) == 0 && (w & 1) == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (kh == 0) bpart[((size_t)job * jobs.nsplit + slice) * D_HID + o0 + wo + a * 32 + i] = v;
        }
    }
}

// dw_split_kernel as ONE wave per SIMD (round 6; the launcher's default): 4 waves of 128 (o) x 128 (k) = 4 x 4 MFMA tiles,
// 256 accumulator registers per lane (the AGPR half of the 512-register file a 256-thread workgroup owns), 16 fragments per k-step
// for 48 MFMAs (the 8-wave form: 12 for 24 -- a third less LDS traffic per MFMA), and room for TWO fragment sets: the fragments of
// k-step j + 1 are read while the MFMAs of k-step j issue, the slab barrier sits BETWEEN the two k-steps of a slab (every LDS read
// of a slab is issued before it, so the other buffer may be overwritten right behind it), and no k-step starts with an LDS burst of
// all waves behind a barrier.  What the 8-wave form does there (ISA): 24 transposing reads at the top of every k-step, waited for,
// then 24 MFMAs; after the barrier of every second k-step all 8 waves read at once -- 96 KiB through the 128 B/clk LDS pipe = 768
// clocks against 1536 clocks of MFMAs per k-step and SIMD.
__global__ void __launch_bounds__(256)
dw_split_wide_kernel(const DwJobs jobs, float *__restrict__ part, float *__restrict__ bpart) {
    typedef Prec<PNR_PREC_F16> P;
    typedef _Float16 T;
    constexpr int SR = 32, LDB = 576;
    extern __shared__ __attribute__((aligned(16))) char dws[];  // [buf][dY head, dY tail, X head, X tail][SR * LDB]
    constexpr int SLAB = SR * LDB;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int lid = blockIdx.x, ngroups = gridDim.x >> 2, full = (ngroups >> 3) * 32;
    int grp, tile4;
    if (lid < full) { const int k = lid >> 3; grp = (k >> 2) * 8 + (lid & 7); tile4 = k & 3; }
    else { const int rem = lid - full; grp = (full >> 2) + (rem >> 2); tile4 = rem & 3; }
    const int job = grp / jobs.nsplit, slice = grp - job * jobs.nsplit;
    const long long rows = jobs.rows[job];
    const int nx = jobs.nx[job];
    const T *dYh = reinterpret_cast<const T *>(jobs.dY[job]), *dYl = dYh + (size_t)rows * D_HID;
    const T *Xh = reinterpret_cast<const T *>(jobs.X[job]), *Xl = Xh + (size_t)rows * nx;
    long long per = (rows + jobs.nsplit - 1) / jobs.nsplit;
    per = (per + SR - 1) / SR * SR;
    const int o0 = (tile4 >> 1) * 256, k0 = (tile4 & 1) * 256;
    if (k0 >= nx) return;  // narrow X (lin_in): only the first column tile exists
    const long long r_begin = (long long)slice * per;
    const long long r_end = r_begin + per < rows ? r_begin + per : rows;
    const int wo = (w >> 1) * 128, wk = (w & 1) * 128;  // wave tile: 128 (o) x 128 (k) = 4 x 4 MFMA tiles
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc[4][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // Staging: 4 x (dY head, dY tail, X head, X tail) 16-byte chunks per thread.  The pipelined loop takes FULL slabs only: its loads
    // are unconditional (uniform base per slab + 32-bit lane offset) and go to LDS as they arrive -- no exec-mask branches and no
    // selects, so the body is one basic block whose issue order is pinned, and nothing but the LDS stores of the NEXT k-step waits
    // for a load.  A narrow X (lin_in: 64 of the 256 columns exist) reads column 0 in place of the missing ones: those columns of
    // the partial products are never read (dw_reduce_kernel stops at the job's width).  A last partial slab of the slice (rows not
    // a multiple of 32) runs behind the loop with predicated loads.
    u32x4 vyh[4], vyl[4], vxh[4], vxl[4];
    unsigned offy[4], offx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int chunk = t + u * 256;  // consecutive lanes -> consecutive 16-byte chunks of a row
        const int srow = chunk >> 5, scol = (chunk & 31) * 8;
        offy[u] = (unsigned)(srow * D_HID + scol);
        offx[u] = (unsigned)(srow * nx + (k0 + scol < nx ? scol : 0));
    }
    const int lds_off = (t >> 5) * LDB + (t & 31) * 16;  // chunk u sits 8 rows further down
    const size_t tail_y = (size_t)rows * D_HID, tail_x = (size_t)rows * nx;
    auto load_slab = [&](long long r0) {
        const T *by = dYh + (size_t)r0 * D_HID + o0, *bx = Xh + (size_t)r0 * nx + k0;  // uniform
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            vyh[u] = *reinterpret_cast<const u32x4 *>(by + offy[u]);
            vyl[u] = *reinterpret_cast<const u32x4 *>(by + tail_y + offy[u]);
            vxh[u] = *reinterpret_cast<const u32x4 *>(bx + offx[u]);
            vxl[u] = *reinterpret_cast<const u32x4 *>(bx + tail_x + offx[u]);
        }
    };
    auto store_slab = [&](int buf) {
        char *base = dws + buf * (4 * SLAB) + lds_off;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<u32x4 *>(base + u * 8 * LDB) = vyh[u];
            *reinterpret_cast<u32x4 *>(base + SLAB + u * 8 * LDB) = vyl[u];
            *reinterpret_cast<u32x4 *>(base + 2 * SLAB + u * 8 * LDB) = vxh[u];
            *reinterpret_cast<u32x4 *>(base + 3 * SLAB + u * 8 * LDB) = vxl[u];
        }
    };
    const int c16 = lane & 15;
    const int frag_off = (8 * kh + (c16 >> 2)) * LDB + (16 * ((lane >> 4) & 1) + 4 * (c16 & 3)) * 2;
    struct Frags { P::T8 ah[4], al[4], bh[4], bl[4]; };
    auto read_frags = [&](Frags &f, int buf, int ks) {
        const char *sYh = dws + buf * (4 * SLAB) + ks * 16 * LDB + frag_off, *sYl = sYh + SLAB, *sXh = sYh + 2 * SLAB, *sXl = sYh + 3 * SLAB;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f.ah[a] = tr_frag<P::T8, LDB>(sYh + (wo + a * 32) * 2);
            f.al[a] = tr_frag<P::T8, LDB>(sYl + (wo + a * 32) * 2);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            f.bh[b] = tr_frag<P::T8, LDB>(sXh + (wk + b * 32) * 2);
            f.bl[b] = tr_frag<P::T8, LDB>(sXl + (wk + b * 32) * 2);
        }
    };
    const bool sums = (tile4 & 1) == 0 && (w & 1) == 0;
    auto mfmas = [&](const Frags &f) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(f.ah[a], f.bh[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(f.ah[a], f.bl[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = P::mfma(f.al[a], f.bh[b], acc[a][b]);
        // column sums of dY (the bias gradient), formed in every wave -- a uniform branch would cut the pinned block in two -- and
        // written by the waves that own them: v_dot2_f32_f16 against (1, 1), fp32 accumulate, 8 VALU per fragment pair
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 ones = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                bsum[a] = __builtin_amdgcn_fdot2(h2{f.ah[a][e], f.ah[a][e + 1]}, ones, bsum[a], false);
                bsum[a] = __builtin_amdgcn_fdot2(h2{f.al[a][e], f.al[a][e + 1]}, ones, bsum[a], false);
            }
    };
    const int nrows = r_begin < r_end ? (int)(r_end - r_begin) : 0;
    const int nfull = nrows / SR;  // slabs of the pipelined loop
    if (nfull > 0) {
        Frags f0, f1;
        load_slab(r_begin);
        store_slab(0);
        __syncthreads();
        load_slab(r_begin + (long long)min(1, nfull - 1) * SR);
        read_frags(f0, 0, 0);
#pragma unroll 1
        for (int sl = 0; sl + 1 < nfull; ++sl) {
            const int cur = sl & 1;
            // k-step 0 of slab sl: its fragments were read under the previous k-step; read k-step 1's, then park slab sl + 1 (requested
            // a whole k-step ago) in the other buffer -- its last readers waited for their data in front of the previous barrier
            // k-step 0 of slab sl: its fragments were read under the previous k-step.  Read k-step 1's (32 slots), then park slab sl + 1 --
            // requested two k-steps ago -- in the other buffer (its last readers waited for their data in front of the previous barrier)
            // and request slab sl + 2 into every staging register right behind the LDS store that read it: 96 MFMA slots in flight
            // (requested behind the barrier instead -- 64 slots -- the launch was 2.5 % slower: profiles/r06_dw_split_notes.md)
            read_frags(f1, cur, 1);
            store_slab(cur ^ 1);
            load_slab(r_begin + (long long)min(sl + 2, nfull - 1) * SR);  // (the last iteration re-requests the last slab: no branch)
            mfmas(f0);
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 transposing read
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);  // 1 VALU (addresses, bias sums)
            }
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // 1 LDS store
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 global load into the register it read
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            __syncthreads();  // slab sl + 1 is visible; nobody reads buffer `cur` any more
            read_frags(f0, cur ^ 1, 0);
            mfmas(f1);
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
        // the last full slab
        read_frags(f1, (nfull - 1) & 1, 1);
        mfmas(f0);
        mfmas(f1);
    }
    if (nrows > nfull * SR) {  // the slice's last, partial slab (rows not a multiple of 32): predicated loads, no pipelining
        __syncthreads();
        const long long r0 = r_begin + (long long)nfull * SR;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int srow = (t + u * 256) >> 5;
            vyh[u] = vyl[u] = vxh[u] = vxl[u] = u32x4{0, 0, 0, 0};
            if (r0 + srow < r_end) {
                const T *by = dYh + (size_t)r0 * D_HID + o0, *bx = Xh + (size_t)r0 * nx + k0;
                vyh[u] = *reinterpret_cast<const u32x4 *>(by + offy[u]);
                vyl[u] = *reinterpret_cast<const u32x4 *>(by + tail_y + offy[u]);
                vxh[u] = *reinterpret_cast<const u32x4 *>(bx + offx[u]);
                vxl[u] = *reinterpret_cast<const u32x4 *>(bx + tail_x + offx[u]);
            }
        }
        store_slab(0);
        __syncthreads();
        Frags f;
        read_frags(f, 0, 0);
        mfmas(f);
        read_frags(f, 0, 1);
        mfmas(f);
    }
    float *pz = part + ((size_t)job * jobs.nsplit + slice) * (D_HID * D_HID);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = o0 + wo + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                pz[(size_t)orow * D_HID + k0 + wk + b * 32 + i] = acc[a][b][r];
            }
    if (sums) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (kh == 0) bpart[((size_t)job * jobs.nsplit + slice) * D_HID + o0 + wo + a * 32 + i] = v;
        }
    }
}

// dW = scale * sum_z part[job][z], db = scale * sum_z bpart[job][z]  (fixed summation order); blockIdx.y = job.
// rows_st / cols_st: the operand that indexes the rows (dY) / columns (X) of dW was dumped in storage order;
// the result is written in feature order (row e -> feature feat_of(e/32, (e%32)/16, e%16)).
__global__ void dw_reduce_kernel(const DwJobs jobs, const float *__restrict__ part, const float *__restrict__ bpart,
                                 float scale, const float *__restrict__ scale_dev) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int job = blockIdx.y, nz = jobs.nsplit;
    if (scale_dev) scale *= *scale_dev;
    const float *pj = part + (size_t)job * nz * (D_HID * D_HID);
    const float *bj = bpart + (size_t)job * nz * D_HID;
    const bool rows_st = jobs.rows_st[job], cols_st = jobs.cols_st[job];
    if (idx < D_HID * D_HID && (idx % D_HID) < ((jobs.nx[job] + 255) / 256) * 256) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += pj[(size_t)z * (D_HID * D_HID) + idx];
        int r = idx / D_HID, c = idx % D_HID;
        if (rows_st) r = feat_of(r >> 5, (r >> 4) & 1, r & 15);
        if (cols_st) c = feat_of(c >> 5, (c >> 4) & 1, c & 15);
        const int ncw = jobs.ncw[job];
        if (c < ncw) jobs.dW[job][r * ncw + c] = s * scale;
    }
    if (jobs.db[job] && idx < D_HID) {
        float s = 0.f;
        for (int z = 0; z < nz; ++z) s += bj[(size_t)z * D_HID + idx];
        jobs.db[job][rows_st ? feat_of(idx >> 5, (idx >> 4) & 1, idx & 15) : idx] = s * scale;
    }
}

// ---------------------------------------------------------------- compositing backward
constexpr int CW = 4;  // wavefronts per block

template <typename T> __device__ __forceinline__ T wsum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wscan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ float wscan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// w_i = a_i T_i, T_i = prod_{j<i} (1 - a_j + 1e-10), a_i = 1 - exp(-delta_i relu(sigma_i))
// g_i = dL/dw_i = d_rgb.c_i + d_depth z_i + d_w_i - [white] sum(d_rgb)
// dL/da_i = g_i T_i - (sum_{j>i} g_j w_j) / (1 - a_i + 1e-10)
template <int PASS>
__device__ __forceinline__ float composite_bwd_pass(const float *zr, const float4 *cr, const float *dwr, int K, float far,
                                                    float3 drgb, float ddepth, float gwhite, int lane, float total,
                                                    float4 *dout, float *dzout, bool preact) {
    float carry = 1.f, run = 0.f, acc = 0.f, ddelta_prev = 0.f;
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < K;
        float zi = 0.f, alpha = 0.f, delta = 0.f, ex = 1.f;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            zi = zr[i];
            delta = ((i + 1 < K) ? zr[i + 1] : far) - zi;
            cs = cr[i];
            ex = expf(-delta * fmaxf(cs.w, 0.f));
            alpha = 1.f - ex;
        }
        const float tf = valid ? (1.f - alpha + 1e-10f) : 1.f;
        const float incl = wscan_mul(tf, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        const float w = alpha * T;
        float g = 0.f;
        if (valid) g = drgb.x * cs.x + drgb.y * cs.y + drgb.z * cs.z + ddepth * zi + (dwr ? dwr[i] : 0.f) - gwhite;
        const float gw = valid ? g * w : 0.f;
        if (PASS == 0) {
            acc += gw;
        } else {
            const float pre = wscan_add(gw, lane) + run;   // sum_{j<=i} g_j w_j
            const float suffix = total - pre;              // sum_{j>i}
            float ddelta = 0.f;
            if (valid) {
                const float dalpha = g * T - suffix / tf;
                const float dsigma = cs.w > 0.f ? dalpha * delta * ex : 0.f;
                float4 go = make_float4(w * drgb.x, w * drgb.y, w * drgb.z, dsigma);
                if (preact) {  // through rgb = sigmoid(.), sigma = relu(.) (models.py:260-263): relu' already applied
                    go.x *= cs.x * (1.f - cs.x); go.y *= cs.y * (1.f - cs.y); go.z *= cs.z * (1.f - cs.z);
                }
                dout[i] = go;
                ddelta = dalpha * fmaxf(cs.w, 0.f) * ex;  // d alpha_i / d delta_i = relu(sigma) exp(-delta relu(sigma))
            }
            if (dzout) {
                // delta_i = z_{i+1} - z_i (last: far - z_i), depth = sum w z:
                //   dL/dz_i = w_i d_depth - ddelta_i + ddelta_{i-1}
                float up = __shfl_up(ddelta, 1, 64);
                if (lane == 0) up = ddelta_prev;
                if (valid) dzout[i] = w * ddepth - ddelta + up;
                ddelta_prev = __shfl(ddelta, 63, 64);
            }
            run = __shfl(pre, 63, 64);
        }
        carry = carry * __shfl(incl, 63, 64);
    }
    return PASS == 0 ? wsum(acc) : 0.f;
}

__global__ void __launch_bounds__(CW * 64)
composite_bwd_kernel(const float *__restrict__ rays, const float *__restrict__ z, const float4 *__restrict__ rgbs, int R,
                     int K, int white_bkgd, const float *__restrict__ d_rgb, const float *__restrict__ d_depth,
                     const float *__restrict__ d_w, float4 *__restrict__ d_rgbs, float *__restrict__ d_z, int preact) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * CW + wv;
    if (r >= R) return;
    const float far = rays[(size_t)r * 8 + 7];
    const float3 drgb = make_float3(d_rgb[(size_t)r * 3], d_rgb[(size_t)r * 3 + 1], d_rgb[(size_t)r * 3 + 2]);
    const float ddepth = d_depth ? d_depth[r] : 0.f;
    const float gwhite = white_bkgd ? (drgb.x + drgb.y + drgb.z) : 0.f;  // rgb += 1 - sum w
    const float *zr = z + (size_t)r * K;
    const float4 *cr = rgbs + (size_t)r * K;
    const float *dwr = d_w ? d_w + (size_t)r * K : nullptr;
    const float total = composite_bwd_pass<0>(zr, cr, dwr, K, far, drgb, ddepth, gwhite, lane, 0.f, nullptr, nullptr, false);
    composite_bwd_pass<1>(zr, cr, dwr, K, far, drgb, ddepth, gwhite, lane, total, d_rgbs + (size_t)r * K,
                          d_z ? d_z + (size_t)r * K : nullptr, preact != 0);
}

// ---------------------------------------------------------------- latent scatter-add
// (global fp32 atomics: since round 6 only the fallback for objects of 2^29+ samples, grids of > 8192 tiles and PIXELNERF_SCATTER_TILED=0 --
// every other grid takes the LDS-slab forms further down.)  One wavefront per (view, run of SCATTER_RUN consecutive points); lane handles channels 8*lane..+7.
// Consecutive samples of a ray mostly fall into the same grid cell, so each of the 4 bilinear
// corners keeps a register accumulator that is flushed with atomics only when its texel changes
// (run-length merging: ~5x fewer atomics on the 32x32 sn64 grid).
constexpr int SCATTER_RUN = 16;
#pragma clang fp contract(off)
__global__ void __launch_bounds__(CW * 64)
latent_scatter_kernel(const EvalParams q, const float *__restrict__ d_zlat, float *__restrict__ d_latent) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long runs_per_view = (q.P + SCATTER_RUN - 1) / SCATTER_RUN;
    const long long run = (long long)blockIdx.x * CW + wv;
    if (run >= runs_per_view * q.NS) return;
    const int view = (int)(run / runs_per_view);
    const long long g0 = (run % runs_per_view) * SCATTER_RUN;
    float acc[4][8];
    uint32_t cur[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
    for (int j = 0; j < SCATTER_RUN; ++j) {
        const long long gl = g0 + j;
        if (gl >= q.P) break;
        const int g = (int)gl;
        const int r = g / q.K;
        const float *ray = q.rays + (size_t)r * 8;
        const float zz = q.z[g];
        const float X = ray[0] + zz * ray[3], Y = ray[1] + zz * ray[4], Z = ray[2] + zz * ray[5];
        const int obj = r / q.per_obj;
        const float *pose = q.poses + (size_t)(obj * q.NS + view) * 12;
        const float xr0 = pose[0] * X + pose[1] * Y + pose[2] * Z;
        const float xr1 = pose[4] * X + pose[5] * Y + pose[6] * Z;
        const float xr2 = pose[8] * X + pose[9] * Y + pose[10] * Z;
        const Proj pr = project_point(q, pose, obj, view, xr0, xr1, xr2, true);
        const float *src = d_zlat + ((size_t)view * q.P + g) * C_LAT + lane * 8;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t off = __builtin_amdgcn_readfirstlane(pr.off[c]);  // identical in every lane
            if (off != cur[c]) {
                if (cur[c] != 0xffffffffu) {
                    float *dst = d_latent + cur[c] + lane * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if (acc[c][e] != 0.f) atomicAdd(dst + e, acc[c][e]);
                        acc[c][e] = 0.f;
                    }
                }
                cur[c] = off;
            }
            const float w = pr.w[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[c][e] += w * a[e];
                acc[c][4 + e] += w * b[e];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (cur[c] != 0xffffffffu) {
            float *dst = d_latent + cur[c] + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (acc[c][e] != 0.f) atomicAdd(dst + e, acc[c][e]);
        }
}

// Small grids (the 32x32 / 64x64 grids of sn64 / SRN, anything whose (image, 4-channel) slab fits the LDS): dozens of samples
// hit every texel, so global atomics serialise on a few thousand addresses.  Instead a workgroup owns an (image, CS-channel
// slice) slab of the gradient grid in LDS (CS = 16 / 8 / 4 by what fits), walks the points of that image's object, and adds
// the slab to HBM once.  When there are at least as many (image, slice) pairs as compute units ONE workgroup walks all
// points and the slab goes out with plain read-add-write; otherwise TWO workgroups share the pair and meet in HBM with
// atomics (config 5 at CS = 16: 4 images x 32 slices x 2) -- never more (scatter_form picks the slice width by the image
// count): two adds onto a zeroed element commute, so the result does not depend on their order.
//
// What bounded the round-2..5 form of this kernel (thread = run of 8 consecutive samples x 8 channels, register accumulator
// per corner flushed when ITS texel changes; 64-bit fixed-point slab) -- tools/ubench/lds_atomic.hip + timing twins,
// profiles/r06_scatter_notes.md:
//   * the NUMBER of LDS atomic instructions, not their lanes: a 64-bit LDS atomic costs ~7.5 clocks of the CU's LDS pipe per
//     wave instruction whether 64 lanes or one are active.  Merging the adds of a run removed 70 % of the lane adds and not
//     one instruction -- some lane of the wave changes texel at every step, so every flush site executes;
//   * VALU: 64 slices each projected every point (~100 instructions), and every add converted fp32 -> int64 (17 instructions
//     in hipcc's expansion: 4350 of the kernel's 7700 static instructions);
//   * the L1: a lane read its 32 bytes of a 2 KiB gradient row as two dwordx4 -- 16 waves x 64 lines in flight against
//     256 lines of cache, each line fetched twice for a quarter of its bytes.
// Here the merge is done BEFORE lanes are assigned:
//   * scatter_segments_kernel projects every (view, point) ONCE -- the clamped grid position (ix, iy) of project_point,
//     8 bytes per point -- and cuts every ray into SEGMENTS: consecutive samples that share the cell (floor ix, floor iy),
//     i.e. all four corners, at most SEG_B of them; per image a sorted list of segment starts (ballot + scan compaction,
//     SEG_NSUB workgroups per image);
//   * latent_scatter_owner_kernel: lane = (segment, 4 channels); the CS / 4 lanes of a segment read adjacent 16-byte pieces of a
//     row in ONE load instruction.  A lane sums w_c * grad over its segment in fp32 registers and issues its 16 LDS adds once,
//     every lane of the wave active; segment bounds are requested two segments ahead, samples one segment ahead.
// The slab is fp64 and takes ds_add_f64 (39 clocks per wave instruction like ds_add_u64, 4.0 lanes/clk/CU with all 64 lanes on
// random banks; ds_add_f32 is executed one lane at a time for the whole CU: 192 clocks): no common scale, so no max pass over the
// gradients, no clamp, one conversion per add.  An fp32 value is exact in fp64 and a texel sums ~1e2-1e3 of them: the slab
// holds the sum to 2^-53, and its fp32 rounding can differ between two orders of the adds only on near-ties (the repeat
// runs of tools/gpu_scatter_bench.py agree bit for bit; not guaranteed).
// Same-box A/B, config 5 (4 x 32x32; 32 768 / 49 152 points), us per call: round-5 kernel 90 / 119, this one 48 / 59 (of which
// 5 are the segment pass); 4 x 64x64 grid: 400 -> 143-170.  What is left is the 16 ds_add_f64 per segment lane (~50 M lane adds
// per pass at 4 per clock and CU: ~25 us) and the chain list -> rows behind them.
constexpr int SLAB_MAX_BYTES = 160 * 1024;  // whole LDS
constexpr int OWNER_NT = 1024;
constexpr int SEG_NT = 1024;
constexpr int SEG_B = 4;  // longest segment = samples per trip of the owner kernel (divides 64: a wave boundary is a cut)

constexpr int SEG_NSUB = 16;  // sub-ranges of an object's samples, one workgroup of scatter_segments_kernel each

#pragma clang fp contract(off)
// Workgroup (image = obj * NS + view, sub-range j of the object's samples: sub_len consecutive samples, a multiple of 64):
// coords[view * P + point] = (ix, iy); segs[(img * SEG_NSUB + j) * sub_len + k], k < nseg[img * SEG_NSUB + j] = first sample
// (index inside the object, ascending) of the k-th segment that starts in the sub-range.  A segment ends where the next one
// starts, or with its sub-range.
__global__ void __launch_bounds__(SEG_NT)
scatter_segments_kernel(const EvalParams q, float2 *__restrict__ coords, int *__restrict__ segs, int *__restrict__ nseg,
                        const int sub_len) {
    __shared__ int wave_total[SEG_NT / 64];
    __shared__ int carry;  // segments written by the previous rounds
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int img = blockIdx.x / SEG_NSUB, sub = blockIdx.x % SEG_NSUB, obj = img / q.NS, view = img % q.NS;
    const int pts = q.per_obj * q.K;  // < 2^31 (P is)
    const int s_begin = sub * sub_len, s_end = s_begin + sub_len < pts ? s_begin + sub_len : pts;
    const size_t row0 = (size_t)view * q.P + (size_t)obj * pts;
    int *list = segs + (size_t)blockIdx.x * sub_len;
    const float *pose = q.poses + (size_t)img * 12;
    const float *fo = q.focal + (q.n_focal > 1 ? obj * 2 : 0);
    const float *cc = q.c + (q.n_c > 1 ? obj * 2 : 0);
    const float Wl = (float)q.Wl, Hl = (float)q.Hl;
    const float lsx = Wl / (Wl - 1.f) * 2.f, lsy = Hl / (Hl - 1.f) * 2.f;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = s_begin; base < s_end; base += SEG_NT) {
        const int s = base + t;
        int cell = -1;
        if (s < s_end) {
            const int g = obj * pts + s;
            const int r = g / q.K;
            const float *ray = q.rays + (size_t)r * 8;
            const float zz = q.z[g];
            const float X = ray[0] + zz * ray[3], Y = ray[1] + zz * ray[4], Z = ray[2] + zz * ray[5];
            // project_point's op order (pnr_device.h)
            const float xr0 = pose[0] * X + pose[1] * Y + pose[2] * Z;
            const float xr1 = pose[4] * X + pose[5] * Y + pose[6] * Z;
            const float xr2 = pose[8] * X + pose[9] * Y + pose[10] * Z;
            const float xc0 = xr0 + pose[3], xc1 = xr1 + pose[7], xc2 = xr2 + pose[11];
            float u = -xc0 / xc2; u = u * fo[0]; u = u + cc[0];
            float v = -xc1 / xc2; v = v * fo[1]; v = v + cc[1];
            const float gx = u * (lsx / q.img_w) - 1.f, gy = v * (lsy / q.img_h) - 1.f;
            float ix = ((gx + 1.f) / 2.f) * (Wl - 1.f), iy = ((gy + 1.f) / 2.f) * (Hl - 1.f);
            ix = fminf(Wl - 1.f, fmaxf(ix, 0.f));
            iy = fminf(Hl - 1.f, fmaxf(iy, 0.f));
            if (!(ix == ix)) ix = 0.f;  // NaN (point on the camera plane): as project_point
            if (!(iy == iy)) iy = 0.f;
            coords[row0 + s] = make_float2(ix, iy);
            cell = (int)floorf(iy) * q.Wl + (int)floorf(ix);
        }
        // segment start: first sample of a ray, a cell that differs from the previous sample's, or every SEG_B-th sample (a
        // segment is ONE trip of the owner kernel's loads: a ray that leaves the image clamps to one border cell for dozens of
        // samples, and the wave that held such a lane waited for 16 dependent trips -- 100 k of the kernel's 130 k cycles)
        const int prev = __shfl_up(cell, 1, 64);
        const bool head = s < s_end && (s % SEG_B == 0 || s % q.K == 0 || cell != prev);
        const unsigned long long m = __ballot(head);
        const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (lane == 0) wave_total[wv] = __popcll(m);
        __syncthreads();
        int off = carry, total = 0;
#pragma unroll
        for (int w = 0; w < SEG_NT / 64; ++w) {
            const int c = wave_total[w];
            if (w < wv) off += c;
            total += c;
        }
        if (head) list[off + before] = s;
        __syncthreads();
        if (t == 0) carry += total;
    }
    __syncthreads();
    if (t == 0) nseg[blockIdx.x] = carry;
}

template <int CS>
__global__ void __launch_bounds__(OWNER_NT)
latent_scatter_owner_kernel(const EvalParams q, const float *__restrict__ d_zlat, const float2 *__restrict__ coords,
                            const int *__restrict__ segs, const int *__restrict__ nseg, const int sub_len,
                            float *__restrict__ d_latent, const int psplit, const int row) {
    extern __shared__ double dslab[];  // [Hl*Wl][row]
    constexpr int LPS = CS / 4;        // lanes per segment (4 channels = one 16-byte load per sample and lane)
    constexpr int GRP = 32 / CS;       // slices that share a 128-byte line of a d_zlat / d_latent row
    const int t = threadIdx.x;
    const int texels = q.Hl * q.Wl, nslices = C_LAT / CS;
    // XCD-aware placement (as in dw_kernel): the slices that share 128-byte lines form a group that lands on ONE XCD
    // (consecutive per-XCD slots), so a line is fetched into one L2 once
    const int lid = blockIdx.x, ngroups = gridDim.x / GRP, full = (ngroups >> 3) * (8 * GRP);
    int grp, sub;
    if (lid < full) { const int k = lid >> 3; grp = (k / GRP) * 8 + (lid & 7); sub = k % GRP; }
    else { const int rem = lid - full; grp = full / GRP + rem / GRP; sub = rem % GRP; }
    const int cs = (grp % (nslices / GRP)) * GRP + sub;
    const int pslice = (grp / (nslices / GRP)) % psplit;
    const int img = grp / ((nslices / GRP) * psplit);  // img = obj * NS + view
    const int obj = img / q.NS, view = img % q.NS;
    for (int i = t; i < texels * row; i += OWNER_NT) dslab[i] = 0.0;
    const long long pts = (long long)q.per_obj * q.K;  // points of this object
    const size_t row0 = (size_t)view * q.P + (size_t)obj * pts;  // first row of this (view, object) in d_zlat / coords
    const float *grad = d_zlat + row0 * C_LAT + cs * CS;
    const float2 *xy = coords + row0;
    // this image's segment lists: SEG_NSUB sub-ranges, read as one list through the running sums of their counts
    const int *list = segs + (size_t)img * SEG_NSUB * sub_len;
    __shared__ int pre[SEG_NSUB + 1];
    if (t == 0) {
        int run = 0;
        for (int j = 0; j < SEG_NSUB; ++j) { pre[j] = run; run += nseg[img * SEG_NSUB + j]; }
        pre[SEG_NSUB] = run;
    }
    __syncthreads();
    const int n = pre[SEG_NSUB];
    const int per = (n + psplit - 1) / psplit;
    const int i_begin = pslice * per, i_end = i_begin + per < n ? i_begin + per : n;
    const int Wl = q.Wl, Hl = q.Hl;
    __syncthreads();  // slab zeroed
    // lane = (segment, 4-channel part of the slice): the LPS lanes of a segment read LPS * 16 contiguous bytes of a gradient row
    // with ONE load instruction (a lane that read its 32 bytes as two dwordx4 fetched the line twice: 16 waves x 64 lines in
    // flight against the 256 lines of the L1)
    const int part = t % LPS;
    // segment i -> [s0, s1): sub-range j with pre[j] <= i < pre[j + 1] (binary search over the SEG_NSUB = 16 running sums in LDS)
    auto seg_bounds = [&](int i, int &s0, int &s1) {
        int j = 0;
#pragma unroll
        for (int step = SEG_NSUB / 2; step > 0; step >>= 1) j += i >= pre[j + step] ? step : 0;
        const int k = i - pre[j];
        const int *lj = list + (size_t)j * sub_len;
        const int sub_end = (j + 1) * sub_len < (int)pts ? (j + 1) * sub_len : (int)pts;
        s0 = lj[k];
        s1 = i + 1 < pre[j + 1] ? lj[k + 1] : sub_end;
    };
    // a segment is at most SEG_B samples: all its loads are issued before the first use
    auto load_batch = [&](int sb, int s1, float2 (&pos)[SEG_B], f32x4 (&v)[SEG_B]) {
#pragma unroll
        for (int b = 0; b < SEG_B; ++b) {
            pos[b] = make_float2(0.f, 0.f);
            v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (sb + b < s1) {
                pos[b] = xy[sb + b];
                v[b] = reinterpret_cast<const f32x4 *>(grad + (size_t)(sb + b) * C_LAT)[part];
            }
        }
    };
    // Two-deep software pipeline over this lane's segments: a segment's bounds come out of the list (one round trip), its
    // samples out of the gradient rows (a second one that depends on the first) -- taken one after the other, ~8 segments
    // per lane were a chain of 16 loaded round trips (~2 us each at this access pattern).  The bounds are requested two
    // segments ahead and the first batch one segment ahead.
    constexpr int STRIDE = OWNER_NT / LPS;
    int i = i_begin + t / LPS;
    int s0 = 0, s1 = 0, s0n = 0, s1n = 0;
    float2 pos[SEG_B];
    f32x4 v[SEG_B];
    if (i < i_end) { seg_bounds(i, s0, s1); load_batch(s0, s1, pos, v); }
    if (i + STRIDE < i_end) seg_bounds(i + STRIDE, s0n, s1n);
    for (; i < i_end; i += STRIDE) {
        int s0nn = 0, s1nn = 0;
        if (i + 2 * STRIDE < i_end) seg_bounds(i + 2 * STRIDE, s0nn, s1nn);
        float2 posn[SEG_B];
        f32x4 vn[SEG_B];
        if (i + STRIDE < i_end) load_batch(s0n, s1n, posn, vn);
        const float ix0 = floorf(pos[0].x), iy0 = floorf(pos[0].y);  // the segment's cell
        const float ix1 = ix0 + 1.f, iy1 = iy0 + 1.f;
        const int x0 = (int)ix0, y0 = (int)iy0;
        const bool x_in = x0 + 1 <= Wl - 1, y_in = y0 + 1 <= Hl - 1;  // out-of-range corner has weight 0 (project_point)
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto accumulate = [&](const float2 (&ps)[SEG_B], const f32x4 (&vs)[SEG_B]) {
#pragma unroll
            for (int b = 0; b < SEG_B; ++b) {
                // corner weights of project_point from (ix, iy); slots beyond the segment carry zero gradients
                const float2 p = ps[b];
                float w[4] = {(ix1 - p.x) * (iy1 - p.y), (p.x - ix0) * (iy1 - p.y), (ix1 - p.x) * (p.y - iy0), (p.x - ix0) * (p.y - iy0)};
                if (!x_in) { w[1] = 0.f; w[3] = 0.f; }
                if (!y_in) { w[2] = 0.f; w[3] = 0.f; }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[c][e] += w[c] * vs[b][e];
            }
        };
        accumulate(pos, v);  // s1 - s0 <= SEG_B: scatter_segments_kernel cuts there
        const int x1 = min(x0 + 1, Wl - 1), y1 = min(y0 + 1, Hl - 1);
        const int tex[4] = {y0 * Wl + x0, y0 * Wl + x1, y1 * Wl + x0, y1 * Wl + x1};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double *dst = dslab + tex[c] * row + part * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dst + e, (double)acc[c][e]);  // ds_add_f64, no return
        }
        s0 = s0n; s1 = s1n; s0n = s0nn; s1n = s1nn;
#pragma unroll
        for (int b = 0; b < SEG_B; ++b) { pos[b] = posn[b]; v[b] = vn[b]; }
    }
    __syncthreads();
    float *out = d_latent + (size_t)img * texels * C_LAT + cs * CS;
    for (int i = t; i < texels * CS; i += OWNER_NT) {
        const double sv = dslab[(i / CS) * row + (i % CS)];
        if (sv != 0.0) {
            float *dst = out + (size_t)(i / CS) * C_LAT + (i % CS);
            if (psplit == 1) *dst += (float)sv;  // this workgroup is the only writer of the (image, slice) in this launch
            else atomicAdd(dst, (float)sv);
        }
    }
}

// ---- LARGE grids (DTU: 150 x 200 texels per image; anything whose (image, 4-channel) slab does not fit the LDS).  Rounds 2-5 sent
// them through global fp32 atomics (latent_scatter_kernel above): 25 M atomics per call of a 1-object x 3-view DTU training step,
// 965 us -- the largest kernel of that step -- and order-dependent.  Round 6: the slab form with the image cut into TILES of
// 32 x 32 texels.  A workgroup owns (image, tile, 16-channel slice): its slab is the tile (139 KiB of fp64), it takes the ray
// segments that touch the tile -- a segment's four corners can straddle up to four tiles, so it is listed in each -- adds only the
// corners that lie INSIDE the tile, and writes the tile back with plain read-add-write: every texel has one owner.
//   scatter_segments_kernel  (as above)           coords, segment starts per image
//   tile_bin_kernel<COUNT>                         per (image, tile): how many segments touch it
//   tile_scan_kernel                               running sums -> list offsets
//   tile_bin_kernel<FILL>                          segment (start | length - 1 << 29) into the lists of the tiles it touches (one atomic per entry:
//                                                  the order inside a list is not fixed -- the fp64 slab sum does not depend on it beyond 2^-53)
//   latent_scatter_tiled_kernel                    lane = (list entry, 4 channels): one trip of <= SEG_B samples, 16 predicated ds_add_f64
constexpr int TILE_W = 32, TILE_TEXELS = TILE_W * TILE_W, TILE_CS = 16, TILE_ROW = TILE_CS + 1;
constexpr int TILE_LDS = TILE_TEXELS * TILE_ROW * 8;  // 139,264 B

struct TileGeom { int tx, ty, ntiles; };  // tiles per image row / column, per image
__device__ __forceinline__ void tiles_of_cell(int x0, int y0, int Wl, int Hl, int tx, int (&tiles)[4], int &n) {
    const int x1 = min(x0 + 1, Wl - 1), y1 = min(y0 + 1, Hl - 1);
    const int ax = x0 / TILE_W, bx = x1 / TILE_W, ay = y0 / TILE_W, by = y1 / TILE_W;
    n = 0;
    tiles[n++] = ay * tx + ax;
    if (bx != ax) tiles[n++] = ay * tx + bx;
    if (by != ay) {
        tiles[n++] = by * tx + ax;
        if (bx != ax) tiles[n++] = by * tx + bx;
    }
}

// Workgroup = 256 slots of the segment list of one (image, sub-range): every touched tile is first counted in an LDS histogram
// (ranks from the returning LDS atomic), then ONE global atomic per (workgroup, tile) -- the counters are a hundred addresses, and
// one global atomic per list entry serialised on them (2 x 66 us for the 36 k segments of a DTU step; 2 x ~5 us this way).
template <bool FILL>
__global__ void __launch_bounds__(256)
tile_bin_kernel(const EvalParams q, const float2 *__restrict__ coords, const int *__restrict__ segs, const int *__restrict__ nseg,
                const int sub_len, const TileGeom tg, int *__restrict__ tile_cnt, const int *__restrict__ tile_off,
                int *__restrict__ cursor, unsigned *__restrict__ entries) {
    extern __shared__ int hist[];  // [ntiles] counts, [ntiles] bases (FILL)
    const int t = threadIdx.x, k = blockIdx.x * 256 + t, lj = blockIdx.y;
    const int n_list = nseg[lj];
    if (blockIdx.x * 256 >= n_list) return;  // uniform
    for (int i = t; i < tg.ntiles; i += 256) hist[i] = 0;
    __syncthreads();
    const int img = lj / SEG_NSUB, j = lj % SEG_NSUB, obj = img / q.NS, view = img % q.NS;
    const int pts = q.per_obj * q.K;
    int tiles[4], rank[4], n = 0;
    unsigned entry = 0;
    if (k < n_list) {
        const int *list = segs + (size_t)lj * sub_len;
        const int s0 = list[k];
        const int sub_end = (j + 1) * sub_len < pts ? (j + 1) * sub_len : pts;
        const int s1 = k + 1 < n_list ? list[k + 1] : sub_end;
        const float2 p0 = coords[(size_t)view * q.P + (size_t)obj * pts + s0];
        tiles_of_cell((int)floorf(p0.x), (int)floorf(p0.y), q.Wl, q.Hl, tg.tx, tiles, n);
        entry = (unsigned)s0 | ((unsigned)(s1 - s0 - 1) << 29);
        for (int c = 0; c < n; ++c) rank[c] = atomicAdd(hist + tiles[c], 1);
    }
    __syncthreads();
    for (int i = t; i < tg.ntiles; i += 256) {
        const int c = hist[i];
        if (c) {
            if (!FILL) atomicAdd(tile_cnt + img * tg.ntiles + i, c);
            else hist[tg.ntiles + i] = atomicAdd(cursor + img * tg.ntiles + i, c);
        }
    }
    if (FILL) {
        __syncthreads();
        for (int c = 0; c < n; ++c) entries[tile_off[img * tg.ntiles + tiles[c]] + hist[tg.ntiles + tiles[c]] + rank[c]] = entry;
    }
}

// tile_off[i] = sum of tile_cnt[0 .. i) over all (image, tile) pairs, one workgroup; clears the FILL pass's cursors
__global__ void __launch_bounds__(1024) tile_scan_kernel(const int *__restrict__ tile_cnt, int n, int *__restrict__ tile_off, int *__restrict__ cursor) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int v = base + t < n ? tile_cnt[base + t] : 0;
        part[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan (n is a few hundred: one round)
            const int add = t >= o ? part[t - o] : 0;
            __syncthreads();
            part[t] += add;
            __syncthreads();
        }
        if (base + t < n) { tile_off[base + t] = carry + part[t] - v; cursor[base + t] = 0; }
        __syncthreads();
        if (t == 1023) carry += part[1023];
        __syncthreads();
    }
    if (t == 0) tile_off[n] = carry;
}

#pragma clang fp contract(off)
__global__ void __launch_bounds__(OWNER_NT)
latent_scatter_tiled_kernel(const EvalParams q, const float *__restrict__ d_zlat, const float2 *__restrict__ coords,
                            const unsigned *__restrict__ entries, const int *__restrict__ tile_off, const TileGeom tg,
                            float *__restrict__ d_latent) {
    extern __shared__ double dslab[];  // [32 x 32 texels][TILE_ROW]
    constexpr int LPS = TILE_CS / 4, NSL = C_LAT / TILE_CS, GRP = 32 / TILE_CS;
    const int t = threadIdx.x;
    // XCD-aware placement as in the slab kernel: the slices that share 128-byte lines get consecutive slots of ONE XCD
    const int lid = blockIdx.x, ngroups = gridDim.x / GRP, full = (ngroups >> 3) * (8 * GRP);
    int grp, sub;
    if (lid < full) { const int k = lid >> 3; grp = (k / GRP) * 8 + (lid & 7); sub = k % GRP; }
    else { const int rem = lid - full; grp = full / GRP + rem / GRP; sub = rem % GRP; }
    const int cs = (grp % (NSL / GRP)) * GRP + sub;
    const int it = grp / (NSL / GRP);  // (image, tile)
    const int img = it / tg.ntiles, tile = it % tg.ntiles;
    const int obj = img / q.NS, view = img % q.NS;
    const int tx0 = (tile % tg.tx) * TILE_W, ty0 = (tile / tg.tx) * TILE_W;
    const int e_begin = tile_off[it], e_end = tile_off[it + 1];
    if (e_begin == e_end) return;  // no ray crosses this tile: nothing to add (uniform over the workgroup)
    for (int i = t; i < TILE_TEXELS * TILE_ROW; i += OWNER_NT) dslab[i] = 0.0;
    const long long pts = (long long)q.per_obj * q.K;
    const size_t row0 = (size_t)view * q.P + (size_t)obj * pts;
    const float *grad = d_zlat + row0 * C_LAT + cs * TILE_CS;
    const float2 *xy = coords + row0;
    const int Wl = q.Wl, Hl = q.Hl;
    const int part = t % LPS;
    __syncthreads();
    for (int i = e_begin + t / LPS; i < e_end; i += OWNER_NT / LPS) {
        const unsigned en = entries[i];
        const int s0 = (int)(en & 0x1fffffffu), len = (int)(en >> 29) + 1;
        float2 pos[SEG_B];
        f32x4 v[SEG_B];
#pragma unroll
        for (int b = 0; b < SEG_B; ++b) {
            pos[b] = make_float2(0.f, 0.f);
            v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (b < len) {
                pos[b] = xy[s0 + b];
                v[b] = reinterpret_cast<const f32x4 *>(grad + (size_t)(s0 + b) * C_LAT)[part];
            }
        }
        const float ix0 = floorf(pos[0].x), iy0 = floorf(pos[0].y);  // the segment's cell
        const float ix1 = ix0 + 1.f, iy1 = iy0 + 1.f;
        const int x0 = (int)ix0, y0 = (int)iy0;
        const bool x_in = x0 + 1 <= Wl - 1, y_in = y0 + 1 <= Hl - 1;  // out-of-range corner has weight 0 (project_point)
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < SEG_B; ++b) {
            const float2 p = pos[b];  // slots beyond the segment carry zero gradients
            float w[4] = {(ix1 - p.x) * (iy1 - p.y), (p.x - ix0) * (iy1 - p.y), (ix1 - p.x) * (p.y - iy0), (p.x - ix0) * (p.y - iy0)};
            if (!x_in) { w[1] = 0.f; w[3] = 0.f; }
            if (!y_in) { w[2] = 0.f; w[3] = 0.f; }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[c][e] += w[c] * v[b][e];
        }
        const int x1 = min(x0 + 1, Wl - 1), y1 = min(y0 + 1, Hl - 1);
        const int cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int lx = cx[c] - tx0, ly = cy[c] - ty0;
            if ((unsigned)lx < (unsigned)TILE_W && (unsigned)ly < (unsigned)TILE_W) {  // this tile owns the corner's texel
                double *dst = dslab + (ly * TILE_W + lx) * TILE_ROW + part * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, (double)acc[c][e]);  // ds_add_f64, no return
            }
        }
    }
    __syncthreads();
    // write-back: read-add-write of the touched elements, this workgroup being their only writer.  Done element by element -- LDS read,
    // branch, global load, add, store -- every touched element of a thread was a round trip of its own (a quarter of a tile's texels are
    // touched: ~4 dependent trips per thread); here all 16 loads of a thread are issued first (an untouched element reads a valid
    // dummy), then the touched ones are added and stored: one trip
    float *out = d_latent + (size_t)img * Hl * Wl * C_LAT + cs * TILE_CS;
    constexpr int WB = TILE_TEXELS * TILE_CS / OWNER_NT;  // 16 elements per thread
    static_assert(TILE_TEXELS * TILE_CS % OWNER_NT == 0, "write-back tiling");
    float add[WB], old[WB];
    float *dst[WB];
#pragma unroll
    for (int j = 0; j < WB; ++j) {
        const int i = t + j * OWNER_NT;
        const int tex = i / TILE_CS, ch = i % TILE_CS;
        const int x = tx0 + tex % TILE_W, y = ty0 + tex / TILE_W;
        const double sv = dslab[tex * TILE_ROW + ch];
        const bool touched = sv != 0.0 && x < Wl && y < Hl;
        add[j] = (float)sv;
        dst[j] = touched ? out + (size_t)(y * Wl + x) * C_LAT + ch : nullptr;
        old[j] = *(touched ? dst[j] : out);  // (`out` itself is always a valid address of the image)
    }
#pragma unroll
    for (int j = 0; j < WB; ++j)
        if (dst[j]) *dst[j] = old[j] + add[j];
}
#pragma clang fp contract(fast)

// dL/dz through the network inputs; one wavefront per (view, point).
// ranks == nullptr: every point, the result is accumulated into d_z[point].
// ranks (R, Kfd): only the depth samples (the ones whose position carries gradient, nerf.py:157-160,292): wave =
// (view, ray, j), point = ray * K + ranks[ray][j]; the total dL/dz of that sample -- this network term plus the compositing
// term dz_comp -- goes through the clamp z = max(min(depth + n * std, far), near) into contrib[view][ray][j].
struct DepthSamples {
    const int *ranks;        // (R, Kfd) position of each depth sample in the sorted z, or null
    const float *n4;         // (R, Kfd) the normal draws
    const float *depth_c;    // (R) coarse depth
    const float *dz_comp;    // (R, K) compositing dL/dz (nullable)
    float *contrib;          // (NS, R, Kfd) out: what sample j of ray r sends to d_depth[r] through view v (summed by the caller
                             // in a fixed order: the result stays bit-reproducible)
    float depth_std;
    int Kfd;
};

__global__ void __launch_bounds__(CW * 64)
position_bwd_kernel(const EvalParams q, const float *__restrict__ d_in42, const float *__restrict__ d_zlat,
                    float *__restrict__ d_z, const DepthSamples ds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long widx = (long long)blockIdx.x * CW + wv;
    int view, g, jd = 0;
    if (ds.ranks) {  // (view, ray, j)
        const long long per_view = (long long)(q.P / q.K) * ds.Kfd;
        if (widx >= per_view * q.NS) return;
        view = (int)(widx / per_view);
        const long long rj = widx - (long long)view * per_view;
        const int ray_i = (int)(rj / ds.Kfd);
        jd = (int)(rj - (long long)ray_i * ds.Kfd);
        g = ray_i * q.K + ds.ranks[(size_t)ray_i * ds.Kfd + jd];
    } else {         // view * P + point
        if (widx >= q.P * q.NS) return;
        view = (int)(widx / q.P);
        g = (int)(widx % q.P);
    }
    const long long idx = (long long)view * q.P + g;
    const int r = g / q.K;
    const float *ray = q.rays + (size_t)r * 8;
    const float zz = q.z[g];
    const float X = ray[0] + zz * ray[3], Y = ray[1] + zz * ray[4], Z = ray[2] + zz * ray[5];
    const int obj = r / q.per_obj;
    const float *pose = q.poses + (size_t)(obj * q.NS + view) * 12;
    const float xr0 = pose[0] * X + pose[1] * Y + pose[2] * Z;
    const float xr1 = pose[4] * X + pose[5] * Y + pose[6] * Z;
    const float xr2 = pose[8] * X + pose[9] * Y + pose[10] * Z;
    const float xc0 = xr0 + pose[3], xc1 = xr1 + pose[7], xc2 = xr2 + pose[11];
    const float *fo = q.focal + (q.n_focal > 1 ? obj * 2 : 0);
    const float *cc = q.c + (q.n_c > 1 ? obj * 2 : 0);
    const float u = -xc0 / xc2 * fo[0] + cc[0], v = -xc1 / xc2 * fo[1] + cc[1];
    const float Wl = (float)q.Wl, Hl = (float)q.Hl;
    const float sx = Wl / (Wl - 1.f) * 2.f / q.img_w, sy = Hl / (Hl - 1.f) * 2.f / q.img_h;
    float ix = ((u * sx - 1.f + 1.f) / 2.f) * (Wl - 1.f), iy = ((v * sy - 1.f + 1.f) / 2.f) * (Hl - 1.f);
    // grid_sample border padding: clip_coordinates_set_grad -> gradient 0 outside (0, size-1)
    const bool gx_on = ix > 0.f && ix < Wl - 1.f, gy_on = iy > 0.f && iy < Hl - 1.f;
    ix = fminf(Wl - 1.f, fmaxf(ix, 0.f));
    iy = fminf(Hl - 1.f, fmaxf(iy, 0.f));
    float six = 0.f, siy = 0.f;
    if ((gx_on || gy_on) && ix == ix && iy == iy) {
        const float ix0 = floorf(ix), iy0 = floorf(iy);
        const int x0 = (int)ix0, y0 = (int)iy0;
        const int x1 = min(x0 + 1, q.Wl - 1), y1 = min(y0 + 1, q.Hl - 1);
        const float ax = ix - ix0, ay = iy - iy0;  // fractional parts
        const size_t rowbase = (size_t)(obj * q.NS + view) * (size_t)(q.Hl * q.Wl);
        const float *nw = q.latent + (rowbase + (size_t)y0 * q.Wl + x0) * C_LAT + lane * 8;
        const float *ne = q.latent + (rowbase + (size_t)y0 * q.Wl + x1) * C_LAT + lane * 8;
        const float *sw = q.latent + (rowbase + (size_t)y1 * q.Wl + x0) * C_LAT + lane * 8;
        const float *se = q.latent + (rowbase + (size_t)y1 * q.Wl + x1) * C_LAT + lane * 8;
        const float *dz = d_zlat + (size_t)idx * C_LAT + lane * 8;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {  // 16-byte loads: 10 in flight per lane
            const f32x4 a = *reinterpret_cast<const f32x4 *>(nw + 4 * hh), b = *reinterpret_cast<const f32x4 *>(ne + 4 * hh);
            const f32x4 c = *reinterpret_cast<const f32x4 *>(sw + 4 * hh), d = *reinterpret_cast<const f32x4 *>(se + 4 * hh);
            const f32x4 gq = *reinterpret_cast<const f32x4 *>(dz + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                six += gq[e] * ((1.f - ay) * (b[e] - a[e]) + ay * (d[e] - c[e]));   // d zlat / d ix
                siy += gq[e] * ((1.f - ax) * (c[e] - a[e]) + ax * (d[e] - b[e]));   // d zlat / d iy
            }
        }
    }
    // positional code: [x, sin(f_k x), sin(f_k x + pi/2)] , f_k = 1.5 * 2^k  (code.py:37-41).  Lane l < 18 differentiates
    // band k = l / 3 of coordinate c = l % 3; lanes 18..20 carry the identity part; summed per coordinate below
    float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
    {
        const float *gi = d_in42 + (size_t)idx * D_IN;
        const float HALF_PI = 1.57079637050628662109375f;
        if (lane < 21) {
            const int k = lane / 3, c = lane - 3 * k;
            const float xc = c == 0 ? xr0 : (c == 1 ? xr1 : xr2);
            float term;
            if (k < 6) {
                const float f = 1.5f * (float)(1 << k), a = xc * f;
                term = f * (cosf(a) * gi[3 + 6 * k + c] + cosf(__builtin_fmaf(xc, f, HALF_PI)) * gi[3 + 6 * k + 3 + c]);
            } else {
                term = gi[c];
            }
            gc0 = c == 0 ? term : 0.f; gc1 = c == 1 ? term : 0.f; gc2 = c == 2 ? term : 0.f;
        }
    }
    gc0 = wsum(gc0); gc1 = wsum(gc1); gc2 = wsum(gc2);
    six = wsum(six);
    siy = wsum(siy);
    if (lane == 0) {
        const float du = gx_on ? six * (Wl - 1.f) * 0.5f * sx : 0.f;
        const float dv = gy_on ? siy * (Hl - 1.f) * 0.5f * sy : 0.f;
        // u = -xc0/xc2 fx + cx ; v = -xc1/xc2 fy + cy   (fy already negated in `focal`)
        float g0 = -fo[0] / xc2 * du;
        float g1 = -fo[1] / xc2 * dv;
        float g2 = (xc0 * fo[0] * du + xc1 * fo[1] * dv) / (xc2 * xc2);
        g0 += gc0; g1 += gc1; g2 += gc2;
        // x_world gradient = R^T g ; dz = ray_dir . that
        const float wx = pose[0] * g0 + pose[4] * g1 + pose[8] * g2;
        const float wy = pose[1] * g0 + pose[5] * g1 + pose[9] * g2;
        const float wz = pose[2] * g0 + pose[6] * g1 + pose[10] * g2;
        float val = ray[3] * wx + ray[4] * wy + ray[5] * wz;
        if (!ds.ranks) {
            if (val == val) atomicAdd(d_z + g, val);
        } else {
            if (view == 0 && ds.dz_comp) val += ds.dz_comp[g];
            const float zraw = ds.depth_c[r] + ds.n4[(size_t)r * ds.Kfd + jd] * ds.depth_std;
            const bool live = zraw < ray[7] && zraw > ray[6];  // inside the clamp: gradient passes (nerf.py:157-160)
            ds.contrib[widx] = (live && val == val) ? val : 0.f;
        }
    }
}
#pragma clang fp contract(fast)

}  // namespace pnr

using namespace pnr;

extern "C" int pnr_storage_perm(int32_t *perm512) {
    if (!perm512) return pnr_fail(PNR_E_INVALID, "pnr_storage_perm: null argument");
    for (int e = 0; e < D_HID; ++e) perm512[e] = feat_of(e / 32, (e % 32) / 16, e % 16);
    return PNR_OK;
}

extern "C" size_t pnr_packed_mlp_bwd_bytes(void) { return BPACKED_BYTES; }

extern "C" int pnr_pack_mlp_bwd(const PnrMlpWeights *w, int precision, void *packed_bwd, void *stream) {
    if (!w || !packed_bwd) return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp_bwd: null argument");
    const size_t n = BWSTREAM_ELEMS_PER_WAVE / 8 * NW;  // one thread per 8 elements
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL(pack_weights_bwd_kernel<_Float16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *w, (_Float16 *)packed_bwd);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL(pack_weights_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *w, (__bf16 *)packed_bwd);
    else
        return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp_bwd: unknown precision");
    return pnr_check_launch("pnr_pack_mlp_bwd");
}

static int bwd_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// scales[0] = 2^(6 - ceil(log2 max|g|)) (1 if g == 0), scales[1] = 1 / scales[0]; NaN if g holds a non-finite value
__global__ void __launch_bounds__(1024) grad_scale_kernel(const float *__restrict__ g, long long n, float *__restrict__ scales) {
    __shared__ float red[1024];
    float m = 0.f;
    bool bad = false;
    // one block, but 8 independent 16-byte loads in flight per thread: a handful of memory round trips for the
    // (P,4) gradient instead of one per element (51 us -> a few us at config-5 sizes)
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (long long i0 = threadIdx.x; i0 < n4; i0 += 8 * 1024) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = i0 + (long long)u * 1024;
            v[u] = i < n4 ? g4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fabsf(v[u][e]);
                bad |= !(a <= 3.0e38f);
                m = fmaxf(m, a);
            }
    }
    for (long long i = n4 * 4 + threadIdx.x; i < n; i += 1024) {
        const float a = fabsf(g[i]);
        bad |= !(a <= 3.0e38f);
        m = fmaxf(m, a);
    }
    red[threadIdx.x] = bad ? __builtin_inff() : m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mx = red[0];
        float sc = 1.f;
        if (!(mx <= 3.0e38f)) sc = __builtin_nanf("");
        // exponent clamped to +-100: scale and 1/scale stay finite normal numbers for vanishing (denormal-range) or huge
        // gradients -- an unclamped 2^(6 - log2 mx) overflows to inf for mx < 2^-121 and poisons the step with 0 * inf
        else if (mx > 0.f) sc = exp2f(fminf(fmaxf(6.f - ceilf(log2f(mx)), -100.f), 100.f));
        scales[0] = sc;
        scales[1] = 1.f / sc;
    }
}

extern "C" size_t pnr_train_masks_bytes(long long P, int NS) {
    if (P <= 0 || NS <= 0) return 0;
    return (size_t)11 * (size_t)NS * (size_t)((P + MT - 1) / MT) * NTHREADS * sizeof(unsigned long long);
}

// Large gradients (the stand-alone linear operators scale a whole (rows, d_out) tensor: 33 M values take 1.5 ms in one workgroup):
// max |g| over many workgroups -- non-negative floats order like their bit patterns, a non-finite value enters as +inf -- into
// scales[0] (zeroed first), then one thread turns the maximum into [scale, 1 / scale] exactly as grad_scale_kernel does.
__global__ void __launch_bounds__(1024) grad_absmax_kernel(const float *__restrict__ g, long long n, unsigned int *__restrict__ out) {
    __shared__ float red[1024];
    float m = 0.f;
    bool bad = false;
    const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    const long long stride = (long long)gridDim.x * 8 * 1024;
    for (long long i0 = (long long)blockIdx.x * 8 * 1024 + threadIdx.x; i0 < n4; i0 += stride) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = i0 + (long long)u * 1024;
            v[u] = i < n4 ? g4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = fabsf(v[u][e]);
                bad |= !(a <= 3.0e38f);
                m = fmaxf(m, a);
            }
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n; i += 1024) {
            const float a = fabsf(g[i]);
            bad |= !(a <= 3.0e38f);
            m = fmaxf(m, a);
        }
    red[threadIdx.x] = bad ? __builtin_inff() : m;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(red[0]));
}
__global__ void grad_scale_from_max_kernel(float *__restrict__ scales) {
    const float mx = __uint_as_float(reinterpret_cast<const unsigned int *>(scales)[0]);
    float sc = 1.f;
    if (!(mx <= 3.0e38f)) sc = __builtin_nanf("");
    else if (mx > 0.f) sc = exp2f(fminf(fmaxf(6.f - ceilf(log2f(mx)), -100.f), 100.f));
    scales[0] = sc;
    scales[1] = 1.f / sc;
}

extern "C" int pnr_grad_scale(const float *g, long long n, float *scales, void *stream) {
    if (!g || !scales || n <= 0) return pnr_fail(PNR_E_INVALID, "pnr_grad_scale: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (n <= (1LL << 20)) {  // the renderer's (P, 4) output gradient: one workgroup, one launch
        hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1024), 0, st, g, n, scales);
        return pnr_check_launch("pnr_grad_scale");
    }
    const hipError_t e = hipMemsetAsync(scales, 0, 2 * sizeof(float), st);
    if (e != hipSuccess) return pnr_check_hip(e, "hipMemsetAsync(pnr_grad_scale)");
    const long long chunks = (n / 4 + 8 * 1024 - 1) / (8 * 1024);
    const unsigned blocks = (unsigned)(chunks < 1024 ? (chunks < 1 ? 1 : chunks) : 1024);
    hipLaunchKernelGGL(grad_absmax_kernel, dim3(blocks), dim3(1024), 0, st, g, n, reinterpret_cast<unsigned int *>(scales));
    hipLaunchKernelGGL(grad_scale_from_max_kernel, dim3(1), dim3(1), 0, st, scales);
    return pnr_check_launch("pnr_grad_scale");
}

extern "C" int pnr_mlp_backward(const void *packed_bwd, int precision, const PnrTrainDumps *fwd, const float *g_out,
                                float grad_scale, const float *grad_scale_dev, long long P, int NS,
                                const PnrBackwardDumps *out, void *stream) {
    if (!packed_bwd || !fwd || !g_out || !out || P <= 0 || NS <= 0 || (!grad_scale_dev && !(grad_scale > 0.f)))
        return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: bad argument");
    if (P > 0x7fffffc0LL || (P + MT - 1) / MT * NS * NTHREADS > 0xffffffffLL)
        return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: too many points");
    BwdParams q = {};
    q.wstream = (const char *)packed_bwd;
    q.g_out = g_out; q.scale = grad_scale; q.scale_dev = grad_scale_dev; q.P = P; q.NS = NS; q.ntiles = (int)((P + MT - 1) / MT);
    q.d_mask = (const unsigned long long *)fwd->d_mask; q.g_x0 = (char *)out->g_x0;
    q.d_zlat = out->d_zlat; q.d_in = out->d_in;
    // d_zlat is REQUIRED: the lin_z^T / lin_in^T GEMMs sit inside the per-view segment of the transposed weight stream, and
    // the prefetch ring only stays in step with that stream when they run (skipping them made every later view / tile
    // multiply by the wrong fragments)
    if (!q.d_zlat) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: PnrBackwardDumps.d_zlat is required (d_in alone is optional)");
    if (!q.d_mask) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: the forward dumps carry no relu bit masks (PnrTrainDumps.d_mask)");
    if (!q.g_x0) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: null dump");
    for (int b = 0; b < 5; ++b) {
        q.g_fc1[b] = (char *)out->g_fc1[b]; q.g_fc0[b] = (char *)out->g_fc0[b];
        if (!q.g_fc1[b] || !q.g_fc0[b]) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: null dump");
    }
    const bool mv = NS > 1;
    const int grid = q.ntiles < bwd_num_cus() ? q.ntiles : bwd_num_cus();
    if (mv) {
        q.mv_ws = mv_scratch((hipStream_t)stream, (size_t)bwd_num_cus() * 96 * D_HID * sizeof(float));
        if (!q.mv_ws) return pnr_fail(PNR_E_HIP, "pnr_mlp_backward: cannot allocate the multi-view scratch (48 MiB)");
    }
    void (*k)(const BwdParams);
    if (precision == PNR_PREC_F16) k = mv ? bwd_kernel<PNR_PREC_F16, true> : bwd_kernel<PNR_PREC_F16, false>;
    else if (precision == PNR_PREC_BF16) k = mv ? bwd_kernel<PNR_PREC_BF16, true> : bwd_kernel<PNR_PREC_BF16, false>;
    else return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward: unknown precision");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(bwd_kernel)");
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), LDS_TOTAL, (hipStream_t)stream, q);
    return pnr_check_launch("bwd_kernel");
}

// lin_out (4 x 512): dW[o][k] = sum_r g[r][o] x5[r][k], db[o] = sum_r g[r][o]; g fp32 (P,4), x5 16-bit (P,512)
// in storage order.  Row slices -> partials -> fixed-order reduce (written in feature order).
constexpr int LO_BLOCKS = 256;
template <typename T>
__global__ void __launch_bounds__(256)
lin_out_grad_kernel(const float *__restrict__ g, const T *__restrict__ x5, long long P, float *__restrict__ part) {
    const int t = threadIdx.x;
    const long long per = (P + LO_BLOCKS - 1) / LO_BLOCKS;
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < P ? r0 + per : P;
    float acc[4][2] = {}, bs[4] = {};
#pragma unroll 8  // 8 rows of loads in flight: the loop is latency-bound otherwise (66 us -> ~10 us at config-5 sizes)
    for (long long r = r0; r < r1; ++r) {
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + r * 4);
        const uint32_t xw = *reinterpret_cast<const uint32_t *>(x5 + r * D_HID + 2 * t);
        const float xa = (float)__builtin_bit_cast(T, (uint16_t)(xw & 0xffffu)), xb = (float)__builtin_bit_cast(T, (uint16_t)(xw >> 16));
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o][0] += gv[o] * xa; acc[o][1] += gv[o] * xb;
            bs[o] += gv[o];
        }
    }
    float *pz = part + (size_t)blockIdx.x * (4 * D_HID + 4);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        pz[o * D_HID + 2 * t] = acc[o][0];
        pz[o * D_HID + 2 * t + 1] = acc[o][1];
    }
    if (t == 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o) pz[4 * D_HID + o] = bs[o];
    }
}
__global__ void lin_out_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * D_HID + 4) return;
    float s = 0.f;
    for (int z = 0; z < LO_BLOCKS; ++z) s += part[(size_t)z * (4 * D_HID + 4) + idx];
    if (idx < 4 * D_HID) {
        const int o = idx / D_HID, e = idx % D_HID;
        dW[o * D_HID + feat_of(e >> 5, (e >> 4) & 1, e & 15)] = s;
    } else if (db) {
        db[idx - 4 * D_HID] = s;
    }
}

extern "C" size_t pnr_lin_out_grad_workspace_bytes(void) { return (size_t)LO_BLOCKS * (4 * D_HID + 4) * sizeof(float); }

extern "C" int pnr_lin_out_grad(const float *g_out, const void *x5, long long P, int precision, float *dW, float *db,
                                void *workspace, void *stream) {
    if (!g_out || !x5 || !dW || !workspace || P <= 0) return pnr_fail(PNR_E_INVALID, "pnr_lin_out_grad: bad argument");
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)workspace;
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL(lin_out_grad_kernel<_Float16>, dim3(LO_BLOCKS), dim3(256), 0, st, g_out, (const _Float16 *)x5, P, part);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL(lin_out_grad_kernel<__bf16>, dim3(LO_BLOCKS), dim3(256), 0, st, g_out, (const __bf16 *)x5, P, part);
    else
        return pnr_fail(PNR_E_INVALID, "pnr_lin_out_grad: unknown precision");
    hipLaunchKernelGGL(lin_out_reduce_kernel, dim3((4 * D_HID + 4 + 255) / 256), dim3(256), 0, st, part, dW, db);
    return pnr_check_launch("pnr_lin_out_grad");
}

constexpr int DW_MAX_SPLIT = 32;
static int dw_nsplit(int n_jobs, long long max_rows) {
    // 4 tiles x jobs x slices workgroups of 8 waves, one per CU at a time: pick the slice count whose workgroup total fills
    // whole rounds of the chip (14 jobs: 9 slices = 504 workgroups = 1.97 rounds; the former "at least 512" rule gave 10
    // slices = 560 = 2.19 rounds, i.e. a third round at 19 % occupancy); slices of at least 256 rows
    const int cus = bwd_num_cus();
    const int per = 4 * n_jobs;
    int best = 1;
    double best_eff = 0.0;
    for (int ns = 1; ns <= DW_MAX_SPLIT && (long long)per * ns <= 3LL * cus; ++ns) {
        const long long blocks = (long long)per * ns;
        const long long rounds = (blocks + cus - 1) / cus;
        const double eff = (double)blocks / (double)(rounds * cus);
        if (blocks >= cus && eff > best_eff + 1e-9) { best_eff = eff; best = ns; }
        else if (blocks < cus) best = ns;  // fewer workgroups than CUs: more slices is always better
    }
    if (const char *e = getenv("PNR_DW_NSPLIT")) {  // experiment hook: force the slice count
        const int v = atoi(e);
        if (v >= 1 && v <= DW_MAX_SPLIT) best = v;
    }
    const long long cap = (max_rows + 255) / 256;
    if (best > cap) best = (int)cap;
    return best < 1 ? 1 : best;
}

extern "C" size_t pnr_weight_grad_batched_workspace_bytes(int n_jobs, long long max_rows) {
    if (n_jobs < 1 || n_jobs > DW_MAX_JOBS || max_rows < 1) return 0;
    return (size_t)n_jobs * dw_nsplit(n_jobs, max_rows) * (D_HID * D_HID + D_HID) * sizeof(float);
}
extern "C" size_t pnr_weight_grad_workspace_bytes(void) { return (size_t)DW_MAX_SPLIT * (D_HID * D_HID + D_HID) * sizeof(float); }

extern "C" int pnr_weight_grad_batched(const PnrWeightGradJob *jobs, int n_jobs, int precision, float out_scale,
                                       const float *out_scale_dev, void *workspace, void *stream) {
    if (!jobs || !workspace || n_jobs < 1 || n_jobs > DW_MAX_JOBS)
        return pnr_fail(PNR_E_INVALID, "pnr_weight_grad_batched: 1..16 jobs and a workspace are required");
    DwJobs J = {};
    long long max_rows = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!jobs[j].dY || !jobs[j].X || !jobs[j].dW || jobs[j].rows <= 0)
            return pnr_fail(PNR_E_INVALID, "pnr_weight_grad_batched: bad job");
        J.dY[j] = jobs[j].dY; J.X[j] = jobs[j].X; J.dW[j] = jobs[j].dW; J.db[j] = jobs[j].db; J.rows[j] = jobs[j].rows;
        J.rows_st[j] = jobs[j].rows_storage_order ? 1 : 0; J.cols_st[j] = jobs[j].cols_storage_order ? 1 : 0;
        const int nx = jobs[j].x_cols ? jobs[j].x_cols : D_HID, ncw = jobs[j].dw_cols ? jobs[j].dw_cols : nx;
        if (nx < 8 || nx > D_HID || nx % 8 != 0 || ncw < 1 || ncw > nx || (nx != D_HID && jobs[j].cols_storage_order))
            return pnr_fail(PNR_E_INVALID, "pnr_weight_grad_batched: x_cols must be a multiple of 8 in [8,512], dw_cols <= x_cols");
        J.nx[j] = (short)nx; J.ncw[j] = (short)ncw;
        if (jobs[j].rows > max_rows) max_rows = jobs[j].rows;
    }
    J.nsplit = dw_nsplit(n_jobs, max_rows);
    float *part = (float *)workspace;
    float *bpart = part + (size_t)n_jobs * J.nsplit * D_HID * D_HID;
    dim3 grid(4u * (unsigned)J.nsplit * (unsigned)n_jobs);  // (tile, slice, job) decoded XCD-aware in the kernel
    hipStream_t st = (hipStream_t)stream;
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL(dw_kernel<PNR_PREC_F16>, grid, dim3(512), 0, st, J, part, bpart);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL(dw_kernel<PNR_PREC_BF16>, grid, dim3(512), 0, st, J, part, bpart);
    else if (precision == PNR_PREC_F16X3) {
        // split-operand form: dY / X are (head | tail) f16 row sets, the tail array behind the head array
        constexpr int lds = 2 * 4 * 32 * 576;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dw_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(dw_split_kernel)");
        // one wave per SIMD with 128 x 128 wave tiles (dw_split_wide_kernel) since round 6: -12 % per launch, same partial sums bit
        // for bit (the bias sums differ in summation order); PNR_DW_FORM=8wave selects the round-3..5 kernel (profiles/r06_dw_split_notes.md)
        static const bool wide = [] { const char *e = getenv("PNR_DW_FORM"); return !(e && !strcmp(e, "8wave")); }();
        if (wide) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(dw_split_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(dw_split_wide_kernel)");
            hipLaunchKernelGGL(dw_split_wide_kernel, grid, dim3(256), lds, st, J, part, bpart);
        } else
            hipLaunchKernelGGL(dw_split_kernel, grid, dim3(512), lds, st, J, part, bpart);
    } else
        return pnr_fail(PNR_E_INVALID, "pnr_weight_grad_batched: unknown precision");
    hipLaunchKernelGGL(dw_reduce_kernel, dim3(D_HID * D_HID / 256, (unsigned)n_jobs), dim3(256), 0, st, J, part, bpart, out_scale, out_scale_dev);
    return pnr_check_launch("pnr_weight_grad_batched");
}

extern "C" int pnr_weight_grad(const void *dY, const void *X, long long rows, int precision, float out_scale,
                               int rows_storage_order, int cols_storage_order, float *dW, float *db, void *workspace,
                               void *stream) {
    if (!dY || !X || !dW || !workspace || rows <= 0) return pnr_fail(PNR_E_INVALID, "pnr_weight_grad: bad argument");
    PnrWeightGradJob job = {dY, X, rows, rows_storage_order, cols_storage_order, dW, db, 0, 0};
    return pnr_weight_grad_batched(&job, 1, precision, out_scale, nullptr, workspace, stream);
}

extern "C" int pnr_composite_backward(const float *rays, const float *z, const float *rgbsigma, int R, int K, int white_bkgd,
                                      const float *d_rgb, const float *d_depth, const float *d_weights, float *d_rgbsigma,
                                      float *d_z, int pre_activation, void *stream) {
    if (R < 0 || K <= 0) return pnr_fail(PNR_E_INVALID, "pnr_composite_backward: bad sizes");
    if (R == 0) return PNR_OK;
    if (!rays || !z || !rgbsigma || !d_rgb || !d_rgbsigma) return pnr_fail(PNR_E_INVALID, "pnr_composite_backward: null argument");
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((R + CW - 1) / CW), dim3(CW * 64), 0, (hipStream_t)stream, rays, z,
                       (const float4 *)rgbsigma, R, K, white_bkgd, d_rgb, d_depth, d_weights, (float4 *)d_rgbsigma, d_z,
                       pre_activation);
    return pnr_check_launch("pnr_composite_backward");
}

extern "C" int pnr_position_backward(const PnrScene *s, const float *rays, const float *z, int R, int rays_per_obj, int K,
                                     const float *d_in42, const float *d_zlat, float *d_z, void *stream) {
    if (!s || !rays || !z || !d_in42 || !d_zlat || !d_z || R <= 0 || K <= 0 || rays_per_obj <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_position_backward: bad argument");
    if ((long long)rays_per_obj * s->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_position_backward: R != SB * rays_per_obj");
    EvalParams q = {};
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K;
    const long long n = q.P * q.NS;
    hipLaunchKernelGGL(position_bwd_kernel, dim3((unsigned)((n + CW - 1) / CW)), dim3(CW * 64), 0, (hipStream_t)stream, q,
                       d_in42, d_zlat, d_z, DepthSamples{});
    return pnr_check_launch("pnr_position_backward");
}

extern "C" int pnr_depth_sample_backward(const PnrScene *s, const float *rays, const float *z, int R, int rays_per_obj, int K,
                                         const int *ranks, const float *n4, int Kfd, const float *depth_c, float depth_std,
                                         const float *d_in42, const float *d_zlat, const float *dz_comp, float *contrib,
                                         void *stream) {
    if (!s || !rays || !z || !ranks || !n4 || !depth_c || !d_in42 || !d_zlat || !contrib || R <= 0 || K <= 0 || Kfd <= 0 ||
        Kfd > K || rays_per_obj <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_depth_sample_backward: bad argument");
    if ((long long)rays_per_obj * s->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_depth_sample_backward: R != SB * rays_per_obj");
    EvalParams q = {};
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K;
    DepthSamples ds = {ranks, n4, depth_c, dz_comp, contrib, depth_std, Kfd};
    const long long n = (long long)R * Kfd * q.NS;
    hipLaunchKernelGGL(position_bwd_kernel, dim3((unsigned)((n + CW - 1) / CW)), dim3(CW * 64), 0, (hipStream_t)stream, q,
                       d_in42, d_zlat, nullptr, ds);
    return pnr_check_launch("pnr_depth_sample_backward");
}

// slab form of the scatter: channels per slab (16 / 8 / 4) and fp64 slots per texel row (padded by one when that fits the LDS;
// 64x64: unpadded); cs = 0: the grid does not fit, global atomics.  Of the widths that fit, the widest one whose (image, slice)
// pairs fill the chip with at most TWO workgroups per pair is taken (4+ images of 32x32: 16 channels; 2 images: 8; 1 image: 4):
// wider slices read the gradient rows in longer pieces (profiles/r06_scatter_notes.md section 4), and at most two atomic adds per
// grid element keep the result independent of their order (onto a zeroed buffer two terms commute).
static bool slab_row(int texels, int cs, int &row) {
    if ((size_t)texels * (cs + 1) * 8 <= SLAB_MAX_BYTES - 128) { row = cs + 1; return true; }
    if (cs == 4 && (size_t)texels * 4 * 8 <= SLAB_MAX_BYTES - 128) { row = 4; return true; }
    return false;
}
static void scatter_form(int texels, int images, int &cs, int &row) {
    cs = 0; row = 0;
    const int cus = bwd_num_cus();
    for (int c = 16; c >= 4; c >>= 1) {
        int r = 0;
        if (!slab_row(texels, c, r)) continue;
        if (!cs) { cs = c; row = r; }                                      // the widest that fits, unless a narrower one fills the chip
        if (2 * images * (C_LAT / c) >= cus) { cs = c; row = r; return; }  // ... with <= 2 workgroups per (image, slice)
    }
    if (cs) { int r = 0; if (slab_row(texels, 4, r)) { cs = 4; row = r; } }  // very few images: the most pairs there are
}
// workspace of the slab form: coords NS*P float2 | segment starts, SEG_NSUB x sub_len ints per image | counts, SEG_NSUB ints per image
static int scatter_sub_len(long long pts) { return (int)(((pts + SEG_NSUB - 1) / SEG_NSUB + 63) / 64 * 64); }
static size_t scatter_ws_bytes(int images, int NS, long long P, long long pts) {
    return (size_t)NS * P * sizeof(float2) + (size_t)images * SEG_NSUB * scatter_sub_len(pts) * sizeof(int) + (size_t)images * SEG_NSUB * sizeof(int);
}

// large grids (tiled form) add: per (image, tile) counts | offsets (+1) | cursors | list entries, at most four per segment
static pnr::TileGeom tile_geom(int Hl, int Wl) {
    pnr::TileGeom g;
    g.tx = (Wl + pnr::TILE_W - 1) / pnr::TILE_W; g.ty = (Hl + pnr::TILE_W - 1) / pnr::TILE_W; g.ntiles = g.tx * g.ty;
    return g;
}
static size_t scatter_tiled_extra_bytes(int images, int ntiles, int NS, long long P) {
    return ((size_t)images * ntiles * 3 + 1) * sizeof(int) + 4 * (size_t)NS * P * sizeof(unsigned);
}
// Grids whose slab only fits 4 channels wide (2275 .. 5116 texels: the 64 x 64 grids of SRN-sized images) read the gradient rows in
// 16-byte pieces; cut into tiles they read 64-byte pieces like the small grids (4 x 64 x 64: 165 -> see profiles/r06_scatter_notes.md).
// PNR_SCATTER_TILED_MIN_TEXELS moves the crossover (experiment hook).
static bool scatter_prefers_tiles(int cs, int texels) {
    static const long long min_texels = [] { const char *e = getenv("PNR_SCATTER_TILED_MIN_TEXELS"); return e ? atoll(e) : -1LL; }();
    if (min_texels >= 0) return texels >= min_texels;
    int r = 0;
    return cs == 4 && !slab_row(texels, 8, r);  // (one or two small images also get cs == 4 -- for the pair count; they stay slabs)
}
// the tiled form needs 29-bit sample indices inside an object; beyond that (and with PIXELNERF_SCATTER_TILED=0) the global-atomic kernel runs
static bool scatter_tiled_ok(long long pts, int ntiles = 1) {
    if (ntiles > 8192) return false;  // (the binning histogram lives in LDS: 2 x 4 bytes per tile)
    static const bool off = [] { const char *e = getenv("PIXELNERF_SCATTER_TILED"); return e && e[0] == '0'; }();
    return !off && pts < (1LL << 29);
}

extern "C" size_t pnr_latent_scatter_workspace_bytes(const PnrScene *s, int R, int rays_per_obj, int K) {
    if (!s || R <= 0 || K <= 0 || rays_per_obj <= 0) return 0;
    int cs, row;
    scatter_form(s->Hl * s->Wl, s->SB * s->NS, cs, row);
    const long long P = (long long)R * K, pts = (long long)rays_per_obj * K;
    if (scatter_prefers_tiles(cs, s->Hl * s->Wl) && scatter_tiled_ok(pts, tile_geom(s->Hl, s->Wl).ntiles)) cs = 0;
    if (cs) return scatter_ws_bytes(s->SB * s->NS, s->NS, P, pts);
    if (!scatter_tiled_ok(pts, tile_geom(s->Hl, s->Wl).ntiles)) return 0;
    return (scatter_ws_bytes(s->SB * s->NS, s->NS, P, pts) + 15) / 16 * 16 +
           scatter_tiled_extra_bytes(s->SB * s->NS, tile_geom(s->Hl, s->Wl).ntiles, s->NS, P);
}

// 1: every element of the grid gradient is written by ONE workgroup with plain read-add-write (the tiled form): successive calls may
// accumulate into one buffer and the result still does not depend on any execution order; 0: the two-workgroups-per-pair / global-atomic
// forms, which are order-free only onto a zeroed buffer
extern "C" int pnr_latent_scatter_single_owner(const PnrScene *s, int R, int rays_per_obj, int K) {
    if (!s || R <= 0 || K <= 0 || rays_per_obj <= 0) return 0;
    int cs, row;
    scatter_form(s->Hl * s->Wl, s->SB * s->NS, cs, row);
    const bool tiles_ok = scatter_tiled_ok((long long)rays_per_obj * K, tile_geom(s->Hl, s->Wl).ntiles);
    if (getenv("PNR_SCATTER_FORM")) return 0;  // (experiment hook active: unknown form)
    return ((cs == 0 || scatter_prefers_tiles(cs, s->Hl * s->Wl)) && tiles_ok) ? 1 : 0;
}

extern "C" int pnr_latent_scatter(const PnrScene *s, const float *rays, const float *z, int R, int rays_per_obj, int K,
                                  const float *d_zlat, float *d_latent_nhwc, void *workspace, size_t workspace_bytes, void *stream) {
    if (!s || !rays || !z || !d_zlat || !d_latent_nhwc || R <= 0 || K <= 0 || rays_per_obj <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_latent_scatter: bad argument");
    if ((long long)rays_per_obj * s->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_latent_scatter: R != SB * rays_per_obj");
    EvalParams q = {};
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K;
    const int texels = q.Hl * q.Wl;
    int cs, row;
    scatter_form(texels, q.SB * q.NS, cs, row);
    if (scatter_prefers_tiles(cs, texels) && scatter_tiled_ok((long long)rays_per_obj * K, tile_geom(q.Hl, q.Wl).ntiles)) cs = 0;
    int force_psplit = 0;
    if (const char *e = getenv("PNR_SCATTER_FORM")) {  // experiment hook "cs,psplit": force the slice width (16 / 8 / 4) and the point split
        int fcs = 0, fps = 0;
        int frow = 0;
        if (cs && sscanf(e, "%d,%d", &fcs, &fps) >= 1 && (fcs == 16 || fcs == 8 || fcs == 4) && slab_row(texels, fcs, frow)) {
            cs = fcs; row = frow;
            force_psplit = fps;
        }
    }
    if (cs) {
        const size_t lds = (size_t)texels * row * 8;
        const long long pts = (long long)rays_per_obj * K;
        // one workgroup per (image, slice) takes all of the image's segments; they are split only when there are fewer
        // (image, slice) pairs than compute units (and never below ~one segment per thread: a segment is >= 1 sample)
        const int owners = q.SB * q.NS * (C_LAT / cs);
        int psplit = (bwd_num_cus() + owners - 1) / owners;
        const long long rounds = (pts + 4LL * OWNER_NT - 1) / (4LL * OWNER_NT);
        if (psplit > rounds) psplit = (int)rounds;
        if (psplit > 2) psplit = 2;  // (scatter_form: at most two adds per grid element)
        if (psplit < 1) psplit = 1;
        if (force_psplit >= 1 && force_psplit <= 8) psplit = force_psplit;
        const int images = q.SB * q.NS;
        const size_t coords_bytes = (size_t)q.NS * q.P * sizeof(float2);
        const int sub_len = scatter_sub_len(pts);
        const size_t segs_bytes = (size_t)images * SEG_NSUB * sub_len * sizeof(int);
        if (!workspace || workspace_bytes < scatter_ws_bytes(images, q.NS, q.P, pts) || ((uintptr_t)workspace & 15) != 0)
            return pnr_fail(PNR_E_INVALID, "pnr_latent_scatter: workspace missing, misaligned (16 bytes) or smaller than "
                                           "pnr_latent_scatter_workspace_bytes()");
        char *scratch = reinterpret_cast<char *>(workspace);
        float2 *coords = reinterpret_cast<float2 *>(scratch);
        int *segs = reinterpret_cast<int *>(scratch + coords_bytes);
        int *nseg = reinterpret_cast<int *>(scratch + coords_bytes + segs_bytes);
        auto k = cs == 16 ? latent_scatter_owner_kernel<16> : (cs == 8 ? latent_scatter_owner_kernel<8> : latent_scatter_owner_kernel<4>);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           SLAB_MAX_BYTES - 128);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(latent_scatter_owner_kernel)");
        hipLaunchKernelGGL(scatter_segments_kernel, dim3((unsigned)(images * SEG_NSUB)), dim3(SEG_NT), 0, (hipStream_t)stream, q, coords,
                           segs, nseg, sub_len);
        hipLaunchKernelGGL(k, dim3((unsigned)(owners * psplit)), dim3(OWNER_NT), lds, (hipStream_t)stream, q, d_zlat, coords, segs, nseg,
                           sub_len, d_latent_nhwc, psplit, row);
        return pnr_check_launch("pnr_latent_scatter");
    }
    const long long pts_obj = (long long)rays_per_obj * K;
    if (scatter_tiled_ok(pts_obj, tile_geom(q.Hl, q.Wl).ntiles)) {
        // large grid: 32 x 32-texel tiles, one owner workgroup per (image, tile, 16-channel slice) (latent_scatter_tiled_kernel)
        const int images = q.SB * q.NS;
        const TileGeom tg = tile_geom(q.Hl, q.Wl);
        const size_t base_bytes = (scatter_ws_bytes(images, q.NS, q.P, pts_obj) + 15) / 16 * 16;
        if (!workspace || workspace_bytes < base_bytes + scatter_tiled_extra_bytes(images, tg.ntiles, q.NS, q.P) || ((uintptr_t)workspace & 15) != 0)
            return pnr_fail(PNR_E_INVALID, "pnr_latent_scatter: workspace missing, misaligned (16 bytes) or smaller than "
                                           "pnr_latent_scatter_workspace_bytes()");
        char *scratch = reinterpret_cast<char *>(workspace);
        const size_t coords_bytes = (size_t)q.NS * q.P * sizeof(float2);
        const int sub_len = scatter_sub_len(pts_obj);
        const size_t segs_bytes = (size_t)images * SEG_NSUB * sub_len * sizeof(int);
        float2 *coords = reinterpret_cast<float2 *>(scratch);
        int *segs = reinterpret_cast<int *>(scratch + coords_bytes);
        int *nseg = reinterpret_cast<int *>(scratch + coords_bytes + segs_bytes);
        const int IT = images * tg.ntiles;
        int *tile_cnt = reinterpret_cast<int *>(scratch + base_bytes);
        int *tile_off = tile_cnt + IT, *cursor = tile_off + IT + 1;
        unsigned *entries = reinterpret_cast<unsigned *>(cursor + IT);
        hipStream_t st = (hipStream_t)stream;
        hipError_t e = hipMemsetAsync(tile_cnt, 0, (size_t)IT * sizeof(int), st);
        if (e != hipSuccess) return pnr_check_hip(e, "hipMemsetAsync(tile counts)");
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(latent_scatter_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TILE_LDS);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(latent_scatter_tiled_kernel)");
        hipLaunchKernelGGL(scatter_segments_kernel, dim3((unsigned)(images * SEG_NSUB)), dim3(SEG_NT), 0, st, q, coords, segs, nseg, sub_len);
        const dim3 bgrid((unsigned)((sub_len + 255) / 256), (unsigned)(images * SEG_NSUB));
        const size_t bin_lds = 2 * (size_t)tg.ntiles * sizeof(int);
        hipLaunchKernelGGL(tile_bin_kernel<false>, bgrid, dim3(256), bin_lds, st, q, coords, segs, nseg, sub_len, tg, tile_cnt, tile_off, cursor, entries);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, tile_cnt, IT, tile_off, cursor);
        hipLaunchKernelGGL(tile_bin_kernel<true>, bgrid, dim3(256), bin_lds, st, q, coords, segs, nseg, sub_len, tg, tile_cnt, tile_off, cursor, entries);
        hipLaunchKernelGGL(latent_scatter_tiled_kernel, dim3((unsigned)(IT * (C_LAT / TILE_CS))), dim3(OWNER_NT), TILE_LDS, st, q, d_zlat, coords,
                           entries, tile_off, tg, d_latent_nhwc);
        return pnr_check_launch("pnr_latent_scatter");
    }
    const long long n = ((q.P + SCATTER_RUN - 1) / SCATTER_RUN) * q.NS;  // wavefronts
    hipLaunchKernelGGL(latent_scatter_kernel, dim3((unsigned)((n + CW - 1) / CW)), dim3(CW * 64), 0, (hipStream_t)stream, q,
                       d_zlat, d_latent_nhwc);
    return pnr_check_launch("pnr_latent_scatter");
}
