// pnr_pack.hip -- one-time repack of a ResnetFC's nn.Linear parameters into the per-wave MFMA
// fragment stream the fused kernel consumes (layout: pnr_layout.h), plus the NCHW->NHWC
// transpose of the encoder feature grid.  gfx950.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pnr_common.h"
#include "pnr_device.h"  // EvalParams, project_point: the sparse fold marks rows with the forward kernels' own projection
#include "pnr_internal.h"
#include "pnr_layout.h"

namespace pnr {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// nn.Linear parameters are read through a 4-BYTE-aligned vector type: a parameter that is a view into a flat buffer (flattened /
// FSDP-style storages, load_state_dict(assign=True) from an mmap) need not start on a 16-byte boundary.  gfx950 code objects run
// in the target's unaligned-access mode, where this still compiles to one global_load_dwordx4; without that mode the compiler
// splits the access instead of faulting.

// which nn.Linear feeds GEMM g, and how its K index is ordered in the stream
struct GemmSrc {
    const float *w;
    int kind;  // 0 lin_in (natural K, 42 wide), 1 natural K (lin_z), 2 LDS-permuted K (fc_0/fc_1), 3 lin_out
};

__device__ inline GemmSrc gemm_source(const PnrMlpWeights &p, int g) {
    switch (g) {
        case G_LIN_IN: return {p.lin_in_w, 0};
        case G_Z0: return {p.lin_z_w[0], 1};
        case G_Z1: return {p.lin_z_w[1], 1};
        case G_Z2: return {p.lin_z_w[2], 1};
        case G_FC0_0: return {p.fc0_w[0], 2};
        case G_FC1_0: return {p.fc1_w[0], 2};
        case G_FC0_1: return {p.fc0_w[1], 2};
        case G_FC1_1: return {p.fc1_w[1], 2};
        case G_FC0_2: return {p.fc0_w[2], 2};
        case G_FC1_2: return {p.fc1_w[2], 2};
        case G_FC0_3: return {p.fc0_w[3], 2};
        case G_FC1_3: return {p.fc1_w[3], 2};
        case G_FC0_4: return {p.fc0_w[4], 2};
        case G_FC1_4: return {p.fc1_w[4], 2};
        default: return {p.lin_out_w, 3};
    }
}

// FOLD: the stream without the three lin_z GEMMs (they are folded into per-texel tables, pnr_fold_latent)
// LO: the f16 TAIL of the weight, f16(w - f16(w)), for the split-operand kernel (pnr_split.hip)
// One thread = one lane's 8-element fragment slice (a 16-byte store; the index arithmetic is paid once per 8 elements).
// OWNK: the K order of the split-operand kernel's 512-wide linears (pnr_split.hip, stage_own / gemm_split_rot): ring steps
//       0-3 of a layer = this wave's OWN K block (features 64 wv .. 64 wv + 63) in REGISTER order -- k-step j, lane half h,
//       element e <-> feat_of(wv IT + (j >> 1), h, 8 (j & 1) + e): what the wave's accumulators hold -- then the blocks of waves
//       wv+1 .. wv+7 (mod 8) in the storage order of the operand image.
// out_lo (split-operand stream only): the tail stream is written by the same thread from the same loads (one pass over the
// weights instead of two).
template <typename T, bool FOLD, bool LO = false, bool OWNK = false>
__global__ void pack_weights_kernel(PnrMlpWeights p, T *__restrict__ out, T *__restrict__ out_lo = nullptr) {
    constexpr int TOTAL = FOLD ? RS_TOTAL_F : RS_TOTAL;
    const size_t idx8 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx8 >= (size_t)TOTAL * IT * (FRAG_ELEMS / 8) * NW) return;
    const int lane = idx8 & 63;
    const int it = (idx8 >> 6) % IT;
    const size_t rest = idx8 / ((FRAG_ELEMS / 8) * IT);
    const int rs = rest % TOTAL;
    const int wv = rest / TOTAL;
    int g = 0, s = 0;
    if (FOLD) {
        for (int i = 0; i < NGEMM; ++i)
            if (!gemm_is_linz(i) && rs >= gemm_offset_fold(i)) g = i;
        s = rs - gemm_offset_fold(g);
    } else {
        while (g + 1 < NGEMM && rs >= gemm_offset(g + 1)) ++g;
        s = rs - gemm_offset(g);
    }
    const int i = lane & 31, h = lane >> 5;
    const int f_out = wv * SL + it * 32 + i;
    const GemmSrc src = gemm_source(p, g);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (src.kind == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = s * 16 + h * 8 + e;
            if (k < D_IN) v[e] = src.w[f_out * D_IN + k];
        }
    } else if (src.kind == 1) {
        const f32x4_param *q = reinterpret_cast<const f32x4_param *>(src.w + (size_t)f_out * C_LAT + s * 16 + h * 8);
        const f32x4 a = q[0], b = q[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else if (src.kind == 2) {
        // B operand comes from the LDS activation buffer: k-step s reads storage elements
        // 16s..16s+15 = what half (s&1) of feature tile (s>>1) wrote as registers 8h+e.
        // feat_of(T, hh, r) for r = r0 .. r0+7 (r0 a multiple of 8) = two runs of four consecutive features, 8 apart
        int k;
        if (OWNK) {
            constexpr int KSB = SL / 16;  // k-steps per wave block (4 at 8 waves)
            const int blk = s / KSB, j = s % KSB;
            if (blk == 0) k = feat_of(wv * IT + (j >> 1), h, 8 * (j & 1));
            else {
                const int ss = ((wv + blk) & (NW - 1)) * KSB + j;  // k-step of the image order this ring step stands for
                k = feat_of(ss >> 1, ss & 1, 8 * h);
            }
        } else k = feat_of(s >> 1, s & 1, 8 * h);
        const float *row = src.w + (size_t)f_out * D_HID + k;
        const f32x4 a = *reinterpret_cast<const f32x4_param *>(row), b = *reinterpret_cast<const f32x4_param *>(row + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
    } else {
        // lin_out: B operand = the wave's own accumulators; k-step q = IT*s + it covers
        // registers 8*(q&1)..+7 of the wave's feature tile (q>>1), for both lane halves.
        if (s < 2 && i < D_OUT) {
            const int q = IT * s + it;
            const float *row = src.w + i * D_HID + feat_of(wv * IT + (q >> 1), h, 8 * (q & 1));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = row[e]; v[4 + e] = row[8 + e]; }
        }
    }
    __attribute__((aligned(16))) T o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = LO ? (T)(v[e] - (float)(T)v[e]) : (T)v[e];
    *reinterpret_cast<uint4 *>(out + idx8 * 8) = *reinterpret_cast<const uint4 *>(o);
    if (out_lo) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (T)(v[e] - (float)(T)v[e]);
        *reinterpret_cast<uint4 *>(out_lo + idx8 * 8) = *reinterpret_cast<const uint4 *>(o);
    }
}

template <bool FOLD>
__global__ void pack_bias_kernel(PnrMlpWeights p, float *__restrict__ bias, float *__restrict__ bout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < D_OUT) bout[idx] = p.lin_out_b[idx];
    if (idx == 0) reinterpret_cast<int *>(bout)[BOUT_FLAGS_INDEX] = p.combine_max ? 1 : 0;
    if (idx >= NBIAS * NW * BIAS_FLOATS_PER_WAVE) return;
    const int r = idx & 15, h = (idx >> 4) & 1, it = (idx >> 5) % IT;
    const int wv = (idx / BIAS_FLOATS_PER_WAVE) % NW;
    const int slot = idx / (BIAS_FLOATS_PER_WAVE * NW);
    const int f = feat_of(wv * IT + it, h, r);
    float v = 0.f;
    switch (slot) {
        case B_IN_Z0: v = p.lin_in_b[f] + (FOLD ? 0.f : p.lin_z_b[0][f]); break;  // folded: the tables carry lin_z's bias
        case B_FC0_0: v = p.fc0_b[0][f]; break;
        case B_FC1_0_Z1: v = p.fc1_b[0][f] + (FOLD ? 0.f : p.lin_z_b[1][f]); break;
        case B_FC0_1: v = p.fc0_b[1][f]; break;
        case B_FC1_1_Z2: v = p.fc1_b[1][f] + (FOLD ? 0.f : p.lin_z_b[2][f]); break;
        case B_FC0_2: v = p.fc0_b[2][f]; break;
        case B_FC1_2: v = p.fc1_b[2][f]; break;
        case B_FC0_3: v = p.fc0_b[3][f]; break;
        case B_FC1_3: v = p.fc1_b[3][f]; break;
        case B_FC0_4: v = p.fc0_b[4][f]; break;
        case B_FC1_4: v = p.fc1_b[4][f]; break;
    }
    bias[idx] = v;
}

// (N,C,H,W) -> (N,H,W,C) through a 32x32 LDS tile: coalesced on both sides.
__global__ void nchw_to_nhwc_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    const float *src = in + (size_t)n * C * HW;
    float *dst = out + (size_t)n * C * HW;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        if (c < C && p < HW) tile[j][tx] = src[(size_t)c * HW + p];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (c < C && p < HW) dst[(size_t)p * C + c] = tile[tx][j];
    }
}

// Folding lin_z into the feature grid (inference): table[b][texel][slot_of(n)] = sum_k W_z[b][n][k] grid[texel][k] + b_z[b][n],
// fp32 on v_mfma_f32_32x32x2_f32 (exact products, fp32 sums), stored 16-bit saturated, hidden features in storage
// order.  lin_z[b](bilinear(grid)) == bilinear(table[b]) by linearity (the bilinear weights sum to 1), so the
// per-point stream loses three of its 13.4 GEMMs.  Block = 4 waves = 64 texels x 64 features, K staged 32 at a time.
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// blockIdx.z = table b: the three tables of a network are ONE launch (the sn64 grid gives 128 blocks per table, a launch-
// and latency-bound 62 us each when launched one by one)
struct FoldJobs {
    const float *W[COMBINE_LAYER], *bias[COMBINE_LAYER];
};
template <typename T>
__global__ void __launch_bounds__(256)
fold_kernel(const float *__restrict__ grid, const FoldJobs jobs, T *__restrict__ tables, long long M, float max_finite) {
    __shared__ float sX[64][33], sW[64][33];
    const float *__restrict__ W = jobs.W[blockIdx.z];
    const float *__restrict__ bias = jobs.bias[blockIdx.z];
    T *__restrict__ table = tables + (size_t)blockIdx.z * (size_t)M * D_HID;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long m0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int i = lane & 31, kh = lane >> 5;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < C_LAT; k0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = t + u * 256, row = e >> 5, col = e & 31;
            sX[row][col] = (m0 + row < M) ? grid[(m0 + row) * C_LAT + k0 + col] : 0.f;
            sW[row][col] = W[(size_t)(n0 + row) * C_LAT + k0 + col];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[wm + i][2 * kk + kh], sW[wn + i][2 * kk + kh], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn + i;  // D layout: column = lane&31 -> feature, row (r&3)+8(r>>2)+4kh -> texel
    const float bn = bias[n];
    const int slot = slot_of(n);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < M) table[m * D_HID + slot] = (T)__builtin_amdgcn_fmed3f(acc[r] + bn, -max_finite, max_finite);
    }
}


// The same fold at fp32-CLASS precision on the f16 matrix cores (round 4): both operands are split into fp16 (head, tail) pairs
// on their way into LDS and every product is the three MFMAs of pnr_split.hip (x w ~= xh wh + xh wl + xl wh, fp32 accumulate;
// the dropped tail-tail term is 2^-22 of the product) -- the arithmetic class of the kernel that consumes the tables, at ~5x the
// rate of the fp32-MFMA fold above (157 TFLOP/s peak against 3 MFMAs on a 2.5 PFLOP/s pipe).  The fold is per scene at
// inference (2.5 ms per network for the DTU grid, on EVERY rank of a sharded render) and per STEP in training (the grid is a
// trained tensor: 0.27 ms of the 6.3 ms fp32-class step).  128 texels x 128 features per 256-thread workgroup (4 waves of
// 64 x 64 = 2 x 2 MFMA tiles), K chunks of 32; [row][k] f16 images with 80-byte rows (conflict-free 16-byte fragment reads).
// sat: when non-null, bit 12 is raised if a grid value or a lin_z weight is beyond the fp16 range (pnr_saturation_guard).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr int FS_TM = 128, FS_TN = 128, FS_K = 32, FS_ROW = FS_K + 8;  // halves per LDS row (32 used)
__global__ void __launch_bounds__(256)
fold_split_kernel(const float *__restrict__ grid, const FoldJobs jobs, float *__restrict__ tables, long long M, unsigned int *sat) {
    __shared__ __attribute__((aligned(16))) _Float16 sXh[FS_TM][FS_ROW], sXl[FS_TM][FS_ROW], sWh[FS_TN][FS_ROW], sWl[FS_TN][FS_ROW];
    const float *__restrict__ W = jobs.W[blockIdx.z];
    const float *__restrict__ bias = jobs.bias[blockIdx.z];
    float *__restrict__ table = tables + (size_t)blockIdx.z * (size_t)M * D_HID;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long m0 = (long long)blockIdx.x * FS_TM;
    const int n0 = blockIdx.y * FS_TN;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int i = lane & 31, kh = lane >> 5;
    f32x16_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float amax = 0.f;
    __builtin_amdgcn_s_setreg(1473, 1);  // MODE.FP16_OVFL = 1 (hwreg(HW_REG_MODE, 23, 1)): f16 conversions saturate, as in the fused kernels
    auto split4 = [&](const f32x4 v, _Float16 *hi, _Float16 *lo) {
        // heads and tails SATURATE at the fp16 limit like the operands of the kernel that reads the tables (MODE.FP16_OVFL): an
        // out-of-range grid value or weight gives a large finite table entry (and raises the guard bit), never inf - inf.
        // head = v_cvt_pk_f16_f32 per pair, tail = one v_fma_mix{lo,hi}_f16 per value (fp32 FMA of the packed head times -1 plus v,
        // rounded once: the bits of convert-back + subtract + convert; pnr_split.hip split8) -- round 6: 1.5 VALU per value, was 8
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        uint32_t h[2], l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float a = v[2 * c], b = v[2 * c + 1];
            h[c] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, f16x2_t));
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l[c]) : "v"(h[c]), "v"(a));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l[c]) : "v"(h[c]), "v"(b));
            amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b)));
        }
        *reinterpret_cast<u32x2_t *>(hi) = u32x2_t{h[0], h[1]};
        *reinterpret_cast<u32x2_t *>(lo) = u32x2_t{l[0], l[1]};
    };
    // Register ring of FS_DEPTH chunks of both operands; chunks are requested FS_DEPTH - 1 ahead.  Round 6 measured depth 4 (256
    // registers): 1024- and 4096-texel grids -- what this kernel serves since fold_split_big_kernel took the large ones -- stay at
    // 46 us per network either way (16 chunks of two barriers each on 96-384 four-wave workgroups: not a load-latency chain), the
    // DTU grid went 840 -> 720 us; depth 2 (one chunk ahead, ~130 registers, four workgroups per CU) is kept.
    constexpr int FS_DEPTH = 2;
    f32x4 xv[FS_DEPTH][4], wv[FS_DEPTH][4];
    // thread -> (row, 4 columns) of a 128-row x 32-column chunk.  Rows of consecutive 8-lane groups are 4 apart (bits 0 and 2 of the
    // group index swapped): the 16 lanes one ds_write_b64 cycle serves then hit 2 x 64 bytes that are 320 bytes = 16 banks (mod 32)
    // apart -- with neighbouring rows (80 bytes apart) four banks were hit twice: 33 % of the LDS cycles were conflicts
    // (profiles/r04_train_step_f16x3_pmc.txt)
    auto row_of = [](int e) { const int g = e >> 3; return (g & ~5) | ((g & 1) << 2) | ((g >> 2) & 1); };
    auto fetch = [&](int k0, f32x4 (&x)[4], f32x4 (&wq)[4]) {  // 128 rows x 8 float4 per operand
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 256, row = row_of(e), c4 = (e & 7) * 4;
            x[u] = (m0 + row < M) ? *reinterpret_cast<const f32x4 *>(grid + (m0 + row) * C_LAT + k0 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            wq[u] = *reinterpret_cast<const f32x4_param *>(W + (size_t)(n0 + row) * C_LAT + k0 + c4);
        }
    };
#pragma unroll
    for (int d = 0; d < FS_DEPTH - 1; ++d) fetch(d * FS_K, xv[d], wv[d]);
#pragma unroll
    for (int kc = 0; kc < C_LAT / FS_K; ++kc) {  // fully unrolled (16 chunks): the ring slot is a compile-time index
        constexpr int NCH = C_LAT / FS_K;
        const int slot = kc % FS_DEPTH;
        __syncthreads();  // the previous chunk's fragments have been read
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 256, row = row_of(e), c4 = (e & 7) * 4;
            split4(xv[slot][u], &sXh[row][c4], &sXl[row][c4]);
            split4(wv[slot][u], &sWh[row][c4], &sWl[row][c4]);
        }
        __syncthreads();
        if (kc + FS_DEPTH - 1 < NCH) fetch((kc + FS_DEPTH - 1) * FS_K, xv[(kc + FS_DEPTH - 1) % FS_DEPTH], wv[(kc + FS_DEPTH - 1) % FS_DEPTH]);
#pragma unroll
        for (int kk = 0; kk < FS_K / 16; ++kk) {
            f16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ah[a] = *reinterpret_cast<const f16x8_t *>(&sXh[wm + a * 32 + i][kk * 16 + kh * 8]);
                al[a] = *reinterpret_cast<const f16x8_t *>(&sXl[wm + a * 32 + i][kk * 16 + kh * 8]);
                bh[a] = *reinterpret_cast<const f16x8_t *>(&sWh[wn + a * 32 + i][kk * 16 + kh * 8]);
                bl[a] = *reinterpret_cast<const f16x8_t *>(&sWl[wn + a * 32 + i][kk * 16 + kh * 8]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
    // D layout: column = lane & 31 -> feature, row (r & 3) + 8 (r >> 2) + 4 kh -> texel
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn + b * 32 + i;
        const float bn = bias[n];
        const int slot = slot_of(n);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m < M) table[m * D_HID + slot] = acc[a][b][r] + bn;
            }
    }
    if (sat && amax >= 65504.f) atomicOr(sat, 1u << 12);
}


// The fold for LARGE grids (round 6; DTU: 90 000 texels, 3 tables x 2 networks per scene and per rank of a sharded render).  The
// 128 x 128 kernel above spends a chunk as split (VALU, matrix pipe idle) | barrier | 24 MFMAs per wave | barrier at four waves
// per workgroup: 24.8 % MFMA-busy.  Here the structure of dw_split_kernel (pnr_bwd.hip): 256 texels x 256 features per 512-thread
// workgroup (8 waves of 64 x 128 = 2 x 4 MFMA tiles, 48 MFMAs per wave and chunk), the four operand images (X / W, head / tail)
// DOUBLE-buffered in LDS -- the next chunk's rows are requested at the top of a chunk and split + stored into the other buffer
// between its two k-steps, under the MFMAs -- one barrier per chunk.  64-byte image rows (32 halves, no padding) with the 16-byte
// units of a row XOR-swizzled by (row >> 1) & 3: the eight rows one ds_read_b128 cycle serves land on eight different
// 4-bank groups; 128 KiB of LDS.  Workgroups are placed XCD-aware: the six (column tile, table) workgroups of a 256-texel
// row tile take consecutive slots of ONE XCD, so its 512 KiB of grid rows cross the fabric once.
// Same-box A/B (profiles/r06_fold_notes.md): DTU grid 840 -> 640 us per network (664 TFLOP/s of executed MFMAs), srn_car 80 -> 65 us;
// 4096 texels and fewer stay on the 128 x 128 kernel (45 vs 60 us: 96 workgroups of this one do not fill the chip).  Timing twins of
// this kernel on the DTU grid: no global loads 595, no table stores 555, no split / LDS stores 537 us -- what is left is the
// fragment-read + MFMA loop itself at two waves per SIMD with one barrier per chunk (the 128 accumulator registers of the 64 x 128
// wave tile leave no room to double-buffer the fragments).
constexpr int FB_TM = 256, FB_TN = 256, FB_K = 32;
constexpr int FB_IMG = FB_TM * FB_K * 2;  // bytes of one image (256 rows x 64 B)
constexpr int FB_LDS = 2 * 4 * FB_IMG;     // [buffer][Xh, Xl, Wh, Wl]
// SPARSE (training on large grids, pnr_fold_latent_f32_rows): the row tile is a run of 256 entries of `rows` -- the texels one
// training pass reads, in ascending order -- and M is the device-side count *nrows; a table row is written at its texel's
// place, so the kernel that reads the tables is unchanged.  Every row is the same sum in the same order as in the dense form.
template <bool SPARSE>
__global__ void __launch_bounds__(512)
fold_split_big_kernel(const float *__restrict__ grid, const FoldJobs jobs, float *__restrict__ tables, long long M, int ngroups,
                      unsigned int *sat, const int *__restrict__ rows, const int *__restrict__ nrows) {
    extern __shared__ __attribute__((aligned(16))) char fb[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // (XCD, slot) -> (row tile, column tile, table)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int group = (slot / 6) * 8 + xcd, sub = slot % 6;
    if (group >= ngroups) return;
    const long long table_rows = M;  // rows of one table (its stride)
    if (SPARSE) {
        M = *nrows;
        if ((long long)group * FB_TM >= M) return;  // uniform
    }
    const int tab = sub >> 1;
    const float *__restrict__ W = jobs.W[tab];
    const float *__restrict__ bias = jobs.bias[tab];
    float *__restrict__ table = tables + (size_t)tab * (size_t)table_rows * D_HID;
    const long long m0 = (long long)group * FB_TM;
    const int n0 = (sub & 1) * FB_TN;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 128;  // wave tile: 64 texels x 128 features
    const int i = lane & 31, kh = lane >> 5;
    __builtin_amdgcn_s_setreg(1473, 1);  // MODE.FP16_OVFL = 1 (hwreg(HW_REG_MODE, 23, 1)): f16 conversions saturate, as in the fused kernels
    // computed TRANSPOSED like the fused kernels (table^T = W grid^T; A = W fragment, B = grid fragment): a lane's 16 D registers of
    // a tile are then features feat_of(T, kh, 0..15) of ONE texel = 16 CONSECUTIVE slots of the table row in storage order
    // (pnr_layout.h): 64 contiguous bytes, four 16-byte stores -- 32 store instructions per thread where the texel-major form
    // issued 128 scattered dwords
    f32x16_t acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float amax = 0.f;
    // byte offset of (row, 16-byte unit u) inside an image
    auto img_off = [](int row, int u) { return row * 64 + ((u ^ ((row >> 1) & 3)) << 4); };
    f32x4 xv[4], wv[4];
    long long xrow[4];  // grid row of this thread's four chunk rows (SPARSE: through the list; -1 beyond its end)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long m = m0 + ((t + u * 512) >> 3);
        xrow[u] = m < M ? (SPARSE ? (long long)rows[m] : m) : -1;
    }
    // thread -> (row, 4 columns) of a 256-row x 32-column chunk: 8 consecutive threads read one row's 128 bytes
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 512, row = e >> 3, c4 = (e & 7) * 4;
            xv[u] = xrow[u] >= 0 ? *reinterpret_cast<const f32x4 *>(grid + xrow[u] * C_LAT + k0 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            wv[u] = *reinterpret_cast<const f32x4_param *>(W + (size_t)(n0 + row) * C_LAT + k0 + c4);
        }
    };
    auto split_store = [&](int buf) {
        char *base = fb + buf * (4 * FB_IMG);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = t + u * 512, row = e >> 3, c4 = (e & 7) * 4;
            const int off = img_off(row, c4 >> 3) + (c4 & 4) * 2;  // 8 bytes: 4 halves
            // head = f16(v), tail = f16(v - head) as pnr_split.hip's split8 makes them: MODE.FP16_OVFL is set, so the conversions
            // SATURATE at the fp16 limit like the operands of the kernel that reads the tables (an out-of-range grid value or weight
            // gives a large finite table entry and raises the guard bit, never inf - inf); one v_cvt_pk_f16_f32 per pair + one
            // v_fma_mix{lo,hi}_f16 per value (fp32 FMA of the packed f16 head times -1 plus v, rounded once: the bits of
            // convert-back + subtract + convert) -- 1.5 VALU per value where clamp / convert / subtract / clamp / convert took 8
            uint32_t xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float x0 = xv[u][2 * c], x1 = xv[u][2 * c + 1], w0 = wv[u][2 * c], w1 = wv[u][2 * c + 1];
                typedef float f32x2_t __attribute__((ext_vector_type(2)));
                typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                xh[c] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){x0, x1}, f16x2_t));
                wh[c] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){w0, w1}, f16x2_t));
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(xl[c]) : "v"(xh[c]), "v"(x0));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(xl[c]) : "v"(xh[c]), "v"(x1));
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(wl[c]) : "v"(wh[c]), "v"(w0));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(wl[c]) : "v"(wh[c]), "v"(w1));
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(w0), fabsf(w1))));
            }
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2_t *>(base + off) = u32x2_t{xh[0], xh[1]};
            *reinterpret_cast<u32x2_t *>(base + FB_IMG + off) = u32x2_t{xl[0], xl[1]};
            *reinterpret_cast<u32x2_t *>(base + 2 * FB_IMG + off) = u32x2_t{wh[0], wh[1]};
            *reinterpret_cast<u32x2_t *>(base + 3 * FB_IMG + off) = u32x2_t{wl[0], wl[1]};
        }
    };
    fetch(0);
    split_store(0);
    fetch(FB_K);  // one register set: chunk c + 2 is requested right behind the split of chunk c + 1, a whole chunk before its use
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < C_LAT; k0 += FB_K, cur ^= 1) {
        const bool more = k0 + FB_K < C_LAT;
        const char *sXh = fb + cur * (4 * FB_IMG), *sXl = sXh + FB_IMG, *sWh = sXh + 2 * FB_IMG, *sWl = sXh + 3 * FB_IMG;
#pragma unroll
        for (int kk = 0; kk < FB_K / 16; ++kk) {
            f16x8_t ah[4], al[4], bh[2], bl[2];  // A: W rows (features), B: grid rows (texels)
            const int u = kk * 2 + kh;  // this lane's 16-byte unit of the row: k = 16 kk + 8 kh
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int o = img_off(wn + a * 32 + i, u);
                ah[a] = *reinterpret_cast<const f16x8_t *>(sWh + o);
                al[a] = *reinterpret_cast<const f16x8_t *>(sWl + o);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int o = img_off(wm + b * 32 + i, u);
                bh[b] = *reinterpret_cast<const f16x8_t *>(sXh + o);
                bl[b] = *reinterpret_cast<const f16x8_t *>(sXl + o);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[b], acc[a][b], 0, 0, 0);
            // the next chunk goes into the OTHER buffer (its readers passed the barrier of the previous chunk) between the two
            // k-steps: its split and its 16 LDS stores per thread ride under the second k-step's MFMAs; the chunk after it is
            // requested at once (requested at the top of a chunk instead, the rows had half a chunk to arrive: the fetch registers
            // are re-used, so hipcc cannot issue the loads before the previous values are split)
            if (kk == 0 && more) {
                split_store(cur ^ 1);
                if (k0 + 2 * FB_K < C_LAT) fetch(k0 + 2 * FB_K);
            }
        }
        __syncthreads();
    }
    // D layout: column = lane & 31 -> texel, row (r & 3) + 8 (r >> 2) + 4 kh -> feature feat_of(T, kh, r) = slot 32 T + 16 kh + r
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int f0 = n0 + wn + a * 32;  // first feature of the tile (a multiple of 32: its slots are f0 .. f0 + 31)
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = bias[f0 + (r & 3) + 8 * (r >> 2) + 4 * kh];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long long m = m0 + wm + b * 32 + i;
            if (m < M) {
                float *dst = table + (SPARSE ? (long long)rows[m] : m) * D_HID + f0 + 16 * kh;
#pragma unroll
                for (int q = 0; q < 4; ++q)  // (ordinary stores: non-temporal ones measured 636 -> 980 us on the DTU grid, profiles/r06_fold_notes.md)
                    *reinterpret_cast<f32x4 *>(dst + 4 * q) = f32x4{acc[a][b][4 * q] + bv[4 * q], acc[a][b][4 * q + 1] + bv[4 * q + 1],
                                                                    acc[a][b][4 * q + 2] + bv[4 * q + 2], acc[a][b][4 * q + 3] + bv[4 * q + 3]};
            }
        }
    }
    if (sat && amax >= 65504.f) atomicOr(sat, 1u << 12);
}

// ---- which texels does a training pass read?  (pnr_fold_latent_f32_rows)
// A training step re-folds lin_z every pass (the weights moved), and on a large grid most of that work is for texels no ray of the
// pass comes near: 128 rays x 64 / 96 samples x 3 views of a DTU step touch 37-43 k of the 90 k texels (4 objects: ~150 of 360 k).
// fold_mark_kernel projects every (view, point) with the forward kernels' own code -- geometry_item's rotation and project_point
// (pnr_device.h), the same operations in the same order without contraction: the same bits, hence the same four corner rows -- and
// raises a byte per corner texel; fold_rows_count_kernel / fold_rows_compact_kernel turn the bytes into the ascending list of
// marked rows (4096 texels per workgroup; a workgroup sums the counts of the ones before it).
constexpr int FM_NT = 256, FR_NT = 1024, FR_PER_WG = 4 * FR_NT;
#pragma clang fp contract(off)
__global__ void __launch_bounds__(FM_NT) fold_mark_kernel(const EvalParams q, unsigned char *__restrict__ flags) {
    const long long idx = (long long)blockIdx.x * FM_NT + threadIdx.x;
    // the padding points of the last tile read texel (0, 0) of object 0's views with weight zero (geometry_item: valid == false)
    if (idx < q.NS) flags[(size_t)idx * q.Hl * q.Wl] = 1;
    if (idx >= q.P * q.NS) return;
    const int view = (int)(idx / q.P), g = (int)(idx % q.P);
    const int r = g / q.K;
    const float *ray = q.rays + (size_t)r * 8;
    const float ox = ray[0], oy = ray[1], oz = ray[2], dx = ray[3], dy = ray[4], dz = ray[5];
    const float zz = q.z[g];
    const float X = ox + zz * dx, Y = oy + zz * dy, Z = oz + zz * dz;
    const int obj = r / q.per_obj;
    const float *pose = q.poses + (size_t)(obj * q.NS + view) * 12;
    const float xr0 = pose[0] * X + pose[1] * Y + pose[2] * Z;
    const float xr1 = pose[4] * X + pose[5] * Y + pose[6] * Z;
    const float xr2 = pose[8] * X + pose[9] * Y + pose[10] * Z;
    const Proj pr = project_point(q, pose, obj, view, xr0, xr1, xr2, true);
#pragma unroll
    for (int c = 0; c < 4; ++c) flags[pr.off[c] / (uint32_t)C_LAT] = 1;
}
#pragma clang fp contract(fast)

__device__ __forceinline__ int popcount_bytes(uint32_t v) { return __popc(v & 0x01010101u); }

__global__ void __launch_bounds__(FR_NT) fold_rows_count_kernel(const uint32_t *__restrict__ flags4, int *__restrict__ block_counts) {
    __shared__ int part[FR_NT / 64];
    const int t = threadIdx.x;
    int c = popcount_bytes(flags4[(size_t)blockIdx.x * FR_NT + t]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((t & 63) == 0) part[t >> 6] = c;
    __syncthreads();
    if (t == 0) {
        int s = 0;
        for (int i = 0; i < FR_NT / 64; ++i) s += part[i];
        block_counts[blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(FR_NT) fold_rows_compact_kernel(const uint32_t *__restrict__ flags4, const int *__restrict__ block_counts,
                                                                  long long M, int *__restrict__ rows, int *__restrict__ nrows) {
    __shared__ int part[FR_NT / 64];
    __shared__ int base_s;
    const int t = threadIdx.x, b = blockIdx.x;
    // rows marked in the workgroups before this one
    int before = 0;
    for (int i = t; i < b; i += FR_NT) before += block_counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((t & 63) == 0) part[t >> 6] = before;
    __syncthreads();
    if (t == 0) {
        int s = 0;
        for (int i = 0; i < FR_NT / 64; ++i) s += part[i];
        base_s = s;
        if (b == (int)gridDim.x - 1) *nrows = s + block_counts[b];
    }
    __syncthreads();
    const uint32_t v = flags4[(size_t)b * FR_NT + t] & 0x01010101u;
    const int c = __popc(v);
    int incl = c;  // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += up;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    int off = base_s + incl - c;
    for (int i = 0; i < (t >> 6); ++i) off += part[i];
    const long long first = (long long)b * FR_PER_WG + 4 * t;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((v >> (8 * k)) & 1u) {
            if (first + k < M) rows[off] = (int)(first + k);
            ++off;
        }
}


}  // namespace pnr

namespace pnr {
// 64-bit content fingerprint of a ResnetFC's 30 parameter tensors: sum over all elements of bits(v) * (2 * position + 1)
// (mod 2^64; position = running index over the tensors in PnrMlpWeights order).  ws[0] = workgroups done, ws[1 + b] = partial sum
// of workgroup b.  The last workgroup sums the partials, publishes the sum (out), compares it with *expect when given
// (mismatch -> *flag = 1) and re-zeroes the counter.
constexpr int CK_PER_THREAD = 16;                                  // independent 16-byte loads per thread
constexpr int CK_TOTAL4 = (D_HID * D_IN + D_HID + 3 * (D_HID * C_LAT + D_HID) + 10 * (D_HID * D_HID + D_HID) + D_OUT * D_HID + D_OUT) / 4;
constexpr int CK_BLOCKS = (CK_TOTAL4 + 256 * CK_PER_THREAD - 1) / (256 * CK_PER_THREAD);  // 420
__global__ void __launch_bounds__(256)
params_checksum_kernel(PnrMlpWeights p, unsigned long long *ws, unsigned long long *out, const unsigned long long *expect, int *flag) {
    // The 30 tensors are walked as ONE virtual array of 16-byte chunks (every element count is a multiple of 4; the loads make no
    // alignment assumption beyond 4 bytes: u32x4_param): a thread owns 16 chunks a block-stride apart (210 workgroups: one round on 256 CUs), finds each chunk's tensor in a prefix
    // table and issues its loads back to back -- one memory round trip for the whole 13.75 MB (the first version walked the
    // tensors one after the other inside every thread: 13 dependent round trips, 40 us).
    const float *ptr[30];
    int end4[30];  // exclusive prefix ends, in 16-byte chunks
    {
        int k = 0, run = 0;
        auto add = [&](const float *q, int cnt) { ptr[k] = q; run += cnt >> 2; end4[k] = run; ++k; };
        add(p.lin_in_w, D_HID * D_IN); add(p.lin_in_b, D_HID);
        for (int b = 0; b < 3; ++b) { add(p.lin_z_w[b], D_HID * C_LAT); add(p.lin_z_b[b], D_HID); }
        for (int b = 0; b < 5; ++b) { add(p.fc0_w[b], D_HID * D_HID); add(p.fc0_b[b], D_HID); }
        for (int b = 0; b < 5; ++b) { add(p.fc1_w[b], D_HID * D_HID); add(p.fc1_b[b], D_HID); }
        add(p.lin_out_w, D_OUT * D_HID); add(p.lin_out_b, D_OUT);
    }
    u32x4 x[CK_PER_THREAD];
    int pos[CK_PER_THREAD];
#pragma unroll
    for (int u = 0; u < CK_PER_THREAD; ++u) {
        const int c = (u * CK_BLOCKS + blockIdx.x) * 256 + threadIdx.x;
        pos[u] = c;
        x[u] = u32x4{0u, 0u, 0u, 0u};
        if (c < CK_TOTAL4) {
            const float *base = ptr[0];
            int begin = 0;
#pragma unroll
            for (int k = 0; k < 29; ++k)  // static indexing only: the tables stay in registers
                if (c >= end4[k]) { base = ptr[k + 1]; begin = end4[k]; }
            x[u] = reinterpret_cast<const u32x4_param *>(base)[c - begin];
        }
    }
    unsigned long long acc = 0ull;
#pragma unroll
    for (int u = 0; u < CK_PER_THREAD; ++u) {
        const unsigned long long w = 8ull * (unsigned long long)pos[u] + 1ull;  // odd weight of the chunk's first element
        acc += (unsigned long long)x[u][0] * w + (unsigned long long)x[u][1] * (w + 2ull) + (unsigned long long)x[u][2] * (w + 4ull) +
               (unsigned long long)x[u][3] * (w + 6ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ unsigned long long part[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // one plain store per workgroup + one counter increment; the last workgroup to arrive sums the partials in a fixed
        // order (no serialised same-address atomic adds of the sums themselves)
        ws[1 + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        __threadfence();
        last = atomicAdd(&ws[0], 1ull) == (unsigned long long)(CK_BLOCKS - 1);
    }
    __syncthreads();
    if (last) {
        __threadfence();
        unsigned long long t = 0ull;
        for (int i = threadIdx.x; i < CK_BLOCKS; i += 256) t += ws[1 + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long total = part[0] + part[1] + part[2] + part[3];
            if (out) *out = total;
            if (expect && flag && *expect != total) *flag = 1;
            ws[0] = 0ull;  // the counter is left zeroed for the next call (the partials are overwritten)
        }
    }
}
}  // namespace pnr

// Content fingerprint of a network's parameters, entirely on the device (no host synchronisation): `ws` = pnr_params_checksum_ws_bytes()
// of device scratch whose first 8 bytes are zero (left zeroed), `sum_out` (nullable) receives the fingerprint, and when `expect` is given a differing
// fingerprint raises *mismatch_flag (device int).  pixelnerf_amd uses it to notice parameter writes that bypass both
// tensor._version and torch.optim (`p.data.copy_()`, custom kernels) behind a cached packed stream.
extern "C" size_t pnr_params_checksum_ws_bytes(void) { return (size_t)(1 + pnr::CK_BLOCKS) * sizeof(unsigned long long); }

extern "C" int pnr_params_checksum(const PnrMlpWeights *w, void *ws, unsigned long long *sum_out, const unsigned long long *expect,
                                   int *mismatch_flag, void *stream) {
    if (!w || !ws || (!sum_out && !(expect && mismatch_flag))) return pnr_fail(PNR_E_INVALID, "pnr_params_checksum: bad argument");
    hipLaunchKernelGGL(pnr::params_checksum_kernel, dim3(pnr::CK_BLOCKS), dim3(256), 0, (hipStream_t)stream, *w,
                       (unsigned long long *)ws, sum_out, expect, mismatch_flag);
    return pnr_check_launch("pnr_params_checksum");
}

extern "C" size_t pnr_folded_tables_bytes(const PnrScene *s) {
    if (!s || s->SB <= 0 || s->NS <= 0 || s->Hl <= 0 || s->Wl <= 0) return 0;
    return (size_t)pnr::COMBINE_LAYER * s->SB * s->NS * s->Hl * s->Wl * pnr::D_HID * 2;
}

extern "C" int pnr_fold_latent(const PnrScene *s, const PnrMlpWeights *w, int precision, void *tables, void *stream) {
    using namespace pnr;
    if (!s || !w || !tables || !s->latent_nhwc) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent: null argument");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent: bad scene shape");
    const long long M = (long long)s->SB * s->NS * s->Hl * s->Wl;
    if ((M + 63) / 64 > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent: grid too large");
    dim3 grid((unsigned)((M + 63) / 64), D_HID / 64, COMBINE_LAYER);
    FoldJobs jobs;
    for (int b = 0; b < COMBINE_LAYER; ++b) {
        if (!w->lin_z_w[b] || !w->lin_z_b[b]) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent: null lin_z parameters");
        jobs.W[b] = w->lin_z_w[b]; jobs.bias[b] = w->lin_z_b[b];
    }
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL(fold_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, s->latent_nhwc, jobs, (_Float16 *)tables, M, 65504.0f);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL(fold_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, s->latent_nhwc, jobs, (__bf16 *)tables, M, 3.3895314e38f);
    else
        return pnr_fail(PNR_E_INVALID, "pnr_fold_latent: unknown precision");
    return pnr_check_launch("pnr_fold_latent");
}

// fp32 tables for the split-operand kernel: same layout as the 16-bit tables, 4 bytes per entry
extern "C" size_t pnr_folded_tables_f32_bytes(const PnrScene *s) { return 2 * pnr_folded_tables_bytes(s); }

extern "C" int pnr_fold_latent_f32(const PnrScene *s, const PnrMlpWeights *w, float *tables, void *stream) {
    using namespace pnr;
    if (!s || !w || !tables || !s->latent_nhwc) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32: null argument");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32: bad scene shape");
    const long long M = (long long)s->SB * s->NS * s->Hl * s->Wl;
    if ((M + 63) / 64 > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32: grid too large");
    FoldJobs jobs;
    for (int b = 0; b < COMBINE_LAYER; ++b) {
        if (!w->lin_z_w[b] || !w->lin_z_b[b]) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32: null lin_z parameters");
        jobs.W[b] = w->lin_z_w[b]; jobs.bias[b] = w->lin_z_b[b];
    }
    // large grids: 256 x 256 tiles, double-buffered (fold_split_big_kernel); small ones keep the 128 x 128 kernel, whose 4-wave
    // workgroups fill the chip at a few thousand texels (PNR_FOLD_BIG_MIN_TEXELS: the crossover, measured -- profiles/r06_fold_notes.md)
    static const long long big_min = [] { const char *e = getenv("PNR_FOLD_BIG_MIN_TEXELS"); return e ? atoll(e) : 8192LL; }();
    if (M >= big_min) {
        const int ngroups = (int)((M + FB_TM - 1) / FB_TM);
        const unsigned wgs = (unsigned)((ngroups + 7) / 8) * 8u * 6u;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fold_split_big_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(fold_split_big_kernel)");
        hipLaunchKernelGGL(fold_split_big_kernel<false>, dim3(wgs), dim3(512), FB_LDS, (hipStream_t)stream, s->latent_nhwc, jobs, tables, M, ngroups,
                           saturation_guard_word(), (const int *)nullptr, (const int *)nullptr);
        return pnr_check_launch("pnr_fold_latent_f32");
    }
    dim3 sgrid((unsigned)((M + FS_TM - 1) / FS_TM), D_HID / FS_TN, COMBINE_LAYER);
    hipLaunchKernelGGL(fold_split_kernel, sgrid, dim3(256), 0, (hipStream_t)stream, s->latent_nhwc, jobs, tables, M, saturation_guard_word());
    return pnr_check_launch("pnr_fold_latent_f32");
}

// fp32 tables of the texels ONE training pass reads (rays (R,8), z (R,K): the pass's samples); every other row of `tables` keeps
// what it held.  workspace: marks (one byte per texel, padded to 4096) | per-workgroup counts | row count | row list
static size_t fold_rows_blocks(long long M) { return (size_t)((M + pnr::FR_PER_WG - 1) / pnr::FR_PER_WG); }
extern "C" size_t pnr_fold_latent_f32_rows_workspace_bytes(const PnrScene *s) {
    if (!s || s->SB <= 0 || s->NS <= 0 || s->Hl <= 0 || s->Wl <= 0) return 0;
    const long long M = (long long)s->SB * s->NS * s->Hl * s->Wl;
    const size_t nb = fold_rows_blocks(M);
    return nb * pnr::FR_PER_WG + (nb + 4 + (size_t)M) * sizeof(int);
}

extern "C" int pnr_fold_latent_f32_rows(const PnrScene *s, const PnrMlpWeights *w, const float *rays, const float *z, int R, int rays_per_obj,
                                        int K, float *tables, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace pnr;
    if (!s || !w || !tables || !s->latent_nhwc || !rays || !z) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: null argument");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2 || R <= 0 || K <= 0 || rays_per_obj <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: bad shape");
    if ((long long)rays_per_obj * s->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: R != SB * rays_per_obj");
    const long long M = (long long)s->SB * s->NS * s->Hl * s->Wl, P = (long long)R * K;
    if (M * C_LAT > 0xffffffffLL || P * s->NS > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: grid or pass too large");
    if (!workspace || workspace_bytes < pnr_fold_latent_f32_rows_workspace_bytes(s) || ((uintptr_t)workspace & 15) != 0)
        return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: workspace missing, misaligned (16 bytes) or smaller than "
                                       "pnr_fold_latent_f32_rows_workspace_bytes()");
    FoldJobs jobs;
    for (int b = 0; b < COMBINE_LAYER; ++b) {
        if (!w->lin_z_w[b] || !w->lin_z_b[b]) return pnr_fail(PNR_E_INVALID, "pnr_fold_latent_f32_rows: null lin_z parameters");
        jobs.W[b] = w->lin_z_w[b]; jobs.bias[b] = w->lin_z_b[b];
    }
    const size_t nb = fold_rows_blocks(M);
    unsigned char *flags = reinterpret_cast<unsigned char *>(workspace);
    int *block_counts = reinterpret_cast<int *>(flags + nb * FR_PER_WG);
    int *nrows = block_counts + nb;
    int *rows = nrows + 4;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(flags, 0, nb * FR_PER_WG, st);
    if (e != hipSuccess) return pnr_check_hip(e, "hipMemsetAsync(fold marks)");
    EvalParams q = {};
    q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = P;
    const long long n = P * s->NS;
    hipLaunchKernelGGL(fold_mark_kernel, dim3((unsigned)((n + FM_NT - 1) / FM_NT)), dim3(FM_NT), 0, st, q, flags);
    hipLaunchKernelGGL(fold_rows_count_kernel, dim3((unsigned)nb), dim3(FR_NT), 0, st, reinterpret_cast<const uint32_t *>(flags), block_counts);
    hipLaunchKernelGGL(fold_rows_compact_kernel, dim3((unsigned)nb), dim3(FR_NT), 0, st, reinterpret_cast<const uint32_t *>(flags), block_counts, M,
                       rows, nrows);
    const int ngroups = (int)((M + FB_TM - 1) / FB_TM);
    const unsigned wgs = (unsigned)((ngroups + 7) / 8) * 8u * 6u;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(fold_split_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(fold_split_big_kernel<sparse>)");
    hipLaunchKernelGGL(fold_split_big_kernel<true>, dim3(wgs), dim3(512), FB_LDS, st, s->latent_nhwc, jobs, tables, M, ngroups,
                       saturation_guard_word(), (const int *)rows, (const int *)nrows);
    return pnr_check_launch("pnr_fold_latent_f32_rows");
}

// split-operand stream: [head blob: folded f16 stream | biases | lin_out bias] [tail blob: folded f16 stream of w - f16(w)]
extern "C" int pnr_pack_mlp_split(const PnrMlpWeights *w, void *packed, void *stream) {
    using namespace pnr;
    if (!w || !packed) return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp_split: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)RS_TOTAL_F * IT * (FRAG_ELEMS / 8) * NW;  // one thread per 8 elements
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    constexpr bool OWNK = true;  // the K order pnr_split.hip's stage_own / gemm_split_rot consume
    hipLaunchKernelGGL((pack_weights_kernel<_Float16, true, false, OWNK>), dim3(blocks), dim3(threads), 0, st, *w, (_Float16 *)packed,
                       (_Float16 *)((char *)packed + PACKED_BYTES));  // heads and tails from one pass over the weights
    const int nb = NBIAS * NW * BIAS_FLOATS_PER_WAVE;
    hipLaunchKernelGGL(pack_bias_kernel<true>, dim3((nb + threads - 1) / threads), dim3(threads), 0, st, *w,
                       (float *)((char *)packed + BIAS_OFFSET_BYTES), (float *)((char *)packed + BOUT_OFFSET_BYTES));
    return pnr_check_launch("pnr_pack_mlp_split");
}

extern "C" size_t pnr_packed_mlp_bytes(void) { return pnr::PACKED_BYTES; }

template <bool FOLD>
static int pack_mlp_impl(const PnrMlpWeights *w, int precision, void *packed, void *stream) {
    using namespace pnr;
    if (!w || !packed) return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)(FOLD ? RS_TOTAL_F : RS_TOTAL) * IT * (FRAG_ELEMS / 8) * NW;  // one thread per 8 elements
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL((pack_weights_kernel<_Float16, FOLD>), dim3(blocks), dim3(threads), 0, st, *w, (_Float16 *)packed, (_Float16 *)nullptr);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL((pack_weights_kernel<__bf16, FOLD>), dim3(blocks), dim3(threads), 0, st, *w, (__bf16 *)packed, (__bf16 *)nullptr);
    else
        return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp: unknown precision");
    const int nb = NBIAS * NW * BIAS_FLOATS_PER_WAVE;
    hipLaunchKernelGGL(pack_bias_kernel<FOLD>, dim3((nb + threads - 1) / threads), dim3(threads), 0, st, *w,
                       (float *)((char *)packed + BIAS_OFFSET_BYTES), (float *)((char *)packed + BOUT_OFFSET_BYTES));
    return pnr_check_launch("pnr_pack_mlp");
}

extern "C" int pnr_pack_mlp(const PnrMlpWeights *w, int precision, void *packed, void *stream) {
    return pack_mlp_impl<false>(w, precision, packed, stream);
}

extern "C" int pnr_pack_mlp_folded(const PnrMlpWeights *w, int precision, void *packed, void *stream) {
    return pack_mlp_impl<true>(w, precision, packed, stream);
}

extern "C" int pnr_nchw_to_nhwc(const float *in, float *out, int N, int C, int H, int W, void *stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return pnr_fail(PNR_E_INVALID, "pnr_nchw_to_nhwc: bad argument");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N), block(32, 8);
    hipLaunchKernelGGL(pnr::nchw_to_nhwc_kernel, grid, block, 0, (hipStream_t)stream, in, out, C, HW);
    return pnr_check_launch("pnr_nchw_to_nhwc");
}
