// pnr_f32.hip -- exact-fp32 evaluation of the per-point network (precision PNR_PREC_F32), gfx950.
//
// A validation-grade companion of the fused 16-bit kernel: the same mathematics with every
// operand in fp32.  Unfused on purpose -- features and activations live in HBM, every linear
// layer is one launch of a plain LDS-tiled GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: each
// product rounded once, accumulation = an fmaf chain in k order) -- so that its results track the
// reference's fp32 PyTorch path to rounding level (~1e-5) and the fast path can be cross-checked
// against it on the GPU at full size.  Throughput is bounded by the fp32 MFMA rate (157 TFLOP/s =
// 1/16 of the f16 rate) and by HBM round trips; it is not the product's headline path.
//
//   feat_f32_kernel   models.py:161-215, code.py:30-42, encoder.py:80-109 -> in42 (rows,64), zlat (rows,512)
//   linear_f32_kernel Y = [Y +] [relu](X) W^T + b                          resnetfc.py:147,175-180,55-62,183
//   pool_f32_kernel   mean over source views                              util.py:461-471
//   out_f32_kernel    sigmoid / relu                                      models.py:260-265
// rows are ordered [view][point] like the training dumps.
//
// The file also holds the host side of the fp32-precision TRAINING paths:
//   exact            pnr_eval_ray_samples_f32_train / pnr_mlp_backward_f32 (split_gemm = 0): the chain above with every
//                    activation kept in fp32, linear_f32_kernel in its transposed form + wgrad_f32_kernel
//   GEMM per layer   the same entries with split_gemm = 1: every product on gemm3_kernel (128x128-tile GEMM that splits its
//                    fp32 operands into (head, tail) f16 pairs on the way into LDS, three f16 MFMAs per product)
//   fused (default)  pnr_eval_ray_samples_split_train / pnr_mlp_backward_split: the launch sequence around the fused
//                    split-operand kernels of pnr_split.hip (forward, data-gradient chain) and pnr_bwd.hip (weight gradients)
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_device.h"
#include "pnr_internal.h"
#include "pnr_layout.h"

namespace pnr {

constexpr int FW = 4;  // wavefronts per block in the feature kernel

// one wavefront per (view, point) of the chunk [p0, p0+np)
// SPLIT: the two outputs leave as (head | tail) f16 row sets instead (operands of the split-operand weight-gradient GEMM: in42 /
// zlat then point to the head arrays, the tail arrays follow behind np*NS rows)
template <bool RAYS, bool SPLIT = false>
__global__ void __launch_bounds__(FW * 64)
feat_f32_kernel(const EvalParams q, long long p0, int np, float *__restrict__ in42, float *__restrict__ zlat) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * FW + wv;  // view * np + local point
    if (idx >= (long long)np * q.NS) return;
    const int view = (int)(idx / np);
    const int g = (int)(p0 + idx % np);
    float X, Y, Z, dx, dy, dz;
    int obj;
    if (RAYS) {
        const int r = g / q.K;
        const float *ray = q.rays + (size_t)r * 8;
        const float zz = q.z[g];
        dx = ray[3]; dy = ray[4]; dz = ray[5];
        X = ray[0] + zz * dx; Y = ray[1] + zz * dy; Z = ray[2] + zz * dz;
        obj = r / q.per_obj;
    } else {
        X = q.xyz[(size_t)g * 3 + 0]; Y = q.xyz[(size_t)g * 3 + 1]; Z = q.xyz[(size_t)g * 3 + 2];
        dx = q.viewdirs[(size_t)g * 3 + 0]; dy = q.viewdirs[(size_t)g * 3 + 1]; dz = q.viewdirs[(size_t)g * 3 + 2];
        obj = g / q.per_obj;
    }
    const float *pose = q.poses + (size_t)(obj * q.NS + view) * 12;
    const float xr[3] = {pose[0] * X + pose[1] * Y + pose[2] * Z, pose[4] * X + pose[5] * Y + pose[6] * Z,
                         pose[8] * X + pose[9] * Y + pose[10] * Z};
    // lin_in operand: [x(3), sin(f_k x)(3), sin(f_k x + pi/2)(3) ... , R d (3), 0-pad]
    float v = 0.f;
    if (lane < 3) {
        v = xr[lane];
    } else if (lane < 39) {
        const int t = lane - 3, k = t / 6, rem = t % 6, c = rem % 3;
        const float f = 1.5f * (float)(1 << k);
        const float a = xr[c] * f;
        // code.py:39 addcmul(phases, x, freqs) = one fused multiply-add (see geometry_item)
        v = rem < 3 ? sinf(a) : sinf(__builtin_fmaf(xr[c], f, 1.57079637050628662109375f));
    } else if (lane < 42) {
        const int c = lane - 39;
        v = pose[4 * c + 0] * dx + pose[4 * c + 1] * dy + pose[4 * c + 2] * dz;
    }
    [[maybe_unused]] const size_t rows_all = (size_t)np * q.NS;
    if constexpr (SPLIT) {
        _Float16 *o = reinterpret_cast<_Float16 *>(in42);
        const _Float16 hh = (_Float16)v;
        o[(size_t)idx * D_IN_PAD + lane] = hh;
        o[rows_all * D_IN_PAD + (size_t)idx * D_IN_PAD + lane] = (_Float16)(v - (float)hh);
    } else {
        in42[(size_t)idx * D_IN_PAD + lane] = v;
    }
    // bilinear latent lookup, 8 channels per lane
    const Proj pr = project_point(q, pose, obj, view, xr[0], xr[1], xr[2], true);
    const float *lat = q.latent + lane * 8;
    float *dst = zlat + (size_t)idx * C_LAT + lane * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float *src = lat + pr.off[c];
        const f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 4);
        if (c == 0) { acc0 = a * pr.w[0]; acc1 = b * pr.w[0]; }
        else { acc0 += a * pr.w[c]; acc1 += b * pr.w[c]; }
    }
    if constexpr (SPLIT) {
        _Float16 *o = reinterpret_cast<_Float16 *>(zlat) + (size_t)idx * C_LAT + lane * 8;
        __attribute__((aligned(16))) _Float16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = e < 4 ? acc0[e] : acc1[e - 4];
            hh[e] = (_Float16)x;
            ll[e] = (_Float16)(x - (float)hh[e]);
        }
        *reinterpret_cast<uint4 *>(o) = *reinterpret_cast<const uint4 *>(hh);
        *reinterpret_cast<uint4 *>(o + rows_all * C_LAT) = *reinterpret_cast<const uint4 *>(ll);
    } else {
        *reinterpret_cast<f32x4 *>(dst) = acc0;
        *reinterpret_cast<f32x4 *>(dst + 4) = acc1;
    }
}

// Y (M,N) = [Yin +] mask( [relu](X (M,K)) W'^T + b ).   Block 256 threads = 4 waves, 64x64 tile, K in chunks of 32
// through LDS, v_mfma_f32_32x32x2_f32 (A: X[i][k], B: W'[j][k], one float each).
//   W' element (n, k) = W[n * ldw + k]  (w_trans == 0: an nn.Linear weight (N,K), the forward)
//                     = W[k * ldw + n]  (w_trans == 1: the SAME tensor read transposed -- the backward's dX = dY W)
//   mask (nullable, leading dimension ldy): the product is kept where mask > 0, zeroed elsewhere (relu' of a saved activation)
//   Yin (nullable, leading dimension ldy, may alias Y): residual / accumulation operand, added after the mask
__global__ void __launch_bounds__(256)
linear_f32_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw, int w_trans, const float *__restrict__ b,
                  const float *Yin, float *Y, int ldy, long long M, int N, int K, int relu_in, const float *__restrict__ mask,
                  float out_scale) {
    __shared__ float sX[64][33], sW[64][33];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long m0 = (long long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32;
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        // stage 64x32 of X and of W': 2048 floats each, 8 per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = t + u * 256;
            {
                const int row = e >> 5, col = e & 31;
                float xv = 0.f;
                if (m0 + row < M && k0 + col < K) xv = X[(m0 + row) * ldx + k0 + col];
                sX[row][col] = relu_in ? fmaxf(xv, 0.f) : xv;
            }
            // W': consecutive threads walk the tensor's fast dimension in either orientation
            const int row = w_trans ? (e & 63) : (e >> 5), col = w_trans ? (e >> 6) : (e & 31);
            float wv_ = 0.f;
            if (n0 + row < N && k0 + col < K)
                wv_ = w_trans ? W[(size_t)(k0 + col) * ldw + n0 + row] : W[(size_t)(n0 + row) * ldw + k0 + col];
            sW[row][col] = wv_;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sX[wm + i][2 * kk + kh], sW[wn + i][2 * kk + kh], acc, 0, 0, 0);
        __syncthreads();
    }
    // D layout: column j = lane&31 -> n, row (r&3)+8(r>>2)+4kh -> m
    const int n = n0 + wn + i;
    if (n < N) {
        const float bias = b ? b[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < M) {
                float v = (acc[r] + bias) * out_scale;
                if (mask && !(mask[m * ldy + n] > 0.f)) v = 0.f;
                if (Yin) v += Yin[m * ldy + n];
                Y[m * ldy + n] = v;
            }
        }
    }
}

// Weight gradient of one linear:  dW (N,Kd) = dY^T (N x M) [relu](X) (M x Kd),  db (N) = sum_m dY[m][.]   -- exact fp32
// (v_mfma_f32_32x32x2_f32), the reduction over the M rows split over blockIdx.z with a fixed-order second pass
// (wgrad_reduce_f32_kernel): bit-reproducible, no atomics.  Block = 64 x 64 tile of dW, 4 waves.
__global__ void __launch_bounds__(256)
wgrad_f32_kernel(const float *__restrict__ dY, int lddy, const float *__restrict__ X, int ldx, int relu_x, long long M, int N, int Kd,
                 float *__restrict__ part, float *__restrict__ bpart) {
    __shared__ float sA[32][65], sB[32][65];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int nsplit = gridDim.z, split = blockIdx.z;
    long long per = (M + nsplit - 1) / nsplit;
    per = (per + 31) / 32 * 32;
    const long long r0 = (long long)split * per, r1 = r0 + per < M ? r0 + per : M;
    const int wn = (w >> 1) * 32, wk = (w & 1) * 32;
    const int i = lane & 31, kh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum = 0.f;
    for (long long m0 = r0; m0 < r1; m0 += 32) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = t + u * 256, row = e >> 6, col = e & 63;
            float a = 0.f, x = 0.f;
            if (m0 + row < r1) {
                if (n0 + col < N) a = dY[(m0 + row) * lddy + n0 + col];
                if (k0 + col < Kd) x = X[(m0 + row) * ldx + k0 + col];
            }
            sA[row][col] = a;
            sB[row][col] = relu_x ? fmaxf(x, 0.f) : x;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[2 * kk + kh][wn + i], sB[2 * kk + kh][wk + i], acc, 0, 0, 0);
        if (blockIdx.y == 0 && t < 64) {
#pragma unroll
            for (int r = 0; r < 32; ++r) bsum += sA[r][t];
        }
        __syncthreads();
    }
    // D: column j = lane&31 -> kd, row (r&3)+8(r>>2)+4kh -> n
    float *pz = part + (size_t)split * N * Kd;
    const int kd = k0 + wk + i;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (n < N && kd < Kd) pz[(size_t)n * Kd + kd] = acc[r];
    }
    if (blockIdx.y == 0 && t < 64 && n0 + t < N) bpart[(size_t)split * N + n0 + t] = bsum;
}

__global__ void wgrad_reduce_f32_kernel(const float *__restrict__ part, const float *__restrict__ bpart, int nsplit, int N, int Kd,
                                        float *__restrict__ dW, float *__restrict__ db, const float *__restrict__ scale_dev) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const float sc = scale_dev ? *scale_dev : 1.f;
    if (idx < N * Kd) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * N * Kd + idx];
        dW[idx] = s * sc;
    }
    if (db && idx < N) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += bpart[(size_t)z * N + idx];
        db[idx] = s * sc;
    }
}

// ---------------------------------------------------------------- fp32-CLASS GEMM on the f16 matrix cores (split operands)
// C[m][n] (+)= sum_k A(m,k) B(n,k) with fp32 operands in HBM, carried through the MFMA as (head, tail) fp16 pairs:
// a b ~= ah bh + ah bl + al bh (fp32 accumulate; the dropped al bl is 2^-22 of the product) -- the arithmetic of
// pnr_split.hip as a plain tiled GEMM, ~6x the rate of the fp32-MFMA kernels above.  It serves all three products of the
// fp32-class TRAINING path (precision "f16x3" under autograd):
//     forward   Y  = relu(X) W^T + b        A = X (k contiguous),      B = W (k contiguous)
//     data grad dX = (dY W) . relu'         A = dY (k contiguous),     B(n,k) = W[k][n]      (row index contiguous)
//     weights   dW = dY^T relu(X)           A(n,m) = dY[m][n], B(kd,m) = X[m][kd]            (both row-index contiguous;
//                                           the reduction over the rows m split over blockIdx.z, fixed-order second pass)
// Operands are split on their way into LDS ([row][k] f16 images, 80-byte rows: conflict-free b128 fragment reads); a device
// scalar can scale A on load (gradients run at a power-of-two scale that keeps them in the fp16 range) and the result on its way
// out.  128 x 128 block tile, 4 waves of 64 x 64, K chunks of 32; one LDS stage + register prefetch of the next chunk, three
// workgroups per CU cover each other's load / convert phases.
struct Gemm3 {
    const float *A, *B;
    long long a_rs, a_ks, b_rs, b_ks;  // element (row, k) at base[row * rs + k * ks]; one of (rs, ks) is 1
    long long M;                       // rows of A / C
    int N;                             // rows of B = columns of C
    long long K;                       // reduction length
    int relu_a, relu_b;
    const float *a_scale;              // device scalar multiplied into A on load (nullable)
    const float *bias, *mask, *Yin;    // epilogue: + bias[n]; keep where mask[m][n] > 0; + Yin[m][n]   (leading dimension ldc)
    float *C;
    int ldc;
    const float *out_scale;            // device scalar multiplied into the product (nullable)
    float *part, *bpart;               // split-K: partial products part[z][M][N] (raw) and column sums of A bpart[z][M] (nullable)
    long long k_per_split;
    int tiles_m, tiles_n, nz;          // 128 x 128 tiles of C and K slices; the 1-D grid is decoded XCD-aware (see the kernel)
};

constexpr int G3_ROW = 80;                 // bytes per [row][32 k] f16 row (64 + 16 pad)
constexpr int G3_IMG = 128 * G3_ROW;       // one 128-row image
// split 4 fp32 values into head / tail f16 (8 bytes each), MODE.FP16_OVFL set by the caller
__device__ __forceinline__ void g3_split4(f32x4 v, bool relu, float sc, uint2 &hi, uint2 &lo) {
    uint32_t h[2], l[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        f32x2 p = {v[2 * k] * sc, v[2 * k + 1] * sc};
        if (relu) { p[0] = fmaxf(p[0], 0.f); p[1] = fmaxf(p[1], 0.f); }
        const f16x2 hh = __builtin_convertvector(p, f16x2);
        h[k] = __builtin_bit_cast(uint32_t, hh);
        uint32_t t;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t) : "v"(h[k]), "v"(p[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(t) : "v"(h[k]), "v"(p[1]));
        l[k] = t;
    }
    hi = make_uint2(h[0], h[1]);
    lo = make_uint2(l[0], l[1]);
}

// STAGES = 2 (weight gradients: long reductions, 1536 rows per slice): two LDS stages, one barrier per chunk, two workgroups
// per CU.  STAGES = 1 (forward / data gradients: K = 512, sixteen chunks, epilogue-heavy): one stage + one chunk of register
// lookahead, 40 KiB and <= 170 registers so that THREE workgroups per CU cover each other (measured: 122 / 179 us per
// 49152-row GEMM against 141 / 211 us in the two-stage form; the weight gradient 102 against 214 us).
template <bool A_KC, bool B_KC, int STAGES>
__global__ void __launch_bounds__(256, STAGES == 2 ? 2 : 3) gemm3_kernel(const Gemm3 g) {
    // two LDS stages of (A head, A tail, B head, B tail): chunk k+1 is split into the other stage while chunk k is multiplied,
    // and the global loads of chunk k+2 are issued before that -- every load has a whole iteration to arrive (the round-3
    // first version, one stage + one chunk of lookahead, spent 50-75 % of its wave cycles parked on s_waitcnt / barriers)
    __shared__ __attribute__((aligned(16))) char lds_all[STAGES * 4 * G3_IMG];
    typedef Prec<PNR_PREC_F16> PH;
    typedef PH::T8 h8;
    f16_ovfl_mode<PH>();
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // XCD-aware placement (workgroups are dealt to the 8 XCDs round-robin by linear id; each XCD has its own L2).  The blocks
    // that read the SAME operand rows -- the column tiles of one row tile (A rows re-read per column tile), and all 16 tiles of
    // one K slice of a weight gradient (both operands) -- get consecutive slots of ONE XCD, so an fp32 operand row (2 KiB)
    // crosses the fabric once instead of once per tile: these GEMMs are HBM-bound (100 MB per activation tensor).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int tm, tn, bz;
    if (g.nz == 1) {
        tn = slot % g.tiles_n; tm = (slot / g.tiles_n) * 8 + xcd; bz = 0;
        if (tm >= g.tiles_m) return;
    } else {
        const int T = g.tiles_m * g.tiles_n, tile = slot % T;
        bz = (slot / T) * 8 + xcd; tm = tile / g.tiles_n; tn = tile % g.tiles_n;
        if (bz >= g.nz) return;
    }
    const long long m0 = (long long)tm * 128;
    const int n0 = tn * 128;
    const long long k_begin = (long long)bz * g.k_per_split;
    const long long k_end = k_begin + g.k_per_split < g.K ? k_begin + g.k_per_split : g.K;
    const float asc = g.a_scale ? *g.a_scale : 1.f;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    const int fi = lane & 31, fh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging maps.  k-contiguous operand: thread -> (row = t/8 + 32u, k = 4 (t%8) .. +3), one 16-byte load per u.
    // row-contiguous operand: thread -> (rows 4 (t/8) .. +3, k = 4 (t%8) + u): four 16-byte loads along the rows, transposed
    // in registers so that every LDS write is again 4 consecutive k of one row (8 bytes of heads + 8 of tails).
    f32x4 ra[4], rb[4];
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};  // split-K + bpart: sums over k of this thread's four A rows
    const int kq = t & 7, rq = t >> 3;
    auto load = [&](long long kc) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (A_KC) {
                const long long m = m0 + rq + 32 * u, k = kc + 4 * kq;
                ra[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < g.M) {
                    const float *src = g.A + m * g.a_rs + k;
                    if (k + 3 < k_end && (g.a_rs & 3) == 0) ra[u] = *reinterpret_cast<const f32x4_param *>(src);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (k + e < k_end) ra[u][e] = src[e];
                }
            } else {
                const long long m = m0 + 4 * rq, k = kc + 4 * kq + u;
                ra[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < k_end) {
                    const float *src = g.A + k * g.a_ks + m;
                    if (m + 3 < g.M && (g.a_ks & 3) == 0) ra[u] = *reinterpret_cast<const f32x4_param *>(src);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (m + e < g.M) ra[u][e] = src[e];
                }
            }
            if (B_KC) {
                const int n = n0 + rq + 32 * u;
                const long long k = kc + 4 * kq;
                rb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < g.N) {
                    const float *src = g.B + (long long)n * g.b_rs + k;
                    if (k + 3 < k_end && (g.b_rs & 3) == 0) rb[u] = *reinterpret_cast<const f32x4_param *>(src);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (k + e < k_end) rb[u][e] = src[e];
                }
            } else {
                const int n = n0 + 4 * rq;
                const long long k = kc + 4 * kq + u;
                rb[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < k_end) {
                    const float *src = g.B + k * g.b_ks + n;
                    if (n + 3 < g.N && (g.b_ks & 3) == 0) rb[u] = *reinterpret_cast<const f32x4_param *>(src);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (n + e < g.N) rb[u][e] = src[e];
                }
            }
        }
    };
    auto stage = [&](char *lds) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint2 hi, lo;
            if (A_KC) {
                g3_split4(ra[u], g.relu_a != 0, asc, hi, lo);
                const int off = (rq + 32 * u) * G3_ROW + kq * 8;
                *reinterpret_cast<uint2 *>(lds + off) = hi;
                *reinterpret_cast<uint2 *>(lds + G3_IMG + off) = lo;
            } else {
                const f32x4 col = {ra[0][u], ra[1][u], ra[2][u], ra[3][u]};  // row 4 rq + u, k = 4 kq .. +3
                g3_split4(col, g.relu_a != 0, asc, hi, lo);
                const int off = (4 * rq + u) * G3_ROW + kq * 8;
                *reinterpret_cast<uint2 *>(lds + off) = hi;
                *reinterpret_cast<uint2 *>(lds + G3_IMG + off) = lo;
                if (g.bpart) colsum[u] += (col[0] + col[1]) + (col[2] + col[3]);
            }
            if (B_KC) {
                g3_split4(rb[u], g.relu_b != 0, 1.f, hi, lo);
                const int off = (rq + 32 * u) * G3_ROW + kq * 8;
                *reinterpret_cast<uint2 *>(lds + 2 * G3_IMG + off) = hi;
                *reinterpret_cast<uint2 *>(lds + 3 * G3_IMG + off) = lo;
            } else {
                const f32x4 col = {rb[0][u], rb[1][u], rb[2][u], rb[3][u]};
                g3_split4(col, g.relu_b != 0, 1.f, hi, lo);
                const int off = (4 * rq + u) * G3_ROW + kq * 8;
                *reinterpret_cast<uint2 *>(lds + 2 * G3_IMG + off) = hi;
                *reinterpret_cast<uint2 *>(lds + 3 * G3_IMG + off) = lo;
            }
        }
    };
    if constexpr (STAGES == 1) {
        if (k_begin < k_end) load(k_begin);
        for (long long kc = k_begin; kc < k_end; kc += 32) {
            __syncthreads();  // the previous chunk's fragment reads are done
            stage(lds_all);
            __syncthreads();
            if (kc + 32 < k_end) load(kc + 32);  // in flight under this chunk's MFMAs
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = (wm + i * 32 + fi) * G3_ROW + ks * 32 + fh * 16;
                    ah[i] = *reinterpret_cast<const h8 *>(lds_all + off);
                    al[i] = *reinterpret_cast<const h8 *>(lds_all + G3_IMG + off);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = (wn + j * 32 + fi) * G3_ROW + ks * 32 + fh * 16;
                    bh[j] = *reinterpret_cast<const h8 *>(lds_all + 2 * G3_IMG + off);
                    bl[j] = *reinterpret_cast<const h8 *>(lds_all + 3 * G3_IMG + off);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
    } else {
    // prologue: chunk 0 staged, chunk 1 in flight
    if (k_begin < k_end) {
        load(k_begin);
        stage(lds_all);
        if (k_begin + 32 < k_end) load(k_begin + 32);
    }
    __syncthreads();
    int buf = 0;
    for (long long kc = k_begin; kc < k_end; kc += 32, buf ^= 1) {
        const char *lds = lds_all + buf * (4 * G3_IMG);
        // fragments of this chunk first (LDS), then the split of chunk k+1 into the other stage (its loads were issued one
        // iteration ago), then the loads of chunk k+2 -- all before / under this chunk's 24 MFMAs
        h8 ah[2][2], al[2][2], bh[2][2], bl[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm + i * 32 + fi) * G3_ROW + ks * 32 + fh * 16;
                ah[ks][i] = *reinterpret_cast<const h8 *>(lds + off);
                al[ks][i] = *reinterpret_cast<const h8 *>(lds + G3_IMG + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (wn + j * 32 + fi) * G3_ROW + ks * 32 + fh * 16;
                bh[ks][j] = *reinterpret_cast<const h8 *>(lds + 2 * G3_IMG + off);
                bl[ks][j] = *reinterpret_cast<const h8 *>(lds + 3 * G3_IMG + off);
            }
        }
        if (kc + 32 < k_end) {
            stage(lds_all + (buf ^ 1) * (4 * G3_IMG));
            if (kc + 64 < k_end) load(kc + 64);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // D[i][j] += sum_k A-frag(row i, k) B-frag(col j, k): the MFMA's D holds column j = lane&31 of the B rows
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();  // stage k+1 is complete for every wave; stage k may be overwritten next iteration
    }
    }
    // epilogue.  D register r of tile (i, j): row m = wm + 32 i + (r&3) + 8 (r>>2) + 4 fh, column n = wn + 32 j + fi
    const float osc = g.out_scale ? *g.out_scale : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 32 + fi;
            if (n >= g.N) continue;
            const float bias = (g.bias && !g.part) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= g.M) continue;
                if (g.part) {
                    g.part[((size_t)bz * g.M + m) * g.N + n] = acc[i][j][r];
                } else {
                    float v = acc[i][j][r] * osc + bias;
                    const size_t o = (size_t)m * g.ldc + n;
                    if (g.mask && !(g.mask[o] > 0.f)) v = 0.f;
                    if (g.Yin) v += g.Yin[o];
                    g.C[o] = v;
                }
            }
        }
    if (!A_KC && g.bpart && tn == 0) {
        // column sums of this block's A rows over its k range: 8 threads (kq) per row quad, reduced through LDS
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds_all);
#pragma unroll
        for (int u = 0; u < 4; ++u) red[(4 * rq + u) * 8 + kq] = colsum[u];
        __syncthreads();
        if (t < 128 && m0 + t < g.M) {
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += red[t * 8 + e];
            g.bpart[(size_t)bz * g.M + m0 + t] = sum * asc;  // same (scaled) domain as the partial products
        }
    }
}


// Gv[v*np + p][f] = Gp[p][f] * inv   (backward of the view mean, util.py:461-466)
__global__ void unpool_f32_kernel(const float *__restrict__ gp, float *__restrict__ gv, long long np, int NS, float inv) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np * D_HID) return;
    const float v = gp[idx] * inv;
    for (int s = 0; s < NS; ++s) gv[(size_t)s * np * D_HID + idx] = v;
}

// xp[p][f] = mean_v x[v*np + p][f]
__global__ void pool_f32_kernel(const float *__restrict__ x, float *__restrict__ xp, int np, int NS, int cmax) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)np * D_HID) return;
    float s = x[idx];
    if (cmax) {  // util.py:467-468
        for (int v = 1; v < NS; ++v) s = fmaxf(s, x[(size_t)v * np * D_HID + idx]);
        xp[idx] = s;
        return;
    }
    for (int v = 1; v < NS; ++v) s += x[(size_t)v * np * D_HID + idx];
    xp[idx] = s / (float)NS;
}

// xp[(o*B + p)][f] = mean_v x[((o*NS + v)*B + p)][f] : util.combine_interleaved with inner dims (NS, B) (util.py:461-471)
__global__ void pool_interleaved_f32_kernel(const float *__restrict__ x, float *__restrict__ xp, long long groups, int NS, int B, int cmax) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups * B * D_HID) return;
    const long long o = idx / ((long long)B * D_HID), rem = idx - o * (long long)B * D_HID;
    float s = x[((size_t)o * NS) * (size_t)B * D_HID + rem];
    if (cmax) {
        for (int v = 1; v < NS; ++v) s = fmaxf(s, x[((size_t)o * NS + v) * (size_t)B * D_HID + rem]);
        xp[idx] = s;
        return;
    }
    for (int v = 1; v < NS; ++v) s += x[((size_t)o * NS + v) * (size_t)B * D_HID + rem];
    xp[idx] = s / (float)NS;
}

__global__ void out_f32_kernel(const float *o, float *rgbs, int np) {  // o may alias rgbs (training forward)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np) return;
    const f32x4 s = *reinterpret_cast<const f32x4 *>(o + (size_t)idx * 4);
    f32x4 r = {1.f / (1.f + expf(-s[0])), 1.f / (1.f + expf(-s[1])), 1.f / (1.f + expf(-s[2])), fmaxf(s[3], 0.f)};
    *reinterpret_cast<f32x4 *>(rgbs + (size_t)idx * 4) = r;
}

// how the fp32-precision products are formed: exact fp32 MFMA (validation grade) or split f16 operands (fp32-class, fast);
// gs / gi: device scalars [s], [1/s] -- the power-of-two scale the gradients of a fast backward run at (NULL: 1)
struct Mm {
    hipStream_t st;
    bool fast;
    const float *gs, *gi;
};

template <bool AK, bool BK> static void launch_g3(const Mm &c, Gemm3 g, int nz) {
    g.tiles_m = (int)((g.M + 127) / 128); g.tiles_n = (g.N + 127) / 128; g.nz = nz;
    const long long groups = nz == 1 ? (g.tiles_m + 7) / 8 : (nz + 7) / 8;  // per XCD: row tiles (x column tiles) or K slices (x all tiles)
    const long long per = nz == 1 ? g.tiles_n : (long long)g.tiles_m * g.tiles_n;
    hipLaunchKernelGGL((gemm3_kernel<AK, BK, (AK ? 1 : 2)>), dim3((unsigned)(groups * per * 8)), dim3(256), 0, c.st, g);
}

static void linear(const Mm &c, const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, long long M,
                   int N, int K, bool relu_in, bool accumulate, const float *Yin = nullptr) {
    if (c.fast) {
        Gemm3 g = {};
        g.A = X; g.a_rs = ldx; g.a_ks = 1; g.B = W; g.b_rs = K; g.b_ks = 1; g.M = M; g.N = N; g.K = K; g.relu_a = relu_in;
        g.bias = b; g.Yin = accumulate ? (Yin ? Yin : Y) : nullptr; g.C = Y; g.ldc = ldy; g.k_per_split = K;
        launch_g3<true, true>(c, g, 1);
        return;
    }
    dim3 grid((unsigned)((M + 63) / 64), (N + 63) / 64);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, c.st, X, ldx, W, K, 0, b, accumulate ? (Yin ? Yin : Y) : nullptr, Y, ldy, M,
                       N, K, relu_in ? 1 : 0, (const float *)nullptr, 1.f);
}
static void linear(hipStream_t st, const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, long long M,
                   int N, int K, bool relu_in, bool accumulate, const float *Yin = nullptr) {
    linear(Mm{st, false, nullptr, nullptr}, X, ldx, W, b, Y, ldy, M, N, K, relu_in, accumulate, Yin);
}

// backward data product: Out (M,J) = [Out +] (dY (M,N) W (N,J)) [* out scale] . [mask > 0]   (W = the nn.Linear weight (N,J) as stored)
// in_scaled: dY is still the caller's unscaled gradient (scaled on load in the fast form); out_unscale: the result leaves the
// scaled domain (d z_lat, d(code)); a_scale / out_scale only act in the fast form
static void linear_bwd(const Mm &c, const float *dY, int lddy, const float *W, int J, int N, float *Out, int ldo, long long M,
                       const float *mask, bool accumulate, bool in_unscaled = false, bool out_unscale = false) {
    if (c.fast) {
        Gemm3 g = {};
        g.A = dY; g.a_rs = lddy; g.a_ks = 1; g.B = W; g.b_rs = 1; g.b_ks = J; g.M = M; g.N = J; g.K = N;
        g.a_scale = in_unscaled ? c.gs : nullptr; g.out_scale = out_unscale ? c.gi : nullptr;
        g.mask = mask; g.Yin = accumulate ? Out : nullptr; g.C = Out; g.ldc = ldo; g.k_per_split = N;
        launch_g3<true, false>(c, g, 1);
        return;
    }
    dim3 grid((unsigned)((M + 63) / 64), (J + 63) / 64);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, c.st, dY, lddy, W, J, 1, (const float *)nullptr,
                       accumulate ? Out : (const float *)nullptr, Out, ldo, M, J, N, 0, mask, 1.f);
}

constexpr int WG_SPLIT = 32;  // row slices of the weight-gradient reduction

// dW (N,Kd), db (N) from dY (M,N) and X (M,Kd); part / bpart: WG_SPLIT * (N*Kd + N) floats of workspace
// lin_out's gradient (resnetfc.py:183; 4 x 512: dW[o][k] = sum_r g[r][o] relu(x5[r][k]), db[o] = sum_r g[r][o]) is a
// REDUCTION over the points, not a GEMM worth the matrix cores: 200 MFLOP over 100 MB of fp32 rows at config-5 size.  Round 3
// ran it on the split-operand GEMM kernel (82 us per pass at 17.6 % MFMA busy: a 4-row A operand in a 128-row tile); here
// every thread owns two features and walks a row slice with plain fp32 FMAs (exact products, one rounding per add), row
// slices -> partials -> fixed-order reduce, like lin_out_grad_kernel of the 16-bit path: HBM-bound, ~25-30 us.
constexpr int LOF_BLOCKS = 256;
__global__ void __launch_bounds__(256)
lin_out_grad_f32_kernel(const float *__restrict__ g, const float *__restrict__ x5, long long P, float *__restrict__ part) {
    const int t = threadIdx.x;
    const long long per = (P + LOF_BLOCKS - 1) / LOF_BLOCKS;
    const long long r0 = (long long)blockIdx.x * per, r1 = r0 + per < P ? r0 + per : P;
    float acc[4][2] = {}, bs[4] = {};
#pragma unroll 8  // 8 rows of loads in flight
    for (long long r = r0; r < r1; ++r) {
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + r * 4);
        const f32x2 xv = *reinterpret_cast<const f32x2 *>(x5 + r * D_HID + 2 * t);
        const float xa = fmaxf(xv[0], 0.f), xb = fmaxf(xv[1], 0.f);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            acc[o][0] = __builtin_fmaf(gv[o], xa, acc[o][0]);
            acc[o][1] = __builtin_fmaf(gv[o], xb, acc[o][1]);
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
__global__ void lin_out_reduce_f32_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * D_HID + 4) return;
    float s = 0.f;
    for (int z = 0; z < LOF_BLOCKS; ++z) s += part[(size_t)z * (4 * D_HID + 4) + idx];
    if (idx < 4 * D_HID) dW[idx] = s;
    else if (db) db[idx - 4 * D_HID] = s;
}
// g (P,4) unscaled fp32, x5 (P,512) fp32 rows in natural feature order; part: >= LOF_BLOCKS * 2052 floats
static void lin_out_grad_f32(hipStream_t st, const float *g, const float *x5, long long P, float *dW, float *db, float *part) {
    hipLaunchKernelGGL(lin_out_grad_f32_kernel, dim3(LOF_BLOCKS), dim3(256), 0, st, g, x5, P, part);
    hipLaunchKernelGGL(lin_out_reduce_f32_kernel, dim3((4 * D_HID + 4 + 255) / 256), dim3(256), 0, st, part, dW, db);
}

static void wgrad(const Mm &c, const float *dY, int lddy, const float *X, int ldx, bool relu_x, long long M, int N, int Kd,
                  float *dW, float *db, float *part, bool in_unscaled = false) {
    float *bpart = part + (size_t)WG_SPLIT * N * Kd;
    if (c.fast) {
        Gemm3 g = {};
        g.A = dY; g.a_rs = 1; g.a_ks = lddy; g.B = X; g.b_rs = 1; g.b_ks = ldx; g.M = N; g.N = Kd; g.K = M; g.relu_b = relu_x;
        g.a_scale = in_unscaled ? c.gs : nullptr; g.part = part; g.bpart = bpart;
        long long per = (M + WG_SPLIT - 1) / WG_SPLIT;
        g.k_per_split = (per + 31) / 32 * 32;
        launch_g3<false, false>(c, g, WG_SPLIT);
        hipLaunchKernelGGL(wgrad_reduce_f32_kernel, dim3((N * Kd + 255) / 256), dim3(256), 0, c.st, part, bpart, WG_SPLIT, N, Kd, dW, db, c.gi);
        return;
    }
    dim3 grid((N + 63) / 64, (Kd + 63) / 64, WG_SPLIT);
    hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(256), 0, c.st, dY, lddy, X, ldx, relu_x ? 1 : 0, M, N, Kd, part, bpart);
    hipLaunchKernelGGL(wgrad_reduce_f32_kernel, dim3((N * Kd + 255) / 256), dim3(256), 0, c.st, part, bpart, WG_SPLIT, N, Kd, dW, db,
                       (const float *)nullptr);
}

// floats of workspace per point of a chunk
static size_t floats_per_point(int NS) { return (size_t)NS * (D_IN_PAD + C_LAT + D_HID + D_HID) + D_HID + D_HID + 4; }

static int eval_f32(const PnrScene *s, const PnrMlpWeights *w, EvalParams q, bool rays, float *ws, size_t ws_bytes,
                    hipStream_t st) {
    if (!s || !w || !ws || !q.out) return pnr_fail(PNR_E_INVALID, "pnr_eval_f32: null argument");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_eval_f32: bad scene shape");
    if (q.P == 0) return PNR_OK;
    if (q.P > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_eval_f32: too many points");
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    const int NS = s->NS;
    long long chunk = (long long)(ws_bytes / sizeof(float) / floats_per_point(NS));
    if (chunk < 64) return pnr_fail(PNR_E_INVALID, "pnr_eval_f32: workspace too small");
    if (chunk >= q.P) chunk = q.P;
    else chunk = chunk / 64 * 64;
    for (long long p0 = 0; p0 < q.P; p0 += chunk) {
        const int np = (int)((q.P - p0) < chunk ? (q.P - p0) : chunk);
        const long long rows = (long long)np * NS;
        float *in42 = ws;
        float *zlat = in42 + rows * D_IN_PAD;
        float *x = zlat + rows * C_LAT;
        float *net = x + rows * D_HID;
        float *xp = net + rows * D_HID;      // pooled stream (np,512)
        float *netp = xp + (size_t)np * D_HID;
        float *o4 = netp + (size_t)np * D_HID;
        const unsigned fb = (unsigned)((rows + FW - 1) / FW);
        if (rays) hipLaunchKernelGGL(feat_f32_kernel<true>, dim3(fb), dim3(FW * 64), 0, st, q, p0, np, in42, zlat);
        else hipLaunchKernelGGL(feat_f32_kernel<false>, dim3(fb), dim3(FW * 64), 0, st, q, p0, np, in42, zlat);
        linear(st, in42, D_IN_PAD, w->lin_in_w, w->lin_in_b, x, D_HID, rows, D_HID, D_IN, false, false);  // resnetfc.py:147
        for (int b = 0; b < COMBINE_LAYER; ++b) {
            linear(st, zlat, C_LAT, w->lin_z_w[b], w->lin_z_b[b], x, D_HID, rows, D_HID, C_LAT, false, true);   // :175-180
            linear(st, x, D_HID, w->fc0_w[b], w->fc0_b[b], net, D_HID, rows, D_HID, D_HID, true, false);          // :55-57
            linear(st, net, D_HID, w->fc1_w[b], w->fc1_b[b], x, D_HID, rows, D_HID, D_HID, true, true);           // :58-62
        }
        const float *xs = x;
        if (NS > 1) {  // util.combine_interleaved
            const long long n = (long long)np * D_HID;
            hipLaunchKernelGGL(pool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xp, np, NS, w->combine_max ? 1 : 0);
            xs = xp;
        }
        float *xw = (NS > 1) ? xp : x;
        for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b) {
            linear(st, xs, D_HID, w->fc0_w[b], w->fc0_b[b], (NS > 1) ? netp : net, D_HID, np, D_HID, D_HID, true, false);
            linear(st, (NS > 1) ? netp : net, D_HID, w->fc1_w[b], w->fc1_b[b], xw, D_HID, np, D_HID, D_HID, true, true);
        }
        linear(st, xw, D_HID, w->lin_out_w, w->lin_out_b, o4, 4, np, D_OUT, D_HID, true, false);  // :183
        hipLaunchKernelGGL(out_f32_kernel, dim3((np + 255) / 256), dim3(256), 0, st, o4, q.out + (size_t)p0 * 4, np);
    }
    return pnr_check_launch("pnr_eval_f32");
}

// ---------------------------------------------------------------- exact-fp32 training path (validation grade)
// The same unfused fp32 chain with every activation the backward needs KEPT (PnrF32Saved), and its backward: data gradients
// dX = (dY W) . relu', weight gradients dW = dY^T relu(X), all on v_mfma_f32_32x32x2_f32 -- the fp32 reference of the
// 16-bit training kernels (pnr_bwd.hip), held to 1e-3 against the reference's own autograd (tests/test_hip_backward_f32.py).
static int check_saved(const PnrF32Saved *sv, int NS) {
    if (!sv || !sv->in42 || !sv->zlat || !sv->x5) return 0;
    for (int b = 0; b < 5; ++b)
        if (!sv->xin[b] || !sv->net[b]) return 0;
    return NS == 1 || sv->pool_in != nullptr;
}

static int eval_f32_train(const PnrScene *s, const PnrMlpWeights *w, EvalParams q, const PnrF32Saved *sv, hipStream_t stream, bool fast) {
    const Mm st = {stream, fast, nullptr, nullptr};
    if (!s || !w || !q.out) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: null argument");
    if (w->combine_max && s->NS > 1) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: combine_type \"max\" is an inference form (no backward)");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: bad scene shape");
    if (!check_saved(sv, s->NS)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: null activation buffer in PnrF32Saved");
    if (q.P == 0) return PNR_OK;
    if (q.P * s->NS > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: too many points");
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    const int NS = s->NS;
    const int np = (int)q.P;
    const long long rows = (long long)np * NS;
    hipLaunchKernelGGL(feat_f32_kernel<true>, dim3((unsigned)((rows + FW - 1) / FW)), dim3(FW * 64), 0, stream, q, 0LL, np, sv->in42, sv->zlat);
    linear(st, sv->in42, D_IN_PAD, w->lin_in_w, w->lin_in_b, sv->xin[0], D_HID, rows, D_HID, D_IN, false, false);  // resnetfc.py:147
    for (int b = 0; b < COMBINE_LAYER; ++b) {
        linear(st, sv->zlat, C_LAT, w->lin_z_w[b], w->lin_z_b[b], sv->xin[b], D_HID, rows, D_HID, C_LAT, false, true);       // :175-180
        linear(st, sv->xin[b], D_HID, w->fc0_w[b], w->fc0_b[b], sv->net[b], D_HID, rows, D_HID, D_HID, true, false);         // :55-57
        float *next = b + 1 < COMBINE_LAYER ? sv->xin[b + 1] : (NS > 1 ? sv->pool_in : sv->xin[COMBINE_LAYER]);
        linear(st, sv->net[b], D_HID, w->fc1_w[b], w->fc1_b[b], next, D_HID, rows, D_HID, D_HID, true, true, sv->xin[b]);    // :58-62
    }
    if (NS > 1) {  // util.combine_interleaved
        const long long n = (long long)np * D_HID;
        hipLaunchKernelGGL(pool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, sv->pool_in, sv->xin[COMBINE_LAYER], np, NS, 0);
    }
    for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b) {
        linear(st, sv->xin[b], D_HID, w->fc0_w[b], w->fc0_b[b], sv->net[b], D_HID, np, D_HID, D_HID, true, false);
        float *next = b + 1 < N_BLOCKS ? sv->xin[b + 1] : sv->x5;
        linear(st, sv->net[b], D_HID, w->fc1_w[b], w->fc1_b[b], next, D_HID, np, D_HID, D_HID, true, true, sv->xin[b]);
    }
    // lin_out's raw output goes through the caller's rgbsigma buffer in place (4 floats per point either way)
    linear(st, sv->x5, D_HID, w->lin_out_w, w->lin_out_b, q.out, 4, np, D_OUT, D_HID, true, false);  // :183
    hipLaunchKernelGGL(out_f32_kernel, dim3((np + 255) / 256), dim3(256), 0, stream, q.out, q.out, np);
    return pnr_check_launch("pnr_eval_ray_samples_f32_train");
}

// reverse of one residual block (resnetfc.py:55-62) on `rows` rows:  G = dL/d(xin + fc_1(relu(fc_0(relu(xin)))))
//   dW1 = G^T relu(net), db1 = sum G;  T = (G W1) . [net > 0];  dW0 = T^T relu(xin), db0 = sum T;  G += (T W0) . [xin > 0]
static void block_bwd_f32(const Mm &st, const PnrMlpWeights *w, const PnrMlpWeights *g, int b, float *G, float *T, const float *xin,
                          const float *net, long long rows, float *part) {
    wgrad(st, G, D_HID, net, D_HID, true, rows, D_HID, D_HID, (float *)g->fc1_w[b], (float *)g->fc1_b[b], part);
    linear_bwd(st, G, D_HID, w->fc1_w[b], D_HID, D_HID, T, D_HID, rows, net, false);
    wgrad(st, T, D_HID, xin, D_HID, true, rows, D_HID, D_HID, (float *)g->fc0_w[b], (float *)g->fc0_b[b], part);
    linear_bwd(st, T, D_HID, w->fc0_w[b], D_HID, D_HID, G, D_HID, rows, xin, true);
}

}  // namespace pnr

extern "C" int pnr_eval_ray_samples_f32_train(const PnrScene *scene, const PnrMlpWeights *w, const float *rays, const float *z, int R,
                                              int rays_per_obj, int K, float *rgbsigma, const PnrF32Saved *saved, int split_gemm,
                                              void *stream) {
    if (R <= 0 || K <= 0 || rays_per_obj <= 0 || !rays || !z) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: bad argument");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32_train: R != SB * rays_per_obj");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma;
    return pnr::eval_f32_train(scene, w, q, saved, (hipStream_t)stream, split_gemm != 0);
}

// ---- fp32-class training, FUSED (round 3).  The training forward is the split-operand inference kernel (pnr_split.hip, TRAIN
// instantiation: one launch, lin_z through the folded fp32 tables) that also copies the (head, tail) operand images of every
// linear out of LDS and writes 1-bit relu masks; the backward is bwd_split_kernel (all 15 transposed products of a network in
// one launch, gradient images copied out the same way) + ONE batched split-operand weight-gradient launch (dw_split_kernel,
// pnr_bwd.hip) straight from those images + lin_out's 4 x 512 gradient.  ~120 launches of the GEMM-per-layer form -> 8.
static int check_split_saved(const PnrSplitSaved *sv) {
    if (!sv || !sv->in_op || !sv->zlat || !sv->x5 || !sv->masks) return 0;
    for (int b = 0; b < 5; ++b)
        if (!sv->a[b] || !sv->n[b]) return 0;
    return 1;
}

extern "C" int pnr_eval_ray_samples_split_train(const PnrScene *scene, const void *packed_split, const void *tables_f32,
                                                const float *rays, const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                                const PnrSplitSaved *saved, void *stream) {
    using namespace pnr;
    if (R <= 0 || K <= 0 || rays_per_obj <= 0 || !rays || !z || !scene || !packed_split || !tables_f32 || !rgbsigma)
        return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: bad argument");
    if ((long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: R != SB * rays_per_obj");
    if (scene->SB <= 0 || scene->NS <= 0 || scene->Hl < 2 || scene->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: bad scene shape");
    if (!check_split_saved(saved)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: null buffer in PnrSplitSaved");
    EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K;
    if (q.P * scene->NS > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_split_train: too many points");
    q.latent = scene->latent_nhwc; q.poses = scene->poses; q.focal = scene->focal; q.c = scene->c;
    q.SB = scene->SB; q.NS = scene->NS; q.Hl = scene->Hl; q.Wl = scene->Wl; q.n_focal = scene->n_focal; q.n_c = scene->n_c;
    q.img_w = scene->img_w; q.img_h = scene->img_h;
    const int np = (int)q.P;
    const long long rows = (long long)np * scene->NS;
    // lin_in operand and interpolated latent as (head | tail) rows: operands of the lin_in / lin_z weight gradients
    hipLaunchKernelGGL((feat_f32_kernel<true, true>), dim3((unsigned)((rows + FW - 1) / FW)), dim3(FW * 64), 0, (hipStream_t)stream, q, 0LL,
                       np, (float *)saved->in_op, (float *)saved->zlat);
    int rc = pnr_check_launch("pnr_eval_ray_samples_split_train (features)");
    if (rc != PNR_OK) return rc;
    return eval_samples_split_train(scene, packed_split, tables_f32, rays, z, R, rays_per_obj, K, rgbsigma, saved->a, saved->n,
                                    saved->x5, saved->masks, (hipStream_t)stream);
}

static size_t split_bwd_images_bytes(long long P, int NS) { return ((size_t)P * NS * 7 + (size_t)P * 4) * pnr::D_HID * 4; }

extern "C" size_t pnr_mlp_backward_split_workspace_bytes(long long P, int NS) {
    if (P <= 0 || NS <= 0) return 0;
    // gradient images: g_fc1 / g_fc0 of blocks 0-2 and g_x0 at NS*P rows, of blocks 3-4 at P rows, 2 x 1024 B per row; the
    // transposed (head, tail) streams; the slice partials of the batched weight-gradient launch and of lin_out's
    return split_bwd_images_bytes(P, NS) + pnr::bwd_split_packed_bytes() + pnr_weight_grad_batched_workspace_bytes(14, P * NS) +
           (size_t)pnr::WG_SPLIT * (pnr::D_HID * pnr::D_HID + pnr::D_HID) * sizeof(float);
}

extern "C" int pnr_mlp_backward_split(const PnrMlpWeights *w, const PnrSplitSaved *sv, const float *g_out, long long P, int NS,
                                      const PnrMlpWeights *grads, float *d_zlat, float *d_in, const float *grad_scale, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    using namespace pnr;
    if (!w || !g_out || !grads || !d_zlat || !workspace || !grad_scale || P <= 0 || NS <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_split: bad argument");
    if (w->combine_max && NS > 1) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_split: combine_type \"max\" is an inference form (no backward)");
    if (!check_split_saved(sv)) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_split: null buffer in PnrSplitSaved");
    if (workspace_bytes < pnr_mlp_backward_split_workspace_bytes(P, NS)) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_split: workspace too small");
    if (P * NS > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_split: too many points");
    hipStream_t hs = (hipStream_t)stream;
    const long long rows = P * NS;
    char *cur = (char *)workspace;
    void *g_fc1[5], *g_fc0[5];
    for (int b = 0; b < 5; ++b) {
        const size_t n = (size_t)(b < COMBINE_LAYER ? rows : P) * D_HID * 4;
        g_fc1[b] = cur; cur += n;
        g_fc0[b] = cur; cur += n;
    }
    void *g_x0 = cur; cur += (size_t)rows * D_HID * 4;
    void *packed = cur; cur += bwd_split_packed_bytes();
    void *dw_ws = cur; cur += pnr_weight_grad_batched_workspace_bytes(14, rows);
    float *part = (float *)cur;
    int rc = pack_bwd_split(w, packed, hs);
    if (rc != PNR_OK) return rc;
    rc = mlp_backward_split_chain(packed, (const unsigned long long *)sv->masks, g_out, grad_scale, P, NS, g_fc1, g_fc0, g_x0, d_zlat, d_in, hs);
    if (rc != PNR_OK) return rc;
    // all 14 wide weight gradients in one launch pair, straight from the (head | tail) images; the chain ran at scale s: 1/s on the way out
    PnrWeightGradJob jobs[14];
    int nj = 0;
    for (int b = N_BLOCKS - 1; b >= 0; --b) {
        const long long r = b < COMBINE_LAYER ? rows : P;
        jobs[nj++] = PnrWeightGradJob{g_fc1[b], sv->n[b], r, 1, 1, (float *)grads->fc1_w[b], (float *)grads->fc1_b[b], 0, 0};
        jobs[nj++] = PnrWeightGradJob{g_fc0[b], sv->a[b], r, 1, 1, (float *)grads->fc0_w[b], (float *)grads->fc0_b[b], 0, 0};
    }
    // xin[b] = (stream in front) + lin_z[b](zlat): dY of lin_z[b] = gradient of the stream entering block b (resnetfc.py:175-180)
    for (int b = COMBINE_LAYER - 1; b >= 0; --b)
        jobs[nj++] = PnrWeightGradJob{b == 0 ? g_x0 : g_fc1[b - 1], sv->zlat, rows, 1, 0, (float *)grads->lin_z_w[b], (float *)grads->lin_z_b[b], 0, 0};
    jobs[nj++] = PnrWeightGradJob{g_x0, sv->in_op, rows, 1, 0, (float *)grads->lin_in_w, (float *)grads->lin_in_b, D_IN_PAD, D_IN};
    rc = pnr_weight_grad_batched(jobs, nj, PNR_PREC_F16X3, 1.f, grad_scale + 1, dw_ws, stream);
    if (rc != PNR_OK) return rc;
    static_assert((size_t)WG_SPLIT * (D_HID * D_HID + D_HID) >= (size_t)LOF_BLOCKS * (4 * D_HID + 4), "lin_out partials fit the slice workspace");
    lin_out_grad_f32(hs, g_out, sv->x5, P, (float *)grads->lin_out_w, (float *)grads->lin_out_b, part);
    return pnr_check_launch("pnr_mlp_backward_split");
}

extern "C" size_t pnr_mlp_backward_f32_workspace_bytes(long long P, int NS) {
    if (P <= 0 || NS <= 0) return 0;
    // G and T at (NS*P, 512), the pooled G at (P, 512), the row-slice partials of one weight gradient
    return ((size_t)P * NS * 2 + (size_t)P) * pnr::D_HID * sizeof(float) +
           (size_t)pnr::WG_SPLIT * (pnr::D_HID * pnr::D_HID + pnr::D_HID) * sizeof(float);
}

extern "C" int pnr_mlp_backward_f32(const PnrMlpWeights *w, const PnrF32Saved *sv, const float *g_out, long long P, int NS,
                                    const PnrMlpWeights *grads, float *d_zlat, float *d_in, int split_gemm, const float *grad_scale,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    using namespace pnr;
    if (!w || !g_out || !grads || !d_zlat || !workspace || P <= 0 || NS <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: bad argument");
    if (w->combine_max && NS > 1) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: combine_type \"max\" is an inference form (no backward)");
    if (split_gemm && !grad_scale)
        return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: the split-operand form needs grad_scale = device [s, 1/s] (pnr_grad_scale)");
    if (!check_saved(sv, NS)) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: null activation buffer in PnrF32Saved");
    if (workspace_bytes < pnr_mlp_backward_f32_workspace_bytes(P, NS)) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: workspace too small");
    if (P * NS > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_mlp_backward_f32: too many points");
    const Mm st = {(hipStream_t)stream, split_gemm != 0, split_gemm ? grad_scale : nullptr, split_gemm ? grad_scale + 1 : nullptr};
    const long long rows = P * NS;
    float *Gv = (float *)workspace;
    float *T = Gv + (size_t)rows * D_HID;
    float *Gp = T + (size_t)rows * D_HID;
    float *part = Gp + (size_t)P * D_HID;
    float *G = NS > 1 ? Gp : Gv;  // pooled part of the chain
    // lin_out (resnetfc.py:183): out = W_out relu(x5) + b
    // split-operand form: g_out enters at the power-of-two scale s (fp16 range); every gradient of the chain stays in that
    // domain and the results that leave it -- dW, db, d z_lat, d(code) -- are multiplied by 1/s on their way out (exact)
    wgrad(st, g_out, D_OUT, sv->x5, D_HID, true, P, D_OUT, D_HID, (float *)grads->lin_out_w, (float *)grads->lin_out_b, part, true);
    linear_bwd(st, g_out, D_OUT, w->lin_out_w, D_HID, D_OUT, G, D_HID, P, sv->x5, false, true);
    for (int b = N_BLOCKS - 1; b >= COMBINE_LAYER; --b) block_bwd_f32(st, w, grads, b, G, T, sv->xin[b], sv->net[b], P, part);
    if (NS > 1) {  // backward of the view mean: every view receives G / NS
        const long long n = P * D_HID;
        hipLaunchKernelGGL(unpool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st.st, Gp, Gv, P, NS, 1.f / (float)NS);
    }
    for (int b = COMBINE_LAYER - 1; b >= 0; --b) {
        block_bwd_f32(st, w, grads, b, Gv, T, sv->xin[b], sv->net[b], rows, part);
        // xin[b] = (stream in front) + lin_z[b](zlat)   (resnetfc.py:175-180)
        wgrad(st, Gv, D_HID, sv->zlat, C_LAT, false, rows, D_HID, C_LAT, (float *)grads->lin_z_w[b], (float *)grads->lin_z_b[b], part);
        linear_bwd(st, Gv, D_HID, w->lin_z_w[b], C_LAT, D_HID, d_zlat, C_LAT, rows, nullptr, b != COMBINE_LAYER - 1, false, true);
    }
    // lin_in (resnetfc.py:147)
    wgrad(st, Gv, D_HID, sv->in42, D_IN_PAD, false, rows, D_HID, D_IN, (float *)grads->lin_in_w, (float *)grads->lin_in_b, part);
    if (d_in) linear_bwd(st, Gv, D_HID, w->lin_in_w, D_IN, D_HID, d_in, D_IN, rows, nullptr, false, false, true);
    return pnr_check_launch("pnr_mlp_backward_f32");
}

extern "C" size_t pnr_resnetfc_forward_f32_workspace_bytes(long long rows, int NS) {
    if (rows <= 0 || NS <= 0 || rows % NS != 0) return 0;
    return ((size_t)rows * 2 + (NS > 1 ? (size_t)(rows / NS) * 2 : 0)) * pnr::D_HID * sizeof(float);
}

extern "C" int pnr_resnetfc_forward_f32(const PnrMlpWeights *w, const float *zx, long long rows, int NS, int B, float *out,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    using namespace pnr;
    if (!w || !out || !workspace || rows < 0 || NS <= 0 || B <= 0)
        return pnr_fail(PNR_E_INVALID, "pnr_resnetfc_forward_f32: bad argument");
    if (rows == 0) return PNR_OK;
    if (!zx) return pnr_fail(PNR_E_INVALID, "pnr_resnetfc_forward_f32: null zx");
    if (rows % ((long long)NS * B) != 0)
        return pnr_fail(PNR_E_INVALID, "pnr_resnetfc_forward_f32: rows must be a multiple of combine_inner_dims = (NS, B)");
    if (workspace_bytes < pnr_resnetfc_forward_f32_workspace_bytes(rows, NS))
        return pnr_fail(PNR_E_INVALID, "pnr_resnetfc_forward_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ld = C_LAT + D_IN;  // zx row: [z (512) | x (42)]   resnetfc.py:141-143
    const long long np = rows / NS;
    float *x = (float *)workspace;
    float *net = x + (size_t)rows * D_HID;
    float *xp = net + (size_t)rows * D_HID;
    float *netp = xp + (size_t)np * D_HID;
    linear(st, zx + C_LAT, ld, w->lin_in_w, w->lin_in_b, x, D_HID, rows, D_HID, D_IN, false, false);   // resnetfc.py:147
    for (int b = 0; b < COMBINE_LAYER; ++b) {
        linear(st, zx, ld, w->lin_z_w[b], w->lin_z_b[b], x, D_HID, rows, D_HID, C_LAT, false, true);   // :175-180
        linear(st, x, D_HID, w->fc0_w[b], w->fc0_b[b], net, D_HID, rows, D_HID, D_HID, true, false);    // :55-57
        linear(st, net, D_HID, w->fc1_w[b], w->fc1_b[b], x, D_HID, rows, D_HID, D_HID, true, true);     // :58-62
    }
    float *xs = x, *ns = net;
    if (NS > 1) {  // util.combine_interleaved(x, (NS, B), "average")   resnetfc.py:168-170
        const long long n = np * D_HID;
        hipLaunchKernelGGL(pool_interleaved_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xp, np / B, NS, B, w->combine_max ? 1 : 0);
        xs = xp; ns = netp;
    }
    for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b) {
        linear(st, xs, D_HID, w->fc0_w[b], w->fc0_b[b], ns, D_HID, np, D_HID, D_HID, true, false);
        linear(st, ns, D_HID, w->fc1_w[b], w->fc1_b[b], xs, D_HID, np, D_HID, D_HID, true, true);
    }
    linear(st, xs, D_HID, w->lin_out_w, w->lin_out_b, out, D_OUT, np, D_OUT, D_HID, true, false);       // :183 (raw output)
    return pnr_check_launch("pnr_resnetfc_forward_f32");
}

// ---- nn.Linear as a stand-alone operator pair (any rows / d_in / d_out): what ResnetFC / ResnetBlockFC of a NON-shipped shape
// are composed of on the host side (model/resnetfc.py: d_hidden != 512, other block counts, shortcut / Softplus blocks, SPADE).
static bool linear_precision_ok(int precision) { return precision == PNR_PREC_F32 || precision == PNR_PREC_F16X3; }

extern "C" int pnr_linear(const float *X, const float *W, const float *b, const float *Yin, float *Y, long long rows, int d_in,
                          int d_out, int relu_in, int precision, void *stream) {
    using namespace pnr;
    if (rows < 0 || d_in <= 0 || d_out <= 0 || !linear_precision_ok(precision))
        return pnr_fail(PNR_E_INVALID, "pnr_linear: rows >= 0, d_in > 0, d_out > 0, precision PNR_PREC_F32 or PNR_PREC_F16X3");
    if (rows == 0) return PNR_OK;
    if (!X || !W || !Y) return pnr_fail(PNR_E_INVALID, "pnr_linear: null X / W / Y");
    const Mm c = {(hipStream_t)stream, precision == PNR_PREC_F16X3, nullptr, nullptr};
    linear(c, X, d_in, W, b, Y, d_out, rows, d_out, d_in, relu_in != 0, Yin != nullptr, Yin);
    return pnr_check_launch("pnr_linear");
}

extern "C" size_t pnr_linear_backward_workspace_bytes(int d_in, int d_out) {
    if (d_in <= 0 || d_out <= 0) return 0;
    return (size_t)pnr::WG_SPLIT * ((size_t)d_in * d_out + d_out) * sizeof(float);
}

extern "C" int pnr_linear_backward(const float *dY, const float *X, const float *W, long long rows, int d_in, int d_out, int relu_in,
                                   float *dX, float *dW, float *db, const float *grad_scale, void *workspace, size_t workspace_bytes,
                                   int precision, void *stream) {
    using namespace pnr;
    if (rows <= 0 || d_in <= 0 || d_out <= 0 || !linear_precision_ok(precision))
        return pnr_fail(PNR_E_INVALID, "pnr_linear_backward: rows > 0, d_in > 0, d_out > 0, precision PNR_PREC_F32 or PNR_PREC_F16X3");
    if (!dY || !X || !W) return pnr_fail(PNR_E_INVALID, "pnr_linear_backward: null dY / X / W");
    if (db && !dW) return pnr_fail(PNR_E_INVALID, "pnr_linear_backward: db comes out of the dW pass -- pass dW as well");
    const bool fast = precision == PNR_PREC_F16X3;
    if (fast && !grad_scale)
        return pnr_fail(PNR_E_INVALID, "pnr_linear_backward: PNR_PREC_F16X3 needs grad_scale = device [s, 1/s] (pnr_grad_scale(dY))");
    const Mm c = {(hipStream_t)stream, fast, fast ? grad_scale : nullptr, fast ? grad_scale + 1 : nullptr};
    if (dW) {
        if (!workspace || workspace_bytes < pnr_linear_backward_workspace_bytes(d_in, d_out))
            return pnr_fail(PNR_E_INVALID, "pnr_linear_backward: workspace too small (pnr_linear_backward_workspace_bytes)");
        wgrad(c, dY, d_out, X, d_in, relu_in != 0, rows, d_out, d_in, dW, db, (float *)workspace, true);
    }
    if (dX) linear_bwd(c, dY, d_out, W, d_in, d_out, dX, d_in, rows, relu_in ? X : nullptr, false, true, true);
    return pnr_check_launch("pnr_linear_backward");
}

extern "C" size_t pnr_eval_f32_workspace_bytes(int NS, long long chunk_points) {
    if (NS <= 0 || chunk_points <= 0) return 0;
    return pnr::floats_per_point(NS) * (size_t)chunk_points * sizeof(float);
}

extern "C" int pnr_eval_ray_samples_f32(const PnrScene *scene, const PnrMlpWeights *w, const float *rays, const float *z, int R,
                                        int rays_per_obj, int K, float *rgbsigma, void *workspace, size_t workspace_bytes,
                                        void *stream) {
    if (R < 0 || K <= 0 || rays_per_obj <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32: bad sizes");
    if (R > 0 && (!rays || !z)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32: null rays/z");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples_f32: R != SB * rays_per_obj");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma;
    return pnr::eval_f32(scene, w, q, true, (float *)workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int pnr_eval_points_f32(const PnrScene *scene, const PnrMlpWeights *w, const float *xyz, const float *viewdirs, int B,
                                   float *rgbsigma, void *workspace, size_t workspace_bytes, void *stream) {
    if (B < 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_points_f32: bad sizes");
    if (B > 0 && (!xyz || !viewdirs)) return pnr_fail(PNR_E_INVALID, "pnr_eval_points_f32: null xyz/viewdirs");
    pnr::EvalParams q = {};
    q.xyz = xyz; q.viewdirs = viewdirs; q.K = 1; q.per_obj = B > 0 ? B : 1;
    q.P = scene ? (long long)scene->SB * B : 0; q.out = rgbsigma;
    return pnr::eval_f32(scene, w, q, false, (float *)workspace, workspace_bytes, (hipStream_t)stream);
}

// The feature phase alone (SURVEY rows R7 + R8 in isolation): lin_in operand rows [code(39) | R d (3) | 0-pad] and the
// bilinearly interpolated latent rows of B points per object, rows ordered [view][object][point] like the dumps.
extern "C" int pnr_point_features_f32(const PnrScene *s, const float *xyz, const float *viewdirs, int B, float *in42,
                                      float *zlat, void *stream) {
    using namespace pnr;
    if (!s || B < 0 || !in42 || !zlat) return pnr_fail(PNR_E_INVALID, "pnr_point_features_f32: bad argument");
    if (s->SB <= 0 || s->NS <= 0 || s->Hl < 2 || s->Wl < 2) return pnr_fail(PNR_E_INVALID, "pnr_point_features_f32: bad scene shape");
    if (B == 0) return PNR_OK;
    if (!xyz || !viewdirs) return pnr_fail(PNR_E_INVALID, "pnr_point_features_f32: null xyz/viewdirs");
    EvalParams q = {};
    q.xyz = xyz; q.viewdirs = viewdirs; q.K = 1; q.per_obj = B; q.P = (long long)s->SB * B;
    if (q.P * s->NS > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_point_features_f32: too many points");
    q.latent = s->latent_nhwc; q.poses = s->poses; q.focal = s->focal; q.c = s->c;
    q.SB = s->SB; q.NS = s->NS; q.Hl = s->Hl; q.Wl = s->Wl; q.n_focal = s->n_focal; q.n_c = s->n_c;
    q.img_w = s->img_w; q.img_h = s->img_h;
    const long long rows = q.P * s->NS;
    hipLaunchKernelGGL(feat_f32_kernel<false>, dim3((unsigned)((rows + FW - 1) / FW)), dim3(FW * 64), 0, (hipStream_t)stream, q, 0LL,
                       (int)q.P, in42, zlat);
    return pnr_check_launch("pnr_point_features_f32");
}
