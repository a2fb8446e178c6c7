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

#include <vector>

#include "pnr_common.h"
#include "pnr_layout.h"

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
    static __device__ __forceinline__ f32x16 mfma(T8 a, T8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Prec<PNR_PREC_BF16> {
    typedef __bf16 T;
    typedef bf16x8 T8;
    typedef bf16x2 T2;
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
    // points: variant A (rays + z) or B (xyz + viewdirs)
    const float *rays, *z, *xyz, *viewdirs;
    int K;             // samples per ray (A)
    int per_obj;       // rays per object (A) or points per object (B)
    long long P;       // total points
    int ntiles;
    float *out;        // (P,4)
    float *dbg;        // optional debug dump of the final residual stream x (P,512), may be null
    unsigned long long *tim;  // phase-timing accumulators (TIMING instantiation only)
};

// phase ids of the TIMING instantiation (wave 0 of workgroup 0, s_memtime ticks)
enum Phase { PH_SYNC_TOP = 0, PH_GEOMETRY, PH_GATHER, PH_GEMM_IN_Z0, PH_BAR1, PH_WRITE_X, PH_BAR2, PH_GEMM_FC0, PH_BAR3,
             PH_WRITE_NET, PH_BAR4, PH_GEMM_FC1_Z, PH_LIN_OUT, PH_BAR_OUT, PH_FINAL, NPHASE };
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

// 8 fp32 -> 8 x 16-bit, optional relu
template <typename P, bool RELU>
__device__ __forceinline__ typename P::T8 pack8(float v0, float v1, float v2, float v3, float v4, float v5,
                                                float v6, float v7) {
    if (RELU) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        v4 = fmaxf(v4, 0.f); v5 = fmaxf(v5, 0.f); v6 = fmaxf(v6, 0.f); v7 = fmaxf(v7, 0.f);
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

template <typename P> __device__ __forceinline__ void ring_advance4(Ring<P> &R, int NS) {
    int rs = R.pf_rs + 4, v = R.pf_view;
    if (rs == RS_VIEW_END) {
        if (v + 1 < NS) { v += 1; rs = 0; }
    } else if (rs == RS_TOTAL) {
        rs = 0; v = 0;
    }
    R.pf_rs = rs; R.pf_view = v;
}

// acc[it][jt] += W-fragments (ring) x B-fragments (LDS rows baddr0/baddr1, 32 B per k-step),
// nbody*4 k-steps.
template <typename P>
__device__ __forceinline__ void gemm(f32x16 (&acc)[IT][JT], const char *smem, uint32_t baddr0, uint32_t baddr1,
                                     int nbody, Ring<P> &R, int NS) {
    typename P::T8 b[2][JT];
    b[0][0] = lds8<P>(smem, baddr0);
    b[0][1] = lds8<P>(smem, baddr1);
#pragma unroll 1
    for (int body = 0; body < nbody; ++body) {

#ifdef PNR_EXP_FAKE_W  // experiment: refill from a fixed 8 KiB window (no L2 streaming); results are wrong
        const char *pf = R.wave_base + (size_t)(R.pf_rs & 0) * (IT * 1024);
#else
        const char *pf = R.wave_base + (size_t)R.pf_rs * (IT * 1024);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            // intended step order: LDS reads for step+1 | 4 MFMAs of this step | refill this ring
            // slot (step+4).  hipcc re-orders this (it batches the 8 refills behind the last MFMA
            // of the body); pinning the order with sched_barrier (-DPNR_PIN_SCHEDULE) gives the
            // textbook stream but measured 2-3 % SLOWER (profiles/r01_gemm_experiments.md), so
            // the compiler's schedule is the default.
            b[cur ^ 1][0] = lds8<P>(smem, baddr0 + (j + 1) * 32);
            b[cur ^ 1][1] = lds8<P>(smem, baddr1 + (j + 1) * 32);
#ifdef PNR_PIN_SCHEDULE
            __builtin_amdgcn_sched_barrier(0);
#endif
            const typename P::T8 a0 = R.r[j][0], a1 = R.r[j][1];
            acc[0][0] = P::mfma(a0, b[cur][0], acc[0][0]);
            acc[0][1] = P::mfma(a0, b[cur][1], acc[0][1]);
            acc[1][0] = P::mfma(a1, b[cur][0], acc[1][0]);
            acc[1][1] = P::mfma(a1, b[cur][1], acc[1][1]);
#ifdef PNR_PIN_SCHEDULE
            __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef PNR_EXP_NO_WLOAD  // experiment: never refill the ring (no weight traffic at all); results are wrong
            R.r[j][0] = gload8<P>(pf + j * (IT * 1024));
            R.r[j][1] = gload8<P>(pf + j * (IT * 1024) + 1024);
#endif
#ifdef PNR_PIN_SCHEDULE
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        baddr0 += 128;
        baddr1 += 128;
        ring_advance4(R, NS);
    }
}

// relu(acc) -> 16-bit -> activation buffer.  Lane (p,h) writes registers 0..15 of feature tile
// T = wave*IT+it as 32 contiguous bytes at element offset 32T + 16h of its point row.
template <typename P>
__device__ __forceinline__ void write_act(const f32x16 (&acc)[IT][JT], char *smem, uint32_t waddr) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const f32x16 &a = acc[it][jt];
            const uint32_t ad = waddr + jt * 32 * ROW_ACT + it * 64;
            *reinterpret_cast<typename P::T8 *>(smem + ad) =
                pack8<P, true>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
            *reinterpret_cast<typename P::T8 *>(smem + ad + 16) =
                pack8<P, true>(a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
        }
}

template <bool INIT>
__device__ __forceinline__ void add_bias(f32x16 (&acc)[IT][JT], const float *bias_lane, int slot) {
    // bias_lane = bias + wave*BIAS_FLOATS_PER_WAVE + h*16 ; slot stride NW*BIAS_FLOATS_PER_WAVE
    const float *b = bias_lane + (size_t)slot * (NW * BIAS_FLOATS_PER_WAVE);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        f32x4 q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const f32x4 *>(b + it * 32 + i * 4);
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (INIT) acc[it][jt][r] = q[r >> 2][r & 3];
                else acc[it][jt][r] += q[r >> 2][r & 3];
            }
    }
}

// ---------------------------------------------------------------- feature phase (geometry)
// Thread (p = tid&63, sub = tid>>6).  Follows the reference op order without FMA contraction
// so that fp32 intermediates round like the PyTorch eager path.
#pragma clang fp contract(off)
template <typename P, bool RAYS>
__device__ __forceinline__ void geometry(const EvalParams &q, char *smem, int tile, int view, int tid) {
    typedef typename P::T T;
    const int p = tid & 63, sub = tid >> 6;
    const int g = tile * MT + p;  // P < 2^31 (checked on the host)
    const bool valid = g < (int)q.P;
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
    int obj = 0;
    float X = 0, Y = 0, Z = 0;
    if (valid) {
        if (RAYS) {
            const int r = g / q.K;
            const float *ray = q.rays + (size_t)r * 8;
            ox = ray[0]; oy = ray[1]; oz = ray[2]; dx = ray[3]; dy = ray[4]; dz = ray[5];
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
    if (sub == 0) {
        // identity part of the code, rotated view direction (models.py:188-196), zero pad
        const float dv0 = pose[0] * dx + pose[1] * dy + pose[2] * dz;
        const float dv1 = pose[4] * dx + pose[5] * dy + pose[6] * dz;
        const float dv2 = pose[8] * dx + pose[9] * dy + pose[10] * dz;
        in_row[0] = (T)(valid ? xr0 : 0.f); in_row[1] = (T)(valid ? xr1 : 0.f); in_row[2] = (T)(valid ? xr2 : 0.f);
        in_row[39] = (T)dv0; in_row[40] = (T)dv1; in_row[41] = (T)dv2;
        // camera-space point and pinhole projection (models.py:165,206-212)
        const float xc0 = xr0 + pose[3], xc1 = xr1 + pose[7], xc2 = xr2 + pose[11];
        const float *fo = q.focal + (q.n_focal > 1 ? obj * 2 : 0);
        const float *cc = q.c + (q.n_c > 1 ? obj * 2 : 0);
        float u = -xc0 / xc2; u = u * fo[0]; u = u + cc[0];
        float v = -xc1 / xc2; v = v * fo[1]; v = v + cc[1];
        // SpatialEncoder.index (encoder.py:96-99,161-163) + grid_sample(bilinear, border,
        // align_corners=True)
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
        uint32_t *mo = reinterpret_cast<uint32_t *>(smem + LDS_META + p * 32);
        float *mw = reinterpret_cast<float *>(smem + LDS_META + p * 32 + 16);
        mo[0] = (rowbase + y0 * q.Wl + x0) * C_LAT; mo[1] = (rowbase + y0 * q.Wl + x1) * C_LAT;
        mo[2] = (rowbase + y1 * q.Wl + x0) * C_LAT; mo[3] = (rowbase + y1 * q.Wl + x1) * C_LAT;
        mw[0] = w_nw; mw[1] = w_ne; mw[2] = w_sw; mw[3] = w_se;
    } else if (sub <= 6) {
        // frequency k = sub-1: sin(f x), sin(f x + pi/2), f = 1.5 * 2^k (code.py:15,37-41)
        const float f = 1.5f * (float)(1 << (sub - 1));
        const float HALF_PI = 1.57079637050628662109375f;  // fp32(pi/2), code.py:26
        const float a0 = xr0 * f, a1 = xr1 * f, a2 = xr2 * f;
        T *o = in_row + 3 + 6 * (sub - 1);
        o[0] = (T)(valid ? sinf(a0) : 0.f); o[1] = (T)(valid ? sinf(a1) : 0.f); o[2] = (T)(valid ? sinf(a2) : 0.f);
        o[3] = (T)(valid ? sinf(a0 + HALF_PI) : 0.f); o[4] = (T)(valid ? sinf(a1 + HALF_PI) : 0.f);
        o[5] = (T)(valid ? sinf(a2 + HALF_PI) : 0.f);
    } else {
        // zero the K padding 42..63 (+ the 8-element row pad read by the last B prefetch)
        for (int i = D_IN; i < D_IN_PAD + 8; ++i) in_row[i] = (T)0.f;
    }
}
#pragma clang fp contract(fast)

// bilinear lookup: wave handles points wave*8..+7; lane handles channels 8*lane..+7
template <typename P>
__device__ __forceinline__ void gather(const EvalParams &q, char *smem, int wv, int lane) {
    const float *lat = q.latent + lane * 8;
#pragma unroll 2
    for (int i = 0; i < MT / NW; i += 2) {
        f32x4 v[2][4][2];
        f32x4 w[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
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
        for (int u = 0; u < 2; ++u) {
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
            *reinterpret_cast<typename P::T8 *>(smem + LDS_Z + p * ROW_ACT + lane * 16) =
                pack8<P, false>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
        }
    }
}

// one residual block (+ the lin_z of the next block when with_z):
//   net = fc_0(relu(x)); x += fc_1(relu(net)) [+ lin_z[b+1](z)]       resnetfc.py:55-62,174-182
template <typename P, bool TIMING>
__device__ __forceinline__ void res_block(f32x16 (&x)[IT][JT], char *smem, int b, bool with_z, Ring<P> &R,
                                          int NS, const float *bias_lane, uint32_t a_rd0, uint32_t a_rd1,
                                          uint32_t z_rd0, uint32_t z_rd1, uint32_t a_wr, int tid,
                                          unsigned long long *tim, unsigned long long &tlast) {
    __syncthreads();  // every wave is done reading LDS_A (previous fc_1)
    PNR_T(PH_BAR1);
    write_act<P>(x, smem, a_wr);
    PNR_T(PH_WRITE_X);
    __syncthreads();
    PNR_T(PH_BAR2);
    {
        f32x16 net[IT][JT];
        add_bias<true>(net, bias_lane, 1 + 2 * b);
        gemm<P>(net, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
        PNR_T(PH_GEMM_FC0);
        __syncthreads();  // every wave is done reading relu(x)
        PNR_T(PH_BAR3);
        write_act<P>(net, smem, a_wr);
        PNR_T(PH_WRITE_NET);
    }
    __syncthreads();
    PNR_T(PH_BAR4);
    add_bias<false>(x, bias_lane, 2 + 2 * b);
    gemm<P>(x, smem, a_rd0, a_rd1, KS_BIG / 4, R, NS);
    if (with_z) gemm<P>(x, smem, z_rd0, z_rd1, KS_BIG / 4, R, NS);
    PNR_T(PH_GEMM_FC1_Z);
}

template <int PREC, bool RAYS, bool MV, bool TIMING = false>
__global__ void __launch_bounds__(NTHREADS, 2) eval_kernel(const EvalParams q) {
    typedef Prec<PREC> P;
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

    Ring<P> R;
    R.wave_base = q.wstream + (size_t)wv * (RS_TOTAL * IT * 1024) + lane * 16;
    R.pf_rs = 0;
    R.pf_view = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        R.r[j][0] = gload8<P>(R.wave_base + j * (IT * 1024));
        R.r[j][1] = gload8<P>(R.wave_base + j * (IT * 1024) + 1024);
    }
    R.pf_rs = 4;
    unsigned long long *tim = q.tim;
    unsigned long long tlast = TIMING ? __builtin_readcyclecounter() : 0ull;

    for (int tile = blockIdx.x; tile < q.ntiles; tile += gridDim.x) {
        f32x16 x[IT][JT];
        f32x16 xsum[MV ? IT : 1][MV ? JT : 1];
#pragma unroll 1
        for (int view = 0; view < NS; ++view) {
            __syncthreads();  // previous users of LDS_IN / LDS_META / LDS_Z are done
            PNR_T(PH_SYNC_TOP);
            geometry<P, RAYS>(q, smem, tile, view, tid);
            __syncthreads();
            PNR_T(PH_GEOMETRY);
            gather<P>(q, smem, wv, lane);
            __syncthreads();
            PNR_T(PH_GATHER);
            add_bias<true>(x, bias_lane, B_IN_Z0);
            gemm<P>(x, smem, in_rd0, in_rd1, KS_IN / 4, R, NS);      // lin_in     resnetfc.py:147
            gemm<P>(x, smem, z_rd0, z_rd1, KS_BIG / 4, R, NS);       // lin_z[0]   resnetfc.py:175-180
            PNR_T(PH_GEMM_IN_Z0);
#pragma unroll 1
            for (int b = 0; b < COMBINE_LAYER; ++b)
                res_block<P, TIMING>(x, smem, b, b + 1 < COMBINE_LAYER, R, NS, bias_lane, a_rd0, a_rd1, z_rd0, z_rd1, a_wr,
                                     tid, tim, tlast);
            if constexpr (MV) {
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        if (view == 0) xsum[it][jt] = x[it][jt];
                        else xsum[it][jt] += x[it][jt];
                    }
            }
        }
        if constexpr (MV) {
            // mean over source views (util.combine_interleaved, util.py:461-466)
            const float inv = 1.f / (float)NS;
#pragma unroll
            for (int it = 0; it < IT; ++it)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) x[it][jt] = xsum[it][jt] * inv;
        }
#pragma unroll 1
        for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b)
            res_block<P, TIMING>(x, smem, b, false, R, NS, bias_lane, a_rd0, a_rd1, z_rd0, z_rd1, a_wr, tid, tim, tlast);

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
#pragma unroll
            for (int qk = 0; qk < 4; ++qk) {
                const int xit = qk >> 1, rr = qk & 1;
                const typename P::T8 a = R.r[qk >> 1][qk & 1];
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const f32x16 &v = x[xit][jt];
                    const typename P::T8 bq =
                        rr == 0 ? pack8<P, true>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
                                : pack8<P, true>(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]);
                    o[jt] = P::mfma(a, bq, o[jt]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R.r[j][0] = gload8<P>(pf + j * (IT * 1024));
                R.r[j][1] = gload8<P>(pf + j * (IT * 1024) + 1024);
            }
            ring_advance4(R, NS);
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

template <int PREC, bool RAYS>
static int launch(const EvalParams &q, bool mv, int grid, hipStream_t st) {
    hipError_t e;
    auto k = mv ? eval_kernel<PREC, RAYS, true> : eval_kernel<PREC, RAYS, false>;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(eval_kernel)");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_profile) {
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), LDS_TOTAL, st, q);
    if (g_profile) {
        hipEventRecord(e1, st);
        g_events.emplace_back(e0, e1);
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

static int eval_common(const PnrScene *s, const void *packed, int precision, EvalParams &q, bool rays, hipStream_t st) {
    if (!s || !packed || !q.out) return pnr_fail(PNR_E_INVALID, "pnr_eval: null argument");
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
    const long long nt = (q.P + MT - 1) / MT;
    if (q.P > 0x7fffffc0LL) return pnr_fail(PNR_E_INVALID, "pnr_eval: too many points (P must stay below 2^31)");
    q.ntiles = (int)nt;
    const int grid = (int)(nt < num_cus() ? nt : num_cus());
    const bool mv = s->NS > 1;
    if (precision == PNR_PREC_F16) return rays ? launch<PNR_PREC_F16, true>(q, mv, grid, st) : launch<PNR_PREC_F16, false>(q, mv, grid, st);
    if (precision == PNR_PREC_BF16) return rays ? launch<PNR_PREC_BF16, true>(q, mv, grid, st) : launch<PNR_PREC_BF16, false>(q, mv, grid, st);
    return pnr_fail(PNR_E_INVALID, "pnr_eval: unknown precision");
}

}  // namespace pnr

// test/diagnostic hook (not in the public header): per-phase s_memtime totals of wave 0 of
// workgroup 0 for one f16 single-view launch.  tim: NW*NPHASE device counters, zeroed by the caller.
extern "C" int pnr_debug_phase_timing(const PnrScene *s, const void *packed, const float *rays, const float *z, int R,
                                      int rays_per_obj, int K, unsigned long long *tim, void *stream) {
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
    q.ntiles = (int)((q.P + MT - 1) / MT);
    static float *scratch_out = nullptr;
    static long long scratch_n = 0;
    if (scratch_n < q.P) {
        if (scratch_out) (void)hipFree(scratch_out);
        if (hipMalloc(&scratch_out, (size_t)q.P * 16) != hipSuccess) return pnr_fail(PNR_E_HIP, "hipMalloc");
        scratch_n = q.P;
    }
    q.out = scratch_out;
    auto k = eval_kernel<PNR_PREC_F16, true, false, true>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute");
    const int grid = q.ntiles < num_cus() ? q.ntiles : num_cus();
    hipLaunchKernelGGL(k, dim3(grid), dim3(NTHREADS), LDS_TOTAL, (hipStream_t)stream, q);
    return pnr_check_launch("eval_kernel<timing>");
}

static float *g_dbg_ptr = nullptr;
// test hook (not part of the public header): dump the final residual stream of the next launches
extern "C" int pnr_debug_set_x_dump(float *ptr) { g_dbg_ptr = ptr; return PNR_OK; }

extern "C" int pnr_eval_ray_samples(const PnrScene *scene, const void *packed, int precision, const float *rays,
                                    const float *z, int R, int rays_per_obj, int K, float *rgbsigma, void *stream) {
    if (R < 0 || K <= 0 || rays_per_obj <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: bad sizes");
    if (R > 0 && (!rays || !z)) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: null rays/z");
    if (scene && (long long)rays_per_obj * scene->SB != R) return pnr_fail(PNR_E_INVALID, "pnr_eval_ray_samples: R != SB * rays_per_obj");
    pnr::EvalParams q = {};
    q.rays = rays; q.z = z; q.K = K; q.per_obj = rays_per_obj; q.P = (long long)R * K; q.out = rgbsigma; q.dbg = g_dbg_ptr;
    return pnr::eval_common(scene, packed, precision, q, true, (hipStream_t)stream);
}

extern "C" int pnr_eval_points(const PnrScene *scene, const void *packed, int precision, const float *xyz,
                               const float *viewdirs, int B, float *rgbsigma, void *stream) {
    if (B < 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_points: bad sizes");
    if (B > 0 && (!xyz || !viewdirs)) return pnr_fail(PNR_E_INVALID, "pnr_eval_points: null xyz/viewdirs");
    pnr::EvalParams q = {};
    q.xyz = xyz; q.viewdirs = viewdirs; q.K = 1; q.per_obj = B > 0 ? B : 1;
    q.P = scene ? (long long)scene->SB * B : 0; q.out = rgbsigma; q.dbg = g_dbg_ptr;
    return pnr::eval_common(scene, packed, precision, q, false, (hipStream_t)stream);
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
