// pnr_render.hip -- ray sampling, inverse-CDF resampling + merge sort, alpha compositing and ray
// generation for gfx950.  One 64-lane wavefront per ray; prefix sums / products are wavefront
// scans (shuffles), reductions are wavefront butterflies.  These stages move <= 2 KiB per ray
// and are bandwidth/latency-trivial next to the fused network.
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_internal.h"
#include "pnr_raysrc.h"

namespace pnr {

constexpr int WAVES_PER_BLOCK = 4;
// sample_fine keeps a ray's cdf (Kc + 1 floats) and its merged sample set (Kc + Kf floats) in LDS, one wavefront per ray:
// dynamic shared memory, so the only limit is the 160 KiB of a CU (about 10 000 samples per ray at 4 rays per workgroup) --
// the reference has none (nerf.py:120-161), and no shipped config comes near it (64 + 128)
constexpr int SAMPLE_FINE_LDS_MAX = 160 * 1024;

#pragma clang fp contract(off)  // keep the reference's separately-rounded mul/add sequences

// z = near (1-t) + far t   or its linear-in-disparity form (nerf.py:112-115,143-147)
__device__ __forceinline__ float z_from_t(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.f - t) + far * t;
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
}

// torch.linspace(0, 1-step, n)[i] as ATen's CPU kernel evaluates it (symmetric halves)
__device__ __forceinline__ float linspace_at(float end, int n, int i) {
    if (n <= 1) return 0.f;
    const float step = end / (float)(n - 1);
    const int half = n / 2;
    return i < half ? step * (float)i : end - step * (float)(n - i - 1);
}

// NeRFRenderer.sample_coarse, nerf.py:98-118
__global__ void sample_coarse_kernel(const RaySrc rs, const NoiseSrc ns, int R, int Kc, int lindisp, float *__restrict__ z) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * Kc) return;
    const int r = (int)(idx / Kc), i = (int)(idx % Kc);
    float near, far;
    load_bounds(rs, r, near, far);
    const float step = 1.0f / (float)Kc;
    float t = linspace_at(1.f - step, Kc, i);
    t = t + noise_u1(ns, r, i, Kc) * step;
    z[idx] = z_from_t(near, far, t, lindisp);
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive scans across the 64 lanes of a wavefront
__device__ __forceinline__ double wave_scan_add(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}

// NeRFRenderer.sample_fine (nerf.py:120-148) + sample_fine_depth (:150-161) + cat/sort (:294-295)
__global__ void __launch_bounds__(WAVES_PER_BLOCK * 64)
sample_fine_kernel(const RaySrc rs, const float *__restrict__ wc, const float *__restrict__ depth_c,
                   const float *__restrict__ zc, const NoiseSrc ns, int R, int Kc, int Kimp, int Kfd, float depth_std, int lindisp,
                   float *__restrict__ zout, int *__restrict__ depth_ranks, float *__restrict__ z_new,
                   int *__restrict__ ranks_all) {
    extern __shared__ float s_fine[];  // per wavefront: cdf[Kc + 1] | z[Kc + Kimp + Kfd]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = blockIdx.x * WAVES_PER_BLOCK + wv;
    const bool active = r0 < R;  // inactive wavefronts still take part in the block barriers
    const int r = active ? r0 : R - 1;
    const int Ktot = Kc + Kimp + Kfd;
    float *cdf = s_fine + (size_t)wv * (Kc + 1 + Ktot), *zs = cdf + Kc + 1;
    float near, far;
    load_bounds(rs, r, near, far);

    if (Kimp > 0) {
        // weights + 1e-5, pdf, cdf with leading 0 (nerf.py:130-133).  The running sum is kept in
        // fp64 like ATen's CPU cumsum (acc_type<float> = double) and rounded per element.
        float part = 0.f;
        for (int i = lane; i < Kc; i += 64) part += wc[(size_t)r * Kc + i] + 1e-5f;
        const float tot = wave_sum(part);
        double run = 0.0;
        for (int c0 = 0; c0 < Kc; c0 += 64) {
            const int i = c0 + lane;
            const float pdf = i < Kc ? (wc[(size_t)r * Kc + i] + 1e-5f) / tot : 0.f;
            const double inc = wave_scan_add((double)pdf, lane) + run;
            if (i < Kc) cdf[i + 1] = (float)inc;
            run = __shfl(inc, 63, 64);
        }
        if (lane == 0) cdf[0] = 0.f;
    }
    for (int i = lane; i < Kc; i += 64) zs[i] = zc[(size_t)r * Kc + i];
    __syncthreads();
    for (int j = lane; j < Kimp; j += 64) {
        const float u = noise_u2(ns, r, j, Kimp);
        // searchsorted(cdf, u, right=True): number of entries <= u      (nerf.py:138)
        int lo = 0, hi = Kc + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const float ind = fmaxf((float)lo - 1.0f, 0.0f);  // :138-139 (may equal Kc)
        const float t = (ind + noise_u3(ns, r, j, Kimp)) / (float)Kc;  // :141
        zs[Kc + j] = z_from_t(near, far, t, lindisp);
    }
    for (int j = lane; j < Kfd; j += 64) {
        float z = depth_c[r] + noise_n4(ns, r, j, Kfd) * depth_std;  // :157-158
        z = fmaxf(fminf(z, far), near);                              // :160
        zs[Kc + Kimp + j] = z;
    }
    __syncthreads();
    if (!active) return;
    // ascending sort by rank counting (ties broken by position; K^2/64 LDS broadcasts per lane)
    for (int e = lane; e < Ktot; e += 64) {
        const float v = zs[e];
        int rank = 0;
        for (int j = 0; j < Ktot; ++j) {
            const float o = zs[j];
            rank += (o < v || (o == v && j < e)) ? 1 : 0;
        }
        zout[(size_t)r * Ktot + rank] = v;
        if (depth_ranks && e >= Kc + Kimp) depth_ranks[(size_t)r * Kfd + (e - Kc - Kimp)] = rank;
        // optional: the new samples in draw order and every source element's sorted position (coarse-network reuse)
        if (ranks_all) ranks_all[(size_t)r * Ktot + e] = rank;
        if (z_new && e >= Kc) z_new[(size_t)r * (Kimp + Kfd) + (e - Kc)] = v;
    }
}

// fine pass on the coarse network (mlp_fine is None, models.py:242): the Kc coarse samples were already evaluated by
// the same network at the same points -- a point's output does not depend on its neighbours -- so only the Kf new
// samples are evaluated and the two result sets are merged into sorted order.
__global__ void merge_rgbsigma_kernel(const float4 *__restrict__ rgbs_c, const float4 *__restrict__ rgbs_new,
                                      const int *__restrict__ ranks_all, int R, int Kc, int Kf, float4 *__restrict__ rgbs_f) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Kt = Kc + Kf;
    if (idx >= (long long)R * Kt) return;
    const int r = (int)(idx / Kt), e = (int)(idx % Kt);
    const float4 v = e < Kc ? rgbs_c[(size_t)r * Kc + e] : rgbs_new[(size_t)r * Kf + (e - Kc)];
    rgbs_f[(size_t)r * Kt + ranks_all[idx]] = v;
}

// NeRFRenderer.composite, nerf.py:178-182 (deltas) and :223-249
__global__ void __launch_bounds__(WAVES_PER_BLOCK * 64)
composite_kernel(const RaySrc rs, const float *__restrict__ z, const float4 *__restrict__ rgbs, int R,
                 int K, int white_bkgd, float *__restrict__ weights, float *__restrict__ rgb,
                 float *__restrict__ depth) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * WAVES_PER_BLOCK + wv;
    if (r >= R) return;
    float near_unused, far;
    load_bounds(rs, r, near_unused, far);
    const float *zr = z + (size_t)r * K;
    float carry = 1.f;  // transmittance in front of this chunk: prod_{j<c0} (1 - a_j + 1e-10)
    float acc_r = 0.f, acc_g = 0.f, acc_b = 0.f, acc_d = 0.f, acc_w = 0.f;
    for (int c0 = 0; c0 < K; c0 += 64) {
        const int i = c0 + lane;
        const bool valid = i < K;
        float zi = 0.f, alpha = 0.f;
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            zi = zr[i];
            const float znext = (i + 1 < K) ? zr[i + 1] : far;  // :181 last delta = far - z_last
            const float delta = znext - zi;
            cs = rgbs[(size_t)r * K + i];
            alpha = 1.f - expf(-delta * fmaxf(cs.w, 0.f));  // :228
        }
        const float tfac = valid ? (1.f - alpha + 1e-10f) : 1.f;  // :230-232
        const float incl = wave_scan_mul(tfac, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;  // cumprod shifted by one, :234-235
        const float w = alpha * T;
        if (valid) {
            if (weights) weights[(size_t)r * K + i] = w;
            acc_r += w * cs.x; acc_g += w * cs.y; acc_b += w * cs.z;  // :239
            acc_d += w * zi;                                          // :240
            acc_w += w;
        }
        carry = carry * __shfl(incl, 63, 64);
    }
    acc_r = wave_sum(acc_r); acc_g = wave_sum(acc_g); acc_b = wave_sum(acc_b);
    acc_d = wave_sum(acc_d); acc_w = wave_sum(acc_w);
    if (lane == 0) {
        if (white_bkgd) {  // :241-244
            acc_r = acc_r + 1.f - acc_w; acc_g = acc_g + 1.f - acc_w; acc_b = acc_b + 1.f - acc_w;
        }
        rgb[(size_t)r * 3 + 0] = acc_r; rgb[(size_t)r * 3 + 1] = acc_g; rgb[(size_t)r * 3 + 2] = acc_b;
        depth[r] = acc_d;
    }
}

// util.gen_rays + unproj_map (util.py:113-143,238-276, ndc=False)
__global__ void gen_rays_kernel(const float *__restrict__ poses, int NV, int W, int H, float fx, float fy, float cx,
                                float cy, float z_near, float z_far, float *__restrict__ rays) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)NV * H * W) return;
    const int px = (int)(idx % W), py = (int)((idx / W) % H), n = (int)(idx / ((long long)W * H));
    const float X = ((float)px - cx) / fx, Y = ((float)py - cy) / fy;
    float d0 = X, d1 = -Y, d2 = -1.f;
    const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 /= nrm; d1 /= nrm; d2 /= nrm;
    const float *P = poses + (size_t)n * 16;
    float *o = rays + idx * 8;
    o[0] = P[3]; o[1] = P[7]; o[2] = P[11];
    o[3] = P[0] * d0 + P[1] * d1 + P[2] * d2;
    o[4] = P[4] * d0 + P[5] * d1 + P[6] * d2;
    o[5] = P[8] * d0 + P[9] * d1 + P[10] * d2;
    o[6] = z_near; o[7] = z_far;
}
#pragma clang fp contract(fast)

// the counter-based draws written out as tensors (the same functions the sampling kernels call): lets a caller -- and
// the tests -- feed the explicit-noise interface with exactly what the seeded interface would draw
__global__ void philox_fill_kernel(const NoiseSrc ns, int R, int Kc, int Kimp, int Kfd, float *__restrict__ u1,
                                   float *__restrict__ u2, float *__restrict__ u3, float *__restrict__ n4) {
    const int per = Kc + 2 * Kimp + Kfd;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * per) return;
    const long long r = idx / per;
    int i = (int)(idx % per);
    if (i < Kc) { u1[r * Kc + i] = gen_uniform(ns, DRAW_U1, r, i); return; }
    i -= Kc;
    if (i < Kimp) { u2[r * Kimp + i] = gen_uniform(ns, DRAW_U2, r, i); return; }
    i -= Kimp;
    if (i < Kimp) { u3[r * Kimp + i] = gen_uniform(ns, DRAW_U3, r, i); return; }
    i -= Kimp;
    n4[r * Kfd + i] = gen_normal(ns, r, i);
}

}  // namespace pnr

using namespace pnr;

static RaySrc explicit_rays(const float *rays) {
    RaySrc s = {};
    s.rays = rays;
    return s;
}
static NoiseSrc explicit_noise(const float *u1, const float *u2, const float *u3, const float *n4) {
    NoiseSrc n = {};
    n.u1 = u1; n.u2 = u2; n.u3 = u3; n.n4 = n4;
    n.id_stride = 1; n.per_obj = 1;
    return n;
}
static NoiseSrc seeded_noise(unsigned long long seed, long long id_offset, int id_stride, int rays_per_obj) {
    NoiseSrc n = {};
    n.seed_lo = (uint32_t)seed; n.seed_hi = (uint32_t)(seed >> 32);
    n.id_offset = id_offset;
    n.per_obj = rays_per_obj > 0 ? rays_per_obj : 1;
    n.id_stride = id_stride > 0 ? id_stride : n.per_obj;
    return n;
}

static int sample_coarse_src(const RaySrc &rs, const NoiseSrc &ns, int R, int Kc, int lindisp, float *z, void *stream) {
    const long long n = (long long)R * Kc;
    hipLaunchKernelGGL(sample_coarse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rs, ns, R, Kc, lindisp, z);
    return pnr_check_launch("pnr_sample_coarse");
}

extern "C" int pnr_sample_coarse(const float *rays, const float *u1, int R, int Kc, int lindisp, float *z,
                                 void *stream) {
    if (R < 0 || Kc <= 0) return pnr_fail(PNR_E_INVALID, "pnr_sample_coarse: bad sizes");
    if (R == 0) return PNR_OK;
    if (!rays || !u1 || !z) return pnr_fail(PNR_E_INVALID, "pnr_sample_coarse: null argument");
    return sample_coarse_src(explicit_rays(rays), explicit_noise(u1, nullptr, nullptr, nullptr), R, Kc, lindisp, z, stream);
}

static int sample_fine_src(const RaySrc &rs, const float *weights_c, const float *depth_c, const float *z_coarse,
                           const NoiseSrc &ns, int R, int Kc, int Kimp, int Kfd, float depth_std, int lindisp, float *z_sorted,
                           int32_t *depth_ranks, float *z_new, int32_t *ranks_all, void *stream) {
    if (R < 0 || Kc <= 0 || Kimp < 0 || Kfd < 0) return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: bad sizes");
    const size_t lds = (size_t)WAVES_PER_BLOCK * (2 * (size_t)Kc + 1 + Kimp + Kfd) * sizeof(float);
    if (lds > (size_t)SAMPLE_FINE_LDS_MAX)
        return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: 2 n_coarse + n_fine + 1 must stay below 10240 samples per ray (the ray's cdf and sample set live in LDS)");
    if (R == 0) return PNR_OK;
    if (!z_coarse || !z_sorted || (Kimp > 0 && !weights_c) || (Kfd > 0 && !depth_c))
        return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: null argument");
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sample_fine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(sample_fine_kernel)");
    }
    hipLaunchKernelGGL(sample_fine_kernel, dim3((R + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(WAVES_PER_BLOCK * 64),
                       lds, (hipStream_t)stream, rs, weights_c, depth_c, z_coarse, ns, R, Kc, Kimp, Kfd, depth_std, lindisp,
                       z_sorted, depth_ranks, z_new, ranks_all);
    return pnr_check_launch("pnr_sample_fine");
}

extern "C" int pnr_sample_fine(const float *rays, const float *weights_c, const float *depth_c, const float *z_coarse,
                               const float *u2, const float *u3, const float *n4, int R, int Kc, int Kimp, int Kfd,
                               float depth_std, int lindisp, float *z_sorted, int32_t *depth_ranks, void *stream) {
    if (R < 0 || Kc <= 0 || Kimp < 0 || Kfd < 0) return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: bad sizes");
    if ((size_t)WAVES_PER_BLOCK * (2 * (size_t)Kc + 1 + Kimp + Kfd) * sizeof(float) > (size_t)SAMPLE_FINE_LDS_MAX)
        return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: 2 n_coarse + n_fine + 1 must stay below 10240 samples per ray (the ray's cdf and sample set live in LDS)");
    if (R > 0 && (!rays || (Kimp > 0 && (!u2 || !u3)) || (Kfd > 0 && !n4)))
        return pnr_fail(PNR_E_INVALID, "pnr_sample_fine: null argument");
    return sample_fine_src(explicit_rays(rays), weights_c, depth_c, z_coarse, explicit_noise(nullptr, u2, u3, n4), R, Kc, Kimp,
                           Kfd, depth_std, lindisp, z_sorted, depth_ranks, nullptr, nullptr, stream);
}

static int composite_src(const RaySrc &rs, const float *z, const float *rgbsigma, int R, int K, int white_bkgd, float *weights,
                         float *rgb, float *depth, void *stream) {
    hipLaunchKernelGGL(composite_kernel, dim3((R + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK), dim3(WAVES_PER_BLOCK * 64), 0,
                       (hipStream_t)stream, rs, z, (const float4 *)rgbsigma, R, K, white_bkgd, weights, rgb, depth);
    return pnr_check_launch("pnr_composite");
}

extern "C" int pnr_composite(const float *rays, const float *z, const float *rgbsigma, int R, int K, int white_bkgd,
                             float *weights, float *rgb, float *depth, void *stream) {
    if (R < 0 || K <= 0) return pnr_fail(PNR_E_INVALID, "pnr_composite: bad sizes");
    if (R == 0) return PNR_OK;
    if (!rays || !z || !rgbsigma || !rgb || !depth) return pnr_fail(PNR_E_INVALID, "pnr_composite: null argument");
    return composite_src(explicit_rays(rays), z, rgbsigma, R, K, white_bkgd, weights, rgb, depth, stream);
}

extern "C" int pnr_gen_rays(const float *poses, int NV, int W, int H, float fx, float fy, float cx, float cy,
                            float z_near, float z_far, float *rays, void *stream) {
    if (NV < 0 || W <= 0 || H <= 0) return pnr_fail(PNR_E_INVALID, "pnr_gen_rays: bad sizes");
    if (NV == 0) return PNR_OK;
    if (!poses || !rays) return pnr_fail(PNR_E_INVALID, "pnr_gen_rays: null argument");
    const long long n = (long long)NV * W * H;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, poses, NV,
                       W, H, fx, fy, cx, cy, z_near, z_far, rays);
    return pnr_check_launch("pnr_gen_rays");
}

extern "C" int pnr_philox_noise(unsigned long long seed, long long id_offset, int id_stride, int rays_per_obj, int R, int Kc,
                                int Kimp, int Kfd, float *u1, float *u2, float *u3, float *n4, void *stream) {
    if (R < 0 || Kc < 0 || Kimp < 0 || Kfd < 0) return pnr_fail(PNR_E_INVALID, "pnr_philox_noise: bad sizes");
    if ((Kc > 0 && !u1) || (Kimp > 0 && (!u2 || !u3)) || (Kfd > 0 && !n4)) return pnr_fail(PNR_E_INVALID, "pnr_philox_noise: null output");
    const long long n = (long long)R * (Kc + 2 * Kimp + Kfd);
    if (n == 0) return PNR_OK;
    hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       seeded_noise(seed, id_offset, id_stride, rays_per_obj), R, Kc, Kimp, Kfd, u1, u2, u3, n4);
    return pnr_check_launch("pnr_philox_noise");
}

extern "C" int pnr_philox_raw(const uint32_t *counter4 /*host*/, const uint32_t *key2 /*host*/, uint32_t *out4 /*host*/) {
    if (!counter4 || !key2 || !out4) return pnr_fail(PNR_E_INVALID, "pnr_philox_raw: null argument");
    const U4 c = {counter4[0], counter4[1], counter4[2], counter4[3]};
    const U4 r = philox4x32_10(c, key2[0], key2[1]);  // the same inline function the kernels call, compiled for the host
    out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
    return PNR_OK;
}

// workspace layout (floats): z_c [R*Kc] | rgbs_c [R*Kc*4] | w_c [R*Kc] | z_f [R*Kt] | rgbs_f [R*Kt*4]
//                            | z_new [R*Kf] | rgbs_new [R*Kf*4] | ranks [R*Kt]   (coarse-network reuse, mlp_fine == NULL)
static size_t align64(size_t n) { return (n + 63) & ~(size_t)63; }

extern "C" size_t pnr_render_workspace_bytes(int R, int Kc, int Kf) {
    if (R <= 0 || Kc <= 0 || Kf < 0) return 0;
    const size_t r = (size_t)R, kc = (size_t)Kc, kt = (size_t)(Kc + Kf);
    size_t fl = align64(r * kc) + align64(r * kc * 4) + align64(r * kc);
    if (Kf > 0) fl += align64(r * kt) + align64(r * kt * 4) + align64(r * (kt - kc)) + align64(r * (kt - kc) * 4) + align64(r * kt);
    return fl * sizeof(float);
}

// NeRFRenderer.forward (nerf.py:251-303) for any ray source / noise source
static int render_impl(const PnrScene *scene, const void *packed_coarse, const void *tables_coarse, const void *packed_fine,
                       const void *tables_fine, int precision, const RaySrc &rs, const NoiseSrc &ns, int R, int rays_per_obj,
                       int Kc, int Kf, int Kfd, float depth_std, int white_bkgd, int lindisp, float *rgb_c, float *depth_c,
                       float *weights_c, float *rgb_f, float *depth_f, float *weights_f, void *workspace, void *stream) {
    if (R < 0 || Kc <= 0 || Kf < 0 || Kfd < 0 || Kfd > Kf)
        return pnr_fail(PNR_E_INVALID, "pnr_render_forward: bad sample counts");
    if (R == 0) return PNR_OK;
    if (!workspace || !rgb_c || !depth_c || (Kf > 0 && (!rgb_f || !depth_f)))
        return pnr_fail(PNR_E_INVALID, "pnr_render_forward: null output / workspace");
    if (!rs.rays && !rs.poses) return pnr_fail(PNR_E_INVALID, "pnr_render_forward: null rays");
    const bool gen = !ns.u1 && !ns.u2 && !ns.u3 && !ns.n4;
    if (!gen && (!ns.u1 || (Kf - Kfd > 0 && (!ns.u2 || !ns.u3)) || (Kfd > 0 && !ns.n4)))
        return pnr_fail(PNR_E_INVALID, "pnr_render_forward: explicit noise needs u1 [, u2, u3] [, n4] (pass none of them for seeded draws)");
    hipStream_t st = (hipStream_t)stream;
    const size_t r = (size_t)R, kc = (size_t)Kc, kt = (size_t)(Kc + Kf);
    float *ws = (float *)workspace;
    float *z_c = ws; ws += align64(r * kc);
    float *rgbs_c = ws; ws += align64(r * kc * 4);
    float *w_c = ws; ws += align64(r * kc);
    float *z_f = ws; ws += align64(r * kt);
    float *rgbs_f = ws; ws += align64(r * kt * 4);
    float *z_new = ws; ws += align64(r * (kt - kc));
    float *rgbs_new = ws; ws += align64(r * (kt - kc) * 4);
    int32_t *ranks = (int32_t *)ws;
    if (weights_c) w_c = weights_c;  // write straight into the caller's buffer
    int rc;
    if ((rc = sample_coarse_src(rs, ns, R, Kc, lindisp, z_c, stream))) return rc;
    saturation_guard_slot(0);  // (fp16-range guard, when armed: the coarse network reports into word 0, a fine network into word 1)
    if ((rc = eval_samples_src(scene, packed_coarse, tables_coarse, precision, rs, z_c, R, rays_per_obj, Kc, rgbs_c, st))) return rc;
    if ((rc = composite_src(rs, z_c, rgbs_c, R, Kc, white_bkgd, w_c, rgb_c, depth_c, stream))) return rc;
    if (Kf > 0) {
        if (packed_fine) {
            if ((rc = sample_fine_src(rs, w_c, depth_c, z_c, ns, R, Kc, Kf - Kfd, Kfd, depth_std, lindisp, z_f, nullptr, nullptr,
                                      nullptr, stream))) return rc;
            saturation_guard_slot(1);
            rc = eval_samples_src(scene, packed_fine, tables_fine, precision, rs, z_f, R, rays_per_obj, Kc + Kf, rgbs_f, st);
            saturation_guard_slot(0);
            if (rc) return rc;
        } else {
            // mlp_fine is None (models.py:242, eval/eval.py:140): the fine pass runs the coarse network on the merged
            // samples, Kc of which it has just evaluated -- evaluate the Kf new ones only and merge in sorted order
            if ((rc = sample_fine_src(rs, w_c, depth_c, z_c, ns, R, Kc, Kf - Kfd, Kfd, depth_std, lindisp, z_f, nullptr, z_new,
                                      ranks, stream))) return rc;
            if ((rc = eval_samples_src(scene, packed_coarse, tables_coarse, precision, rs, z_new, R, rays_per_obj, Kf, rgbs_new, st))) return rc;
            const long long n = (long long)R * (Kc + Kf);
            hipLaunchKernelGGL(merge_rgbsigma_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const float4 *)rgbs_c, (const float4 *)rgbs_new, ranks, R, Kc, Kf, (float4 *)rgbs_f);
            if ((rc = pnr_check_launch("merge_rgbsigma_kernel"))) return rc;
        }
        if ((rc = composite_src(rs, z_f, rgbs_f, R, Kc + Kf, white_bkgd, weights_f, rgb_f, depth_f, stream))) return rc;
    }
    return PNR_OK;
}

extern "C" int pnr_render_forward(const PnrScene *scene, const void *packed_coarse, const void *packed_fine,
                                  int precision, const float *rays, int R, int rays_per_obj, int Kc, int Kf, int Kfd,
                                  float depth_std, int white_bkgd, int lindisp, const float *u1, const float *u2,
                                  const float *u3, const float *n4, float *rgb_c, float *depth_c, float *weights_c,
                                  float *rgb_f, float *depth_f, float *weights_f, void *workspace, void *stream) {
    if (R > 0 && !u1) return pnr_fail(PNR_E_INVALID, "pnr_render_forward: null u1 (pnr_render_forward_seeded draws in-kernel)");
    return render_impl(scene, packed_coarse, nullptr, packed_fine, nullptr, precision, explicit_rays(rays),
                       explicit_noise(u1, u2, u3, n4), R, rays_per_obj, Kc, Kf, Kfd, depth_std, white_bkgd, lindisp, rgb_c, depth_c,
                       weights_c, rgb_f, depth_f, weights_f, workspace, stream);
}

extern "C" int pnr_render_forward_folded(const PnrScene *scene, const void *packed_coarse, const void *tables_coarse,
                                         const void *packed_fine, const void *tables_fine, int precision, const float *rays,
                                         int R, int rays_per_obj, int Kc, int Kf, int Kfd, float depth_std, int white_bkgd,
                                         int lindisp, const float *u1, const float *u2, const float *u3, const float *n4,
                                         float *rgb_c, float *depth_c, float *weights_c, float *rgb_f, float *depth_f,
                                         float *weights_f, void *workspace, void *stream) {
    if (!tables_coarse || (packed_fine && !tables_fine))
        return pnr_fail(PNR_E_INVALID, "pnr_render_forward_folded: every folded network needs its tables");
    if (R > 0 && !u1) return pnr_fail(PNR_E_INVALID, "pnr_render_forward_folded: null u1 (pnr_render_forward_seeded draws in-kernel)");
    return render_impl(scene, packed_coarse, tables_coarse, packed_fine, tables_fine, precision, explicit_rays(rays),
                       explicit_noise(u1, u2, u3, n4), R, rays_per_obj, Kc, Kf, Kfd, depth_std, white_bkgd, lindisp, rgb_c, depth_c,
                       weights_c, rgb_f, depth_f, weights_f, workspace, stream);
}

extern "C" int pnr_render_forward_seeded(const PnrScene *scene, const void *packed_coarse, const void *tables_coarse,
                                         const void *packed_fine, const void *tables_fine, int precision, const float *rays,
                                         int R, int rays_per_obj, int Kc, int Kf, int Kfd, float depth_std, int white_bkgd,
                                         int lindisp, unsigned long long seed, long long ray_id_offset, int ray_id_stride,
                                         float *rgb_c, float *depth_c, float *weights_c, float *rgb_f, float *depth_f,
                                         float *weights_f, void *workspace, void *stream) {
    if ((tables_coarse == nullptr) != (tables_fine == nullptr) && packed_fine)
        return pnr_fail(PNR_E_INVALID, "pnr_render_forward_seeded: both networks folded or neither");
    return render_impl(scene, packed_coarse, tables_coarse, packed_fine, tables_fine, precision, explicit_rays(rays),
                       seeded_noise(seed, ray_id_offset, ray_id_stride, rays_per_obj), R, rays_per_obj, Kc, Kf, Kfd, depth_std,
                       white_bkgd, lindisp, rgb_c, depth_c, weights_c, rgb_f, depth_f, weights_f, workspace, stream);
}

// util.gen_rays + NeRFRenderer.forward for whole target views (eval/eval.py:247-279): the rays are never materialised
extern "C" size_t pnr_render_views_workspace_bytes(int NV, int W, int H, int Kc, int Kf) {
    if (NV <= 0 || W <= 0 || H <= 0) return 0;
    const long long R = (long long)NV * W * H;
    if (R > 0x7fffffffLL) return 0;
    return pnr_render_workspace_bytes((int)R, Kc, Kf);
}

extern "C" int pnr_render_views(const PnrScene *scene, const void *packed_coarse, const void *tables_coarse,
                                const void *packed_fine, const void *tables_fine, int precision, const float *poses_c2w, int NV,
                                int W, int H, float fx, float fy, float cx, float cy, float z_near, float z_far, int Kc, int Kf,
                                int Kfd, float depth_std, int white_bkgd, int lindisp, const float *u1, const float *u2,
                                const float *u3, const float *n4, unsigned long long seed, float *rgb_c, float *depth_c,
                                float *weights_c, float *rgb_f, float *depth_f, float *weights_f, void *workspace, void *stream) {
    if (!scene || NV < 0 || W <= 0 || H <= 0) return pnr_fail(PNR_E_INVALID, "pnr_render_views: bad sizes");
    if (NV == 0) return PNR_OK;
    if (!poses_c2w) return pnr_fail(PNR_E_INVALID, "pnr_render_views: null poses");
    if (scene->SB <= 0 || NV % scene->SB != 0) return pnr_fail(PNR_E_INVALID, "pnr_render_views: NV must be a multiple of SB (views grouped per object)");
    const long long R = (long long)NV * W * H;
    if (R > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_render_views: too many rays for one call");
    if (packed_fine && ((tables_coarse == nullptr) != (tables_fine == nullptr)))
        return pnr_fail(PNR_E_INVALID, "pnr_render_views: both networks folded or neither");
    RaySrc rs = {};
    rs.poses = poses_c2w; rs.W = W; rs.H = H; rs.fx = fx; rs.fy = fy; rs.cx = cx; rs.cy = cy; rs.z_near = z_near; rs.z_far = z_far;
    const int per_obj = (int)(R / scene->SB);
    const NoiseSrc ns = (u1 || u2 || u3 || n4) ? explicit_noise(u1, u2, u3, n4) : seeded_noise(seed, 0, per_obj, per_obj);
    return render_impl(scene, packed_coarse, tables_coarse, packed_fine, tables_fine, precision, rs, ns, (int)R, per_obj, Kc, Kf, Kfd,
                       depth_std, white_bkgd, lindisp, rgb_c, depth_c, weights_c, rgb_f, depth_f, weights_f, workspace, stream);
}
