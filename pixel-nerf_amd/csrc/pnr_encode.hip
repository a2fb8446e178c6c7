// pnr_encode.hip -- the neighbours of the hot path: encoder output formatting in front
// of it (rank 2), whole-view rendering from camera poses (rank 1), the eval epilogue behind it (rank 3).
//
// Encoder output formatting, src/model/encoder.py:150-163.  The reference upsamples every ResNet stage to the
// first stage's resolution (F.interpolate, bilinear, align_corners=True), concatenates them on the
// channel axis into `latent` (NV,512,Hl,Wl) NCHW, and the lookup later reads that tensor with a
// 512-way strided access.  Here ONE pass reads the pyramid and writes the grid channel-last (the
// layout the fused kernel gathers from: one bilinear corner = one contiguous 2 KiB row) and, when
// the caller wants the reference's NCHW tensor too (net.encoder.latent stays readable), that as
// well -- no separate interpolate / cat / transpose passes over up to 176 MiB.
//
// HBM-bound: algorithmic bytes = pyramid read once + 4*512 B per output pixel per written layout.
// 64-channel x 32-pixel x 4-row tiles; interpolation with lanes along x (coalesced source rows and NCHW
// stores), transposed through LDS, stored with lanes along channels (16-byte NHWC stores).
#include <hip/hip_runtime.h>

#include "pnr_common.h"

namespace pnr {

constexpr int MAX_STAGES = 5;  // encoder.py:68 num_layers <= 5
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Pyramid {
    const float *src[MAX_STAGES];
    int c_begin[MAX_STAGES + 1];  // first output channel of stage s; [n] = total
    int H[MAX_STAGES], W[MAX_STAGES];
    int n;
};

constexpr int FT_C = 64;  // channels per tile
constexpr int FT_X = 32;  // pixels (along x) per tile
constexpr int FT_Y = 4;   // rows per workgroup

// ATen upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1); i0 = (int)src;
// i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1;
// val = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)     (no FMA contraction)
// Workgroup = 256 threads: FT_C channels x FT_X pixels x FT_Y rows.  Phase 1 (lanes along x): interpolate,
// store NCHW, park in LDS; phase 2 (lanes along channels): 16-byte NHWC stores, 256 B contiguous per pixel.
__global__ void __launch_bounds__(256)
pyramid_to_latent_kernel(const Pyramid p, int NV, int H0, int W0, float *__restrict__ nhwc, float *__restrict__ nchw) {
#pragma clang fp contract(off)
    __shared__ float tile[FT_C][FT_X + 1];
    const int t = threadIdx.x;
    const int tx = t & 31, ty = t >> 5;  // phase 1: x, channel group (8)
    const int Ctot = p.c_begin[p.n];
    const int ctiles = Ctot / FT_C;
    const int n = blockIdx.z / ctiles, c0 = (blockIdx.z % ctiles) * FT_C;
    const int x0 = blockIdx.x * FT_X;
    int s = 0;
    while (s + 1 < p.n && c0 >= p.c_begin[s + 1]) ++s;
    const int Hs = p.H[s], Ws = p.W[s], Cs = p.c_begin[s + 1] - p.c_begin[s];
    const float sy = H0 > 1 ? (float)(Hs - 1) / (float)(H0 - 1) : 0.f;
    const float sx = W0 > 1 ? (float)(Ws - 1) / (float)(W0 - 1) : 0.f;
    const int x = x0 + tx;
    const float fx = sx * (float)x;
    const int xa = min((int)fx, Ws - 1), xb = xa + (xa < Ws - 1 ? 1 : 0);
    const float w1 = fx - (float)xa, w0 = 1.f - w1;
    const int xx2 = t >> 4, c4 = (t & 15) * 4;  // phase 2: pixel (16 per pass), 4 channels
    for (int yy = 0; yy < FT_Y; ++yy) {
        const int y = blockIdx.y * FT_Y + yy;
        if (y >= H0) break;
        const float fy = sy * (float)y;
        const int y0 = (int)fy, y1 = y0 + (y0 < Hs - 1 ? 1 : 0);
        const float h1 = fy - (float)y0, h0 = 1.f - h1;
        if (x < W0) {
#pragma unroll
            for (int cc = ty; cc < FT_C; cc += 8) {
                const int c = c0 + cc;
                const float *plane = p.src[s] + ((size_t)n * Cs + (c - p.c_begin[s])) * Hs * Ws;
                const float *r0 = plane + (size_t)y0 * Ws, *r1 = plane + (size_t)y1 * Ws;
                const float v = h0 * (w0 * r0[xa] + w1 * r0[xb]) + h1 * (w0 * r1[xa] + w1 * r1[xb]);
                tile[cc][tx] = v;
                if (nchw) nchw[(((size_t)n * Ctot + c) * H0 + y) * W0 + x] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int xx = xx2; xx < FT_X; xx += 16)
            if (x0 + xx < W0) {
                const f32x4_t v = {tile[c4][xx], tile[c4 + 1][xx], tile[c4 + 2][xx], tile[c4 + 3][xx]};
                *reinterpret_cast<f32x4_t *>(nhwc + (((size_t)n * H0 + y) * W0 + x0 + xx) * Ctot + c0 + c4) = v;
            }
        __syncthreads();
    }
}

// eval/eval.py:283-290,327-329 + util.psnr (util.py:474-481): one workgroup per view.  rgb -> clamp[0,1]
// (-> uint8 by truncation of x*255, numpy .astype(np.uint8)); depth -> (d - near)/(far - near); sum of
// squared error of the clamped image against ground truth, accumulated in fp64 in a fixed order.
__global__ void __launch_bounds__(256)
eval_epilogue_kernel(const float *__restrict__ rgb, const float *__restrict__ depth, int pixels, float z_near, float z_far,
                     const float *__restrict__ gt, unsigned char *__restrict__ u8, float *__restrict__ clamped,
                     float *__restrict__ depth_norm, double *__restrict__ sse) {
#pragma clang fp contract(off)
    __shared__ double part[256];
    const int v = blockIdx.x, t = threadIdx.x;
    const size_t base = (size_t)v * pixels;
    double acc = 0.0;
    for (int i = t; i < pixels * 3; i += 256) {
        const float c = fminf(fmaxf(rgb[base * 3 + i], 0.f), 1.f);
        if (clamped) clamped[base * 3 + i] = c;
        if (u8) u8[base * 3 + i] = (unsigned char)(c * 255.f);
        if (gt) {
            const double d = (double)c - (double)gt[base * 3 + i];
            acc += d * d;
        }
    }
    if (depth_norm && depth)
        for (int i = t; i < pixels; i += 256) depth_norm[base + i] = (depth[base + i] - z_near) / (z_far - z_near);
    if (sse) {
        part[t] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (t < s) part[t] += part[t + s];
            __syncthreads();
        }
        if (t == 0) sse[v] = part[0];
    }
}

// train/train.py:143-182 + util.bbox_sample (util.py:220-235): one thread per selected pixel; the ray is
// built from (pose, intrinsics, pixel) exactly as gen_rays_kernel does, the colour is images*0.5+0.5.
__global__ void sample_training_rays_kernel(const float *__restrict__ poses, const float *__restrict__ images,
                                            const float *__restrict__ focal, const float *__restrict__ c,
                                            const float *__restrict__ bboxes, const long long *__restrict__ ids,
                                            const float *__restrict__ ux, const float *__restrict__ uy, int SB, int NV, int W,
                                            int H, int B, float z_near, float z_far, float *__restrict__ rays,
                                            float *__restrict__ rgb_gt) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SB * B) return;
    const int obj = idx / B;
    long long img, px, py;
    if (bboxes) {  // util.bbox_sample: x = long(u * (x1 + 1 - x0) + x0)
        img = ids[idx];
        const float *bb = bboxes + ((size_t)obj * NV + img) * 4;
        px = (long long)(ux[idx] * (bb[2] + 1.f - bb[0]) + bb[0]);
        py = (long long)(uy[idx] * (bb[3] + 1.f - bb[1]) + bb[1]);
    } else {  // flat index into (NV, H, W)
        const long long f = ids[idx];
        img = f / ((long long)H * W); py = (f / W) % H; px = f % W;
    }
    img = min(max(img, 0LL), (long long)NV - 1); px = min(max(px, 0LL), (long long)W - 1); py = min(max(py, 0LL), (long long)H - 1);
    const float fx = focal[obj * 2], fy = focal[obj * 2 + 1];
    const float cx = c ? c[obj * 2] : (float)W * 0.5f, cy = c ? c[obj * 2 + 1] : (float)H * 0.5f;
    const float X = ((float)px - cx) / fx, Y = ((float)py - cy) / fy;
    float d0 = X, d1 = -Y, d2 = -1.f;
    const float nrm = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    d0 /= nrm; d1 /= nrm; d2 /= nrm;
    const float *P = poses + ((size_t)obj * NV + img) * 16;
    float *o = rays + (size_t)idx * 8;
    o[0] = P[3]; o[1] = P[7]; o[2] = P[11];
    o[3] = P[0] * d0 + P[1] * d1 + P[2] * d2;
    o[4] = P[4] * d0 + P[5] * d1 + P[6] * d2;
    o[5] = P[8] * d0 + P[9] * d1 + P[10] * d2;
    o[6] = z_near; o[7] = z_far;
    const float *im = images + (((size_t)obj * NV + img) * 3) * H * W + (size_t)py * W + px;
    float *g = rgb_gt + (size_t)idx * 3;
    g[0] = im[0] * 0.5f + 0.5f; g[1] = im[(size_t)H * W] * 0.5f + 0.5f; g[2] = im[(size_t)2 * H * W] * 0.5f + 0.5f;
}

}  // namespace pnr

extern "C" int pnr_sample_training_rays(const float *poses, const float *images, const float *focal, const float *c,
                                        const float *bboxes, const long long *ids, const float *ux, const float *uy, int SB,
                                        int NV, int W, int H, int B, float z_near, float z_far, float *rays, float *rgb_gt,
                                        void *stream) {
    if (SB < 0 || B < 0 || NV <= 0 || W <= 0 || H <= 0) return pnr_fail(PNR_E_INVALID, "pnr_sample_training_rays: bad sizes");
    if (SB == 0 || B == 0) return PNR_OK;
    if (!poses || !images || !focal || !ids || !rays || !rgb_gt || (bboxes && (!ux || !uy)))
        return pnr_fail(PNR_E_INVALID, "pnr_sample_training_rays: null argument");
    const int n = SB * B;
    hipLaunchKernelGGL(pnr::sample_training_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, poses, images,
                       focal, c, bboxes, ids, ux, uy, SB, NV, W, H, B, z_near, z_far, rays, rgb_gt);
    return pnr_check_launch("pnr_sample_training_rays");
}

extern "C" int pnr_eval_epilogue(const float *rgb, const float *depth, int n_views, int pixels_per_view, float z_near,
                                 float z_far, const float *gt_rgb, unsigned char *rgb_u8, float *rgb_clamped,
                                 float *depth_norm, double *sq_err_sum, void *stream) {
    if (n_views < 0 || pixels_per_view <= 0) return pnr_fail(PNR_E_INVALID, "pnr_eval_epilogue: bad sizes");
    if (n_views == 0) return PNR_OK;
    if (!rgb || (depth_norm && !depth) || (sq_err_sum && !gt_rgb)) return pnr_fail(PNR_E_INVALID, "pnr_eval_epilogue: null argument");
    hipLaunchKernelGGL(pnr::eval_epilogue_kernel, dim3(n_views), dim3(256), 0, (hipStream_t)stream, rgb, depth, pixels_per_view,
                       z_near, z_far, gt_rgb, rgb_u8, rgb_clamped, depth_norm, sq_err_sum);
    return pnr_check_launch("pnr_eval_epilogue");
}

extern "C" int pnr_pyramid_to_latent(const float *const *stages, const int *channels, const int *heights, const int *widths,
                                     int n_stages, int NV, float *latent_nhwc, float *latent_nchw, void *stream) {
    if (n_stages < 1 || n_stages > pnr::MAX_STAGES) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: 1..5 stages");
    if (!stages || !channels || !heights || !widths || !latent_nhwc) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: null argument");
    if (NV < 0) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: bad sizes");
    pnr::Pyramid p = {};
    p.n = n_stages;
    int c = 0;
    for (int s = 0; s < n_stages; ++s) {
        if (!stages[s] || channels[s] <= 0 || channels[s] % pnr::FT_C != 0 || heights[s] <= 0 || widths[s] <= 0)
            return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: stage channels must be positive multiples of 64, sizes positive");
        p.src[s] = stages[s]; p.c_begin[s] = c; p.H[s] = heights[s]; p.W[s] = widths[s];
        c += channels[s];
    }
    p.c_begin[n_stages] = c;
    if (NV == 0) return PNR_OK;
    const int H0 = heights[0], W0 = widths[0];
    if ((long long)NV * (c / pnr::FT_C) > 65535) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: grid too large");
    dim3 grid((W0 + pnr::FT_X - 1) / pnr::FT_X, (H0 + pnr::FT_Y - 1) / pnr::FT_Y, NV * (c / pnr::FT_C));
    hipLaunchKernelGGL(pnr::pyramid_to_latent_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, NV, H0, W0, latent_nhwc,
                       latent_nchw);
    return pnr_check_launch("pnr_pyramid_to_latent");
}
