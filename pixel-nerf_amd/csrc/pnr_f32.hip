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
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_device.h"
#include "pnr_layout.h"

namespace pnr {

constexpr int FW = 4;  // wavefronts per block in the feature kernel

// one wavefront per (view, point) of the chunk [p0, p0+np)
template <bool RAYS>
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
        v = rem < 3 ? sinf(a) : sinf(a + 1.57079637050628662109375f);
    } else if (lane < 42) {
        const int c = lane - 39;
        v = pose[4 * c + 0] * dx + pose[4 * c + 1] * dy + pose[4 * c + 2] * dz;
    }
    in42[(size_t)idx * D_IN_PAD + lane] = v;
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
    *reinterpret_cast<f32x4 *>(dst) = acc0;
    *reinterpret_cast<f32x4 *>(dst + 4) = acc1;
}

// Y (M,N) = [Y +] [relu](X (M,K)) W^T (W is (N,K)) + b.   Block 256 threads = 4 waves, 64x64 tile,
// K in chunks of 32 through LDS, v_mfma_f32_32x32x2_f32 (A: X[i][k], B: W[j][k], one float each).
__global__ void __launch_bounds__(256)
linear_f32_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, const float *__restrict__ b,
                  float *__restrict__ Y, int ldy, long long M, int N, int K, int relu_in, int accumulate) {
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
        // stage 64x32 of X and of W: 2048 floats each, 8 per thread
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = t + u * 256, row = e >> 5, col = e & 31;
            float xv = 0.f, wv_ = 0.f;
            if (m0 + row < M && k0 + col < K) xv = X[(m0 + row) * ldx + k0 + col];
            if (n0 + row < N && k0 + col < K) wv_ = W[(size_t)(n0 + row) * K + k0 + col];
            sX[row][col] = relu_in ? fmaxf(xv, 0.f) : xv;
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
                float *y = Y + m * ldy + n;
                const float v = acc[r] + bias;
                *y = accumulate ? *y + v : v;
            }
        }
    }
}

// xp[p][f] = mean_v x[v*np + p][f]
__global__ void pool_f32_kernel(const float *__restrict__ x, float *__restrict__ xp, int np, int NS) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)np * D_HID) return;
    float s = 0.f;
    for (int v = 0; v < NS; ++v) s += x[(size_t)v * np * D_HID + idx];
    xp[idx] = s / (float)NS;
}

// xp[(o*B + p)][f] = mean_v x[((o*NS + v)*B + p)][f] : util.combine_interleaved with inner dims (NS, B) (util.py:461-471)
__global__ void pool_interleaved_f32_kernel(const float *__restrict__ x, float *__restrict__ xp, long long groups, int NS, int B) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= groups * B * D_HID) return;
    const long long o = idx / ((long long)B * D_HID), rem = idx - o * (long long)B * D_HID;
    float s = 0.f;
    for (int v = 0; v < NS; ++v) s += x[((size_t)o * NS + v) * (size_t)B * D_HID + rem];
    xp[idx] = s / (float)NS;
}

__global__ void out_f32_kernel(const float *__restrict__ o, float *__restrict__ rgbs, int np) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np) return;
    const f32x4 s = *reinterpret_cast<const f32x4 *>(o + (size_t)idx * 4);
    f32x4 r = {1.f / (1.f + expf(-s[0])), 1.f / (1.f + expf(-s[1])), 1.f / (1.f + expf(-s[2])), fmaxf(s[3], 0.f)};
    *reinterpret_cast<f32x4 *>(rgbs + (size_t)idx * 4) = r;
}

static void linear(hipStream_t st, const float *X, int ldx, const float *W, const float *b, float *Y, int ldy, long long M,
                   int N, int K, bool relu_in, bool accumulate) {
    dim3 grid((unsigned)((M + 63) / 64), (N + 63) / 64);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, st, X, ldx, W, b, Y, ldy, M, N, K, relu_in ? 1 : 0,
                       accumulate ? 1 : 0);
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
            hipLaunchKernelGGL(pool_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xp, np, NS);
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

}  // namespace pnr

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
        hipLaunchKernelGGL(pool_interleaved_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xp, np / B, NS, B);
        xs = xp; ns = netp;
    }
    for (int b = COMBINE_LAYER; b < N_BLOCKS; ++b) {
        linear(st, xs, D_HID, w->fc0_w[b], w->fc0_b[b], ns, D_HID, np, D_HID, D_HID, true, false);
        linear(st, ns, D_HID, w->fc1_w[b], w->fc1_b[b], xs, D_HID, np, D_HID, D_HID, true, true);
    }
    linear(st, xs, D_HID, w->lin_out_w, w->lin_out_b, out, D_OUT, np, D_OUT, D_HID, true, false);       // :183 (raw output)
    return pnr_check_launch("pnr_resnetfc_forward_f32");
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
