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
// HBM-bound: algorithmic bytes = pyramid read once + 4*C B per output pixel per written layout.  Round 2's kernel fetched the
// four corners of every output value straight from global memory: 16 B per value through the CU's 64 B/clk vector-memory return
// path, i.e. as many cycles there as the output stream takes on the HBM side (0.38 of the HBM roofline).  Here a workgroup owns
// a run of FT_P pixels of one image row x ALL channels and first copies the few source texels the run touches -- for an
// r-times upsampled stage 2 rows x (FT_P / r + 2) texels per channel -- into LDS, channel-last; the interpolation then reads
// LDS: lanes along the channels for the channel-last grid (four 16-byte LDS reads and one 16-byte global store per 4 values:
// every wave instruction writes 1 KiB of ONE pixel's row), and, when the reference's NCHW tensor is wanted too, a second sweep
// with lanes along the pixels (128-byte runs of a channel plane).
#include <hip/hip_runtime.h>

#include "pnr_common.h"

namespace pnr {

constexpr int MAX_STAGES = 5;  // encoder.py:68 num_layers <= 5
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Pyramid {
    const float *src[MAX_STAGES];
    int c_begin[MAX_STAGES + 1];  // first output channel of stage s; [n] = total
    int H[MAX_STAGES], W[MAX_STAGES];
    float sy[MAX_STAGES], sx[MAX_STAGES];  // (in - 1) / (out - 1), 0 when out == 1
    int same[MAX_STAGES];         // stage at the output resolution: its texels ARE the values
    int nxw[MAX_STAGES];          // texels per window row the LDS layout reserves (>= what any run touches)
    int wsh[MAX_STAGES];          // log2 of the lanes a window row is loaded with (2^wsh >= nxw)
    int win_off[MAX_STAGES + 1];  // float offset of stage s's window in LDS; [n] = total
    int n;
};

constexpr int FT_C = 64;   // stage channel counts are multiples of this
constexpr int FT_PAD = 4;  // floats of padding per window texel: neighbouring texels sit 4 banks apart (16-byte reads of 8 lanes cover the 32 banks)
constexpr int FT_U = 8;    // channel steps per thread and stage prefetched in registers (x 2 window rows); covers the shipped shapes

static float axis_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

// texels of a window row that P output pixels can touch: xa(last) - xa(first) <= ceil(sx (P-1)) + 1 (float rounding), + the
// right neighbour, + 1 for the count
static int window_texels(int P, int Ws, int W0) {
    const int n = (int)ceilf(axis_scale(Ws, W0) * (float)(P - 1)) + 3;
    return n < Ws ? n : Ws;
}

// ATen upsample_bilinear2d, align_corners=True: src = dst * (in-1)/(out-1); i0 = (int)src;
// i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1;
// val = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)     (no FMA contraction)
// A stage at the output resolution (the first one) has src == dst, l1 == 0: the value is the texel itself (finite inputs).
struct StageRun {  // what a run of pixels needs of one stage; every field is uniform over the workgroup
    int cb, Cs, stride, Ws, nxw, woff, y0, y1, xa0, nx;
    float h0, h1, sx;
    bool same;
};

// One workgroup = one run of P pixels of an image row x all channels:
//   1. the source texels under the run go global -> LDS, [row][texel][channel]; a window row is loaded with 2^wsh lanes along x so
//      that the walk over (channel, row) is a pointer increment;
//   2. channel-last sweep: a thread keeps one group of 4 channels and walks the pixels: four 16-byte LDS reads, the blend, one
//      16-byte store -- a wave instruction writes 1 KiB of one pixel's row;
//   3. NCHW sweep (optional): lanes along the pixels, 128-byte runs of a channel plane.
// Instruction count is what this kernel lives on (rocprofv3: the first LDS-window form issued 1700 VALU + 900 SALU instructions
// per wave, 60 % of its cycles, most of them 64-bit index arithmetic on per-stage geometry that had spilled into scratch
// arrays): stage geometry is recomputed in uniform loops (scalar registers), lane-dependent addresses are formed once per stage
// and then stepped, per-pixel x weights come from a small LDS table.
template <int P>
__global__ void __launch_bounds__(256, 3)
pyramid_to_latent_kernel(const Pyramid p, int H0, int W0, float *__restrict__ nhwc, float *__restrict__ nchw) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float win[];
    const int t = threadIdx.x;
    const int Ctot = p.c_begin[p.n];
    const int n = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * P;
    const int npx = min(P, W0 - x0);
    auto stage = [&](int s) {  // s uniform
        StageRun g;
        g.cb = p.c_begin[s]; g.Cs = p.c_begin[s + 1] - g.cb; g.stride = g.Cs + FT_PAD; g.Ws = p.W[s]; g.nxw = p.nxw[s]; g.woff = p.win_off[s];
        g.same = p.same[s] != 0; g.sx = p.sx[s];
        if (g.same) {
            g.y0 = g.y1 = y; g.xa0 = x0; g.nx = npx; g.h0 = 1.f; g.h1 = 0.f;
        } else {
            const int Hs = p.H[s];
            const float fy = p.sy[s] * (float)y;
            g.y0 = __builtin_amdgcn_readfirstlane((int)fy);
            g.y1 = g.y0 + (g.y0 < Hs - 1 ? 1 : 0);
            g.h1 = fy - (float)g.y0; g.h0 = 1.f - g.h1;
            g.xa0 = __builtin_amdgcn_readfirstlane(min((int)(g.sx * (float)x0), g.Ws - 1));
            const int xl = __builtin_amdgcn_readfirstlane(min((int)(g.sx * (float)(x0 + npx - 1)), g.Ws - 1));
            g.nx = min(xl + (xl < g.Ws - 1 ? 1 : 0) - g.xa0 + 1, g.nxw);
        }
        return g;
    };
    f32x4_t *xtab = reinterpret_cast<f32x4_t *>(win + p.win_off[p.n]);
    const int quads = Ctot / 4;
    // this thread's constants of the channel-last sweep (its channel group q lies in exactly one stage)
    const bool fixed_q = 256 % quads == 0 && P % (256 / quads) == 0;  // every ResNet trunk
    const int q = t % quads;
    int a_w = 0, a_wb = 0, a_tab = 0;
    float a_h0 = 1.f, a_h1 = 0.f;
    bool a_same = true;

    for (int s = 0; s < p.n; ++s) {
        const StageRun g = stage(s);
        // ---- 1. window of stage s: lane i = t & (2^wsh - 1) is the texel of a row, j = t >> wsh the first channel, channels step
        // by 256 >> wsh; FT_U steps (both rows) are loaded before their LDS writes
        {
            const int wsh = p.wsh[s];
            const int i = t & ((1 << wsh) - 1), j = t >> wsh, step = 256 >> wsh;
            // 32-bit element offsets inside one view of the stage (the host checks Cs Hs Ws < 2^29); only the view base is 64-bit
            const int plane = p.H[s] * g.Ws;
            const float *base = p.src[s] + (size_t)n * ((size_t)g.Cs * plane) + (g.y0 * g.Ws + g.xa0);  // uniform
            const int row1 = (g.y1 - g.y0) * g.Ws;                                                       // uniform
            const int hop = step * plane;                                                                // uniform
            const int nsteps = g.Cs / step;  // uniform; step <= 64 divides every stage's channel count (host: wsh >= 2)
            if (i < g.nx) {
                const float *r0 = base + (unsigned)(j * plane + i);
                float *w0 = win + g.woff + i * g.stride + j;
                const int wrow = g.nxw * g.stride;
                for (int k = 0; k < nsteps; k += FT_U) {
                    float v0[FT_U], v1[FT_U];
#pragma unroll
                    for (int u = 0; u < FT_U; ++u) {
                        if (k + u >= nsteps) break;  // uniform
                        v0[u] = r0[u * hop];
                        if (!g.same) v1[u] = r0[u * hop + row1];
                    }
#pragma unroll
                    for (int u = 0; u < FT_U; ++u) {
                        if (k + u >= nsteps) break;
                        w0[u * step] = v0[u];
                        if (!g.same) w0[u * step + wrow] = v1[u];
                    }
                    r0 += (unsigned)(FT_U * hop);
                    w0 += FT_U * step;
                }
            }
        }
        // ---- per pixel of the run: float offsets of the two window texels and the x weights -- computed once, read back by every
        // thread that touches the pixel (one LDS read instead of ~20 VALU operations per output group)
        if (t < P) {
            const int x = min(x0 + t, W0 - 1);
            const float fx = g.sx * (float)x;
            const int xa = min((int)fx, g.Ws - 1), xb = xa + (xa < g.Ws - 1 ? 1 : 0);
            const float w1 = fx - (float)xa, w0 = 1.f - w1;
            int i0 = min(max(xa - g.xa0, 0), g.nxw - 1), i1 = min(max(xb - g.xa0, 0), g.nxw - 1);
            if (g.same) i0 = i1 = t;
            xtab[s * P + t] = f32x4_t{__int_as_float(i0 * g.stride), __int_as_float(i1 * g.stride), w0, w1};
        }
        if (q * 4 >= g.cb && q * 4 < g.cb + g.Cs) {
            a_w = g.woff + (q * 4 - g.cb); a_wb = a_w + g.nxw * g.stride; a_tab = s * P; a_h0 = g.h0; a_h1 = g.h1; a_same = g.same;
        }
    }
    __syncthreads();
    auto blend = [&](float h0, float h1, f32x4_t xp, f32x4_t a0, f32x4_t a1, f32x4_t b0, f32x4_t b1) {
        return h0 * (xp[2] * a0 + xp[3] * a1) + h1 * (xp[2] * b0 + xp[3] * b1);
    };
    // ---- 2. channel-last grid: lanes along the channels, 16 bytes per lane
    {
        float *dst = nhwc + (((size_t)n * H0 + y) * W0 + x0) * Ctot;  // uniform
        if (fixed_q) {
            const int PPG = P / (256 / quads);
            const float *w = win + a_w, *wb = win + a_wb;
            const f32x4_t *tab = xtab + a_tab;
            const int p_begin = (t / quads) * PPG, p_end = min(p_begin + PPG, npx);
            float *d = dst + ((size_t)p_begin * Ctot + q * 4);
            for (int px = p_begin; px < p_end; ++px, d += Ctot) {
                const f32x4_t xp = tab[px];
                const int o0 = __float_as_int(xp[0]), o1 = __float_as_int(xp[1]);
                f32x4_t val = *reinterpret_cast<const f32x4_t *>(w + o0);
                if (!a_same)
                    val = blend(a_h0, a_h1, xp, val, *reinterpret_cast<const f32x4_t *>(w + o1), *reinterpret_cast<const f32x4_t *>(wb + o0),
                                *reinterpret_cast<const f32x4_t *>(wb + o1));
                *reinterpret_cast<f32x4_t *>(d) = val;
            }
        } else {  // channel counts that do not tile 256 threads: every (pixel, group) on its own
            for (int s = 0; s < p.n; ++s) {
                const StageRun g = stage(s);
                const float *w = win + g.woff, *wb = w + g.nxw * g.stride;
                const int sq = g.Cs / 4;
                for (int k = t; k < npx * sq; k += 256) {
                    const int px = k / sq, c4 = (k - px * sq) * 4;
                    const f32x4_t xp = xtab[s * P + px];
                    const int o0 = __float_as_int(xp[0]), o1 = __float_as_int(xp[1]);
                    f32x4_t val = *reinterpret_cast<const f32x4_t *>(w + c4 + o0);
                    if (!g.same)
                        val = blend(g.h0, g.h1, xp, val, *reinterpret_cast<const f32x4_t *>(w + c4 + o1), *reinterpret_cast<const f32x4_t *>(wb + c4 + o0),
                                    *reinterpret_cast<const f32x4_t *>(wb + c4 + o1));
                    *reinterpret_cast<f32x4_t *>(dst + (size_t)px * Ctot + g.cb + c4) = val;
                }
            }
        }
    }
    // ---- 3. the reference's NCHW tensor: lanes along the pixels (128-byte runs of a channel plane)
    if (nchw) {
        const int px = t % P, grp = t / P;
        if (px < npx) {
            const size_t HW = (size_t)H0 * W0;
            float *dpx = nchw + ((size_t)n * Ctot * H0 + y) * W0 + x0 + px;
            for (int s = 0; s < p.n; ++s) {
                const StageRun g = stage(s);
                const f32x4_t xp = xtab[s * P + px];
                const int o0 = __float_as_int(xp[0]), o1 = __float_as_int(xp[1]);
                const float *w = win + g.woff + grp * 4, *wb = w + g.nxw * g.stride;
                float *d = dpx + (size_t)(g.cb + grp * 4) * HW;
                const size_t dstep = (size_t)(256 / P) * 4 * HW;
                for (int c4 = grp * 4; c4 < g.Cs; c4 += (256 / P) * 4, w += (256 / P) * 4, wb += (256 / P) * 4, d += dstep) {
                    f32x4_t val = *reinterpret_cast<const f32x4_t *>(w + o0);
                    if (!g.same)
                        val = blend(g.h0, g.h1, xp, val, *reinterpret_cast<const f32x4_t *>(w + o1), *reinterpret_cast<const f32x4_t *>(wb + o0),
                                    *reinterpret_cast<const f32x4_t *>(wb + o1));
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e * HW] = val[e];
                }
            }
        }
    }
}

// PositionalEncoding.forward, src/model/code.py:30-42, for callers that use the module on its own (the fused kernels form the
// code of their 3-vectors in registers, pnr_device.h):  out[n] = [x[n]] ++ sin(phases[j] + x[n][d] * freqs[j]) with j = 0 .. 2F-1
// outer and d inner; freqs / phases = the module's `_freqs` / `_phases` buffers (each frequency twice, phases 0, pi/2).  The
// argument is ONE fused multiply-add like ATen's addcmul.  One thread per output element: lanes along a row's d_out values.
__global__ void positional_encoding_kernel(const float *__restrict__ x, const float *__restrict__ freqs, const float *__restrict__ phases,
                                           long long total, int d_in, int d_out, int lead, float *__restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / d_out;
    const int k = (int)(idx - n * d_out);
    const float *row = x + n * d_in;
    if (k < lead) { out[idx] = row[k]; return; }
    const int j = (k - lead) / d_in, d = (k - lead) - j * d_in;
    out[idx] = sinf(__builtin_fmaf(row[d], freqs[j], phases[j]));
}

// backward of the above: g_x[n][d] = [g[n][d]] + sum_j g[n][lead + j d_in + d] cos(phases[j] + x[n][d] freqs[j]) freqs[j]
__global__ void positional_encoding_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ freqs,
                                               const float *__restrict__ phases, long long total, int d_in, int d_out, int lead,
                                               int F2, float *__restrict__ gx) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / d_in;
    const int d = (int)(idx - n * d_in);
    const float v = x[idx];
    const float *gr = g + n * d_out;
    float acc = lead ? gr[d] : 0.f;
    for (int j = 0; j < F2; ++j) acc += gr[lead + j * d_in + d] * (cosf(__builtin_fmaf(v, freqs[j], phases[j])) * freqs[j]);
    gx[idx] = acc;
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

// ---------------------------------------------------------------- SpatialEncoder.index as a stand-alone operator
// src/model/encoder.py:80-109: F.grid_sample(latent, uv, bilinear, padding "border", align_corners=True) on a (NV,C,Hl,Wl)
// grid at (NV,N) normalised coordinates -> (NV,C,N).  The grid is read in the channel-last copy the fused kernels use (one
// corner = one contiguous row); a workgroup owns 64 points of one view: per 64-channel chunk every wave blends the four
// corner rows of its 16 points with lanes along the channels (256-byte coalesced reads) into an LDS tile, which then leaves
// with lanes along the points (256-byte runs of one channel's output row) -- the reference's (NV,C,N) layout without a
// strided store.  Corner arithmetic in ATen's op order (no FMA contraction); a NaN coordinate reads texel 0 like
// project_point (pnr_device.h) and the fused kernels do.
struct IndexCorner {
    int off[4];   // texel index (y * Wl + x) of nw, ne, sw, se
    float w[4];
    float ix, iy;
    int x_on, y_on;  // coordinate strictly inside (0, size-1): grid_sample's border clip passes the gradient
};
#pragma clang fp contract(off)
__device__ __forceinline__ IndexCorner index_corner(float gx, float gy, int Wl_, int Hl_) {
    const float Wl = (float)Wl_, Hl = (float)Hl_;
    float ix = ((gx + 1.f) / 2.f) * (Wl - 1.f), iy = ((gy + 1.f) / 2.f) * (Hl - 1.f);
    IndexCorner c;
    c.x_on = ix > 0.f && ix < Wl - 1.f;
    c.y_on = iy > 0.f && iy < Hl - 1.f;
    ix = fminf(Wl - 1.f, fmaxf(ix, 0.f));
    iy = fminf(Hl - 1.f, fmaxf(iy, 0.f));
    if (!(ix == ix)) { ix = 0.f; c.x_on = 0; }
    if (!(iy == iy)) { iy = 0.f; c.y_on = 0; }
    const float ix0 = floorf(ix), iy0 = floorf(iy);
    const float ix1 = ix0 + 1.f, iy1 = iy0 + 1.f;
    c.w[0] = (ix1 - ix) * (iy1 - iy); c.w[1] = (ix - ix0) * (iy1 - iy);
    c.w[2] = (ix1 - ix) * (iy - iy0); c.w[3] = (ix - ix0) * (iy - iy0);
    const int x0 = (int)ix0, y0 = (int)iy0;
    const int x1 = min(x0 + 1, Wl_ - 1), y1 = min(y0 + 1, Hl_ - 1);  // an out-of-range corner carries weight 0
    if (x0 + 1 > Wl_ - 1) { c.w[1] = 0.f; c.w[3] = 0.f; }
    if (y0 + 1 > Hl_ - 1) { c.w[2] = 0.f; c.w[3] = 0.f; }
    c.off[0] = y0 * Wl_ + x0; c.off[1] = y0 * Wl_ + x1; c.off[2] = y1 * Wl_ + x0; c.off[3] = y1 * Wl_ + x1;
    c.ix = ix; c.iy = iy;
    return c;
}

constexpr int GI_P = 64;   // points per workgroup
constexpr int GI_C = 64;   // channels per LDS tile

__global__ void __launch_bounds__(256)
grid_index_kernel(const float *__restrict__ grid, int Hl, int Wl, int C, const float *__restrict__ uv, long long N,
                  float *__restrict__ out) {
    __shared__ float tile[GI_P][GI_C + 1];
    __shared__ IndexCorner corner[GI_P];
    const int v = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long n0 = (long long)blockIdx.x * GI_P;
    if (threadIdx.x < GI_P) {
        const long long n = n0 + threadIdx.x;
        IndexCorner c = {};
        if (n < N) c = index_corner(uv[((size_t)v * N + n) * 2], uv[((size_t)v * N + n) * 2 + 1], Wl, Hl);
        corner[threadIdx.x] = c;
    }
    __syncthreads();
    const float *gv = grid + (size_t)v * Hl * Wl * C;
    for (int c0 = 0; c0 < C; c0 += GI_C) {
        const int ch = c0 + lane;
#pragma unroll 4
        for (int i = 0; i < GI_P / 4; ++i) {
            const int p = wv * (GI_P / 4) + i;
            const IndexCorner &k = corner[p];
            float r = 0.f;
            if (ch < C && n0 + p < N) {
                // ATen's accumulation order: nw, ne, sw, se
                r = gv[(size_t)k.off[0] * C + ch] * k.w[0];
                r = r + gv[(size_t)k.off[1] * C + ch] * k.w[1];
                r = r + gv[(size_t)k.off[2] * C + ch] * k.w[2];
                r = r + gv[(size_t)k.off[3] * C + ch] * k.w[3];
            }
            tile[p][lane] = r;
        }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < GI_C / 4; ++i) {
            const int cc = wv * (GI_C / 4) + i;
            if (c0 + cc < C && n0 + lane < N) out[((size_t)v * C + c0 + cc) * N + n0 + lane] = tile[lane][cc];
        }
        __syncthreads();
    }
}

// backward: d grid (channel-last, accumulated with atomics: 256-byte runs per corner) and d uv (NV,N,2).
//   d out / d ix = (ne - nw)(iy1 - iy) + (se - sw)(iy - iy0), times (Wl - 1) / 2 for the normalised coordinate; zero where the
//   border clip is active (ATen clip_coordinates_set_grad) -- the rule position_bwd_kernel (pnr_bwd.hip) applies on the fused path.
__global__ void __launch_bounds__(256)
grid_index_bwd_kernel(const float *__restrict__ grid, int Hl, int Wl, int C, const float *__restrict__ uv, long long N,
                      const float *__restrict__ g_out, float *__restrict__ d_grid, float *__restrict__ d_uv) {
    __shared__ float tile[GI_P][GI_C + 1];
    __shared__ IndexCorner corner[GI_P];
    __shared__ float duv[GI_P][2];
    const int v = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long n0 = (long long)blockIdx.x * GI_P;
    if (threadIdx.x < GI_P) {
        const long long n = n0 + threadIdx.x;
        IndexCorner c = {};
        if (n < N) c = index_corner(uv[((size_t)v * N + n) * 2], uv[((size_t)v * N + n) * 2 + 1], Wl, Hl);
        corner[threadIdx.x] = c;
        duv[threadIdx.x][0] = 0.f; duv[threadIdx.x][1] = 0.f;
    }
    __syncthreads();
    const float *gv = grid + (size_t)v * Hl * Wl * C;
    float *dg = d_grid ? d_grid + (size_t)v * Hl * Wl * C : nullptr;
    float ax[GI_P / 4], ay[GI_P / 4];
#pragma unroll
    for (int i = 0; i < GI_P / 4; ++i) { ax[i] = 0.f; ay[i] = 0.f; }
    for (int c0 = 0; c0 < C; c0 += GI_C) {
#pragma unroll 4
        for (int i = 0; i < GI_C / 4; ++i) {
            const int cc = wv * (GI_C / 4) + i;
            tile[lane][cc] = (c0 + cc < C && n0 + lane < N) ? g_out[((size_t)v * C + c0 + cc) * N + n0 + lane] : 0.f;
        }
        __syncthreads();
        const int ch = c0 + lane;
#pragma unroll
        for (int i = 0; i < GI_P / 4; ++i) {
            const int p = wv * (GI_P / 4) + i;
            if (ch >= C || n0 + p >= N) continue;
            const IndexCorner &k = corner[p];
            const float g = tile[p][lane];
            if (dg) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k.w[q] != 0.f && g != 0.f) atomicAdd(dg + (size_t)k.off[q] * C + ch, k.w[q] * g);
            }
            if (d_uv) {
                const float nw = gv[(size_t)k.off[0] * C + ch], ne = gv[(size_t)k.off[1] * C + ch];
                const float sw = gv[(size_t)k.off[2] * C + ch], se = gv[(size_t)k.off[3] * C + ch];
                const float fx = k.ix - floorf(k.ix), fy = k.iy - floorf(k.iy);
                ax[i] += g * ((ne - nw) * (1.f - fy) + (se - sw) * fy);
                ay[i] += g * ((sw - nw) * (1.f - fx) + (se - ne) * fx);
            }
        }
        __syncthreads();
    }
    if (d_uv) {
#pragma unroll
        for (int i = 0; i < GI_P / 4; ++i) {
            float sx = ax[i], sy = ay[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); }
            const int p = wv * (GI_P / 4) + i;
            if (lane == 0 && n0 + p < N) {
                const IndexCorner &k = corner[p];
                d_uv[((size_t)v * N + n0 + p) * 2] = k.x_on ? sx * (((float)Wl - 1.f) / 2.f) : 0.f;
                d_uv[((size_t)v * N + n0 + p) * 2 + 1] = k.y_on ? sy * (((float)Hl - 1.f) / 2.f) : 0.f;
            }
        }
    }
}
#pragma clang fp contract(fast)

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

extern "C" int pnr_positional_encoding(const float *x, long long N, int d_in, int num_freqs, const float *freqs2, const float *phases2,
                                       int include_input, float *out, void *stream) {
    if (N < 0 || d_in <= 0 || num_freqs < 0) return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding: bad sizes");
    if (N == 0) return PNR_OK;
    if (!x || !out || (num_freqs > 0 && (!freqs2 || !phases2))) return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding: null argument");
    const int lead = include_input ? d_in : 0, d_out = lead + 2 * num_freqs * d_in;
    if (d_out == 0) return PNR_OK;
    const long long total = N * d_out;
    if ((total + 255) / 256 > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding: too many rows");
    hipLaunchKernelGGL(pnr::positional_encoding_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, freqs2,
                       phases2, total, d_in, d_out, lead, out);
    return pnr_check_launch("pnr_positional_encoding");
}

extern "C" int pnr_positional_encoding_backward(const float *x, const float *g_out, long long N, int d_in, int num_freqs,
                                                const float *freqs2, const float *phases2, int include_input, float *g_x, void *stream) {
    if (N < 0 || d_in <= 0 || num_freqs < 0) return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding_backward: bad sizes");
    if (N == 0) return PNR_OK;
    if (!x || !g_out || !g_x || (num_freqs > 0 && (!freqs2 || !phases2)))
        return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding_backward: null argument");
    const int lead = include_input ? d_in : 0, d_out = lead + 2 * num_freqs * d_in;
    const long long total = N * d_in;
    if ((total + 255) / 256 > 0x7fffffffLL) return pnr_fail(PNR_E_INVALID, "pnr_positional_encoding_backward: too many rows");
    hipLaunchKernelGGL(pnr::positional_encoding_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       g_out, freqs2, phases2, total, d_in, d_out, lead, 2 * num_freqs, g_x);
    return pnr_check_launch("pnr_positional_encoding_backward");
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
        if ((long long)channels[s] * heights[s] * widths[s] >= (1LL << 29))
            return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: a stage of one view must stay below 2 GiB");
        p.src[s] = stages[s]; p.c_begin[s] = c; p.H[s] = heights[s]; p.W[s] = widths[s];
        c += channels[s];
    }
    p.c_begin[n_stages] = c;
    if (NV == 0) return PNR_OK;
    const int H0 = heights[0], W0 = widths[0];
    // pixels per workgroup: the longest run whose source windows leave room for >= 3 workgroups per CU (else the shortest)
    int P = 32;
    size_t lds = 0;
    auto layout = [&](int run) {
        int off = 0;
        for (int s = 0; s < n_stages; ++s) {
            p.same[s] = heights[s] == H0 && widths[s] == W0;
            p.sy[s] = pnr::axis_scale(heights[s], H0); p.sx[s] = pnr::axis_scale(widths[s], W0);
            p.nxw[s] = p.same[s] ? run : pnr::window_texels(run, widths[s], W0);
            p.wsh[s] = 2;  // >= 4 lanes per row: the channel step 256 >> wsh is then <= 64 and divides every stage's channel count
            while ((1 << p.wsh[s]) < p.nxw[s]) ++p.wsh[s];
            p.win_off[s] = off;
            off += (p.same[s] ? 1 : 2) * p.nxw[s] * (channels[s] + pnr::FT_PAD);
        }
        p.win_off[n_stages] = off;
        return ((size_t)off + (size_t)n_stages * run * 4) * sizeof(float);  // windows + the per-pixel x table
    };
    for (;; P /= 2) {
        lds = layout(P);
        if (lds <= 48 * 1024 || P == 8) break;
    }
    if (lds > 160 * 1024) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: source windows do not fit the LDS (too many channels)");
    for (int s = 0; s < n_stages; ++s)
        if (p.nxw[s] > 256) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: a stage more than 32x wider than stage 0 is not supported");
    auto k = P == 32 ? pnr::pyramid_to_latent_kernel<32> : (P == 16 ? pnr::pyramid_to_latent_kernel<16> : pnr::pyramid_to_latent_kernel<8>);
    if (lds > 64 * 1024) {  // beyond the default dynamic-LDS limit (not reached by the shipped encoder shapes)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pnr_check_hip(e, "hipFuncSetAttribute(pyramid_to_latent_kernel)");
    }
    const int tiles_x = (W0 + P - 1) / P;
    if (H0 > 65535 || NV > 65535) return pnr_fail(PNR_E_INVALID, "pnr_pyramid_to_latent: grid too large");
    hipLaunchKernelGGL(k, dim3(tiles_x, H0, NV), dim3(256), lds, (hipStream_t)stream, p, H0, W0, latent_nhwc, latent_nchw);
    return pnr_check_launch("pnr_pyramid_to_latent");
}

extern "C" int pnr_grid_index(const float *latent_nhwc, int NV, int Hl, int Wl, int C, const float *uv, long long N, float *out,
                              void *stream) {
    if (NV < 0 || Hl <= 0 || Wl <= 0 || C <= 0 || N < 0) return pnr_fail(PNR_E_INVALID, "pnr_grid_index: bad sizes");
    if (NV == 0 || N == 0) return PNR_OK;
    if (!latent_nhwc || !uv || !out) return pnr_fail(PNR_E_INVALID, "pnr_grid_index: null argument");
    if ((long long)Hl * Wl * C >= (1LL << 31) || NV > 65535 || (N + pnr::GI_P - 1) / pnr::GI_P > 0x7fffffffLL)
        return pnr_fail(PNR_E_INVALID, "pnr_grid_index: one view's grid must stay below 2^31 elements, NV <= 65535");
    hipLaunchKernelGGL(pnr::grid_index_kernel, dim3((unsigned)((N + pnr::GI_P - 1) / pnr::GI_P), (unsigned)NV), dim3(256), 0,
                       (hipStream_t)stream, latent_nhwc, Hl, Wl, C, uv, N, out);
    return pnr_check_launch("pnr_grid_index");
}

extern "C" int pnr_grid_index_backward(const float *latent_nhwc, int NV, int Hl, int Wl, int C, const float *uv, long long N,
                                       const float *g_out, float *d_latent_nhwc, float *d_uv, void *stream) {
    if (NV < 0 || Hl <= 0 || Wl <= 0 || C <= 0 || N < 0) return pnr_fail(PNR_E_INVALID, "pnr_grid_index_backward: bad sizes");
    if (NV == 0 || N == 0 || (!d_latent_nhwc && !d_uv)) return PNR_OK;
    if (!latent_nhwc || !uv || !g_out) return pnr_fail(PNR_E_INVALID, "pnr_grid_index_backward: null argument");
    if ((long long)Hl * Wl * C >= (1LL << 31) || NV > 65535 || (N + pnr::GI_P - 1) / pnr::GI_P > 0x7fffffffLL)
        return pnr_fail(PNR_E_INVALID, "pnr_grid_index_backward: one view's grid must stay below 2^31 elements, NV <= 65535");
    hipLaunchKernelGGL(pnr::grid_index_bwd_kernel, dim3((unsigned)((N + pnr::GI_P - 1) / pnr::GI_P), (unsigned)NV), dim3(256), 0,
                       (hipStream_t)stream, latent_nhwc, Hl, Wl, C, uv, N, g_out, d_latent_nhwc, d_uv);
    return pnr_check_launch("pnr_grid_index_backward");
}
