// pnr_pack.hip -- one-time repack of a ResnetFC's nn.Linear parameters into the per-wave MFMA
// fragment stream the fused kernel consumes (layout: pnr_layout.h), plus the NCHW->NHWC
// transpose of the encoder feature grid.  gfx950.
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_layout.h"

namespace pnr {

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

template <typename T>
__global__ void pack_weights_kernel(PnrMlpWeights p, T *__restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= WSTREAM_ELEMS_PER_WAVE * NW) return;
    const int e = idx & 7;
    const int lane = (idx >> 3) & 63;
    const int it = (idx >> 9) % IT;
    const size_t rest = idx / (FRAG_ELEMS * IT);
    const int rs = rest % RS_TOTAL;
    const int wv = rest / RS_TOTAL;
    int g = 0;
    while (g + 1 < NGEMM && rs >= gemm_offset(g + 1)) ++g;
    const int s = rs - gemm_offset(g);
    const int i = lane & 31, h = lane >> 5;
    const int f_out = wv * SL + it * 32 + i;
    const GemmSrc src = gemm_source(p, g);
    float v = 0.f;
    if (src.kind == 0) {
        const int k = s * 16 + h * 8 + e;
        if (k < D_IN) v = src.w[f_out * D_IN + k];
    } else if (src.kind == 1) {
        const int k = s * 16 + h * 8 + e;
        v = src.w[f_out * C_LAT + k];
    } else if (src.kind == 2) {
        // B operand comes from the LDS activation buffer: k-step s reads storage elements
        // 16s..16s+15 = what half (s&1) of feature tile (s>>1) wrote as registers 8h+e.
        const int k = feat_of(s >> 1, s & 1, 8 * h + e);
        v = src.w[f_out * D_HID + k];
    } else {
        // lin_out: B operand = the wave's own accumulators; k-step q = IT*s + it covers
        // registers 8*(q&1)..+7 of the wave's feature tile (q>>1), for both lane halves.
        if (s < 2) {
            const int q = IT * s + it;
            const int k = feat_of(wv * IT + (q >> 1), h, 8 * (q & 1) + e);
            if (i < D_OUT) v = src.w[i * D_HID + k];
        }
    }
    out[idx] = (T)v;
}

__global__ void pack_bias_kernel(PnrMlpWeights p, float *__restrict__ bias, float *__restrict__ bout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < D_OUT) bout[idx] = p.lin_out_b[idx];
    if (idx >= NBIAS * NW * BIAS_FLOATS_PER_WAVE) return;
    const int r = idx & 15, h = (idx >> 4) & 1, it = (idx >> 5) % IT;
    const int wv = (idx / BIAS_FLOATS_PER_WAVE) % NW;
    const int slot = idx / (BIAS_FLOATS_PER_WAVE * NW);
    const int f = feat_of(wv * IT + it, h, r);
    float v = 0.f;
    switch (slot) {
        case B_IN_Z0: v = p.lin_in_b[f] + p.lin_z_b[0][f]; break;
        case B_FC0_0: v = p.fc0_b[0][f]; break;
        case B_FC1_0_Z1: v = p.fc1_b[0][f] + p.lin_z_b[1][f]; break;
        case B_FC0_1: v = p.fc0_b[1][f]; break;
        case B_FC1_1_Z2: v = p.fc1_b[1][f] + p.lin_z_b[2][f]; break;
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

}  // namespace pnr

extern "C" size_t pnr_packed_mlp_bytes(void) { return pnr::PACKED_BYTES; }

extern "C" int pnr_pack_mlp(const PnrMlpWeights *w, int precision, void *packed, void *stream) {
    using namespace pnr;
    if (!w || !packed) return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp: null argument");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = WSTREAM_ELEMS_PER_WAVE * NW;
    const int threads = 256;
    const unsigned blocks = (unsigned)((n + threads - 1) / threads);
    if (precision == PNR_PREC_F16)
        hipLaunchKernelGGL(pack_weights_kernel<_Float16>, dim3(blocks), dim3(threads), 0, st, *w, (_Float16 *)packed);
    else if (precision == PNR_PREC_BF16)
        hipLaunchKernelGGL(pack_weights_kernel<__bf16>, dim3(blocks), dim3(threads), 0, st, *w, (__bf16 *)packed);
    else
        return pnr_fail(PNR_E_INVALID, "pnr_pack_mlp: unknown precision");
    const int nb = NBIAS * NW * BIAS_FLOATS_PER_WAVE;
    hipLaunchKernelGGL(pack_bias_kernel, dim3((nb + threads - 1) / threads), dim3(threads), 0, st, *w,
                       (float *)((char *)packed + BIAS_OFFSET_BYTES), (float *)((char *)packed + BOUT_OFFSET_BYTES));
    return pnr_check_launch("pnr_pack_mlp");
}

extern "C" int pnr_nchw_to_nhwc(const float *in, float *out, int N, int C, int H, int W, void *stream) {
    if (!in || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return pnr_fail(PNR_E_INVALID, "pnr_nchw_to_nhwc: bad argument");
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, N), block(32, 8);
    hipLaunchKernelGGL(pnr::nchw_to_nhwc_kernel, grid, block, 0, (hipStream_t)stream, in, out, C, HW);
    return pnr_check_launch("pnr_nchw_to_nhwc");
}
