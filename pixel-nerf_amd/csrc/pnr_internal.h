// pnr_internal.h -- entry points shared between the translation units of libpixelnerf_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "pnr_common.h"
#include "pnr_raysrc.h"

namespace pnr {

// fused network on (ray, z) samples with the rays taken from `src` (explicit array or camera): dispatches on
// `precision` (F16 / BF16 kernels of pnr_mlp.hip, F16X3 split-operand kernel of pnr_split.hip); tables == NULL selects
// the unfolded stream.
int eval_samples_src(const PnrScene *scene, const void *packed, const void *tables, int precision, const RaySrc &src,
                     const float *z, int R, int rays_per_obj, int K, float *rgbsigma, hipStream_t stream);
int eval_samples_split_src(const PnrScene *scene, const void *packed_split, const void *tables_f32, const RaySrc &src,
                           const float *z, int R, int rays_per_obj, int K, float *rgbsigma, hipStream_t stream);

// training forward of the fp32-class path: the split-operand kernel + fp32 rows (natural feature order) of the residual stream
// entering each block (xin[5]), each fc_0 output (net[5]), the stream in front of lin_out (x5) and, several views, every view's
// stream in front of the view mean (pool_in); row = view * P + point for the per-view tensors.  Defined in pnr_split.hip.
int eval_samples_split_train(const PnrScene *scene, const void *packed_split, const void *tables_f32, const float *rays,
                             const float *z, int R, int rays_per_obj, int K, float *rgbsigma, float *const *xin, float *const *net,
                             float *x5, float *pool_in, void *masks /* pnr_train_masks_bytes(P, NS), nullable */, hipStream_t stream);

// fused data-gradient chain of the fp32-class training path (bwd_split_kernel, pnr_split.hip): transposed (head, tail) weight
// streams packed from the raw parameters, relu masks of the TRAIN forward, g_out (P,4) unscaled + device [s, 1/s]; every layer's
// output gradient leaves as fp32 rows at scale s (g_fc1[b] = dY of blocks[b].fc_1 = gradient of the stream behind block b,
// g_fc0[b] = dY of blocks[b].fc_0, g_x0 = gradient of the stream entering block 0; b < 3 and g_x0: [view][point] rows);
// d z_lat = sum_b dY_b W_z[b] and d(code) = dY_0 W_in come out of the same launch.
size_t bwd_split_packed_bytes();
int pack_bwd_split(const PnrMlpWeights *w, void *packed, hipStream_t st);
int mlp_backward_split_chain(const void *packed_bwd_split, const unsigned long long *masks, const float *g_out, const float *scale_dev,
                             long long P, int NS, float *const *g_fc1, float *const *g_fc0, float *g_x0, float *d_zlat /* (NS*P,512) unscaled */,
                             float *d_in /* (NS*P,42) unscaled, nullable */, hipStream_t st);

// per (device, stream) scratch for the parked view sum of multi-view launches (one tile of fp32 accumulators per workgroup),
// allocated at the first multi-view launch on a stream and kept; NULL on allocation failure.  Defined in pnr_mlp.hip.
float *mv_scratch(hipStream_t st, size_t bytes);

// HIP events around one network-kernel launch, recorded on the launch's own stream while pnr_profile_enable(1) is
// active (pnr_profile_read sums them): the live kernel-time figure of bench.py's roofline block.  Defined in pnr_mlp.hip.
struct ProfileScope {
    hipEvent_t e0 = nullptr;
    hipStream_t st;
    explicit ProfileScope(hipStream_t s);
    ~ProfileScope();
};

}  // namespace pnr
