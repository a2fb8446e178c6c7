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
                             float *x5, float *pool_in, hipStream_t stream);

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
