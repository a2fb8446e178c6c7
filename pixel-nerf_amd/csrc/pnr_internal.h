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

// training forward of the fp32-class path: the split-operand kernel + what the backward keeps -- the (head | tail) 16-bit operand
// images of every 512-wide linear in storage order (img_a[b]: relu(x) entering blocks[b].fc_0, img_n[b]: relu(net) entering fc_1;
// b < 3: NS*P rows [view][point], else P rows; 2 x rows x 1024 bytes each), the stream in front of lin_out as fp32 rows (x5) and
// the relu bit masks (pnr_train_masks_bytes).  Defined in pnr_split.hip.
int eval_samples_split_train(const PnrScene *scene, const void *packed_split, const void *tables_f32, const float *rays,
                             const float *z, int R, int rays_per_obj, int K, float *rgbsigma, void *const *img_a, void *const *img_n,
                             float *x5, void *masks, hipStream_t stream);

// fused data-gradient chain of the fp32-class training path (bwd_split_kernel, pnr_split.hip): transposed (head, tail) weight
// streams packed from the raw parameters, relu masks of the TRAIN forward, g_out (P,4) unscaled + device [s, 1/s]; every layer's
// output gradient leaves at scale s as (head | tail) 16-bit rows in storage order (g_fc1[b] = dY of blocks[b].fc_1 = gradient of
// the stream behind block b, g_fc0[b] = dY of blocks[b].fc_0, g_x0 = gradient of the stream entering block 0; b < 3 and g_x0:
// [view][point] rows); d z_lat = sum_b dY_b W_z[b] and d(code) = dY_0 W_in come out of the same launch, unscaled fp32.
size_t bwd_split_packed_bytes();
int pack_bwd_split(const PnrMlpWeights *w, void *packed, hipStream_t st);
int mlp_backward_split_chain(const void *packed_bwd_split, const unsigned long long *masks, const float *g_out, const float *scale_dev,
                             long long P, int NS, void *const *g_fc1, void *const *g_fc0, void *g_x0, float *d_zlat /* (NS*P,512) */,
                             float *d_in /* (NS*P,42), nullable */, hipStream_t st);

// fp16-range guard of the fp32-class kernels (pnr_saturation_guard, pnr_api.hip): the flag word the next split-operand launch
// of THIS host thread reports into (NULL = guard off), and which of the caller's two words that is (render entries set the
// slot: 0 = coarse-network launch, 1 = fine-network launch).
unsigned int *saturation_guard_word();
void saturation_guard_slot(int slot);

// XCDs the dispatcher deals workgroups to round-robin on the current device (hipDeviceAttributeNumberOfXccs, i.e. of the
// current compute-partition mode: 8 in SPX, 1 in CPX; PIXELNERF_XCD_COUNT=n overrides, 0 selects the plain grid-stride
// tile order); cached per process.  Feeds EvalParams::n_xcd / tile_range() (pnr_device.h).  Defined in pnr_api.hip.
int device_xcd_count();

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
