// pnr_layout.h -- tile geometry and packed-weight layout shared by the repack kernel
// (pnr_pack.hip) and the fused network kernel (pnr_mlp.hip).  gfx950 only.
//
// Fused-kernel geometry:
//   workgroup = NW=8 waves (512 threads, 2 waves per SIMD), one workgroup per CU, persistent;
//   point tile = MT=64 sample points = JT=2 MFMA column tiles of 32 points;
//   wave w owns hidden features [64w, 64w+64) = IT=2 MFMA row tiles of 32 features;
//   every 512x512 linear is computed as  Y^T[feature][point] = W[feature][k] * X^T[k][point]
//   with v_mfma_f32_32x32x16_{f16,bf16}: A operand = weights (streamed from L2 straight into
//   VGPRs, pre-packed in fragment order), B operand = activations (LDS, or registers for
//   lin_out), C/D = fp32 accumulators that hold the residual stream.
//
// MFMA 32x32x16 fragment maps (lane l, h = l>>5):
//   A: row i = l&31, k = 8h+e (e=0..7)      B: col j = l&31, k = 8h+e
//   D: col j = l&31, row i = (r&3) + 8*(r>>2) + 4h  (r=0..15)
#pragma once
#include <stdint.h>

namespace pnr {

// Loads through a PnrMlpWeights pointer: the parameters are CALLER memory and may be views into a flat buffer at any 4-byte
// offset (torch's own allocations are 16-byte aligned, a `flat[1:]` view is not).  Every kernel that reads a parameter tensor with
// vector loads goes through these 4-byte-aligned types -- global_load_dwordx4 either way, without the compiler assuming more
// (one alignment contract for pack, fold, checksum, the fp32 GEMMs and the backward transposes).
typedef float f32x4_param __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x4_param __attribute__((ext_vector_type(4), aligned(4)));


constexpr int C_LAT = 512;   // latent channels (encoder.latent_size, encoder.py:68)
constexpr int D_HID = 512;   // d_hidden
constexpr int D_IN = 42;     // 39 positional code + 3 view direction (models.py:47-60)
constexpr int D_IN_PAD = 64; // padded K of lin_in
constexpr int D_OUT = 4;
constexpr int N_BLOCKS = 5;
constexpr int COMBINE_LAYER = 3;

#ifndef PNR_NW
#define PNR_NW 8
#endif
constexpr int NW = PNR_NW;        // waves per workgroup (8: 2 per SIMD, 16: 4 per SIMD)
constexpr int NTHREADS = NW * 64;
constexpr int SL = D_HID / NW;    // hidden features per wave (64)
constexpr int IT = SL / 32;       // feature tiles per wave (2)
constexpr int MT = 64;            // points per tile
constexpr int JT = MT / 32;       // point tiles (2)

// ---- per-wave weight stream, in "ring steps" (one ring step = IT fragments of 1 KiB) ----
constexpr int KS_IN = D_IN_PAD / 16;  // 4
constexpr int KS_BIG = D_HID / 16;    // 32
constexpr int KS_OUT = 4;             // lin_out: 4 real k-steps packed into ring steps 0,1 (x IT); 2,3 zero
// execution order of the GEMMs
enum Gemm {
    G_LIN_IN = 0, G_Z0, G_FC0_0, G_FC1_0, G_Z1, G_FC0_1, G_FC1_1, G_Z2, G_FC0_2, G_FC1_2,
    G_FC0_3, G_FC1_3, G_FC0_4, G_FC1_4, G_OUT, NGEMM
};
__host__ __device__ constexpr int gemm_ksteps(int g) {
    return g == G_LIN_IN ? KS_IN : (g == G_OUT ? KS_OUT : KS_BIG);
}
__host__ __device__ constexpr int gemm_offset(int g) {  // first ring step of GEMM g
    int o = 0;
    for (int i = 0; i < g; ++i) o += gemm_ksteps(i);
    return o;
}
constexpr int RS_VIEW_END = gemm_offset(G_FC0_3);  // 292: end of the per-view segment (blocks 0-2)
constexpr int RS_TOTAL = gemm_offset(NGEMM);       // 424
static_assert(RS_VIEW_END % 4 == 0 && RS_TOTAL % 4 == 0, "ring depth 4 needs aligned segments");

// "folded" inference stream: lin_z[b](z) is linear in the interpolated latent, so W_z[b] is applied to the feature
// grid once per scene (per-texel tables, pnr_fold_latent) and the three lin_z GEMMs leave the per-point stream.
__host__ __device__ constexpr bool gemm_is_linz(int g) { return g == G_Z0 || g == G_Z1 || g == G_Z2; }
__host__ __device__ constexpr int gemm_offset_fold(int g) {  // first ring step of GEMM g in the folded stream
    int o = 0;
    for (int i = 0; i < g; ++i) o += gemm_is_linz(i) ? 0 : gemm_ksteps(i);
    return o;
}
constexpr int RS_VIEW_END_F = gemm_offset_fold(G_FC0_3);  // 196
constexpr int RS_TOTAL_F = gemm_offset_fold(NGEMM);       // 328
static_assert(RS_VIEW_END_F % 4 == 0 && RS_TOTAL_F % 4 == 0, "ring depth 4 needs aligned segments");

constexpr int FRAG_ELEMS = 64 * 8;  // 64 lanes x 8 elements (1 KiB of 16-bit)
constexpr size_t WSTREAM_ELEMS_PER_WAVE = (size_t)RS_TOTAL * IT * FRAG_ELEMS;
constexpr size_t WSTREAM_BYTES = WSTREAM_ELEMS_PER_WAVE * NW * 2;  // 6,946,816 B

// ---- bias slots: one per accumulator (re)initialisation, already summed where two linears
// accumulate into the residual stream back to back ----
enum BiasSlot {
    B_IN_Z0 = 0,   // lin_in.bias + lin_z[0].bias
    B_FC0_0, B_FC1_0_Z1,  // blocks[0].fc_0 ; blocks[0].fc_1 + lin_z[1]
    B_FC0_1, B_FC1_1_Z2,
    B_FC0_2, B_FC1_2,
    B_FC0_3, B_FC1_3, B_FC0_4, B_FC1_4, NBIAS
};
constexpr int BIAS_FLOATS_PER_WAVE = IT * 2 * 16;  // [it][h][r]
constexpr size_t BIAS_OFFSET_BYTES = WSTREAM_BYTES;
constexpr size_t BIAS_BYTES = (size_t)NBIAS * NW * BIAS_FLOATS_PER_WAVE * 4;
constexpr size_t BOUT_OFFSET_BYTES = BIAS_OFFSET_BYTES + BIAS_BYTES;
constexpr size_t PACKED_BYTES = BOUT_OFFSET_BYTES + 32;  // lin_out bias (4 floats) | network flags word (+ pad)
constexpr int BOUT_FLAGS_INDEX = 4;  // 32-bit word behind lin_out's bias: bit 0 = combine_type "max" (util.py:467-468)

// hidden feature held by D-register r of half h in feature tile T (global tile index 0..15)
__host__ __device__ constexpr int feat_of(int T, int h, int r) {
    return 32 * T + (r & 3) + 8 * (r >> 2) + 4 * h;
}
// inverse: position of hidden feature f in "storage order" (element 32T + 16h + r of an activation row)
__host__ __device__ constexpr int slot_of(int f) {
    return (f & ~31) + 16 * ((f >> 2) & 1) + (f & 3) + 4 * ((f & 31) >> 3);
}

// ---- backward chain: transposed weight stream, consumption order per tile ----
//   head (pooled, once): lin_out^T (4 ring steps, k-step 0 real) | fc_1[4]^T fc_0[4]^T fc_1[3]^T fc_0[3]^T
//   per source view    : fc_1[2]^T fc_0[2]^T fc_1[1]^T fc_0[1]^T fc_1[0]^T fc_0[0]^T
//                        | lin_z[2]^T lin_z[1]^T lin_z[0]^T  (d z_lat = sum_b dY_b W_z[b])
//                        | lin_in^T (4 ring steps: every wave contracts its own 64 hidden features, K-split)
enum BGemm { BG_OUT = 0, BG_FC1_4, BG_FC0_4, BG_FC1_3, BG_FC0_3, BG_FC1_2, BG_FC0_2, BG_FC1_1, BG_FC0_1, BG_FC1_0, BG_FC0_0,
             BG_Z2, BG_Z1, BG_Z0, BG_IN, NBGEMM };
__host__ __device__ constexpr int bgemm_offset(int g) { return g == 0 ? 0 : KS_IN + (g - 1) * KS_BIG; }
constexpr int BRS_HEAD_END = bgemm_offset(BG_FC1_2);  // 132
constexpr int BRS_TOTAL = bgemm_offset(BG_IN) + KS_IN;  // 424
static_assert(BRS_HEAD_END % 4 == 0 && BRS_TOTAL % 4 == 0, "ring depth 4 needs aligned segments");
constexpr size_t BWSTREAM_ELEMS_PER_WAVE = (size_t)BRS_TOTAL * IT * FRAG_ELEMS;
constexpr size_t BPACKED_BYTES = BWSTREAM_ELEMS_PER_WAVE * NW * 2;
constexpr int ROW_GOUT = 48;  // d(lin_out output) operand rows: 16 x 16-bit (4 real) + pad -> 3 slots/row

// ---- LDS map of the fused kernel (bytes), per point-tile size ----
constexpr int ROW_ACT = D_HID * 2 + 16;      // 1040 B per point row: 1024 B of 16-bit activations + one
                                             // 16-B slot of padding, so the 32 rows a ds_read_b128 touches
                                             // fall on distinct 16-B bank slots (65 slots/row, odd)
constexpr int ROW_IN = D_IN_PAD * 2 + 16;    // 144 B: 9 slots/row -> conflict-free b128 reads
// Tile<64>: two images -- Z (interpolated latent / table rows) and A (relu(x) / relu(net) operand) -- as the
//           unfolded, training and multi-view instantiations need (lin_z reads Z while A is live).
// Tile<96>: the folded single-view inference tile.  x + net for 96 points are 2 x 96 accumulator registers per
//           wave (the 256-register budget of 2 waves/SIMD allows no more), and ONE image: the table rows of
//           lin_z[b+1] are looked up into the activation image after fc_1[b] has finished reading it.  A weight
//           fragment fetched from L2 now feeds 3 MFMAs instead of 2: the L2->CU stream per point drops by a third.
template <int MT_> struct Tile {
    static_assert(MT_ == 64 || MT_ == 96, "supported point tiles");
    static constexpr int MT = MT_;            // points per tile
    static constexpr int JT = MT_ / 32;       // MFMA column tiles of 32 points
    static constexpr bool SINGLE_IMAGE = MT_ > 64;
    static constexpr int IN_LO_DELTA = 0;     // no split-operand image (pnr_split.hip has its own map)
    static constexpr int LDS_Z = 0;
    static constexpr int LDS_A = SINGLE_IMAGE ? 0 : MT_ * ROW_ACT;
    static constexpr int LDS_IN = LDS_A + MT_ * ROW_ACT;       // positional code + viewdir
    static constexpr int LDS_META = LDS_IN + MT_ * ROW_IN;     // per point: 4 corner offsets + 4 weights
    static constexpr int LDS_OUT = LDS_META + MT_ * 32;        // lin_out partials [NW][MT][4] floats
    // the 96-point tile has room for the network's bias table (11 slots x 512 fp32, accumulator order): the 20 accumulator
    // initialisations of a tile then read LDS instead of waiting for an L2 round trip in front of every GEMM
    static constexpr bool BIAS_IN_LDS = SINGLE_IMAGE;
    static constexpr int LDS_BIAS = LDS_OUT + NW * MT_ * 16;
    static constexpr int LDS_TOTAL = LDS_BIAS + (BIAS_IN_LDS ? 11 * NW * IT * 32 * 4 : 0);  // 152,576 B (64) / 151,552 B (96)
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
};
// the 64-point map under its historical names (backward chain, training instantiation)
constexpr int LDS_Z = Tile<64>::LDS_Z;
constexpr int LDS_A = Tile<64>::LDS_A;
constexpr int LDS_IN = Tile<64>::LDS_IN;
constexpr int LDS_META = Tile<64>::LDS_META;
constexpr int LDS_OUT = Tile<64>::LDS_OUT;
constexpr int LDS_TOTAL = Tile<64>::LDS_TOTAL;

}  // namespace pnr
