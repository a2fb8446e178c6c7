// split_gemm.hip -- stand-alone micro-benchmark of ONE 512 x 512 split-operand linear of the fused fp32-class network
// (pnr::eval_split_kernel, pixel-nerf_amd/csrc/pnr_split.hip), VERDICT r04 "next" item 2.
//
// What it settles: eval_split_kernel re-streams the 10.8 MB (head + tail) weight set from L2 for every 64-point tile -- 2 KiB of
// weight fragments per 3 JT MFMAs -- and the only lever on bytes per MFMA is JT, the number of 32-point column tiles a fragment
// feeds.  LDS pins the shipped form at JT = 2 (two 66.5 KB operand images).  The candidate is a ONE-wave-per-SIMD, 512-register
// form: 4 waves x 128 features (IT = 4), JT = 3 or 4, accumulators in AGPRs, operand images holding one K half at a time.  This
// program times the GEMM loop of both forms on the real footprint (a 10 x 512 x 512 (head, tail) stream per workgroup pass, so
// the per-XCD L2 behaves as in the product), with the issue order of the product loop, and nothing else of the network:
// if the wide form's loop is not >= 10 % faster per point here, no integration of it can be.
//
//   form          waves  IT JT  regs/lane (acc + ring + B)   LDS image (head + tail)       extras
//   shipped         8     2  2  64 [+64] + 64 + 32           64 pts x 512 K  = 133 KB      1 barrier per GEMM
//   wide-96         4     4  3  192 [+192] + 64|128 + 48     96 pts x 256 K  = 101 KB      3 barriers per GEMM (K halves)
//   wide-128        4     4  4  256 + 64|128 + 64            128 pts x 256 K = 135 KB      3 barriers per GEMM, PARK: the residual
//                                                                                          stream x leaves the registers for an
//                                                                                          L2-resident scratch around every fc_0
//                                                                                          (256 KB out + 256 KB in per block)
// [+..] = a second accumulator set kept live (x next to net), as the product holds it.
//
// Build + run (one MI355X):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/split_gemm.hip -o /tmp/split_gemm && /tmp/split_gemm
// Output: one line per form -- us per launch, points x GEMMs per second, executed MFMA TFLOP/s (3 MFMAs per product), and the
// rate relative to the shipped form (register counts: tools/kernel_regs.py on the binary).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int D = 512, KS = D / 16;  // 32 k-steps of 16
constexpr int NG = 10;               // GEMMs per tile pass: the ten 512 x 512 linears of a folded single-view network

__device__ __forceinline__ f32x16 mf(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// NWV waves, wave w owns IT feature tiles of 32; JT point tiles of 32.  KHALF: the LDS images hold 256 of the 512 K (two extra
// barriers per GEMM stand for the second half's write phase).  NACC accumulator sets alternate GEMM by GEMM (x / net).
// PARK: after every second GEMM the accumulators make a round trip through a per-workgroup global scratch (the residual stream
// parked around fc_0).  LOADS = false: the ring is never refilled (upper bound without the weight stream).
template <int NWV, int IT, int JT, int RING, int NACC, bool KHALF, bool PARK, bool LOADS>
__global__ void __launch_bounds__(NWV * 64) gemm_ub(const char *__restrict__ wstream, size_t tail_delta, float *__restrict__ out,
                                                    f32x4 *__restrict__ park_ws, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = JT * 32;
    constexpr int KIMG = KHALF ? D / 2 : D;
    constexpr int ROW = KIMG * 2 + 16;  // 16-byte pad: conflict-free ds_read_b128 (odd number of 16-byte slots per row)
    constexpr int LO_DELTA = MT * ROW;
    constexpr int KSTEPS_IMG = KIMG / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 31, h = lane >> 5;
    // operand images: something finite and non-trivial
    for (int i = tid; i < 2 * MT * ROW / 2; i += NWV * 64)
        reinterpret_cast<_Float16 *>(smem)[i] = (_Float16)(((i * 37 + 11) % 61) * (1.f / 64.f) - 0.45f);
    __syncthreads();

    const size_t wave_bytes = (size_t)NG * KS * IT * 1024;
    const char *base_h = wstream + (size_t)wv * wave_bytes + lane * 16;
    const char *base_l = base_h + tail_delta;
    h8 rh[RING][IT], rl[RING][IT];
#pragma unroll
    for (int j = 0; j < RING; ++j)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            rh[j][it] = *reinterpret_cast<const h8 *>(base_h + (size_t)(j * IT + it) * 1024);
            rl[j][it] = *reinterpret_cast<const h8 *>(base_l + (size_t)(j * IT + it) * 1024);
        }
    int pf = RING;  // next k-step (of NG * KS) to prefetch

    f32x16 acc[NACC][IT][JT];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][it][jt][r] = 0.f;

    const uint32_t b0 = pl * ROW + h * 16;
    float sink = 0.f;

    auto gemm = [&](f32x16 (&c)[IT][JT]) {
#pragma unroll 1
        for (int half = 0; half < (KHALF ? 2 : 1); ++half) {
            if (KHALF) {  // the half's write phase is published / the previous half's readers are done
                __syncthreads();
                if (half == 1) __syncthreads();
            } else {
                __syncthreads();
            }
            h8 bh[2][JT], bl[2][JT];
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                bh[0][jt] = *reinterpret_cast<const h8 *>(smem + b0 + jt * 32 * ROW);
                bl[0][jt] = *reinterpret_cast<const h8 *>(smem + b0 + jt * 32 * ROW + LO_DELTA);
            }
            uint32_t bk = b0;
#pragma unroll 1
            for (int body = 0; body < KSTEPS_IMG / RING; ++body) {
                const size_t pfo = (size_t)pf * (IT * 1024);
#pragma unroll
                for (int j = 0; j < RING; ++j) {
                    const int cur = j & 1;
                    const bool lastk = (j == RING - 1) && (body == KSTEPS_IMG / RING - 1);
                    const uint32_t nx = lastk ? b0 : bk + (j + 1) * 32;  // (after the last k-step: step 0 again, read and dropped)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        bh[cur ^ 1][jt] = *reinterpret_cast<const h8 *>(smem + nx + jt * 32 * ROW);
                        bl[cur ^ 1][jt] = *reinterpret_cast<const h8 *>(smem + nx + jt * 32 * ROW + LO_DELTA);
                    }
                    h8 ah[IT], al[IT];
#pragma unroll
                    for (int it = 0; it < IT; ++it) { ah[it] = rh[j][it]; al[it] = rl[j][it]; }
                    if constexpr (RING == 1) {
                        // just-in-time ring: feature tile by feature tile -- a fragment pair is refilled for the NEXT k-step as soon as its
                        // 3 JT MFMAs are issued (3 JT (IT - 1) MFMAs of cover), 8 IT registers of ring instead of 16 IT
#pragma unroll
                        for (int it = 0; it < IT; ++it) {
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(ah[it], bh[cur][jt], c[it][jt]);
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(ah[it], bl[cur][jt], c[it][jt]);
#pragma unroll
                            for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(al[it], bh[cur][jt], c[it][jt]);
                            if (LOADS) {
                                rh[0][it] = *reinterpret_cast<const h8 *>(base_h + pfo + (size_t)it * 1024);
                                rl[0][it] = *reinterpret_cast<const h8 *>(base_l + pfo + (size_t)it * 1024);
                            }
                        }
#pragma unroll
                        for (int it = 0; it < IT; ++it) {
                            if (it == 0) {
#pragma unroll
                                for (int i = 0; i < 2 * JT; ++i) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                                }
                                __builtin_amdgcn_sched_group_barrier(0x008, JT, 0);
                            } else {
                                __builtin_amdgcn_sched_group_barrier(0x008, 3 * JT, 0);
                            }
                            if (LOADS) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                        }
                    } else {
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(ah[it], bh[cur][jt], c[it][jt]);
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(ah[it], bl[cur][jt], c[it][jt]);
#pragma unroll
                    for (int it = 0; it < IT; ++it)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt) c[it][jt] = mf(al[it], bh[cur][jt], c[it][jt]);
                    if (LOADS) {
#pragma unroll
                        for (int it = 0; it < IT; ++it) {
                            rh[j][it] = *reinterpret_cast<const h8 *>(base_h + pfo + (size_t)(j * IT + it) * 1024);
                            rl[j][it] = *reinterpret_cast<const h8 *>(base_l + pfo + (size_t)(j * IT + it) * 1024);
                        }
                    }
                    // the product loop's pinned order, generalised: (M L) x 2JT, (n M, G) x 2IT, rest
                    constexpr int NM = 3 * IT * JT, NL = 2 * JT, NV = LOADS ? 2 * IT : 0;
#pragma unroll
                    for (int i = 0; i < NL; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if constexpr (NV > 0) {
                        constexpr int PER = (NM - NL) / NV;
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        }
                        if constexpr (NM - NL - PER * NV > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - NL - PER * NV, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, NM - NL, 0);
                    }
                    }
                }
                bk += RING * 32;
                pf += RING;
                if (pf == NG * KS) pf = 0;
            }
        }
    };

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll 1
        for (int g = 0; g < NG; g += 2) {
            if constexpr (PARK) {  // the residual stream leaves for the scratch while fc_0 accumulates, and comes back for fc_1
                f32x4 *ws = park_ws + (size_t)blockIdx.x * (IT * JT * 4 * NWV * 64);
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 v = {acc[0][it][jt][4 * k], acc[0][it][jt][4 * k + 1], acc[0][it][jt][4 * k + 2], acc[0][it][jt][4 * k + 3]};
                            ws[((it * JT + jt) * 4 + k) * (NWV * 64) + tid] = v;
                        }
            }
            gemm(acc[0]);
            if constexpr (PARK) {
                // the product's split epilogue consumes net here; one value per accumulator tile keeps the GEMM alive
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) sink += acc[0][it][jt][(it + jt) & 15];
                const f32x4 *ws = park_ws + (size_t)blockIdx.x * (IT * JT * 4 * NWV * 64);
#pragma unroll
                for (int it = 0; it < IT; ++it)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        __builtin_amdgcn_sched_barrier(0);  // one accumulator tile (4 loads) in flight at a time: no register spike
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            // (a rotated slot: the compiler cannot forward the stores; loads may target AGPRs directly)
                            const f32x4 v = ws[(((it * JT + jt) * 4 + k + 1) % (IT * JT * 4)) * (NWV * 64) + tid];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[0][it][jt][4 * k + e] = v[e];
                        }
                    }
            }
            gemm(acc[NACC - 1]);
        }
    }
    float s = sink;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int it = 0; it < IT; ++it)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[a][it][jt][r];
    out[(size_t)blockIdx.x * (NWV * 64) + tid] = s;
}

struct Result { double us, pts_gemm_per_s, tflops; };

template <int NWV, int IT, int JT, int RING, int NACC, bool KHALF, bool PARK, bool LOADS>
static Result run(const char *name, const char *d_w, size_t tail_delta, float *d_out, f32x4 *d_ws, int tiles_per_wg, int reps, double ref_rate) {
    constexpr int MT = JT * 32;
    constexpr int KIMG = KHALF ? D / 2 : D;
    constexpr int ROW = KIMG * 2 + 16;
    const size_t lds = 2 * (size_t)MT * ROW;
    auto kern = gemm_ub<NWV, IT, JT, RING, NACC, KHALF, PARK, LOADS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)));
    const int ncu = 256, ntiles = ncu * tiles_per_wg;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(NWV * 64), lds, 0, d_w, tail_delta, d_out, d_ws, ntiles);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(ncu), dim3(NWV * 64), lds, 0, d_w, tail_delta, d_out, d_ws, ntiles);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    Result r;
    r.us = ms * 1e3 / reps;
    const double pts_gemm = (double)ntiles * MT * NG;
    r.pts_gemm_per_s = pts_gemm / (r.us * 1e-6);
    r.tflops = pts_gemm * (2.0 * D * D * 3) / (r.us * 1e-6) / 1e12;
    printf("%-34s waves %d IT %d JT %d ring %d acc-sets %d | regs %3d (spill/scratch %d B) LDS %6zu | %9.1f us | %7.2f G pt.gemm/s | %7.1f TF executed | x%.3f\n",
           name, NWV, IT, JT, RING, NACC, fa.numRegs, (int)fa.localSizeBytes, lds, r.us, r.pts_gemm_per_s / 1e9, r.tflops,
           ref_rate > 0 ? r.pts_gemm_per_s / ref_rate : 1.0);
    fflush(stdout);
    return r;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    // one (head | tail) stream: NG x 512 x 512 x 2 B per blob -- 5.24 MB + 5.24 MB, the product's 10.8 MB footprint
    const size_t blob = (size_t)NG * D * D * 2;
    std::vector<_Float16> hw(blob);  // (elements: blob bytes = 2 blobs x blob/2 elements x 2 B)
    uint32_t s = 12345u;
    for (size_t i = 0; i < hw.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const float v = ((int)(s >> 9) % 2001 - 1000) * (i < blob / 2 ? 4.0e-5f : 2.0e-8f);  // heads ~ +-0.04, tails ~ +-2e-5
        hw[i] = (_Float16)v;
    }
    char *d_w;
    float *d_out;
    f32x4 *d_ws;
    CK(hipMalloc(&d_w, 2 * blob));
    CK(hipMemcpy(d_w, hw.data(), 2 * blob, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&d_ws, (size_t)256 * 256 * 1024));  // 256 KiB per workgroup
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs; %d GEMMs of 512 x 512 (head + tail = %.2f MB) per tile pass; persistent, one workgroup per CU\n", pr.gcnArchName,
           pr.multiProcessorCount, NG, 2 * blob / 1e6);
    // every form does the same number of (point, GEMM) products per workgroup: 768 points
    //                 NWV IT JT RING NACC KHALF  PARK   LOADS
    const Result a = run<8, 2, 2, 4, 2, false, false, true>("shipped (64 pt, 2 waves/SIMD)", d_w, blob, d_out, d_ws, 12, reps, 0);
    const double ref = a.pts_gemm_per_s;
    run<8, 2, 2, 4, 2, false, false, false>("shipped, ring never refilled", d_w, blob, d_out, d_ws, 12, reps, ref);
    run<4, 4, 2, 2, 2, false, false, true>("1 wave/SIMD, 64 pt (JT 2)", d_w, blob, d_out, d_ws, 12, reps, ref);
    run<4, 4, 3, 2, 2, true, false, true>("wide-96  (x + net live), ring 2", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 3, 2, 1, true, false, true>("wide-96  (one acc set), ring 2", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 3, 4, 1, true, false, true>("wide-96  (one acc set), ring 4", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 4, 2, 1, true, false, true>("wide-128 (x parked: no traffic)", d_w, blob, d_out, d_ws, 6, reps, ref);
    run<4, 4, 3, 1, 2, true, false, true>("wide-96  (x + net live), JIT ring", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 3, 1, 1, true, false, true>("wide-96  (one acc set), JIT ring", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 3, 2, 1, true, true, true>("wide-96  + park round trips", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<4, 4, 4, 2, 1, true, true, true>("wide-128 + park round trips", d_w, blob, d_out, d_ws, 6, reps, ref);
    run<4, 4, 4, 2, 1, true, false, false>("wide-128, ring never refilled", d_w, blob, d_out, d_ws, 6, reps, ref);
    // the shipped wave geometry (8 waves x 64 features: the packed stream's layout) with the residual stream parked: one
    // accumulator set of 32 JT registers per feature tile, K-half-staged images
    run<8, 2, 3, 4, 1, true, false, true>("8 waves, 96 pt, K halves", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<8, 2, 3, 4, 1, true, true, true>("8 waves, 96 pt, K halves + park", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<8, 2, 3, 2, 1, true, true, true>("8 waves, 96 pt + park, ring 2", d_w, blob, d_out, d_ws, 8, reps, ref);
    run<8, 2, 4, 2, 1, true, true, true>("8 waves, 128 pt + park, ring 2", d_w, blob, d_out, d_ws, 6, reps, ref);
    run<8, 2, 4, 4, 1, true, true, true>("8 waves, 128 pt + park, ring 4", d_w, blob, d_out, d_ws, 6, reps, ref);
    run<8, 2, 2, 4, 1, false, true, true>("8 waves, 64 pt + park (x1 check)", d_w, blob, d_out, d_ws, 12, reps, ref);
    return 0;
}
