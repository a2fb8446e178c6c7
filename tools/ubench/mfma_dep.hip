// MFMA issue-rate microbenchmark: cycles per v_mfma_f32_32x32x16_f16 as a function of the number
// of independent accumulators per wave (dependency distance) and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k(const h8 *in, float *out, int iters, unsigned long long *cyc) {
    h8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f16v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC> void run(int threads, const h8 *in, float *out, unsigned long long *cyc) {
    const int iters = 4096 / NACC * 4;
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC;  // MFMAs per wave
    const int wps = threads / 256;            // waves per SIMD
    printf("NACC=%2d waves/SIMD=%d: %6.1f cycles per MFMA per wave  -> %5.1f cycles per MFMA per SIMD   %.0f TFLOP/s\n", NACC, wps,
           c / nm, c / nm / wps, 256.0 * (threads / 64) * nm * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    h8 *in; float *out; unsigned long long *cyc;
    hipMalloc(&in, 128 * sizeof(h8)); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    _Float16 host[1024];
    for (int i = 0; i < 1024; ++i) host[i] = (_Float16)(0.01f * ((i * 37) % 101 - 50));
    hipMemcpy(in, host, sizeof(host), hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<1>(threads, in, out, cyc); run<2>(threads, in, out, cyc); run<4>(threads, in, out, cyc);
        run<6>(threads, in, out, cyc); run<8>(threads, in, out, cyc);
    }
    return 0;
}
