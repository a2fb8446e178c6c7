// LDS atomic throughput on gfx950: cycles per wave-level instruction for ds_add_f32 / ds_add_u32 / ds_add_u64 and a plain
// read-modify-write, under three address patterns (lane-distinct banks, pseudo-random, all lanes one address).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_atomic.hip -o tools/ubench/lds_atomic.bin
#include <hip/hip_runtime.h>
#include <cstdio>

enum { OP_F32 = 0, OP_U32 = 1, OP_U64 = 2, OP_RMW = 3, OP_F32_RTN = 4, OP_F64 = 5 };

template <int OP>
__global__ void __launch_bounds__(1024) k(int pattern, int iters, float *out, unsigned long long *cyc, int active = 64) {
    __shared__ unsigned long long lds[8192];  // 64 KiB
    float *lf = reinterpret_cast<float *>(lds);
    unsigned *lu = reinterpret_cast<unsigned *>(lds);
    double *ld = reinterpret_cast<double *>(lds);
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = pattern == 0 ? lane : pattern == 1 ? (lane * 2654435761u >> 20) & 4095 : 7;
    const unsigned step = pattern == 0 ? 64 : pattern == 1 ? 977 : 0;
    float ret = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
        if (OP == OP_F32) atomicAdd(lf + idx, 1.0f);
        if (OP == OP_F32_RTN) ret += atomicAdd(lf + idx, 1.0f);
        if (OP == OP_U32) atomicAdd(lu + idx, 1u);
        if (OP == OP_U64 && lane < active) atomicAdd(lds + idx, 1ull);
        if (OP == OP_F64 && lane < active) atomicAdd(ld + idx, 1.0);  // ds_add_f64 (round 6: is the fp64 form as slow as ds_add_f32?)
        if (OP == OP_RMW) lf[idx] += 1.0f;
        idx = (idx + step) & 4095;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lf[threadIdx.x] + ret;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP> void run(const char *name, float *out, unsigned long long *cyc) {
    for (int pattern = 0; pattern < 3; ++pattern) {
        const int iters = pattern == 2 ? 256 : 4096;
        for (int threads : {64, 256, 1024}) {
            hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, pattern, iters, out, cyc);
            hipDeviceSynchronize();
            unsigned long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-12s %-8s waves/CU=%d: %8.1f clk per wave instruction, %6.2f lanes/clk/CU\n", name,
                   pattern == 0 ? "distinct" : pattern == 1 ? "random" : "same", threads / 64, (double)c / iters,
                   64.0 * (threads / 64) * iters / (double)c);
        }
    }
}

// partial exec masks (round 6): does a 64-bit LDS atomic cost per instruction or per active lane?
template <int OP> void run_masked(const char *name, float *out, unsigned long long *cyc) {
    for (int active : {64, 32, 16, 8, 4, 1}) {
        const int iters = 4096, threads = 1024;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, 1, iters, out, cyc, active);
        hipDeviceSynchronize();
        unsigned long long c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-12s random   waves/CU=16 active lanes %2d: %8.1f clk per wave instruction (%5.1f clk per instruction CU-wide)\n", name, active,
               (double)c / iters, (double)c / iters / 16.0);
    }
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<OP_F32>("ds_add_f32", out, cyc);
    run<OP_F32_RTN>("ds_add_rtn_f32", out, cyc);
    run<OP_U32>("ds_add_u32", out, cyc);
    run<OP_U64>("ds_add_u64", out, cyc);
    run<OP_F64>("ds_add_f64", out, cyc);
    run<OP_RMW>("plain rmw", out, cyc);
    run_masked<OP_U64>("ds_add_u64", out, cyc);
    run_masked<OP_F64>("ds_add_f64", out, cyc);
    return 0;
}
