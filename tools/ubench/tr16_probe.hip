#include <hip/hip_runtime.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short lds[64*4*4];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    // lane supplies address of 4 contiguous elements
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + l*4));
    for (int j = 0; j < 4; ++j) out[l*4+j] = v[j];
}
int main() {
    short h[1024]; for (int i=0;i<1024;++i) h[i]=i;
    short *d,*o; hipMalloc(&d,2048); hipMalloc(&o,512);
    hipMemcpy(d,h,2048,hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,d,o);
    short r[256]; hipMemcpy(r,o,512,hipMemcpyDeviceToHost);
    for (int l=0;l<64;++l){ printf("lane %2d:",l); for(int j=0;j<4;++j) printf(" %4d",r[l*4+j]); printf("\n"); }
}
