#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int ld) {
    __shared__ __attribute__((aligned(16))) unsigned short S[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) S[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    // group g reads the 4x16 block at rows 0..3, cols 16*g.. ; lane i -> &S[(i/4)*ld + 16*g + (i%4)*4]
    const unsigned short* p = &S[(i / 4) * ld + 16 * g + (i % 4) * 4];
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short h[4096], *d, *o, r[256];
    for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 64);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
    return 0;
}
