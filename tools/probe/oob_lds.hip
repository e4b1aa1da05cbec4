// Probe: does an out-of-range lane of buffer_load ... lds (LDS-DMA) write zeros to LDS or skip the write?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t* src, uint32_t* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x7FFFFFF0u;  // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    uint32_t h[256], *d, *o;
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    hipMalloc(&d, 1024); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
