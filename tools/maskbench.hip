// tools/maskbench.hip -- does a gathered 12-byte load cost less when half of the wave's lanes are masked off?
// (prices "share one tap load between two adjacent pixels" for the fused kernel: the second pixel's load would be issued
// with only the lanes that cannot share.)  Cache-resident footprint, K1-like addresses: lane l reads 12 bytes at
// row (l/16)*stride + 12*(l%16) + phase.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
struct alignas(4) D3 { uint32_t a, b, c; };
template <int MASK>   // 0: all lanes, 1: odd lanes off, 2: three of four lanes off, 3: lanes 32..63 off
__global__ void __launch_bounds__(256) k(const uint8_t* base, unsigned stride, int loops, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    unsigned off = (threadIdx.x >> 6)*4096u + (lane >> 4)*stride + (lane & 15)*12u;
    uint32_t acc = 0;
    const bool on = MASK == 0 ? true : MASK == 1 ? !(lane & 1) : MASK == 2 ? !(lane & 3) : lane < 32;
    for (int l = 0; l < loops; l++) {
        if (on) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const D3 v = *(const D3*)(base + ((off + u*stride*4u) & 0x3FFFFu & ~3u));
                acc += v.a ^ v.b ^ v.c;
            }
        }
        off += 48u;
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}
int main()
{
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc((void**)&buf, 1 << 19)); CK(hipMemset(buf, 1, 1 << 19)); CK(hipMalloc((void**)&out, 4096*256*4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int loops = 2000, blocks = 256*8;
    const char* names[4] = {"all 64 lanes", "odd lanes off (32 active)", "1 of 4 lanes active (16)", "upper half off (32 active)"};
    for (int m = 0; m < 4; m++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, 0));
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, buf, 11520u, loops, out);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, buf, 11520u, loops, out);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, buf, 11520u, loops, out);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, buf, 11520u, loops, out);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("%-30s %.3f ms -> %.1f cycles per wave-load per CU (2.4 GHz)\n", names[m], ms, ms*1e-3*2.4e9/(32.0*loops*8));
        }
    }
    return 0;
}
