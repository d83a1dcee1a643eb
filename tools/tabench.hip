// tools/tabench.hip -- cost of per-lane gathers through the vector memory path (TA/TCP) on gfx950, as a function of
// bytes per lane and of the byte distance between neighbouring lanes.  The region is small (L1/L2 resident), so the
// number is the address/tag pipeline's, not HBM's.  Prints ns and cycles (at 2.4 GHz) of CU time per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u3 __attribute__((ext_vector_type(3)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

struct __attribute__((packed, aligned(1))) P1 { uint32_t a; };
struct __attribute__((packed, aligned(1))) P2 { uint32_t a, b; };
struct __attribute__((packed, aligned(1))) P3 { uint32_t a, b, c; };
struct __attribute__((packed, aligned(1))) P4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) PS { uint16_t a; };

template <int W> __device__ __forceinline__ uint32_t ld(const uint8_t* p);
template <> __device__ __forceinline__ uint32_t ld<2>(const uint8_t* p) { return ((const PS*)p)->a; }
template <> __device__ __forceinline__ uint32_t ld<4>(const uint8_t* p) { return ((const P1*)p)->a; }
template <> __device__ __forceinline__ uint32_t ld<8>(const uint8_t* p) { P2 v = *(const P2*)p; return v.a ^ v.b; }
template <> __device__ __forceinline__ uint32_t ld<12>(const uint8_t* p) { P3 v = *(const P3*)p; return v.a ^ v.b ^ v.c; }
template <> __device__ __forceinline__ uint32_t ld<16>(const uint8_t* p) { P4 v = *(const P4*)p; return v.a ^ v.b ^ v.c ^ v.d; }

// Each lane: address = base + row*rowstep + lane*lane_stride + small per-iteration shift; 8 independent loads per iteration.
template <int W>
__global__ void __launch_bounds__(256) k_gather(const uint8_t* __restrict__ base, uint32_t* out, int lane_stride, int rowstep, int iters, uint32_t region)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    uint32_t off0 = lane*lane_stride + wave*rowstep*8 + (blockIdx.x & 7)*64;
    for (int it = 0; it < iters; it++) {
        uint32_t o = off0 + (it & 15)*3;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t a = o + k*rowstep;
            a = a < region ? a : a - region;
            acc ^= ld<W>(base + a);
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}

template <int W>
static void run(const uint8_t* buf, uint32_t* out, int lane_stride, int rowstep, uint32_t region)
{
    const int iters = 2000, blocks = 256*8;          // 8 blocks of 4 waves per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, 10, region);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, iters, region);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*4*iters*8/256.0;
    const double ns = ms*1e6/instr_per_cu;
    printf("  width %2d B  lane stride %3d B : %6.2f ns = %5.1f cycles per wave-instr per CU  (%5.1f B/clk/CU)\n", W, lane_stride, ns, ns*2.4, 64.0*W/(ns*2.4));
}

int main()
{
    const uint32_t region = 1u << 20;      // 1 MiB: L2 resident, rows spread like an image tile
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, region + 65536)); CK(hipMemset(buf, 1, region + 65536));
    CK(hipMalloc(&out, 256*8*256*4));
    const int rowstep = 11520;             // a 4K u8 row
    const int strides[] = {3, 4, 6, 12, 16, 24, 64};
    for (int s : strides) {
        printf("lane stride %d B\n", s);
        run<2>(buf, out, s, rowstep, region);
        run<4>(buf, out, s, rowstep, region);
        run<8>(buf, out, s, rowstep, region);
        run<12>(buf, out, s, rowstep, region);
        run<16>(buf, out, s, rowstep, region);
    }
    return 0;
}
