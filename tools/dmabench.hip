// tools/dmabench.hip -- can a workgroup's LDS-DMA staging of its next source box run under the VALU work of the current one?
// The question behind an LDS-staged form of the fused kernel: per half tile (64 x 32 output pixels) a workgroup
// would copy a 38-row x 320-byte box of the source frame into LDS (12 x global_load_lds_dwordx4 of 1 KB, cold data) while its
// four waves spend ~2,000 issue cycles each on coordinates and interpolation, read their taps from LDS and store 2 x 768 bytes
// per wave; one barrier per half tile hands the double buffer over.  This harness runs exactly that skeleton with the
// arithmetic replaced by an f64 FMA chain of the same issue time, in every combination of {VALU, DMA, LDS tap reads, stores},
// 4 workgroups of 256 threads per CU (39 KB of LDS each), over 128 4K frames in the fused kernel's tile order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

constexpr int W = 3840, H = 2160, STEP = W*3, TX = W/64, TY = (H + 63)/64;      // 60 x 34 tiles
constexpr int BOX_ROWS = 38, BOX_CHUNKS = 20;                                   // 20 x 16 bytes per row
constexpr int BOX_BYTES = 12*1024;                                              // 760 chunks used
constexpr int COEF_BYTES = 14*1024 + 768;                                       // stands in for the RP tables a workgroup keeps

typedef __attribute__((address_space(3))) void* lds_ptr;

template <bool VALU, bool DMA, bool TAPS, bool STORE, int PAD = 0>
__global__ void __launch_bounds__(256, 4) k_tile(const uint8_t* __restrict__ src, size_t sframe, uint8_t* __restrict__ dst, size_t dframe,
                                                 unsigned n_work, int fma_iters, double* sink)
{
    __shared__ __attribute__((aligned(16))) uint8_t box[2][BOX_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t coef[COEF_BYTES + PAD];         // PAD: LDS nobody uses, to lower the number of resident workgroups
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < BOX_BYTES/4; i += 256) { ((unsigned*)box[0])[i] = i; ((unsigned*)box[1])[i] = ~i; }
    for (int i = tid; i < COEF_BYTES/8; i += 256) ((double*)coef)[i] = 1.0 + 1e-9*i;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)box[0], lds1 = (unsigned)(uintptr_t)(lds_ptr)box[1];
    double a0 = 1.0 + tid*1e-9, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
    const double m = ((double*)coef)[tid & 7], c = 1e-12;
    unsigned acc = 0;
    auto issue = [&](unsigned v, int half, int buf) {          // this wave's 3 of the 12 copies of box (item v, half)
        const unsigned frame = v/(TX*TY), t = v - frame*(TX*TY), ty = t/TX, tx = t - ty*TX;
        const size_t base = (size_t)frame*sframe + (size_t)min((int)(ty*64 + half*32), H - BOX_ROWS)*STEP + (size_t)min((int)(tx*192), STEP - 320);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int i = wave*3 + j, ch = min(i*64 + lane, BOX_ROWS*BOX_CHUNKS - 1);
            const unsigned off = (unsigned)(ch/BOX_CHUNKS)*STEP + (unsigned)(ch % BOX_CHUNKS)*16;
            const uint8_t* p = src + base;
            const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)((buf ? lds1 : lds0) + (unsigned)i*1024u));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(m0v), "s"(p) : "memory");
        }
    };
    const unsigned n_it = (n_work - blockIdx.x + gridDim.x - 1)/gridDim.x;
    if (DMA && n_it) issue(blockIdx.x, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (unsigned k = 0; k < n_it; k++) {
        const unsigned v = blockIdx.x + k*gridDim.x;
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            if (DMA) {                                           // start the copy of the next half tile's box into the other buffer
                if (half == 0) issue(v, 1, buf ^ 1);
                else if (k + 1 < n_it) issue(v + gridDim.x, 0, buf ^ 1);
            }
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                if (TAPS) {                                      // 4 pixels x 2 rows x (2 + 1 aligned dwords) from the box
                    const unsigned a = (buf ? lds1 : lds0) + (unsigned)((pass*16 + (tid >> 4)) % 36)*320u + (unsigned)(tid & 15)*12u;
                    unsigned r0, r1, r2, r3; unsigned long long q0, q1, q2, q3;
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(q0) : "v"(a + e*24)); asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(r0) : "v"(a + e*24));
                        asm volatile("ds_read2_b32 %0, %1 offset0:80 offset1:81" : "=v"(q1) : "v"(a + e*24)); asm volatile("ds_read_b32 %0, %1 offset:328" : "=v"(r1) : "v"(a + e*24));
                        asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:4" : "=v"(q2) : "v"(a + e*24)); asm volatile("ds_read_b32 %0, %1 offset:20" : "=v"(r2) : "v"(a + e*24));
                        asm volatile("ds_read2_b32 %0, %1 offset0:83 offset1:84" : "=v"(q3) : "v"(a + e*24)); asm volatile("ds_read_b32 %0, %1 offset:340" : "=v"(r3) : "v"(a + e*24));
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        acc ^= r0 ^ r1 ^ r2 ^ r3 ^ (unsigned)q0 ^ (unsigned)q1 ^ (unsigned)q2 ^ (unsigned)q3;
                    }
                }
                if (VALU) {
#pragma unroll 1
                    for (int i = 0; i < fma_iters; i++) {
                        a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                        a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
                    }
                }
                if (STORE) {
                    const unsigned frame = v/(TX*TY), t = v - frame*(TX*TY), ty = t/TX, tx = t - ty*TX;
                    const int y = min((int)(ty*64 + half*32 + pass*16 + (tid >> 4)), H - 1);
                    uint8_t* d = dst + (size_t)frame*dframe + (size_t)y*STEP + tx*192 + (tid & 15)*12;
                    struct alignas(4) D3 { unsigned a, b, c; };
                    *(D3*)d = D3{acc, (unsigned)__double2loint(a0), (unsigned)k};
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the copies started at the top of this half tile have landed
            __syncthreads();                                      // everybody has finished reading `buf`; the other buffer is complete
            buf ^= 1;
        }
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456 || acc == 0x12345u) sink[tid] = a0;
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 128;
    const int fma_iters = argc > 2 ? atoi(argv[2]) : 28;           // 8 x 28 f64 fma = 224 per pass: ~1,000 issue cycles
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t sframe = (size_t)STEP*H;
    uint8_t *src, *dst; double* sink;
    CK(hipMalloc((void**)&src, sframe*frames)); CK(hipMalloc((void**)&dst, sframe*frames)); CK(hipMalloc((void**)&sink, 256*8));
    CK(hipMemset(src, 3, sframe*frames)); CK(hipMemset(dst, 0, sframe*frames));
    const unsigned n_work = (unsigned)(TX*TY)*frames, blocks = cus*32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("device %s, %d CUs; %d 4K frames, %u work items (64 x 64 tiles, two half tiles each), %u workgroups, %d x 8 f64 fma per pass\n",
           prop.name, cus, frames, n_work, blocks, fma_iters);
    auto run = [&](const char* name, auto kern) {
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, sframe, dst, sframe, n_work, fma_iters, sink);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, src, sframe, dst, sframe, n_work, fma_iters, sink);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        CK(hipGetLastError());
        printf("%-44s %8.3f ms = %6.2f us per frame\n", name, best, 1e3*best/frames);
    };
    run("VALU only", k_tile<true, false, false, false>);
    run("DMA only (cold boxes)", k_tile<false, true, false, false>);
    run("VALU + DMA", k_tile<true, true, false, false>);
    run("VALU + LDS tap reads", k_tile<true, false, true, false>);
    run("VALU + DMA + LDS tap reads", k_tile<true, true, true, false>);
    run("stores only", k_tile<false, false, false, true>);
    run("VALU + stores", k_tile<true, false, false, true>);
    run("VALU + DMA + LDS tap reads + stores", k_tile<true, true, true, true>);
    run("DMA + stores", k_tile<false, true, false, true>);
    run("everything, 3 workgroups per CU (52 KB LDS)", k_tile<true, true, true, true, 13*1024>);
    run("everything, 2 workgroups per CU (79 KB LDS)", k_tile<true, true, true, true, 40*1024>);
    run("VALU only, 3 workgroups per CU", k_tile<true, false, false, false, 13*1024>);
    run("VALU only, 2 workgroups per CU", k_tile<true, false, false, false, 40*1024>);
    return 0;
}
