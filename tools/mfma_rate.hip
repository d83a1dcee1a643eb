// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and beside f64 VALU work (tools only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: 4 independent MFMA chains; 1: VALU fma only; 2: both interleaved; 3: one dependent MFMA chain
__global__ void __launch_bounds__(256) k(double* out, int n, double a0)
{
    double4_t c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    double a = a0 + threadIdx.x, b = a0*2, f0 = a, f1 = b, f2 = a + b, f3 = a - b;
    for (int i = 0; i < n; i++) {
        if (MODE == 0 || MODE == 2) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
        }
        if (MODE == 3) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c0, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; j++) { f0 = __fma_rn(f0, a, b); f1 = __fma_rn(f1, a, b); f2 = __fma_rn(f2, a, b); f3 = __fma_rn(f3, a, b); }
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
}
template <int MODE> void run(const char* name, int waves_per_simd, double* d)
{
    const int n = 2000, blocks = 256*waves_per_simd;      // 256 CUs x 4 SIMDs: one 256-thread block = 1 wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1e-9);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, n, 1e-9); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms*1e-3*2.4e9/n/waves_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms -> %.0f cycles @2.4GHz per iteration per wave (4 MFMA and/or 64 f64 FMA)\n", name, waves_per_simd, ms, cyc);
}
int main()
{
    double* d; hipMalloc(&d, 256*8*256*8);
    for (int w = 1; w <= 2; w++) { run<0>("mfma x4 independent", w, d); run<3>("mfma x4 dependent", w, d); run<1>("valu 64 fma", w, d); run<2>("mfma x4 + valu 64 fma", w, d); }
    return 0;
}
