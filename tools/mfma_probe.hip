// Probe: operand/result lane layout of v_mfma_f64_16x16x4_f64 on gfx950 (tools only; not product code).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D)   // A[16][4], B[4][16] row-major; D[64 lanes][4]
{
    const int l = threadIdx.x, g = l / 16, c = l % 16;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c*4 + g], B[g*16 + c], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) D[l*4 + v] = acc[v];
}
int main()
{
    double hA[64], hB[64], hD[256], ref[16][16];
    srand(1);
    for (int i = 0; i < 64; i++) { hA[i] = rand() % 17 - 8; hB[i] = rand() % 13 - 6; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int kk = 0; kk < 4; kk++) s += hA[i*4+kk]*hB[kk*16+j]; ref[i][j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    if (hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    int ok1 = 1, ok2 = 1, ok3 = 1, ok4 = 1;
    for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
        const int g = l / 16, c = l % 16; const double d = hD[l*4+v];
        if (d != ref[4*g+v][c]) ok1 = 0;      // i = 4g+v, j = c
        if (d != ref[g+4*v][c]) ok2 = 0;      // i = g+4v, j = c
        if (d != ref[c][4*g+v]) ok3 = 0;      // transposed variants
        if (d != ref[c][g+4*v]) ok4 = 0;
    }
    printf("A:(i=l%%16,k=l/16) B:(k=l/16,j=l%%16) assumed.  D layout: i=4g+v:%d  i=g+4v:%d  T(4g+v):%d  T(g+4v):%d\n", ok1, ok2, ok3, ok4);
    return 0;
}
