// tools/ashrpk_probe.hip -- what v_ashr_pk_u8_i32 (gfx950) writes: which source lands in which byte, how it saturates, and
// what happens to the upper half of the destination.  Also the issue cost of v_dot2c_i32_i16 / v_dot2_i32_i16 / v_med3_i32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void k_probe(const int* a, const int* b, uint32_t* out, int n)
{
    const int i = threadIdx.x;
    if (i < n) {
        uint32_t d = 0xAAAAAAAAu;
        asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 15" : "+v"(d) : "v"(a[i]), "v"(b[i]));
        out[i] = d;
    }
}
typedef short short2_t __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, uint32_t seed, int loops)
{
    uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed*3 + threadIdx.x, c = b ^ 0x1234567u;
#define R8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#define I0(x) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
#define I1(x) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c));
#define I2(x) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(x) : "v"(b));
#define I3(x) asm volatile("v_ashr_pk_u8_i32 %0, %0, %1, 15" : "+v"(x) : "v"(b));
#define I4(x) asm volatile("v_ashrrev_i32 %0, 15, %0" : "+v"(x));
#define I5(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(b));
#define I6(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));
#define I7(x) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(x) : "v"(b));
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { R8(I0) } if (OP == 1) { R8(I1) } if (OP == 2) { R8(I2) } if (OP == 3) { R8(I3) }
            if (OP == 4) { R8(I4) } if (OP == 5) { R8(I5) } if (OP == 6) { R8(I6) } if (OP == 7) { R8(I7) }
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    const int n = 8;
    int ha[n] = {0, 1 << 15, 255 << 15, 256 << 15, -1, 100 << 15, (77 << 15) + 32767, 0x7fffffff};
    int hb[n] = {3 << 15, -(5 << 15), 254 << 15, 1000 << 15, 9 << 15, 200 << 15, (78 << 15) - 1, (int)0x80000000};
    int *da, *db; uint32_t* dout; uint32_t ho[n];
    CK(hipMalloc((void**)&da, sizeof ha)); CK(hipMalloc((void**)&db, sizeof hb)); CK(hipMalloc((void**)&dout, 256*8*256*4));
    CK(hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, da, db, dout, n);
    CK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) printf("v_ashr_pk_u8_i32(s0 = %d*2^15%+d, s1 = %d*2^15%+d, 15) on 0xAAAAAAAA -> 0x%08x\n", ha[i] >> 15, ha[i] & 32767, hb[i] >> 15, hb[i] & 32767, ho[i]);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[8] = {"v_dot2c_i32_i16 (VOP2)", "v_dot2_i32_i16 (VOP3P)", "v_med3_i32", "v_ashr_pk_u8_i32", "v_ashrrev_i32", "v_mov_b32", "v_cndmask_b32", "v_mov_b32_dpp"};
    const int loops = 2000, blocks = 256*8;
    for (int op = 0; op < 8; op++) for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0, 0));
        switch (op) {
        case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 5: hipLaunchKernelGGL(k_rate<5>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 6: hipLaunchKernelGGL(k_rate<6>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        case 7: hipLaunchKernelGGL(k_rate<7>, dim3(blocks), dim3(256), 0, 0, dout, 5u, loops); break;
        }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("%-26s %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", names[op], ms, ms*1e-3*2.4e9/(8.0*loops*64));
    }
    return 0;
}
