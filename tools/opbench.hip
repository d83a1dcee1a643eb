// tools/opbench.hip -- issue cost of single VALU instructions on gfx950 (cycles of SIMD time per wave-instruction),
// to price the fused kernel's per-pixel budget.  Each kernel runs a long unrolled stream of independent copies of
// one instruction on every SIMD (8 waves per SIMD) and reports  time * clock / instructions-per-SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>
#include <math.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

constexpr int REP = 64;      // instructions per loop body (x8 independent register sets)
constexpr int LOOPS = 2000;

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

#define KERNEL(NAME, DECL, INS, SINK)                                                        \
__global__ void __launch_bounds__(256) NAME(double* out, double seed, int loops)             \
{                                                                                            \
    DECL                                                                                     \
    for (int l = 0; l < loops; l++) {                                                        \
        BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) \
    }                                                                                        \
    SINK                                                                                     \
}

#define DD double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double b=seed*0.5+threadIdx.x;
#define DSINK out[blockIdx.x*256+threadIdx.x] = a0+a1+a2+a3+a4+a5+a6+a7;
#define FD float a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; float b=seed*0.5f+threadIdx.x;
#define ID uint32_t a0=(uint32_t)seed,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7; uint32_t b=a0*3+threadIdx.x;

#define I_FMA64(i)  asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_MUL64(i)  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_ADD64(i)  asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_RCP64(i)  asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
#define I_RSQ64(i)  asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
#define I_SQRT64(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a##i));
#define I_RND64(i)  asm volatile("v_rndne_f64 %0, %0" : "+v"(a##i));
#define I_FMA32(i)  asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_RCP32(i)  asm volatile("v_rcp_f32 %0, %0" : "+v"(a##i));
#define I_MULLO(i)  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_MAD24(i)  asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_BFE(i)    asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a##i));
#define I_PERM(i)   asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
#define I_ADD32(i)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b));
#define I_PKMAD16(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
#define I_DOT4(i)   asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));

#define I_CVTUB1(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a##i));
#define I_CVTU32F(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a##i));
#define I_CVTF32U(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a##i));
#define I_FLOOR32(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a##i));
#define I_CVTPKU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %0, 1, %1" : "+v"(a##i) : "v"(b));
#define I_AND32(i)  asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_LSHR32(i) asm volatile("v_lshrrev_b32 %0, 5, %0" : "+v"(a##i));
#define I_MUL32(i)  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_ANDOR(i)  asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
#define I_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(a##i) : "v"(b));
#define I_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 26" : "+v"(a##i) : "v"(b));
#define I_MUL24(i)  asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define I_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a##i) : "v"(b2));
#define DD2 double a0=seed,a1=seed+1,a2=seed+2,a3=seed+3,a4=seed+4,a5=seed+5,a6=seed+6,a7=seed+7; double b2=seed*0.5+threadIdx.x;
KERNEL(k_cvtub1, ID, I_CVTUB1, DSINK)
KERNEL(k_cvtu32f, ID, I_CVTU32F, DSINK)
KERNEL(k_cvtf32u, ID, I_CVTF32U, DSINK)
KERNEL(k_floor32, FD, I_FLOOR32, DSINK)
KERNEL(k_cvtpku8, ID, I_CVTPKU8, DSINK)
KERNEL(k_and32, ID, I_AND32, DSINK)
KERNEL(k_lshr32, ID, I_LSHR32, DSINK)
KERNEL(k_mul32, FD, I_MUL32, DSINK)
KERNEL(k_andor, ID, I_ANDOR, DSINK)
KERNEL(k_lshlor, ID, I_LSHLOR, DSINK)
KERNEL(k_alignbit, ID, I_ALIGNBIT, DSINK)
KERNEL(k_mul24, ID, I_MUL24, DSINK)
KERNEL(k_pkfma32, DD2, I_PKFMA32, DSINK)
KERNEL(k_fma64, DD, I_FMA64, DSINK)
KERNEL(k_mul64, DD, I_MUL64, DSINK)
KERNEL(k_add64, DD, I_ADD64, DSINK)
KERNEL(k_rcp64, DD, I_RCP64, DSINK)
KERNEL(k_rsq64, DD, I_RSQ64, DSINK)
KERNEL(k_sqrt64, DD, I_SQRT64, DSINK)
KERNEL(k_rnd64, DD, I_RND64, DSINK)
KERNEL(k_fma32, FD, I_FMA32, DSINK)
KERNEL(k_rcp32, FD, I_RCP32, DSINK)
KERNEL(k_mullo, ID, I_MULLO, DSINK)
KERNEL(k_mad24, ID, I_MAD24, DSINK)
KERNEL(k_bfe, ID, I_BFE, DSINK)
KERNEL(k_perm, ID, I_PERM, DSINK)
KERNEL(k_add32, ID, I_ADD32, DSINK)
KERNEL(k_lshladd, ID, I_LSHLADD, DSINK)
KERNEL(k_pkmad16, ID, I_PKMAD16, DSINK)
KERNEL(k_dot4, ID, I_DOT4, DSINK)

// conversions need distinct in/out register classes
__global__ void __launch_bounds__(256) k_cvt_i32_f64(double* out, double seed, int loops)
{
    double a = seed + threadIdx.x; int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r1) : "v"(a));
            asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r2) : "v"(a)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r3) : "v"(a));
        }
    }
    out[blockIdx.x*256+threadIdx.x] = r0 + r1 + r2 + r3;
}
__global__ void __launch_bounds__(256) k_cvt_f64_i32(double* out, double seed, int loops)
{
    int a = (int)seed + threadIdx.x; double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r1) : "v"(a));
            asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r2) : "v"(a)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r3) : "v"(a));
        }
    }
    out[blockIdx.x*256+threadIdx.x] = r0 + r1 + r2 + r3;
}
__global__ void __launch_bounds__(256) k_cvt_f32_f64(double* out, double seed, int loops)
{
    double a = seed + threadIdx.x; float r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r0) : "v"(a)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r1) : "v"(a));
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r2) : "v"(a)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r3) : "v"(a));
        }
    }
    out[blockIdx.x*256+threadIdx.x] = r0 + r1 + r2 + r3;
}
// LDS b128 broadcast-ish reads (all lanes of a wave within 2 adjacent 48-byte entries)
__global__ void __launch_bounds__(256) k_ds_read_b128(double* out, double seed, int loops)
{
    __shared__ double tab[3072];
    for (int i = threadIdx.x; i < 3072; i += 256) tab[i] = seed + i;
    __syncthreads();
    double2 acc = {0, 0};
    int idx = ((threadIdx.x >> 5) & 1) + (int)seed;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 64; k++) {
            const double2 v = *(const double2*)&tab[((idx + k) & 511)*6 + 2*(k % 3)];
            acc.x += v.x; acc.y += v.y;
        }
    }
    out[blockIdx.x*256+threadIdx.x] = acc.x + acc.y;
}

__global__ void k_probe(const double* in, double* out_rcp, const float* fin, unsigned* out_pk, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        double r; asm volatile("v_rcp_f64 %0, %1" : "=v"(r) : "v"(in[i])); out_rcp[i] = r;
        unsigned d = 0; asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(d) : "v"(fin[i])); out_pk[i] = d;
    }
}

typedef void (*kern_t)(double*, double, int);

int main()
{
    double* out; CK(hipMalloc((void**)&out, 256*8*2048*sizeof(double)));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate*1e3;      // Hz (max)
    printf("device %s, %d CUs, max clock %.0f MHz\n", prop.name, cus, clk/1e6);
    struct E { const char* name; kern_t k; int per_loop; };
    std::vector<E> ks = {
        {"v_fma_f64", k_fma64, 64}, {"v_mul_f64", k_mul64, 64}, {"v_add_f64", k_add64, 64}, {"v_rcp_f64", k_rcp64, 64},
        {"v_rsq_f64", k_rsq64, 64}, {"v_sqrt_f64", k_sqrt64, 64}, {"v_rndne_f64", k_rnd64, 64}, {"v_cvt_i32_f64", k_cvt_i32_f64, 64},
        {"v_cvt_f64_i32", k_cvt_f64_i32, 64}, {"v_cvt_f32_f64", k_cvt_f32_f64, 64},
        {"v_fma_f32", k_fma32, 64}, {"v_rcp_f32", k_rcp32, 64}, {"v_mul_lo_u32", k_mullo, 64}, {"v_mad_u32_u24", k_mad24, 64},
        {"v_bfe_u32", k_bfe, 64}, {"v_perm_b32", k_perm, 64}, {"v_add_u32", k_add32, 64}, {"v_lshl_add_u32", k_lshladd, 64},
        {"v_pk_mad_u16", k_pkmad16, 64}, {"v_cvt_f32_ubyte1", k_cvtub1, 64}, {"v_cvt_u32_f32", k_cvtu32f, 64},
        {"v_cvt_f32_u32", k_cvtf32u, 64}, {"v_floor_f32", k_floor32, 64}, {"v_cvt_pk_u8_f32", k_cvtpku8, 64},
        {"v_and_b32", k_and32, 64}, {"v_lshrrev_b32", k_lshr32, 64}, {"v_mul_f32", k_mul32, 64}, {"v_and_or_b32", k_andor, 64},
        {"v_lshl_or_b32", k_lshlor, 64}, {"v_alignbit_b32", k_alignbit, 64}, {"v_mul_u32_u24", k_mul24, 64},
        {"v_pk_fma_f32", k_pkfma32, 64}, {"v_dot4_u32_u8", k_dot4, 64}, {"ds_read_b128+2add", k_ds_read_b128, 64},
    };
    {   // probes: accuracy of v_rcp_f64, rounding of v_cvt_pk_u8_f32
        const int n = 256;
        std::vector<double> hin(n), hr(n); std::vector<float> hf = {0.25f, 0.5f, 0.75f, 1.5f, 2.5f, 2.7f, 3.5f, 254.5f, 255.4f, 255.5f, 255.6f, 300.f, -1.f, -0.4f, 1e9f, 0.49999997f};
        hf.resize(n, 7.5f);
        for (int i = 0; i < n; i++) hin[i] = 0.5 + 1.7*i/n + 1e-3*(i*i % 17);
        double *din, *dr; float* df; unsigned *dp; std::vector<unsigned> hp(n);
        CK(hipMalloc((void**)&din, n*8)); CK(hipMalloc((void**)&dr, n*8)); CK(hipMalloc((void**)&df, n*4)); CK(hipMalloc((void**)&dp, n*4));
        CK(hipMemcpy(din, hin.data(), n*8, hipMemcpyHostToDevice)); CK(hipMemcpy(df, hf.data(), n*4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, din, dr, df, dp, n);
        CK(hipMemcpy(hr.data(), dr, n*8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hp.data(), dp, n*4, hipMemcpyDeviceToHost));
        double worst = 0; for (int i = 0; i < n; i++) { double e = fabs(hr[i]*hin[i] - 1.0); if (e > worst) worst = e; }
        printf("v_rcp_f64 max relative error over %d inputs: %.3e (2^%.1f)\n", n, worst, log2(worst));
        printf("v_cvt_pk_u8_f32:");
        for (int i = 0; i < 16; i++) printf(" %g->%u", hf[i], hp[i] & 0xff);
        printf("\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = cus*8;                   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    for (auto& e : ks) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5, 10);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5, LOOPS);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_per_simd = 8.0*LOOPS*e.per_loop;      // 8 waves per SIMD
        printf("%-20s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at max clock)\n", e.name, ms, ms*1e-3*clk/inst_per_simd);
    }
    return 0;
}
