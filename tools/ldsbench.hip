// tools/ldsbench.hip -- what it costs on gfx950 to move the fused kernel's byte extraction from the VALU to the LDS pipe:
// per-lane private LDS slots written from the tap loads (ds_write / LDS-DMA) and read back at byte offsets (unaligned
// ds_read_b32, ds_read_u8_d16), alone and beside a VALU stream.  Every kernel runs 4 workgroups of 256 threads per CU (4 waves
// per SIMD, the fused kernel's occupancy) and reports cycles of CU time per wave-instruction at the 2.4 GHz maximum clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(2); } } while (0)

typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int SLOT = 12;                  // bytes per lane slot
constexpr int NSLOT = 8;                  // slots per lane (8 tap rows of a 4-pixel pass)

#define RD16(INS, OFF) \
    asm volatile(INS " %0, %1 offset:" #OFF : "=v"(r0) : "v"(a)); asm volatile(INS " %0, %1 offset:" #OFF "+3072" : "=v"(r1) : "v"(a)); \
    asm volatile(INS " %0, %1 offset:" #OFF "+6144" : "=v"(r2) : "v"(a)); asm volatile(INS " %0, %1 offset:" #OFF "+9216" : "=v"(r3) : "v"(a));

// mode 0: ds_read_b32; 1: ds_read_u8_d16 / _hi pairs; 2: ds_read_b64; 3: ds_read_u16_d16 pairs
template <int MODE>
__global__ void __launch_bounds__(256) k_read(unsigned* out, int phase, int loops)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256*SLOT*NSLOT + 64];
    for (int i = threadIdx.x; i < (256*SLOT*NSLOT + 64)/4; i += 256) ((unsigned*)buf)[i] = i*2654435761u;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)buf + threadIdx.x*SLOT + phase;
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0, acc = 0;
    unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (MODE == 0) { RD16("ds_read_b32", 0) RD16("ds_read_b32", 1) RD16("ds_read_b32", 2) RD16("ds_read_b32", 3072*4 + 0) }
            else if (MODE == 1) { RD16("ds_read_u8_d16", 0) RD16("ds_read_u8_d16_hi", 3) RD16("ds_read_u8_d16", 1) RD16("ds_read_u8_d16_hi", 4) }
            else if (MODE == 3) { RD16("ds_read_u16_d16", 0) RD16("ds_read_u16_d16_hi", 3) RD16("ds_read_u16_d16", 1) RD16("ds_read_u16_d16_hi", 4) }
            else {
                asm volatile("ds_read_b64 %0, %1" : "=v"(q0) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(q1) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(q2) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(q3) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:12288" : "=v"(q0) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:15360" : "=v"(q1) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:18432" : "=v"(q2) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:21504" : "=v"(q3) : "v"(a));
                asm volatile("ds_read_b64 %0, %1" : "=v"(q0) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(q1) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:6144" : "=v"(q2) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:9216" : "=v"(q3) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:12288" : "=v"(q0) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:15360" : "=v"(q1) : "v"(a));
                asm volatile("ds_read_b64 %0, %1 offset:18432" : "=v"(q2) : "v"(a)); asm volatile("ds_read_b64 %0, %1 offset:21504" : "=v"(q3) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc ^= r0 ^ r1 ^ r2 ^ r3 ^ (unsigned)q0 ^ (unsigned)q1 ^ (unsigned)q2 ^ (unsigned)q3;
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}

// ds_write_b96 / ds_write_b128 to the lane's slots (12-byte / 16-byte stride)
template <int W>
__global__ void __launch_bounds__(256) k_write(unsigned* out, int phase, int loops)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256*16*NSLOT + 64];
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)buf + threadIdx.x*(W == 3 ? 12 : 16);
    u3 d = {threadIdx.x, threadIdx.x*3u, (unsigned)phase};
    u4 d4 = {threadIdx.x, threadIdx.x*3u, (unsigned)phase, 7u};
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (W == 3) {
                asm volatile("ds_write_b96 %0, %1" :: "v"(a), "v"(d) : "memory"); asm volatile("ds_write_b96 %0, %1 offset:3072" :: "v"(a), "v"(d) : "memory");
                asm volatile("ds_write_b96 %0, %1 offset:6144" :: "v"(a), "v"(d) : "memory"); asm volatile("ds_write_b96 %0, %1 offset:9216" :: "v"(a), "v"(d) : "memory");
            } else {
                asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(d4) : "memory"); asm volatile("ds_write_b128 %0, %1 offset:4096" :: "v"(a), "v"(d4) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:8192" :: "v"(a), "v"(d4) : "memory"); asm volatile("ds_write_b128 %0, %1 offset:12288" :: "v"(a), "v"(d4) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    out[blockIdx.x*256 + threadIdx.x] = ((unsigned*)buf)[threadIdx.x];
}

// Gathered 12-byte tap loads with the fused kernel's footprint (16 lanes along a row at ~12-byte steps, 4 rows per wave, all
// workgroups of a CU inside one 64-row band so that the taps hit L1/L2), into registers (MODE 0) or straight into LDS (MODE 1).
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(unsigned* out, const uint8_t* src, unsigned step, int loops)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256*SLOT*NSLOT + 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned col = (unsigned)((blockIdx.x & 15)*192 + (lane & 15)*12), row0 = (unsigned)(lane >> 4) + 4u*wave;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)buf + wave*768u;
    unsigned acc = 0;
    u3 t0 = {0, 0, 0}, t1 = t0, t2 = t0, t3 = t0;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned row = (row0 + 16u*(unsigned)k + (unsigned)l) & 63u;
            const unsigned off = (row*step + col) & ~3u;
            if (MODE == 0) {
                asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(t0) : "v"(off), "s"(src) : "memory");
                asm volatile("global_load_dwordx3 %0, %1, %2 offset:12" : "=v"(t1) : "v"(off), "s"(src) : "memory");
                asm volatile("global_load_dwordx3 %0, %1, %2 offset:24" : "=v"(t2) : "v"(off), "s"(src) : "memory");
                asm volatile("global_load_dwordx3 %0, %1, %2 offset:36" : "=v"(t3) : "v"(off), "s"(src) : "memory");
            } else {
                const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)k*3072u));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\tglobal_load_lds_dwordx3 %1, %3 offset:12\n\t"
                             "global_load_lds_dwordx3 %1, %3 offset:24\n\tglobal_load_lds_dwordx3 %1, %3 offset:36\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(off), "s"(m), "s"(src) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 0) acc ^= t0.x ^ t1.y ^ t2.z ^ t3.x;
    }
    __syncthreads();
    out[blockIdx.x*256 + threadIdx.x] = acc ^ ((unsigned*)buf)[threadIdx.x];
}

// A VALU stream (64 v_dot4 per iteration, 8 independent chains) with NR LDS byte reads and NW 12-byte LDS writes beside it.
template <int NR, int NW>
__global__ void __launch_bounds__(256) k_mixed(unsigned* out, int phase, int loops)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256*SLOT*NSLOT + 64];
    for (int i = threadIdx.x; i < (256*SLOT*NSLOT + 64)/4; i += 256) ((unsigned*)buf)[i] = i*2654435761u;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)buf + threadIdx.x*SLOT + phase;
    unsigned v[8], b = threadIdx.x*3 + 1, r[16] = {};
    u3 d = {threadIdx.x, threadIdx.x*3u, (unsigned)phase};
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int l = 0; l < loops; l++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(v[i]) : "v"(b));
            if (k < NW) asm volatile("ds_write_b96 %0, %1 offset:12288" :: "v"(a), "v"(d) : "memory");
            if (2*k < NR)     asm volatile("ds_read_u8_d16 %0, %1 offset:1" : "=v"(r[2*k]) : "v"(a));
            if (2*k + 1 < NR) asm volatile("ds_read_u8_d16_hi %0, %1 offset:4" : "=v"(r[2*k + 1]) : "v"(a));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; i++) if (i < NR) b ^= r[i] & 1u;
    }
    unsigned s = b;
    for (int i = 0; i < 8; i++) s ^= v[i];
    out[blockIdx.x*256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = 2.4e9;
    printf("device %s, %d CUs; cycles at 2.4 GHz, 4 workgroups of 256 threads per CU (16 waves per CU)\n", prop.name, cus);
    unsigned* out; CK(hipMalloc((void**)&out, (size_t)cus*4*256*4));
    const unsigned step = 11520;
    uint8_t* src; CK(hipMalloc((void**)&src, (size_t)step*80)); CK(hipMemset(src, 7, (size_t)step*80));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = cus*4;
    auto time = [&](auto launch) { launch(20); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); launch(2000); CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
                                   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms; };
    auto report = [&](const char* name, double ms, double inst_per_wave) {
        printf("%-58s %8.3f ms -> %6.2f cycles of CU time per wave-instruction\n", name, ms, ms*1e-3*clk/(16.0*inst_per_wave));
    };
    for (int ph = 0; ph < 4; ph++) {
        char nm[96];
        snprintf(nm, sizeof nm, "ds_read_b32, 12-byte lane stride, byte phase %d", ph);
        report(nm, time([&](int n) { hipLaunchKernelGGL(k_read<0>, dim3(blocks), dim3(256), 0, 0, out, ph, n); }), 2000.0*64);
    }
    report("ds_read_u8_d16 / _hi", time([&](int n) { hipLaunchKernelGGL(k_read<1>, dim3(blocks), dim3(256), 0, 0, out, 0, n); }), 2000.0*64);
    report("ds_read_u16_d16 / _hi, even offsets", time([&](int n) { hipLaunchKernelGGL(k_read<3>, dim3(blocks), dim3(256), 0, 0, out, 0, n); }), 2000.0*64);
    report("ds_read_u16_d16 / _hi, odd offsets", time([&](int n) { hipLaunchKernelGGL(k_read<3>, dim3(blocks), dim3(256), 0, 0, out, 1, n); }), 2000.0*64);
    for (int ph = 0; ph < 4; ph += 1) {
        char nm[96];
        snprintf(nm, sizeof nm, "ds_read_b64, 12-byte lane stride, byte phase %d", ph);
        report(nm, time([&](int n) { hipLaunchKernelGGL(k_read<2>, dim3(blocks), dim3(256), 0, 0, out, ph, n); }), 2000.0*64);
    }
    report("ds_write_b96, 12-byte lane stride", time([&](int n) { hipLaunchKernelGGL(k_write<3>, dim3(blocks), dim3(256), 0, 0, out, 0, n); }), 2000.0*32);
    report("ds_write_b128, 16-byte lane stride", time([&](int n) { hipLaunchKernelGGL(k_write<4>, dim3(blocks), dim3(256), 0, 0, out, 0, n); }), 2000.0*32);
    report("global_load_dwordx3, K1-like gather (cache resident)", time([&](int n) { hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, out, src, step, n); }), 2000.0*16);
    report("global_load_lds_dwordx3, same gather", time([&](int n) { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, out, src, step, n); }), 2000.0*16);
    const double valu = time([&](int n) { hipLaunchKernelGGL((k_mixed<0, 0>), dim3(blocks), dim3(256), 0, 0, out, 0, n); });
    report("64 v_dot4 per iteration (per dot4)", valu, 2000.0*64);
    report("  + 8 ds_read_u8_d16 per iteration (per dot4)", time([&](int n) { hipLaunchKernelGGL((k_mixed<8, 0>), dim3(blocks), dim3(256), 0, 0, out, 1, n); }), 2000.0*64);
    report("  + 16 ds_read_u8_d16 per iteration (per dot4)", time([&](int n) { hipLaunchKernelGGL((k_mixed<16, 0>), dim3(blocks), dim3(256), 0, 0, out, 1, n); }), 2000.0*64);
    report("  + 16 reads + 2 ds_write_b96 per iteration (per dot4)", time([&](int n) { hipLaunchKernelGGL((k_mixed<16, 2>), dim3(blocks), dim3(256), 0, 0, out, 1, n); }), 2000.0*64);
    report("  + 16 reads + 4 ds_write_b96 per iteration (per dot4)", time([&](int n) { hipLaunchKernelGGL((k_mixed<16, 4>), dim3(blocks), dim3(256), 0, 0, out, 1, n); }), 2000.0*64);
    return 0;
}
