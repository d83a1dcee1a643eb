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
__global__ void __launch_bounds__(256) k_gather(const uint8_t* __restrict__ base, uint32_t* out, int lane_stride, int rowstep, int iters, uint32_t region, int active_mod)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    if ((int)((lane*7u + 3u) % 16u) >= active_mod) { out[blockIdx.x*256 + threadIdx.x] = 0; return; }   // active_mod/16 of the lanes load
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

// Streaming variant (cold lines): a wave walks down an image-like region; per "pass" it issues 8 loads the way the fused
// kernel does for 4 pixels x 2 source rows.  layout 0: a lane owns 4 adjacent pixels (lane stride 12 B, the 4 loads of a
// row are 3 B apart); layout 1: adjacent lanes own adjacent pixels (lane stride 3 B, the 4 loads of a row are 192 B apart).
// mode bits: 1 = a wave covers two output rows (lanes 32..63 one row further down), 2 = sloped rows (the source row
// advances by one every 16 pixels, like a 3.6 degree roll), 4 = one 12-byte store per lane and pass, 8 = odd-byte alignment
template <int W>
__global__ void __launch_bounds__(256) k_stream(const uint8_t* __restrict__ base, uint32_t* out, int layout, int rowstep, int rows, size_t region, int mode,
                                                uint8_t* __restrict__ sink)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    size_t col = ((size_t)blockIdx.x*768u) % (size_t)(rowstep - 2048);
    size_t strip = ((size_t)blockIdx.x*768u) / (size_t)(rowstep - 2048);
    const uint8_t* p0 = base + (strip*(size_t)(rows + 80)*rowstep) % region + col;
    const unsigned l2 = (mode & 1) ? (lane & 31) : lane;
    const unsigned rowsel = (mode & 1) ? (lane >> 5) : 0;
    for (int y = wave; y < rows; y += 4) {
        const uint8_t* r0 = p0 + (size_t)((mode & 1) ? 2*y + (int)rowsel : y)*rowstep;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned px = layout == 0 ? l2*4u + i : l2 + i*64u;
            unsigned o = px*3u + ((mode & 8) ? 1u : 0u);
            if (mode & 2) o += (px >> 4)*(unsigned)rowstep;
            acc ^= ld<W>(r0 + o);
            acc ^= ld<W>(r0 + rowstep + o);
        }
        if (mode & 4) {
            struct alignas(4) W3 { uint32_t a, b, c; };
            *(W3*)(sink + ((size_t)blockIdx.x*rows + y)*768u*2 + lane*12u) = W3{acc, acc + 1, acc + 2};
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}

template <int W>
static void run_stream(const uint8_t* buf, uint32_t* out, int layout, int rowstep, size_t region, int mode = 0, uint8_t* sink = nullptr)
{
    const int rows = 256, blocks = 256*8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_stream<W>, dim3(blocks), dim3(256), 0, 0, buf, out, layout, rowstep, 8, region, mode, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream<W>, dim3(blocks), dim3(256), 0, 0, buf, out, layout, rowstep, rows, region, mode, sink);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*rows*8/256.0;
    const double ns = ms*1e6/instr_per_cu;
    printf("  stream layout %d mode %2d width %2d B: %6.2f ns per wave-load per CU (= %5.1f cycles at 2.1 GHz)\n", layout, mode, W, ns, ns*2.1);
}

// Store cost: every lane writes W contiguous bytes (lane stride W), rows of 64*W bytes; NT = nontemporal hint.
template <int N> struct alignas(4) SW { uint32_t v[N]; };
template <int W, int NT>
__global__ void __launch_bounds__(256) k_store(uint8_t* __restrict__ sink, int iters, uint32_t* out)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t v = lane*2654435761u;
    uint8_t* p = sink + ((size_t)blockIdx.x*4 + wave)*(size_t)iters*64*W + lane*W;      // host checks the extent
    for (int it = 0; it < iters; it++) {
        uint32_t* q = (uint32_t*)(p + (size_t)it*64*W);
        if (NT) {
#pragma unroll
            for (int k = 0; k < W/4; k++) __builtin_nontemporal_store(v + k, q + k);
        } else {
            SW<W/4> x;
#pragma unroll
            for (int k = 0; k < W/4; k++) x.v[k] = v + k;
            *(SW<W/4>*)q = x;
        }
        v = v*3 + 1;
    }
    out[blockIdx.x*256 + threadIdx.x] = v;
}

template <int W, int NT>
static void run_store(uint8_t* sink, size_t sink_bytes, uint32_t* out)
{
    const int iters = 256, blocks = 256*8;
    if ((size_t)blocks*4*iters*64*W > sink_bytes) { printf("  store width %d: sink too small, skipped\n", W); return; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_store<W, NT>), dim3(blocks), dim3(256), 0, 0, sink, 8, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_store<W, NT>), dim3(blocks), dim3(256), 0, 0, sink, iters, out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*4*iters/256.0;
    const double ns = ms*1e6/instr_per_cu;
    printf("  store width %2d B%s: %6.2f ns per wave-store per CU (= %5.1f cycles at 2.1 GHz, %5.1f B/clk/CU, %.0f GB/s)\n", W, NT ? " nontemporal" : "            ",
           ns, ns*2.1, 64.0*W/(ns*2.1), (double)blocks*4*iters*64.0*W/(ms*1e-3)/1e9);
}

template <int W>
static void run(const uint8_t* buf, uint32_t* out, int lane_stride, int rowstep, uint32_t region, int active_mod = 16)
{
    const int iters = 2000, blocks = 256*8;          // 8 blocks of 4 waves per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, 10, region, active_mod);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<W>, dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, iters, region, active_mod);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*4*iters*8/256.0;
    const double ns = ms*1e6/instr_per_cu;
    if (active_mod != 16) printf("  [%2d/16 lanes active]", active_mod);
    printf("  width %2d B  lane stride %3d B : %6.2f ns = %5.1f cycles per wave-instr per CU  (%5.1f B/clk/CU)\n", W, lane_stride, ns, ns*2.4, 64.0*W/(ns*2.4));
}

// Aligned variant: every lane address is a multiple of ALIGN bytes (ALIGN = 4 or the access width), vector loads of W bytes.
template <int W> struct AlT;
template <> struct AlT<4>  { typedef uint32_t T; };
template <> struct AlT<8>  { typedef u2 T; };
template <> struct AlT<12> { typedef u3 T; };
template <> struct AlT<16> { typedef u4 T; };
template <int W, int ALIGN> __device__ __forceinline__ uint32_t ld_al(const uint8_t* p)
{
    struct alignas(ALIGN) V { uint32_t v[W/4]; };
    const V x = *(const V*)p;
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < W/4; k++) r ^= x.v[k];
    return r;
}
template <int W, int ALIGN>
__global__ void __launch_bounds__(256) k_gather_al(const uint8_t* __restrict__ base, uint32_t* out, int lane_stride, int rowstep, int iters, uint32_t region)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    uint32_t off0 = lane*lane_stride + wave*rowstep*8 + (blockIdx.x & 7)*64;
    for (int it = 0; it < iters; it++) {
        uint32_t o = off0 + (it & 15)*ALIGN;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t a = o + k*rowstep;
            a = a < region ? a : a - region;
            acc ^= ld_al<W, ALIGN>(base + a);
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}
template <int W, int ALIGN>
static void run_al(const uint8_t* buf, uint32_t* out, int lane_stride, int rowstep, uint32_t region)
{
    const int iters = 2000, blocks = 256*8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather_al<W, ALIGN>), dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, 10, region);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather_al<W, ALIGN>), dim3(blocks), dim3(256), 0, 0, buf, out, lane_stride, rowstep, iters, region);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*4*iters*8/256.0;
    const double ns = ms*1e6/instr_per_cu;
    printf("  aligned %2d  width %2d B  lane stride %3d B : %6.2f ns = %5.1f cycles (2.1 GHz) per wave-instr per CU  (%5.1f B/clk/CU)\n", ALIGN, W, lane_stride, ns, ns*2.1, 64.0*W/(ns*2.1));
}

// K1-like footprint, cache resident: a wave-load covers R source rows (64/R lanes each, lane stride 12 B, aligned 12-byte loads),
// optionally with the row advancing by one every `slope` lanes (a rolled camera).  Cost per wave-instruction vs R.
template <int W>
__global__ void __launch_bounds__(256) k_rows(const uint8_t* __restrict__ base, uint32_t* out, int R, int rowstep, int iters, uint32_t region, int slope)
{
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lpr = 64u/(unsigned)R, lr = lane/lpr, lc = lane%lpr;
    uint32_t acc = 0;
    uint32_t off0 = lr*(unsigned)rowstep + lc*12u + (slope ? (lc/(unsigned)slope)*(unsigned)rowstep : 0u) + wave*(unsigned)rowstep*(unsigned)R + (blockIdx.x & 7)*64;
    for (int it = 0; it < iters; it++) {
        uint32_t o = off0 + (it & 15)*4;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint32_t a = o + (k & 1)*rowstep + (k >> 1)*4;
            a = a < region ? a : a - region;
            acc ^= ld_al<W, 4>(base + a);
        }
    }
    out[blockIdx.x*256 + threadIdx.x] = acc;
}
template <int W>
static void run_rows(const uint8_t* buf, uint32_t* out, int R, int rowstep, uint32_t region, int slope)
{
    const int iters = 2000, blocks = 256*8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rows<W>), dim3(blocks), dim3(256), 0, 0, buf, out, R, rowstep, 10, region, slope);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rows<W>), dim3(blocks), dim3(256), 0, 0, buf, out, R, rowstep, iters, region, slope);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_cu = (double)blocks*4*iters*8/256.0;
    const double ns = ms*1e6/instr_per_cu;
    printf("  rows per wave-load %2d (slope %2d)  width %2d B : %6.2f ns = %5.1f cycles (2.1 GHz) per wave-instr per CU\n", R, slope, W, ns, ns*2.1);
}

int main()
{
    const uint32_t region = 1u << 20;      // 1 MiB: L2 resident, rows spread like an image tile
    uint8_t* buf; uint32_t* out;
    CK(hipMalloc(&buf, region + 65536)); CK(hipMemset(buf, 1, region + 65536));
    CK(hipMalloc(&out, 256*8*256*4));
    const int rowstep = 11520;             // a 4K u8 row
    const int strides[] = {12};
    for (int s : strides) {
        printf("lane stride %d B\n", s);
        run<2>(buf, out, s, rowstep, region);
        run<4>(buf, out, s, rowstep, region);
        run<8>(buf, out, s, rowstep, region);
        run<12>(buf, out, s, rowstep, region);
        run<16>(buf, out, s, rowstep, region);
    }
    {
        const size_t big = (size_t)1 << 31;
        uint8_t* b2; CK(hipMalloc(&b2, big + (1 << 24))); CK(hipMemset(b2, 1, big + (1 << 24)));
        printf("streaming (cold) gathers, 4 px x 2 rows per lane and pass\n");
        const size_t sink_bytes = ((size_t)1 << 31) + 65536;       // k_stream mode 4 needs 2048*256*1536 = 805 MB, k_store up to 2 GiB
        uint8_t* sink; CK(hipMalloc(&sink, sink_bytes));
        for (int mode : {0, 1, 2, 3, 4, 8, 7, 15}) run_stream<8>(b2, out, 0, rowstep, big, mode, sink);
        run_stream<8>(b2, out, 1, rowstep, big, 0, sink);
        printf("streaming stores\n");
        run_store<4, 0>(sink, sink_bytes, out); run_store<8, 0>(sink, sink_bytes, out); run_store<12, 0>(sink, sink_bytes, out); run_store<16, 0>(sink, sink_bytes, out);
        run_store<4, 1>(sink, sink_bytes, out); run_store<8, 1>(sink, sink_bytes, out); run_store<12, 1>(sink, sink_bytes, out); run_store<16, 1>(sink, sink_bytes, out);
        CK(hipFree(sink));
        CK(hipFree(b2));
    }
    printf("aligned loads (cache resident)\n");
    run_al<4, 4>(buf, out, 4, rowstep, region);   run_al<4, 4>(buf, out, 12, rowstep, region);  run_al<4, 4>(buf, out, 64, rowstep, region);
    run_al<8, 4>(buf, out, 12, rowstep, region);  run_al<8, 8>(buf, out, 8, rowstep, region);   run_al<8, 8>(buf, out, 16, rowstep, region);
    run_al<12, 4>(buf, out, 12, rowstep, region); run_al<12, 4>(buf, out, 16, rowstep, region);
    run_al<16, 4>(buf, out, 12, rowstep, region); run_al<16, 4>(buf, out, 16, rowstep, region); run_al<16, 16>(buf, out, 16, rowstep, region);
    run_al<16, 16>(buf, out, 32, rowstep, region);
    printf("K1-like footprints (aligned, cache resident)\n");
    for (int R : {1, 2, 4, 8, 16}) run_rows<12>(buf, out, R, rowstep, region, 0);
    for (int R : {1, 2, 4, 8}) run_rows<12>(buf, out, R, rowstep, region, 4);
    for (int R : {1, 4}) run_rows<4>(buf, out, R, rowstep, region, 0);
    for (int R : {1, 4}) run_rows<16>(buf, out, R, rowstep, region, 0);
    printf("partial exec masks, width 8, lane stride 12\n");
    for (int a : {16, 12, 8, 4, 2, 1}) run<8>(buf, out, 12, rowstep, region, a);
    return 0;
}
