// Micro-benchmark: cost of 8 x ds_read_b128 (+ one wait) per wave for the address patterns the banded forward
// kernels could use, with 1..6 waves of one workgroup reading at the same time (s_memtime around an unrolled loop).
// Patterns (byte address of lane l, read q = 0..7 adds 16*q):
//   0  contiguous     : 16*l                          (ideal: every lane its own 16 bytes)
//   1  window/4copies : delta window starting at lo = l + off, taken from the copy (lo&3) that aligns it;
//                       copy stride DC floats          (banded_floor_forward_kernel)
//   2  pair/4copies   : lo = 2*l + off                 (banded_floor_pair_forward_kernel)
//   3  window, one copy, b32 : 32 x ds_read_b32 at 4*(l + off + w)   (no copies, 4-byte reads)
//   4  window, one copy, 16 x ds_read_b64 from two copies (shift 0/1)
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_patterns lds_patterns.hip ; run: ./lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DC = 400;   // copy stride in floats (NP + 16 at S = 361)

template <int PAT>
__global__ void k(float* out, unsigned long long* cyc, int iters, int off, int dc) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned addr = 0;
    if (PAT == 0) addr = 16u * tid;
    if (PAT == 1) { const int lo = tid + off; addr = 4u * ((lo & 3) * dc + (lo & ~3)); }
    if (PAT == 2) { const int lo = 2 * tid + off; addr = 4u * ((lo & 3) * dc + (lo & ~3)); }
    if (PAT == 3) addr = 4u * (tid + off);
    if (PAT == 4) { const int lo = tid + off; addr = 4u * ((lo & 1) * dc + (lo & ~1)); }
    f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PAT <= 2) {
            asm volatile(
                "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
            acc += r0.x + r7.w;
        } else if (PAT == 3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a0, a1, a2, a3, a4, a5, a6, a7;
                asm volatile(
                    "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:4\n\tds_read_b32 %2, %8 offset:8\n\tds_read_b32 %3, %8 offset:12\n\t"
                    "ds_read_b32 %4, %8 offset:16\n\tds_read_b32 %5, %8 offset:20\n\tds_read_b32 %6, %8 offset:24\n\tds_read_b32 %7, %8 offset:28\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr + 32u * g) : "memory");
                acc += a0 + a7;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x2 a0, a1, a2, a3, a4, a5, a6, a7;
                asm volatile(
                    "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                    "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr + 64u * g) : "memory");
                acc += a0.x + a7.y;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[tid + blockIdx.x * blockDim.x] = acc;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + tid) >> 6] = t1 - t0;
}

template <int PAT>
void run(const char* name, int off, int dc) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * sizeof(float));
    hipMalloc(&cyc, 16 * sizeof(unsigned long long));
    const int iters = 2000;
    printf("%-34s off=%2d dc=%3d :", name, off, dc);
    for (int waves : {1, 2, 3, 4, 6, 8}) {
        hipLaunchKernelGGL(k<PAT>, dim3(1), dim3(64 * waves), 8192 * sizeof(float), 0, out, cyc, iters, off, dc);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(16);
        hipMemcpy(h.data(), cyc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
        // cycles per "window" (32 floats per lane) per wave, and implied bytes/clk for the CU
        const double per = (double)mx / iters;
        printf("  %dw %6.1f cyc (%5.1f B/clk)", waves, per, waves * 64 * 128.0 / per);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("contiguous b128", 0, DC);
    for (int off : {0, 1, 2, 3}) run<1>("window, 4 copies, b128", off, DC);
    for (int dc : {392, 396, 404, 408, 416}) run<1>("window, 4 copies, b128", 2, dc);
    for (int off : {0, 1, 2, 3}) run<2>("pair window, 4 copies, b128", off, DC);
    run<3>("window, one copy, 32 x b32", 0, DC);
    for (int off : {0, 1}) run<4>("window, 2 copies, 16 x b64", off, DC);
    return 0;
}
