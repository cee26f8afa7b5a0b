// Micro-benchmark: cycles per instruction per wave on gfx950 for the instructions the Viterbi kernels
// are made of, at 1 / 2 / 4 waves per SIMD (one workgroup, s_memtime around an unrolled loop).
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip ; run: ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = 1.f, a5 = 2.f, a6 = 3.f, a7 = 4.f;
    f32x2 p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7};
    const float* lp = lds + (threadIdx.x & 255);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) {  // 4 independent v_add_f32
                asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));
            } else if (OP == 1) {  // 4 independent v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2));
            } else if (OP == 2) {  // 4 independent v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %4, %5\n\tv_max3_f32 %2, %2, %4, %5\n\tv_max3_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5));
            } else if (OP == 3) {  // dependent chain of v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %1, %2"
                             : "+v"(a0) : "v"(a4), "v"(a5));
            } else if (OP == 4) {  // 4 ds_read_b32 + wait
                asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"((unsigned)(size_t)lp) : "memory");
            } else if (OP == 5) {  // 4 ds_read2_b32 + wait
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:64 offset1:65\n\tds_read2_b32 %2, %4 offset0:128 offset1:129\n\tds_read2_b32 %3, %4 offset0:192 offset1:193\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"((unsigned)(size_t)lp) : "memory");
            } else if (OP == 6) {  // 4 s_nop 0
                asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            } else if (OP == 7) {  // mixed: pk_add feeding max3 (2 independent chains), written in C
                p0 = p1 + p2; a0 = fmaxf(fmaxf(a0, p0.x), p0.y);
                p3 = p1 + p3; a1 = fmaxf(fmaxf(a1, p3.x), p3.y);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(p1), "+v"(p3));
            } else if (OP == 8) {  // v_cmp + v_cndmask pairs
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cmp_gt_f32 vcc, %1, %0\n\tv_cndmask_b32 %3, %3, %2, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "vcc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + p0.x + p1.x + p2.x + p3.y;
    if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_u) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 4 * 64); hipMalloc(&cyc, 64 * 8 * 64);
    const int iters = 2000;
    printf("%-28s", name);
    for (int waves : {1, 4, 8, 16}) {   // waves per workgroup (one workgroup on one CU): 4 = 1/SIMD, 8 = 2/SIMD, 16 = 4/SIMD
        hipLaunchKernelGGL(k<OP>, dim3(1), dim3(waves * 64), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        double mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        printf("  %2dw: %6.2f", waves, mx / (double)(iters * 16 * instr_per_u));
    }
    printf("   cycles per instruction per wave\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_add_f32 (indep)", 4);
    run<1>("v_pk_add_f32 (indep)", 4);
    run<2>("v_max3_f32 (indep)", 4);
    run<3>("v_max3_f32 (dependent)", 4);
    run<4>("ds_read_b32 x4 + wait", 5);
    run<5>("ds_read2_b32 x4 + wait", 5);
    run<6>("s_nop 0", 4);
    run<7>("pk_add -> max3 (2 chains)", 4);
    run<8>("v_cmp + v_cndmask", 4);
    return 0;
}
