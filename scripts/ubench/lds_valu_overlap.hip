// Micro-benchmark: do LDS returns (ds_read_b128) and VALU work overlap on gfx950
//   (a) inside one wave (8 reads issued, then 64 independent VALU instructions, then the wait),
//   (b) between two waves that share a SIMD (wave 0 reads only, wave 4 VALU only)?
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_valu_overlap lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define READS8 \
    "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t" \
    "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\t"

// role: 0 idle, 1 reads only, 2 VALU only (64 pk_add/max3), 3 reads then VALU then wait (same wave), 4 reads, wait, VALU,
//       5 reads, then per read: stepped wait + 4 pk_add + 4 max3 that consume it (the kernels' pattern), 6 the same VALU without reads
__global__ void k(float* out, unsigned long long* cyc, int iters, const int* roles) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int role = roles[wv];
    const unsigned addr = 16u * lane + 1024u * wv;
    f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
    f32x2 p0{1.f, 2.f}, p1{3.f, 4.f}, p2{5.f, 6.f}, p3{7.f, 8.f};
    float a0 = lane, a1 = 1.f, a2 = 2.f, a3 = 3.f, acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role != 0)
    for (int it = 0; it < iters; ++it) {
        if (role == 5 || role == 6) {
            if (role == 5)
                asm volatile(READS8 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
#define STEP(N, R)                                                        \
            asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(R));           \
            {                                                             \
                const f32x2 lo_ = f32x2{R.x, R.y}, hi_ = f32x2{R.z, R.w}; \
                const f32x2 c0 = lo_ + q0, c1 = hi_ + q0, c2 = lo_ + q1, c3 = hi_ + q1; \
                a0 = fmaxf(fmaxf(a0, c0.x), c0.y);                        \
                a3 = fmaxf(fmaxf(a3, c1.x), c1.y);                        \
                b0 = fmaxf(fmaxf(b0, c2.x), c2.y);                        \
                b1 = fmaxf(fmaxf(b1, c3.x), c3.y);                        \
            }
            f32x2 q0{1.f, 2.f}, q1{3.f, 4.f};
            float b0 = 0.f, b1 = 0.f;
            if (role == 6) { r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = f32x4{a0, a1, a2, a3}; asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); }
            STEP(7, r0) STEP(6, r1) STEP(5, r2) STEP(4, r3) STEP(3, r4) STEP(2, r5) STEP(1, r6) STEP(0, r7)
            acc += b0 + b1;
            continue;
        }
        if (role == 1 || role == 3 || role == 4)
            asm volatile(READS8 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        if (role == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (role >= 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                             "v_max3_f32 %5, %5, %6, %7\n\tv_max3_f32 %6, %6, %5, %7\n\tv_max3_f32 %7, %7, %5, %6\n\tv_max3_f32 %8, %8, %5, %6"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2), "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        }
        if (role == 1 || role == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (role == 1 || role >= 3) acc += r0.x + r7.w;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = acc + a0 + a1 + a2 + a3 + p0.x + p1.x + p3.y;
    if (lane == 0) cyc[wv] = t1 - t0;
}

void run(const char* name, std::vector<int> roles) {
    float* out; unsigned long long* cyc; int* dr;
    (void)hipMalloc(&out, 1024 * sizeof(float));
    (void)hipMalloc(&cyc, 16 * sizeof(unsigned long long));
    (void)hipMalloc(&dr, 16 * sizeof(int));
    const int waves = (int)roles.size();
    roles.resize(16, 0);
    (void)hipMemcpy(dr, roles.data(), 16 * sizeof(int), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 8192 * sizeof(float), 0, out, cyc, iters, dr);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(16);
    (void)hipMemcpy(h.data(), cyc, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    printf("%-58s:", name);
    for (int w = 0; w < waves; ++w) if (roles[w]) printf("  wave%d(role %d) %6.1f cyc/iter", w, roles[w], (double)h[w] / iters);
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(dr);
}

int main() {
    run("reads only (8 x b128 + wait)", {1});
    run("VALU only (32 pk_add + 32 max3)", {2});
    run("one wave: reads, VALU, then wait", {3});
    run("one wave: reads, wait, VALU", {4});
    run("two waves, different SIMDs: reads | VALU", {1, 2});
    run("two waves, same SIMD (0 and 4): reads | VALU", {1, 0, 0, 0, 2});
    run("two waves, same SIMD: reads | reads", {1, 0, 0, 0, 1});
    run("two waves, same SIMD: VALU | VALU", {2, 0, 0, 0, 2});
    run("two waves, same SIMD: both reads+VALU+wait", {3, 0, 0, 0, 3});
    run("two waves, different SIMDs: both reads+VALU+wait", {3, 3});
    run("one wave: reads, stepped waits, consuming VALU", {5});
    run("one wave: the same VALU, no reads", {6});
    run("three waves (3 SIMDs): reads, stepped waits, VALU", {5, 5, 5});
    run("six waves: reads, stepped waits, consuming VALU", {5, 5, 5, 5, 5, 5});
    run("six waves: the same VALU, no reads", {6, 6, 6, 6, 6, 6});
    run("six waves: reads only", {1, 1, 1, 1, 1, 1});
    return 0;
}
