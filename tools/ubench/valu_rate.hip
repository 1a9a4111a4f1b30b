// Micro-benchmark: VALU issue rate on gfx950 as a function of waves per SIMD, for v_fma_f32, v_pk_fma_f32,
// v_mul+v_add (unfused), v_min/v_max, v_cndmask.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
    float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {   // 8 independent v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 1) {   // 8 x (4 v_pk_fma_f32) = 8 packed instrs... do 8 pk instrs
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            } else if (KIND == 2) {   // 8 v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 3) {   // 8 v_max_f32
                asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 4) {   // 8 v_cndmask_b32 (vcc)
                asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                             "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "s"(mask));
            } else if (KIND == 5) {   // 8 v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa));
            } else if (KIND == 6) {   // 8 v_max3_f32
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            } else if (KIND == 7) {   // 8 dependent-chain v_fma on ONE register (latency)
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(x0) : "v"(a), "v"(b));
            } else if (KIND == 8) {   // v_cmp_lt_f32 -> sgpr pair
                asm volatile("v_cmp_lt_f32_e64 %8, %0, %9\n v_cmp_lt_f32_e64 %8, %1, %9\n v_cmp_lt_f32_e64 %8, %2, %9\n v_cmp_lt_f32_e64 %8, %3, %9\n"
                             "v_cmp_lt_f32_e64 %8, %4, %9\n v_cmp_lt_f32_e64 %8, %5, %9\n v_cmp_lt_f32_e64 %8, %6, %9\n v_cmp_lt_f32_e64 %8, %7, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+s"(mask) : "v"(a));
            } else if (KIND == 9) {   // v_add_u32
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 10) {   // v_sub_f32 / v_add_f32
                asm volatile("v_sub_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_sub_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 11) {   // v_min_u32 / v_max_u32
                asm volatile("v_min_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_max_u32 %3, %3, %8\n"
                             "v_min_u32 %4, %4, %8\n v_max_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_max_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 12) {   // v_rcp_f32 (transcendental)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if (KIND == 13) {   // v_mul_lo_u32
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
            } else if (KIND == 14) {   // v_med3_f32
                asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                             "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(mask & 1) + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND> double run(int blocks, int iters, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    const int iters = 20000;                   // x 64 wave-instrs per iteration
    const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_min/max_f32", "v_cndmask_b32", "v_pk_mul/add_f32", "v_min3/max3_f32", "v_fma dependent", "v_cmp_lt_f32", "v_add_u32", "v_sub/add_f32", "v_min/max_u32", "v_rcp_f32", "v_mul_lo_u32", "v_med3_f32"};
    printf("cycles per wave-instruction per SIMD (assuming 2.4 GHz), blocks/CU = waves/SIMD\n");
    for (int kind = 0; kind < 15; ++kind) {
        printf("%-18s", names[kind]);
        for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {
            int blocks = 256 * per_cu;
            double ms = 0;
            switch (kind) { case 0: ms = run<0>(blocks, iters, d); break; case 1: ms = run<1>(blocks, iters, d); break; case 2: ms = run<2>(blocks, iters, d); break;
                            case 3: ms = run<3>(blocks, iters, d); break; case 4: ms = run<4>(blocks, iters, d); break; case 5: ms = run<5>(blocks, iters, d); break;
                            case 6: ms = run<6>(blocks, iters, d); break; case 7: ms = run<7>(blocks, iters, d); break; case 8: ms = run<8>(blocks, iters, d); break; case 9: ms = run<9>(blocks, iters, d); break; case 10: ms = run<10>(blocks, iters, d); break; case 11: ms = run<11>(blocks, iters, d); break; case 12: ms = run<12>(blocks, iters, d); break; case 13: ms = run<13>(blocks, iters, d); break; case 14: ms = run<14>(blocks, iters, d); break; }
            double instr_per_simd = (double)iters * 64 * per_cu;          // each wave issues iters*64 instrs; per_cu waves per SIMD
            printf("  w=%d: %6.2f ms %5.2f cyc/instr", per_cu, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
