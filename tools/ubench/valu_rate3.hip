// Micro-benchmark 3: issue rate of integer min/max, 64-bit address ops and a few others on gfx950 (cycles per wave-instruction per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define DEFK(NAME, ASM) \
template <int D> __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) { \
 float x0=threadIdx.x+1.5f,x1=x0+1,x2=x0+2,x3=x0+3,x4=x0+4,x5=x0+5,x6=x0+6,x7=x0+7; \
 for (int i=0;i<iters;++i) { _Pragma("unroll") for (int u=0;u<8;++u) { \
   asm volatile(ASM : "+v"(x0),"+v"(x1),"+v"(x2),"+v"(x3),"+v"(x4),"+v"(x5),"+v"(x6),"+v"(x7) : "v"(a),"v"(b) : "vcc"); } } \
 out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7; }
#define R2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8"
#define R3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9"
#define RC(op) op " vcc, %0, %8\n" op " vcc, %1, %8\n" op " vcc, %2, %8\n" op " vcc, %3, %8\n" op " vcc, %4, %8\n" op " vcc, %5, %8\n" op " vcc, %6, %8\n" op " vcc, %7, %8"
DEFK(k_max_i32, R2("v_max_i32"))
DEFK(k_min_u32, R2("v_min_u32"))
DEFK(k_max3_i32, R3("v_max3_i32"))
DEFK(k_min3_u32, R3("v_min3_u32"))
DEFK(k_med3_f32, R3("v_med3_f32"))
DEFK(k_or_b32, R2("v_or_b32"))
DEFK(k_or3_b32, R3("v_or3_b32"))
DEFK(k_add3_u32, R3("v_add3_u32"))
DEFK(k_cmp_le_i32, RC("v_cmp_le_i32"))
DEFK(k_cmp_le_f32, RC("v_cmp_le_f32"))
DEFK(k_max_f32, R2("v_max_f32"))
DEFK(k_fma, R3("v_fma_f32"))
DEFK(k_xad, R3("v_xad_u32"))
DEFK(k_sad, R3("v_sad_u32"))
DEFK(k_mul_lo, R2("v_mul_lo_u32"))
DEFK(k_mul_u24, R2("v_mul_u32_u24"))
DEFK(k_cvt_pk, R2("v_cvt_pkrtz_f16_f32"))
DEFK(k_max_f16, R2("v_max_f16"))
DEFK(k_pk_max_f16, R2("v_pk_max_f16"))
DEFK(k_max_u16, R2("v_max_u16"))
template <class K> double run(K kern, int blocks, int iters, float* d) { hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,10,1.0001f,0.5f); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,iters,1.0001f,0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); return ms; }
#define LINE(NAME) { printf("%-22s", #NAME); for (int w=1; w<=4; w*=2) { double ms=run(NAME<0>,256*w,iters,d); printf("  w=%d %5.2f", w, ms*1e-3*2.4e9/((double)iters*64*w)); } printf("\n"); }
int main(){ float* d; (void)hipMalloc(&d,256*256*8*sizeof(float)); const int iters=20000;
 printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal; w = waves per SIMD\n");
 LINE(k_fma) LINE(k_max_f32) LINE(k_max_i32) LINE(k_min_u32) LINE(k_max3_i32) LINE(k_min3_u32) LINE(k_med3_f32) LINE(k_or_b32) LINE(k_or3_b32) LINE(k_add3_u32)
 LINE(k_cmp_le_i32) LINE(k_cmp_le_f32) LINE(k_xad) LINE(k_sad) LINE(k_mul_lo) LINE(k_mul_u24) LINE(k_cvt_pk) LINE(k_max_f16) LINE(k_pk_max_f16) LINE(k_max_u16)
 return 0; }
