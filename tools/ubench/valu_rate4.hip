// Micro-benchmark 4: does the VGPR bank (register number mod 4) of the three sources of v_fma_f32 change its issue rate on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define DEFK(NAME, BODY) \
template <int D> __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) { \
 asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %1\n v_mov_b32 v45, %1\n v_mov_b32 v46, %1\n v_mov_b32 v47, %1\n" \
              "v_mov_b32 v48, %1\n v_mov_b32 v49, %1\n v_mov_b32 v50, %1\n v_mov_b32 v51, %1\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n s_mov_b32 s20, 0.5\n" \
              :: "v"(a), "v"(b) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20"); \
 for (int i=0;i<iters;++i) { _Pragma("unroll") for (int u=0;u<8;++u) { \
   asm volatile(BODY ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","s20"); } } \
 float r; asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v52" : "=v"(r)); out[blockIdx.x*blockDim.x+threadIdx.x]=r; }
// dst/src0 = v40..v43 (banks 0..3) ; a-type regs v44..v47 ; b-type v48..v51
DEFK(k_all_diff,  "v_fma_f32 v40, v40, v45, v50\n v_fma_f32 v41, v41, v46, v51\n v_fma_f32 v42, v42, v47, v48\n v_fma_f32 v43, v43, v44, v49\n v_fma_f32 v40, v40, v45, v50\n v_fma_f32 v41, v41, v46, v51\n v_fma_f32 v42, v42, v47, v48\n v_fma_f32 v43, v43, v44, v49")
DEFK(k_two_same,  "v_fma_f32 v40, v40, v45, v49\n v_fma_f32 v41, v41, v46, v50\n v_fma_f32 v42, v42, v47, v51\n v_fma_f32 v43, v43, v44, v48\n v_fma_f32 v40, v40, v45, v49\n v_fma_f32 v41, v41, v46, v50\n v_fma_f32 v42, v42, v47, v51\n v_fma_f32 v43, v43, v44, v48")
DEFK(k_src0_same, "v_fma_f32 v40, v40, v44, v49\n v_fma_f32 v41, v41, v45, v50\n v_fma_f32 v42, v42, v46, v51\n v_fma_f32 v43, v43, v47, v48\n v_fma_f32 v40, v40, v44, v49\n v_fma_f32 v41, v41, v45, v50\n v_fma_f32 v42, v42, v46, v51\n v_fma_f32 v43, v43, v47, v48")
DEFK(k_all_same,  "v_fma_f32 v40, v40, v44, v48\n v_fma_f32 v41, v41, v45, v49\n v_fma_f32 v42, v42, v46, v50\n v_fma_f32 v43, v43, v47, v51\n v_fma_f32 v40, v40, v44, v48\n v_fma_f32 v41, v41, v45, v49\n v_fma_f32 v42, v42, v46, v50\n v_fma_f32 v43, v43, v47, v51")
DEFK(k_two_ops,   "v_fma_f32 v40, v40, v45, v45\n v_fma_f32 v41, v41, v46, v46\n v_fma_f32 v42, v42, v47, v47\n v_fma_f32 v43, v43, v44, v44\n v_fma_f32 v40, v40, v45, v45\n v_fma_f32 v41, v41, v46, v46\n v_fma_f32 v42, v42, v47, v47\n v_fma_f32 v43, v43, v44, v44")
DEFK(k_dst_other, "v_fma_f32 v52, v40, v45, v50\n v_fma_f32 v53, v41, v46, v51\n v_fma_f32 v54, v42, v47, v48\n v_fma_f32 v55, v43, v44, v49\n v_fma_f32 v52, v40, v45, v50\n v_fma_f32 v53, v41, v46, v51\n v_fma_f32 v54, v42, v47, v48\n v_fma_f32 v55, v43, v44, v49")
DEFK(k_mul_diff,  "v_mul_f32 v40, v40, v45\n v_mul_f32 v41, v41, v46\n v_mul_f32 v42, v42, v47\n v_mul_f32 v43, v43, v44\n v_mul_f32 v40, v40, v45\n v_mul_f32 v41, v41, v46\n v_mul_f32 v42, v42, v47\n v_mul_f32 v43, v43, v44")
DEFK(k_mul_same,  "v_mul_f32 v40, v40, v44\n v_mul_f32 v41, v41, v45\n v_mul_f32 v42, v42, v46\n v_mul_f32 v43, v43, v47\n v_mul_f32 v40, v40, v44\n v_mul_f32 v41, v41, v45\n v_mul_f32 v42, v42, v46\n v_mul_f32 v43, v43, v47")
DEFK(k_fma_sgpr,  "v_fma_f32 v40, v40, v45, s20\n v_fma_f32 v41, v41, v46, s20\n v_fma_f32 v42, v42, v47, s20\n v_fma_f32 v43, v43, v44, s20\n v_fma_f32 v40, v40, v45, s20\n v_fma_f32 v41, v41, v46, s20\n v_fma_f32 v42, v42, v47, s20\n v_fma_f32 v43, v43, v44, s20")
DEFK(k_mul_sgpr,  "v_mul_f32 v40, s20, v40\n v_mul_f32 v41, s20, v41\n v_mul_f32 v42, s20, v42\n v_mul_f32 v43, s20, v43\n v_mul_f32 v40, s20, v40\n v_mul_f32 v41, s20, v41\n v_mul_f32 v42, s20, v42\n v_mul_f32 v43, s20, v43")
DEFK(k_mul_lit,   "v_mul_f32 v40, 0x3f8ccccd, v40\n v_mul_f32 v41, 0x3f8ccccd, v41\n v_mul_f32 v42, 0x3f8ccccd, v42\n v_mul_f32 v43, 0x3f8ccccd, v43\n v_mul_f32 v40, 0x3f8ccccd, v40\n v_mul_f32 v41, 0x3f8ccccd, v41\n v_mul_f32 v42, 0x3f8ccccd, v42\n v_mul_f32 v43, 0x3f8ccccd, v43")
DEFK(k_mul_inline,"v_mul_f32 v40, 0.5, v40\n v_mul_f32 v41, 0.5, v41\n v_mul_f32 v42, 0.5, v42\n v_mul_f32 v43, 0.5, v43\n v_mul_f32 v40, 2.0, v40\n v_mul_f32 v41, 2.0, v41\n v_mul_f32 v42, 2.0, v42\n v_mul_f32 v43, 2.0, v43")
DEFK(k_add_sgpr,  "v_add_f32 v40, s20, v40\n v_add_f32 v41, s20, v41\n v_add_f32 v42, s20, v42\n v_add_f32 v43, s20, v43\n v_add_f32 v40, s20, v40\n v_add_f32 v41, s20, v41\n v_add_f32 v42, s20, v42\n v_add_f32 v43, s20, v43")
DEFK(k_fma_inline,"v_fma_f32 v40, v40, v45, 1.0\n v_fma_f32 v41, v41, v46, 1.0\n v_fma_f32 v42, v42, v47, 1.0\n v_fma_f32 v43, v43, v44, 1.0\n v_fma_f32 v40, v40, v45, 1.0\n v_fma_f32 v41, v41, v46, 1.0\n v_fma_f32 v42, v42, v47, 1.0\n v_fma_f32 v43, v43, v44, 1.0")
DEFK(k_fmaak,     "v_fmaak_f32 v40, v40, v45, 0x3f8ccccd\n v_fmaak_f32 v41, v41, v46, 0x3f8ccccd\n v_fmaak_f32 v42, v42, v47, 0x3f8ccccd\n v_fmaak_f32 v43, v43, v44, 0x3f8ccccd\n v_fmaak_f32 v40, v40, v45, 0x3f8ccccd\n v_fmaak_f32 v41, v41, v46, 0x3f8ccccd\n v_fmaak_f32 v42, v42, v47, 0x3f8ccccd\n v_fmaak_f32 v43, v43, v44, 0x3f8ccccd")
DEFK(k_fma_neg,   "v_fma_f32 v40, v40, v45, -v50\n v_fma_f32 v41, v41, v46, -v51\n v_fma_f32 v42, v42, v47, -v48\n v_fma_f32 v43, v43, v44, -v49\n v_fma_f32 v40, v40, v45, -v50\n v_fma_f32 v41, v41, v46, -v51\n v_fma_f32 v42, v42, v47, -v48\n v_fma_f32 v43, v43, v44, -v49")
DEFK(k_xor_lit,   "v_xor_b32 v40, 0x80000000, v40\n v_xor_b32 v41, 0x80000000, v41\n v_xor_b32 v42, 0x80000000, v42\n v_xor_b32 v43, 0x80000000, v43\n v_xor_b32 v40, 0x80000000, v40\n v_xor_b32 v41, 0x80000000, v41\n v_xor_b32 v42, 0x80000000, v42\n v_xor_b32 v43, 0x80000000, v43")
DEFK(k_mul_e64,   "v_mul_f32_e64 v40, v40, v45\n v_mul_f32_e64 v41, v41, v46\n v_mul_f32_e64 v42, v42, v47\n v_mul_f32_e64 v43, v43, v44\n v_mul_f32_e64 v40, v40, v45\n v_mul_f32_e64 v41, v41, v46\n v_mul_f32_e64 v42, v42, v47\n v_mul_f32_e64 v43, v43, v44")
DEFK(k_mul_e64_neg,"v_mul_f32_e64 v40, -v40, v45\n v_mul_f32_e64 v41, -v41, v46\n v_mul_f32_e64 v42, -v42, v47\n v_mul_f32_e64 v43, -v43, v44\n v_mul_f32_e64 v40, -v40, v45\n v_mul_f32_e64 v41, -v41, v46\n v_mul_f32_e64 v42, -v42, v47\n v_mul_f32_e64 v43, -v43, v44")
template <class K> double run(K kern, int blocks, int iters, float* d) { hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,10,1.0001f,0.5f); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,iters,1.0001f,0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); return ms; }
#define LINE(NAME) { printf("%-22s", #NAME); for (int w=1; w<=4; w*=2) { double ms=run(NAME<0>,256*w,iters,d); printf("  w=%d %5.2f", w, ms*1e-3*2.4e9/((double)iters*64*w)); } printf("\n"); }
int main(){ float* d; (void)hipMalloc(&d,256*256*8*sizeof(float)); const int iters=20000;
 printf("v_fma_f32 dst,src0,src1,src2 — cycles per wave-instruction per SIMD at 2.4 GHz nominal; w = waves per SIMD\n");
 LINE(k_all_diff) LINE(k_two_same) LINE(k_src0_same) LINE(k_all_same) LINE(k_two_ops) LINE(k_dst_other) LINE(k_mul_diff) LINE(k_mul_same) LINE(k_fma_sgpr) LINE(k_mul_sgpr) LINE(k_mul_lit) LINE(k_mul_inline) LINE(k_add_sgpr) LINE(k_fma_inline) LINE(k_fmaak) LINE(k_fma_neg) LINE(k_xor_lit) LINE(k_mul_e64) LINE(k_mul_e64_neg)
 return 0; }
