// Micro-benchmark 6: issue rate of the conversions a compact BVH node would need on gfx950 — v_fma_mix_f32 with an f16 source
// (low / high half), v_cvt_f32_f16, v_cvt_f32_ubyte0..3, v_cvt_f32_u32 — against v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CL "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55"
#define DEFK(NAME, BODY) \
template <int D> __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) { \
 asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %1\n v_mov_b32 v45, %1\n v_mov_b32 v46, %1\n v_mov_b32 v47, %1\n" \
              "v_mov_b32 v48, 0x3c003c00\n v_mov_b32 v49, 0x3c003c00\n v_mov_b32 v50, 0x3c003c00\n v_mov_b32 v51, 0x3c003c00\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n" \
              :: "v"(a), "v"(b) : CL); \
 for (int i=0;i<iters;++i) { _Pragma("unroll") for (int u=0;u<8;++u) { asm volatile(BODY ::: CL); } } \
 float r; asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v52\n v_add_f32 %0, %0, v53" : "=v"(r)); out[blockIdx.x*blockDim.x+threadIdx.x]=r; }
// v48..v51 hold two f16 1.0 each; v44..v47 = b (0.5); accumulators v40..v43 / v52..v55
DEFK(k_fma,        "v_fma_f32 v40, v48, v44, v40\n v_fma_f32 v41, v49, v45, v41\n v_fma_f32 v42, v50, v46, v42\n v_fma_f32 v43, v51, v47, v43\n v_fma_f32 v52, v48, v44, v52\n v_fma_f32 v53, v49, v45, v53\n v_fma_f32 v54, v50, v46, v54\n v_fma_f32 v55, v51, v47, v55")
DEFK(k_mix_lo,     "v_fma_mix_f32 v40, v48, v44, v40 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, v49, v45, v41 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, v50, v46, v42 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v43, v51, v47, v43 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v52, v48, v44, v52 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v53, v49, v45, v53 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v54, v50, v46, v54 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v55, v51, v47, v55 op_sel_hi:[1,0,0]")
DEFK(k_mix_hi,     "v_fma_mix_f32 v40, v48, v44, v40 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, v49, v45, v41 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, v50, v46, v42 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v43, v51, v47, v43 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v52, v48, v44, v52 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v53, v49, v45, v53 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v54, v50, v46, v54 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v55, v51, v47, v55 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEFK(k_mix_f32,    "v_fma_mix_f32 v40, v52, v44, v40\n v_fma_mix_f32 v41, v53, v45, v41\n v_fma_mix_f32 v42, v54, v46, v42\n v_fma_mix_f32 v43, v55, v47, v43\n v_fma_mix_f32 v40, v52, v44, v40\n v_fma_mix_f32 v41, v53, v45, v41\n v_fma_mix_f32 v42, v54, v46, v42\n v_fma_mix_f32 v43, v55, v47, v43")
DEFK(k_mix_neg,    "v_fma_mix_f32 v40, v48, v44, -v52 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, v49, v45, -v53 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, v50, v46, -v54 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v43, v51, v47, -v55 op_sel_hi:[1,0,0]\n v_fma_mix_f32 v40, v48, v44, -v52 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v41, v49, v45, -v53 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v42, v50, v46, -v54 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 v43, v51, v47, -v55 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEFK(k_cvt_f16,    "v_cvt_f32_f16 v40, v48\n v_cvt_f32_f16 v41, v49\n v_cvt_f32_f16 v42, v50\n v_cvt_f32_f16 v43, v51\n v_cvt_f32_f16 v52, v48\n v_cvt_f32_f16 v53, v49\n v_cvt_f32_f16 v54, v50\n v_cvt_f32_f16 v55, v51")
DEFK(k_cvt_ubyte,  "v_cvt_f32_ubyte0 v40, v48\n v_cvt_f32_ubyte1 v41, v49\n v_cvt_f32_ubyte2 v42, v50\n v_cvt_f32_ubyte3 v43, v51\n v_cvt_f32_ubyte0 v52, v48\n v_cvt_f32_ubyte1 v53, v49\n v_cvt_f32_ubyte2 v54, v50\n v_cvt_f32_ubyte3 v55, v51")
DEFK(k_cvt_u32,    "v_cvt_f32_u32 v40, v48\n v_cvt_f32_u32 v41, v49\n v_cvt_f32_u32 v42, v50\n v_cvt_f32_u32 v43, v51\n v_cvt_f32_u32 v52, v48\n v_cvt_f32_u32 v53, v49\n v_cvt_f32_u32 v54, v50\n v_cvt_f32_u32 v55, v51")
DEFK(k_max3,       "v_max3_f32 v40, v48, v44, v40\n v_max3_f32 v41, v49, v45, v41\n v_max3_f32 v42, v50, v46, v42\n v_max3_f32 v43, v51, v47, v43\n v_max3_f32 v52, v48, v44, v52\n v_max3_f32 v53, v49, v45, v53\n v_max3_f32 v54, v50, v46, v54\n v_max3_f32 v55, v51, v47, v55")
DEFK(k_pk_max_f16, "v_pk_max_f16 v40, v48, v44\n v_pk_max_f16 v41, v49, v45\n v_pk_max_f16 v42, v50, v46\n v_pk_max_f16 v43, v51, v47\n v_pk_max_f16 v52, v48, v44\n v_pk_max_f16 v53, v49, v45\n v_pk_max_f16 v54, v50, v46\n v_pk_max_f16 v55, v51, v47")
DEFK(k_pk_fma_f16, "v_pk_fma_f16 v40, v48, v44, v40\n v_pk_fma_f16 v41, v49, v45, v41\n v_pk_fma_f16 v42, v50, v46, v42\n v_pk_fma_f16 v43, v51, v47, v43\n v_pk_fma_f16 v52, v48, v44, v52\n v_pk_fma_f16 v53, v49, v45, v53\n v_pk_fma_f16 v54, v50, v46, v54\n v_pk_fma_f16 v55, v51, v47, v55")
template <class K> double run(K kern, int blocks, int iters, float* d) { hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,10,1.0001f,0.5f); (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
 hipLaunchKernelGGL(kern,dim3(blocks),dim3(256),0,0,d,iters,1.0001f,0.5f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); return ms; }
#define LINE(NAME) { printf("%-22s", #NAME); for (int w=1; w<=4; w*=2) { double ms=run(NAME<0>,256*w,iters,d); printf("  w=%d %5.2f", w, ms*1e-3*2.4e9/((double)iters*64*w)); } printf("\n"); }
int main(){ float* d; (void)hipMalloc(&d,256*256*8*sizeof(float)); const int iters=20000;
 printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal; w = waves per SIMD\n");
 LINE(k_fma) LINE(k_mix_lo) LINE(k_mix_hi) LINE(k_mix_f32) LINE(k_mix_neg) LINE(k_cvt_f16) LINE(k_cvt_ubyte) LINE(k_cvt_u32) LINE(k_max3) LINE(k_pk_max_f16) LINE(k_pk_fma_f16)
 return 0; }
