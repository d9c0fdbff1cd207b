// instruction issue-rate microbenchmark (gfx950): cycles per wave-instruction on one SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s\n",hipGetErrorString(e)); return 1;}}while(0)
template<int OP> __global__ void k(uint32_t* out, uint32_t a0, uint32_t b0, int iters) {
  uint32_t x0=a0+threadIdx.x, x1=x0*3+1, x2=x0*5+2, x3=x0*7+3, x4=x0+4,x5=x0+5,x6=x0+6,x7=x0+7;
  uint64_t y0=x0,y1=x1,y2=x2,y3=x3,y4=x4,y5=x5,y6=x6,y7=x7;
  uint32_t b=b0|1;
  for (int i=0;i<iters;i++){
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
    if (OP==0) {
#define S(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==1) {
#define S(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==2) {
#define S(n) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==3) {
#define S(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y##n) : "v"(x##n), "v"(b) : "vcc");
      R8(S) R8(S)
#undef S
    } else if (OP==4) {
#define S(n) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==5) {
#define S(n) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==6) {
#define S(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==7) {
#define S(n) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==8) {
#define S(n) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x##n));
      R8(S) R8(S)
#undef S
    } else if (OP==9) {
#define S(n) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    } else if (OP==10) {
#define S(n) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x##n) : "v"(b));
      R8(S) R8(S)
#undef S
    }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=x0+x1+x2+x3+x4+x5+x6+x7+(uint32_t)(y0+y1+y2+y3+y4+y5+y6+y7);
}
template<int OP> double run(const char* name, uint32_t* d, int waves_per_simd) {
  int iters=20000; hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  int blocks=256*4*waves_per_simd/4; // 256 threads per block = 4 waves
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256),0,0,d,1u,3u,100);
  hipDeviceSynchronize();
  hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256),0,0,d,1u,3u,iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms,a,b);
  // per SIMD: waves_per_simd waves, each iters*16 instrs
  double instr_per_simd=(double)waves_per_simd*iters*16;
  double cyc=ms*1e-3*2.4e9/instr_per_simd;
  printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction (at 2.4 GHz)\n",name,waves_per_simd,ms,cyc);
  return cyc;
}
int main(){ uint32_t* d; CHK(hipMalloc(&d, 256*4*8*64*4*4));
  for (int w : {1,4}) {
    run<0>("v_add_u32",d,w); run<1>("v_mul_lo_u32",d,w); run<2>("v_mul_hi_u32",d,w); run<3>("v_mad_u64_u32",d,w);
    run<4>("v_mul_u32_u24",d,w); run<5>("v_mul_hi_u32_u24",d,w); run<6>("v_mad_u32_u24",d,w); run<7>("v_alignbit_b32",d,w);
    run<8>("v_bfe_u32",d,w); run<9>("v_lshl_or_b32",d,w); run<10>("v_dot4_u32_u8",d,w);
  }
  return 0; }
