// What does a VALU instruction cost beside v_mfma_i32_16x16x64_i8 on one SIMD, by kind?
//   4 waves per SIMD, every wave runs the same stream: MFMAs round-robin over five accumulator
//   tiles (the exact blur's chain) with V independent VALU instructions after each MFMA:
//   KIND 0 v_fma_f32, 1 v_fma_f64, 2 v_cvt_f64_i32, 3 v_lshl_add_u32, 4 v_perm_b32, 5 v_fract_f64,
//   6 v_mul_f64, 7 v_cvt_u32_f64
// Prints shader cycles per MFMA per SIMD (wave 0's clock).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_i8_valu_mix mfma_i8_valu_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int intx4 __attribute__((ext_vector_type(4)));

template<int V,int KIND,bool MFMA>
__global__ __launch_bounds__(1024) void probe(int *out,long long *cycles,int iters)
{
  intx4 a,b;
  for (int i=0; i < 4; i++) { a[i]=(int) threadIdx.x*0x01010101+i; b[i]=0x01020304*(i+1); }
  intx4 acc[5];
  for (int i=0; i < 5; i++) acc[i]=intx4{0,0,0,0};
  float x[8]; double d[8]; int n[8];
  for (int i=0; i < 8; i++) { x[i]=threadIdx.x*0.5f+i; d[i]=threadIdx.x*0.25+i; n[i]=(int) threadIdx.x+i; }
  const float m=1.0001f,c=0.5f;
  const double dm=1.0000001,dc=0.5;
  __syncthreads();
  const long long t0=__builtin_readcyclecounter();
  for (int it=0; it < iters; it++)
#pragma unroll
    for (int rep=0; rep < 10; rep++)
      {
        if (MFMA)
          acc[rep % 5]=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,acc[rep % 5],0,0,0);
#pragma unroll
        for (int v=0; v < V; v++)
          {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v % 8]) : "v"(m),"v"(c));
            if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[v % 8]) : "v"(dm),"v"(dc));
            if (KIND == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[v % 8]) : "v"(n[v % 8]));
            if (KIND == 3) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(n[v % 8]) : "v"(n[(v+1) % 8]));
            if (KIND == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(n[v % 8]) : "v"(n[(v+1) % 8]),"v"(0x05010400));
            if (KIND == 5) asm volatile("v_fract_f64 %0, %1" : "=v"(d[v % 8]) : "v"(d[(v+1) % 8]));
            if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[v % 8]) : "v"(dm));
            if (KIND == 7) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(n[v % 8]) : "v"(d[v % 8]));
          }
      }
  const long long t1=__builtin_readcyclecounter();
  int s=0;
  for (int i=0; i < 5; i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for (int i=0; i < 8; i++) s+=(int) x[i]+(int) d[i]+n[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if ((threadIdx.x == 0) && (blockIdx.x == 0))
    cycles[0]=t1-t0;
}

template<int V,int KIND,bool MFMA>
static void run(const char *name)
{
  int *out; long long *cycles;
  hipMalloc(&out,sizeof(int)*256*1024);
  hipMalloc(&cycles,sizeof(long long));
  const int iters=1000;
  hipLaunchKernelGGL((probe<V,KIND,MFMA>),dim3(256),dim3(1024),0,0,out,cycles,100);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V,KIND,MFMA>),dim3(256),dim3(1024),0,0,out,cycles,iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms=0.0f; hipEventElapsedTime(&ms,e0,e1);
  long long host=0;
  hipMemcpy(&host,cycles,sizeof(host),hipMemcpyDeviceToHost);
  const double slots=(double) iters*10*4;          // MFMA slots per SIMD
  printf("%-16s %s  VALU per MFMA slot %2d : %6.1f ticks per slot per SIMD, %.3f ms = %.2f ns per slot per SIMD (tick = %.2f ns)\n",
    name,MFMA ? "with MFMA" : "VALU only",V,(double) host/slots,ms,1.0e6*ms/slots,1.0e6*ms/(double) host);
  hipFree(out); hipFree(cycles);
}

#define KIND_RUNS(K,NAME) \
  run<1,K,true>(NAME); run<2,K,true>(NAME); run<3,K,true>(NAME); run<4,K,true>(NAME); run<6,K,true>(NAME); \
  run<4,K,false>(NAME);

int main()
{
  run<0,0,true>("bare chain");
  KIND_RUNS(0,"v_fma_f32")
  KIND_RUNS(1,"v_fma_f64")
  KIND_RUNS(2,"v_cvt_f64_i32")
  KIND_RUNS(3,"v_lshl_add_u32")
  KIND_RUNS(4,"v_perm_b32")
  KIND_RUNS(5,"v_fract_f64")
  KIND_RUNS(6,"v_mul_f64")
  KIND_RUNS(7,"v_cvt_u32_f64")
  return 0;
}
