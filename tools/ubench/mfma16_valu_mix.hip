// How much VALU work hides behind v_mfma_f32_16x16x32_f16 on one SIMD?
//   W waves per SIMD (1, 2 or 4), every wave runs the same stream: a dependent MFMA chain with
//   V independent v_fma_f32 interleaved after each MFMA (V = 0 .. 12), issued as scalar
//   v_fma_f32 (inline asm; PACKED = false) or left to the compiler, which pairs them into
//   v_pk_fma_f32 (PACKED = true).  The first version of this probe only had the second form and
//   read its result as "MFMA and VALU do not overlap": what does not overlap is PACKED f32.
// Prints cycles per MFMA per SIMD (s_memtime of wave 0).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template<int V,bool PACKED>
__global__ __launch_bounds__(1024) void probe(float *out,long long *cycles,int iters)
{
  half8 a,b;
  for (int i=0; i < 8; i++) { a[i]=(_Float16) (threadIdx.x*0.001f+i); b[i]=(_Float16) (0.5f+i*0.01f); }
  floatx4 acc={0.0f,0.0f,0.0f,0.0f};
  float x[12];
  for (int i=0; i < 12; i++) x[i]=threadIdx.x*0.5f+i;
  const float m=1.0001f,c=0.5f;
  __syncthreads();
  const long long t0=__builtin_readcyclecounter();
  for (int it=0; it < iters; it++)
#pragma unroll
    for (int rep=0; rep < 8; rep++)
      {
        acc=__builtin_amdgcn_mfma_f32_16x16x32_f16(a,b,acc,0,0,0);
#pragma unroll
        for (int v=0; v < V; v++)
          if (PACKED)
            x[v]=__builtin_fmaf(x[v],m,c);
          else
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v]) : "v"(m),"v"(c));
      }
  const long long t1=__builtin_readcyclecounter();
  float s=acc[0]+acc[1]+acc[2]+acc[3];
  for (int i=0; i < 12; i++) s+=x[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
  if ((threadIdx.x == 0) && (blockIdx.x == 0))
    cycles[0]=t1-t0;
}

template<int V,bool PACKED>
static void run(int waves_per_simd)
{
  float *out; long long *cycles;
  hipMalloc(&out,sizeof(float)*256*1024);
  hipMalloc(&cycles,sizeof(long long));
  const int iters=2000;
  hipLaunchKernelGGL((probe<V,PACKED>),dim3(256),dim3(256*waves_per_simd),0,0,out,cycles,100);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<V,PACKED>),dim3(256),dim3(256*waves_per_simd),0,0,out,cycles,iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  long long host=0;
  hipMemcpy(&host,cycles,sizeof(host),hipMemcpyDeviceToHost);
  const double mfmas=(double) iters*8*waves_per_simd;
  printf("%s waves/SIMD %d  VALU per MFMA %2d : %.3f ms, %.1f shader cycles per MFMA per SIMD (%.1f per wave-MFMA)\n",
    PACKED ? "compiler (v_pk_fma_f32)" : "scalar v_fma_f32       ",waves_per_simd,V,ms,(double) host/mfmas,(double) host/(iters*8.0));
  hipFree(out); hipFree(cycles);
}

int main()
{
  for (int w=1; w <= 4; w*=2)
    {
      run<0,false>(w); run<2,false>(w); run<4,false>(w); run<6,false>(w); run<8,false>(w); run<12,false>(w);
      run<4,true>(w); run<8,true>(w); run<12,true>(w);
    }
  return 0;
}
