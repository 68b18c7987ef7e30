// Sustained v_fma_f64 rate of the whole chip: 32 independent accumulators per lane, two / four waves a SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fma_f64_rate tools/ubench/fma_f64_rate.hip && /tmp/fma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void fma64(double *out,double a,double b,int iterations)
{
  double acc[32];
#pragma unroll
  for (int i=0; i < 32; i++)
    acc[i]=(double) threadIdx.x+i;
  for (int it=0; it < iterations; it++)
    {
#pragma unroll
      for (int i=0; i < 32; i++)
        acc[i]=__builtin_fma(acc[i],a,b);
    }
  double s=0.0;
#pragma unroll
  for (int i=0; i < 32; i++)
    s+=acc[i];
  out[blockIdx.x*256+threadIdx.x]=s;
}
int main()
{
  double *out;
  hipMalloc(&out,sizeof(double)*256*8192);
  hipEvent_t e0,e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {512,1024,2048,4096})
    {
      const int iterations=20000;
      hipLaunchKernelGGL(fma64,dim3(blocks),dim3(256),0,0,out,1.0000001,1.0e-9,iterations);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(fma64,dim3(blocks),dim3(256),0,0,out,1.0000001,1.0e-9,iterations);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms=0; hipEventElapsedTime(&ms,e0,e1);
      const double fma=(double) blocks*256*32*iterations;
      printf("blocks %5d (%.1f waves a SIMD): %.3f ms  %.1f T fma/s = %.1f TFLOP/s\n",blocks,blocks*4/1024.0,ms,fma/ms/1e9,2*fma/ms/1e9);
    }
  return 0;
}
