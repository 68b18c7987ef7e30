// v_mfma_f64_16x16x4_f64: sustained rate alone and beside fp64 vector work of another wave /
// of the same wave (what resize_mfma.hip relies on: matrix pipe busy while the vector pipe
// finishes pixels).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/mfma_f64_rate tools/ubench/mfma_f64_rate.hip && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// MODE 0: 8 independent MFMA chains; 1: the same + 8 independent v_fma_f64 per MFMA in the same
// wave; 2: even waves MFMA only, odd waves v_fma_f64 only
template<int MODE>
__global__ __launch_bounds__(256) void kernel(double *out,double a,double b,int iterations)
{
  d4 acc[8];
  double v[8];
#pragma unroll
  for (int i=0; i < 8; i++)
    {
      acc[i]=(d4) {(double) threadIdx.x,1.0,2.0,3.0+i};
      v[i]=(double) threadIdx.x+i;
    }
  const bool odd=((threadIdx.x >> 6) & 1) != 0;
  for (int it=0; it < iterations; it++)
    {
      if ((MODE != 2) || !odd)
        {
#pragma unroll
          for (int i=0; i < 8; i++)
            {
              acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
              if (MODE == 1)
                {
#pragma unroll
                  for (int j=0; j < 8; j++)
                    v[j]=__builtin_fma(v[j],a,b);
                }
            }
        }
      else
        {
#pragma unroll
          for (int i=0; i < 8; i++)
#pragma unroll
            for (int j=0; j < 8; j++)
              v[j]=__builtin_fma(v[j],a,b);
        }
    }
  double s=0.0;
#pragma unroll
  for (int i=0; i < 8; i++)
    s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3]+v[i];
  out[blockIdx.x*256+threadIdx.x]=s;
}
template<int MODE>
static void run(double *out,const char *what)
{
  hipEvent_t e0,e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256,512,1024})
    {
      const int iterations=4000;
      hipLaunchKernelGGL(kernel<MODE>,dim3(blocks),dim3(256),0,0,out,1.0000001,1.0e-9,iterations);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(kernel<MODE>,dim3(blocks),dim3(256),0,0,out,1.0000001,1.0e-9,iterations);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms=0; hipEventElapsedTime(&ms,e0,e1);
      const double waves=(double) blocks*4*(MODE == 2 ? 0.5 : 1.0);
      const double mfma=waves*8*iterations;
      const double cycles_per=ms*1e-3*2.4e9/((double) blocks*4/1024.0*(MODE == 2 ? 0.5 : 1.0)*8*iterations);
      printf("%-34s blocks %5d (%.0f waves a SIMD): %.3f ms  %.2f G mfma/s = %.1f TFLOP/s, %.1f cycles a SIMD per mfma at 2.4 GHz\n",
        what,blocks,blocks*4/1024.0,ms,mfma/ms/1e6,mfma*2048/ms/1e9,cycles_per);
    }
}
int main()
{
  double *out;
  hipMalloc(&out,sizeof(double)*256*4096);
  run<0>(out,"mfma only");
  run<1>(out,"mfma + 8 v_fma_f64 each, one wave");
  run<2>(out,"mfma waves beside v_fma_f64 waves");
  return 0;
}
