// Do the matrix pipe and the vector ALU of one SIMD overlap across waves?  Two waves per SIMD:
// both MFMA, both VALU, or one of each.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// mode bit0: even waves run MFMA (else VALU); bit1: odd waves run MFMA (else VALU)
__global__ __launch_bounds__(512) void probe(float *out,int iters,int mode)
{
  const int wave=threadIdx.x >> 6;
  // waves 0-3 land on SIMD 0-3, waves 4-7 on SIMD 0-3 again: partner = wave^4
  const bool second=wave >= 4;
  const bool mfma=second ? (mode & 2) != 0 : (mode & 1) != 0;
  float s=0.0f;
  if (mfma)
    {
      half8 a,b;
      for (int i=0; i < 8; i++) { a[i]=(_Float16) (threadIdx.x*0.001f+i); b[i]=(_Float16) (0.5f+i*0.01f); }
      floatx16 acc;
      for (int r=0; r < 16; r++) acc[r]=0.0f;
      for (int it=0; it < iters; it++)
#pragma unroll
        for (int rep=0; rep < 8; rep++)
          acc=__builtin_amdgcn_mfma_f32_32x32x16_f16(a,b,acc,0,0,0);
      for (int r=0; r < 16; r++) s+=acc[r];
    }
  else
    {
      float x0=threadIdx.x*0.5f,x1=1.0f,x2=2.0f,x3=3.0f,x4=4.0f,x5=5.0f,x6=6.0f,x7=7.0f;
      const float m=1.0001f,c=0.5f;
      for (int it=0; it < iters; it++)
#pragma unroll
        for (int rep=0; rep < 10; rep++)          // 80 independent-ish FMAs ~ 8 MFMAs of issue time
          {
            x0=__builtin_fmaf(x0,m,c); x1=__builtin_fmaf(x1,m,c); x2=__builtin_fmaf(x2,m,c); x3=__builtin_fmaf(x3,m,c);
            x4=__builtin_fmaf(x4,m,c); x5=__builtin_fmaf(x5,m,c); x6=__builtin_fmaf(x6,m,c); x7=__builtin_fmaf(x7,m,c);
          }
      s=x0+x1+x2+x3+x4+x5+x6+x7;
    }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

static float run(int mode,int iters)
{
  float *out; hipMalloc(&out,sizeof(float)*256*512);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe,dim3(256),dim3(512),0,0,out,100,mode);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe,dim3(256),dim3(512),0,0,out,iters,mode);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  hipFree(out);
  return ms;
}

int main()
{
  const int iters=20000;
  printf("both waves VALU   : %.3f ms\n",run(0,iters));
  printf("both waves MFMA   : %.3f ms\n",run(3,iters));
  printf("one MFMA, one VALU: %.3f ms\n",run(1,iters));
  printf("one VALU, one MFMA: %.3f ms\n",run(2,iters));
  return 0;
}
