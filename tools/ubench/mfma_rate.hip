// Issue-rate probe for v_mfma_f32_32x32x16_f16: dependent chain on one accumulator vs two
// interleaved accumulators, 1 / 2 / 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template<int ACCS>
__global__ __launch_bounds__(256) void probe(float *out,int iters)
{
  half8 a,b;
  for (int i=0; i < 8; i++) { a[i]=(_Float16) (threadIdx.x*0.001f+i); b[i]=(_Float16) (0.5f+i*0.01f); }
  floatx16 acc[ACCS];
  for (int k=0; k < ACCS; k++) for (int r=0; r < 16; r++) acc[k][r]=0.0f;
  for (int it=0; it < iters; it++)
    {
#pragma unroll
      for (int rep=0; rep < 8; rep++)
#pragma unroll
        for (int k=0; k < ACCS; k++)
          acc[k]=__builtin_amdgcn_mfma_f32_32x32x16_f16(a,b,acc[k],0,0,0);
    }
  float s=0;
  for (int k=0; k < ACCS; k++) for (int r=0; r < 16; r++) s+=acc[k][r];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

template<int ACCS> void run(int blocks,int threads,const char *what)
{
  float *out; hipMalloc(&out,sizeof(float)*blocks*threads);
  const int iters=20000;
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<ACCS>,dim3(blocks),dim3(threads),0,0,out,100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<ACCS>,dim3(blocks),dim3(threads),0,0,out,iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1);
  double mfma_per_wave=(double) iters*8*ACCS;
  double ns_per=ms*1e6/mfma_per_wave;
  printf("%-40s %8.3f ms  %.1f ns per MFMA per wave (%.0f cycles @2.4GHz)\n",what,ms,ns_per,ns_per*2.4);
  hipFree(out);
}

int main()
{
  run<1>(256,256,"1 acc, 1 wave/SIMD (256 blk x 4 waves)");
  run<2>(256,256,"2 accs, 1 wave/SIMD");
  run<4>(256,256,"4 accs, 1 wave/SIMD");
  run<1>(512,256,"1 acc, 2 waves/SIMD");
  run<2>(512,256,"2 accs, 2 waves/SIMD");
  run<1>(1024,256,"1 acc, 4 waves/SIMD");
  return 0;
}
