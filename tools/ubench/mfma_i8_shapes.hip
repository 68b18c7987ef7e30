// Issue rate of the i8 matrix instructions of gfx950, four waves per SIMD, five independent tiles:
//   v_mfma_i32_16x16x64_i8 (new), v_mfma_i32_16x16x32_i8 (legacy K), v_mfma_i32_32x32x32_i8
//   hipcc --offload-arch=gfx950 -O3 -o mfma_i8_shapes mfma_i8_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx16 __attribute__((ext_vector_type(16)));

template<int SHAPE>
__global__ __launch_bounds__(1024) void probe(int *out,int iters)
{
  intx4 a,b;
  for (int i=0; i < 4; i++) { a[i]=(int) threadIdx.x*0x01010101+i; b[i]=0x01020304*(i+1); }
  long a8=((long) a[0] << 32) | (unsigned) a[1],b8=((long) b[0] << 32) | (unsigned) b[1];
  intx4 acc[5];
  intx16 big[2];
  for (int i=0; i < 5; i++) acc[i]=intx4{0,0,0,0};
  for (int i=0; i < 2; i++) for (int j=0; j < 16; j++) big[i][j]=0;
  for (int it=0; it < iters; it++)
#pragma unroll
    for (int rep=0; rep < 10; rep++)
      {
        if (SHAPE == 0) acc[rep % 5]=__builtin_amdgcn_mfma_i32_16x16x64_i8(a,b,acc[rep % 5],0,0,0);
        if (SHAPE == 1) acc[rep % 5]=__builtin_amdgcn_mfma_i32_16x16x32_i8(a8,b8,acc[rep % 5],0,0,0);
        if (SHAPE == 2) big[rep % 2]=__builtin_amdgcn_mfma_i32_32x32x32_i8(a,b,big[rep % 2],0,0,0);
      }
  int s=0;
  for (int i=0; i < 5; i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for (int i=0; i < 2; i++) for (int j=0; j < 16; j++) s+=big[i][j];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

template<int SHAPE>
static void run(const char *name,double macs)
{
  int *out;
  hipMalloc(&out,sizeof(int)*256*1024);
  const int iters=4000;
  hipLaunchKernelGGL((probe<SHAPE>),dim3(256),dim3(1024),0,0,out,400);
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<SHAPE>),dim3(256),dim3(1024),0,0,out,iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms=0.0f; hipEventElapsedTime(&ms,e0,e1);
  const double per_simd=(double) iters*10*4;
  printf("%-26s %.3f ms: %.2f ns per instruction per SIMD, %.0f TOPS\n",name,ms,1.0e6*ms/per_simd,
    2.0*macs*per_simd*1024.0/(ms*1.0e-3)/1.0e12);
  hipFree(out);
}

int main()
{
  run<0>("v_mfma_i32_16x16x64_i8",16.0*16*64);
  run<1>("v_mfma_i32_16x16x32_i8",16.0*16*32);
  run<2>("v_mfma_i32_32x32x32_i8",32.0*32*32);
  return 0;
}
