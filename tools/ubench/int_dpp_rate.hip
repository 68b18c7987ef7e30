// Micro-benchmark: sustained issue rate of the instructions the morphology kernels lean on:
//   v_pk_max_u16, v_max_u32, v_mov_b32_dpp wave_shr:1 / row_shr:1, ds_bpermute_b32, and the
//   Row(1) step of morph_rects_kernel (2 DPP + 8 v_pk_max_u16 per 4 words).
//   hipcc --offload-arch=gfx950 -O3 -o int_dpp_rate int_dpp_rate.hip && ./int_dpp_rate
// Prints shader cycles per wave-instruction per SIMD with 4 waves on every SIMD (256 threads x
// 4 blocks per CU share a SIMD four ways), measured from the kernel time of 256*4 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int NACC=32;
constexpr int ITER=2048;

template<int MODE>
__global__ __launch_bounds__(256) void rate(unsigned *out)
{
  unsigned acc[NACC];
#pragma unroll
  for (int i=0; i < NACC; i++) acc[i]=threadIdx.x*2654435761u+i;
  for (int it=0; it < ITER; it++)
    {
#pragma unroll
      for (int i=0; i < NACC; i++)
        {
          if constexpr (MODE == 0)
            asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 1)
            asm volatile("v_max_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 2)
            asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 3)
            asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 4)
            asm volatile("v_max_u16_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 5)
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 6)
            asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
          else if constexpr (MODE == 7)
            asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(acc[(i+5)%NACC]));
        }
    }
  unsigned r=0;
#pragma unroll
  for (int i=0; i < NACC; i++) r+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}

// the kernel's Row(1) step on 8 rows x 4 words
template<bool DPP>
__global__ __launch_bounds__(256) void row1(unsigned *out)
{
  unsigned s[8][4];
#pragma unroll
  for (int i=0; i < 8; i++)
#pragma unroll
    for (int p=0; p < 4; p++) s[i][p]=threadIdx.x*2654435761u+i*4+p;
  for (int it=0; it < ITER; it++)
#pragma unroll
    for (int i=0; i < 8; i++)
      {
        unsigned n[4];
#pragma unroll
        for (int p=0; p < 4; p++)
          {
            typedef unsigned short U2 __attribute__((ext_vector_type(2)));
            unsigned left,right;
            if (p >= 2)
              left=s[i][p-2];
            else
              left=DPP ? (unsigned) __builtin_amdgcn_update_dpp((int) s[i][p+2],(int) s[i][p+2],0x138,0xf,0xf,false) :
                (unsigned) __shfl_up((int) s[i][p+2],1,64);
            if (p < 2)
              right=s[i][p+2];
            else
              right=DPP ? (unsigned) __builtin_amdgcn_update_dpp((int) s[i][p-2],(int) s[i][p-2],0x130,0xf,0xf,false) :
                (unsigned) __shfl_down((int) s[i][p-2],1,64);
            const U2 a=__builtin_bit_cast(U2,left),b=__builtin_bit_cast(U2,s[i][p]),c=__builtin_bit_cast(U2,right);
            n[p]=__builtin_bit_cast(unsigned,__builtin_elementwise_max(__builtin_elementwise_max(a,b),c));
          }
#pragma unroll
        for (int p=0; p < 4; p++) s[i][p]=n[p];
      }
  unsigned r=0;
#pragma unroll
  for (int i=0; i < 8; i++)
#pragma unroll
    for (int p=0; p < 4; p++) r+=s[i][p];
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
}

template<typename K>
static double time_kernel(K kernel,unsigned *out,int blocks)
{
  hipLaunchKernelGGL(kernel,dim3(blocks),dim3(256),0,0,out);
  hipEvent_t e0,e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
  (void) hipEventRecord(e0);
  hipLaunchKernelGGL(kernel,dim3(blocks),dim3(256),0,0,out);
  (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
  float ms=0; (void) hipEventElapsedTime(&ms,e0,e1);
  return ms;
}

int main()
{
  unsigned *out;
  (void) hipMalloc(&out,sizeof(unsigned)*256*256*8);
  const int blocks=256*4;                  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  const char *names[]={"v_pk_max_u16","v_max_u32","v_mov_b32_dpp wave_shr:1","v_mov_b32_dpp row_shr:1",
    "v_max_u16_dpp wave_shr:1","v_add_u32","v_pk_add_u16","v_mov_b32_dpp wave_shl:1"};
  double ms[8];
  ms[0]=time_kernel(rate<0>,out,blocks); ms[1]=time_kernel(rate<1>,out,blocks);
  ms[2]=time_kernel(rate<2>,out,blocks); ms[3]=time_kernel(rate<3>,out,blocks);
  ms[4]=time_kernel(rate<4>,out,blocks); ms[5]=time_kernel(rate<5>,out,blocks);
  ms[6]=time_kernel(rate<6>,out,blocks); ms[7]=time_kernel(rate<7>,out,blocks);
  for (int m=0; m < 8; m++)
    {
      // per SIMD: 4 waves x ITER x NACC instructions
      const double instr=4.0*ITER*NACC;
      printf("%-28s %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n",names[m],ms[m],
        ms[m]*1e-3*2.4e9/instr);
    }
  const double a=time_kernel(row1<true>,out,blocks),b=time_kernel(row1<false>,out,blocks);
  const double steps=4.0*ITER*8;           // Row(1) steps of one row (4 words) per SIMD
  printf("Row(1) of 4 words, DPP       %.3f ms  %.1f cycles per row step per SIMD\n",a,a*1e-3*2.4e9/steps);
  printf("Row(1) of 4 words, bpermute  %.3f ms  %.1f cycles per row step per SIMD\n",b,b*1e-3*2.4e9/steps);
  (void) hipFree(out);
  return 0;
}
