// Issue interval (shader-clock cycles per wave-instruction, ONE wave on its SIMD, sixteen independent
// destinations) of the fp64-class and integer instructions the exact blur's epilogues are made of
// (blur_exact_common.hpp: exact_sums, exact_levels).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_misc_rate tools/ubench/f64_misc_rate.hip && /tmp/f64_misc_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER=2048;
constexpr int NOPS=20;

#define REPEAT16(OP) \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template<int MODE>
__global__ __launch_bounds__(64) void rate(double *out,unsigned long long *cycles,double seed)
{
  double d[16],e[16];
  unsigned u[16];
  float f[16];
#pragma unroll
  for (int i=0; i < 16; i++)
    {
      d[i]=seed+(double) threadIdx.x+i;
      e[i]=seed*0.5+i;
      u[i]=threadIdx.x*7u+i;
      f[i]=(float) i+0.25f;
    }
  const unsigned long long c0=__builtin_readcyclecounter();
  for (int it=0; it < ITER; it++)
    {
#define OP_FMA(i)    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e[i]), "v"(e[(i+1) & 15]));
#define OP_ADD(i)    asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e[i]));
#define OP_MUL(i)    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(d[i]) : "v"(e[i]), "v"(e[(i+1) & 15]));
#define OP_CVTI(i)   asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define OP_CVTU(i)   asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(e[i]));
#define OP_FRACT(i)  asm volatile("v_fract_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define OP_RCP(i)    asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define OP_CMP(i)    asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
#define OP_CMPABS(i) asm volatile("v_cmp_gt_f64 vcc, |%0|, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
#define OP_CVT32(i)  asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(e[i]));
#define OP_LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %1, 8, %0" : "+v"(d[i]) : "v"(e[i]));
#define OP_ADD32(i)  asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(u[(i+1) & 15]));
#define OP_LSHLADD32(i) asm volatile("v_lshl_add_u32 %0, %1, 8, %0" : "+v"(u[i]) : "v"(u[(i+1) & 15]));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %1, %2, 24" : "=v"(u[i]) : "v"(u[(i+1) & 15]), "v"(u[(i+2) & 15]));
#define OP_CMP32(i)  asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i+1) & 15]) : "vcc");
#define OP_FMA32(i)  asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(f[(i+1) & 15]), "v"(f[(i+2) & 15]));
#define OP_CVTF64U(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define OP_CVTI32F64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(e[i]));
#define OP_FLOOR(i)  asm volatile("v_floor_f64 %0, %1" : "=v"(d[i]) : "v"(e[i]));
#define OP_MAX(i)    asm volatile("v_max_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e[i]));
      if constexpr (MODE == 0) { REPEAT16(OP_FMA) }
      else if constexpr (MODE == 1) { REPEAT16(OP_ADD) }
      else if constexpr (MODE == 2) { REPEAT16(OP_MUL) }
      else if constexpr (MODE == 3) { REPEAT16(OP_CVTI) }
      else if constexpr (MODE == 4) { REPEAT16(OP_CVTU) }
      else if constexpr (MODE == 5) { REPEAT16(OP_FRACT) }
      else if constexpr (MODE == 6) { REPEAT16(OP_RCP) }
      else if constexpr (MODE == 7) { REPEAT16(OP_CMP) }
      else if constexpr (MODE == 8) { REPEAT16(OP_CMPABS) }
      else if constexpr (MODE == 9) { REPEAT16(OP_CVT32) }
      else if constexpr (MODE == 10) { REPEAT16(OP_LSHLADD64) }
      else if constexpr (MODE == 11) { REPEAT16(OP_ADD32) }
      else if constexpr (MODE == 12) { REPEAT16(OP_LSHLADD32) }
      else if constexpr (MODE == 13) { REPEAT16(OP_ALIGNBIT) }
      else if constexpr (MODE == 14) { REPEAT16(OP_CMP32) }
      else if constexpr (MODE == 15) { REPEAT16(OP_FMA32) }
      else if constexpr (MODE == 16) { REPEAT16(OP_CVTF64U) }
      else if constexpr (MODE == 17) { REPEAT16(OP_CVTI32F64) }
      else if constexpr (MODE == 18) { REPEAT16(OP_FLOOR) }
      else if constexpr (MODE == 19) { REPEAT16(OP_MAX) }
    }
  const unsigned long long c1=__builtin_readcyclecounter();
  double s=0.0;
#pragma unroll
  for (int i=0; i < 16; i++)
    s+=d[i]+(double) u[i]+(double) f[i];
  out[threadIdx.x]=s;
  if (threadIdx.x == 0)
    cycles[MODE]=c1-c0;
}

template<int MODE>
static void run(double *out,unsigned long long *cycles)
{
  hipLaunchKernelGGL(rate<MODE>,dim3(1),dim3(64),0,0,out,cycles,1.25);
  hipLaunchKernelGGL(rate<MODE>,dim3(1),dim3(64),0,0,out,cycles,1.25);
}

int main()
{
  double *out;
  unsigned long long *cycles,host[NOPS];
  hipMalloc(&out,sizeof(double)*64);
  hipMalloc(&cycles,sizeof(host));
  hipMemset(cycles,0,sizeof(host));
  run<0>(out,cycles); run<1>(out,cycles); run<2>(out,cycles); run<3>(out,cycles); run<4>(out,cycles);
  run<5>(out,cycles); run<6>(out,cycles); run<7>(out,cycles); run<8>(out,cycles); run<9>(out,cycles);
  run<10>(out,cycles); run<11>(out,cycles); run<12>(out,cycles); run<13>(out,cycles); run<14>(out,cycles);
  run<15>(out,cycles); run<16>(out,cycles); run<17>(out,cycles); run<18>(out,cycles); run<19>(out,cycles);
  hipDeviceSynchronize();
  hipMemcpy(host,cycles,sizeof(host),hipMemcpyDeviceToHost);
  const char *name[NOPS]={"v_fma_f64","v_add_f64","v_mul_f64","v_cvt_f64_i32","v_cvt_u32_f64","v_fract_f64","v_rcp_f64",
    "v_cmp_gt_f64","v_cmp_gt_f64 |x|","v_cvt_f32_f64","v_lshl_add_u64","v_add_u32","v_lshl_add_u32","v_alignbit_b32",
    "v_cmp_gt_u32","v_fma_f32","v_cvt_f64_u32","v_cvt_i32_f64","v_floor_f64","v_max_f64"};
  // the cycle counter's unit against the shader clock: v_fma_f32 issues every 4 shader cycles from one wave
  const double unit=(double) host[15]/(16.0*ITER);
  for (int i=0; i < NOPS; i++)
    printf("%-18s %7.2f counter ticks an instruction = %5.2f x v_fma_f32\n",name[i],(double) host[i]/(16.0*ITER),
      (double) host[i]/(16.0*ITER)/unit);
  return 0;
}
