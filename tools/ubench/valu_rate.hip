// Micro-benchmark: sustained issue rate of the f32 FMA forms the blur kernels can use.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e=(x); if (e != hipSuccess) { printf("%s: %s\n",#x,hipGetErrorString(e)); return 1; } } while (0)

constexpr int NACC=32;
constexpr int ITER=4096;

template<int MODE>
__global__ __launch_bounds__(256) void rate(float *out,const float *taps,unsigned long long *cycles)
{
  float acc[NACC];
#pragma unroll
  for (int i=0; i < NACC; i++) acc[i]=(float) threadIdx.x*0.001f+i;
  float s=out[threadIdx.x & 7];
  float t0=taps[0],t1=taps[1];          // uniform -> SGPR
  unsigned long long c0=__builtin_readcyclecounter();
  for (int it=0; it < ITER; it++)
    {
      if constexpr (MODE == 0)          // v_fma_f32, all-VGPR operands
        {
#pragma unroll
          for (int i=0; i < NACC; i++)
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(s), "v"(acc[(i+1)%NACC]));
        }
      else if constexpr (MODE == 1)     // v_fmac_f32 with an SGPR multiplier
        {
#pragma unroll
          for (int i=0; i < NACC; i++)
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(t0), "v"(s));
        }
      else if constexpr (MODE == 2)     // v_pk_fma_f32 all-VGPR
        {
#pragma unroll
          for (int i=0; i < NACC; i+=2)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(float2 *) &acc[i]) : "v"(*(float2 *) &acc[(i+2)%NACC]), "v"(*(float2 *) &acc[(i+4)%NACC]));
        }
      else if constexpr (MODE == 3)     // v_pk_fma_f32 with an SGPR pair broadcast (as hipcc emits)
        {
          float2 tt=make_float2(t0,t1);
#pragma unroll
          for (int i=0; i < NACC; i+=2)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(*(float2 *) &acc[i]) : "s"(tt), "v"(*(float2 *) &acc[(i+2)%NACC]));
        }
      else if constexpr (MODE == 4)     // v_fma_f64
        {
#pragma unroll
          for (int i=0; i < NACC; i+=2)
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double *) &acc[i]) : "v"(*(double *) &acc[(i+2)%NACC]), "v"(*(double *) &acc[(i+4)%NACC]));
        }
      else if constexpr (MODE == 6)     // v_fmac_f64 with an SGPR-pair multiplier (the EXACT kernels' tap form)
        {
          const double td=(double) t0;
#pragma unroll
          for (int i=0; i < NACC; i+=2)
            asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(*(double *) &acc[i]) : "s"(td), "v"(*(double *) &acc[(i+2)%NACC]));
        }
      else if constexpr (MODE == 7)     // v_cvt_f64_u32
        {
#pragma unroll
          for (int i=0; i < NACC; i+=2)
            asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(*(double *) &acc[i]) : "v"(acc[(i+3)%NACC]));
        }
      else if constexpr (MODE == 5)     // v_mul_f64 + v_add_f64 pairs (EXACT path)
        {
#pragma unroll
          for (int i=0; i < NACC; i+=4)
            {
              asm volatile("v_mul_f64 %0, %1, %2" : "=v"(*(double *) &acc[i]) : "v"(*(double *) &acc[(i+4)%NACC]), "v"(*(double *) &acc[(i+8)%NACC]));
              asm volatile("v_add_f64 %0, %1, %0" : "+v"(*(double *) &acc[i+2]) : "v"(*(double *) &acc[i]));
            }
        }
    }
  unsigned long long c1=__builtin_readcyclecounter();
  float r=0;
#pragma unroll
  for (int i=0; i < NACC; i++) r+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=r;
  if ((threadIdx.x == 0) && (blockIdx.x == 0)) cycles[MODE]=c1-c0;
}

// The blur inner block with compiler-allocated registers, no memory traffic:
// TAPS_SGPR=false keeps the taps in VGPRs (loaded from a per-lane address),
// true in SGPRs (uniform address -> s_load).
template<bool TAPS_SGPR>
__global__ __launch_bounds__(256) void block_rate(float *out,const float *taps,int iters)
{
  constexpr int R=16,U=8,C=4;
  float acc[R][C];
#pragma unroll
  for (int r=0; r < R; r++)
#pragma unroll
    for (int c=0; c < C; c++) acc[r][c]=0.f;
  float in[C];
#pragma unroll
  for (int c=0; c < C; c++) in[c]=out[threadIdx.x+c];
  const float *tp=TAPS_SGPR ? taps : taps+(threadIdx.x >> 8);
  for (int it=0; it < iters; it++)
    {
      float tw[R+U-1];
#pragma unroll
      for (int i=0; i < R+U-1; i++) tw[i]=tp[(it & 7)*U+i];
#pragma unroll
      for (int jj=0; jj < U; jj++)
        {
#pragma unroll
          for (int c=0; c < C; c++) in[c]=in[c]*1.0001f;
#pragma unroll
          for (int r=0; r < R; r++)
#pragma unroll
            for (int c=0; c < C; c++)
              acc[r][c]=__builtin_fmaf(tw[jj-r+R-1],in[c],acc[r][c]);
        }
    }
  float rsum=0;
#pragma unroll
  for (int r=0; r < R; r++)
#pragma unroll
    for (int c=0; c < C; c++) rsum+=acc[r][c];
  out[blockIdx.x*blockDim.x+threadIdx.x]=rsum;
}

template<bool TAPS_SGPR> int run_block(const char *name,int blocks_per_cu)
{
  float *out,*taps;
  int grid=256*blocks_per_cu,iters=512;
  CHECK(hipMalloc(&out,(size_t) grid*256*4+64)); CHECK(hipMalloc(&taps,4096));
  CHECK(hipMemset(out,0,(size_t) grid*256*4+64)); CHECK(hipMemset(taps,0,4096));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  block_rate<TAPS_SGPR><<<grid,256>>>(out,taps,iters);
  CHECK(hipDeviceSynchronize());
  hipEventRecord(a);
  for (int i=0; i < 5; i++) block_rate<TAPS_SGPR><<<grid,256>>>(out,taps,iters);
  hipEventRecord(b); CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,a,b); ms/=5;
  double fma=(double) grid*256*iters*8*16*4;
  double per_simd=(double) blocks_per_cu*iters*512;
  printf("%-34s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  => if 2.4GHz: %.2f cyc/fma-instr/SIMD\n",name,blocks_per_cu,ms,
    fma*2/(ms*1e-3)/1e12,ms*1e-3*2.4e9/per_simd);
  hipFree(out); hipFree(taps);
  return 0;
}

template<int MODE> int run(const char *name,double flop_per_instr,int instr_per_iter,int blocks_per_cu)
{
  float *out,*taps; unsigned long long *cyc;
  int grid=256*blocks_per_cu;
  CHECK(hipMalloc(&out,(size_t) grid*256*4)); CHECK(hipMalloc(&taps,64)); CHECK(hipMalloc(&cyc,64));
  CHECK(hipMemset(out,0,(size_t) grid*256*4)); CHECK(hipMemset(taps,0,64));
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  rate<MODE><<<grid,256>>>(out,taps,cyc);
  CHECK(hipDeviceSynchronize());
  hipEventRecord(a);
  for (int i=0; i < 5; i++) rate<MODE><<<grid,256>>>(out,taps,cyc);
  hipEventRecord(b); CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms,a,b); ms/=5;
  unsigned long long h[8]; CHECK(hipMemcpy(h,cyc,64,hipMemcpyDeviceToHost));
  double instr=(double) grid*4/*waves*/*ITER*instr_per_iter;
  double tflops=instr*64*flop_per_instr/(ms*1e-3)/1e12;
  // per-SIMD wave-instructions: waves per SIMD = blocks_per_cu (4 waves/block over 4 SIMDs)
  double per_simd=(double) blocks_per_cu*ITER*instr_per_iter;
  printf("%-34s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  clock-counter cycles=%llu (%.2f per wave-instr of one wave)  => if 2.4GHz: %.2f cyc/instr/SIMD\n",
    name,blocks_per_cu,ms,tflops,h[MODE],(double) h[MODE]/(ITER*instr_per_iter),ms*1e-3*2.4e9/per_simd);
  hipFree(out); hipFree(taps); hipFree(cyc);
  return 0;
}

int main()
{
  for (int bpc : {1,2,4})
    {
      run<0>("v_fma_f32 vgpr",2,NACC,bpc);
      run<1>("v_fmac_f32 sgpr",2,NACC,bpc);
      run<2>("v_pk_fma_f32 vgpr",4,NACC/2,bpc);
      run<3>("v_pk_fma_f32 sgpr-bcast",4,NACC/2,bpc);
      run<4>("v_fma_f64",2,NACC/2,bpc);
      run<5>("v_mul_f64+v_add_f64",1,NACC/2,bpc);
      run<6>("v_fmac_f64 sgpr",2,NACC/2,bpc);
      run<7>("v_cvt_f64_u32 (1 op/instr)",1,NACC/2,bpc);
      run_block<false>("blur block, taps in VGPRs",bpc);
      run_block<true>("blur block, taps in SGPRs",bpc);
    }
  return 0;
}
