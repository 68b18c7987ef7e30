// Write-bandwidth of the store patterns the resize kernels can use (17.2 GB target:
// 32768 x 32768 float RGBA).   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e=(x); if (e != hipSuccess) { printf("%s: %s\n",#x,hipGetErrorString(e)); return 1; } } while (0)

// block = 256 threads; writes ROWS rows x (256*CHUNK) columns of 16-byte pixels
template<int ROWS,int CHUNK>
__global__ __launch_bounds__(256) void tile_store(float4 *dst,int columns,int rows)
{
  const int x0=(int) blockIdx.x*256*CHUNK,y0=(int) blockIdx.y*ROWS;
  float4 v=make_float4((float) threadIdx.x,1.f,2.f,3.f);
  for (int r=0; r < ROWS; r++)
#pragma unroll
    for (int c=0; c < CHUNK; c++)
      dst[(size_t) (y0+r)*columns+x0+c*256+threadIdx.x]=v;
}

__global__ __launch_bounds__(256) void linear_store(float4 *dst,size_t n)
{
  float4 v=make_float4((float) threadIdx.x,1.f,2.f,3.f);
  for (size_t i=(size_t) blockIdx.x*256+threadIdx.x; i < n; i+=(size_t) gridDim.x*256)
    dst[i]=v;
}

template<typename F> double timeit(F f)
{
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i=0; i < 3; i++) f(); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms,a,b); return ms/3;
}

int main()
{
  const int W=32768,H=32768;
  const size_t n=(size_t) W*H;
  float4 *dst; CHECK(hipMalloc(&dst,n*16));
  double gb=n*16/1e9;
  double ms=timeit([&]{ linear_store<<<256*16,256>>>(dst,n); });
  printf("linear grid-stride          %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<16,1><<<dim3(W/256,H/16),256>>>(dst,W,H); });
  printf("tile 16 rows x 4 KB         %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<32,1><<<dim3(W/256,H/32),256>>>(dst,W,H); });
  printf("tile 32 rows x 4 KB         %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<4,1><<<dim3(W/256,H/4),256>>>(dst,W,H); });
  printf("tile 4 rows x 4 KB          %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<1,1><<<dim3(W/256,H/1),256>>>(dst,W,H); });
  printf("tile 1 row x 4 KB           %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<4,4><<<dim3(W/1024,H/4),256>>>(dst,W,H); });
  printf("tile 4 rows x 16 KB         %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<16,4><<<dim3(W/1024,H/16),256>>>(dst,W,H); });
  printf("tile 16 rows x 16 KB        %.3f ms  %.2f TB/s\n",ms,gb/ms);
  ms=timeit([&]{ tile_store<1,16><<<dim3(W/4096,H/1),256>>>(dst,W,H); });
  printf("tile 1 row x 64 KB          %.3f ms  %.2f TB/s\n",ms,gb/ms);
  return 0;
}
