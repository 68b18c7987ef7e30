#!/usr/bin/env python3
"""Generates bank_rate.hip: v_fmac_f32 / v_pk_fma_f32 streams with controlled VGPR bank
patterns (bank = register index mod 4), to find what limits VALU issue on gfx950."""
N = 256          # instructions per loop body


def fmac(pattern):
    lines = []
    for k in range(N):
        acc = 32 + (k % 64)
        b = acc % 4
        if pattern == "A":      # three different banks
            t, s = 8 + (b + 1) % 4, 12 + (b + 2) % 4
        elif pattern == "B":    # tap and sample in one bank
            t, s = 8 + (b + 1) % 4, 12 + (b + 1) % 4
        elif pattern == "C":    # accumulator and tap in one bank
            t, s = 8 + b, 12 + (b + 2) % 4
        elif pattern == "D":    # all in one bank
            t, s = 8 + b, 12 + b
        elif pattern == "E":    # accumulator and sample in one bank
            t, s = 8 + (b + 1) % 4, 12 + b
        lines.append("v_fmac_f32 v%d, v%d, v%d" % (acc, t, s))
    return lines


def fma3(pattern):              # v_fma_f32 d, a, b, c  (VOP3, d == c)
    return [l.replace("v_fmac_f32 v%s," % l.split()[1][1:-1], "v_fma_f32 v%s," % l.split()[1][1:-1]) + ", " + l.split()[1][:-1]
            for l in fmac(pattern)]


def pk(pattern):
    lines = []
    for k in range(N // 2):
        acc = 32 + 2 * (k % 32)
        b = acc % 4             # 0 or 2
        if pattern == "A":
            t, s = 8 + ((b + 2) % 4), 16 + ((b + 2) % 4)
            # tap pair in bank pair (b+2), sample pair same bank pair as tap -> use distinct below
            t, s = 8 + (b + 2) % 4, 20 + b      # tap other pair, sample same pair as acc? see B
        if pattern == "A":      # acc pair banks (b,b+1); tap pair (b+2,b+3); sample pair (b+2,b+3) too is a conflict, so:
            t, s = 8 + (b + 2) % 4, 16 + (b + 2) % 4
        lines.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[%d:%d], v[%d:%d] op_sel_hi:[0,1,1]" %
                     (acc, acc + 1, t, t + 1, s, s + 1, acc, acc + 1))
    return lines


def kernel(name, lines):
    body = "\\n\\t".join(lines)
    clob = ",".join('"v%d"' % i for i in range(8, 96))
    init = "\\n\\t".join("v_mov_b32 v%d, %%0\\n\\tv_xor_b32 %%0, %%0, %%1\\n\\tv_add_u32 %%0, 0x%x, %%0\\n\\tv_and_b32 %%0, 0x007fffff, %%0\\n\\tv_or_b32 %%0, 0x3f800000, %%0" % (i, 0x9e37 * (i + 1)) for i in range(8, 96))
    return '''
__global__ __launch_bounds__(256) void %s(float *out,int iters,int randomize)
{
  if (randomize)
    {
      unsigned seed=(threadIdx.x*2654435761u+blockIdx.x*40503u) & 0x007fffffu | 0x3f800000u,salt=threadIdx.x*0x45d9f3bu;
      asm volatile("%s" : "+v"(seed) : "v"(salt) : %s);
    }
  for (int it=0; it < iters; it++)
    asm volatile("%s" ::: %s);
  if (iters < 0) out[threadIdx.x]=0;
}
''' % (name, init, clob, body, clob)


src = ['#include <hip/hip_runtime.h>', '#include <cstdio>']
names = []
for p in "AD":
    src.append(kernel("fmac_" + p, fmac(p)))
    names.append(("fmac_" + p, N, 2))
for p in "":
    src.append(kernel("fma3_" + p, fma3(p)))
    names.append(("fma3_" + p, N, 2))
src.append(kernel("pk_A", pk("A")))
names.append(("pk_A", N // 2, 4))
src.append('''
template<typename K> void run(const char *name,K k,int instr,int flop,int bpc,int randomize)
{
  float *out; hipMalloc(&out,4096);
  int iters=2048,grid=256*bpc;
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  k<<<grid,256>>>(out,iters,randomize); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i=0; i < 3; i++) k<<<grid,256>>>(out,iters,randomize);
  hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms,a,b); ms/=3;
  double per_simd=(double) bpc*iters*instr;
  printf("%-8s random=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  %.2f cyc/instr/SIMD @2.4GHz\\n",name,randomize,bpc,ms,
    (double) grid*4*iters*instr*64.0*flop/(ms*1e-3)/1e12,ms*1e-3*2.4e9/per_simd);
  hipFree(out);
}
int main()
{
  for (int randomize : {0,1})
  for (int bpc : {2,4})
    {
''')
for n, instr, flop in names:
    src.append('      run("%s",%s,%d,%d,bpc,randomize);' % (n, n, instr, flop))
src.append("    }\n  return 0;\n}\n")
open("bank_rate.hip", "w").write("\n".join(src))
