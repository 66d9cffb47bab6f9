// Microbenchmark: f32 VALU issue rates on gfx950 (plain vs packed ops, SGPR operand), to size
// the exact-L2 scan kernel's ceiling.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s0, float s1) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float x = threadIdx.x * 0.5f;
  for (int i = 0; i < N_ITER; ++i) {
    if (MODE == 0) {  // 8 independent v_add_f32 (VGPR operands)
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    } else if (MODE == 1) {  // 4 v_pk_add_f32 = 8 adds
      asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                   : "v"(*(double*)&a0));
    } else if (MODE == 2) {  // sub(SGPR) / mul / add chain pattern of the scan kernel, 2 queries
      asm volatile("v_sub_f32 %2, %4, %6\n v_sub_f32 %3, %5, %6\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                   "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s0), "s"(s1), "v"(x));
    } else if (MODE == 3) {  // same with packed ops: one pk_add (neg), pk_mul, pk_add for 2 queries
      asm volatile("v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %1, %1, %1\n v_pk_add_f32 %0, %0, %1\n"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2) : "v"(*(double*)&a4), "v"(*(double*)&a6));
    } else if (MODE == 4) {  // v_fma chain (reference point for peak)
      asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                   "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
double run(int blocks, float* d, int ops_per_iter) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<MODE><<<blocks, 256>>>(d, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double lane_ops = 5.0 * blocks * 256.0 * N_ITER * ops_per_iter;
  return lane_ops / (ms * 1e-3) / 1e12;
}
int main() {
  float* d;
  hipMalloc(&d, 8192 * 256 * 4);
  for (int wpc : {4, 8, 16, 32}) {
    int blocks = 256 * wpc / 4;
    printf("waves/CU=%2d  v_add %.1f  v_pk_add %.1f  sub/mul/add(sgpr) %.1f  pk-sub/mul/add %.1f  v_fma(x2 flop) %.1f  [T lane-ops/s]\n",
           wpc, run<0>(blocks, d, 8), run<1>(blocks, d, 8), run<2>(blocks, d, 6), run<3>(blocks, d, 6), run<4>(blocks, d, 8));
  }
  return 0;
}
