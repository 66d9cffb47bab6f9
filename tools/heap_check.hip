// development aid: wave-parallel heap operations (kernels_graph.h) against the serial BinaryHeap restatement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../fabstir-vectordb_amd/csrc/kernels_graph.h"
using namespace fvdb;

__global__ void check_kernel(const float* vals, uint32_t nsteps, uint32_t ef, uint32_t* bad) {
  __shared__ HItem hs[80], cs[1100], cp[1100];
  const int lane = threadIdx.x;
  uint32_t ns = 0, nr = 0, ncs = 0, ncp = 0;
  uint32_t rn = 0;
  float rd = 0;
  for (uint32_t s = 0; s < nsteps; ++s) {
    const float d = vals[blockIdx.x * nsteps + s];
    if (lane == 0) {
      h_push(hs, ns, HItem{s, -d});
      h_push(cs, ncs, HItem{s, d});
    }
    lds_push_parallel(cp, ncp, HItem{s, d}, lane);
    rh_push(rn, rd, nr, s, -d, lane);
    if (nr > ef) {
      if (lane == 0) (void)h_pop(hs, ns);
      rh_pop(rn, rd, nr, lane);
    }
    __syncthreads();
    ns = __shfl(ns, 0);
    ncs = __shfl(ncs, 0);
    bool ok = true;
    if ((uint32_t)lane < nr) ok = hs[lane].node == rn && hs[lane].d == rd;
    if (ns != nr) ok = false;
    for (uint32_t i = lane; i < ncp; i += 64) ok = ok && cs[i].node == cp[i].node && cs[i].d == cp[i].d;
    if (ncs != ncp) ok = false;
    if (!ok) atomicAdd(bad, 1u);
    if ((s % 3) == 2 && ncs > 0) {
      HItem r1 = HItem{0, 0};
      if (lane == 0) r1 = h_pop(cs, ncs);
      const HItem r2 = lds_pop_parallel(cp, ncp, lane);
      ncs = __shfl(ncs, 0);
      if (lane == 0 && (r1.node != r2.node || r1.d != r2.d)) atomicAdd(bad, 1000u);
    }
    __syncthreads();
  }
}

int main() {
  const uint32_t nsteps = 400, blocks = 256;
  std::vector<float> v(nsteps * blocks);
  srand(1);
  for (auto& x : v) x = (rand() % 4 == 0) ? (float)(rand() % 10) * 0.1f : (float)rand() / RAND_MAX;
  float* dv;
  uint32_t* dbad;
  hipMalloc(&dv, v.size() * 4);
  hipMalloc(&dbad, 4);
  hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice);
  for (uint32_t ef : {1u, 2u, 5u, 50u, 63u}) {
    hipMemset(dbad, 0, 4);
    hipLaunchKernelGGL(check_kernel, dim3(blocks), dim3(64), 0, 0, dv, nsteps, ef, dbad);
    uint32_t bad = 0;
    hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("ef %u: mismatching steps %u\n", ef, bad);
  }
  return 0;
}
