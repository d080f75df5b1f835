// Micro-benchmark of the wave-cooperative scan operators of csrc/ci_seasonal_tp.h: cycles of the
// FIRST call of tp_combine / tp_bcompose in a workgroup (cold instruction cache, operands in L2) and
// of the following calls (hot), with 1, 4 or 8 wavefronts of the workgroup calling at once.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I tfp-causalimpact_amd/csrc \
//         tools/bench_tp_combine.hip -o tools/build/bench_tp_combine && tools/build/bench_tp_combine
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "ci_seasonal_tp.h"

template <int NR>
__global__ __launch_bounds__(512) void bench(float* ws, long long* out, int D, int reps, int active) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ci::TpL base = (ci::TpL)(smem + (size_t)wave * 16384);
  ci::TpL scr = base; ci::TpL vb = base + NR * NR; ci::TpL piv = vb + 2 * NR;
  const size_t ESZ = ci::tp_esz(NR), BSZ = ci::tp_bsz(NR);
  ci::TpG e1 = (ci::TpG)ws + (size_t)wave * 4 * ESZ; ci::TpG e2 = e1 + ESZ; ci::TpG eo = e2 + ESZ; ci::TpG so = eo + ESZ;
  if (wave >= active) return;
  for (int r = 0; r < reps; ++r) {
    const long long t0 = clock64();
    ci::tp_combine<NR, false>(e1, e2, eo, scr, vb, piv, D, lane);
    const long long t1 = clock64();
    ci::tp_combine<NR, true>(e1, e2, so, scr, vb, piv, D, lane);
    const long long t2 = clock64();
    ci::tp_bcompose<NR>(so, so, eo, scr, vb, D, lane);
    const long long t3 = clock64();
    if (lane == 0) { out[(wave * reps + r) * 3] = t1 - t0; out[(wave * reps + r) * 3 + 1] = t2 - t1; out[(wave * reps + r) * 3 + 2] = t3 - t2; }
  }
}

template <int NR> void run(int D) {
  const size_t ESZ = ci::tp_esz(NR);
  std::vector<float> h(8 * 4 * ESZ, 0.f);
  for (int w = 0; w < 8; ++w)
    for (int e = 0; e < 2; ++e) {
      float* p = h.data() + ((size_t)w * 4 + e) * ESZ;
      for (int i = 0; i < D; ++i) {
        float* row = p + (size_t)i * (3 * NR + 4);
        row[i] = 0.9f; row[NR + i] = 0.5f; row[2 * NR + i] = 2.0f;        // A, C, J diagonal
        for (int j = 0; j < D; ++j) if (j != i) { row[NR + j] = 0.01f; row[2 * NR + j] = 0.02f; }
        row[3 * NR] = 0.1f * i; row[3 * NR + 1] = 0.2f;
      }
    }
  float* d; long long* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 8 * 8 * 3 * 8);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)bench<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
  for (int active : {1, 4, 8}) {
    const int reps = 6;
    hipLaunchKernelGGL(bench<NR>, dim3(1), dim3(512), 8 * 16384, 0, d, o, D, reps, active);
    hipDeviceSynchronize();
    std::vector<long long> ho(8 * reps * 3);
    hipMemcpy(ho.data(), o, ho.size() * 8, hipMemcpyDeviceToHost);
    printf("NR=%d D=%d waves=%d  combine first %lld then %lld %lld | state first %lld then %lld | bcompose first %lld then %lld\n",
           NR, D, active, ho[0], ho[3], ho[3 * (reps - 1)], ho[1], ho[3 * (reps - 1) + 1], ho[2], ho[3 * (reps - 1) + 2]);
  }
  hipFree(d); hipFree(o);
}

int main() {
  run<8>(8); run<16>(13); run<20>(18); run<24>(18); run<32>(32);
  return 0;
}
