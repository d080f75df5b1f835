// Micro-benchmark of the order-statistics kernel (ci_summary.h) on the fit_causalimpact shape:
// two [T, N] float64 matrices, R ranks.  Build: see tools/bench_select.sh.  Prints the average
// launch time and the algorithmic HBM rate (2 * T * N * 8 bytes read once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#include "../tfp-causalimpact_amd/csrc/ci_summary.h"

#ifndef BENCH_NT
#define BENCH_NT 256
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1000, N = argc > 2 ? atoi(argv[2]) : 8000;
  const int preT = argc > 3 ? atoi(argv[3]) : 0;       // leading rows of M1 that are constant
  std::vector<int> ranks = {N / 40 - 1, N / 40, N / 2 - 1, N / 2, N - N / 40 - 1, N - N / 40};
  const int R = (int)ranks.size();
  std::vector<double> h((size_t)2 * T * N);
  std::mt19937_64 g(1);
  const double mean = argc > 4 ? atof(argv[4]) : 100.0;
  std::normal_distribution<double> nd(mean, 2.0);
  for (auto& v : h) v = nd(g);
  for (size_t i = 0; i < (size_t)preT * N; ++i) h[(size_t)T * N + i] = 0.0;
  double *dM, *dout; int* dr;
  CK(hipMalloc(&dM, h.size() * 8)); CK(hipMalloc(&dout, (size_t)2 * R * T * 8)); CK(hipMalloc(&dr, R * 4));
  CK(hipMemcpy(dM, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dr, ranks.data(), R * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    hipLaunchKernelGGL((ci::summ_select_reg_kernel<BENCH_NT, 8192 / BENCH_NT>), dim3(2 * T), dim3(BENCH_NT), 0, 0,
                       N, T, R, T, dr, dM, dM + (size_t)T * N, dout, dout + (size_t)R * T);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  printf("NT=%d T=%d N=%d preT=%d mean=%g: %.1f us per launch, %.2f TB/s\n", BENCH_NT, T, N, preT, mean, us,
         2.0 * T * N * 8 / (us * 1e-6) / 1e12);
#ifdef CI_SEL_PROF
  {
    int zero = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(ci::sel_prof_n), &zero, 4));
    launch(); CK(hipDeviceSynchronize());
    unsigned long long st[64]; int n;
    CK(hipMemcpyFromSymbol(&n, HIP_SYMBOL(ci::sel_prof_n), 4));
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ci::sel_prof), sizeof st));
    for (int i = 0; i < n && i < 32; ++i)
      printf("  tick %llu  +%llu cycles (total %llu)\n", st[2 * i], i ? st[2 * i + 1] - st[2 * i - 1] : 0ull, st[2 * i + 1] - st[1]);
  }
#endif
#ifndef CI_SEL_STOP_AFTER
  std::vector<double> out((size_t)2 * R * T);
  CK(hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int m = 0; m < 2; ++m)
    for (int t = 0; t < T; t += 97) {
      std::vector<double> row(h.begin() + ((size_t)m * T + t) * N, h.begin() + ((size_t)m * T + t + 1) * N);
      std::sort(row.begin(), row.end());
      for (int r = 0; r < R; ++r) bad += out[((size_t)m * R + r) * T + t] != row[ranks[r]];
    }
  printf("mismatches: %d\n", bad);
#endif
  return 0;
}
