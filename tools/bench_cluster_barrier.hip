// Cost of one hand-over barrier between the workgroups of a chain's cluster (ci_wide_quad.h dk_barrier),
// as a function of the number of workgroups and of the mode (one XCD: relaxed arrival after the
// stores have reached the shared L2; several XCDs: agent-scope release with L2 write-back).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I tfp-causalimpact_amd/csrc \
//         tools/bench_cluster_barrier.hip -o tools/build/bench_cluster_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "ci_wide.h"

using namespace ci;

// blocks are dealt to the XCDs round-robin: block b sits on XCD b % 8; a "chain" takes blocks equal mod 8
__global__ __launch_bounds__(256) void k_bar(int* cnt, float* buf, long long* out, int G, int light, int reps,
                                             int stores) {
  const int xcd = blockIdx.x & 7, role = blockIdx.x >> 3, tid = threadIdx.x;
  if (role >= G) return;
  DkSync sy;
  sy.cnt = cnt + xcd * 256; sy.flag = cnt + xcd * 256 + 64; sy.latcnt = cnt + xcd * 256 + 128; sy.latflag = cnt + xcd * 256 + 192;
  sy.Gd = G; sy.epoch = 0; sy.cluster = G > 1; sy.light = light != 0;
  float* mine = buf + (size_t)(xcd * 16 + role) * 256 * 64 + tid;      // coalesced: [float][lane]
  dk_barrier(sy, tid);
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < stores; ++i) mine[i * 256] = (float)(r + i);      // something for the release to cover
    dk_barrier(sy, tid);
  }
  const long long t1 = clock64();
  if (tid == 0 && role == 0) out[xcd] = (t1 - t0) / reps;
}

int main() {
  int* cnt; float* buf; long long* out;
  hipMalloc(&cnt, 8 * 256 * 4); hipMalloc(&buf, (size_t)8 * 16 * 256 * 64 * 4); hipMalloc(&out, 8 * 8);
  for (int light : {1, 0})
    for (int G : {2, 4, 8, 16})
      for (int stores : {0, 16, 64}) {
        hipMemset(cnt, 0, 8 * 256 * 4);
        hipLaunchKernelGGL(k_bar, dim3(8 * G), dim3(256), 0, 0, cnt, buf, out, G, light, 200, stores);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%s, %2d workgroups per chain (8 chains), %2d floats stored per lane before it: %lld cycles per barrier\n",
               light ? "one XCD (relaxed arrival)" : "agent-scope release    ", G, stores, h[0]);
      }
  return 0;
}
