// What does a cheap, CORRECT hand-over between workgroups that share an XCD (and its L2) cost on gfx950?
//  * agent-scope read-modify-writes and polls work; workgroup-scope ones do not (the poll hits in L1).
//  * the acquire side: the vector L1 of the reader must drop its stale lines.  Variants measured here,
//    each with a data check (every workgroup re-reads what all others rewrote, so its L1 holds stale
//    copies from the round before): none | fence(acquire, agent) on every wave | ONE wave's
//    buffer_inv sc1 | ONE wave's buffer_inv sc0.
// Every spin is capped: a variant that does not work reports it instead of hanging.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ACQ>
__global__ __launch_bounds__(256) void k_bar(int* cnt, float* buf, long long* out, int G, int reps) {
  const int xcd = blockIdx.x & 7, role = blockIdx.x >> 3, tid = threadIdx.x;
  if (role >= G) return;
  int* c = cnt + xcd * 256;
  int* flag = c + 64;
  float* base = buf + (size_t)xcd * 16 * 256;
  int fails = 0, stale = 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 1; r <= reps; ++r) {
    base[role * 256 + tid] = (float)(r * 100 + role);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == r * G) {
        __hip_atomic_store(flag, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > 2000000) { fails = 1; break; }
        }
      }
      if (ACQ == 2) asm volatile("buffer_inv sc1" ::: "memory");
      if (ACQ == 3) asm volatile("buffer_inv sc0" ::: "memory");
    }
    __syncthreads();
    if (ACQ == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (fails) break;
    // read what every workgroup of the chain wrote this round
    for (int g = 0; g < G; ++g) {
      asm volatile("" ::: "memory");
      const float v = base[g * 256 + tid];            // a plain load: may be served by the vector L1
      if (v != (float)(r * 100 + g)) ++stale;
    }
    __syncthreads();        // nobody rewrites its row before all have read it ... (next barrier orders it)
    // second barrier so that the rewrite of round r + 1 cannot overtake a slow reader
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(c + 128, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == r * G) __hip_atomic_store(flag + 128, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { int spins = 0; while (__hip_atomic_load(flag + 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) { __builtin_amdgcn_s_sleep(1); if (++spins > 2000000) { fails = 1; break; } } }
    }
    __syncthreads();
  }
  const long long t1 = clock64();
  if (stale) atomicAdd((int*)(out + 9), stale);
  if (tid == 0) { if (fails) atomicAdd((int*)(out + 8), 1); if (role == 0) out[xcd] = (t1 - t0) / reps; }
}

template <int ACQ> void run(const char* name, int* cnt, float* buf, long long* out) {
  for (int G : {2, 8, 16}) {
    hipMemset(cnt, 0, 8 * 256 * 4);
    hipMemset(out, 0, 16 * 8);
    hipLaunchKernelGGL(k_bar<ACQ>, dim3(8 * G), dim3(256), 0, 0, cnt, buf, out, G, 300);
    hipDeviceSynchronize();
    long long h[10];
    hipMemcpy(h, out, 80, hipMemcpyDeviceToHost);
    printf("%-42s %2d workgroups: %6lld cycles per round (2 barriers + %d reads)  stale reads %lld%s\n", name, G, h[0], G,
           h[9], h[8] ? "  ** poll timed out **" : "");
  }
}
int main() {
  int* cnt; long long* out; float* buf;
  hipMalloc(&cnt, 8 * 256 * 4); hipMalloc(&out, 16 * 8); hipMalloc(&buf, 8 * 16 * 256 * 4);
  hipMemset(buf, 0, 8 * 16 * 256 * 4);
  run<0>("no acquire", cnt, buf, out);
  run<1>("fence(acquire, agent) on every wave", cnt, buf, out);
  run<2>("buffer_inv sc1 by one wave", cnt, buf, out);
  run<3>("buffer_inv sc0 by one wave", cnt, buf, out);
  return 0;
}
