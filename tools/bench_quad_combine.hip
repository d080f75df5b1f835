// Micro-benchmark + check of the QUAD-split scan operators of csrc/ci_quad.h against the one-lane
// operators of csrc/ci_linalg.h they replace in the trend + seasonal kernel (round-5 review, item 1:
// "micro-benchmark the d = 7 quad combine first and publish it either way").
//   * correctness: random well-conditioned elements, quad combine vs felems_combine (max abs diff);
//     quad compose vs aelem_compose; the in-quad transpose
//   * cycles per combine with one wavefront per SIMD (the kernel's occupancy), operands in registers
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -I tfp-causalimpact_amd/csrc \
//         tools/bench_quad_combine.hip -o tools/build/bench_quad_combine && tools/build/bench_quad_combine
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "ci_quad.h"

using namespace ci;

// per element: A (D*D) | b (D) | C (D*D, symmetric) | eta (D) | J (D*D, symmetric)
template <int D> constexpr int esz() { return 3 * D * D + 2 * D; }

template <int D> __device__ FElemS<D> load_one(const float* p) {
  FElemS<D> e;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) e.A.m[i][j] = p[i * D + j];
  for (int i = 0; i < D; ++i) e.b.v[i] = p[D * D + i];
  for (int i = 0; i < D; ++i)
    for (int j = i; j < D; ++j) e.C[symidx<D>(i, j)] = p[D * D + D + i * D + j];
  for (int i = 0; i < D; ++i) e.eta.v[i] = p[2 * D * D + D + i];
  for (int i = 0; i < D; ++i)
    for (int j = i; j < D; ++j) e.J[symidx<D>(i, j)] = p[2 * D * D + 2 * D + i * D + j];
  return e;
}
template <int D> __device__ void store_one(float* p, const FElemS<D>& e) {
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) p[i * D + j] = e.A.m[i][j];
  for (int i = 0; i < D; ++i) p[D * D + i] = e.b.v[i];
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) p[D * D + D + i * D + j] = e.C[symidx<D>(i, j)];
  for (int i = 0; i < D; ++i) p[2 * D * D + D + i] = e.eta.v[i];
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) p[2 * D * D + 2 * D + i * D + j] = e.J[symidx<D>(i, j)];
}
template <int D> __device__ QFElem<D> load_quad(const float* p, int q) {
  QFElem<D> e;
  constexpr int H = QMat<D>::H;
  for (int h = 0; h < H; ++h) {
    const int j = q + 4 * h;
    for (int i = 0; i < D; ++i) {
      qp_set(e.A.r[i], h, j < D ? p[i * D + j] : 0.f);
      qp_set(e.AT.r[i], h, j < D ? p[j * D + i] : 0.f);
      qp_set(e.C.r[i], h, j < D ? p[D * D + D + i * D + j] : 0.f);
      qp_set(e.J.r[i], h, j < D ? p[2 * D * D + 2 * D + i * D + j] : 0.f);
    }
    qp_set(e.b.v, h, j < D ? p[D * D + j] : 0.f);
    qp_set(e.eta.v, h, j < D ? p[2 * D * D + D + j] : 0.f);
  }
  return e;
}
template <int D> __device__ void store_quad(float* p, const QFElem<D>& e, int q) {
  constexpr int H = QMat<D>::H;
  for (int h = 0; h < H; ++h) {
    const int j = q + 4 * h;
    if (j >= D) continue;
    for (int i = 0; i < D; ++i) {
      p[i * D + j] = qp_get(e.A.r[i], h);
      p[D * D + D + i * D + j] = qp_get(e.C.r[i], h);
      p[2 * D * D + 2 * D + i * D + j] = qp_get(e.J.r[i], h);
    }
    p[D * D + j] = qp_get(e.b.v, h);
    p[2 * D * D + D + j] = qp_get(e.eta.v, h);
  }
}

// one-lane: thread t combines pair t.  quad: quad t / 4 combines pair t / 4.
template <int D>
__global__ __launch_bounds__(256) void k_one(const float* in, float* out, long long* cyc, int reps) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  FElemS<D> e1 = load_one<D>(in + (size_t)(2 * t) * esz<D>());
  const FElemS<D> e2 = load_one<D>(in + (size_t)(2 * t + 1) * esz<D>());
  FElemS<D> r = felems_combine(e1, e2);
  store_one<D>(out + (size_t)t * esz<D>(), r);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < reps; ++i) e1 = felems_combine(e1, e2);
  const long long t1 = clock64();
  if (e1.b.v[0] == 1234.5f) out[0] = 1.f;       // keep the loop
  if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}
template <int D>
__global__ __launch_bounds__(256) void k_quad(const float* in, float* out, float* out_t, long long* cyc,
                                              int reps) {
  const int t = blockIdx.x * 256 + threadIdx.x, pair = t >> 2, q = t & 3;
  QFElem<D> e1 = load_quad<D>(in + (size_t)(2 * pair) * esz<D>(), q);
  const QFElem<D> e2 = load_quad<D>(in + (size_t)(2 * pair + 1) * esz<D>(), q);
  QFElem<D> r = qf_combine<D>(e1, e2, q);
  store_quad<D>(out + (size_t)pair * esz<D>(), r, q);
  {
    // the transpose: K(A) -> K(A') must equal the AT the combine carries
    const QMat<D> tr = q_transpose(r.A, q);
    for (int h = 0; h < QMat<D>::H; ++h)
      for (int i = 0; i < D; ++i)
        if (q + 4 * h < D) out_t[(size_t)pair * D * D + (q + 4 * h) * D + i] = qp_get(tr.r[i], h) - qp_get(r.AT.r[i], h);
  }
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < reps; ++i) e1 = qf_combine<D>(e1, e2, q);
  const long long t1 = clock64();
  long long ts = 0;
  {
    const long long a = clock64();
    for (int i = 0; i < reps; ++i) e1 = qf_combine<D, true>(e1, e2, q);
    ts = (clock64() - a) / reps;
  }
  if (qp_get(e1.b.v, 0) == 1234.5f) out[0] = 1.f;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = (t1 - t0) / reps; cyc[2 * blockIdx.x + 1] = ts; }
}
// one Kogge-Stone level inside a wavefront, as the kernel runs it: 60 ds_bpermutes + a masked combine
template <int D>
__global__ __launch_bounds__(256) void k_level(const float* in, float* out, long long* cyc, int reps) {
  const int t = blockIdx.x * 256 + threadIdx.x, pair = t >> 2, q = t & 3, qi = (threadIdx.x & 63) >> 2;
  QFElem<D> incl = load_quad<D>(in + (size_t)(2 * pair) * esz<D>(), q);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < reps; ++i) {
    const int off = 1 << (i & 3);
    const QFElem<D> o = q_shfl_up(incl, off);
    if (qi >= off) incl = qf_combine<D>(o, incl, q);
  }
  const long long t1 = clock64();
  QFElem<D> sh = incl;
  const long long t2 = clock64();
  for (int i = 0; i < reps; ++i) sh = q_shfl_up(sh, 1 << (i & 3));
  const long long t3 = clock64();
  store_quad<D>(out + (size_t)pair * esz<D>(), incl, q);
  if (qp_get(sh.b.v, 0) == 1234.5f) out[0] = 1.f;
  if (threadIdx.x == 0) { cyc[0] = (t1 - t0) / reps; cyc[1] = (t3 - t2) / reps; }
}

// backward maps: pairs of (M, c)
template <int D>
__global__ __launch_bounds__(256) void k_amap(const float* in, float* out_one, float* out_quad, long long* cyc,
                                              int reps) {
  const int t = blockIdx.x * 256 + threadIdx.x, pair = t >> 2, q = t & 3;
  constexpr int AS = D * D + D;
  const float* p1 = in + (size_t)(2 * pair) * AS;
  const float* p2 = p1 + AS;
  AElem<D> a1, a2;
  QAElem<D> q1, q2;
  for (int i = 0; i < D; ++i) {
    for (int j = 0; j < D; ++j) { a1.M.m[i][j] = p1[i * D + j]; a2.M.m[i][j] = p2[i * D + j]; }
    a1.c.v[i] = p1[D * D + i]; a2.c.v[i] = p2[D * D + i];
    q1.c[i] = a1.c.v[i]; q2.c[i] = a2.c.v[i];
  }
  q1.M = qm_from(a1.M, q); q2.M = qm_from(a2.M, q);
  const AElem<D> r1 = aelem_compose(a1, a2);
  QAElem<D> rq = qa_compose(q1, q2, q);
  if (q == 0)
    for (int i = 0; i < D; ++i) {
      for (int j = 0; j < D; ++j) out_one[(size_t)pair * AS + i * D + j] = r1.M.m[i][j];
      out_one[(size_t)pair * AS + D * D + i] = r1.c.v[i];
      out_quad[(size_t)pair * AS + D * D + i] = rq.c[i];
    }
  for (int h = 0; h < QMat<D>::H; ++h)
    for (int i = 0; i < D; ++i)
      if (q + 4 * h < D) out_quad[(size_t)pair * AS + i * D + q + 4 * h] = qp_get(rq.M.r[i], h);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < reps; ++i) rq = qa_compose(rq, q2, q);
  const long long t1 = clock64();
  if (rq.c[0] == 1234.5f) out_one[0] = 1.f;
  if (threadIdx.x == 0) cyc[blockIdx.x] = (t1 - t0) / reps;
}

template <int D> void run() {
  const int NP = 256, ES = esz<D>();        // pairs
  std::mt19937 g(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> h((size_t)2 * NP * ES);
  for (int e = 0; e < 2 * NP; ++e) {
    float* p = h.data() + (size_t)e * ES;
    std::vector<float> L(D * D), M(D * D);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        p[i * D + j] = (i == j ? 0.7f : 0.f) + 0.15f * nd(g);       // A
        L[i * D + j] = 0.3f * nd(g);
        M[i * D + j] = 0.5f * nd(g);
      }
    for (int i = 0; i < D; ++i) { p[D * D + i] = nd(g); p[2 * D * D + D + i] = nd(g); }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        float c = i == j ? 0.05f : 0.f, jj = i == j ? 0.2f : 0.f;
        for (int k = 0; k < D; ++k) { c += L[i * D + k] * L[j * D + k]; jj += M[i * D + k] * M[j * D + k]; }
        p[D * D + D + i * D + j] = c;                               // C = L L' + 0.05 I
        p[2 * D * D + 2 * D + i * D + j] = jj;                      // J = M M' + 0.2 I
      }
  }
  float *din, *o1, *oq, *ot;
  long long* cy;
  hipMalloc(&din, h.size() * 4); hipMalloc(&o1, (size_t)NP * ES * 4); hipMalloc(&oq, (size_t)NP * ES * 4);
  hipMalloc(&ot, (size_t)NP * D * D * 4); hipMalloc(&cy, 64 * 8);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(ot, 0, (size_t)NP * D * D * 4);
  const int reps = 50;
  hipLaunchKernelGGL(k_one<D>, dim3(NP / 256), dim3(256), 0, 0, din, o1, cy, reps);
  hipDeviceSynchronize();
  long long c_one;
  hipMemcpy(&c_one, cy, 8, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_quad<D>, dim3(NP * 4 / 256), dim3(256), 0, 0, din, oq, ot, cy, reps);
  hipDeviceSynchronize();
  long long c_quad[2];
  hipMemcpy(c_quad, cy, 16, hipMemcpyDeviceToHost);
  {
    hipLaunchKernelGGL(k_level<D>, dim3(1), dim3(256), 0, 0, din, oq, cy, 8);
    hipDeviceSynchronize();
    long long c2[2];
    hipMemcpy(c2, cy, 16, hipMemcpyDeviceToHost);
    printf("d=%d one scan level inside a wavefront (shuffle of the element + masked combine): %lld cycles; the shuffle alone %lld\n",
           D, c2[0], c2[1]);
    hipLaunchKernelGGL(k_quad<D>, dim3(NP * 4 / 256), dim3(256), 0, 0, din, oq, ot, cy, reps);
    hipDeviceSynchronize();
  }
  std::vector<float> a((size_t)NP * ES), b((size_t)NP * ES), tt((size_t)NP * D * D);
  hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), oq, b.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(tt.data(), ot, tt.size() * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0, mt = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = std::fmax(md, std::fabs((double)a[i] - b[i])); mx = std::fmax(mx, std::fabs((double)a[i])); }
  for (float v : tt) mt = std::fmax(mt, std::fabs((double)v));
  printf("d=%d filtering elements: quad vs one-lane combine over %d random pairs: max |diff| %.3g (max |value| %.3g); "
         "K(A') carried vs transposed K(A): max |diff| %.3g\n", D, NP, md, mx, mt);
  printf("d=%d cycles per combine (one wavefront per SIMD, hot): one lane %lld | quad %lld (%.2fx) | quad, state only %lld\n",
         D, c_one, c_quad[0], (double)c_one / (double)c_quad[0], c_quad[1]);
  // backward maps
  {
    constexpr int AS = D * D + D;
    std::vector<float> hm((size_t)2 * NP * AS);
    for (auto& v : hm) v = 0.4f * nd(g);
    float *dm, *r1, *r2;
    hipMalloc(&dm, hm.size() * 4); hipMalloc(&r1, (size_t)NP * AS * 4); hipMalloc(&r2, (size_t)NP * AS * 4);
    hipMemcpy(dm, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_amap<D>, dim3(NP * 4 / 256), dim3(256), 0, 0, dm, r1, r2, cy, reps);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cy, 8, hipMemcpyDeviceToHost);
    std::vector<float> x((size_t)NP * AS), y((size_t)NP * AS);
    hipMemcpy(x.data(), r1, x.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(y.data(), r2, y.size() * 4, hipMemcpyDeviceToHost);
    double d2 = 0;
    for (size_t i = 0; i < x.size(); ++i) d2 = std::fmax(d2, std::fabs((double)x[i] - y[i]));
    printf("d=%d backward maps: quad vs one-lane compose: max |diff| %.3g; quad compose %lld cycles\n", D, d2, c);
    hipFree(dm); hipFree(r1); hipFree(r2);
  }
  hipFree(din); hipFree(o1); hipFree(oq); hipFree(ot); hipFree(cy);
}

int main() {
  run<7>();
  run<8>();
  run<3>();
  return 0;
}
