"""Prototype (numpy, float64) of the chunked time-parallel Durbin-Koopman mean pass in SLOT
coordinates -- the algebra of csrc/ci_seasonal_tp.h before it was written for the device.

Sequential reference = the recursions of csrc/ci_seasonal.h (filter storing K_t, v_t/F_t; backward
r; fast state smoother forward).  Chunked version: per-chunk filtering elements (A, b, C, eta, J)
built by rank-one folds, prefix composition, replay of the filter from each chunk's true predicted
moments; the chunk's backward map is M = (I + J P_start)^-1 A'  (the transpose of the closed-loop
propagator A (I + P_start J)^-1 -- no per-step accumulation), its offset the zero-input backward
pass; suffix chain; second backward pass; forward reconstruction from x^ = a + P r.

  python tools/proto_tp_seasonal.py
"""
import numpy as np


def make_model(T, ns, slope, rng, steps=None):
  tr = 2 if slope else 1
  D = tr + sum(ns)
  offs = np.cumsum([tr] + list(ns))[:-1]
  steps = steps or [1] * len(ns)
  cur = np.zeros((len(ns), T), int)
  change = np.zeros((len(ns), T), bool)
  for k, n in enumerate(ns):
    c = 0
    for t in range(T):
      cur[k, t] = c
      if (t + 1) % steps[k] == 0 and t + 1 < T:
        change[k, t] = True
        c = (c + 1) % n
  mask = rng.random(T) < 0.1
  mask[T - T // 5:] = True
  y = rng.normal(size=T)
  return dict(T=T, D=D, tr=tr, ns=ns, offs=offs, cur=cur, change=change, mask=mask, y=y,
              H=0.5, ql=0.04, qs=0.01, qd=[0.03 * (k + 1) for k in range(len(ns))])


def zvec(m, t):
  z = np.zeros(m["D"])
  z[0] = 1.0
  for k in range(len(m["ns"])):
    z[m["offs"][k] + m["cur"][k, t]] = 1.0
  return z


def trans(m):
  Tm = np.eye(m["D"])
  if m["tr"] == 2:
    Tm[0, 1] = 1.0
  return Tm


def qmat(m, t):
  Q = np.zeros((m["D"], m["D"]))
  Q[0, 0] = m["ql"]
  if m["tr"] == 2:
    Q[1, 1] = m["qs"]
  for k, n in enumerate(m["ns"]):
    if m["change"][k, t]:
      g = np.zeros(m["D"])
      g[m["offs"][k]:m["offs"][k] + n] = -1.0 / n
      g[m["offs"][k] + m["cur"][k, t]] += 1.0
      Q += m["qd"][k] * np.outer(g, g)
  return Q


def prior(m, rng):
  D = m["D"]
  a = rng.normal(size=D) * 0.1
  P = np.zeros((D, D))
  P[0, 0] = 1.3
  if m["tr"] == 2:
    P[1, 1] = 0.7
  for k, n in enumerate(m["ns"]):
    o = m["offs"][k]
    P[o:o + n, o:o + n] = 0.9 * (np.eye(n) - 1.0 / n)
    a[o:o + n] -= a[o:o + n].mean()
  return a, P


def filter_range(m, a, P, s, e, K, VF):
  Tm = trans(m)
  for t in range(s, e):
    z = zvec(m, t)
    if not m["mask"][t]:
      pz = P @ z
      F = z @ pz + m["H"]
      v = m["y"][t] - z @ a
      K[t] = pz / F
      VF[t] = v / F
      a = a + K[t] * v
      P = P - np.outer(pz, pz) / F
    else:
      K[t] = 0.0
      VF[t] = 0.0
    if t + 1 < m["T"]:
      a = Tm @ a
      P = Tm @ P @ Tm.T + qmat(m, t)
  return a, P


def backward_range(m, r, s, e, K, VF, RS=None):
  Tm = trans(m)
  for t in range(e - 1, s - 1, -1):
    if t + 1 < m["T"]:
      r = Tm.T @ r
    else:
      r = np.zeros_like(r)
    if not m["mask"][t]:
      z = zvec(m, t)
      r = r + z * (VF[t] - K[t] @ r)
    if RS is not None:
      RS[t] = r
  return r


def forward_range(m, xh, s, e, RS, r_end, X):
  Tm = trans(m)
  for t in range(s, e):
    X[t] = xh
    if t + 1 < m["T"]:
      rn = RS[t + 1] if t + 1 < e else r_end
      xh = Tm @ xh + qmat(m, t) @ rn
  return xh


def sequential(m, a1, P1):
  T, D = m["T"], m["D"]
  K, VF, RS, X = np.zeros((T, D)), np.zeros(T), np.zeros((T, D)), np.zeros((T, D))
  filter_range(m, a1, P1, 0, T, K, VF)
  backward_range(m, np.zeros(D), 0, T, K, VF, RS)
  forward_range(m, a1 + P1 @ RS[0], 0, T, RS, np.zeros(D), X)
  return X


def element(m, s, e):
  D = m["D"]
  Tm = trans(m)
  A, b, C, eta, J = np.eye(D), np.zeros(D), np.zeros((D, D)), np.zeros(D), np.zeros((D, D))
  for t in range(s, e):
    if not m["mask"][t]:
      z = zvec(m, t)
      cz = C @ z
      za = A.T @ z
      S = z @ cz + m["H"]
      inn = (m["y"][t] - z @ b) / S
      eta = eta + za * inn
      J = J + np.outer(za, za) / S
      b = b + cz * inn
      A = A - np.outer(cz, za) / S
      C = C - np.outer(cz, cz) / S
    if t + 1 < m["T"]:
      A = Tm @ A
      b = Tm @ b
      C = Tm @ C @ Tm.T + qmat(m, t)
  return A, b, C, eta, J


def combine(e1, e2):
  A1, b1, C1, h1, J1 = e1
  A2, b2, C2, h2, J2 = e2
  D = len(b1)
  W = np.eye(D) + C1 @ J2
  Wi = np.linalg.inv(W)
  G = Wi @ A1
  A = A2 @ G
  b = A2 @ Wi @ (b1 + C1 @ h2) + b2
  C = A2 @ Wi @ C1 @ A2.T + C2
  h = G.T @ (h2 - J2 @ b1) + h1
  J = G.T @ J2 @ A1 + J1
  return A, b, C, h, J


def chunked(m, a1, P1, N):
  T, D = m["T"], m["D"]
  Lc = -(-T // N)
  Lc = (Lc + 3) // 4 * 4
  bounds = [(min(c * Lc, T), min((c + 1) * Lc, T)) for c in range(N)]
  els = [element(m, s, e) for (s, e) in bounds]
  # Kogge-Stone inclusive prefix, the prior as the element in front of chunk 0
  pri = (np.zeros((D, D)), a1, P1, np.zeros(D), np.zeros((D, D)))
  incl = [combine(pri, els[0])] + els[1:]
  off = 1
  while off < N:
    incl = [combine(incl[c - off], incl[c]) if c >= off else incl[c] for c in range(N)]
    off *= 2
  starts = [(a1, P1)] + [(incl[c][1], incl[c][2]) for c in range(N - 1)]
  K, VF, RS, X = np.zeros((T, D)), np.zeros(T), np.zeros((T, D)), np.zeros((T, D))
  for c, (s, e) in enumerate(bounds):
    a_end, P_end = filter_range(m, starts[c][0], starts[c][1], s, e, K, VF)
    if c + 1 < N and bounds[c + 1][0] < T:
      assert np.allclose(a_end, starts[c + 1][0], atol=1e-9) and np.allclose(P_end, starts[c + 1][1], atol=1e-9)
  # backward maps from the elements: M = (I + J P_start)^-1 A', offset = zero-input pass
  Ms, cs = [], []
  for c, (s, e) in enumerate(bounds):
    A, _, _, _, J = els[c]
    Ms.append(np.linalg.solve(np.eye(D) + J @ starts[c][1], A.T))
    cs.append(backward_range(m, np.zeros(D), s, e, K, VF))
  r_end = [np.zeros(D) for _ in range(N)]
  for c in range(N - 2, -1, -1):
    r_end[c] = Ms[c + 1] @ r_end[c + 1] + cs[c + 1]
  for c, (s, e) in enumerate(bounds):
    backward_range(m, r_end[c], s, e, K, VF, RS)
  for c, (s, e) in enumerate(bounds):
    if s >= T:
      continue
    xh = starts[c][0] + starts[c][1] @ RS[s]
    forward_range(m, xh, s, e, RS, r_end[c], X)
  return X


if __name__ == "__main__":
  rng = np.random.default_rng(0)
  for (T, ns, slope, steps, N) in [(97, [4, 7, 6], False, [1, 2, 1], 8), (203, [7], True, [1], 16),
                                   (60, [2, 3], True, [3, 1], 4), (331, [], True, None, 16),
                                   (500, [24, 7], False, [1, 24], 32)]:
    m = make_model(T, ns, slope, rng, steps)
    a1, P1 = prior(m, rng)
    X0 = sequential(m, a1, P1)
    X1 = chunked(m, a1, P1, N)
    print(T, ns, slope, N, "max |chunked - sequential| =", np.abs(X0 - X1).max())
    assert np.abs(X0 - X1).max() < 1e-8
