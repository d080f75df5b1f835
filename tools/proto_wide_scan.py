"""Numerical prototype (float32) of the chunked Sarkka scan for the weekly-seasonal model.

Checks that chunk elements built by the O(d^2) recursion + a Hillis-Steele scan of 256 dense
elements in float32 reproduce the float64 sequential Kalman filter at the chunk boundaries.
"""
import sys
import numpy as np

def model(TR, NS):
  D = TR + NS - 1
  o = TR
  Tm = np.eye(D)
  if TR == 2: Tm[0, 1] = 1.0
  Tc = Tm.copy()
  n1 = NS - 1
  Tc[o:, o:] = 0.0
  for i in range(n1 - 1): Tc[o + i, o + i + 1] = 1.0
  Tc[o + n1 - 1, o:] = -1.0
  Z = np.zeros(D); Z[0] = 1.0; Z[o] = 1.0
  return D, o, Tm, Tc, Z

def gj_nopivot(M):
  n = M.shape[-1]
  a = M.copy(); inv = np.broadcast_to(np.eye(n, dtype=M.dtype), M.shape).copy()
  for c in range(n):
    rp = (M.dtype.type(1) / a[:, c, c])[:, None]
    a[:, c, :] = a[:, c, :] * rp; inv[:, c, :] = inv[:, c, :] * rp
    for r in range(n):
      if r == c: continue
      f = a[:, r, c][:, None].copy()
      a[:, r, :] = a[:, r, :] - f * a[:, c, :]
      inv[:, r, :] = inv[:, r, :] - f * inv[:, c, :]
  return inv

INV = None

def run(T=10000, TR=1, NS=7, steps_per_season=1, dt=np.float32, seed=0, so=0.45, sl=0.01, ss=0.003, sd=0.002):
  rng = np.random.default_rng(seed)
  D, o, Tm, Tc, Z = model(TR, NS)
  n1 = NS - 1
  ch = np.array([(t + 1) % steps_per_season == 0 for t in range(T)])
  Tpre = int(0.7 * T)
  mask = np.zeros(T, bool); mask[Tpre:] = True; mask[rng.integers(0, Tpre, 20)] = True
  y = np.cumsum(rng.normal(0, 0.05, T)) + np.tile(rng.normal(0, 1, NS), T // NS + 1)[:T] + rng.normal(0, so, T)
  H = so * so
  def Q(c):
    q = np.zeros((D, D)); q[0, 0] = sl * sl
    if TR == 2: q[1, 1] = ss * ss
    if c: q[o:, o:] += (sd / NS) ** 2
    return q
  a1 = np.zeros(D); a1[0] = y[0]
  P1 = np.zeros((D, D)); P1[0, 0] = 1.0
  if TR == 2: P1[1, 1] = 0.01
  P1[o:, o:] = np.eye(n1) - 1.0 / NS
  # ---- float64 sequential filter (filtered moments after each step)
  a, P = a1.copy(), P1.copy()
  mf, Pf = np.zeros((T, D)), np.zeros((T, D, D))
  for t in range(T):
    if t > 0:
      Tt = Tc if ch[t - 1] else Tm
      a = Tt @ a; P = Tt @ P @ Tt.T + Q(ch[t - 1])
    if not mask[t]:
      pz = P @ Z; F = Z @ pz + H
      a = a + pz * (y[t] - Z @ a) / F; P = P - np.outer(pz, pz) / F
    mf[t], Pf[t] = a, P
  # ---- chunk elements in dt
  NT = 256
  Lc = -(-T // NT); Lc = (Lc + 3) & ~3
  A = np.zeros((NT, D, D), dt); b = np.zeros((NT, D), dt); C = np.zeros((NT, D, D), dt)
  eta = np.zeros((NT, D), dt); J = np.zeros((NT, D, D), dt)
  Tm_, Tc_, Z_ = Tm.astype(dt), Tc.astype(dt), Z.astype(dt)
  Hd = dt(H)
  for i in range(NT):
    Ai = np.eye(D, dtype=dt); bi = np.zeros(D, dt); Ci = np.zeros((D, D), dt)
    ei = np.zeros(D, dt); Ji = np.zeros((D, D), dt)
    for l in range(Lc):
      t = i * Lc + l
      if t >= T: break
      if t == 0:
        Ai[:] = 0; bi = a1.astype(dt); Ci = P1.astype(dt)
      else:
        Tt = Tc_ if ch[t - 1] else Tm_
        Ai = Tt @ Ai; bi = Tt @ bi; Ci = Tt @ Ci @ Tt.T + Q(ch[t - 1]).astype(dt)
      if not mask[t]:
        za = Z_ @ Ai; zb = Z_ @ bi; cz = Ci @ Z_; S = Z_ @ cz + Hd
        rS = dt(1) / S
        Ji = Ji + np.outer(za, za) * rS; ei = ei + za * ((dt(y[t]) - zb) * rS)
        K = cz * rS
        Ai = Ai - np.outer(K, za); bi = bi + K * (dt(y[t]) - zb); Ci = Ci - np.outer(K, cz)
        Ci = dt(0.5) * (Ci + Ci.T)
    A[i], b[i], C[i], eta[i], J[i] = Ai, bi, Ci, ei, Ji
  # ---- Hillis-Steele inclusive scan in dt
  I = np.eye(D, dtype=dt)
  def combine(e1, e2):
    A1, b1, C1, n1_, J1 = e1; A2, b2, C2, n2, J2 = e2
    M = I + C1 @ J2
    Mi = (gj_nopivot(M.astype(dt)) if INV == 'nopivot' else np.linalg.inv(M.astype(dt))).astype(dt)
    G = Mi @ A1; A2Mi = A2 @ Mi
    rA = A2 @ G
    rb = np.einsum('nij,nj->ni', A2Mi, b1 + np.einsum('nij,nj->ni', C1, n2)) + b2
    rC = A2Mi @ C1 @ A2.transpose(0, 2, 1) + C2
    re = np.einsum('nji,nj->ni', G, n2 - np.einsum('nij,nj->ni', J2, b1)) + n1_
    rJ = G.transpose(0, 2, 1) @ (J2 @ A1) + J1
    rC = dt(0.5) * (rC + rC.transpose(0, 2, 1)); rJ = dt(0.5) * (rJ + rJ.transpose(0, 2, 1))
    return [x.astype(dt) for x in (rA, rb, rC, re, rJ)]
  e = [A, b, C, eta, J]
  off = 1
  while off < NT:
    e1 = [x[:-off] for x in e]; e2 = [x[off:] for x in e]
    r = combine(e1, e2)
    e = [np.concatenate([x[:off], rx]) for x, rx in zip(e, r)]
    off *= 2
  # inclusive element i gives filtered moments at the end of chunk i
  errm, errP = 0.0, 0.0
  for i in range(NT):
    t = min((i + 1) * Lc, T) - 1
    if i * Lc >= T: break
    sdv = np.sqrt(np.maximum(np.diag(Pf[t]), 1e-12))
    errm = max(errm, np.max(np.abs(e[1][i] - mf[t]) / sdv))
    errP = max(errP, np.max(np.abs(e[2][i] - Pf[t])) / np.max(np.abs(Pf[t])))
  print(f"T={T} TR={TR} NS={NS} sps={steps_per_season} dtype={dt.__name__}: max |mean err|/sd = {errm:.2e}, rel cov err = {errP:.2e}, max|J|={np.max(np.abs(e[4])):.1e}")

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "nopivot":
  INV = 'nopivot'
  for T in (1000, 10000):
    for TR in (1, 2):
      run(T=T, TR=TR)
      run(T=T, TR=TR, so=0.05, sl=0.2, sd=0.1)     # high signal-to-noise: large C J
  run(T=10000, TR=1, steps_per_season=3)
  sys.exit(0)
if __name__ == "__main__":
  for T in (1000, 10000):
    for TR in (1, 2):
      run(T=T, TR=TR)
  run(T=10000, TR=1, steps_per_season=3)
  run(T=10000, TR=1, dt=np.float64)
