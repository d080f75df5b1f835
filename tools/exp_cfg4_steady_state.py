"""How soon does the covariance side of the weekly model's Kalman filter reach its periodic steady
state at cfg4's posterior scales?  (DESIGN.md 3.2 "The periodic steady state of the covariance
side": the review's lever, measured before building it.)  numpy, float64; the recursion of
oracle/ci_oracle.c propagate_cov in the (n-1)-effect coordinates.

  python tools/exp_cfg4_steady_state.py [sigma_obs sigma_level sigma_drift]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  if len(sys.argv) == 4:
    so, sl, sd = (float(v) for v in sys.argv[1:4])
  else:
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "parity_fullsize_cfg4.json")))
    so, sl, sd = (d[k]["device"] for k in ("sigma_obs.mean", "sigma_level.mean", "sigma_drift.mean"))
  n = 7
  D = n
  T = np.eye(D)
  B = np.zeros((n - 1, n - 1))
  B[:-1, 1:] = np.eye(n - 2)
  B[-1, :] = -1.0
  T[1:, 1:] = B
  Q = np.zeros((D, D))
  Q[0, 0] = sl ** 2
  Q[1:, 1:] = (sd / n) ** 2
  z = np.zeros(D)
  z[0] = z[1] = 1.0
  P = np.eye(D)
  P[1:, 1:] = np.eye(n - 1) - 1.0 / n
  hist = []
  for _ in range(12000):
    pz = P @ z
    P = T @ (P - np.outer(pz, pz) / (z @ pz + so ** 2)) @ T.T + Q
    P = 0.5 * (P + P.T)
    hist.append(P)
  print(f"sigma_obs {so:.4g}  sigma_level {sl:.4g}  sigma_drift {sd:.4g}")
  for tol in (1e-3, 1e-4, 1e-5, 1e-6, 1e-7):
    first = next((t for t in range(7, len(hist))
                  if np.abs(hist[t] - hist[t - 7]).max() / np.abs(hist[t]).max() < tol), None)
    print(f"max|P_t - P_(t-7)| / max|P_t| < {tol:g} from t = {first}")


if __name__ == "__main__":
  main()
