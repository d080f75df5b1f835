"""How do the rows fit_causalimpact hands the select kernel look to its first digit?"""
import sys
import numpy as np, pandas as pd
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _synthetic as syn

T, p = 1000, 10
y, X = syn.make_raw_series(T, p, 0)
df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{j}" for j in range(p)])
pre, post = (0, int(0.7 * T) - 1), (int(0.7 * T), T - 1)
from causalimpact import _native
cap = {}
_orig = _native.Session.summarize
def _spy(self, scale, shift, observed, flags, ranks):
  cap.update(traj=self.fetch(["posterior_trajectories"])["posterior_trajectories"], scale=scale, shift=shift,
             obs=np.asarray(observed), flags=np.asarray(flags), ranks=list(ranks))
  return _orig(self, scale, shift, observed, flags, ranks)
_native.Session.summarize = _spy
an = ci.fit_causalimpact(df, pre, post, seed=1, inference_options=ci.InferenceOptions(num_results=1000, num_chains=8),
                         model_options=ci.ModelOptions(local_linear_trend=True))
tr = cap["traj"]
print(tr.shape, tr.dtype, "ranks", cap["ranks"], "scale", cap["scale"], "shift", cap["shift"])
val = (tr.reshape(-1, T).astype(np.float64) * float(np.ravel(cap["scale"])[0]) + float(np.ravel(cap["shift"])[0])).T.copy()   # [T, N]
obs = np.ravel(cap["obs"])[:T]
point = obs[:, None] - val
start = int(np.argmax(np.ravel(cap["flags"])[:T] & 1))
cum = np.cumsum(np.where(np.arange(T)[:, None] >= start, point, 0.0), axis=0)
post = (start, T - 1)
def key(x):
  b = x.view(np.uint64)
  return np.where(b >> np.uint64(63), ~b, b | np.uint64(1 << 63))
for name, M in (("value", val), ("cum", cum[post[0]:])):
  k = key(np.ascontiguousarray(M))
  kor = np.bitwise_or.reduce(k, axis=1); kand = np.bitwise_and.reduce(k, axis=1)
  diff = kor ^ kand
  top = np.array([int(d).bit_length() - 1 for d in diff])
  shift = np.maximum(top + 1 - 11, 0)
  N = M.shape[1]
  ranks = cap['ranks']
  tot, longest = [], []
  for t in range(0, M.shape[0], 7):
    d = ((k[t] >> np.uint64(shift[t])) & np.uint64(2047)).astype(np.int64)
    srt = np.sort(k[t]); bins = set(int((srt[r] >> np.uint64(shift[t])) & np.uint64(2047)) for r in ranks)
    c = np.bincount(d, minlength=2048)
    tot.append(sum(c[b] for b in bins)); longest.append(max(c[b] for b in bins))
  print(name, "top bit", np.percentile(top, [0, 50, 100]), "range/sd", np.percentile((M.max(1) - M.min(1)) / M.std(1), [0, 50, 100]).round(1),
        "total cands", np.percentile(tot, [0, 50, 90, 100]), "longest", np.percentile(longest, [0, 50, 90, 100]),
        "occupied bins", len(np.unique(d)))
