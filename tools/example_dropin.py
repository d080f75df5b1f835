import sys; sys.path.insert(0, "tfp-causalimpact_amd")
import numpy as np, pandas as pd
import causalimpact
rng = np.random.default_rng(0)
n = 200
x = 100 + np.cumsum(rng.normal(size=n)) * 0.3
y = 1.2 * x + rng.normal(size=n); y[140:] += 5
data = pd.DataFrame({"y": y, "x": x}, index=pd.date_range("2022-01-01", periods=n))
pre_period, post_period = ("2022-01-01", "2022-05-20"), ("2022-05-21", "2022-07-19")
impact = causalimpact.fit_causalimpact(data, pre_period, post_period)
print(impact.series.columns.tolist()[:6], impact.summary.shape, type(impact.posterior_samples.level).__name__, impact.posterior_samples.level.numpy().shape)
print(causalimpact.summary(impact))
impact = causalimpact.fit_causalimpact(
    data, pre_period, post_period,
    inference_options=causalimpact.InferenceOptions(num_results=1000, num_chains=8, devices=[0]),
    model_options=causalimpact.ModelOptions(local_linear_trend=True, seasons=[causalimpact.Seasons(num_seasons=7)]))
print(impact.diagnostics)
res = causalimpact.fit_causalimpact_batch([data, data * 1.0], pre_period, post_period)
print(res.summary.iloc[:, :3]); print(res[1].series.shape)
try:
  causalimpact.plot(impact)
except NotImplementedError as e:
  print("plot:", str(e)[:60])
