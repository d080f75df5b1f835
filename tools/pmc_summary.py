#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes into a small JSON under profiles/.

HBM traffic (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes):

    python tools/pmc_summary.py hbm <fetch_counter_collection.csv> <write_counter_collection.csv> \
        --kernel gibbs_kernel --out profiles/r02_pmc.json [--also latents_kernel ...]

rocprofv3 reports both counters in KiB.  gfx950 correction from the guide: FETCH_SIZE reports
exactly half of a wide (16 B/lane) coalesced streaming read -> doubled here; WRITE_SIZE is
uncalibrated there and is reported as is.

SQ issue counters (one pass):

    python tools/pmc_summary.py sq <counter_collection.csv> --kernel gibbs_kernel --out profiles/...json

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (guide,
"rocprofv3 PMC slots").
"""
import argparse
import collections
import csv
import json


LAST = 0   # --last N: only the last N dispatches of the kernel (e.g. the 8-chain launches of a sweep)


def per_launch(path, kernel_substr):
  per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> sum
  with open(path) as f:
    for row in csv.DictReader(f):
      if kernel_substr in row.get("Kernel_Name", ""):
        per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
  ids = sorted(per)
  if LAST:
    ids = ids[-LAST:]
  tot = collections.defaultdict(float)
  for d in ids:
    for k, v in per[d].items():
      tot[k] += v
  return {k: tot[k] / len(ids) for k in tot}, len(ids)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("mode", choices=("hbm", "sq"))
  ap.add_argument("csvs", nargs="+")
  ap.add_argument("--kernel", required=True)
  ap.add_argument("--also", nargs="*", default=[])
  ap.add_argument("--out", required=True)
  ap.add_argument("--algorithmic-bytes", type=float, default=None)
  ap.add_argument("--last", type=int, default=0)
  args = ap.parse_args()
  global LAST
  LAST = args.last
  if args.mode == "hbm":
    out = {"kernels": {}}
    total = 0.0
    for k in [args.kernel] + list(args.also):
      fetch, nf = per_launch(args.csvs[0], k)
      write, nw = per_launch(args.csvs[1], k)
      f_kb, w_kb = fetch.get("FETCH_SIZE"), write.get("WRITE_SIZE")
      ent = {"launches": {"fetch_pass": nf, "write_pass": nw},
             "FETCH_SIZE_kb_per_launch_raw": f_kb, "WRITE_SIZE_kb_per_launch_raw": w_kb,
             "fetch_bytes_per_launch_corrected": None if f_kb is None else 2.0 * f_kb * 1024.0,
             "write_bytes_per_launch": None if w_kb is None else w_kb * 1024.0}
      if f_kb is not None and w_kb is not None:
        ent["hbm_bytes_per_launch"] = ent["fetch_bytes_per_launch_corrected"] + ent["write_bytes_per_launch"]
        total += ent["hbm_bytes_per_launch"]
      out["kernels"][k] = ent
    out["kernel"] = args.kernel
    out["hbm_bytes_per_launch"] = total
    if args.algorithmic_bytes:
      out["algorithmic_bytes_per_launch"] = args.algorithmic_bytes
      out["traffic_over_algorithmic"] = total / args.algorithmic_bytes
  else:
    c, n = per_launch(args.csvs[0], args.kernel)
    out = {"kernel": args.kernel, "launches": n, "per_launch": c}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
      out["fractions_of_wave_cycles"] = {k: v / wc for k, v in c.items()
                                         if k.startswith(("SQ_WAIT", "SQ_ACTIVE"))}
      if "SQ_INSTS_VALU" in c:
        out["valu_instructions_per_wave_cycle_x4"] = c["SQ_INSTS_VALU"] / wc
  with open(args.out, "w") as f:
    json.dump(out, f, indent=1)
  print(json.dumps(out))


if __name__ == "__main__":
  main()
