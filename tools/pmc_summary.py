#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes) into profiles/r01_pmc.json.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>

Units: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B? No: rocprofv3 reports them in
kilobytes (x1024 bytes).  gfx950 correction from the guide: FETCH_SIZE reports exactly half of
a wide (16 B/lane) coalesced streaming read -> doubled here; WRITE_SIZE is uncalibrated there
and is reported as is.
"""
import csv
import json
import os
import sys


def per_launch(path, counter, kernel_substr="gibbs_kernel"):
  tot, n = 0.0, 0
  with open(path) as f:
    for row in csv.DictReader(f):
      if kernel_substr in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
        tot += float(row["Counter_Value"])
        n += 1
  return (tot / n if n else None), n


def main():
  fetch_csv, write_csv = sys.argv[1], sys.argv[2]
  fetch_kb, nf = per_launch(fetch_csv, "FETCH_SIZE")
  write_kb, nw = per_launch(write_csv, "WRITE_SIZE")
  out = {
      "kernel": "ci::gibbs_kernel<2,4,1,false>",
      "launches": {"fetch_pass": nf, "write_pass": nw},
      "FETCH_SIZE_kb_per_launch_raw": fetch_kb,
      "WRITE_SIZE_kb_per_launch_raw": write_kb,
      "fetch_bytes_per_launch_corrected": None if fetch_kb is None else 2.0 * fetch_kb * 1024.0,
      "write_bytes_per_launch": None if write_kb is None else write_kb * 1024.0,
  }
  if fetch_kb is not None and write_kb is not None:
    out["hbm_bytes_per_launch"] = (out["fetch_bytes_per_launch_corrected"] +
                                   out["write_bytes_per_launch"])
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  with open(os.path.join(root, "profiles", "r01_pmc.json"), "w") as f:
    json.dump(out, f, indent=1)
  print(json.dumps(out))


if __name__ == "__main__":
  main()
