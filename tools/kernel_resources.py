"""Tabulates the compiler's resource remarks (build/*.log): registers, spills, scratch, occupancy."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "*.log"
for f in sorted(glob.glob(os.path.join(ROOT, "tfp-causalimpact_amd", "build", pat))):
  rows, cur = [], None
  for line in open(f):
    m = re.search(r"remark:\s+([A-Za-z][\w /\[\]]*): (\S+) \[-Rpass", line)
    if not m:
      continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
      cur = {"name": v}
      rows.append(cur)
    elif cur is not None:
      cur[k] = v
  if not rows:
    continue
  names = subprocess.run(["c++filt"] + [r["name"] for r in rows],
                         capture_output=True, text=True).stdout.split("\n")
  for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n)[:64]
    print(f"{os.path.basename(f):26s} {n:64s} vgpr {r.get('VGPRs'):>3s} agpr {r.get('AGPRs'):>3s} "
          f"scratch {r.get('ScratchSize [bytes/lane]'):>5s} occ {r.get('Occupancy [waves/SIMD]')} "
          f"sgpr-spill {r.get('SGPRs Spill'):>4s} vgpr-spill {r.get('VGPRs Spill'):>4s} "
          f"lds {r.get('LDS Size [bytes/block]')}")
