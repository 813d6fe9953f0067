"""Post-processes a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv` pass over
`bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline --no-alt-modes` into profiles/<round>_pmc_mfma_busy.json:
per kernel the clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the matrix-core busy fraction
(MFMA busy cycles / (1024 SIMDs x cycles)).
usage: python tools/pmc_mfma_busy.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\((anonymous namespace::)?\w*Params\)$|\(.*\)$", "", name).strip()
    a = acc[name]
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[3] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        a[2] += float(r["Counter_Value"])
out = {}
for k, (n, gui, mf, ns) in sorted(acc.items(), key=lambda kv: -kv[1][3]):
    if n == 0 or ns == 0:
        continue
    cyc = gui / 8.0
    out[k] = dict(launches=n, total_ms=ns / 1e6, clock_ghz=cyc / ns, mfma_busy_frac=(mf / (1024.0 * cyc)) if cyc else 0.0)
json.dump(dict(note="rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 1 --warmup 1 "
                    "--single-stream` (3 steps); clock = GRBM_GUI_ACTIVE/8 XCDs/duration (inflated for kernels of a few "
                    "microseconds); mfma_busy_frac = MFMA busy cycles / (1024 SIMDs x cycles)", kernels=out),
          open(sys.argv[2], "w"), indent=1)
print(len(out), "kernels ->", sys.argv[2])
