"""per-dispatch FETCH_SIZE of winograd_wgrad_kernel from a rocprofv3 --pmc FETCH_SIZE counter csv of tools/bench_winograd.py
(REPS=2) -> MB read per launch in dispatch order.   usage: python tools/ww_xcd_traffic.py <counter_collection.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "winograd_wgrad_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [float(r["Counter_Value"]) * 1024 * 2 / 1e6 for r in rows]
print("winograd_wgrad_kernel read MB per launch, in dispatch order (l1..l4, 1 + 1 + REPS launches each):", [round(v, 1) for v in vals])
