"""per-dispatch FETCH_SIZE of bgemm_kernel<2, ...> from a rocprofv3 --pmc FETCH_SIZE csv of tools/bench_bwgrad.py (REPS=2):
MB read per launch (last launch of every shape), in dispatch order.  usage: python tools/bwgrad_traffic.py <counter csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "bgemm_kernel<2" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
vals = [(r["Kernel_Name"].split("bgemm_kernel")[1][:22], r["Grid_Size_X"] if "Grid_Size_X" in r else "", float(r["Counter_Value"]) * 1024 * 2 / 1e6) for r in rows]
per = 3  # 1 warm call + REPS=2
out = [vals[i + per - 1] for i in range(0, len(vals), per)]
for k, g, v in out:
    print(f"{k:24s} {v:8.1f} MB read")
