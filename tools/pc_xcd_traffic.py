"""per-dispatch FETCH_SIZE of winograd_pc_kernel from a rocprofv3 --pmc FETCH_SIZE counter csv of tools/pc_xcd_probe.py
(REPS=2: per shape 3 forward + 3 accumulate dispatches, in shape order) -> MB read per launch and shape
usage: python tools/pc_xcd_traffic.py <counter_collection.csv> [reps]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "winograd_pc_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
per = int(sys.argv[2]) + 1 if len(sys.argv) > 2 else 3
vals = [float(r["Counter_Value"]) * 1024 * 2 / 1e6 for r in rows]     # KiB -> bytes, x2 wide-read correction (gfx950)
names = ("l1 64x64 c64", "l2 32x32 c128", "l3 16x16 c256", "l4 8x8 c512")
for i, n in enumerate(names):
    f = vals[i * 2 * per: i * 2 * per + per]
    a = vals[i * 2 * per + per: (i + 1) * 2 * per]
    if f:
        print(f"{n:16s} read MB/launch: fwd {sum(f)/len(f):7.1f}   accumulate {sum(a)/len(a):7.1f}")
