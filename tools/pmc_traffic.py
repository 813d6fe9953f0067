"""Post-processes the two rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of
`bench.py --steps 1 --warmup 1 --single-stream --no-cpu-baseline`, --output-format csv) into
profiles/<round>_pmc_traffic.json: HBM-side bytes per launch for every kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes (counter unit KiB -> x1024; FETCH_SIZE x2 on gfx950 for wide reads).
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"void |\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\((anonymous namespace::)?\w*Params\)$|\(.*\)$", "", name).strip()
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f = per_kernel(fetch, "FETCH_SIZE")
    w = per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = f.get(k, w.get(k))[0]
        rd = f[k][1] * 1024 * 2 / f[k][0] if k in f else 0.0
        wr = w[k][1] * 1024 / w[k][0] if k in w else 0.0
        kernels[k] = dict(launches=n, read_bytes_per_launch=rd, write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr)
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                        "--warmup 1 --single-stream --no-cpu-baseline` (3 steps incl. the instrumented one); bytes = "
                        "KiB*1024, FETCH x2 (gfx950 wide-read correction)", kernels=kernels), open(out, "w"), indent=1)
    print(len(kernels), "kernels ->", out)


if __name__ == "__main__":
    main()
