"""Idle-time analysis of a rocprofv3 --kernel-trace csv (kernel_trace.csv): busy = union of kernel intervals over the
steady-state window; gap histogram between consecutive kernels on the merged timeline.
usage: python tools/timeline_gaps.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip      # steady-state window: the last (1 - skip) of the trace
iv = [x for x in iv if x[0] >= t0]
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _ in iv)
print(f"kernels {len(iv)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms ({100*busy/span:.1f} %)  "
      f"sum of kernel durations {ksum/1e6:.2f} ms (overlap factor {ksum/busy:.2f})")
g = sorted(x[0] for x in gaps)
if g:
    tot = sum(g)
    print(f"idle gaps: {len(g)}  total {tot/1e6:.2f} ms  median {g[len(g)//2]/1e3:.1f} us  p90 {g[int(len(g)*.9)]/1e3:.1f} us  max {g[-1]/1e3:.1f} us")
    big = sorted(gaps, reverse=True)[:12]
    for d, n in big:
        print(f"  {d/1e3:8.1f} us before {n[:90]}")
