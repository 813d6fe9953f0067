"""Multi-stream timeline analysis of a rocprofv3 --kernel-trace csv of bench.py: finds the last full step (between two
adamw_kernel launches), then reports wall time, union-busy time, the concurrency histogram (how long k kernels were
resident at once), per-queue busy time and launch counts, per-queue idle gaps, and the time per kernel family.
usage: python tools/timeline_streams.py <kernel_trace.csv> [step_index_from_end=1]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]) for r in rows))
ad = [i for i, e in enumerate(ev) if "adamw_kernel" in e[3]]
lo, hi = ad[-1 - back], ad[-back]
step = ev[lo + 1:hi + 1]
t0, t1 = ev[lo][1], ev[hi][1]
wall = t1 - t0
print(f"step: {len(step)} launches, wall {wall/1e6:.2f} ms (end of previous adamw -> end of adamw)")
# union + concurrency
pts = []
for s, e, q, n in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
conc = collections.Counter(); cur = 0; last = t0
for t, d in pts:
    conc[cur] += t - last; last = t; cur += d
conc[cur] += t1 - last
tot = sum(conc.values())
print("concurrency (kernels resident): " + "  ".join(f"{k}:{v/1e6:.2f}ms({100*v/tot:.0f}%)" for k, v in sorted(conc.items())))
ksum = sum(e - s for s, e, _, _ in step)
print(f"kernel-duration sum {ksum/1e6:.2f} ms, union busy {(tot-conc[0])/1e6:.2f} ms, idle {conc[0]/1e6:.2f} ms")
byq = collections.defaultdict(list)
for x in step: byq[x[2]].append(x)
for q, l in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in l)
    gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print(f"  queue {q}: {len(l):5d} launches, busy {busy/1e6:6.2f} ms, span {(l[-1][1]-l[0][0])/1e6:6.2f} ms, "
          f"gaps<20us: n={len(small)} sum {sum(small)/1e6:.2f} ms median {sorted(small)[len(small)//2]/1e3 if small else 0:.1f} us")
def fam(n):
    n = re.sub(r"^void ", "", n); n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(_ZN12_GLOBAL__N_1\d+)?([A-Za-z_0-9]+)", n)
    if n.startswith("_ZN"):
        m2 = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(I|E)", n); return m2.group(1) if m2 else n[:30]
    return re.split(r"[<(]", n)[0]
f = collections.defaultdict(lambda: [0, 0])
for s, e, q, n in step:
    k = fam(n); f[k][0] += 1; f[k][1] += e - s
for k, (c, d) in sorted(f.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:36s} {c:5d} launches {d/1e6:7.2f} ms  avg {d/c/1e3:6.1f} us")
