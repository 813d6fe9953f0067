cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/time.txt || { tail -20 $O/bench_default.err; exit 1; }
cat $O/time.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04k/bench_default.json"))
print(d["value"], d["ms_per_step"])
print(json.dumps(d["hbm_families"], indent=1))
print(d["eval"]); print(d["seq10"])
print({k:(v.get("value"), v.get("step_frac")) for k,v in d["other_modes"].items()})
print(d["cpu_baseline"]["value"], d["dba"]["value"])
PY
