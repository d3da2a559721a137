cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
t0=$(date +%s)
python3 bench.py > gpurun_out/r04_bench_full.json 2> gpurun_out/r04_bench_full.err || { tail -20 gpurun_out/r04_bench_full.err; exit 1; }
t1=$(date +%s); echo "wall $((t1 - t0)) s"
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_full.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r.get("traffic"), r.get("traffic_note"), r.get("traffic_measurement"))
PY
