#!/bin/bash
# Round 5: the driver's bench command with the new legs at their real sizes (network 5e6 nodes, live PMC passes), wall clock.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5k
rm -rf $OUT; mkdir -p $OUT
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "wall $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5k/bench.json").read().strip().splitlines()[-1])
print("value %.2f ms/16 %.1f iters %.2f roof %.3f" % (d["value"], d["ms_per_16_pairs"], d["iters_mean"], d["roofline"]["frac"]))
print("traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source"), d["roofline"].get("traffic_live_failed"))
for k in ("nodata15","config3_fp32","config4_network","network_geometric"):
    v=d.get(k,{}); print(k, {kk: v.get(kk) for kk in ("value","ms_per_16_pairs","iters_mean","iters_max","enrich_vectors","levels","setup_s","solve_s_all_sources","generate_s","failed")}, "PARITY", v.get("parity"))
print("stream", d["nodata15"].get("stream"), "leg_seconds", d["leg_seconds"])
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","measured_full_size")})
print("parity", d["parity"])
PY
tail -n 5 $OUT/bench.err
