#!/bin/bash
# Round 4, GPU call D: run-to-run reproducibility probe of the mixed path (batch / stream twice on one handle), polygon
# rasters at 5000^2 through csgpu_raster_setup_poly against the polygon-free raster, the default bench with the new legs.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4d
rm -rf $OUT; mkdir -p $OUT
REPEAT=1 PBS=4,0 BATCHES=16 PAIRS=48 timeout 300 python tools/stream_bench.py 3000 valid > $OUT/repro_3000.jsonl 2> $OUT/repro.err
python - $OUT/repro_3000.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("repro pb%d %-8s identical %s maxreldiff %.2e iters %.2f slots %d" % (d["precond_bytes"], d["mode"], d["identical_to_batch"], d["max_rel_diff_vs_batch"], d["iters_mean"], d["stream_slots"]))
PY
PBS=0,4 timeout 600 python tools/polygon_bench.py 5000 50 > $OUT/polygons_5000.jsonl 2> $OUT/polygons.err; cut -c1-420 $OUT/polygons_5000.jsonl; tail -3 $OUT/polygons.err
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --host-csr 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench value %.2f ms/16 %.1f" % (d["value"], d["ms_per_16_pairs"]))
    for k in ("shortcut", "with_voltages", "config3_fp32", "config4_network"):
        print(k, json.dumps(d.get(k))[:600])
except Exception as e:
    print("bench line missing", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
