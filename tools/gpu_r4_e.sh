#!/bin/bash
# Round 4, GPU call E: polygons on the lattice path -- GPU twin, 5000^2 with 50 polygons (lattice path vs merged CSR path vs
# polygon-free), column-position probe of the mixed path, the streaming twin again.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "polygon_rasters_on_the_lattice or streaming" > $OUT/pytest_poly.log 2>&1; grep -v "^csgpu" $OUT/pytest_poly.log | tail -12
PBS=0,4 timeout 600 python tools/polygon_bench.py 5000 50 > $OUT/polygons_5000_lattice.jsonl 2> $OUT/polygons.err; cut -c1-200 $OUT/polygons_5000_lattice.jsonl
python - $OUT/polygons_5000_lattice.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-38s pb%d lat %d setup %.3fs (device %.3f) ms/batch %.1f iters %.2f/%d" % (d["case"][:38], d["precond_bytes"], d["lattice_period"], d["setup_wall_s"], d["setup_device_s"], d["ms_per_batch"], d["iters_mean"], d["iters_max"]))
PY
CSGPU_NO_POLY_LATTICE=1 PBS=0 timeout 300 python tools/polygon_bench.py 5000 50 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  (merged CSR path, same box) ms/batch %.1f iters %.2f setup %.3f' % (d['ms_per_batch'], d['iters_mean'], d['setup_wall_s']))"
BATCH=16 timeout 200 python tools/debug/column_probe.py 1500 > $OUT/column_probe.jsonl 2> $OUT/column_probe.err; cut -c1-600 $OUT/column_probe.jsonl
