#!/bin/bash
# Round 5: (a) large device buffers through one hipMalloc against a VA reservation with 1 GB chunks mapped in (cold-start
# probe); (b) A/B of the fused first-two-sweeps pass of the 25-point levels (CSGPU_DIA25_NO_J0=1 = the two passes it
# replaces); (c) the new device tests.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5h
rm -rf $OUT; mkdir -p $OUT
for args in "26 malloc" "26 vmm 1" "26 vmm 4" "100 vmm 1"; do timeout 120 tools/debug/vmm_probe.bin $args >> $OUT/vmm_probe.jsonl 2>> $OUT/vmm.err; done
cat $OUT/vmm_probe.jsonl
CSGPU_DIA25_NO_J0=1 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 > $OUT/j0_off.jsonl 2> $OUT/err.log
timeout 300 python tools/nodata_iters.py 10000 2468 0.06 > $OUT/j0_on.jsonl 2>> $OUT/err.log
CSGPU_DIA25_NO_J0=1 PB=4 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/j0_off.jsonl 2>> $OUT/err.log
PB=4 timeout 300 python tools/nodata_iters.py 10000 2468 0.06 >> $OUT/j0_on.jsonl 2>> $OUT/err.log
python - <<'PY'
import json,os
for f in ("j0_off","j0_on"):
    for ln in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r5h/%s.jsonl"%f):
        d=json.loads(ln); print(f, "pb %d iters %.2f/%d ms16 %.1f setup %.0f R0 %.15g" % (d["precond_bytes"],d["iters_mean"],d["iters_max"],d["ms_per_16_pairs"],d["setup_device_ms"],d["R0"]))
PY
timeout 900 python -m pytest tests -m gpu -q -x -k "issue341 or enrichment or 25_point or lattice_level1" > $OUT/pytest_subset.log 2>&1; tail -4 $OUT/pytest_subset.log
tail -n 3 $OUT/err.log $OUT/vmm.err
