#!/bin/bash
# Round 4, GPU call I: the CSR SpMM's two-halves form at K = 32 (fp64): GPU twin of the K = 32 test, 10000^2 with 15 % NODATA at
# K = 16 / 32 (batch mode), all-valid at K = 32; polygon cases under the final shape rule.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4i
rm -rf $OUT; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batches_of_32" > $OUT/pytest_k32.log 2>&1; tail -3 $OUT/pytest_k32.log
MODES=batch PBS=0,4 BATCHES=16,32 PAIRS=96 timeout 600 python tools/stream_bench.py 10000 holes15 > $OUT/nodata_k16_k32.jsonl 2> $OUT/nodata.err
MODES=batch PBS=0 BATCHES=16,32 PAIRS=96 timeout 600 python tools/stream_bench.py 10000 valid >> $OUT/nodata_k16_k32.jsonl 2>> $OUT/nodata.err
python - $OUT/nodata_k16_k32.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print("  %-8s pb%d K%-2d ms/16 %.1f iters %.2f/%d" % (d["case"], d["precond_bytes"], d["batch"], d["ms_per_16_pairs"], d["iters_mean"], d["iters_max"]))
PY
for CASE in "400 20 rect" "300 60 rect"; do
  set -- $CASE
  POLY_MAX=$2 POLY_SHAPE=$3 PBS=0 timeout 200 python tools/polygon_bench.py 5000 $1 2> /dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  $1 $3 <= $2: lat %d ms/batch %.1f iters %.2f' % (d['lattice_period'], d['ms_per_batch'], d['iters_mean']))"
done
