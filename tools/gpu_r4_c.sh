#!/bin/bash
# Round 4, GPU call C: streaming pair solves -- GPU twin of the bit-identity test, A/B batch / stream / adaptive at
# 10000^2 (15 % NODATA, all-valid) and 3000^2 (sigma = 3), then the default bench line (batch 32, one call).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or batches_of_32" > $OUT/pytest_stream.log 2>&1; tail -5 $OUT/pytest_stream.log
PBS=0,4 BATCHES=16,32 timeout 600 python tools/stream_bench.py 10000 holes15 > $OUT/stream_holes15_10000.jsonl 2> $OUT/stream_holes.err; cut -c1-330 $OUT/stream_holes15_10000.jsonl; tail -3 $OUT/stream_holes.err
PBS=0 BATCHES=32 timeout 300 python tools/stream_bench.py 10000 valid > $OUT/stream_valid_10000.jsonl 2> $OUT/stream_valid.err; cut -c1-330 $OUT/stream_valid_10000.jsonl
PBS=0 BATCHES=16 PAIRS=64 timeout 400 python tools/stream_bench.py 3000 sigma3 > $OUT/stream_sigma3_3000.jsonl 2> $OUT/stream_sigma3.err; cut -c1-330 $OUT/stream_sigma3_3000.jsonl
timeout 400 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --host-csr 0 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m = d.get("mixed_path", {})
    print("bench batch", d["config"]["batch"], "fp64: value %.2f ms/16 %.1f iters %.2f/%d roof %.3f | mixed: value %.2f ms/16 %.1f | shortcut %.1f volt %.1f | %s"
          % (d["value"], d["ms_per_16_pairs"], d["iters_mean"], d["iters_max"], d["roofline"]["frac"], m.get("value", 0),
             m.get("ms_per_16_pairs", 0), d.get("value_shortcut", 0), d.get("value_with_voltages", 0), d["stream"]))
except Exception as e:
    print("bench line missing", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
