#!/bin/bash
# Round 4, last GPU call: the whole GPU suite, smoke() and the driver's bench command on the closing build.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4o
rm -rf $OUT; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m = d.get("mixed_path", {})
print("value %.2f ms/16 %.1f iters %.2f/%d roof %.3f traffic %s | mixed %.2f ms/16 %.1f roof %.3f | shortcut %.1f volt %.1f fp32 %.1f net %.1f parity %s" % (
    d["value"], d["ms_per_16_pairs"], d["iters_mean"], d["iters_max"], d["roofline"]["frac"], d["roofline"]["traffic"], m["value"], m["ms_per_16_pairs"],
    m["roofline"]["frac"], d["value_shortcut"], d["value_with_voltages"], d["config3_fp32"]["value"], d["config4_network"]["value"], d["parity"]["max_rel_err_vs_oracle"]))
PY
