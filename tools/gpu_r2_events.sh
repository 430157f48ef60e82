#!/bin/bash
# Do the HIP-event pairs around the CG product cost time themselves? bench at 10000^2 with 512 / 0 / 2 timed launches per solve.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2events
rm -rf $OUT; mkdir -p $OUT
for t in 512 0 2 512; do
  tag=timed$t; [ -f $OUT/$tag.json ] && tag=${tag}_b
  CSGPU_TIMED_LAUNCHES=$t timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "pcg_device_ms_per_step", "iters_mean")}, d["roofline"].get("avg_ms"), d["roofline"].get("launches_timed"))
PY
done
