#!/bin/bash
# Round 2: setup-time A/B (direct A*T kernel, trusted lattice fill) at the headline size + the new GPU test.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2setup
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x -k "direct_tentative or solve_paths or golden" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
for tag in direct general direct2; do
  if [ $tag = general ]; then export CSGPU_NO_DIRECT_AT=1; else unset CSGPU_NO_DIRECT_AT; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "setup_s", "setup_cold_s", "setup_device_s", "setup_upload_s")},
      d.get("iters_mean"))
PY
done
