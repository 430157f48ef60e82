#!/bin/bash
# near-kernel eigenpair of fp32 hierarchies: gain of the smoothest kept mode (default) vs zero gain (CSGPU_KERNEL_GAIN_ZERO=1)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2deflate2
rm -rf $OUT; mkdir -p $OUT
for s in 1000 5000 10000; do for d in ref zero; do
  if [ $d = zero ]; then export CSGPU_KERNEL_GAIN_ZERO=1; else unset CSGPU_KERNEL_GAIN_ZERO; fi
  st=20; [ $s -ge 5000 ] && st=6
  timeout 600 python bench.py --size $s --steps $st --warmup 2 --cpu-sample 0 --compare-steps $st --host-csr 0 > $OUT/s${s}_$d.json 2> $OUT/s${s}_$d.err
  python - $OUT/s${s}_$d.json $s $d <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("size", sys.argv[2], "kernel gain", sys.argv[3], {k: d.get(k) for k in ("value", "ms_per_step", "iters_mean", "iters_max", "max_relres")}, "fp64:", {k: d["fp64_path"].get(k) for k in ("ms_per_step", "iters_mean", "max_rel_diff_R_vs_mixed_path")})
PY
done; done
