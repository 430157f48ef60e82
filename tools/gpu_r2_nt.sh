#!/bin/bash
# Non-temporal stores / loads in the marching kernels (-DCSGPU_DIA_NT=1 / 2 builds) against the default build, 10000^2.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2nt
rm -rf $OUT; mkdir -p $OUT
for tag in base nt1 nt2 base_b nt1_b; do
  case $tag in base*) L=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu.so;; nt1*) L=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_nt1.so;; nt2*) L=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_nt2.so;; esac
  CSGPU_LIB=$L timeout 600 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --compare-steps 0 --host-csr 0 > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "pcg_device_ms_per_step", "iters_mean", "max_relres")}, d["roofline"].get("avg_ms"))
PY
done
