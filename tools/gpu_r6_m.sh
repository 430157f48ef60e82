#!/bin/bash
# Round 6, GPU call M: (1) single precision (BASELINE configs[3]'s path: T = TP = float): two passes / fused at 512 threads /
# fused at 256 threads; (2) fp32 hierarchy under the fp64 iteration (--precond fp32): the first restriction as a scatter
# (sparse_init) on / off; (3) fp64 once more: two passes with and without the scatter, fused + never-stored r0.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6m
rm -rf $OUT; mkdir -p $OUT
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0"
P=$GRAFT_REPO_ROOT/circuitscape.jl_amd
for rep in 1 2; do
  timeout 600 python bench.py $B --precision single --opt fused_restrict=-1 --opt sparse_init=-1 > $OUT/fp32_twopass_dense_$rep.json 2> $OUT/err
  timeout 600 python bench.py $B --precision single --opt fused_restrict=-1 > $OUT/fp32_twopass_scatter_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --precision single --opt fused_restrict=1 > $OUT/fp32_fused512_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_fnt256.so timeout 600 python bench.py $B --precision single --opt fused_restrict=1 > $OUT/fp32_fused256_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --precond fp32 --opt sparse_init=-1 > $OUT/mixed_dense_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --precond fp32 > $OUT/mixed_scatter_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --opt fused_restrict=-1 --opt sparse_init=-1 > $OUT/fp64_twopass_dense_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --opt fused_restrict=-1 > $OUT/fp64_twopass_scatter_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B > $OUT/fp64_fused_virtual_$rep.json 2>> $OUT/err
done
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6m"
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": round(d["value"], 2), "ms_per_16_pairs": round(d.get("ms_per_16_pairs"), 2), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": round(d.get("pcg_device_ms_per_step"), 1), "max_relres": d.get("max_relres"), "value_job": d.get("value_job"), "dtype": d.get("dtype")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/precision_ab.json", "w"), indent=1)
PY
tail -3 $OUT/err
