#!/bin/bash
# Round 6, GPU call N: level 1's x = S b and b_c = Q2' b in one pass (csgpu_opts.fused_level1) on / off: fp64, fp32 hierarchy
# under fp64 CG, single precision; device parity test first.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6n
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_residual or lattice_level1" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0"
for rep in 1 2; do
  for F in -1 1; do
    timeout 600 python bench.py $B --opt fused_level1=$F > $OUT/fp64_l1f${F}_$rep.json 2>> $OUT/err
    timeout 600 python bench.py $B --precond fp32 --opt fused_level1=$F > $OUT/mixed_l1f${F}_$rep.json 2>> $OUT/err
    timeout 600 python bench.py $B --precision single --opt fused_level1=$F > $OUT/fp32_l1f${F}_$rep.json 2>> $OUT/err
  done
done
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6n"
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": round(d["value"], 2), "ms_per_16_pairs": round(d.get("ms_per_16_pairs"), 2), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": round(d.get("pcg_device_ms_per_step"), 1), "max_relres": d.get("max_relres"), "dtype": d.get("dtype")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/fused_level1_ab.json", "w"), indent=1)
PY
tail -3 $OUT/err
