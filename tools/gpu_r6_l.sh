#!/bin/bash
# Round 6, GPU call L: the batch's right-hand side never stored (csgpu_opts.sparse_init) on top of the fused residual update +
# restriction (512 threads, 64 coarse columns per tile): device parity test, then the bench's headline leg three ways on one
# box, interleaved, twice: two passes / fused / fused + sparse_init.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6l
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_residual or ragged or stream" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0"
for rep in 1 2; do
  timeout 600 python bench.py $B --opt fused_restrict=-1 > $OUT/twopass_$rep.json 2> $OUT/err
  timeout 600 python bench.py $B --opt sparse_init=-1 > $OUT/fused_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B > $OUT/fused_sparse_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --batch 16 --opt fused_restrict=-1 > $OUT/k16_twopass_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --batch 16 > $OUT/k16_fused_sparse_$rep.json 2>> $OUT/err
done
timeout 600 python bench.py $B --precision single --opt fused_restrict=-1 > $OUT/fp32_twopass.json 2>> $OUT/err
timeout 600 python bench.py $B --precision single > $OUT/fp32_fused_sparse.json 2>> $OUT/err
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6l"
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": d["value"], "ms_per_16_pairs": d.get("ms_per_16_pairs"), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": d.get("pcg_device_ms_per_step"), "max_relres": d.get("max_relres"), "value_job": d.get("value_job"),
           "job_100_pairs_s": d.get("job_100_pairs_s"), "dtype": d.get("dtype")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/sparse_init_ab.json", "w"), indent=1)
PY
tail -3 $OUT/err
