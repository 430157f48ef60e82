#!/bin/bash
# Round 6, GPU call J: the fused residual update + restriction (csgpu_opts.fused_restrict = 1, lattice.h) against the two-pass
# path on one box: device parity test, the bench's headline leg both ways (interleaved, twice), 16-column batches, and the
# rocprofv3 kernel stats of the fused run.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6j
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_residual or stream" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0"
for rep in 1 2; do
  for F in -1 1; do
    timeout 600 python bench.py $B --opt fused_restrict=$F > $OUT/bench_f${F}_$rep.json 2> $OUT/bench_f${F}_$rep.err
    timeout 600 python bench.py $B --batch 16 --opt fused_restrict=$F > $OUT/bench16_f${F}_$rep.json 2>> $OUT/bench_f${F}_$rep.err
    timeout 600 python bench.py $B --calls per-step --opt fused_restrict=$F > $OUT/benchps_f${F}_$rep.json 2>> $OUT/bench_f${F}_$rep.err
  done
done
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6j"
rows = []
for f in sorted(glob.glob(out + "/bench*_f*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": d["value"], "ms_per_16_pairs": d.get("ms_per_16_pairs"), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": d.get("pcg_device_ms_per_step"), "max_relres": d.get("max_relres"), "batch": d["config"]["batch"], "calls": d.get("calls")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/fused_ab.json", "w"), indent=1)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0 --opt fused_restrict=1 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -8 $f; cp $f $OUT/kernel_stats_fused.csv
find $OUT -name "*.csv" -size +4M -delete
