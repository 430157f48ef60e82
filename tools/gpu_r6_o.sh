#!/bin/bash
# Round 6, GPU call O: the fused pass with a thread's update entries on ADJACENT rows (3 (BU + 2) ring reads per column) against
# strided rows (9 BU): fp64 K = 32 and K = 16, fp32 hierarchy under fp64 CG (level-1 pass only); parity test first; kernel
# stats of the adjacent build.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6o
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused_residual or lattice_level1 or ragged" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0"
P=$GRAFT_REPO_ROOT/circuitscape.jl_amd
for rep in 1 2; do
  CSGPU_LIB=$P/libcsgpu_strided.so timeout 600 python bench.py $B > $OUT/fp64_strided_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B > $OUT/fp64_adjacent_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_strided.so timeout 600 python bench.py $B --batch 16 > $OUT/fp64k16_strided_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --batch 16 > $OUT/fp64k16_adjacent_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_strided.so timeout 600 python bench.py $B --precond fp32 > $OUT/mixed_strided_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --precond fp32 > $OUT/mixed_adjacent_$rep.json 2>> $OUT/err
done
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6o"
rows = []
for f in sorted(glob.glob(out + "/*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": round(d["value"], 2), "ms_per_16_pairs": round(d.get("ms_per_16_pairs"), 2), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": round(d.get("pcg_device_ms_per_step"), 1), "max_relres": d.get("max_relres"), "dtype": d.get("dtype")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/fused_rows_ab.json", "w"), indent=1)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); head -9 $f | cut -c1-200; cp $f $OUT/kernel_stats_adjacent.csv
find $OUT -name "*.csv" -size +4M -delete
tail -2 $OUT/err
