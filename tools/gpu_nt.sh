#!/bin/bash
# A/B of non-temporal stores (nts) / stores+loads (ntb) against the default build; variants are swapped in on the box copy.
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/nt
rm -rf $OUT; mkdir -p $OUT
D=$GRAFT_REPO_ROOT/circuitscape.jl_amd
cp $D/libcsgpu.so $D/libcsgpu_plain.so
B="python bench.py --compare-steps 0 --cpu-sample 0 --steps 4 --opt itmax=200"
for v in nts ntb plain; do
cp $D/libcsgpu_$v.so $D/libcsgpu.so
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "level_products or spmv" > $OUT/pytest_$v.log 2>&1; tail -1 $OUT/pytest_$v.log
timeout 300 $B --batch 16 > $OUT/b16_$v.json 2> $OUT/b16_$v.err
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/nt/b*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.2f ms/step %.1f spmm_ms %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["avg_ms"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
cp $D/libcsgpu_nts.so $D/libcsgpu.so
BENCH_ARGS="--opt itmax=200" bash tools/gpu_pmc.sh > $OUT/pmc_stdout.txt 2>&1; grep -E "spmv_kernel<double, 16, 0, true|spmv_kernel<float, 16, 0, true|spmm_longrow_kernel<float, 16, 32" $OUT/pmc_stdout.txt
