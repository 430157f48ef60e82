#!/bin/bash
# Round 6: set-up kernels without per-cell 64-bit divisions, raster kernels on a column-major copy of the raster: A/B at 10000^2
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6v
rm -rf $OUT; mkdir -p $OUT
for args in "10000 --transpose" "10000 0.15 --transpose" "3001 0.15 --transpose" "10000 fp32 --transpose"; do
  timeout 600 python tools/setup_kernels_ab.py $args >> $OUT/setup_ab.jsonl 2>> $OUT/err.log
done
python - <<'PY'
import json, os
for ln in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6v/setup_ab.jsonl"):
    d = json.loads(ln)
    print({k: (round(d[k], 1) if isinstance(d.get(k), float) else d.get(k)) for k in ("size", "holes", "precond", "raster_transpose", "setup_ms", "upload_ms", "iters", "digest", "rc", "err")})
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/setup_kernels_ab.py --child 10000 0 0 same > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6v")
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:26]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["MaxNs"])
    os.system("cp %s %s/kernel_stats_setup.csv" % (f, out))
PY
find $OUT -name "*.csv" -size +2M -delete
