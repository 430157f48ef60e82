#!/bin/bash
# Round 6: A/B of the lattice set-up kernels at 10000^2 (staged Galerkin product, cells per workgroup of lattice_ap_q_kernel)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p
rm -rf $OUT; mkdir -p $OUT
for args in "10000" "10000 fp32" "10000 0.15" "3000 0.15"; do
  timeout 600 python tools/setup_kernels_ab.py $args >> $OUT/setup_kernels_ab.jsonl 2>> $OUT/err.log
done
python - <<'PY'
import json, os
for ln in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6p/setup_kernels_ab.jsonl"):
    d = json.loads(ln)
    print({k: d.get(k) for k in ("size", "holes", "precond", "galerkin", "apq_nt", "setup_device_ms", "iters", "digest", "rc", "err")})
PY
cd /tmp && export TMPDIR=/tmp
CSGPU_APQ_NT=128 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/tools/setup_kernels_ab.py --child 10000 0 0 same > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6p")
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["MaxNs"])
    os.system("cp %s %s/kernel_stats_setup.csv" % (f, out))
PY
find $OUT -name "*.csv" -size +2M -delete
