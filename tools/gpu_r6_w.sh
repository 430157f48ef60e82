#!/bin/bash
# Round 6: set-up kernels with their loads issued up front (lattice_ap_q_kernel, lattice_p_kernel: v1; + lattice_galerkin_kernel: HEAD)
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6w
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in v1 head; do
  if [ $v = v1 ]; then export CSGPU_LIB=$GRAFT_REPO_ROOT/circuitscape.jl_amd/libcsgpu_v1.so; else unset CSGPU_LIB; fi
  for args in "10000 0 0 same" "10000 0.15 0 same" "3001 0.15 0 same" "10000 0 0 fp32"; do
    python $GRAFT_REPO_ROOT/tools/setup_kernels_ab.py --child $args >> $OUT/setup_$v.jsonl 2>> $OUT/err.log
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- python $GRAFT_REPO_ROOT/tools/setup_kernels_ab.py --child 10000 0 0 same > $OUT/trace_$v.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, os, json
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r6w")
for v in ("v1", "head"):
    print(v)
    for ln in open(out + "/setup_%s.jsonl" % v):
        d = json.loads(ln)
        print("  ", {k: (round(d[k], 1) if isinstance(d.get(k), float) else d.get(k)) for k in ("size", "holes", "precond", "setup_ms", "upload_ms", "iters", "digest")})
    for f in glob.glob(out + "/trace_%s/**/*kernel_stats.csv" % v, recursive=True):
        for r in list(csv.DictReader(open(f)))[:40]:
            if any(k in r["Name"] for k in ("ap_q", "galerkin_kernel", "lattice_p_kernel", "dia_build_s", "lattice_sizes")):
                print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["MaxNs"])
        os.system("cp %s %s/kernel_stats_setup_%s.csv" % (f, out, v))
PY
tail -3 $OUT/err.log
find $OUT -name "*.csv" -size +2M -delete
