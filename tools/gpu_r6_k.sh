#!/bin/bash
# Round 6, GPU call K: shape of the fused residual update + restriction -- threads per workgroup (128 / 256 / 512: coarse rows
# per workgroup 8 / 16 / 32 at K = 32, i.e. 8 halo rows of 32 / 56 / 104 staged) and coarse columns per tile (restrict_seg).
ulimit -c 0
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6k
rm -rf $OUT; mkdir -p $OUT
B="--gpus 1 --steps 10 --warmup 3 --host-csr 0 --extra-legs 0 --pmc-live 0 --cpu-sample 0 --compare-steps 0 --cpu-full-size 0 --opt fused_restrict=1"
P=$GRAFT_REPO_ROOT/circuitscape.jl_amd
for rep in 1 2; do
  timeout 600 python bench.py $B > $OUT/nt256_seg32_$rep.json 2> $OUT/err
  timeout 600 python bench.py $B --opt restrict_seg=64 > $OUT/nt256_seg64_$rep.json 2>> $OUT/err
  timeout 600 python bench.py $B --opt restrict_seg=128 > $OUT/nt256_seg128_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_fnt128.so timeout 600 python bench.py $B > $OUT/nt128_seg32_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_fnt512.so timeout 600 python bench.py $B > $OUT/nt512_seg32_$rep.json 2>> $OUT/err
  CSGPU_LIB=$P/libcsgpu_fnt512.so timeout 600 python bench.py $B --opt restrict_seg=64 > $OUT/nt512_seg64_$rep.json 2>> $OUT/err
done
python - <<'PY'
import json, glob, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6k"
rows = []
for f in sorted(glob.glob(out + "/nt*.json")):
    ln = [l for l in open(f) if l.strip().startswith("{")]
    if not ln:
        print(os.path.basename(f), "NO LINE"); continue
    d = json.loads(ln[-1])
    row = {"file": os.path.basename(f), "value": d["value"], "ms_per_16_pairs": d.get("ms_per_16_pairs"), "iters_mean": d.get("iters_mean"),
           "pcg_device_ms_per_step": d.get("pcg_device_ms_per_step"), "max_relres": d.get("max_relres")}
    rows.append(row); print(row)
json.dump(rows, open(out + "/fused_shape_ab.json", "w"), indent=1)
PY
tail -3 $OUT/err
