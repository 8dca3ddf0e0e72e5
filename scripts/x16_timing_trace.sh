cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06q; mkdir -p $OUT
for v in base x16t; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o run -- python3 bench.py --no-cpu-baseline --no-other-configs --repeats 1 > $OUT/bench_$v.json 2> $OUT/kt_$v.err
  DB=$(find $OUT/kt_$v -name "*.db" | head -1)
  python scripts/rocprof_kernel_stats.py $DB $OUT/kernel_stats_$v.csv > $OUT/kernel_stats_$v.txt 2>&1
  rm -rf $OUT/kt_$v
  echo "== $v"; head -6 $OUT/kernel_stats_$v.txt | cut -c1-140
  python - "$OUT/bench_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("ms/tile", d["ms_per_step"], "kernel_us", d["roofline"]["kernel_us"], d["accounting"])
PY
done
