mkdir -p gpurun_out/r06k
for infl in 4 6 8; do for grp in 1 2 4; do
  LBDRN_FIT_GROUP=$grp python bench.py --bands 4 --in-flight $infl --steps $((infl*3)) --warmup $infl --repeats 2 --no-cpu-baseline --no-other-configs > /tmp/g.json 2>/dev/null
  python - "$infl" "$grp" <<'PY'
import json,sys
d=json.loads(open("/tmp/g.json").read().strip().splitlines()[-1])
print(f"bands4 in_flight={sys.argv[1]} group={sys.argv[2]}: {d['ms_per_step_all_repeats']} ms/tile  equals lone {d['timed_equals_lone']}", flush=True)
PY
done; done
