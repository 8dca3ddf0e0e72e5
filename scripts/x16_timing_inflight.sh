# timing-only library (LBDRN_EXP_X16_TIMING: layer 0 and dW0 of k_train_stream on the 16-bit matrix pipe, results garbage) against the
# shipped one, by fits in flight: is the tile bound by the training launch or by the rest of a chain's step?
for infl in 4 6 8; do for v in base x16t; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  python bench.py --in-flight $infl --steps $((infl*3)) --warmup $infl --repeats 2 --no-cpu-baseline --no-other-configs > /tmp/g.json 2>/dev/null
  python - "$infl" "$v" <<'PY'
import json,sys
d=json.loads(open("/tmp/g.json").read().strip().splitlines()[-1])
print(f"in_flight={sys.argv[1]} {sys.argv[2]:5s}: {d['ms_per_step_all_repeats']} ms/tile  kernel_us {d['roofline']['kernel_us']}", flush=True)
PY
done; done
