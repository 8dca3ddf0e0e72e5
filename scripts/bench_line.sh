#!/bin/bash
# One-line summary of a quick bench run per library variant: scripts/bench_line.sh base VARIANT ... (inside one gpurun call:
# BENCH_FLAGS: what replaces "--steps 12 --warmup 4" (e.g. "-bc 256 --in-flight 3 --steps 6 --warmup 3")
# boxes differ by a few percent).  Prints ms per tile with four fits in flight, the lone tile, the step of a fit alone.
for v in "$@"; do
  if [ "$v" = base ]; then unset LBDRN_HIP_LIB; else export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_$v.so; fi
  python bench.py ${BENCH_FLAGS:---steps 12 --warmup 4} --repeats 2 --no-cpu-baseline --no-other-configs > /tmp/bl_$v.json 2>/tmp/bl_$v.err || { tail -5 /tmp/bl_$v.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"/tmp/bl_{v}.json").read().strip().splitlines()[-1]); r = d["roofline"]
print(f"{v:10s} in flight {d['ms_per_step_all_repeats']} ms/tile | lone tile {d.get('single_tile_ms')} ms | kernel_us {r['kernel_us']} reduce {r.get('reduce_adam_us')} step pair {r.get('train_step_pair_us')} | "
      f"single-fit launch {r.get('single_fit_launch', {}).get('kernel_us')} step {r.get('single_fit_launch', {}).get('train_step_pair_us')} | equals lone {d.get('timed_equals_lone')}")
PY
done
