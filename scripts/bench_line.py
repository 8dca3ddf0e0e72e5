"""Print the headline fields of bench.py JSON lines.  usage: bench_line.py FILE..."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, c = d.get("roofline", {}), d.get("cpu_baseline", {})
    print(f.split("/")[-1], d["value"], d["unit"], "ms/step", d["ms_per_step_all_repeats"], "single", d.get("single_tile_ms"),
          "in flight", d["config"].get("tiles_in_flight_per_gpu"), "per launch", d["config"].get("tiles_per_launch"),
          "| kernel_us", r.get("kernel_us"), "reduce", r.get("reduce_adam_us"), "rest", r.get("unaccounted_us"),
          "achieved", r.get("achieved"), "frac", r.get("frac"), "apply_ms", r.get("apply_pass_ms"),
          "e2e", r.get("end_to_end_tflops"), r.get("end_to_end_frac"), "| cpu", c.get("value"), c.get("cores"), "| mse", d.get("recon_mse_last_tile"))
