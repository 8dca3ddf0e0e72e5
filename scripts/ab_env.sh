mkdir -p gpurun_out/fm/e
run() { tag=$1; shift; env "$@" python bench.py --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline > gpurun_out/fm/e/$tag.json 2>>gpurun_out/fm/e/err.log || exit 1; python -c "
import json
d=json.loads(open('gpurun_out/fm/e/$tag.json').read().strip().splitlines()[-1])
print('$tag', d['ms_per_step_all_repeats'], 'single', d['single_tile_ms'], d['roofline']['train_step_pair_us'])"; }
run base A=1
run kernarg16M HSA_KERNARG_POOL_SIZE=16777216
run kernarg256K HSA_KERNARG_POOL_SIZE=262144
run sig256 ROC_SIGNAL_POOL_SIZE=256
run batchsync DEBUG_CLR_BATCH_CPU_SYNC_SIZE=100000 DEBUG_CLR_MAX_BATCH_SIZE=100000
run activewait ROC_ACTIVE_WAIT_TIMEOUT=1000000
run base2 A=1
