#!/bin/bash
# Profile pass of a round, run ON the GPU box (gpurun): scripts/profile_round.sh TAG [kt|cfg|pmc|sq ...]
#   kta  the training launch of bc256 / embed / embed pairs alone on the device; ink  in-kernel stamps + timeline JSON
#   b4   the reference's 4-band shape (F = 100): bench.py --bands 4 with four in flight, one fit alone, one pair alone
#   kt   rocprofv3 --kernel-trace --stats of bench.py with one fit in flight and with the default (four, as two pairs),
#        and of one pair of fits alone (scripts/prof_pair.py)
#   cfg  the same for BASELINE.json configs[2] (bc = 256) and configs[4] (USE_COORDINATES + EMBEDDING)
#   pmc  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (--kernel-trace only), per configuration
#        (bc64, bc256, embed, pair = two fits per launch; CFGS="pair" selects) -> per-kernel mean bytes
#   sq   two SQ counter passes (8 counters each) per configuration -> per-kernel means
# Small CSVs land under gpurun_out/prof_TAG/; the databases are deleted as soon as they are summarised.
# (the program itself follows `--`: the profiler's library has initialised the GPU by then)
TAG=${1:-r03}; shift
WHAT=${@:-kt kta cfg pmc sq ink}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
summarise() { # dir out-csv script [filters]
  DB=$(find $1 -name "*.db" | head -1)
  if [ -n "$DB" ]; then python $3 $DB $2 ${@:4} > ${2%.csv}.txt 2>&1; else echo "no database in $1" > ${2%.csv}.txt; fi
  rm -rf $1
}
declare -A PROG=( [bc64]="scripts/prof_fit.py 2048 64 4" [bc256]="scripts/prof_fit.py 2048 256 2" [embed]="scripts/prof_fit.py 2048 64 4 embed" [pair]="scripts/prof_pair.py 2048 3" [bands4]="scripts/prof_fit.py 2048 64 4 bands4" [bands4_pair]="scripts/prof_pair.py 2048 3 bands4" )
CFGS=${CFGS:-bc64 bc256 embed pair}
for W in $WHAT; do case $W in
kt)
  rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o run -- python3 bench.py --no-cpu-baseline --no-other-configs --in-flight 1 --steps 1 --warmup 0 --repeats 1 > $OUT/bench_one_in_flight.json 2> $OUT/kt1.err
  summarise $OUT/kt1 $OUT/kernel_stats_one_in_flight.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kt4 -o run -- python3 bench.py --no-cpu-baseline --no-other-configs --repeats 1 > $OUT/bench_four_in_flight.json 2> $OUT/kt4.err
  summarise $OUT/kt4 $OUT/kernel_stats_four_in_flight.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/ktp -o run -- python3 scripts/prof_pair.py 2048 4 > $OUT/pair_alone.txt 2> $OUT/ktp.err
  summarise $OUT/ktp $OUT/kernel_stats_pair_alone.csv scripts/rocprof_kernel_stats.py ;;
kta)  # the training launch of every configuration ALONE on the device (one chain): what bench.py's rocprof_kernel_us cites
  rocprofv3 --kernel-trace --stats -d $OUT/kta_bc256 -o run -- python3 scripts/prof_fit.py 2048 256 2 > $OUT/alone_bc256.txt 2> $OUT/kta_bc256.err
  summarise $OUT/kta_bc256 $OUT/kernel_stats_alone_bc256.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kta_embed -o run -- python3 scripts/prof_fit.py 2048 64 3 embed > $OUT/alone_embed.txt 2> $OUT/kta_embed.err
  summarise $OUT/kta_embed $OUT/kernel_stats_alone_embed.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kta_embedp -o run -- python3 scripts/prof_pair.py 2048 3 embed > $OUT/pair_alone_embed.txt 2> $OUT/kta_embedp.err
  summarise $OUT/kta_embedp $OUT/kernel_stats_pair_alone_embed.csv scripts/rocprof_kernel_stats.py ;;
b4)
  rocprofv3 --kernel-trace --stats -d $OUT/kt_b4 -o run -- python3 bench.py --no-cpu-baseline --no-other-configs --bands 4 --repeats 1 > $OUT/bench_bands4.json 2> $OUT/kt_b4.err
  summarise $OUT/kt_b4 $OUT/kernel_stats_bands4.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kta_b4 -o run -- python3 scripts/prof_fit.py 2048 64 3 bands4 > $OUT/alone_bands4.txt 2> $OUT/kta_b4.err
  summarise $OUT/kta_b4 $OUT/kernel_stats_alone_bands4.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kta_b4p -o run -- python3 scripts/prof_pair.py 2048 3 bands4 > $OUT/pair_alone_bands4.txt 2> $OUT/kta_b4p.err
  summarise $OUT/kta_b4p $OUT/kernel_stats_pair_alone_bands4.csv scripts/rocprof_kernel_stats.py ;;
ink)  # in-kernel stamps / timeline as JSON (diagnostic builds; scripts/collect_inkernel.py)
  python3 scripts/collect_inkernel.py $TAG > $OUT/inkernel.log 2>&1; echo "ink rc=$?" >> $OUT/status.txt ;;
cfg)
  rocprofv3 --kernel-trace --stats -d $OUT/kt_bc256 -o run -- python3 bench.py --no-cpu-baseline --no-other-configs -bc 256 --in-flight 2 --steps 2 --repeats 1 > $OUT/bench_bc256.json 2> $OUT/kt_bc256.err
  summarise $OUT/kt_bc256 $OUT/kernel_stats_bc256.csv scripts/rocprof_kernel_stats.py
  rocprofv3 --kernel-trace --stats -d $OUT/kt_embed -o run -- python3 bench.py --no-cpu-baseline --no-other-configs --coords-embedding --steps 4 --repeats 1 > $OUT/bench_embed.json 2> $OUT/kt_embed.err
  summarise $OUT/kt_embed $OUT/kernel_stats_embed.csv scripts/rocprof_kernel_stats.py ;;
pmc)
  for CFG in $CFGS; do for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${CFG}_$C -o run -- python3 ${PROG[$CFG]} > /dev/null 2> $OUT/pmc_${CFG}_$C.err; echo "pmc $CFG $C rc=$?" >> $OUT/status.txt
    summarise $OUT/pmc_${CFG}_$C $OUT/pmc_${CFG}_$C.csv scripts/pmc_by_kernel.py k_train k_dw k_reduce k_apply k_build
  done; done ;;
sq)
  for CFG in $CFGS; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES \
      -d $OUT/sq_a_$CFG -o run -- python3 ${PROG[$CFG]} > /dev/null 2> $OUT/sq_a_$CFG.err; echo "sq_a $CFG rc=$?" >> $OUT/status.txt
    summarise $OUT/sq_a_$CFG $OUT/sq_a_$CFG.csv scripts/pmc_by_kernel.py k_train k_dw k_reduce k_apply
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM \
      -d $OUT/sq_b_$CFG -o run -- python3 ${PROG[$CFG]} > /dev/null 2> $OUT/sq_b_$CFG.err; echo "sq_b $CFG rc=$?" >> $OUT/status.txt
    summarise $OUT/sq_b_$CFG $OUT/sq_b_$CFG.csv scripts/pmc_by_kernel.py k_train k_dw k_reduce k_apply
  done ;;
esac; done
ls -la $OUT; cat $OUT/status.txt 2>/dev/null
