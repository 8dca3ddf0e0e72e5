cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/wide_probe.py 256 > gpurun_out/r04_wide_probe.txt 2>&1; cat gpurun_out/r04_wide_probe.txt | tail -3
rocprofv3 --kernel-trace --stats -d gpurun_out/wk -o run -- python3 scripts/prof_fit.py 2048 256 2 > /dev/null 2> gpurun_out/wk.err
DB=$(find gpurun_out/wk -name "*.db" | head -1); python scripts/rocprof_kernel_stats.py $DB gpurun_out/r04_wide_kstats.csv | head -8; rm -rf gpurun_out/wk
