cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 scripts/randperm_probe.py 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d gpurun_out/rp -o run -- python3 scripts/randperm_probe.py > /dev/null 2> gpurun_out/rp.err
DB=$(find gpurun_out/rp -name "*.db" | head -1); python scripts/rocprof_kernel_stats.py $DB gpurun_out/r04_randperm_kstats.csv | head -8; rm -rf gpurun_out/rp
