cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sqx
export LBDRN_HIP_LIB=lbdrn-msic_amd/liblbdrn_hip_xNONE.so
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d gpurun_out/sqx/a -o run -- python3 scripts/prof_fit.py 2048 64 1 > /dev/null 2> gpurun_out/sqx/a.err
python scripts/pmc_by_kernel.py $(find gpurun_out/sqx/a -name "*.db" | head -1) gpurun_out/sqx/none.csv k_train
unset LBDRN_HIP_LIB
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d gpurun_out/sqx/b -o run -- python3 scripts/prof_fit.py 2048 64 1 > /dev/null 2> gpurun_out/sqx/b.err
python scripts/pmc_by_kernel.py $(find gpurun_out/sqx/b -name "*.db" | head -1) gpurun_out/sqx/base.csv k_train
rm -rf gpurun_out/sqx/a gpurun_out/sqx/b
echo "--- no DMA"; cat gpurun_out/sqx/none.csv; echo "--- base"; cat gpurun_out/sqx/base.csv; tail -3 gpurun_out/sqx/a.err
