# MFMA / VALU / LDS counters of the cell attention kernels (stage 0 and 2 of the bench scene), one rocprofv3 --pmc pass each
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2pmc
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/r2pmc/a -- python3 $R/tools/bench_cell.py 100000 0,2 3 > $R/gpurun_out/r2pmc/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/r2pmc/b -- python3 $R/tools/bench_cell.py 100000 0,2 3 > $R/gpurun_out/r2pmc/b.log 2>&1; echo "b rc=$?"
tail -3 $R/gpurun_out/r2pmc/a.log
