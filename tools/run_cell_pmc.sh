# SQ counters of the cell attention kernels on stage 0 of the bench scene (two rocprofv3 --pmc passes + one kernel trace)
# usage (GPU box): bash tools/run_cell_pmc.sh <tag> [stages]
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3pmc}
ST=${2:-0}
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/t -- python3 $R/tools/bench_cell.py 100000 $ST 5 > $R/gpurun_out/$TAG/t.log 2>&1; echo "t rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/$TAG/a -- python3 $R/tools/bench_cell.py 100000 $ST 3 > $R/gpurun_out/$TAG/a.log 2>&1; echo "a rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/$TAG/b -- python3 $R/tools/bench_cell.py 100000 $ST 3 > $R/gpurun_out/$TAG/b.log 2>&1; echo "b rc=$?"
python3 $R/tools/cell_pmc_summary.py $R/gpurun_out/$TAG > $R/gpurun_out/$TAG/summary.txt 2>&1
cat $R/gpurun_out/$TAG/summary.txt
