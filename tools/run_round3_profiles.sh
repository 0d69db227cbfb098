# Round-3 evidence: GPU tests, default bench line, kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace flags).
# usage (GPU box): bash tools/run_round3_profiles.sh <tag>      -> gpurun_out/<tag>/
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r3a}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
( time timeout -k 10 500 python bench.py > $R/gpurun_out/$TAG/bench_default.json 2> $R/gpurun_out/$TAG/bench_default.err ) 2>&1 | grep real
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $R/gpurun_out/$TAG/pf.log 2>&1; echo "pf rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $R/gpurun_out/$TAG/pw.log 2>&1; echo "pw rc=$?"
cd $R
find gpurun_out/$TAG/kt -name "*kernel_trace.csv" -delete
python3 tools/pmc_traffic.py $(find gpurun_out/$TAG/pf -name "*counter_collection.csv" | head -1) $(find gpurun_out/$TAG/pw -name "*counter_collection.csv" | head -1) gpurun_out/$TAG/pmc_traffic.json > gpurun_out/$TAG/pmc_traffic.log 2>&1; tail -3 gpurun_out/$TAG/pmc_traffic.log
head -c 300 gpurun_out/$TAG/bench_default.json
