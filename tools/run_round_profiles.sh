set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2a/gpu_tests.log 2>&1 ; echo "tests rc=$?"; tail -3 gpurun_out/r2a/gpu_tests.log
P2_FPS_STAMPS=1 timeout -k 10 200 python tools/fps_only.py 100000 2>&1 | tail -3
( time timeout -k 10 500 python bench.py > gpurun_out/r2a/bench_default.log 2>&1 ) 2>&1 | grep real; tail -c 600 gpurun_out/r2a/bench_default.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r2a/pf.log 2>&1; echo "pf rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2a/pw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --in-flight 1 > $GRAFT_REPO_ROOT/gpurun_out/r2a/pw.log 2>&1; echo "pw rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/r2a -name "*.csv" | head -20
# keep only the small summaries
find gpurun_out/r2a/kt -name "*kernel_trace.csv" -delete
