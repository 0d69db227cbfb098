# kernel trace (start / end of every kernel) of one warm pass: tools/span_timeline.py under rocprofv3 --kernel-trace
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3trace}
mkdir -p $R/gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/span_timeline.py > $R/gpurun_out/$TAG/log.txt 2>&1; echo "rc=$?"
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass = after the last but one 'fps_lazy_kernel' group: find the start of the last pass by the last fps_bbox / first kernel after a long idle gap
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
# find gaps > 300 us (between passes the host synchronises)
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[:i][-50:]) > 300000:
        cut = i
t0 = starts[cut]
out = open("$R/gpurun_out/$TAG/last_pass.txt", "w")
for r in rows[cut:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    out.write("%9.1f %9.1f %7.1f q%-3s %s\n" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
out.close()
print("kernels in last pass:", len(rows) - cut)
PY
