"""HBM-side traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same
bench.py command, --output-format csv).  Units and corrections as MI355X_MICROARCH.md (HBM) prescribes: the
counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide (16 B per lane) coalesced reads at
64 B, so the read side is doubled (an upper bound for narrower access shapes).
usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys, collections


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("p2::", "")
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def group_of(name):
    if name.startswith("fps_"):
        return "fps"
    if name.startswith("knn_"):
        return "knn"
    if name.startswith("cell_fwd_kernel") or name.startswith("cell_fwd_mfma_kernel"):
        return "cell_attn_fwd"
    if name.startswith("cell_bwd_kernel") or name.startswith("cell_table_grad"):
        return "cell_attn_bwd"
    for k in ("a1_fwd", "a2_fwd", "a4_fwd", "seg_softmax_fwd"):
        if name.startswith(k):
            return "attn_fwd"
    for k in ("gather_accum", "key_accum", "rows_table_sum", "a4_bwd_attn", "table_grad", "seg_softmax_bwd", "csc_", "csr_expand"):
        if name.startswith(k):
            return "attn_bwd"
    return "other"


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
blocks_per_pass = int(sys.argv[4]) if len(sys.argv) > 4 else 12  # attention blocks of the config (s3dis: 2+2+6+2)
def launches(prefix):
    return sum(v[0] for k, v in fetch.items() if k.startswith(prefix))


# passes of each kind in the profiled process: the operator path launches a1_fwd once per block, the cell module cell_fwd_kernel;
# every pass of either kind launches the grid kNN six times
ops_passes = max(1, launches("a1_fwd_kernel") // blocks_per_pass)
cell_passes = max(1, (launches("cell_fwd_kernel") + launches("cell_fwd_mfma_kernel")) // blocks_per_pass)
all_passes = max(1, (launches("knn_grid_kernel") + launches("knn_lanes_kernel")) // 6)
steps = all_passes
# (the bench also runs forward-only passes: the backward group is normalised by the passes that HAVE a backward)
cell_bwd_passes = max(1, launches("cell_bwd_kernel") // blocks_per_pass)
per = {"attn_fwd": ops_passes, "attn_bwd": ops_passes, "cell_attn_fwd": cell_passes, "cell_attn_bwd": cell_bwd_passes}
kernels, groups = {}, collections.defaultdict(float)
for name in sorted(set(fetch) | set(write)):
    n = max(fetch.get(name, [0, 0])[0], write.get(name, [0, 0])[0])
    rd = 2.0 * fetch.get(name, [0, 0.0])[1] * 1024.0
    wr = write.get(name, [0, 0.0])[1] * 1024.0
    kernels[name] = dict(launches=n, read_bytes_per_launch=rd / max(n, 1), write_bytes_per_launch=wr / max(n, 1))
    groups[group_of(name)] += (rd + wr) / per.get(group_of(name), all_passes)
out = dict(passes_profiled=dict(all=all_passes, operator_api=ops_passes, cell=cell_passes, cell_with_backward=cell_bwd_passes), note="bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, summed over the kernels of a group, per scene pass",
           bytes_per_pass_by_group=dict(groups), kernels=kernels)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["bytes_per_pass_by_group"], indent=1), "passes", steps)
top = sorted(kernels.items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:12]
for k, v in top:
    print("%-60s x%-4d  read %8.2f MB  write %8.2f MB per launch" % (k[:60], v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
