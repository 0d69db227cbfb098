#!/usr/bin/env python
"""bench.py — points/sec through StratifiedAttention fwd+bwd on a 100k-point scene (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = one pass of the hot path over one synthetic S3DIS-like scene of 100 000 points per GPU
(stratified_transformer_amd/pipeline.py: for each of the 4 stages of s3dis_stratified_transformer.yaml
the index build incl. stratified FPS, depth x [A1,A2,add,A3,A4] forward+backward, TransitionDown FPS +
kNN(16), Upsample kNN(3)).  All inputs are resident in HBM before the timed region.  The K timed passes run with
--in-flight L (default 3-5, by the length of the run) batches in flight: the sampling chains of the next L-1 batches - functions of the
coordinates alone - are queued ahead of this batch's index builds and attention blocks, forward+backward of
consecutive batches strictly in order (pipeline.passes_in_flight); the same K passes one at a time are reported
as `single_batch`, and `same_results_as_single_pass` says that both give the same tensors.  N>1: one process
per GPU, one scene per rank (scenes are independent units: windows never cross a batch element, so the
path shards with no data-path collective) -> weak scaling; value = all ranks' points / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel of the timed region, algorithmic bytes / its live-measured mean
                 launch duration (HIP events on the launch stream) vs the 8 TB/s HBM peak
  cpu_baseline : the oracle (CPU port of the reference kernels, OpenMP) on the same step, same box
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

# Hardware queues of the HIP runtime (read when the runtime initialises, i.e. at the first device call below): with
# the default of 4 the ~15 streams of the lanes share queues and a sampler kernel of one batch blocks unrelated
# kernels of another that happen to sit behind it in the same queue.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 100_000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(op, info):
    """SURVEY.md §8(d) per-op compulsory bytes (fp32/int32), for one launch of `op` at stage `info`."""
    if op.startswith("fps/"):
        # per iteration: read xyz (12 B) + min-dist (4 B) and write min-dist (4 B) for every point
        return 20 * info["n"] * max(info["m"] - 1, 0)
    if op.startswith("knn/"):
        return 12 * info["n"] + 12 * info["m"] + 8 * info["m"] * info["k"]
    N, M, C, h = info["N"], info["M"], info["C"], info["h"]
    if op == "attn_fwd/A1":
        return 8 * N * C + 4 * M + 4 * N + 4 * M * h
    if op == "attn_fwd/A2":
        return 8 * N * C + 16 * M + 4 * N + 4 * M * h
    if op == "attn_fwd/A4":
        return 8 * N * C + 16 * M + 4 * N + 4 * M * h
    if op == "attn_fwd/A3":
        return 8 * M * h + 4 * N
    if op == "attn_fwd/add":
        return 12 * M * h
    if op == "attn_bwd":
        return 44 * N * C + 12 * N + 36 * M + 16 * M * h + 16 * M * h  # + A3 backward (y, gy read; gx write) + add
    return None


def run_gpu(args, rank, world):
    from stratified_transformer_amd import pipeline, scene
    # (the modulo only matters for a rehearsal of the N>1 path with several ranks on a one-GPU box)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    cfg = pipeline.s3dis_config()
    xyz_np = scene.make_room(N_POINTS, seed=rank)
    xyz = torch.from_numpy(xyz_np).to(dev)
    offset = torch.tensor([N_POINTS], dtype=torch.int32, device=dev)

    states, results = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=1234 + rank)  # creates resident tensors

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world > 1:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        return seconds

    # One lane = one set of streams + one set of resident state tensors.  With L lanes the sampling chains (FPS, kNN:
    # functions of the coordinates alone, one workgroup wide for most of their time) of the next L-1 batches are
    # queued in front of the index builds and attention blocks of batch k, the way a training loop prefetches its
    # data-side geometry; the forward+backward of consecutive batches stay strictly in order (an event between
    # them, where the optimizer step would sit).  Same kernels, same results as a pass on its own (checked below).
    HOST_OFFS = [[N_POINTS]]  # (the host copy of the batch offsets a data loader has)
    lanes = []
    for li in range(args.in_flight):
        lane_stream = torch.cuda.Stream(dev)
        lane_stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(lane_stream):
            lane_states, _ = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=1234 + rank, lane=li)
        lanes.append((lane_stream, lane_states))
    barrier()
    if args.warmup > 0:
        pipeline.passes_in_flight([xyz], [offset], cfg, lanes, args.warmup, offset_host_list=HOST_OFFS)
    # In the timed region only the sampler launches (the dominant op, 7 per pass) carry event pairs; the
    # per-op table of all components comes from passes run again afterwards with events around every
    # op (~270 event records per pass are host work, and the late stages are close to host-bound).
    live = pipeline.Timer(True, only=("fps/",))
    barrier()
    t0 = time.perf_counter()
    last = pipeline.passes_in_flight([xyz], [offset], cfg, lanes, args.steps, timer=live, offset_host_list=HOST_OFFS)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # one batch at a time (the latency of a pass on its own), on the default stream
    pipeline.scene_pass(xyz, offset, cfg, states)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        states, results = pipeline.scene_pass(xyz, offset, cfg, states)
    barrier()
    single_elapsed = max_over_ranks(time.perf_counter() - t1)
    same = True
    for lane_results in last:
        if lane_results is None:
            continue
        for a, b in zip(results, lane_results):
            same = same and torch.equal(a["downsample_idx"], b["downsample_idx"]) and torch.equal(a["even"].index_1, b["even"].index_1) \
                and torch.equal(a["odd"].rel_idx, b["odd"].rel_idx) and torch.equal(a["out"], b["out"])
    timer = pipeline.Timer(True)
    for _ in range(args.steps):
        states, results = pipeline.scene_pass(xyz, offset, cfg, states, timer)
    barrier()
    # the optional fused module on the same passes: reported beside the headline, which stays on
    # the reference's operator API
    fused_elapsed = None
    if not args.no_fused:
        pipeline.passes_in_flight([xyz], [offset], cfg, lanes, min(args.warmup, 2) or 1, fused=True, offset_host_list=HOST_OFFS)
        barrier()
        t2 = time.perf_counter()
        pipeline.passes_in_flight([xyz], [offset], cfg, lanes, args.steps, fused=True, offset_host_list=HOST_OFFS)
        barrier()
        fused_elapsed = max_over_ranks(time.perf_counter() - t2)
    return dict(cfg=cfg, xyz_np=xyz_np, states=states, results=results, timer=timer, live=live, elapsed=elapsed, fused_elapsed=fused_elapsed,
                single_elapsed=single_elapsed, inflight_same=bool(same), dev=dev)


def component_table(timer, steps):
    comp = {}
    for name, (ms, calls) in timer.totals().items():
        comp[name] = dict(ms_per_step=ms / steps, calls_per_step=calls / steps, ms_per_call=ms / calls)
    return comp


def roofline(comp, run, among=None):
    """dominant timed op (optionally only among the ops whose name starts with `among`) -> achieved algorithmic GB/s"""
    cfg, results = run["cfg"], run["results"]
    name = max((k for k in comp if among is None or k.startswith(among)), key=lambda k: comp[k]["ms_per_step"])
    r0 = results[0]
    st0 = cfg.stages[0]
    info = dict(N=r0["n"], M=(r0["M_even"] + r0["M_odd"]) // 2, C=st0.channels, h=st0.num_heads)
    note = ""
    if name.startswith("fps/"):
        # per-launch mean over the 3-4 launches per step: weight bytes by every launch of that op
        ns = [r["n"] for r in results]
        if name == "fps/stratified":
            launches = [dict(n=n, m=n // cfg.downsample_scale + 1) for n in ns]
        else:
            launches = [dict(n=n, m=int(n * cfg.ratio) + 1) for n in ns[:-1]]
        bytes_per_launch = float(np.mean([algorithmic_bytes(name, l) for l in launches]))
        note = "mean over the op's launches in one step (one per stage); latency-bound: %d dependent iterations" % sum(l["m"] for l in launches)
    elif name.startswith("knn/") or algorithmic_bytes(name, info) is None:  # (an op without a byte model can only
        bytes_per_launch = None                                              # dominate under a serializing profiler)
    else:
        # attention ops run depth times per stage; bytes are summed over all launches of a step / launches
        tot, cnt = 0, 0
        for si, r in enumerate(results):
            st = cfg.stages[r["stage"]]
            for b in range(st.depth):
                M = r["M_even"] if b % 2 == 0 else r["M_odd"]
                tot += algorithmic_bytes(name, dict(N=r["n"], M=M, C=st.channels, h=st.num_heads))
                cnt += 1
        bytes_per_launch = tot / cnt
        note = "mean over the op's %d launches per step (all stages/blocks)" % cnt
    if bytes_per_launch is None:
        return dict(bound="hbm", kernel=name, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                    note="no byte model for this op (kNN is VALU/latency-bound; see DESIGN.md)")
    dur_s = comp[name]["ms_per_call"] / 1e3
    achieved = bytes_per_launch / dur_s / 1e9
    # HBM-side bytes per launch from the committed PMC passes (tools/pmc_traffic.py: separate FETCH_SIZE and
    # WRITE_SIZE runs of this command, (2 x FETCH_SIZE + WRITE_SIZE) x 1024): the group's bytes per pass over the
    # group's op launches per pass
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        group = name.split("/")[0] if name.split("/")[0] in ("fps", "knn") else name.split("/")[0]
        per_pass = json.load(open(pmc)).get("bytes_per_pass_by_group", {}).get(group)
        launches = sum(c["calls_per_step"] for k, c in comp.items() if k.split("/")[0] == group)
        if per_pass and launches:
            traffic = int(per_pass / launches)
    return dict(bound="hbm", kernel=name, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, algorithmic_bytes_per_launch=int(bytes_per_launch),
                mean_launch_ms=round(comp[name]["ms_per_call"], 4), note=note)


def cpu_baseline(run):
    """The oracle (CPU port of the reference's kernels, OpenMP over all host cores) on one full step of
    the same scene; also the full-size integer parity check of the GPU results."""
    from oracle import index_ref, pointops_ref as ref
    cfg, results, states = run["cfg"], run["results"], run["states"]
    ref.set_num_threads(ref.host_cores(16))  # a one-GPU box owns 16 host cores; OpenMP would otherwise spawn one thread per visible CPU
    cores = ref.num_threads()
    parity = {}
    t_total = 0.0
    xyz = run["xyz_np"]
    offset = np.array([N_POINTS], np.int32)
    for r in results:
        si = r["stage"]
        st = cfg.stages[si]
        state = states[si]
        n = xyz.shape[0]
        t0 = time.perf_counter()
        new_offset = index_ref.stratified_new_offset(offset, cfg.downsample_scale)
        ds = ref.furthestsampling(xyz, offset, new_offset)
        x_t = torch.from_numpy(xyz)
        blocks = [index_ref.build_stage_indices(x_t, offset, st.window_size, st.quant_size, torch.from_numpy(ds), par, "cuda") for par in (0, 1)]
        t_total += time.perf_counter() - t0
        parity[f"stage{si}/fps_stratified"] = bool(np.array_equal(ds, r["downsample_idx"].cpu().numpy()))
        for par, name in ((0, "even"), (1, "odd")):
            g = r[name]
            parity[f"stage{si}/{name}/index_1"] = bool(np.array_equal(blocks[par]["index_1"].numpy(), g.index_1.cpu().numpy()))
            parity[f"stage{si}/{name}/offsets"] = bool(np.array_equal(blocks[par]["offsets"].numpy(), g.offsets.cpu().numpy()))
            parity[f"stage{si}/{name}/rel_idx"] = bool(np.array_equal(blocks[par]["rel_idx"].numpy(), g.rel_idx.cpu().numpy()))
        q, k, v = (t.detach().cpu().numpy() for t in (state.q, state.k, state.v))
        tq, tk, tv = (t.detach().cpu().numpy() for t in state.tables)
        go = state.grad_out.cpu().numpy()
        L = tq.shape[0]
        t0 = time.perf_counter()
        for b in range(st.depth):
            blk = blocks[b % 2]
            i1, offs = blk["index_1"].numpy().astype(np.int32), blk["offsets"].numpy().astype(np.int32)
            rel = np.clip(blk["rel_idx"].numpy(), 0, L - 1).astype(np.int32)
            a1 = ref.attention_step1_v2(q, k, i1, offs)
            a2 = ref.dot_prod_with_idx_v3(q, offs, k, i1, tq, tk, rel)
            s = a1 + a2
            sm = ref.segment_softmax(s, offs)
            out = ref.attention_step2_with_rel_pos_value_v2(sm, v, offs, i1, tv, rel)
            ga, gv, gt = ref.attention_step2_with_rel_pos_value_v2_backward(go, sm, v, offs, i1, tv, rel)
            gs = ref.segment_softmax_backward(sm, ga, offs)
            ref.attention_step1_v2_backward(gs, q, k, i1, offs)
            ref.dot_prod_with_idx_v3_backward(gs, q, offs, k, i1, tq, tk, rel)
        t_total += time.perf_counter() - t0
        if st.depth % 2 == 0:
            pass
        last = r["out"].detach().cpu().numpy()
        parity[f"stage{si}/attention_out_max_abs_err"] = float(np.abs(last - out).max())
        if si < len(cfg.stages) - 1:
            t0 = time.perf_counter()
            n_offset = index_ref.transition_down_offset(offset, cfg.ratio)
            idx = ref.furthestsampling(xyz, offset, n_offset)
            n_xyz = np.ascontiguousarray(xyz[idx])
            kidx, _ = ref.knnquery(cfg.k, xyz, n_xyz, offset, n_offset)
            ref.knnquery(cfg.up_k, n_xyz, xyz, n_offset, offset)  # the Upsample kNN between the two stages
            t_total += time.perf_counter() - t0
            parity[f"stage{si}/transition_knn"] = bool(np.array_equal(kidx, r["transition_knn"].cpu().numpy()))
            xyz, offset = n_xyz, n_offset
    ok = all(v for k_, v in parity.items() if isinstance(v, bool)) and all(v < 1e-3 for v in parity.values() if isinstance(v, float))
    return dict(value=round(N_POINTS / t_total, 1), unit="points/s", cores=cores, kind="port",
                sample="1 full step (all 4 stages of the same 100k-point scene, every op incl. index build), 1 repetition, %.1f s" % t_total,
                seconds=round(t_total, 2)), dict(all_ok=ok, checks=parity)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="skip the fused-module leg (counter passes)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="batches in flight (lanes); 1 = one batch at a time; 0 = by the length of the run: a deeper pipeline "
                         "has a higher steady-state rate (24 steps: 24.0 / 23.0 / 21.8 ms per step with 3 / 4 / 5 lanes) but the "
                         "timed region starts with an empty one, and filling it costs more (5 steps: 24.7 / 25.7 / 26.9 ms)")
    args = ap.parse_args()
    if args.in_flight <= 0:
        args.in_flight = 3 if args.steps < 8 else 4 if args.steps < 16 else 5
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL; BENCH_DIST_BACKEND=gloo only to rehearse the N>1 code path with several ranks on one GPU
        torch.distributed.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    run = run_gpu(args, rank, world)
    if rank == 0:
        ms_per_step = run["elapsed"] / args.steps * 1e3
        comp = component_table(run["timer"], args.steps)
        for name, row in component_table(run["live"], args.steps).items():
            comp[name] = row  # the sampler ops: as measured inside the timed region
        line = {
            "metric": "points/sec through StratifiedAttention fwd+bwd, 100k-pt scene",
            "value": round(N_POINTS * world / (ms_per_step / 1e3), 1), "unit": "points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "one synthetic S3DIS-like room of 100000 points per GPU (BASELINE config 3: fwd+bwd), "
                                   "s3dis_stratified_transformer.yaml stages w=[.16,.32,.64,1.28] C=[48,96,192,384] h=[3,6,12,24] depths=[2,2,6,2]; "
                                   "unit = index build + FPS + depth x (A1,A2,add,A3,A4 fwd+bwd) + TransitionDown FPS/kNN16 + Upsample kNN3 per stage",
                       "points_per_gpu": N_POINTS, "pairs_stage0": run["results"][0]["M_even"],
                       "stage_points": [r["n"] for r in run["results"]], "parallelism": "1 scene per rank, no data-path collective"},
            "overlap": "within a batch: sampling chain, kNN and the next stage's index build on side streams beside the attention blocks; across batches: the sampling chains of the next batches_in_flight-1 batches are queued ahead (data-side prefetch), forward+backward of consecutive batches strictly in order; components_ms_per_step are per-op device times (fps/*: events inside the timed region; the others: single passes repeated with events around every op) and overlap in wall time",
            "batches_in_flight": args.in_flight,
            "same_results_as_single_pass": run["inflight_same"],
            "single_batch": {"ms_per_step": round(run["single_elapsed"] / args.steps * 1e3, 3),
                             "value": round(N_POINTS * world / (run["single_elapsed"] / args.steps), 1),
                             "note": "the same K passes one batch at a time on the default stream (latency of a pass on its own) in this process, i.e. with GPU_MAX_HW_QUEUES=%s; with the runtime's default of 4 hardware queues a pass on its own takes ~38 ms (profiles/)" % os.environ.get("GPU_MAX_HW_QUEUES", "default")},
            "fused_module": None if run["fused_elapsed"] is None else {"ms_per_step": round(run["fused_elapsed"] / args.steps * 1e3, 3),
                             "note": "same timed loop with stratified_transformer_amd.fused.window_attention (fused logits+softmax forward, two-walk backward, one autograd node) instead of the five operators; not the headline"},
            "roofline": roofline(comp, run),
            "roofline_attention": roofline(comp, run, among="attn"),
            "components_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in sorted(comp.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, parity = cpu_baseline(run)
            line["cpu_baseline"] = base
            line["gpu_over_cpu"] = round(line["value"] / base["value"], 2)
            line["parity_at_full_size"] = parity["all_ok"]
            if not parity["all_ok"]:
                line["parity_failures"] = {k: v for k, v in parity["checks"].items() if v is False or (isinstance(v, float) and v >= 1e-3)}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
