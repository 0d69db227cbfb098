#!/usr/bin/env python
"""bench.py — points/sec through StratifiedAttention fwd+bwd on a 100k-point scene (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shard]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = one pass of the hot path over one synthetic S3DIS-like scene of 100 000 points (SURVEY.md 8d;
stratified_transformer_amd/pipeline.py): for each of the 4 stages of s3dis_stratified_transformer.yaml the index build
incl. stratified FPS, depth x attention block forward+backward, TransitionDown FPS + kNN(16), Upsample kNN(3).  All
inputs are resident in HBM before the timed region.

What is timed, and what the line calls it:
  single_pass   K passes one after the other, each complete before the next starts (the unit of SURVEY 8d; `value`):
                  .cell          attention blocks = fused.cell_attention (window-centric kernels, csrc/cell_attn.hip)   <- value
                  .operator_api  attention blocks = the reference's five operators (A1, A2, add, A3, A4), what an
                                 unmodified model/stratified_transformer.py calls
  in_flight     the same K passes with --in-flight L (default 4, fixed) batches in flight: the sampling chains of the
                next L-1 batches - functions of the coordinates alone - queued ahead; forward+backward of consecutive
                batches strictly in order (pipeline.passes_in_flight).  A stated throughput configuration, not `value`.
N > 1 ranks (one process per GPU; `--gpus N` without a launcher starts the N ranks itself, before any GPU call):
  default       one scene per rank, no data-path collective: "scaling": "weak", value = N x 100 000 / max-over-ranks time
  --shard       ONE scene over the N ranks (queries sharded by pair count, all-gather of k/v rows, reduce-scatter
                of their gradients, all-reduce of the table gradients over RCCL): "scaling": "strong"

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline      the attention kernels of the timed region (the dominant group once FPS is off the critical path; FPS is
                latency-bound and reported as steps/s in `fps`): SURVEY 8(d)'s algorithmic bytes of every block of a step,
                summed, over the HIP-event time of those blocks inside the timed region, vs the 8 TB/s HBM peak;
                `traffic` = PMC bytes beyond L2 per step for the same kernels (profiles/, tools/pmc_traffic.py)
  cpu_baseline  the oracle (CPU port of the reference kernels, OpenMP) on the same step on this box's host cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# Hardware queues of the HIP runtime (read when the runtime initialises, i.e. at the first device call): with the default
# of 4 the ~15 streams of the in-flight lanes share queues and a sampler kernel of one batch blocks unrelated kernels of
# another that happen to sit behind it in the same queue.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 100_000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
IN_FLIGHT_DEFAULT = 4


# ---------------------------------------------------------------------------------------------------------------------
# unique compulsory bytes of one attention block under the five-operator boundary (fp32 / int32)
# ---------------------------------------------------------------------------------------------------------------------
def attention_bytes(N, M, C, h):
    fwd = 24 * N * C + 12 * N + 36 * M + 12 * M * h      # A1 + A2 + A4 forward
    fwd_glue = 20 * M * h + 8 * M                        # add (12 Mh) + scatter_softmax (8 Mh + 8 M)
    bwd = 44 * N * C + 12 * N + 36 * M + 16 * M * h      # A1 + A2 + A4 backward
    return dict(fwd=fwd, fwd_glue=fwd_glue, bwd=bwd)


def step_attention_bytes(cfg, results):
    tot = dict(fwd=0, fwd_glue=0, bwd=0)
    per_stage = []
    for r in results:
        st = cfg.stages[r["stage"]]
        for b in range(st.depth):
            M = r["M_even"] if b % 2 == 0 else r["M_odd"]
            for k, v in attention_bytes(r["n"], M, st.channels, st.num_heads).items():
                tot[k] += v
        per_stage.append(dict(stage=r["stage"], N=r["n"], M_even=r["M_even"], M_odd=r["M_odd"], C=st.channels, h=st.num_heads, depth=st.depth))
    return tot, per_stage


# ---------------------------------------------------------------------------------------------------------------------
import contextlib
import gc


@contextlib.contextmanager
def no_gc():
    """The host enqueues a pass in ~11 of its ~11.4 ms: a full collection of Python's cyclic garbage collector (tens of ms with the
    process's millions of live objects; due every few dozen passes) inside a timed region shows up as +2-4 ms per pass of a 20-step
    leg and not at all in a 5-step one.  Collected before, switched off inside (as a training loop that cares would)."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run), before this
    process touches the GPU, and return their exit code (non-zero if any rank fails or does not join)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def run_gpu(args, rank, world):
    import numpy as np
    import torch
    from stratified_transformer_amd import pipeline, scene
    local = int(os.environ.get("LOCAL_RANK", 0))
    ndev = torch.cuda.device_count()
    if world > ndev and not os.environ.get("BENCH_ALLOW_SHARED_GPU"):
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPU(s) visible (BENCH_ALLOW_SHARED_GPU=1 rehearses the N>1 path on fewer)")
    dev = torch.device("cuda", local % max(ndev, 1))
    torch.cuda.set_device(dev)
    cfg = pipeline.s3dis_config()
    # --shard: ONE scene over the ranks; ownership by window with halo exchange (sharding.py) unless --shard-mode range
    shard = ((rank, world, "halo") if args.shard_mode == "halo" else (rank, world)) if (args.shard and world > 1) else None
    xyz_np = scene.make_room(N_POINTS, seed=0 if shard else rank)
    xyz = torch.from_numpy(xyz_np).to(dev)
    offset = torch.tensor([N_POINTS], dtype=torch.int32, device=dev)
    HOST_OFFS = [[N_POINTS]]  # (the host copy of the batch offsets a data loader has)

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

    out = dict(cfg=cfg, xyz_np=xyz_np, dev=dev, xyz=xyz, offset=offset)
    # resident synthetic tensors (q/k/v/tables/grad_out stand in for the Linear layers); created once, not timed
    states, results = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=1234 + (0 if shard else rank), fused="cell", shard=shard)

    def summary(res):
        """what the report needs of a pass's results (sizes only)"""
        return [dict(stage=r["stage"], n=r["n"], M_even=r["M_even"], M_odd=r["M_odd"]) for r in res]

    def single_pass_leg(fused, overlap=True):
        # A leg keeps NOTHING of its passes but sizes and event timers: with the tensors of an earlier leg's last pass still alive
        # (pair lists, outputs) the next leg's backward kernels ran 45 % slower for the whole leg - 16.6 instead of 14.2 ms per
        # pass at --steps 20 (tools/leg_trace2.py; no device allocations involved: the same kernels on memory the allocator
        # carved differently).  The tensors the parity checks need come from one more pass after all timed legs.
        nonlocal states
        st, res = states, None
        for _ in range(max(args.warmup, 1)):
            st, res = pipeline.scene_pass(xyz, offset, cfg, st, fused=fused, shard=shard, overlap=overlap)
        live = pipeline.Timer(True, only=("attn", "fps/", "comm/") + (("index/",) if shard else ()))
        if shard:
            from stratified_transformer_amd import sharding
            sharding.reset_bytes()
        barrier()
        from stratified_transformer_amd import _lib as _l
        calls0, allocs0 = _l.CALLS[0], torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0)
        with no_gc():
            t0 = time.perf_counter()
            for _ in range(args.steps):
                st, res = pipeline.scene_pass(xyz, offset, cfg, st, live, fused=fused, shard=shard, overlap=overlap)
            barrier()
            elapsed = max_over_ranks(time.perf_counter() - t0)
        states = st
        leg = dict(elapsed=elapsed, live=live, results=summary(res))
        leg["host"] = dict(library_calls_per_pass=round((_l.CALLS[0] - calls0) / args.steps, 1),
                           torch_allocations_per_pass=round((torch.cuda.memory_stats(dev).get("allocation.all.allocated", 0) - allocs0) / args.steps, 1))
        if shard:
            leg["bytes_moved"] = sharding.BYTES_MOVED
            leg["halo_fraction"] = [r.get("halo_fraction") for r in res]
        return leg

    del results
    out["single_ops"] = single_pass_leg(False)
    # (sharded scene: operator_api = queries cut by range, all-gather k / v; cell = every world-th cell of the size-sorted list per
    #  rank, all-gather q / k / v, reduce-scatter of the output - sharding.py)
    out["single_cell"] = single_pass_leg("cell")
    out["speculation"] = dict(pipeline.SPECULATION)
    if not shard:
        out["single_cell_one_stream"] = single_pass_leg("cell", overlap=False)
        out["single_model"] = single_pass_leg("model")
    if not shard:  # BASELINE configs 2 ("fwd only") and 3 ("fp32 vs bf16") on the same scene
        out["single_fwd"] = single_pass_leg("cell_fwd")
        out["single_bf16"] = single_pass_leg("cell_bf16")

    if not shard:
        out["installed_layers"] = installed_layers_leg(args, cfg, xyz, offset, barrier, max_over_ranks)

    # every component, from passes run again with events around every op (not the timed region)
    timer = pipeline.Timer(True)
    for _ in range(args.steps):
        pipeline.scene_pass(xyz, offset, cfg, states, timer, fused="cell", shard=shard)
    barrier()
    out["timer"] = timer

    if not shard and args.in_flight > 1:
        lanes = []
        for li in range(args.in_flight):
            lane_stream = torch.cuda.Stream(dev)
            lane_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(lane_stream):
                lane_states, _ = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=1234 + rank, lane=li, fused="cell")
            lanes.append((lane_stream, lane_states))
        barrier()

        def in_flight_leg(fused):
            pipeline.passes_in_flight([xyz], [offset], cfg, lanes, max(args.warmup, 1), fused=fused, offset_host_list=HOST_OFFS)
            barrier()
            with no_gc():
                t0 = time.perf_counter()
                last = pipeline.passes_in_flight([xyz], [offset], cfg, lanes, args.steps, fused=fused, offset_host_list=HOST_OFFS)
                barrier()
                return max_over_ranks(time.perf_counter() - t0), last

        out["inflight_cell"], last = in_flight_leg("cell")
        del last
        out["inflight_ops"], last_ops = in_flight_leg(False)
        # the pass the parity checks refer to: one more single pass through the operators, after everything timed
        states, out["results"] = pipeline.scene_pass(xyz, offset, cfg, states, fused=False, shard=shard)
        barrier()
        same = True
        for lane_results in last_ops:
            if lane_results is None:
                continue
            for a, b in zip(out["results"], lane_results):
                same = same and torch.equal(a["downsample_idx"], b["downsample_idx"]) and torch.equal(a["even"].index_1, b["even"].index_1) \
                    and torch.equal(a["odd"].rel_idx, b["odd"].rel_idx) and torch.equal(a["out"], b["out"])
        out["inflight_same"] = bool(same)
    if "results" not in out:
        states, out["results"] = pipeline.scene_pass(xyz, offset, cfg, states, fused=False, shard=shard)
        barrier()

    def capture(res):
        """the last block's output and six gradients of every stage, as host arrays (attention_block clears .grad per block)"""
        return [dict(out=r["out"].detach().cpu().numpy(),
                     grads=[t.grad.detach().float().cpu().numpy() for t in (s_.q, s_.k, s_.v) + tuple(s_.tables)])
                for r, s_ in zip(res, states)]
    out["parity"] = {}
    if not shard:
        # the pass the parity leg checks, per kernel family: the operators (out["results"], made last above) and - one more pass -
        # the cell kernels, i.e. the leg that produces `value`
        out["parity"]["operator_api"] = capture(out["results"])
        states, res_cell = pipeline.scene_pass(xyz, offset, cfg, states, fused="cell", shard=shard)
        barrier()
        out["parity"]["cell"] = capture(res_cell)
        del res_cell
    out["states"] = states
    return out


def installed_layers_leg(args, cfg, xyz, offset, barrier, max_over_ranks):
    """The installed layer forwards under the model's own call order: four stand-in BasicLayers (stratified_transformer_amd.standin:
    the reference's attribute names, call structure and parameter shapes; the reference itself cannot travel to the GPU box) strung
    as Stratified.forward strings them (:470-477), forward + backward on the bench scene.  NOT the metric's unit: a layer also runs
    its qkv / proj / MLP Linear layers, LayerNorms and TransitionDown's grouping + Linear + max-pool, which the unit leaves out."""
    import torch
    from stratified_transformer_amd import layers, standin
    from stratified_transformer_amd import pointops as P
    torch.manual_seed(0)
    dev = xyz.device
    st = cfg.stages
    net = torch.nn.ModuleList([standin.BasicLayer(cfg.downsample_scale, s.depth, s.channels, s.num_heads, s.window_size, s.quant_size, ratio=cfg.ratio, k=cfg.k,
                                                  out_channels=st[i + 1].channels if i + 1 < len(st) else None) for i, s in enumerate(st)]).to(dev)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "relative_pos" in name:
                p.normal_(0.0, 0.02)
    feats0 = torch.randn(xyz.shape[0], st[0].channels, device=dev)

    def step():
        # nothing is carried over from the step before (the same tensor object is fed again): sampler states and what the layers
        # remember about the clouds are dropped, as pipeline.scene_pass does at the start of every pass
        P.clear_caches()
        layers.forget_clouds()
        net.zero_grad(set_to_none=True)
        f, x, o = feats0.clone().requires_grad_(True), xyz, offset
        loss = None
        for layer in net:
            f_out, _, _, f, x, o = layer(f, x, o)
            term = f_out.square().mean()
            loss = term if loss is None else loss + term
        loss.backward()

    res = {}
    chain_was = layers.CHAIN
    try:
        layers.patch_classes(standin.BasicLayer, standin.WindowAttention, standin.TransitionDown)
        for name, chain in (("one_stream", False), ("chained", True)):
            layers.CHAIN = chain
            before = dict(layers.STATS)
            for _ in range(max(args.warmup, 1)):
                step()
            barrier()
            with no_gc():
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                barrier()
                res[name] = max_over_ranks(time.perf_counter() - t0)
            res[name + "_stats"] = {k: layers.STATS[k] - before[k] for k in before}
    finally:
        layers.CHAIN = chain_was
        layers.uninstall_fast_layers()
    return res


def component_table(timer, steps):
    comp = {}
    for name, (ms, calls) in timer.totals().items():
        comp[name] = dict(ms_per_step=ms / steps, calls_per_step=calls / steps, ms_per_call=ms / calls)
    return comp


def attention_roofline(leg, run, steps, label):
    """SURVEY 8(d) bytes of all attention blocks of a step over their HIP-event time inside the timed region."""
    comp = component_table(leg["live"], steps)
    fwd_ms = sum(v["ms_per_step"] for k, v in comp.items() if k.startswith("attn_fwd"))
    bwd_ms = sum(v["ms_per_step"] for k, v in comp.items() if k.startswith("attn_bwd"))
    by, per_stage = step_attention_bytes(run["cfg"], leg["results"])
    fwd_bytes = by["fwd"] + by["fwd_glue"]
    total = fwd_bytes + by["bwd"]
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            groups = json.load(open(pmc)).get("bytes_per_pass_by_group", {})
            key = "cell_attn" if label == "cell" else "attn"
            if key + "_fwd" in groups and key + "_bwd" in groups:
                traffic = int(groups[key + "_fwd"] + groups[key + "_bwd"])
        except (ValueError, OSError):
            traffic = None
    ach = total / ((fwd_ms + bwd_ms) / 1e3) / 1e9 if fwd_ms + bwd_ms > 0 else None
    # the same per stage (the late stages' fractions are much lower than stage 0's: few points, many heads)
    f_by, b_by = leg["live"].totals_by_stage("attn_fwd"), leg["live"].totals_by_stage("attn_bwd")
    stages = []
    for row in per_stage:
        si, cfg_st = row["stage"], run["cfg"].stages[row["stage"]]
        sb = dict(fwd=0, fwd_glue=0, bwd=0)
        for b in range(cfg_st.depth):
            for k_, v_ in attention_bytes(row["N"], row["M_even"] if b % 2 == 0 else row["M_odd"], cfg_st.channels, cfg_st.num_heads).items():
                sb[k_] += v_
        fm, bm = f_by.get(si, (0.0, 0))[0] / steps, b_by.get(si, (0.0, 0))[0] / steps
        stages.append(dict(stage=si, blocks=cfg_st.depth, forward_ms=round(fm, 3), backward_ms=round(bm, 3),
                           forward_ms_per_block=round(fm / cfg_st.depth, 4), backward_ms_per_block=round(bm / cfg_st.depth, 4),
                           forward_frac=round((sb["fwd"] + sb["fwd_glue"]) / (fm / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if fm else None,
                           backward_frac=round(sb["bwd"] / (bm / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if bm else None))
    return dict(bound="hbm", kernel="attention blocks, forward+backward (%s)" % label, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                algorithmic_bytes_per_step=int(total), attention_ms_per_step=round(fwd_ms + bwd_ms, 3), per_stage=stages,
                forward=dict(bytes=int(fwd_bytes), ms=round(fwd_ms, 3), frac=round(fwd_bytes / (fwd_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if fwd_ms else None,
                             formula="sum over blocks of 24NC+12N+36M+12Mh (A1,A2,A4) + 20Mh+8M (add, softmax)"),
                backward=dict(bytes=int(by["bwd"]), ms=round(bwd_ms, 3), frac=round(by["bwd"] / (bwd_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if bwd_ms else None,
                              formula="sum over blocks of 44NC+12N+36M+16Mh"),
                note="bytes: SURVEY 8(d) unique compulsory bytes under the five-operator boundary, every block of every stage of one step; "
                     "time: HIP events around each block's forward and backward on the launch stream, inside the timed region; "
                     "traffic: FETCH_SIZE/WRITE_SIZE passes of the same kernels per step (profiles/pmc_traffic.json), null until collected")


def fps_report(leg, run, steps, xyz, offset):
    comp = component_table(leg["live"], steps)
    ms = sum(v["ms_per_step"] for k, v in comp.items() if k.startswith("fps/"))
    cfg, results = run["cfg"], leg["results"]
    n_steps = sum((r["n"] // cfg.downsample_scale + 1) for r in results) + sum(int(r["n"] * cfg.ratio) + 1 for r in results[:-1])
    # dependent sampling steps that are not an identity prefix verified in parallel: the first stage's n*ratio+1
    seq = int(results[0]["n"] * cfg.ratio) + 1
    evals = sum(float(r["n"]) * (int(r["n"] * cfg.ratio) + 1) for r in results[:-1]) + float(results[-1]["n"]) * (results[-1]["n"] // cfg.downsample_scale + 1)
    out = dict(ms_per_step=round(ms, 3), samples_per_pass=int(n_steps), sequential_steps_per_pass=seq,
               steps_per_s=round(seq / (ms / 1e3), 1) if ms else None,
               reference_distance_evals_per_s=round(evals / (ms / 1e3), 1) if ms else None,
               note="latency-bound (rounds of dependent decisions; up to 16 workgroups per cloud meet at one barrier per round); steps/s = dependent "
                    "sampling steps of stage 0 over the event time of all sampler launches of a pass; the reference's formulation would evaluate n "
                    "distances per step (reference_distance_evals_per_s prices the run at that count); no HBM roofline is claimed for it")
    # the sampler on its own, and on a batch of 8 such scenes in ONE call (the reference's batch dimension: offset / new_offset)
    import torch
    from stratified_transformer_amd import pointops as P

    def timed(x, off, n_off, reps=3):
        best = None
        for _ in range(reps):
            P.clear_caches()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            P.furthestsampling(x, off, n_off)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1)
            best = t if best is None else min(best, t)
        return best
    n = int(xyz.shape[0])
    m = int(n * cfg.ratio) + 1
    one = timed(xyz, offset, torch.tensor([m], dtype=torch.int32, device=xyz.device))
    B = 8
    xb = torch.cat([xyz + torch.tensor([7.0 * i, 0.0, 0.0], device=xyz.device) for i in range(B)]).contiguous()
    ob = torch.tensor([n * (i + 1) for i in range(B)], dtype=torch.int32, device=xyz.device)
    mb = torch.tensor([m * (i + 1) for i in range(B)], dtype=torch.int32, device=xyz.device)
    many = timed(xb, ob, mb)
    out["alone"] = {"scenes": 1, "points": n, "samples": m, "ms": round(one, 3), "samples_per_s": round(m / (one / 1e3), 1)}
    out["batch_of_8"] = {"scenes": B, "points": n * B, "samples": m * B, "ms": round(many, 3), "samples_per_s": round(m * B / (many / 1e3), 1),
                         "note": "8 copies of the scene in one furthestsampling call (one cloud per batch element, 16 workgroups each)"}
    return out


def cpu_baseline(run):
    """The oracle (CPU port of the reference's kernels, OpenMP) on the same step of the same scene: 2 warm-ups + median of 5
    on all host cores of the box, plus a 1-thread figure on a stated sub-sample."""
    import numpy as np
    import torch
    from oracle import index_ref, pointops_ref as ref
    cfg, states = run["cfg"], run["states"]
    cores = ref.host_cores(16)  # a one-GPU box owns 16 host cores; OpenMP would otherwise spawn one thread per visible CPU
    ref.set_num_threads(cores)
    # host copies of the resident tensors and the point cloud of every stage (not CPU-path work: outside the timing)
    host = []
    xyz, offset = run["xyz_np"], np.array([N_POINTS], np.int32)
    for si, st in enumerate(cfg.stages):
        state = states[si]
        row = dict(xyz=xyz, offset=offset, q=state.q.detach().cpu().numpy(), k=state.k.detach().cpu().numpy(), v=state.v.detach().cpu().numpy(),
                   tables=[t.detach().cpu().numpy() for t in state.tables], go=state.grad_out.cpu().numpy())
        host.append(row)
        if si < len(cfg.stages) - 1:
            n_offset = np.asarray(index_ref.transition_down_offset(offset, cfg.ratio), np.int32)
            idx = ref.furthestsampling(xyz, offset, n_offset)
            xyz, offset = np.ascontiguousarray(xyz[idx]), n_offset

    kept = {}  # (first warm-up only) the CPU results of every stage, for the parity check against the GPU pass

    def stage_seconds(si, keep=False):
        """everything the unit does at stage si: stratified FPS, both index patterns, depth x block fwd+bwd, TransitionDown FPS + kNN16, Upsample kNN3"""
        st, hs = cfg.stages[si], host[si]
        xyz, offset = hs["xyz"], hs["offset"]
        t0 = time.perf_counter()
        new_offset = index_ref.stratified_new_offset(offset, cfg.downsample_scale)
        ds = ref.furthestsampling(xyz, offset, new_offset)
        x_t = torch.from_numpy(xyz)
        blocks = [index_ref.build_stage_indices(x_t, offset, st.window_size, st.quant_size, torch.from_numpy(ds), par, "cuda") for par in (0, 1)]
        q, k, v, go = hs["q"], hs["k"], hs["v"], hs["go"]
        tq, tk, tv = hs["tables"]
        L = tq.shape[0]
        for b in range(st.depth):
            blk = blocks[b % 2]
            i1, offs = blk["index_1"].numpy().astype(np.int32), blk["offsets"].numpy().astype(np.int32)
            rel = np.clip(blk["rel_idx"].numpy(), 0, L - 1).astype(np.int32)
            a1 = ref.attention_step1_v2(q, k, i1, offs)
            a2 = ref.dot_prod_with_idx_v3(q, offs, k, i1, tq, tk, rel)
            sm = ref.segment_softmax(a1 + a2, offs)
            o = ref.attention_step2_with_rel_pos_value_v2(sm, v, offs, i1, tv, rel)
            ga, gv, gt = ref.attention_step2_with_rel_pos_value_v2_backward(go, sm, v, offs, i1, tv, rel)
            if keep and b == st.depth - 1:
                kept[si] = dict(ds=ds, out=o, blocks=[(blocks[p_]["index_1"].numpy().astype(np.int32), blocks[p_]["offsets"].numpy().astype(np.int32),
                                                       blocks[p_]["rel_idx"].numpy().astype(np.int32)) for p_ in (0, 1)])  # (unclipped, as the index build emits it)
            gs = ref.segment_softmax_backward(sm, ga, offs)
            gq1, gk1 = ref.attention_step1_v2_backward(gs, q, k, i1, offs)
            gq2, gk2, gtq, gtk = ref.dot_prod_with_idx_v3_backward(gs, q, offs, k, i1, tq, tk, rel)
            if keep and b == st.depth - 1:
                kept[si]["grads"] = [gq1 + gq2, gk1 + gk2, gv, gtq, gtk, gt]   # dq dk dv dTq dTk dTv of the stage's last block
        if si < len(cfg.stages) - 1:
            n_offset = np.asarray(index_ref.transition_down_offset(offset, cfg.ratio), np.int32)
            idx = ref.furthestsampling(xyz, offset, n_offset)
            n_xyz = np.ascontiguousarray(xyz[idx])
            kidx, _ = ref.knnquery(cfg.k, xyz, n_xyz, offset, n_offset)
            ref.knnquery(cfg.up_k, n_xyz, xyz, n_offset, offset)  # the Upsample kNN between the two stages
            if keep and si in kept:
                kept[si]["knn"] = kidx
        return time.perf_counter() - t0

    def step_seconds(stages, keep=False):
        return sum(stage_seconds(si, keep) for si in stages)

    full = range(len(cfg.stages))
    step_seconds(full, keep=True)
    step_seconds(full)
    # parity at full size ("results parity-checked against the GPU output in that run"): every integer tensor of the GPU
    # pass bit for bit, the last block's output of every stage within 1e-3
    worst, differ = 0.0, []

    def same(name, got, want):
        got = got.cpu().numpy()
        if got.shape != want.shape or not np.array_equal(got, want):
            differ.append(name)

    for si, r in enumerate(run["results"]):
        c = kept[si]
        same("stage%d/downsample_idx" % si, r["downsample_idx"], c["ds"])
        for pname, blk, (i1, offs, rel) in zip(("even", "odd"), (r["even"], r["odd"]), c["blocks"]):
            same("stage%d/%s/index_1" % (si, pname), blk.index_1, i1)
            same("stage%d/%s/offsets" % (si, pname), blk.offsets, offs)
            same("stage%d/%s/rel_idx" % (si, pname), blk.rel_idx, rel)
        if "knn" in c and "transition_knn" in r:
            same("stage%d/transition_knn" % si, r["transition_knn"], c["knn"])
        worst = max(worst, float(np.abs(r["out"].detach().cpu().numpy() - c["out"]).max()))
    ints_ok = not differ
    # per kernel family (operators / cell kernels = the `value` leg): the last block's output and its six gradients of every stage
    # against the CPU port.  Outputs: largest absolute difference.  Gradients: largest absolute difference relative to the
    # tensor's largest entry, and where it occurred.
    families = {}
    gnames = ("grad_q", "grad_k", "grad_v", "grad_table_q", "grad_table_k", "grad_table_v")
    for fam, caps in run.get("parity", {}).items():
        o_abs, g_rel, g_abs, where = 0.0, 0.0, 0.0, None
        for si, cap in enumerate(caps):
            c = kept[si]
            o_abs = max(o_abs, float(np.abs(cap["out"] - c["out"]).max()))
            for name, got, want in zip(gnames, cap["grads"], c["grads"]):
                d = float(np.abs(got - want).max())
                rel = d / max(float(np.abs(want).max()), 1e-30)
                g_abs = max(g_abs, d)
                if rel > g_rel:
                    g_rel, where = rel, "stage%d/%s" % (si, name)
        families[fam] = dict(max_abs_output_difference=float("%.3g" % o_abs), max_rel_grad_difference=float("%.3g" % g_rel),
                             max_abs_grad_difference=float("%.3g" % g_abs), worst_gradient=where,
                             within_1e_3=bool(o_abs < 1e-3 and g_rel < 1e-3))
    kept.clear()
    times = sorted(step_seconds(full) for _ in range(5))
    med = times[2]
    # 1-thread figure on a sub-sample: stages 2 and 3 of the same step, all-core time of the same sub-sample beside it
    sub = (2, 3)
    sub_all = sorted(step_seconds(sub) for _ in range(3))[1]
    ref.set_num_threads(1)
    sub_one = step_seconds(sub)
    ref.set_num_threads(cores)
    return dict(value=round(N_POINTS / med, 1), unit="points/s", cores=cores, kind="port",
                parity_at_full_size=bool(ints_ok and worst < 1e-3 and all(f["within_1e_3"] for f in families.values())),
                integers_bit_identical=bool(ints_ok), integer_tensors_that_differ=differ, max_abs_output_difference=float("%.3g" % worst),
                parity_by_kernel_family=families,
                sample="the full step (all 4 stages of the same 100k-point scene, every op incl. index build and FPS): 2 warm-ups, median of 5 "
                       "repetitions on %d OpenMP threads; times %s s" % (cores, [round(t, 2) for t in times]),
                seconds=round(med, 2),
                one_thread=dict(sample="stages 2 and 3 of the same step (%d + %d points)" % (host[2]["xyz"].shape[0], host[3]["xyz"].shape[0]),
                                seconds_1_thread=round(sub_one, 2), seconds_all_cores=round(sub_all, 2), thread_scaling=round(sub_one / sub_all, 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=IN_FLIGHT_DEFAULT, help="batches in flight of the in_flight leg (fixed; 1 = skip the leg)")
    ap.add_argument("--shard", action="store_true", help="N > 1: ONE scene sharded over the ranks (SURVEY 8e) instead of one scene per rank")
    ap.add_argument("--shard-mode", choices=("halo", "range"), default="halo",
                    help="halo: rows owned by large window, only boundary rows travel; range: contiguous index ranges, all-gather of every row")
    args = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))  # no GPU call has been made in this process
    world = int(env_world or 1)
    rank = int(os.environ.get("RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    import torch
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL; BENCH_DIST_BACKEND=gloo only to rehearse the N>1 code path with several ranks on one GPU
        torch.distributed.init_process_group(os.environ.get("BENCH_DIST_BACKEND", "nccl"))
        if torch.distributed.get_world_size() != args.gpus:
            raise SystemExit("bench.py: fewer ranks joined than --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    run = run_gpu(args, rank, world)
    if rank == 0:
        K = args.steps
        sharded = bool(args.shard and world > 1)
        scenes = 1 if sharded else world

        def leg(seconds):
            ms = seconds / K * 1e3
            return dict(ms_per_step=round(ms, 3), value=round(N_POINTS * scenes / (ms / 1e3), 1))

        cell, ops = leg(run["single_cell"]["elapsed"]), leg(run["single_ops"]["elapsed"])
        cell["reached_by"] = ("pipeline.scene_pass: the index build and kernels of stratified_transformer_amd.install(fast_layers=True) (index built once per stage, "
                              "fused.cell_attention per block) under the package's own schedule - sampling chain on a side stream beside the blocks, later stages' "
                              "samples taken as the identity prefix while the sampler verifies them; a training loop that calls the package's pass, not the "
                              "unmodified model file (that one: cell_model_order)")
        cell["speculation"] = run.get("speculation")
        cell["host"] = dict(run["single_cell"]["host"], note="launcher calls of the library (each a handful of kernel launches: ~900 kernels per pass, "
                            "profiles/r03_c_kernel_stats.csv) and caching-allocator allocations per pass; VERDICT r2 #5 asked for <= 120 launches and <= 20 allocations")
        ops["host"] = run["single_ops"]["host"]
        ops["reached_by"] = ("the five operators of the drop-in pointops API on a pair list built once per stage (index_build.stage_index_hip): a caller that "
                             "owns its BasicLayer but keeps the reference's operators")
        extra_legs = {}
        if "single_model" in run:
            extra_legs["model_call_order"] = dict(
                leg(run["single_model"]["elapsed"]),
                reached_by="stratified_transformer_amd.install() alone: the unmodified BasicLayer / WindowAttention drive the operator API",
                note="per block the call sequence of WindowAttention.forward (:183-208) with the model's own tensors: int64 indices, fresh .int() copies per "
                     "operator, rel-pos index by torch ops + two range asserts (host syncs), attn + bias, scatter_softmax shim; the block's pattern is rebuilt "
                     "for every block beyond the first two (:302-317) with the package's device index build - the model's own torch index build "
                     "(grid_sample / get_indice_pairs / sort) is NOT in this number (it lives in the model file and cannot run on the GPU box): a lower bound")
        if "single_cell_one_stream" in run:
            extra_legs["cell_model_order"] = dict(
                leg(run["single_cell_one_stream"]["elapsed"]),
                reached_by="stratified_transformer_amd.install(fast_layers=True): BasicLayer.forward / WindowAttention.forward of the unmodified model file rebound "
                           "to stratified_transformer_amd.layers",
                note="the same pass with every call on ONE stream in the model's order (sampler -> index build -> blocks -> TransitionDown's sampler -> kNN -> "
                     "next stage), which is what the rebound methods of the unmodified model issue: no side stream, nothing speculated")
        if "installed_layers" in run:
            il = run["installed_layers"]
            extra_legs["installed_layers_model"] = dict(
                one_stream_ms=round(il["one_stream"] / K * 1e3, 3), chained_ms=round(il["chained"] / K * 1e3, 3), chained_stats=il["chained_stats"],
                reached_by="stratified_transformer_amd.install(fast_layers=True) on the unmodified model file",
                note="NOT the metric's unit: forward + backward of four whole BasicLayers (stand-in containers with the reference's attribute names, "
                     "call structure and parameter shapes, S3DIS config) strung as Stratified.forward strings them on the same scene - attention "
                     "blocks INCLUDING their qkv / proj / MLP Linear layers and LayerNorms, TransitionDown's grouping + Linear + max-pool.  one_stream: "
                     "every call on the caller's stream in the model's order; chained: the installed forwards put the samplers and the "
                     "TransitionDown geometry on side streams and take later stages' samples as the identity prefix while the sampler verifies them "
                     "(layers.py) - the schedule of single_pass.cell under the unmodified model's call order")
        if "single_fwd" in run:
            extra_legs["cell_forward_only"] = dict(leg(run["single_fwd"]["elapsed"]), note="BASELINE config 2: the same pass without the blocks' backward")
            extra_legs["cell_bf16_storage"] = dict(leg(run["single_bf16"]["elapsed"]), note="BASELINE config 3, second leg: q / k / v / tables stored as bf16, "
                                                   "fp32 arithmetic, outputs and gradient sums (an extension: the reference's operators are fp32-only)")
        comp = component_table(run["timer"], K)
        line = {
            "metric": "points/sec through StratifiedAttention fwd+bwd, 100k-pt scene",
            "value": cell["value"], "unit": "points/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": cell["ms_per_step"],
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "one synthetic S3DIS-like room of 100000 points %s (BASELINE config 3, fp32: fwd+bwd), "
                                   "s3dis_stratified_transformer.yaml stages w=[.16,.32,.64,1.28] C=[48,96,192,384] h=[3,6,12,24] depths=[2,2,6,2]; "
                                   "unit = index build + FPS + depth x attention block fwd+bwd + TransitionDown FPS/kNN16 + Upsample kNN3 per stage; "
                                   "value = single_pass.%s" % ("sharded over the ranks" if sharded else "per GPU", "cell (sharded)" if sharded else "cell"),
                       "points_per_scene": N_POINTS, "stages": step_attention_bytes(run["cfg"], run["results"])[1],
                       "parallelism": ("1 scene over %d ranks: cells dealt to the ranks by size order, all-gather q/k/v, reduce-scatter out and dq/dk/dv, all-reduce table grads "
                                       "(operator_api: queries sharded by pair count, all-gather k/v, reduce-scatter dk/dv)" % world)
                       if sharded else "1 scene per rank, no data-path collective"},
            "single_pass": {"cell": cell, "operator_api": ops, **extra_legs,
                            "note": "K passes, each complete before the next starts; every leg says how a user of the reference reaches it (reached_by)"},
            "roofline": attention_roofline(run["single_cell"], run, K, "cell"),
            "roofline_operator_api": attention_roofline(run["single_ops"], run, K, "operator_api"),
            "fps": fps_report(run["single_cell"], run, K, run["xyz"], run["offset"]),
            "components_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in sorted(comp.items())},
            "components_note": "per-op device times of passes repeated with events around every op (cell attention); chains overlap in wall time",
        }
        if "inflight_cell" in run:
            line["in_flight"] = {"batches_in_flight": args.in_flight, "cell": leg(run["inflight_cell"]), "operator_api": leg(run["inflight_ops"]),
                                 "same_results_as_single_pass": run["inflight_same"],
                                 "note": "the same K passes with the sampling chains (FPS, kNN: functions of the coordinates alone) of the next batches "
                                         "queued ahead, forward+backward of consecutive batches strictly in order; a throughput configuration for a "
                                         "training loop that owns its data-side prefetch, not reachable by the unmodified model file; GPU_MAX_HW_QUEUES=%s"
                                         % os.environ.get("GPU_MAX_HW_QUEUES", "default")}
        if sharded:
            live = component_table(run["single_cell"]["live"], K)
            line["collective_ms"] = round(sum(v["ms_per_step"] for k, v in live.items() if k.startswith("comm/")), 3)
            # what does NOT shrink with the rank count (every rank samples and builds the whole index) and what travels: the
            # Amdahl bound of the strong-scaling leg is in the line.  UNMEASURED on multi-GPU hardware from the build container.
            line["replicated_ms"] = round(sum(v["ms_per_step"] for k, v in live.items() if k.startswith(("fps/", "index/"))), 3)
            line["collective_bytes_per_step"] = {"cell": int(run["single_cell"].get("bytes_moved", 0) // K),
                                                 "operator_api": int(run["single_ops"].get("bytes_moved", 0) // K)}
            line["shard_mode"] = args.shard_mode
            line["halo_fraction_per_stage_even_odd"] = run["single_cell"].get("halo_fraction")
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(run)
            line["cpu_baseline"] = base
            line["gpu_over_cpu"] = round(line["value"] / base["value"], 2)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
