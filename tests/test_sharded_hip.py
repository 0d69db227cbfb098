"""GPU (-m gpu): ONE scene sharded over two ranks, each rank a process of its own running the HIP operators
on the one GPU of the box; collectives over gloo with host staging (RCCL refuses two ranks on one device; on a multi-GPU
node the same code runs with backend "nccl").  Parity target: the N-rank result equals the 1-rank result - integer
tensors bit-identical, fp32 within 1e-3 (the reference has no such path: train.py:105,160-161 is DDP only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.util import window_problem

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene_cfg():
    from stratified_transformer_amd import pipeline
    return pipeline.SceneConfig("two_stage_test", [pipeline.StageConfig(0.16, 0.01, 48, 3, 2), pipeline.StageConfig(0.32, 0.02, 96, 6, 2)],
                                downsample_scale=8, ratio=0.25, k=16, up_k=3, stem_transformer=True)


def _worker(rank, world, port, prob, scene_xyz, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stratified_transformer_amd import pipeline, pointops as P, sharding
        from stratified_transformer_amd.index_build import BlockIndex
        dev = torch.device("cuda", 0)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        # (1) the operators on a shard of a window problem
        block = BlockIndex(t(prob["index_0"]), t(prob["index_1"]), t(prob["offsets"]), None, t(prob["rel_idx"]), None)
        shard, bounds = sharding.make_shard(block, rank, world)
        lo, hi = shard.lo, shard.hi
        q, k, v = (t(prob[x][lo:hi]).clone().requires_grad_(True) for x in ("q", "k", "v"))
        tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
        out = sharding.sharded_window_attention(P, shard, bounds, rank, q, k, v, tq, tk, tv)
        out.backward(t(prob["go_rows"][lo:hi]))
        res = dict(lo=lo, hi=hi, out=out.detach().cpu().numpy(), gq=q.grad.cpu().numpy(), gk=k.grad.cpu().numpy(), gv=v.grad.cpu().numpy(),
                   gtq=tq.grad.cpu().numpy(), gtk=tk.grad.cpu().numpy(), gtv=tv.grad.cpu().numpy())
        # (2) a whole sharded pass: pipeline.scene_pass(shard=(rank, world))
        cfg = _scene_cfg()
        xyz = t(scene_xyz)
        offset = torch.tensor([scene_xyz.shape[0]], dtype=torch.int32, device=dev)
        states, results = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=5, shard=(rank, world))
        torch.cuda.synchronize()
        for si, r in enumerate(results):
            res[f"s{si}_rows"] = np.array(r["out_rows"])
            res[f"s{si}_out"] = r["out"].detach().cpu().numpy()
            res[f"s{si}_ds"] = r["downsample_idx"].cpu().numpy()
            res[f"s{si}_index_1"] = r["odd"].index_1.cpu().numpy()
            res[f"s{si}_gq"] = states[si].q.grad.cpu().numpy()
            res[f"s{si}_gtv"] = states[si].tables[2].grad.cpu().numpy()
        # (3) the same pass with the window-centric kernels on the rank's share of the cells (sharding.sharded_cell_attention)
        states_c, results_c = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=5, fused="cell", shard=(rank, world))
        torch.cuda.synchronize()
        for si, r in enumerate(results_c):
            res[f"c{si}_rows"] = np.array(r["out_rows"])
            res[f"c{si}_out"] = r["out"].detach().cpu().numpy()
            res[f"c{si}_gq"] = states_c[si].q.grad.cpu().numpy()
            res[f"c{si}_gk"] = states_c[si].k.grad.cpu().numpy()
            res[f"c{si}_gtk"] = states_c[si].tables[1].grad.cpu().numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    from stratified_transformer_amd import pipeline, pointops as P, scene
    world = 2
    prob = window_problem(3000, seed=23, h=3, d=16, nbatch=2)
    scene_xyz = scene.make_room(6000, seed=3)
    mp.spawn(_worker, args=(world, _free_port(), prob, scene_xyz, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    # ---- one rank, same kernels ----
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    q, k, v = (t(prob[x]).clone().requires_grad_(True) for x in ("q", "k", "v"))
    tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
    offs, i1, rel = t(prob["offsets"]), t(prob["index_1"]), t(prob["rel_idx"])
    a1 = P.attention_step1_v2(q, k, i1, offs, 0)
    a2 = P.dot_prod_with_idx_v3(q, offs, 0, k, i1, tq, tk, rel)
    out = P.attention_step2_with_rel_pos_value_v2(P.segment_softmax(a1 + a2, offs), v, offs, 0, i1, tv, rel)
    out.backward(t(prob["go_rows"]))
    n = lambda x: x.detach().cpu().numpy()
    assert ranks[0]["lo"] == 0 and ranks[0]["hi"] == ranks[1]["lo"] and ranks[1]["hi"] == prob["N"]
    tol = dict(rtol=1e-4, atol=1e-4)  # bar: 1e-3
    for r in ranks:
        sl = slice(int(r["lo"]), int(r["hi"]))
        np.testing.assert_allclose(r["out"], n(out)[sl], **tol)
        np.testing.assert_allclose(r["gq"], n(q.grad)[sl], **tol)
        np.testing.assert_allclose(r["gk"], n(k.grad)[sl], **tol)
        np.testing.assert_allclose(r["gv"], n(v.grad)[sl], **tol)
        for name, p in (("gtq", tq), ("gtk", tk), ("gtv", tv)):
            np.testing.assert_allclose(r[name], n(p.grad), rtol=5e-4, atol=5e-4)
    # ---- the whole pass ----
    cfg = _scene_cfg()
    xyz = t(scene_xyz)
    offset = torch.tensor([scene_xyz.shape[0]], dtype=torch.int32, device=dev)
    states, results = pipeline.scene_pass(xyz, offset, cfg, None, None, seed=5)
    torch.cuda.synchronize()
    for si, res in enumerate(results):
        full = n(res["out"])
        cover = []
        for r in ranks:
            lo, hi = (int(x) for x in r[f"s{si}_rows"])
            cover.append((lo, hi))
            assert np.array_equal(r[f"s{si}_ds"], n(res["downsample_idx"]))          # integers: bit-identical on every rank
            assert np.array_equal(r[f"s{si}_index_1"], n(res["odd"].index_1))
            np.testing.assert_allclose(r[f"s{si}_out"], full[lo:hi], **tol)
            np.testing.assert_allclose(r[f"s{si}_gq"][lo:hi], n(states[si].q.grad)[lo:hi], **tol)
            np.testing.assert_allclose(r[f"s{si}_gtv"], n(states[si].tables[2].grad), rtol=5e-4, atol=5e-4)
        assert cover[0][0] == 0 and cover[0][1] == cover[1][0] and cover[1][1] == full.shape[0]
        for r in ranks:  # the cell-sharded pass: the same rows, the same numbers up to the order of the sums
            lo, hi = (int(x) for x in r[f"c{si}_rows"])
            np.testing.assert_allclose(r[f"c{si}_out"], full[lo:hi], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(r[f"c{si}_gq"][lo:hi], n(states[si].q.grad)[lo:hi], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(r[f"c{si}_gk"][lo:hi], n(states[si].k.grad)[lo:hi], rtol=2e-4, atol=2e-4)
            np.testing.assert_allclose(r[f"c{si}_gtk"], n(states[si].tables[1].grad), rtol=5e-4, atol=5e-4)
