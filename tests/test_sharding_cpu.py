"""CPU, world_size 2, gloo: the sharded window attention (stratified_transformer_amd/sharding.py) equals the
unsharded one — forward rows, q/k/v gradient rows, table gradients.  Local compute = the oracle ops."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import window_problem


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, prob, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from oracle import pointops_ref
        pointops_ref.set_num_threads(2)
        from stratified_transformer_amd import sharding
        from stratified_transformer_amd.index_build import BlockIndex
        from tests import oracle_ops
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        block = BlockIndex(t(prob["index_0"]), t(prob["index_1"]), t(prob["offsets"]), None, t(prob["rel_idx"]), None)
        shard, bounds = sharding.make_shard(block, rank, world)
        lo, hi = shard.lo, shard.hi
        q, k, v = (t(prob[x][lo:hi]).clone().requires_grad_(True) for x in ("q", "k", "v"))
        tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
        out = sharding.sharded_window_attention(oracle_ops, shard, bounds, rank, q, k, v, tq, tk, tv)
        out.backward(t(prob["go_rows"][lo:hi]))
        ret[rank] = dict(lo=lo, hi=hi, bounds=bounds, out=out.detach().numpy(), gq=q.grad.numpy(), gk=k.grad.numpy(), gv=v.grad.numpy(),
                         gtq=tq.grad.numpy(), gtk=tk.grad.numpy(), gtv=tv.grad.numpy(), pairs=shard.pair_hi - shard.pair_lo)
    finally:
        dist.destroy_process_group()


def test_sharded_attention_equals_unsharded_world2():
    prob = window_problem(1500, seed=21, h=3, d=16, nbatch=2)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), prob, ret), nprocs=world, join=True)
    # unsharded reference (single process, same ops)
    from tests import oracle_ops as O
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    q, k, v = (t(prob[x]).clone().requires_grad_(True) for x in ("q", "k", "v"))
    tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
    offs, i1, rel = t(prob["offsets"]), t(prob["index_1"]), t(prob["rel_idx"])
    a1 = O.attention_step1_v2(q, k, i1, offs, 0)
    a2 = O.dot_prod_with_idx_v3(q, offs, 0, k, i1, tq, tk, rel)
    out = O.attention_step2_with_rel_pos_value_v2(O.segment_softmax(a1 + a2, offs), v, offs, 0, i1, tv, rel)
    out.backward(t(prob["go_rows"]))
    r0, r1 = ret[0], ret[1]
    assert r0["bounds"] == r1["bounds"] and r0["hi"] == r1["lo"] and r0["lo"] == 0 and r1["hi"] == prob["N"]
    # the cut balances PAIRS, not points
    assert abs(r0["pairs"] - r1["pairs"]) < 0.1 * prob["M"]
    tol = dict(rtol=1e-5, atol=1e-5)
    for r in (r0, r1):
        sl = slice(r["lo"], r["hi"])
        np.testing.assert_allclose(r["out"], out.detach().numpy()[sl], **tol)
        np.testing.assert_allclose(r["gq"], q.grad.numpy()[sl], **tol)
        np.testing.assert_allclose(r["gk"], k.grad.numpy()[sl], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r["gv"], v.grad.numpy()[sl], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r["gtq"], tq.grad.numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(r["gtk"], tk.grad.numpy(), rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(r["gtv"], tv.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_balanced_bounds_edge_cases():
    from stratified_transformer_amd import sharding
    offs = torch.tensor([0, 0, 10, 10, 11, 30], dtype=torch.int32)
    assert sharding.balanced_bounds(offs, 1) == [0, 5]
    b = sharding.balanced_bounds(offs, 2)
    assert b[0] == 0 and b[-1] == 5 and b == sorted(b)
    b8 = sharding.balanced_bounds(offs, 8)   # more ranks than rows with pairs: empty shards are fine
    assert len(b8) == 9 and b8 == sorted(b8) and b8[-1] == 5


def _halo_worker(rank, world, port, prob, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from oracle import pointops_ref
        pointops_ref.set_num_threads(2)
        from stratified_transformer_amd import index_build, sharding
        from stratified_transformer_amd.index_build import BlockIndex
        from tests import oracle_ops
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        block = BlockIndex(t(prob["index_0"]), t(prob["index_1"]), t(prob["offsets"]), None, t(prob["rel_idx"]), None)
        parts = index_build.stage_partitions(t(prob["xyz"]), t(prob["offset"]), prob["window_size"])
        owner_of, bounds, order = sharding.window_owners(parts["large"], block.offsets, world)
        shard = sharding.make_halo_shard(block, owner_of, order, bounds, rank, world)
        own = shard.halo.own_ids
        q, k, v = (t(prob[x])[own].clone().requires_grad_(True) for x in ("q", "k", "v"))
        tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
        sharding.reset_bytes()
        out = sharding.halo_window_attention(oracle_ops, shard, q, k, v, tq, tk, tv)
        fwd_bytes = sharding.BYTES_MOVED
        out.backward(t(prob["go_rows"])[own])
        # FPS of a batch over the ranks: every rank ends with the single-process index list
        fps = lambda x, o, no: t(pointops_ref.furthestsampling(x.numpy(), o.numpy(), no.numpy()))
        offs3 = [500, 900, 1500]
        new3 = index_build.stratified_new_offset(offs3, 8)
        ds = sharding.sharded_furthestsampling(fps, t(prob["xyz"]), offs3, new3, rank, world)
        ret[rank] = dict(own=own.numpy(), bounds=bounds, out=out.detach().numpy(), gq=q.grad.numpy(), gk=k.grad.numpy(), gv=v.grad.numpy(),
                         gtq=tq.grad.numpy(), gtk=tk.grad.numpy(), gtv=tv.grad.numpy(), pairs=int(shard.index_1.shape[0]),
                         n_need=int(shard.halo.need_ids.shape[0]), n_send=int(shard.halo.send_ids.shape[0]), fwd_bytes=fwd_bytes,
                         halo_fraction=shard.halo.halo_fraction(), ds=ds.numpy())
    finally:
        dist.destroy_process_group()


def test_window_ownership_moves_only_halo_rows_world2():
    """SURVEY 8e / north_star "all-gather of boundary keys": ownership by large window, all_to_all of the halo rows only.  Same
    rows, same gradients as the unsharded run; what travels is a small fraction of what the all-gather variant moves."""
    prob = window_problem(3000, seed=22, h=3, d=16, nbatch=1)
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_halo_worker, args=(world, _free_port(), prob, ret), nprocs=world, join=True)
    from oracle import pointops_ref
    from stratified_transformer_amd import index_build
    from tests import oracle_ops as O
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    q, k, v = (t(prob[x]).clone().requires_grad_(True) for x in ("q", "k", "v"))
    tq, tk, tv = (t(prob[x]).clone().requires_grad_(True) for x in ("table_q", "table_k", "table_v"))
    offs, i1, rel = t(prob["offsets"]), t(prob["index_1"]), t(prob["rel_idx"])
    a1 = O.attention_step1_v2(q, k, i1, offs, 0)
    a2 = O.dot_prod_with_idx_v3(q, offs, 0, k, i1, tq, tk, rel)
    out = O.attention_step2_with_rel_pos_value_v2(O.segment_softmax(a1 + a2, offs), v, offs, 0, i1, tv, rel)
    out.backward(t(prob["go_rows"]))
    r0, r1 = ret[0], ret[1]
    assert r0["bounds"] == r1["bounds"]
    assert sorted(np.concatenate([r0["own"], r1["own"]]).tolist()) == list(range(prob["N"]))      # every row has exactly one owner
    assert abs(r0["pairs"] - r1["pairs"]) < 0.1 * prob["M"] and r0["pairs"] + r1["pairs"] == prob["M"]
    assert r0["n_need"] == r1["n_send"] and r1["n_need"] == r0["n_send"]                          # the two ends of every transfer agree
    C = prob["h"] * prob["d"]
    for r in (r0, r1):
        own = r["own"]
        assert r["fwd_bytes"] == 2 * r["n_send"] * C * 4                                            # k and v: the rows the peer needs, nothing else
        assert r["fwd_bytes"] < 0.25 * 2 * prob["N"] * C * 4, (r["fwd_bytes"], 2 * prob["N"] * C * 4)
        assert r["halo_fraction"] < 0.25
        np.testing.assert_allclose(r["out"], out.detach().numpy()[own], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r["gq"], q.grad.numpy()[own], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(r["gk"], k.grad.numpy()[own], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(r["gv"], v.grad.numpy()[own], rtol=1e-4, atol=1e-4)
        for name, p_ in (("gtq", tq), ("gtk", tk), ("gtv", tv)):
            np.testing.assert_allclose(r[name], p_.grad.numpy(), rtol=2e-4, atol=2e-4)
        want = pointops_ref.furthestsampling(prob["xyz"], np.array([500, 900, 1500], np.int32),
                                             np.asarray(index_build.stratified_new_offset([500, 900, 1500], 8), np.int32))
        assert np.array_equal(r["ds"], want)
