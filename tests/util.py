"""Shared helpers of the test-suite (CPU side: builds seeded problems; the oracle is the checker)."""
import numpy as np
import torch

from oracle import index_ref
from stratified_transformer_amd import index_build, scene


def window_problem(n, seed, h=3, d=16, w=0.16, quant=0.01, scale=8, nbatch=1, shifted=False, L=None):
    """A seeded room with its block index (built by the oracle's restatement of the reference) and
    random q/k/v/tables.  Everything numpy/CPU."""
    sizes = [n // nbatch + (1 if i < n % nbatch else 0) for i in range(nbatch)]
    xyz, offset = scene.make_batch(sizes, seed)
    x = torch.from_numpy(xyz)
    rng = np.random.default_rng(seed)
    new_offset = index_ref.stratified_new_offset(offset, scale)
    ds, lo, mlo = [], 0, 0
    for b in range(nbatch):
        m_b = int(new_offset[b]) - mlo
        ds.append(lo + np.sort(rng.permutation(int(offset[b]) - lo)[:m_b]))
        lo, mlo = int(offset[b]), int(new_offset[b])
    ds = torch.from_numpy(np.concatenate(ds).astype(np.int32))
    idx = index_ref.build_stage_indices(x, offset, w, quant, ds, 1 if shifted else 0, div_mode="cuda")
    L = L or 2 * int((2 * w + 1e-4) // quant)
    rel = idx["rel_idx"].numpy().clip(0, L - 1)  # the model asserts this range (:189-190)
    C = h * d
    p = dict(xyz=xyz, offset=offset, N=n, M=int(idx["index_1"].shape[0]), h=h, d=d, L=L,
             index_0=idx["index_0"].numpy().astype(np.int32), index_1=idx["index_1"].numpy().astype(np.int32),
             offsets=idx["offsets"].numpy().astype(np.int32), n_max=int(idx["n_max"]), rel_idx=np.ascontiguousarray(rel, dtype=np.int32),
             q=rng.standard_normal((n, h, d), dtype=np.float32), k=rng.standard_normal((n, h, d), dtype=np.float32),
             v=rng.standard_normal((n, h, d), dtype=np.float32),
             table_q=(rng.standard_normal((L, h, d, 3), dtype=np.float32) * 0.5),
             table_k=(rng.standard_normal((L, h, d, 3), dtype=np.float32) * 0.5),
             table_v=(rng.standard_normal((L, h, d, 3), dtype=np.float32) * 0.5),
             downsample_idx=ds.numpy(), window_size=w, quant_size=quant)
    p["go_pairs"] = rng.standard_normal((p["M"], h), dtype=np.float32)
    p["go_rows"] = rng.standard_normal((n, h, d), dtype=np.float32)
    p["attn"] = rng.random((p["M"], h), dtype=np.float32)
    return p


def random_csr_problem(n, seed, h, d, L, mean_len=12, max_len=None, empty_frac=0.1):
    """Unstructured CSR (random keys, ragged lengths incl. empty and very long segments)."""
    rng = np.random.default_rng(seed)
    lens = rng.poisson(mean_len, n)
    lens[rng.random(n) < empty_frac] = 0
    if max_len:
        lens[rng.integers(0, n)] = max_len
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    M = int(offsets[-1])
    index_1 = np.concatenate([np.sort(rng.choice(n, l, replace=l > n)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    index_0 = np.repeat(np.arange(n), lens).astype(np.int32)
    p = dict(N=n, M=M, h=h, d=d, L=L, offsets=offsets, index_0=index_0, index_1=index_1, n_max=int(lens.max()),
             rel_idx=rng.integers(0, L, (M, 3)).astype(np.int32),
             q=rng.standard_normal((n, h, d), dtype=np.float32), k=rng.standard_normal((n, h, d), dtype=np.float32),
             v=rng.standard_normal((n, h, d), dtype=np.float32),
             table_q=rng.standard_normal((L, h, d, 3), dtype=np.float32), table_k=rng.standard_normal((L, h, d, 3), dtype=np.float32),
             table_v=rng.standard_normal((L, h, d, 3), dtype=np.float32),
             go_pairs=rng.standard_normal((M, h), dtype=np.float32), go_rows=rng.standard_normal((n, h, d), dtype=np.float32),
             attn=rng.random((M, h), dtype=np.float32))
    return p


def dev(a, device="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def oracle_scene_pass(xyz, offset, cfg, states):
    """The oracle's restatement of pipeline.scene_pass (numpy/CPU): per stage the stratified FPS, both block
    patterns, the last block's attention output, TransitionDown FPS + kNN.  `states` are the GPU pass's resident
    synthetic tensors (q/k/v/tables/grad_out), copied to the host.  Returns a list of per-stage dicts."""
    from oracle import index_ref, pointops_ref as ref
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    offset = np.asarray(offset, np.int32)
    out = []
    first = 0 if cfg.stem_transformer else 1

    def transition(xyz, offset):
        n_offset = np.asarray(index_ref.transition_down_offset(offset, cfg.ratio), np.int32)
        idx = ref.furthestsampling(xyz, offset, n_offset)
        n_xyz = np.ascontiguousarray(xyz[idx])
        kidx, _ = ref.knnquery(cfg.k, xyz, n_xyz, offset, n_offset)
        return n_xyz, n_offset, idx, kidx

    if not cfg.stem_transformer:
        xyz, offset, _, _ = transition(xyz, offset)
    for si in range(first, len(cfg.stages)):
        st, state = cfg.stages[si], states[si - first]
        new_offset = np.asarray(index_ref.stratified_new_offset(offset, cfg.downsample_scale), np.int32)
        ds = ref.furthestsampling(xyz, offset, new_offset)
        x_t = torch.from_numpy(xyz)
        blocks = [index_ref.build_stage_indices(x_t, offset, st.window_size, st.quant_size, torch.from_numpy(ds), par, "cuda") for par in (0, 1)]
        q, k, v = (t.detach().cpu().numpy() for t in (state.q, state.k, state.v))
        tq, tk, tv = (t.detach().cpu().numpy() for t in state.tables)
        L = tq.shape[0]
        blk = blocks[(st.depth - 1) % 2]
        i1, offs = blk["index_1"].numpy().astype(np.int32), blk["offsets"].numpy().astype(np.int32)
        rel = np.clip(blk["rel_idx"].numpy(), 0, L - 1).astype(np.int32)
        sm = ref.segment_softmax(ref.attention_step1_v2(q, k, i1, offs) + ref.dot_prod_with_idx_v3(q, offs, k, i1, tq, tk, rel), offs)
        att = ref.attention_step2_with_rel_pos_value_v2(sm, v, offs, i1, tv, rel)
        row = dict(stage=si, n=xyz.shape[0], downsample_idx=ds, blocks=blocks, out=att)
        if si < len(cfg.stages) - 1:
            xyz, offset, _, kidx = transition(xyz, offset)
            row["transition_knn"] = kidx
        out.append(row)
    return out
