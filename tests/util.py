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
