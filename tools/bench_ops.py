"""Per-op timing of the attention operators at one stage's size (diagnostic; GPU box only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P, index_build, pipeline

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    cfg = pipeline.s3dis_config(); st = cfg.stages[0]
    xyz_np = scene.make_room(N, 0)
    if os.environ.get('SORTED'):
        c = np.floor(xyz_np / 0.16).astype(np.int64)
        key = (c[:, 2] * 64 + c[:, 1]) * 64 + c[:, 0]
        xyz_np = np.ascontiguousarray(xyz_np[np.argsort(key, kind='stable')])
    xyz = torch.from_numpy(xyz_np).cuda()
    off = torch.tensor([N], dtype=torch.int32, device='cuda')
    ds = P.furthestsampling(xyz, off, torch.tensor([N // 8 + 1], dtype=torch.int32, device='cuda'))
    even, odd, _ = index_build.stage_index_hip(xyz, off, st.window_size, st.quant_size, ds)
    s = pipeline.make_stage_state(xyz, off, st, 0)
    q, k, v = s.q, s.k, s.v; tq, tk, tv = s.tables; b = even
    M = b.index_1.shape[0]; print('N', N, 'M', M, 'h', st.num_heads)
    a1 = P.attention_step1_v2(q, k, b.index_1, b.offsets, 0)
    a2 = P.dot_prod_with_idx_v3(q, b.offsets, 0, k, b.index_1, tq, tk, b.rel_idx)
    sm = P.segment_softmax((a1 + a2).detach(), b.offsets)
    out = P.attention_step2_with_rel_pos_value_v2(sm.detach().requires_grad_(True), v, b.offsets, 0, b.index_1, tv, b.rel_idx)
    g_pairs = torch.randn(M, st.num_heads, device='cuda'); g_rows = s.grad_out
    res = {}
    res['A1 fwd'] = timeit(lambda: P.attention_step1_v2(q, k, b.index_1, b.offsets, 0))
    res['A2 fwd'] = timeit(lambda: P.dot_prod_with_idx_v3(q, b.offsets, 0, k, b.index_1, tq, tk, b.rel_idx))
    res['A3 fwd'] = timeit(lambda: P.segment_softmax(a1.detach(), b.offsets))
    res['A4 fwd'] = timeit(lambda: P.attention_step2_with_rel_pos_value_v2(sm.detach(), v, b.offsets, 0, b.index_1, tv, b.rel_idx))
    def bwd(t, g):
        for x in (q, k, v, tq, tk, tv): x.grad = None
        t.backward(g, retain_graph=True)
    res['A1 bwd'] = timeit(lambda: bwd(a1, g_pairs))
    res['A2 bwd'] = timeit(lambda: bwd(a2, g_pairs))
    res['A4 bwd'] = timeit(lambda: bwd(out, g_rows))
    print('ablate', os.environ.get('P2_ABLATE', '0'), {k_: round(v_) for k_, v_ in res.items()})

main()
