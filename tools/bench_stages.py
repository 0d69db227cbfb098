"""Per-op, per-stage timing of the attention operators on the bench scene (diagnostic; GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P, pipeline


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
    cfg = pipeline.s3dis_config()
    xyz = torch.from_numpy(scene.make_room(N, 0)).cuda()
    off = torch.tensor([N], dtype=torch.int32, device='cuda')
    states, results = pipeline.scene_pass(xyz, off, cfg, cells=True)
    torch.cuda.synchronize()
    total = {}
    for si, (s, r) in enumerate(zip(states, results)):
        st = cfg.stages[si]
        b = r['even']
        q, k, v = s.q, s.k, s.v
        tq, tk, tv = s.tables
        M = b.index_1.shape[0]
        a1 = P.attention_step1_v2(q, k, b.index_1, b.offsets, 0)
        a2 = P.dot_prod_with_idx_v3(q, b.offsets, 0, k, b.index_1, tq, tk, b.rel_idx)
        sm = P.segment_softmax((a1 + a2).detach().requires_grad_(True), b.offsets)
        out = P.attention_step2_with_rel_pos_value_v2(sm.detach().requires_grad_(True), v, b.offsets, 0, b.index_1, tv, b.rel_idx)
        g_pairs = torch.randn(M, st.num_heads, device='cuda')

        def bwd(t, g):
            for x in (q, k, v, tq, tk, tv): x.grad = None
            t.backward(g, retain_graph=True)
        res = {}
        res['A1f'] = timeit(lambda: P.attention_step1_v2(q, k, b.index_1, b.offsets, 0))
        res['A2f'] = timeit(lambda: P.dot_prod_with_idx_v3(q, b.offsets, 0, k, b.index_1, tq, tk, b.rel_idx))
        res['A3f'] = timeit(lambda: P.segment_softmax(a1.detach(), b.offsets))
        res['A4f'] = timeit(lambda: P.attention_step2_with_rel_pos_value_v2(sm.detach(), v, b.offsets, 0, b.index_1, tv, b.rel_idx))
        res['A1b'] = timeit(lambda: bwd(a1, g_pairs))
        res['A2b'] = timeit(lambda: bwd(a2, g_pairs))
        res['A3b'] = timeit(lambda: bwd(sm, g_pairs))
        res['A4b'] = timeit(lambda: bwd(out, s.grad_out))
        from stratified_transformer_amd import fused
        fo = fused.window_attention(q, k, v, tq, tk, tv, b.offsets, b.index_1, b.rel_idx)
        fwd_sum, bwd_sum = sum(v_ for k_, v_ in res.items() if k_.endswith('f')), sum(v_ for k_, v_ in res.items() if k_.endswith('b'))
        tot = sum(res.values())
        res['FUSEDf'] = timeit(lambda: fused.window_attention(q, k, v, tq, tk, tv, b.offsets, b.index_1, b.rel_idx))
        res['FUSEDb'] = timeit(lambda: bwd(fo, s.grad_out))
        res['ops_f'], res['ops_b'] = fwd_sum, bwd_sum
        for pat in ('even', 'odd'):
            plan = r[pat].cells
            co = fused.cell_attention(q, k, v, tq, tk, tv, plan)
            res['CELLf_' + pat] = timeit(lambda: fused.cell_attention(q, k, v, tq, tk, tv, plan))
            res['CELLb_' + pat] = timeit(lambda: bwd(co, s.grad_out))
        print('   cells even/odd', r['even'].cells.n_cells, r['odd'].cells.n_cells, 'P', r['even'].cells.n_pairs, r['odd'].cells.n_pairs,
              'nk_max', r['even'].cells.nk_max, r['odd'].cells.nk_max)
        print('stage', si, 'N', s.xyz.shape[0], 'M', M, 'h', st.num_heads, 'L', tq.shape[0], 'depth', st.depth,
              {k_: round(v_) for k_, v_ in res.items()}, 'block us', round(tot), 'stage ms', round(tot * st.depth / 1e3, 2))
        for k_, v_ in res.items():
            total[k_] = total.get(k_, 0) + v_ * st.depth / 1e3
    print('per step ms', {k_: round(v_, 2) for k_, v_ in total.items()}, 'sum', round(sum(total.values()), 2))


main()
