// A1: attention_step1 (QK^T over the CSR pair list) and the table-free AV product, gfx950.
//
// Replaces lib/pointops2/src/attention_v2/attention_cuda_kernel_v2.cu:7-146 and
// lib/pointops2/src/attention/attention_cuda_kernel.cu:7-105 behind the same launchers.
//
// Mapping (not the reference's one-block-per-(query,head)): one 64-wide wavefront owns one query
// and ALL its heads.  A head vector of D floats is spread over LPG = D/4 lanes (one 16-byte load
// each), so a wave holds PPW = 64/LPG pairs side by side and every key row is fetched as whole
// 64-byte (D=16) / 128-byte (D=32) segments.  The q row is staged once per wave in LDS and read
// back with broadcast ds_read_b128.  Per-pair results for LPG consecutive heads are parked in the
// LPG lanes of the pair's group so the [PPW, h] output tile is written as one contiguous run.
//
// Backward never scatters with global float atomics when a key-major (CSC) view of the pair list
// is available (pointops2_set_csc): grad_q is a by-query gather-accumulate, grad_k the same kernel
// run by key.  Without a CSC the by-query kernel falls back to global atomics (the reference's own
// scheme, attention_cuda_kernel_v2.cu:84).
#include "common.h"

namespace p2 {

// ------------------------------------------------------------------------------------------------
// A1 forward: attn[m, hh] = <q[query(m), hh, :], k[index1[m], hh, :]>
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void a1_fwd_kernel(int N, int h, const float *__restrict__ q,
                                                     const float *__restrict__ k,
                                                     const int *__restrict__ offs,
                                                     const int *__restrict__ idx1,
                                                     float *__restrict__ attn, const int *__restrict__ rord) {
    constexpr int LPG = Geo<D>::LPG, PPW = Geo<D>::PPW;
    extern __shared__ float4 lds4[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RowSlots slots(rord, N, 4, wave);
    if (!slots.more()) return;
    const int qi = slots.row();
    const int C = h * D, C4 = C / 4;
    float4 *qs = lds4 + wave * C4;
    for (int i = lane; i < C4; i += 64) qs[i] = ldg4(q + (size_t)qi * C + 4 * i);
    // same wave writes and reads: LDS ops of one wave retire in order
    __builtin_amdgcn_wave_barrier();
    const int p = lane / LPG, c = lane % LPG;
    const int s = offs[qi], e = offs[qi + 1];
    int jn = idx1[max(0, min(s + p, e - 1))];  // the key id of the next pass travels with the key rows of this one
    for (int m0 = s; m0 < e; m0 += PPW) {
        const int m = m0 + p;
        const bool valid = m < e;
        const int j = jn;
        jn = idx1[min(m + PPW, e - 1)];
        const float *krow = k + (size_t)j * C + 4 * c;
        for (int hb = blockIdx.y * LPG; hb < h; hb += gridDim.y * LPG) {  // head groups over blockIdx.y on small clouds
            float keep = 0.f;
            float4 kv[LPG];
#pragma unroll
            for (int t = 0; t < LPG; t++) kv[t] = ldg4(krow + min(hb + t, h - 1) * D);  // all key rows of the group at once
#pragma unroll
            for (int t = 0; t < LPG; t++) {
                float part = dot4(qs[min(hb + t, h - 1) * LPG + c], kv[t]);
                float tot = xor_sum<1, LPG>(part);
                if (c == t) keep = tot;
            }
            const int hh = hb + c;
            if (valid && hh < h) attn[(size_t)m * h + hh] = keep;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gather-accumulate: out[row, hh, :] (+)= sum over the row's slots of w[widx(slot), hh] * src[sidx(slot), hh, :]
//   by query : slots = CSR segment, widx = identity, sidx = index1      (grad_q of A1, plain AV forward)
//   by key   : slots = CSC segment, widx = csc_pair,  sidx = csc_query  (grad_k of A1, grad_v of A4)
// HC heads are accumulated per pass (HC float4 accumulators per lane).
// ------------------------------------------------------------------------------------------------
template <int D, bool ACCUMULATE>
__global__ __launch_bounds__(256) void gather_accum_kernel(int N, int h, const int *__restrict__ offs,
                                                           const int *__restrict__ sidx,
                                                           const int *__restrict__ widx,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ src,
                                                           float *__restrict__ out, const int *__restrict__ rord) {
    constexpr int LPG = Geo<D>::LPG, PPW = Geo<D>::PPW, HC = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RowSlots slots(rord, N, 4, wave);
    if (!slots.more()) return;
    const int row = slots.row();
    const int C = h * D;
    const int p = lane / LPG, c = lane % LPG;
    const int s = offs[row], e = offs[row + 1];
    // head chunks are spread over blockIdx.y: the index walk is repeated per chunk anyway, and on the late
    // stages (few rows, many heads) one wave per row leaves most of the chip idle
    for (int hb = blockIdx.y * HC; hb < h; hb += gridDim.y * HC) {
        float4 acc[HC];
#pragma unroll
        for (int t = 0; t < HC; t++) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 prev[HC];  // ACCUMULATE: the row's running sum, requested now and consumed after the walk
        if (ACCUMULATE) {
#pragma unroll
            for (int t = 0; t < HC; t++) prev[t] = ldg4(out + (size_t)row * C + min(hb + t, h - 1) * D + 4 * c);
        }
        // the ids of the next pass are requested together with the weights and rows of this one (one round trip per
        // pass); a slot past the row's end repeats the row's last slot and is not added
        const int first = max(0, min(s + p, e - 1));
        int sn = sidx[first], wn = ACCUMULATE ? widx[first] : first;  // by key: CSC pair ids; by query: identity
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            const int nslot = min(slot + PPW, e - 1);
            const float *srow = src + (size_t)sn * C + 4 * c;
            const float *wrow = w + (size_t)wn * h;
            sn = sidx[nslot];
            wn = ACCUMULATE ? widx[nslot] : nslot;
            // no per-head guards: a slot past the last head repeats it (its sum is not stored), so the HC weight and
            // row loads of a pass are issued together instead of one dependent round trip per head
            if (slot < e) {
                float wv[HC];
                float4 sv[HC];
#pragma unroll
                for (int t = 0; t < HC; t++) {
                    const int hh = min(hb + t, h - 1);
                    wv[t] = wrow[hh];
                    sv[t] = ldg4(srow + hh * D);
                }
                __builtin_amdgcn_sched_barrier(0);  // every load of the pass before its first use
#pragma unroll
                for (int t = 0; t < HC; t++) acc[t] = fma4(wv[t], sv[t], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < HC; t++) {
            const int hh = hb + t;
            if (hh < h) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) {
                    float *o = out + (size_t)row * C + hh * D + 4 * c;
                    if (ACCUMULATE) tot = add4(tot, prev[t]);
                    stg4(o, tot);
                }
            }
        }
    }
}

// fallback when no CSC is set: dst[index1[m], hh, :] += w[m, hh] * src[query(m), hh, :] with global atomics
template <int D>
__global__ __launch_bounds__(256) void scatter_atomic_kernel(int N, int h, const int *__restrict__ offs,
                                                             const int *__restrict__ idx1,
                                                             const float *__restrict__ w,
                                                             const float *__restrict__ src,
                                                             float *__restrict__ dst) {
    constexpr int LPG = Geo<D>::LPG, PPW = Geo<D>::PPW;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int C = h * D;
    const int p = lane / LPG, c = lane % LPG;
    const int s = offs[qi], e = offs[qi + 1];
    for (int m0 = s; m0 < e; m0 += PPW) {
        const int m = m0 + p;
        if (m < e) {
            const int j = idx1[m];
            for (int hh = 0; hh < h; hh++) {
                const float g = w[(size_t)m * h + hh];
                const float4 s4 = ldg4(src + (size_t)qi * C + hh * D + 4 * c);
                float *d = dst + (size_t)j * C + hh * D + 4 * c;
                atomicAdd(d + 0, g * s4.x);
                atomicAdd(d + 1, g * s4.y);
                atomicAdd(d + 2, g * s4.z);
                atomicAdd(d + 3, g * s4.w);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v1 pair-indexed forms (arbitrary, unsorted index0): one thread per (pair, head), atomics on the
// scattered side exactly where the reference has them.  Off the model's hot path
// (model/stratified_transformer.py:210 only when rel_value=False).
// ------------------------------------------------------------------------------------------------
__global__ void step1_v1_fwd_kernel(int M, int h, int d, const float *q, const float *k, const int *i0,
                                    const int *i1, float *attn) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h, C = h * d;
    const float *qv = q + (size_t)i0[m] * C + hh * d, *kv = k + (size_t)i1[m] * C + hh * d;
    float sum = 0.f;
    for (int i = 0; i < d; i += 4) sum += dot4(ldg4(qv + i), ldg4(kv + i));
    attn[t] += sum;
}
__global__ void step1_v1_bwd_kernel(int M, int h, int d, const float *go, const int *i0, const int *i1,
                                    const float *q, const float *k, float *gq, float *gk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h, C = h * d;
    const size_t qb = (size_t)i0[m] * C + hh * d, kb = (size_t)i1[m] * C + hh * d;
    const float g = go[t];
    for (int i = 0; i < d; i++) {
        atomicAdd(gq + qb + i, g * k[kb + i]);
        atomicAdd(gk + kb + i, g * q[qb + i]);
    }
}
__global__ void step2_v1_fwd_kernel(int M, int h, int d, const float *attn, const float *v, const int *i0,
                                    const int *i1, float *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h, C = h * d;
    const size_t ob = (size_t)i0[m] * C + hh * d, vb = (size_t)i1[m] * C + hh * d;
    const float a = attn[t];
    for (int i = 0; i < d; i++) atomicAdd(out + ob + i, a * v[vb + i]);
}
__global__ void step2_v1_bwd_kernel(int M, int h, int d, const float *go, const int *i0, const int *i1,
                                    const float *attn, const float *v, float *ga, float *gv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h, C = h * d;
    const size_t ob = (size_t)i0[m] * C + hh * d, vb = (size_t)i1[m] * C + hh * d;
    const float a = attn[t];
    float sum = 0.f;
    for (int i = 0; i < d; i++) {
        sum = fmaf(go[ob + i], v[vb + i], sum);
        atomicAdd(gv + vb + i, go[ob + i] * a);
    }
    ga[t] += sum;
}

template <typename F16, typename F32>
static bool dispatch_d(int d, F16 f16, F32 f32) {
    if (d == 16) { f16(); return true; }
    if (d == 32) { f32(); return true; }
    set_error("d != 16 and d != 32");
    return false;
}

}  // namespace p2

using namespace p2;

extern "C" {

void attention_step1_forward_cuda_launcher_v2(int N, int M, int h, int C, const unsigned int n_max,
                                              const float *q, const float *k, const int *index0_offsets,
                                              const int *index1, float *attn) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    hipStream_t st = state().stream;
    const int *rord = rows_in_order(N);
    const int blocks = rord ? ordered_grid(N, 4) : div_up(N, 4);
    const size_t lds = 4 * (size_t)C * sizeof(float);
    dispatch_d(C / h,
               [&] { hipLaunchKernelGGL(a1_fwd_kernel<16>, dim3(blocks, N < 20000 ? div_up(h, 4) : 1), dim3(256), lds, st, N, h, q, k, index0_offsets, index1, attn, rord); },
               [&] { hipLaunchKernelGGL(a1_fwd_kernel<32>, dim3(blocks, N < 20000 ? div_up(h, 8) : 1), dim3(256), lds, st, N, h, q, k, index0_offsets, index1, attn, rord); });
    check_launch();
}

void attention_step1_backward_cuda_launcher_v2(int N, int M, int h, int C, const unsigned int n_max,
                                               const float *grad_out, const int *index0_offsets,
                                               const int *index1, const float *q, const float *k,
                                               float *grad_q, float *grad_k) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    hipStream_t st = state().stream;
    const LaunchState &ls = state();
    const int *rord = rows_in_order(N);
    const int blocks = rord ? ordered_grid(N, 4) : div_up(N, 4);
    const int *co = ls.csc_offsets, *cp = ls.csc_pair, *cq = ls.csc_query;
    ForkJoin fj(st, fork_worthwhile((int64_t)M * h));  // grad_q and grad_k are independent
    auto run = [&](auto dtag) {
        constexpr int D = decltype(dtag)::value;
        const int chunks = div_up(h, 4);
        hipLaunchKernelGGL((gather_accum_kernel<D, false>), dim3(blocks, chunks), dim3(256), 0, st, N, h, index0_offsets,
                           index1, (const int *)nullptr, grad_out, k, grad_q, rord);
        if (co) {
            const int NK = ls.key_rows > 0 ? ls.key_rows : N;
            const int *kord = rows_in_order(NK);  // (the keys of a window are its queries: the same order serves the transposed list)
            hipLaunchKernelGGL((gather_accum_kernel<D, true>), dim3(kord ? ordered_grid(NK, 4) : div_up(NK, 4), chunks), dim3(256), 0, fj.lane(1), NK, h, co, cq, cp,
                               grad_out, q, grad_k, kord);
        }
        else
            hipLaunchKernelGGL(scatter_atomic_kernel<D>, dim3(blocks), dim3(256), 0, st, N, h, index0_offsets,
                               index1, grad_out, q, grad_k);
    };
    dispatch_d(C / h, [&] { run(std::integral_constant<int, 16>{}); }, [&] { run(std::integral_constant<int, 32>{}); });
    check_launch();
}

void attention_step1_forward_cuda_launcher(int N, int M, int h, int C, const float *q, const float *k,
                                           const int *index0, const int *index1, float *attn) {
    (void)N;
    if (M <= 0) return;
    if ((C / h) % 4) { set_error("head dim must be a multiple of 4"); return; }
    hipLaunchKernelGGL(step1_v1_fwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, C / h, q, k, index0, index1, attn);
    check_launch();
}
void attention_step1_backward_cuda_launcher(int N, int M, int h, int C, const float *grad_out,
                                            const int *index0, const int *index1, const float *q,
                                            const float *k, float *grad_q, float *grad_k) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(step1_v1_bwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, C / h, grad_out, index0, index1, q, k, grad_q, grad_k);
    check_launch();
}
void attention_step2_forward_cuda_launcher(int N, int M, int h, int C, const float *attn, const float *v,
                                           const int *index0, const int *index1, float *output) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(step2_v1_fwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, C / h, attn, v, index0, index1, output);
    check_launch();
}
void attention_step2_backward_cuda_launcher(int N, int M, int h, int C, const float *grad_out,
                                            const int *index0, const int *index1, const float *attn,
                                            const float *v, float *grad_attn, float *grad_v) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(step2_v1_bwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, C / h, grad_out, index0, index1, attn, v, grad_attn, grad_v);
    check_launch();
}
// attention_cuda_kernel_v2.cu:148-195 is a byte-for-byte copy of the v1 step2 kernels
void attention_step2_forward_cuda_launcher_v2(int N, int M, int h, int C, const float *attn, const float *v,
                                              const int *index0, const int *index1, float *output) {
    attention_step2_forward_cuda_launcher(N, M, h, C, attn, v, index0, index1, output);
}
void attention_step2_backward_cuda_launcher_v2(int N, int M, int h, int C, const float *grad_out,
                                               const int *index0, const int *index1, const float *attn,
                                               const float *v, float *grad_attn, float *grad_v) {
    attention_step2_backward_cuda_launcher(N, M, h, C, grad_out, index0, index1, attn, v, grad_attn, grad_v);
}

}  // extern "C"
