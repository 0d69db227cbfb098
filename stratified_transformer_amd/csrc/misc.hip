// Launch state, row gathers (grouping / interpolation) and the CSR <-> key-major (CSC) transposition.
#include "common.h"
#include <mutex>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

namespace p2 {

LaunchState &state() {
    static thread_local LaunchState s;
    return s;
}

// ---- fork / join ---------------------------------------------------------------------------------------------
namespace {
constexpr int kSide = 3, kMaxDev = 16;
struct SideStreams {
    bool ready = false;
    hipStream_t stream[kSide];
    hipEvent_t fork, joined[kSide];
};
SideStreams *side_streams() {
    static thread_local SideStreams per_dev[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    SideStreams &s = per_dev[dev];
    if (!s.ready) {
        for (int i = 0; i < kSide; i++) {
            if (hipStreamCreateWithFlags(&s.stream[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&s.joined[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.ready = true;
    }
    return &s;
}
}  // namespace

ForkJoin::ForkJoin(hipStream_t main, bool small) : main_(main) {
    static const bool off = getenv("P2_NO_FORK") != nullptr;
    enabled_ = !off && small;
    if (enabled_) {  // the fork point is where the call starts: side lanes do not wait for this call's own lane-0 kernels
        SideStreams *s = side_streams();
        if (s) (void)hipEventRecord(s->fork, main_);
        else enabled_ = false;
    }
}
hipStream_t ForkJoin::lane(int i) {
    if (i <= 0 || !enabled_) return main_;
    SideStreams *s = side_streams();
    if (!s) return main_;
    i = (i - 1) % kSide;
    if (!(used_ & (1u << i))) {
        (void)hipStreamWaitEvent(s->stream[i], s->fork, 0);
        used_ |= 1u << i;
    }
    return s->stream[i];
}
void ForkJoin::join() {
    if (!used_) return;
    SideStreams *s = side_streams();
    for (int i = 0; i < kSide; i++)
        if (used_ & (1u << i)) {
            (void)hipEventRecord(s->joined[i], s->stream[i]);
            (void)hipStreamWaitEvent(main_, s->joined[i], 0);
        }
    used_ = 0;
}

// ---- grouping (grouping_cuda_kernel.cu:5-25) / interpolation (interpolation_cuda_kernel.cu:5-33) ----
__global__ void grouping_fwd_kernel(int64_t total, int c, const float *__restrict__ input, const int *__restrict__ idx,
                                    float *__restrict__ output) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    output[t] = input[(size_t)idx[t / c] * c + t % c];
}
__global__ void grouping_bwd_kernel(int64_t total, int c, const float *__restrict__ go, const int *__restrict__ idx,
                                    float *__restrict__ gi) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    atomicAdd(gi + (size_t)idx[t / c] * c + t % c, go[t]);
}
__global__ void interp_fwd_kernel(int n, int c, int k, const float *__restrict__ input, const int *__restrict__ idx,
                                  const float *__restrict__ weight, float *__restrict__ output) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * c) return;
    const int ci = t % c, ni = t / c;
    float o = output[t];
    for (int i = 0; i < k; i++) o += input[(size_t)idx[ni * k + i] * c + ci] * weight[ni * k + i];
    output[t] = o;
}
__global__ void interp_bwd_kernel(int n, int c, int k, const float *__restrict__ go, const int *__restrict__ idx,
                                  const float *__restrict__ weight, float *__restrict__ gi) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n * c) return;
    const int ci = t % c, ni = t / c;
    for (int i = 0; i < k; i++) atomicAdd(gi + (size_t)idx[ni * k + i] * c + ci, go[t] * weight[ni * k + i]);
}

// ---- CSR helpers ----
// The segment bounds are clamped into [0, M]: offsets that do not describe an M-pair list (a caller's mistake, a stale
// remembered CSR) can then leave entries unwritten but never reach outside index0[0, M).
__global__ __launch_bounds__(256) void csr_expand_kernel(int N, int M, const int *__restrict__ offs, int *__restrict__ index0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int s = max(offs[qi], 0), e = min(offs[qi + 1], M);
    for (int m = s + lane; m < e; m += 64) index0[m] = qi;
}
// Does `offs [N+1]` describe the per-pair query ids `index [M]` (ascending runs, run i = query i)?  *bad stays 0 iff
// offs[0] == 0, offs[N] == M, every segment is ordered and inside [0, M] (=> the segments tile [0, M) exactly) and every
// pair of segment i carries the id i.  One wave per query; nothing outside offs[0..N] / index[0..M) is touched whatever the
// offsets hold.  IDX64: the model's index_0 is int64 (model/stratified_transformer.py:205).
template <typename IndexT>
__global__ __launch_bounds__(256) void csr_matches_kernel(int N, int M, const int *__restrict__ offs, const IndexT *__restrict__ index,
                                                          int *__restrict__ bad) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int s = offs[qi], e = offs[qi + 1];
    bool wrong = s < 0 || e > M || s > e || (qi == 0 && s != 0) || (qi == N - 1 && e != M);
    if (!wrong)
        for (int m = s + lane; m < e; m += 64) wrong |= index[m] != (IndexT)qi;
    if (wrong) *bad = 1;  // benign race: every writer stores the same value
}
__global__ void iota_kernel(int M, int *v) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < M) v[t] = t;
}
// after the stable sort by key: segment starts per key (empty keys included) and the query of each pair
__global__ void csc_finish_kernel(int N, int M, const int *__restrict__ sorted_keys, const int *__restrict__ csc_pair,
                                  const int *__restrict__ index0, int *__restrict__ csc_offsets, int *__restrict__ csc_query) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const int kcur = sorted_keys[t];
    const int kprev = t == 0 ? -1 : sorted_keys[t - 1];
    for (int kk = kprev + 1; kk <= kcur; kk++) csc_offsets[kk] = t;
    if (t == M - 1)
        for (int kk = kcur + 1; kk <= N; kk++) csc_offsets[kk] = M;
    csc_query[t] = index0[csc_pair[t]];
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static int key_bits(int N) {
    int b = 1;
    while ((1ll << b) < (long long)N) b++;
    return b;
}
static size_t cub_sort_bytes(int N, int M) {
    size_t bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const int *)nullptr, (int *)nullptr, (const int *)nullptr,
                                       (int *)nullptr, M, 0, key_bits(N), (hipStream_t) nullptr);
    return bytes;
}

}  // namespace p2

using namespace p2;

namespace p2 {
namespace {
struct HeldSlot { hipEvent_t ev = nullptr; int wgs = 0; bool live = false; };
HeldSlot held_slots[16];
int held_next = 0;
std::mutex held_mutex;
}  // namespace
void held_cus_note(hipStream_t st, int workgroups) {
    std::lock_guard<std::mutex> g(held_mutex);
    HeldSlot &s = held_slots[held_next];
    held_next = (held_next + 1) % 16;
    if (s.ev == nullptr && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { s.ev = nullptr; return; }
    if (hipEventRecord(s.ev, st) != hipSuccess) { s.live = false; return; }
    s.wgs = workgroups;
    s.live = true;
}
int held_cus_now() {
    static const bool off = getenv("P2_NO_HELD_CUS") != nullptr;
    if (off) return 0;
    std::lock_guard<std::mutex> g(held_mutex);
    int total = 0;
    for (HeldSlot &s : held_slots) {
        if (!s.live) continue;
        if (hipEventQuery(s.ev) == hipErrorNotReady) total += s.wgs;
        else s.live = false;
    }
    return total;
}
}  // namespace p2

namespace p2 {
__global__ __launch_bounds__(1024) void diag_hold_kernel(int micros, int mode, unsigned *word) {
    extern __shared__ unsigned char diag_lds[];
    const unsigned long long t0 = wall_clock64(), ticks = (unsigned long long)micros * 100ull;  // 100 MHz
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) {
        if (mode >= 1 && threadIdx.x == 0) acc += __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode >= 2) {
            __hip_atomic_store(word + 64 + blockIdx.x * 1024 + threadIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __hip_atomic_load(word + 64 + ((blockIdx.x + 1) % gridDim.x) * 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < 300; i++) __builtin_amdgcn_s_sleep(1);
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (acc == 0xdeadbeefu) diag_lds[threadIdx.x] = 1;  // (keeps acc and the LDS allocation alive)
}
}  // namespace p2

namespace p2 {
static unsigned *g_async_status = nullptr;
unsigned long long g_fps_patience = 200000000ull;  // 2 s of the 100 MHz clock (fps_lazy.hip, LZ_PATIENCE)
unsigned *async_status_word() {
    static unsigned *dev_ptr = [] {
        unsigned *host = nullptr;
        void *dev = nullptr;
        if (hipHostMalloc(reinterpret_cast<void **>(&host), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return (unsigned *)nullptr;
        *host = 0u;
        if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess) return (unsigned *)nullptr;
        g_async_status = host;
        return reinterpret_cast<unsigned *>(dev);
    }();
    return dev_ptr;
}
}  // namespace p2

extern "C" {

void pointops2_set_stream(void *hip_stream) { state().stream = reinterpret_cast<hipStream_t>(hip_stream); }
void *pointops2_get_stream(void) { return reinterpret_cast<void *>(state().stream); }
const char *pointops2_last_error(void) {
    const char *e = state().error;
    state().error = nullptr;
    if (e == nullptr) {
        unsigned *w = g_async_status;  // (never allocates here)
        if (w != nullptr) {
            const unsigned bits = __atomic_exchange_n(w, 0u, __ATOMIC_RELAXED);
            if (bits & ASYNC_FPS_BARRIER_TIMEOUT)
                e = "furthestsampling: a workgroup of the round sampler waited for its grid barrier beyond its patience and the sampler "
                    "gave up - the sample indices of that call are INVALID (fps_lazy.hip; P2_FPS_STEPWISE=1 selects the step-by-step sampler)";
        }
    }
    return e;
}
void pointops2_diag_set_fps_patience(unsigned long long ticks_100mhz) { g_fps_patience = ticks_100mhz; }
int pointops2_abi_version(void) { return 2; }  // 2: pointops2_cell_plan.table_rows, pointops2_csr_matches_launcher
void pointops2_set_table_rows(int L) { state().table_rows = L; }
void pointops2_set_point_count(int N) { state().total_points = N; }
void pointops2_set_batch_count(int b) { state().batch_count = b; }
void pointops2_set_key_rows(int n) { state().key_rows = n; }
void pointops2_set_row_order(const int *order, int n_rows) {
    LaunchState &s = state();
    s.row_order = order;
    s.row_order_n = order != nullptr ? n_rows : 0;
}
void pointops2_set_csc(const int *csc_offsets, const int *csc_pair, const int *csc_query) {
    LaunchState &s = state();
    s.csc_offsets = csc_offsets;
    s.csc_pair = csc_pair;
    s.csc_query = csc_query;
}

void grouping_forward_cuda_launcher(int m, int nsample, int c, const float *input, const int *idx, float *output) {
    const int64_t total = (int64_t)m * nsample * c;
    if (total <= 0) return;
    hipLaunchKernelGGL(grouping_fwd_kernel, dim3(div_up64(total, 256)), dim3(256), 0, state().stream, total, c, input, idx, output);
    check_launch();
}
void grouping_backward_cuda_launcher(int m, int nsample, int c, const float *grad_output, const int *idx, float *grad_input) {
    const int64_t total = (int64_t)m * nsample * c;
    if (total <= 0) return;
    hipLaunchKernelGGL(grouping_bwd_kernel, dim3(div_up64(total, 256)), dim3(256), 0, state().stream, total, c, grad_output, idx, grad_input);
    check_launch();
}
void interpolation_forward_cuda_launcher(int n, int c, int k, const float *input, const int *idx, const float *weight, float *output) {
    if ((int64_t)n * c <= 0) return;
    hipLaunchKernelGGL(interp_fwd_kernel, dim3(div_up64((int64_t)n * c, 256)), dim3(256), 0, state().stream, n, c, k, input, idx, weight, output);
    check_launch();
}
void interpolation_backward_cuda_launcher(int n, int c, int k, const float *grad_output, const int *idx, const float *weight, float *grad_input) {
    if ((int64_t)n * c <= 0) return;
    hipLaunchKernelGGL(interp_bwd_kernel, dim3(div_up64((int64_t)n * c, 256)), dim3(256), 0, state().stream, n, c, k, grad_output, idx, weight, grad_input);
    check_launch();
}

void csr_expand_launcher(int N, int M, const int *offsets, int *index0) {
    if (N <= 0 || M <= 0) return;
    hipLaunchKernelGGL(csr_expand_kernel, dim3(div_up(N, 4)), dim3(256), 0, state().stream, N, M, offsets, index0);
    check_launch();
}

void pointops2_csr_matches_launcher(int N, int M, const int *offsets, const void *index, int index_is_int64, int *bad) {
    hipStream_t st = state().stream;
    // N == 0 describes M == 0 only; a non-empty pair list needs at least one row
    if (hipMemsetAsync(bad, (N <= 0 && M > 0) ? 1 : 0, sizeof(int), st) != hipSuccess) { set_error("pointops2_csr_matches: memset failed"); return; }
    if (N <= 0) return;
    if (index_is_int64)
        hipLaunchKernelGGL(csr_matches_kernel<long long>, dim3(div_up(N, 4)), dim3(256), 0, st, N, M, offsets, (const long long *)index, bad);
    else
        hipLaunchKernelGGL(csr_matches_kernel<int>, dim3(div_up(N, 4)), dim3(256), 0, st, N, M, offsets, (const int *)index, bad);
    check_launch();
}

// workspace layout: [index0 M][iota M][sorted keys M][hipcub temp]
size_t pointops2_csc_workspace_bytes(int N, int M) {
    if (N <= 0 || M <= 0) return 0;
    return 3 * align256((size_t)M * sizeof(int)) + align256(cub_sort_bytes(N, M));
}

void pointops2_csc_build(int N, int M, const int *index0_offsets, const int *index1,
                         int *csc_offsets, int *csc_pair, int *csc_query, void *workspace, size_t workspace_bytes) {
    if (N <= 0 || M <= 0) return;
    const int NK = state().key_rows > 0 ? state().key_rows : N;  // keys may outnumber the CSR's queries (sharded attention)
    state().key_rows = 0;
    if (workspace_bytes < pointops2_csc_workspace_bytes(NK > N ? NK : N, M)) { set_error("pointops2_csc_build: workspace too small"); return; }
    hipStream_t st = state().stream;
    char *ws = reinterpret_cast<char *>(workspace);
    const size_t seg = align256((size_t)M * sizeof(int));
    int *index0 = reinterpret_cast<int *>(ws);
    int *iota = reinterpret_cast<int *>(ws + seg);
    int *keys = reinterpret_cast<int *>(ws + 2 * seg);
    void *cub_tmp = ws + 3 * seg;
    size_t cub_bytes = workspace_bytes - 3 * seg;
    hipLaunchKernelGGL(csr_expand_kernel, dim3(div_up(N, 4)), dim3(256), 0, st, N, M, index0_offsets, index0);
    hipLaunchKernelGGL(iota_kernel, dim3(div_up(M, 256)), dim3(256), 0, st, M, iota);
    // stable LSD radix sort: per key the pair ids stay ascending => deterministic summation order
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, index1, keys, (const int *)iota, csc_pair, M, 0, key_bits(NK), st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(csc_finish_kernel, dim3(div_up(M, 256)), dim3(256), 0, st, NK, M, keys, csc_pair, index0, csc_offsets, csc_query);
    check_launch();
}

// Diagnostic only (tools/interference.py; not part of the header): `blocks` workgroups of 1024 threads with 128 VGPRs' worth of
// occupancy hold their CUs for `micros` microseconds; mode 0 spins in registers, 1 polls `word` with device-scope atomic loads,
// 2 also stores to / loads from `word + 64...` with device-scope accesses every ~10 us.
void pointops2_diag_hold_cus_launcher(int blocks, int micros, int mode, unsigned *word, int lds_kb) {
    allow_big_lds(diag_hold_kernel, (size_t)lds_kb * 1024);  // (120 KB: no cell workgroup fits beside it)
    hipLaunchKernelGGL(diag_hold_kernel, dim3(blocks), dim3(1024), (size_t)lds_kb * 1024, state().stream, micros, mode, word);
}

}  // extern "C"
