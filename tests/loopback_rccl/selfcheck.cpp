// TEST INFRASTRUCTURE: checks the loopback collective double (loopback_rccl.hip) by itself, without libminarrow_hip.so, on
// one GPU. What it rehearses is the combine step of the reference's partitioned reduction (benches/benchmark_parallel_simd.rs:81-98).
//
//   selfcheck queues K [C]    how many of K streams share a hardware queue with a held one (GPU_MAX_HW_QUEUES in effect),
//                             after C streams were created, used and destroyed
//   selfcheck slots K [0|1]   K - 1 queues busy (1: spinning kernels, 0: value waits): does a K-th queue still get to run?
//   selfcheck single N ITERS  ncclCommInitAll over N ranks on device 0, one host thread per rank, ITERS all-gathers + all-reduces
//   selfcheck grouped N ITERS the same from ONE thread inside ncclGroupStart/End
//   selfcheck abort N         rank N-1 never posts: the others block on the GPU; ncclCommAbort ends them within a second
//   selfcheck procs N ITERS   N processes (forked before any HIP call), ncclCommInitRank over a shared segment
// Exit code 0 and a line "ok ..." on success.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define HIP_OK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_));      \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)
#define NCCL_OK(call)                                                                                 \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess) {                                                                      \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, ncclGetErrorString(r_));     \
            exit(2);                                                                                  \
        }                                                                                             \
    } while (0)

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static bool drained(hipStream_t s, double limit_ms) {
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(s) != hipSuccess) {
        (void)hipGetLastError();
        if (ms_since(t0) > limit_ms) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

constexpr size_t kWords = 128;  // 1 KiB per rank: the size of a group's record block

static uint64_t tag(int rank, int iter, size_t w) { return ((uint64_t)(rank + 1) << 48) ^ ((uint64_t)iter << 16) ^ w; }

struct RankState {
    ncclComm_t comm;
    hipStream_t stream;
    uint64_t *send, *recv, *sum, *h_send, *h_recv, *h_sum;
};

static void rank_setup(RankState& r, int n) {
    HIP_OK(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
    HIP_OK(hipMalloc((void**)&r.send, kWords * 8));
    HIP_OK(hipMalloc((void**)&r.recv, (size_t)n * kWords * 8));
    HIP_OK(hipMalloc((void**)&r.sum, kWords * 8));
    HIP_OK(hipHostMalloc((void**)&r.h_send, 4 * kWords * 8, hipHostMallocDefault));  // four iterations may be in flight
    HIP_OK(hipHostMalloc((void**)&r.h_recv, (size_t)n * kWords * 8, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&r.h_sum, kWords * 8, hipHostMallocDefault));
}

// one iteration's operations of one rank; `check` after the stream has drained
static void rank_post(RankState& r, int rank, int n, int iter) {
    uint64_t* h = r.h_send + (size_t)(iter & 3) * kWords;
    for (size_t w = 0; w < kWords; ++w) h[w] = tag(rank, iter, w);
    HIP_OK(hipMemcpyAsync(r.send, h, kWords * 8, hipMemcpyHostToDevice, r.stream));
    NCCL_OK(ncclAllGather(r.send, r.recv, kWords * 8, ncclChar, r.comm, r.stream));
    NCCL_OK(ncclAllReduce(r.send, r.sum, kWords, ncclInt64, ncclSum, r.comm, r.stream));
}

// ... and the read-back (behind ncclGroupEnd when the collectives were posted inside a group: that is where they are enqueued)
static void rank_read_back(RankState& r, int n) {
    HIP_OK(hipMemcpyAsync(r.h_recv, r.recv, (size_t)n * kWords * 8, hipMemcpyDeviceToHost, r.stream));
    HIP_OK(hipMemcpyAsync(r.h_sum, r.sum, kWords * 8, hipMemcpyDeviceToHost, r.stream));
}

static bool rank_check(RankState& r, int n, int iter) {
    for (int q = 0; q < n; ++q)
        for (size_t w = 0; w < kWords; ++w)
            if (r.h_recv[(size_t)q * kWords + w] != tag(q, iter, w)) return false;
    for (size_t w = 0; w < kWords; ++w) {
        uint64_t s = 0;
        for (int q = 0; q < n; ++q) s += tag(q, iter, w);
        if (r.h_sum[w] != s) return false;
    }
    return true;
}

static int mode_queues(int k, int churn) {
    HIP_OK(hipSetDevice(0));
    // streams created, used once and destroyed beforehand: does the runtime hand their hardware queues back?
    for (int done = 0; done < churn; done += 24) {
        std::vector<hipStream_t> t(24);
        uint64_t* w = nullptr;
        HIP_OK(hipHostMalloc((void**)&w, 24 * 8, hipHostMallocMapped | hipHostMallocCoherent));
        for (auto& x : t) HIP_OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (size_t j = 0; j < t.size(); ++j) HIP_OK(hipStreamWriteValue64(t[j], w + j, 1, 0));
        for (auto& x : t) HIP_OK(hipStreamSynchronize(x));
        for (auto& x : t) HIP_OK(hipStreamDestroy(x));
        HIP_OK(hipHostFree(w));
    }
    std::vector<hipStream_t> s((size_t)k);
    for (auto& x : s) HIP_OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    uint64_t* words = nullptr;  // pinned host memory: [0] holds stream 0, [j] is written by stream j
    HIP_OK(hipHostMalloc((void**)&words, (size_t)(k + 1) * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(words, 0, (size_t)(k + 1) * 8);
    HIP_OK(hipStreamWaitValue64(s[0], words, 1, hipStreamWaitValueGte, ~(uint64_t)0));
    for (int j = 1; j < k; ++j) HIP_OK(hipStreamWriteValue64(s[(size_t)j], words + j, 1, 0));
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    int behind = 0;
    for (int j = 1; j < k; ++j) behind += __atomic_load_n(words + j, __ATOMIC_ACQUIRE) == 0;
    __atomic_store_n(words, (uint64_t)1, __ATOMIC_RELEASE);  // release stream 0
    for (auto& x : s)
        if (!drained(x, 5000.0)) {
            fprintf(stderr, "a stream did not drain after the release\n");
            return 1;
        }
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("ok queues: %d streams (after %d created and destroyed), GPU_MAX_HW_QUEUES=%s: %d of %d sit behind the held stream's hardware queue\n",
           k, churn, q ? q : "unset", behind, k - 1);
    return 0;
}

__global__ void spin_until(const uint64_t* flag, uint64_t* seen) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) __builtin_amdgcn_s_sleep(32);
    *seen = 1;
}
__global__ void set_flag(uint64_t* flag) { __hip_atomic_store(flag, (uint64_t)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// K - 1 streams (a hardware queue each) hold a kernel that spins — or a hipStreamWaitValue64 — on a flag that a kernel on the K-th
// stream, enqueued LAST, sets: does the K-th queue get a hardware slot while the others are all busy? (the limit of a rehearsal)
static int mode_slots(int k, bool spin) {
    HIP_OK(hipSetDevice(0));
    std::vector<hipStream_t> s((size_t)k);
    for (auto& x : s) HIP_OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    uint64_t* w = nullptr;
    HIP_OK(hipHostMalloc((void**)&w, (size_t)(k + 1) * 8, hipHostMallocMapped | hipHostMallocCoherent));
    memset(w, 0, (size_t)(k + 1) * 8);
    for (int j = 0; j + 1 < k; ++j) {
        if (spin) hipLaunchKernelGGL(spin_until, dim3(1), dim3(1), 0, s[(size_t)j], w, w + 1 + j);
        else HIP_OK(hipStreamWaitValue64(s[(size_t)j], w, 1, hipStreamWaitValueGte, ~(uint64_t)0));
    }
    HIP_OK(hipGetLastError());
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s[(size_t)k - 1], w);
    HIP_OK(hipGetLastError());
    bool all = true;
    for (auto& x : s) all = drained(x, 3000.0) && all;
    const double ms = ms_since(t0);
    if (!all) {
        __atomic_store_n(w, (uint64_t)1, __ATOMIC_RELEASE);  // the host releases what the last queue could not
        for (auto& x : s) (void)drained(x, 5000.0);
    }
    printf("ok slots: %d queues busy with %s, the releasing kernel on queue %d %s (%.1f ms)\n", k - 1, spin ? "spinning kernels" : "value waits", k,
           all ? "RAN" : "NEVER ran within 3 s: more queues than hardware slots starve", ms);
    return 0;
}

static int mode_single(int n, int iters, bool grouped) {
    HIP_OK(hipSetDevice(0));
    // SELFCHECK_IDLE_STREAMS=E: E more streams (each used once, so that its hardware queue exists) stay alive beside the ranks'
    static std::vector<hipStream_t> idle(getenv("SELFCHECK_IDLE_STREAMS") ? (size_t)atoi(getenv("SELFCHECK_IDLE_STREAMS")) : 0);
    if (!idle.empty()) {
        uint64_t* w = nullptr;
        HIP_OK(hipHostMalloc((void**)&w, idle.size() * 8, hipHostMallocMapped | hipHostMallocCoherent));
        for (auto& x : idle) HIP_OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (size_t j = 0; j < idle.size(); ++j) HIP_OK(hipStreamWriteValue64(idle[j], w + j, 1, 0));
        for (auto& x : idle) HIP_OK(hipStreamSynchronize(x));
    }
    std::vector<ncclComm_t> comms((size_t)n);
    std::vector<int> devs((size_t)n, 0);
    NCCL_OK(ncclCommInitAll(comms.data(), n, devs.data()));
    std::vector<RankState> ranks((size_t)n);
    for (int r = 0; r < n; ++r) {
        ranks[(size_t)r].comm = comms[(size_t)r];
        rank_setup(ranks[(size_t)r], n);
    }
    std::atomic<int> bad{0};
    const auto t0 = std::chrono::steady_clock::now();
    if (grouped) {
        for (int it = 0; it < iters; ++it) {
            NCCL_OK(ncclGroupStart());
            for (int r = 0; r < n; ++r) rank_post(ranks[(size_t)r], r, n, it);
            NCCL_OK(ncclGroupEnd());
            for (int r = 0; r < n; ++r) rank_read_back(ranks[(size_t)r], n);
            for (int r = 0; r < n; ++r) {
                if (!drained(ranks[(size_t)r].stream, 20000.0)) {
                    fprintf(stderr, "rank %d stuck at iteration %d\n", r, it);
                    return 1;
                }
                if (!rank_check(ranks[(size_t)r], n, it)) ++bad;
            }
        }
    } else {
        std::vector<std::thread> th;
        for (int r = 0; r < n; ++r)
            th.emplace_back([&, r] {
                HIP_OK(hipSetDevice(0));
                for (int it = 0; it < iters; ++it) {
                    rank_post(ranks[(size_t)r], r, n, it);
                    rank_read_back(ranks[(size_t)r], n);
                    if (it % 3 == 2 || it == iters - 1) {  // two iterations in flight in between: the staging slots alternate
                        if (!drained(ranks[(size_t)r].stream, 20000.0)) {
                            fprintf(stderr, "rank %d stuck at iteration %d\n", r, it);
                            exit(1);
                        }
                        if (!rank_check(ranks[(size_t)r], n, it)) ++bad;
                    }
                }
            });
        for (auto& t : th) t.join();
    }
    const double ms = ms_since(t0);
    int count = 0;
    NCCL_OK(ncclCommCount(comms[0], &count));
    for (auto c : comms) NCCL_OK(ncclCommDestroy(c));
    if (bad.load() || count != n) {
        fprintf(stderr, "%d checks failed (count %d)\n", bad.load(), count);
        return 1;
    }
    printf("ok %s: %d ranks x %d iterations (all-gather + all-reduce each), %.1f us per iteration (%zu idle streams beside them)\n",
           grouped ? "grouped" : "single", n, iters, ms * 1e3 / iters, idle.size());
    return 0;
}

static int mode_abort(int n) {
    HIP_OK(hipSetDevice(0));
    std::vector<ncclComm_t> comms((size_t)n);
    std::vector<int> devs((size_t)n, 0);
    NCCL_OK(ncclCommInitAll(comms.data(), n, devs.data()));
    std::vector<RankState> ranks((size_t)n);
    for (int r = 0; r < n; ++r) {
        ranks[(size_t)r].comm = comms[(size_t)r];
        rank_setup(ranks[(size_t)r], n);
    }
    for (int r = 0; r < n; ++r) {  // a complete round first
        rank_post(ranks[(size_t)r], r, n, 0);
        rank_read_back(ranks[(size_t)r], n);
    }
    for (int r = 0; r < n; ++r)
        if (!drained(ranks[(size_t)r].stream, 20000.0) || !rank_check(ranks[(size_t)r], n, 0)) return 1;
    for (int r = 0; r + 1 < n; ++r) rank_post(ranks[(size_t)r], r, n, 1);  // the last rank never arrives
    std::this_thread::sleep_for(std::chrono::milliseconds(300));
    for (int r = 0; r + 1 < n; ++r)
        if (hipStreamQuery(ranks[(size_t)r].stream) == hipSuccess) {
            fprintf(stderr, "rank %d finished a collective its peer never joined\n", r);
            return 1;
        }
    (void)hipGetLastError();
    const auto t0 = std::chrono::steady_clock::now();
    for (auto c : comms) NCCL_OK(ncclCommAbort(c));
    for (int r = 0; r < n; ++r)
        if (!drained(ranks[(size_t)r].stream, 5000.0)) {
            fprintf(stderr, "rank %d still busy after the abort\n", r);
            return 1;
        }
    printf("ok abort: %d ranks blocked on the missing one for 300 ms; aborted and drained in %.1f ms\n", n - 1, ms_since(t0));
    return 0;
}

static int mode_procs(int n, int iters) {
    ncclUniqueId id;
    NCCL_OK(ncclGetUniqueId(&id));  // touches no GPU: the parent stays HIP-free and only waits
    std::vector<pid_t> kids;
    for (int r = 0; r < n; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            HIP_OK(hipSetDevice(0));
            RankState st;
            NCCL_OK(ncclCommInitRank(&st.comm, n, id, r));
            rank_setup(st, n);
            int bad = 0;
            for (int it = 0; it < iters; ++it) {
                rank_post(st, r, n, it);
                rank_read_back(st, n);
                if (!drained(st.stream, 20000.0)) {
                    fprintf(stderr, "process rank %d stuck at iteration %d\n", r, it);
                    _exit(1);
                }
                bad += !rank_check(st, n, it);
            }
            NCCL_OK(ncclCommDestroy(st.comm));
            _exit(bad ? 1 : 0);
        }
        kids.push_back(pid);
    }
    int failed = 0;
    for (pid_t k : kids) {
        int status = 0;
        waitpid(k, &status, 0);
        failed += !(WIFEXITED(status) && WEXITSTATUS(status) == 0);
    }
    if (failed) {
        fprintf(stderr, "%d of %d rank processes failed\n", failed, n);
        return 1;
    }
    printf("ok procs: %d processes x %d iterations over a shared segment\n", n, iters);
    return 0;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "";
    const int a = argc > 2 ? atoi(argv[2]) : 2, b = argc > 3 ? atoi(argv[3]) : 50;
    if (!strcmp(mode, "queues")) return mode_queues(a, argc > 3 ? b : 0);
    if (!strcmp(mode, "slots")) return mode_slots(a, argc > 3 ? b != 0 : true);
    if (!strcmp(mode, "single")) return mode_single(a, b, false);
    if (!strcmp(mode, "grouped")) return mode_single(a, b, true);
    if (!strcmp(mode, "abort")) return mode_abort(a);
    if (!strcmp(mode, "procs")) return mode_procs(a, b);
    fprintf(stderr, "usage: selfcheck queues K | single N ITERS | grouped N ITERS | abort N | procs N ITERS\n");
    return 64;
}
