// TEST INFRASTRUCTURE — not part of the product, never linked into libminarrow_hip.so.
//
// A loopback stand-in for librccl.so with collective LIVENESS semantics on ONE device, so that every multi-rank branch of
// ma_group_* / ma_comm_* (eight issue threads each blocked in its own all-gather until the peers arrive, stamp waits in front
// of a collective that has peers, a multi-member abort and rebuild, communicators across processes) runs on a one-GPU box.
// The library opens it instead of RCCL when MINARROW_HIP_RCCL_PATH names it (minarrow_amd/csrc/ma_rccl.hip: load_rccl) and says so
// (ma_rccl_path, ma_group_exchange_note); bench.py labels every line produced through it REHEARSAL. The job being rehearsed is
// the combine step of the reference's partitioned reduction, benches/benchmark_parallel_simd.rs:81-98.
//
// It exports the twelve RCCL entry points load_rccl() resolves plus ncclLoopbackDoubleInfo (how the library tells the double
// from a fabric: only the double accepts several ranks on one device).
//
// How a collective behaves here — as on a fabric:
//   * the host call returns at once: ONE kernel (one workgroup) goes onto the caller's stream, like RCCL's;
//   * that kernel copies the rank's `send` into the staging slot of this operation, publishes the operation's sequence number
//     in the rank's ARRIVAL word, then SPINS until every rank's arrival word has reached the sequence — a rank whose peer never
//     enqueues its collective (or whose peer's stream is held in front of it) blocks on the GPU exactly as it would over xGMI —
//     and copies the staged blocks into `recv` in rank order (all-gather) or adds them in rank order (all-reduce, wrapping);
//   * ncclCommAbort stores the rank's ABORT word from the host: the spinning kernel (and any kernel of that communicator still
//     queued) ends without touching `recv`, and the communicator is dead; peers keep waiting until they are aborted too.
//
// Where things live. Arrival words, abort words and the staging slots sit in HOST-COHERENT memory: hipHostMalloc for
// ncclCommInitAll (one process), a POSIX shared-memory segment registered with hipHostRegister in every process for
// ncclCommInitRank (the segment's name travels in the ncclUniqueId; the ranks rendezvous on a join counter in it). Device
// memory shared over hipIpc would be the fabric-like choice, but two kernels that run at the same time on different XCDs are
// not coherent through coarse-grained device memory (each XCD has its own L2); fine-grained host memory is, and the
// payloads here are reduction records (a few hundred bytes per rank).
//
// One device, many ranks: every rank's kernel spins while it waits, and streams that share a hardware queue run in order. The
// streams that carry the ranks' collectives must therefore each have a hardware queue of their own. The library sees to it: in
// a rehearsal it puts those streams (the members' and their exchange streams) into the high priority class, whose queue pool
// they have to themselves — the process needs GPU_MAX_HW_QUEUES >= ranks (the runtime's default is 4; ncclCommInitAll says so on
// stderr when it is not) and no more than that: a device runs 23 hardware queues at a time (selfcheck `slots`), and with more
// alive, work that waits on work in another queue crawls from time slice to time slice.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdarg>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;
constexpr uint64_t kMagic = 0x4C425243434C3031ull;  // "LBRCCL01"
constexpr int kVersion = 9900;                       // what ncclGetVersion reports: no RCCL release has this code
constexpr size_t kLine = 64;

// The control block of one world of ranks. Host-coherent memory; every multi-writer field is a whole 64-byte line.
struct alignas(64) Control {
    uint64_t magic;
    uint32_t n_ranks;
    uint32_t slot_bytes;  // staging bytes per rank and operation
    uint8_t pad0[kLine - 16];
    alignas(64) std::atomic<uint32_t> joined;  // ranks that have mapped the segment (ncclCommInitRank)
    alignas(64) std::atomic<uint32_t> gone;    // ranks that have destroyed / aborted their communicator
    struct alignas(64) Word {
        uint64_t v;
        uint8_t pad[kLine - 8];
    };
    Word arrival[kMaxRanks];  // rank r's: the sequence of the last operation r has staged its block for
    Word abort[kMaxRanks];    // rank r's: non-zero once r's communicator was aborted
    // followed by the staging area: [2][n_ranks][slot_bytes]
};

size_t world_bytes(uint32_t n_ranks, uint32_t slot_bytes) { return sizeof(Control) + (size_t)2 * n_ranks * slot_bytes; }

uint32_t slot_bytes_from_env() {
    const char* e = getenv("LOOPBACK_RCCL_SLOT_BYTES");
    long v = e ? atol(e) : 0;
    if (v <= 0) v = 64 << 10;
    return (uint32_t)((v + 63) & ~63l);
}

void say(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void say(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[loopback_rccl] %s\n", buf);
    fflush(stderr);
}

struct World {
    Control* host = nullptr;  // the host's view
    Control* dev = nullptr;   // the device's view of the same memory
    size_t bytes = 0;
    bool shm = false;
    std::atomic<int> comms{0};     // communicators of THIS process still alive
    std::atomic<bool> leak{false};  // an operation of a dropped communicator may still be in flight: the memory stays
};

}  // namespace

struct ncclComm {
    std::shared_ptr<World> world;
    int rank = 0, n_ranks = 1, device = 0;
    uint64_t seq = 0;           // operations posted so far
    hipStream_t last_stream = nullptr;
    hipEvent_t last_event = nullptr;  // recorded behind every operation: orders operations that move to another stream
    bool have_last = false;
    bool dead = false;
};

namespace {

enum Op : int { kGather = 0, kReduceI64 = 1, kReduceF64 = 2 };

struct Args {
    Control* c;
    const uint8_t* send;
    uint8_t* recv;
    uint64_t seq;
    uint32_t bytes;  // per rank
    int rank, n_ranks, op;
};

__device__ inline uint64_t load_sys(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

__device__ inline void copy_bytes(uint8_t* dst, const uint8_t* src, size_t n) {
    if ((((uintptr_t)dst | (uintptr_t)src | n) & 7) == 0) {
        for (size_t i = threadIdx.x; i < n / 8; i += blockDim.x) ((uint64_t*)dst)[i] = ((const uint64_t*)src)[i];
    } else {
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    }
}

// One workgroup per collective and rank. Every path reaches the end: the spin leaves on arrival of all ranks OR on the rank's
// abort word, which ncclCommAbort / ncclCommDestroy store from the host.
__global__ void __launch_bounds__(256) collective_kernel(Args a) {
    Control* c = a.c;
    __shared__ int aborted;
    if (threadIdx.x == 0) aborted = load_sys(&c->abort[a.rank].v) != 0;
    __syncthreads();
    if (aborted) return;
    uint8_t* staging = (uint8_t*)(c + 1) + (size_t)(a.seq & 1) * a.n_ranks * c->slot_bytes;
    copy_bytes(staging + (size_t)a.rank * c->slot_bytes, a.send, a.bytes);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&c->arrival[a.rank].v, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < a.n_ranks) {
        const uint64_t* peer = &c->arrival[threadIdx.x].v;
        while (load_sys(peer) < a.seq) {
            if (load_sys(&c->abort[a.rank].v) != 0) {
                aborted = 1;  // benign race: every writer stores 1
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    if (aborted) return;
    __threadfence_system();
    if (a.op == kGather) {
        for (int r = 0; r < a.n_ranks; ++r) copy_bytes(a.recv + (size_t)r * a.bytes, staging + (size_t)r * c->slot_bytes, a.bytes);
    } else if (a.op == kReduceI64) {
        for (size_t i = threadIdx.x; i < a.bytes / 8; i += blockDim.x) {
            uint64_t s = 0;  // wrapping, rank order
            for (int r = 0; r < a.n_ranks; ++r) s += ((const uint64_t*)(staging + (size_t)r * c->slot_bytes))[i];
            ((uint64_t*)a.recv)[i] = s;
        }
    } else {
        for (size_t i = threadIdx.x; i < a.bytes / 8; i += blockDim.x) {
            double s = 0.0;
            for (int r = 0; r < a.n_ranks; ++r) s += ((const double*)(staging + (size_t)r * c->slot_bytes))[i];
            ((double*)a.recv)[i] = s;
        }
    }
}

// ---- group calls: collectives posted between ncclGroupStart and the matching ncclGroupEnd are launched by the latter --------
thread_local int t_group_depth = 0;
thread_local std::vector<std::function<ncclResult_t()>> t_deferred;

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t launch(ncclComm* comm, Args a, hipStream_t stream) {
    int before = 0;
    if (hipGetDevice(&before) != hipSuccess) return ncclUnhandledCudaError;
    if (before != comm->device && hipSetDevice(comm->device) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = ncclSuccess;
    // RCCL serialises a communicator's operations whatever streams they are given: so does this
    if (comm->have_last && comm->last_stream != stream && hipStreamWaitEvent(stream, comm->last_event, 0) != hipSuccess)
        r = ncclUnhandledCudaError;
    if (r == ncclSuccess) {
        hipLaunchKernelGGL(collective_kernel, dim3(1), dim3(256), 0, stream, a);
        if (hipGetLastError() != hipSuccess || hipEventRecord(comm->last_event, stream) != hipSuccess) r = ncclUnhandledCudaError;
    }
    comm->last_stream = stream;
    comm->have_last = r == ncclSuccess;
    if (before != comm->device) (void)hipSetDevice(before);
    if (r != ncclSuccess) (void)hipGetLastError();
    return r;
}

ncclResult_t post(ncclComm* comm, int op, const void* send, void* recv, size_t bytes, hipStream_t stream) {
    if (!comm || comm->dead) {
        say("a collective on a communicator that was aborted or destroyed");
        return ncclInvalidUsage;
    }
    if (bytes == 0) return ncclSuccess;
    if (!send || !recv) return ncclInvalidArgument;
    if (bytes > comm->world->host->slot_bytes) {
        say("%zu bytes per rank exceed the staging slot (%u bytes: LOOPBACK_RCCL_SLOT_BYTES)", bytes, comm->world->host->slot_bytes);
        return ncclInvalidArgument;
    }
    Args a{comm->world->dev, (const uint8_t*)send, (uint8_t*)recv, ++comm->seq, (uint32_t)bytes, comm->rank, comm->n_ranks, op};
    if (t_group_depth > 0) {
        t_deferred.push_back([comm, a, stream] { return launch(comm, a, stream); });
        return ncclSuccess;
    }
    return launch(comm, a, stream);
}

void init_control(Control* c, uint32_t n_ranks, uint32_t slot_bytes) {
    memset((void*)c, 0, sizeof(Control));
    c->n_ranks = n_ranks;
    c->slot_bytes = slot_bytes;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    c->magic = kMagic;
}

// Blocks of one-process worlds that are no longer in use, by size: handed out again instead of freed. hipHostFree waits for the
// DEVICE to run empty, and a communicator is often dropped exactly when some stream of its host is stuck (an abort).
std::mutex g_blocks_mu;
std::vector<std::pair<size_t, void*>> g_blocks;

std::shared_ptr<World> make_local_world(int n_ranks) {
    auto w = std::make_shared<World>();
    const uint32_t slot = slot_bytes_from_env();
    w->bytes = world_bytes((uint32_t)n_ranks, slot);
    void* p = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_blocks_mu);
        for (size_t i = 0; i < g_blocks.size(); ++i)
            if (g_blocks[i].first == w->bytes) {
                p = g_blocks[i].second;
                g_blocks.erase(g_blocks.begin() + (long)i);
                break;
            }
    }
    if (!p && hipHostMalloc(&p, w->bytes, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    w->host = w->dev = (Control*)p;
    init_control(w->host, (uint32_t)n_ranks, slot);
    w->host->joined.store((uint32_t)n_ranks);
    return w;
}

// The shared world of a multi-process communicator: the first rank to arrive creates and sizes the segment, every rank maps
// and registers it, then all wait for each other on the join counter.
struct IdPayload {
    uint64_t magic;
    char name[64];
    uint32_t slot_bytes;
};

std::shared_ptr<World> join_shared_world(const IdPayload& id, int n_ranks, int rank) {
    auto w = std::make_shared<World>();
    w->shm = true;
    w->bytes = world_bytes((uint32_t)n_ranks, id.slot_bytes);
    bool creator = true;
    int fd = shm_open(id.name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) {
        creator = false;
        fd = shm_open(id.name, O_RDWR, 0600);
    }
    if (fd < 0) {
        say("rank %d: shm_open(%s): %s", rank, id.name, strerror(errno));
        return nullptr;
    }
    const double limit_s = getenv("LOOPBACK_RCCL_INIT_TIMEOUT_S") ? atof(getenv("LOOPBACK_RCCL_INIT_TIMEOUT_S")) : 120.0;
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s; };
    if (creator) {
        if (ftruncate(fd, (off_t)w->bytes) != 0) {
            say("rank %d: ftruncate: %s", rank, strerror(errno));
            close(fd);
            return nullptr;
        }
    } else {
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < w->bytes) {
            if (late()) {
                say("rank %d: the segment %s was never sized", rank, id.name);
                close(fd);
                return nullptr;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void* p = mmap(nullptr, w->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        say("rank %d: mmap: %s", rank, strerror(errno));
        return nullptr;
    }
    w->host = (Control*)p;
    if (creator) init_control(w->host, (uint32_t)n_ranks, id.slot_bytes);
    while (((volatile Control*)w->host)->magic != kMagic) {
        if (late()) {
            say("rank %d: the segment %s was never initialised", rank, id.name);
            munmap(p, w->bytes);
            return nullptr;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (w->host->n_ranks != (uint32_t)n_ranks) {
        say("rank %d: the ranks disagree on the communicator's size (%u here %d)", rank, w->host->n_ranks, n_ranks);
        munmap(p, w->bytes);
        return nullptr;
    }
    void* d = nullptr;
    if (hipHostRegister(p, w->bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess ||
        hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
        say("rank %d: hipHostRegister of the shared segment failed: %s", rank, hipGetErrorString(hipGetLastError()));
        munmap(p, w->bytes);
        return nullptr;
    }
    w->dev = (Control*)d;
    const uint32_t mine = w->host->joined.fetch_add(1) + 1;
    if (mine == (uint32_t)n_ranks) shm_unlink(id.name);  // everyone has it mapped: the name can go
    while (w->host->joined.load() < (uint32_t)n_ranks) {
        if (late()) {
            say("rank %d: only %u of %d ranks joined within %.0f s", rank, w->host->joined.load(), n_ranks, limit_s);
            if (creator) shm_unlink(id.name);
            (void)hipHostUnregister(p);
            munmap(p, w->bytes);
            return nullptr;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return w;
}

// The world's memory goes when its last local communicator does — unless one of their operations may still be in flight
// (a kernel still queued behind something that was never released): then it is left to the process.
void drop(ncclComm* comm) {
    std::shared_ptr<World> w = comm->world;
    bool quiet = true;
    if (comm->have_last) {
        quiet = hipEventQuery(comm->last_event) == hipSuccess;
        (void)hipGetLastError();
    }
    if (quiet && comm->last_event) (void)hipEventDestroy(comm->last_event);
    w->host->gone.fetch_add(1);
    if (!quiet) w->leak.store(true);
    if (w->comms.fetch_sub(1) == 1) {
        if (!w->leak.load()) {
            if (w->shm) {
                (void)hipHostUnregister(w->host);
                munmap(w->host, w->bytes);
            } else {
                std::lock_guard<std::mutex> lock(g_blocks_mu);  // kept for the next world of this size (never hipHostFree: it waits for the device)
                g_blocks.emplace_back(w->bytes, (void*)w->host);
            }
            (void)hipGetLastError();
        }
        w->host = w->dev = nullptr;
    }
    comm->dead = true;
    delete comm;
}

ncclComm* make_comm(std::shared_ptr<World> w, int rank, int n_ranks, int device) {
    ncclComm* c = new ncclComm();
    c->world = std::move(w);
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->device = device;
    int before = 0;
    (void)hipGetDevice(&before);
    if (before != device) (void)hipSetDevice(device);
    if (hipEventCreateWithFlags(&c->last_event, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->last_event = nullptr;
    }
    if (before != device) (void)hipSetDevice(before);
    c->world->comms.fetch_add(1);
    return c;
}

std::atomic<uint64_t> g_id_counter{0};

}  // namespace

extern "C" {

// Only the double has this symbol: "ranks may share a device".
const char* ncclLoopbackDoubleInfo(void) {
    return "loopback collective double (test infrastructure): liveness semantics of RCCL on one device, host-coherent staging";
}

ncclResult_t ncclGetVersion(int* version) {
    if (!version) return ncclInvalidArgument;
    *version = kVersion;
    return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* out) {
    if (!out) return ncclInvalidArgument;
    static_assert(sizeof(IdPayload) <= NCCL_UNIQUE_ID_BYTES, "the id carries the segment's name");
    IdPayload id;
    memset(&id, 0, sizeof(id));
    id.magic = kMagic;
    id.slot_bytes = slot_bytes_from_env();
    const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    snprintf(id.name, sizeof(id.name), "/lbrccl-%d-%llu-%llx", (int)getpid(), (unsigned long long)g_id_counter.fetch_add(1),
             (unsigned long long)(t & 0xFFFFFFFFFFull));
    memset(out->internal, 0, NCCL_UNIQUE_ID_BYTES);
    memcpy(out->internal, &id, sizeof(id));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    *comm = nullptr;
    IdPayload id;
    memcpy(&id, commId.internal, sizeof(id));
    if (id.magic != kMagic) {
        say("ncclCommInitRank: the id was not made by this library's ncclGetUniqueId");
        return ncclInvalidArgument;
    }
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return ncclUnhandledCudaError;
    std::shared_ptr<World> w = nranks == 1 ? make_local_world(1) : join_shared_world(id, nranks, rank);
    if (!w) return ncclSystemError;
    *comm = make_comm(w, rank, nranks, device);
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1 || ndev > kMaxRanks) return ncclInvalidArgument;
    bool shared_device = false;
    for (int i = 0; i < ndev; ++i)
        for (int j = i + 1; j < ndev; ++j) shared_device |= (devlist ? devlist[i] : i) == (devlist ? devlist[j] : j);
    if (shared_device) {
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        if ((q ? atoi(q) : 4) < ndev)
            say("%d ranks share a device and GPU_MAX_HW_QUEUES is %s: streams that share a hardware queue run in order, and a rank "
                "whose peer's kernel sits behind its own spinning kernel never finishes. Start the process with GPU_MAX_HW_QUEUES >= %d",
                ndev, q ? q : "unset (4)", ndev);
    }
    std::shared_ptr<World> w = make_local_world(ndev);
    if (!w) return ncclUnhandledCudaError;
    for (int i = 0; i < ndev; ++i) comms[i] = make_comm(w, i, ndev, devlist ? devlist[i] : i);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    // anything of this communicator still spinning ends: a destroyed rank does not arrive any more
    comm->world->host->abort[comm->rank].v = 1;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    drop(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    Control* c = comm->world->host;
    __atomic_store_n(&c->abort[comm->rank].v, (uint64_t)1, __ATOMIC_SEQ_CST);  // host memory: no GPU queue is involved
    // like RCCL's, this returns once the communicator's own work has ended — bounded here (2 s): the caller may hold the
    // stream behind something else
    if (comm->have_last) {
        const auto t0 = std::chrono::steady_clock::now();
        while (hipEventQuery(comm->last_event) != hipSuccess) {
            (void)hipGetLastError();
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
    }
    drop(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->n_ranks;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t w = dtype_bytes(datatype);
    if (!w) return ncclInvalidArgument;
    return post(comm, kGather, send, recv, sendcount * w, stream);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    if (op != ncclSum || (datatype != ncclInt64 && datatype != ncclUint64 && datatype != ncclFloat64)) {
        say("ncclAllReduce: only sums of 64-bit integers and doubles are rehearsed here");
        return ncclInvalidArgument;
    }
    return post(comm, datatype == ncclFloat64 ? kReduceF64 : kReduceI64, send, recv, count * 8, stream);
}

ncclResult_t ncclGroupStart() {
    ++t_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    if (--t_group_depth > 0) return ncclSuccess;
    ncclResult_t first = ncclSuccess;
    for (auto& fn : t_deferred) {
        const ncclResult_t r = fn();
        if (first == ncclSuccess) first = r;
    }
    t_deferred.clear();
    return first;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (loopback double)";
        case ncclSystemError: return "system error (loopback double: shared segment or rendezvous)";
        case ncclInternalError: return "internal error (loopback double)";
        case ncclInvalidArgument: return "invalid argument (loopback double)";
        case ncclInvalidUsage: return "invalid usage (loopback double)";
        default: return "error (loopback double)";
    }
}

}  // extern "C"
