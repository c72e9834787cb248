// Chunk-pipelined staging of host-resident operands: H2D of tile k+1 ‖ kernels of tile k ‖ D2H of tile k-1.
//
// A Rust `&[T]` / `Vec64<T>` that was not allocated through ma_alloc64_pinned is pageable host memory. Staging a
// whole operand (CallScope) costs a device allocation of its full size and runs copy-in, kernel and copy-out one
// after the other, so an elementwise call moves 3 operands over a half-used PCIe link. Here the operands cross in
// tiles through a small context-owned ring of device buffers: the calling thread feeds the H2D stream, the kernels
// run on the context's stream, and a helper thread drains results on a D2H stream, so both directions of the link
// are busy at once and the device footprint is the ring (kSlots x operands x tile), whatever the column size.
//
// The reference has nothing to compare with (its kernels read host memory in place, src/kernels/arithmetic/
// dispatch.rs:74-133); this is purely the cost model of the boundary for callers that keep columns on the host.
#include <condition_variable>
#include <deque>
#include <exception>
#include <string>
#include <thread>

#include "ma_common.hpp"

namespace ma {

namespace {

constexpr int kSlots = 3;

struct Pipe {
    hipStream_t h2d = nullptr;
    hipStream_t d2h = nullptr;
    hipEvent_t ev_in[kSlots] = {};
    hipEvent_t ev_done[kSlots] = {};
    void* buf[kSlots][kMaxPipeOperands] = {};
    size_t buf_bytes = 0;  // size of every ring buffer
};

void pipe_free_buffers(Pipe* p) {
    for (int s = 0; s < kSlots; ++s)
        for (int i = 0; i < kMaxPipeOperands; ++i)
            if (p->buf[s][i]) {
                (void)hipFree(p->buf[s][i]);
                p->buf[s][i] = nullptr;
            }
    p->buf_bytes = 0;
}

ma_status pipe_acquire(ma_ctx* ctx, size_t buf_bytes, Pipe** out) {
    Pipe* p = (Pipe*)ctx->pipe;
    if (!p) {
        p = new Pipe();
        ctx->pipe = p;
        MA_HIP(hipStreamCreateWithFlags(&p->h2d, hipStreamNonBlocking));
        MA_HIP(hipStreamCreateWithFlags(&p->d2h, hipStreamNonBlocking));
        for (int s = 0; s < kSlots; ++s) {
            MA_HIP(hipEventCreateWithFlags(&p->ev_in[s], hipEventDisableTiming));
            MA_HIP(hipEventCreateWithFlags(&p->ev_done[s], hipEventDisableTiming));
        }
    }
    if (buf_bytes > p->buf_bytes) {
        MA_HIP(hipStreamSynchronize(ctx->stream));  // an earlier call's kernels may still read the old ring
        pipe_free_buffers(p);
        for (int s = 0; s < kSlots; ++s)
            for (int i = 0; i < kMaxPipeOperands; ++i) MA_HIP(device_malloc(ctx->device, &p->buf[s][i], buf_bytes));
        p->buf_bytes = buf_bytes;
    }
    *out = p;
    return MA_OK;
}

// What the helper thread and the caller share.
struct Drain {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<long> queue;  // tile indices whose kernels are enqueued; -1 = no more
    long done = 0;           // tiles whose results have landed in host memory
    ma_status status = MA_OK;
    std::string error;
};

}  // namespace

void pipe_destroy(ma_ctx* ctx) {
    Pipe* p = (Pipe*)ctx->pipe;
    if (!p) return;
    pipe_free_buffers(p);
    for (int s = 0; s < kSlots; ++s) {
        if (p->ev_in[s]) (void)hipEventDestroy(p->ev_in[s]);
        if (p->ev_done[s]) (void)hipEventDestroy(p->ev_done[s]);
    }
    if (p->h2d) (void)hipStreamDestroy(p->h2d);
    if (p->d2h) (void)hipStreamDestroy(p->d2h);
    delete p;
    ctx->pipe = nullptr;
}

ma_status run_tiled(ma_ctx* ctx, size_t n_rows, size_t tile_rows, const PipeOperand* ops, int n_ops, TileFn fn,
                    void* user) {
    MA_REQUIRE(n_ops > 0 && n_ops <= kMaxPipeOperands && tile_rows > 0, MA_ERR_INVALID_ARGUMENT, "bad tiling request");
    MA_NO_CAPTURE(ctx, "staging a pageable host operand");
    size_t widest = 0;
    bool any_out = false;
    for (int i = 0; i < n_ops; ++i)
        if (ops[i].staged) {
            widest = ops[i].elem_bytes > widest ? ops[i].elem_bytes : widest;
            any_out = any_out || ops[i].out != nullptr;
        }
    Pipe* p = nullptr;
    MA_TRY(pipe_acquire(ctx, tile_rows * widest, &p));
    const long n_tiles = (long)((n_rows + tile_rows - 1) / tile_rows);
    const int device = ctx->device;

    Drain drain;
    std::thread helper;
    if (any_out) try {
        helper = std::thread([&drain, p, ops, n_ops, tile_rows, n_rows, device]() {
            hipError_t e = hipSetDevice(device);
            for (;;) {
                long k;
                {
                    std::unique_lock<std::mutex> lk(drain.mu);
                    drain.cv.wait(lk, [&] { return !drain.queue.empty(); });
                    k = drain.queue.front();
                    drain.queue.pop_front();
                }
                if (k < 0) return;
                const int s = (int)(k % kSlots);
                const size_t row0 = (size_t)k * tile_rows;
                const size_t rows = n_rows - row0 < tile_rows ? n_rows - row0 : tile_rows;
                if (e == hipSuccess) e = hipStreamWaitEvent(p->d2h, p->ev_done[s], 0);
                for (int i = 0; i < n_ops && e == hipSuccess; ++i)
                    if (ops[i].staged && ops[i].out)
                        e = hipMemcpyAsync((char*)ops[i].out + row0 * ops[i].elem_bytes, p->buf[s][i],
                                           rows * ops[i].elem_bytes, hipMemcpyDeviceToHost, p->d2h);
                if (e == hipSuccess) e = hipStreamSynchronize(p->d2h);
                std::lock_guard<std::mutex> lk(drain.mu);
                if (e != hipSuccess && drain.status == MA_OK) {
                    drain.status = MA_ERR_DEVICE;
                    drain.error = std::string("HIP error in the D2H drain: ") + hipGetErrorString(e);
                    (void)hipGetLastError();
                }
                drain.done = k + 1;  // counted even after a failure so that the feeder never waits forever
                drain.cv.notify_all();
            }
        });
    } catch (const std::exception& e) {  // no thread to be had: nothing has been enqueued yet
        set_error("cannot start the D2H drain thread: %s", e.what());
        return MA_ERR_DEVICE;
    }
    auto finish = [&](ma_status s) -> ma_status {
        if (helper.joinable()) {
            {
                std::lock_guard<std::mutex> lk(drain.mu);
                drain.queue.push_back(-1);
            }
            drain.cv.notify_all();
            helper.join();
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);  // nothing of this call is in flight when it returns
        if (s == MA_OK && drain.status != MA_OK) {
            set_error("%s", drain.error.c_str());
            s = drain.status;
        }
        if (s == MA_OK && e != hipSuccess) s = hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
        return s;
    };
#define MA_PIPE_HIP(expr)                                                            \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        if (_e != hipSuccess) return finish(hip_fail(_e, #expr, __FILE__, __LINE__)); \
    } while (0)

    void* ptrs[kMaxPipeOperands];
    for (long k = 0; k < n_tiles; ++k) {
        const int s = (int)(k % kSlots);
        const size_t row0 = (size_t)k * tile_rows;
        const size_t rows = n_rows - row0 < tile_rows ? n_rows - row0 : tile_rows;
        if (k >= kSlots) {  // the slot is free once tile k - kSlots has left it
            if (any_out) {
                std::unique_lock<std::mutex> lk(drain.mu);
                drain.cv.wait(lk, [&] { return drain.done > k - kSlots; });
                if (drain.status != MA_OK) {
                    lk.unlock();
                    return finish(MA_OK);
                }
            } else {
                MA_PIPE_HIP(hipEventSynchronize(p->ev_done[s]));
            }
        }
        bool any_in = false;
        for (int i = 0; i < n_ops; ++i) {
            const PipeOperand& o = ops[i];
            if (!o.staged) {  // device-reachable (or absent: a scalar side): used where it lies
                char* base = o.out ? (char*)o.out : (char*)const_cast<void*>(o.in);
                ptrs[i] = base ? base + row0 * o.elem_bytes : nullptr;
                continue;
            }
            ptrs[i] = p->buf[s][i];
            if (o.in) {
                MA_PIPE_HIP(hipMemcpyAsync(p->buf[s][i], (const char*)o.in + row0 * o.elem_bytes, rows * o.elem_bytes,
                                           hipMemcpyHostToDevice, p->h2d));
                any_in = true;
            }
        }
        if (any_in) {
            MA_PIPE_HIP(hipEventRecord(p->ev_in[s], p->h2d));
            MA_PIPE_HIP(hipStreamWaitEvent(ctx->stream, p->ev_in[s], 0));
        }
        ma_status st = fn(user, row0, rows, ptrs);
        if (st != MA_OK) return finish(st);
        MA_PIPE_HIP(hipEventRecord(p->ev_done[s], ctx->stream));
        if (any_out) {
            {
                std::lock_guard<std::mutex> lk(drain.mu);
                drain.queue.push_back(k);
            }
            drain.cv.notify_all();
        }
    }
#undef MA_PIPE_HIP
    return finish(MA_OK);
}

}  // namespace ma

extern "C" ma_status ma_ctx_set_staging_tile(ma_ctx* ctx, size_t tile_bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(tile_bytes == 0 || tile_bytes >= ((size_t)1 << 16), MA_ERR_INVALID_ARGUMENT,
               "staging tile must be 0 (whole-operand staging) or at least 64 KiB");
    MA_ENTER_PRIMARY(ctx);
    ctx->staging_tile_bytes = tile_bytes;
    return MA_OK;
}
