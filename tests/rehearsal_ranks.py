"""One RANK PROCESS of the multi-process rehearsal (tests/test_gpu_guard.py, tests/test_gpu_sharded_table.py start `world` of
these on this box's one GPU): a plain host of libminarrow_hip.so with one context on device 0 and one ma_comm rank, the
communicator made by ncclCommInitRank of the loopback collective double (MINARROW_HIP_RCCL_PATH, inherited). The ranks share
nothing but the communicator ids on the command line — every rendezvous is the communicator's own. What is rehearsed: the combine
step of benches/benchmark_parallel_simd.rs:81-98 with one process per GPU.

    python tests/rehearsal_ranks.py <scenario> <rank> <world> <id hex> <id hex> <id hex>

Prints one JSON line; exit code 0 when the scenario ran to its end."""
import json
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from minarrow_amd import ffi  # noqa: E402
from minarrow_amd.host import Comm, Context  # noqa: E402

N = 1 << 20


class Records:
    """n_columns x slots reduction records (8 x u64 each) of this rank, every rank's gathered copy, 4 x u64 finals per column."""

    def __init__(self, ctx, world, n_columns=1, slots=1):
        self.ctx, self.world, self.n_columns, self.slots = ctx, world, n_columns, slots
        per_rank = 64 * n_columns * slots
        self.local, self.gathered, self.final = ctx.alloc(per_rank), ctx.alloc(per_rank * world), ctx.alloc(32 * n_columns)
        for b, nbytes in ((self.local, per_rank), (self.gathered, per_rank * world), (self.final, 32 * n_columns)):
            ctx.dev_memset(b, 0, nbytes)

    def ptr(self, word, column=0, slot=0):
        return self.local.ptr + 8 * ((slot * self.n_columns + column) * 8 + word)

    def finals(self):
        return [int(v) for v in self.final.download(np.uint64, 4 * self.n_columns)]


def scan_into(ctx, rec, col_i, col_f, n, column=0, slot=0):
    ctx.sum_into("i64", col_i, n, out_sum=rec.ptr(0, column, slot), out_count=rec.ptr(1, column, slot))
    ctx.sum_into("f64", col_f, n, out_sum=rec.ptr(2, column, slot), dd_lo=rec.ptr(3, column, slot), out_count=rec.ptr(4, column, slot))


def scenario_exchange(rank, world, ids):
    ctx = Context(0)
    lo = rank * N
    col_i, col_f = ctx.alloc(N * 8), ctx.alloc(N * 8)
    ctx.synth_iota("i64", col_i, N, lo)
    ctx.synth_iota("f64", col_f, N, lo)
    ctx.synchronize()
    comm = Comm(ctx, ids[0], rank, world)
    rep = comm.selftest(30_000)  # in-stream, overlapped on an event, overlapped on a stamp: both record sets each
    out = {"rank": rank, "selftest_ok": rep["ok"], "rccl_ranks": rep["rccl_ranks"], "forms": sorted(k for k, f in rep["forms"].items() if f["ok"]),
           "selftest_text": rep["text"], "rehearsal": ctx.lib.ma_rccl_path().decode().startswith("REHEARSAL")}
    total = N * world
    ctx.set_async(True)
    # in-stream
    rec = Records(ctx, world)
    scan_into(ctx, rec, col_i, col_f, N)
    comm.sum_exchange(rec.local, 1, 1, rec.gathered, rec.final)
    comm.synchronize_for(30_000)
    f = rec.finals()
    out["in_stream"] = f[:2]
    fsum = float(np.array([f[2]], dtype=np.uint64).view(np.float64)[0])
    exact = float(total * (total - 1) // 2)  # < 2^53: exactly representable
    out["f64_within_1ulp"] = abs(fsum - exact) <= math.ulp(exact) and f[3] == total
    out["finals_bits"] = f
    # overlapped, hand-off by event: two record sets alternate, the scans of step k + 1 run while step k's records are exchanged
    sets = [Records(ctx, world), Records(ctx, world)]
    for step in range(6):
        k = step & 1
        comm.slot_wait(k)
        scan_into(ctx, sets[k], col_i, col_f, N)
        comm.sum_exchange_overlapped(k, sets[k].local, 1, 1, sets[k].gathered, sets[k].final)
    comm.synchronize_for(30_000)
    out["overlapped_event"] = sets[1].finals()[:2]
    assert sets[0].finals() == sets[1].finals() == f
    # overlapped, hand-off by the fused scan's stamp
    stamps = [ctx.stamp_alloc(), ctx.stamp_alloc()]
    calls = [ctx.prepare_sum_fused([("l", col_i, N, sets[k].ptr(0)), ("g", col_f, N, sets[k].ptr(2))], stamp=stamps[k]) for k in (0, 1)]
    seq = [0, 0]
    for k in (0, 1):
        ctx.dev_memset(sets[k].final, 0, 32)
    for step in range(6):
        k = step & 1
        comm.slot_wait(k)
        seq[k] += 1
        calls[k](seq[k])
        comm.sum_exchange_overlapped_on_stamp(k, stamps[k], seq[k], sets[k].local, 1, 1, sets[k].gathered, sets[k].final)
    comm.synchronize_for(30_000)
    out["overlapped_stamp"] = sets[1].finals()[:2]
    assert sets[0].finals() == sets[1].finals() == f, (sets[0].finals(), sets[1].finals(), f)
    # the bare collectives
    send, recv = ctx.alloc(16), ctx.alloc(16)
    mine = np.array([int(np.arange(lo, lo + N, dtype=np.int64).sum()), N], dtype=np.int64)
    send.upload(mine)
    comm.all_reduce_sum_i64(send, recv, 2)
    comm.synchronize_for(30_000)
    out["all_reduce"] = [int(v) for v in recv.download(np.int64, 2)]
    st = comm.exchange_stats()
    out["exchange_samples"] = st["samples"]
    ctx.set_async(False)
    comm.close()
    ctx.close()
    return out


def scenario_sharded(rank, world, ids):
    """The rank-sharded SuperTable step: 3 batches ("slots") x 2 columns of records per rank in ONE exchange, folded per column
    over (rank, slot) in that order — minarrow_amd/parallel.py's ScalarExchange layout through ma_comm_sum_exchange."""
    ctx = Context(0)
    comm = Comm(ctx, ids[0], rank, world)
    slots, rows = 3, [70_001, 64, 1]
    rec = Records(ctx, world, n_columns=2, slots=slots)
    ctx.set_async(True)
    keep = []
    for slot in range(slots):
        n = rows[slot]
        rng = np.random.default_rng(1000 * rank + slot)  # every rank can rebuild every other rank's batches
        ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
        flts = rng.standard_normal(n) * 10.0 ** rng.integers(0, 14, size=n)
        bits = rng.integers(0, 256, size=n // 8 + 24, dtype=np.uint8)
        d_i, d_f, d_m = ctx.to_device(ints, 64), ctx.to_device(flts, 64), ctx.to_device(bits, 16)
        keep += [d_i, d_f, d_m]
        ctx.sum_into("i64", d_i, n, out_sum=rec.ptr(0, 0, slot), out_count=rec.ptr(1, 0, slot), mask=d_m, mask_bit_offset=5)
        ctx.sum_into("f64", d_f, n, out_sum=rec.ptr(2, 1, slot), dd_lo=rec.ptr(3, 1, slot), out_count=rec.ptr(4, 1, slot), mask=d_m,
                     mask_bit_offset=5)
    comm.sum_exchange(rec.local, slots, 2, rec.gathered, rec.final)
    comm.synchronize_for(30_000)
    want_i, want_c, all_f = 0, 0, []
    for r in range(world):
        for slot in range(slots):
            n = rows[slot]
            rng = np.random.default_rng(1000 * r + slot)
            ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
            flts = rng.standard_normal(n) * 10.0 ** rng.integers(0, 14, size=n)
            bits = rng.integers(0, 256, size=n // 8 + 24, dtype=np.uint8)
            valid = np.unpackbits(bits, bitorder="little")[5:5 + n].astype(bool)
            want_i += int(ints[valid].astype(object).sum())
            want_c += int(valid.sum())
            all_f += flts[valid].tolist()
    f = rec.finals()
    fsum = float(np.array([f[6]], dtype=np.uint64).view(np.float64)[0])
    exact = math.fsum(all_f)
    out = {"rank": rank, "i64_ok": f[0] == want_i & ((1 << 64) - 1) and f[1] == want_c, "f64_ok": abs(fsum - exact) <= math.ulp(exact) and f[7] == want_c,
           "finals_bits": f}
    ctx.set_async(False)
    comm.close()
    ctx.close()
    return out


def scenario_stall(rank, world, ids):
    assert world == 2
    ctx = Context(0)
    col_i, col_f = ctx.alloc(N * 8), ctx.alloc(N * 8)
    ctx.synth_iota("i64", col_i, N, rank * N)
    ctx.synth_iota("f64", col_f, N, rank * N)
    ctx.synchronize()
    comm = Comm(ctx, ids[0], rank, world)
    rec = Records(ctx, world)
    ctx.set_async(True)
    scan_into(ctx, rec, col_i, col_f, N)
    comm.sum_exchange_overlapped(0, rec.local, 1, 1, rec.gathered, rec.final)
    comm.synchronize_for(30_000)
    out = {"rank": rank}
    if rank == 1:
        comm.test_stall_next_exchange()  # this rank's exchange stream is held in front of its all-gather
    scan_into(ctx, rec, col_i, col_f, N)
    comm.sum_exchange_overlapped(1, rec.local, 1, 1, rec.gathered, rec.final)
    t0 = time.perf_counter()
    try:
        comm.synchronize_for(500)  # rank 0: its all-gather spins on the GPU for rank 1's; rank 1: its stream is held
        out["timed_out"] = False
    except ffi.MinarrowHipError as e:
        out["timed_out"] = e.status == ffi.MA_ERR_DEVICE and "did not finish within 500 ms" in str(e)
        out["error"] = str(e)
    out["waited_s"] = time.perf_counter() - t0
    out["broken"] = comm.is_broken
    comm.close()
    ctx.set_async(False)
    ctx.synchronize()
    out["ctx_ok"] = ctx.sum("i64", col_i, N)[1] == N
    # the ranks agree on a new communicator: a fresh id, the same contexts
    ctx.set_async(True)
    again = Comm(ctx, ids[1], rank, world)
    rec2 = Records(ctx, world)
    scan_into(ctx, rec2, col_i, col_f, N)
    again.sum_exchange(rec2.local, 1, 1, rec2.gathered, rec2.final)
    again.synchronize_for(30_000)
    total = 2 * N
    out["second_comm_ok"] = rec2.finals()[:2] == [total * (total - 1) // 2, total]
    if rank == 1:  # ... and leaves: no abort, no destroy of the last communicator's work in flight — the process is simply gone
        print(json.dumps(out), flush=True)
        import os
        os._exit(0)
    time.sleep(0.5)  # rank 1 is gone by now, or going
    scan_into(ctx, rec2, col_i, col_f, N)
    again.sum_exchange(rec2.local, 1, 1, rec2.gathered, rec2.final)
    t0 = time.perf_counter()
    try:
        again.synchronize_for(500)
        out["vanished_peer_timed_out"] = False
    except ffi.MinarrowHipError as e:
        out["vanished_peer_timed_out"] = e.status == ffi.MA_ERR_DEVICE
    out["vanished_waited_s"] = time.perf_counter() - t0
    again.close()
    ctx.set_async(False)
    ctx.synchronize()
    out["ctx_ok_after_vanish"] = ctx.sum("i64", col_i, N)[1] == N
    ctx.close()
    return out


def main():
    scenario, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    ids = [bytes.fromhex(h) for h in sys.argv[4:]]
    out = {"exchange": scenario_exchange, "sharded": scenario_sharded, "stall": scenario_stall}[scenario](rank, world, ids)
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
