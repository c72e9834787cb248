"""Single-process multi-GPU group API (ma_group_*) and the multi-process communicator (ma_comm_*).

Host exchange: members may share a device (independent contexts and streams), so the partition + fold logic is
covered with 1, 3 and 8 members on the pool's one-GPU boxes. RCCL exchange: needs distinct devices — the test takes
min(device_count, 8) of them (one on this pool: ncclCommInitAll, the grouped all-gather and the device fold all run, with
one rank; on an 8-GPU node the same test covers 8)."""
import ctypes as C
import math
import os
import time

import numpy as np
import pytest

from minarrow_amd import ffi
from minarrow_amd.parallel import row_chunks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members", [1, 3, 8])
def test_group_sums_match_single_array(ctx, oracle, members):
    lib = ffi.load_library()
    n = 2_000_003
    rng = np.random.default_rng(members)
    ints = rng.integers(-(1 << 50), 1 << 50, size=n, dtype=np.int64)
    flts = rng.standard_normal(n) * 1e9
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    di, df, dm = ctx.to_device(ints, 64), ctx.to_device(flts, 64), ctx.to_device(bits, 16)
    g = C.c_void_p()
    devs = (C.c_int32 * members)(*([0] * members))
    ffi.check(lib.ma_group_create(C.cast(devs, C.c_void_p), members, C.byref(g)))
    try:
        assert lib.ma_group_size(g) == members and lib.ma_group_ctx(g, 0)
        chunks = row_chunks(n, members)
        lens = (C.c_size_t * members)(*[b - a for a, b in chunks])
        pi = (C.c_void_p * members)(*[di.ptr + a * 8 for a, _ in chunks])
        pf = (C.c_void_p * members)(*[df.ptr + a * 8 for a, _ in chunks])
        pm = (C.c_void_p * members)(*([dm.ptr] * members))        # the un-windowed validity buffer ...
        offs = (C.c_size_t * members)(*[a for a, _ in chunks])    # ... plus each chunk's first row as bit offset
        s, c = C.c_int64(), C.c_uint64()
        cast = lambda arr: C.cast(arr, C.c_void_p)  # noqa: E731
        ffi.check(lib.ma_group_sum_i64(g, cast(pi), cast(lens), None, None, C.byref(s), C.byref(c)))
        assert (s.value, c.value) == (int(ints.sum()), n)
        ffi.check(lib.ma_group_sum_i64(g, cast(pi), cast(lens), cast(pm), cast(offs), C.byref(s), C.byref(c)))
        assert (s.value, c.value) == oracle.masked_sum(ints, bits, 0)
        f = C.c_double()
        ffi.check(lib.ma_group_sum_f64(g, cast(pf), cast(lens), None, None, C.byref(f), C.byref(c)))
        exact = math.fsum(flts.tolist())
        assert c.value == n and abs(f.value - exact) <= math.ulp(exact)
        valid = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
        ffi.check(lib.ma_group_sum_f64(g, cast(pf), cast(lens), cast(pm), cast(offs), C.byref(f), C.byref(c)))
        exact = math.fsum(flts[valid].tolist())
        assert c.value == int(valid.sum()) and abs(f.value - exact) <= math.ulp(exact)
    finally:
        lib.ma_group_destroy(g)


def test_device_fold_of_records_equals_the_host_fold(ctx):
    """ma_fold_sum_records (the step after the all-gather) against minarrow_amd.parallel's host fold, bit for bit,
    incl. cancellation between ranks, and against the exact sum."""
    import math

    from minarrow_amd.parallel import fold_dd, fold_int

    rng = np.random.default_rng(4)
    for world in (1, 2, 8, 64):
        rec = np.zeros((world, 8), dtype=np.uint64)
        ints = rng.integers(-2**62, 2**62, size=world, dtype=np.int64)
        his = (rng.standard_normal(world) * 1e17)
        his[::2] *= -1
        los = his * 2.0**-55 * rng.standard_normal(world)
        rec[:, 0] = ints.view(np.uint64)
        rec[:, 1] = rng.integers(0, 2**40, size=world, dtype=np.uint64)
        rec[:, 2] = his.view(np.uint64)
        rec[:, 3] = los.view(np.uint64)
        rec[:, 4] = rng.integers(0, 2**40, size=world, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        ctx.fold_sum_records(rec, world, 8, out)
        assert int(out[0]) == fold_int([int(v) for v in rec[:, 0]])
        assert int(out[1]) == int(rec[:, 1].astype(object).sum()) and int(out[3]) == int(rec[:, 4].astype(object).sum())
        want = fold_dd(list(zip(his.tolist(), los.tolist())))
        got = float(out[2:3].view(np.float64)[0])
        assert got == want
        exact = math.fsum(his.tolist() + los.tolist())
        assert abs(got - exact) <= 2 * math.ulp(exact) + abs(exact) * 2**-100
        # device-resident records and output (the async bench path)
        drec, dout = ctx.to_device(rec), ctx.alloc(32)
        ctx.fold_sum_records(drec, world, 8, dout)
        np.testing.assert_array_equal(dout.download(np.uint64, 4), out)


def _uploaders(ctxs):
    """The context each member's chunk is uploaded through: its own — or, when the members share ONE device (a rehearsal), member 0's
    for all: an upload grows a context two copy streams, and a process should not keep more hardware queues alive on a device
    than it runs at a time (23: DESIGN.md section 6)."""
    return [ctxs[0]] * len(ctxs) if len({c.hip_device for c in ctxs}) == 1 else list(ctxs)


def _chunk_tables(ctxs, ints, flts, bits, chunks):
    """Uploads chunk i to member i's device; returns the per-member buffers."""
    up = _uploaders(ctxs)
    di = [c.to_device(ints[a:b], 64) for c, (a, b) in zip(up, chunks)]
    df = [c.to_device(flts[a:b], 64) for c, (a, b) in zip(up, chunks)]
    dm = [c.to_device(bits, 16) for c in up]  # every device holds the un-windowed validity buffer
    return di, df, dm


def _device_lists(members):
    """The member -> device maps the group tests run over: every member on device 0 (partition logic on a one-GPU box)
    and one member per visible GPU, up to 8 (the cross-device paths — on this pool's one-GPU boxes that is [0])."""
    distinct = list(range(min(ffi.device_count(), 8)))
    return [[0] * members] + ([distinct] if distinct != [0] * members else [])


@pytest.mark.parametrize("issue", ["threads", "caller"])
@pytest.mark.parametrize("exchange", ["rccl", "rccl-overlap", "host"])
def test_group_over_distinct_devices(oracle, exchange, issue):
    """One member per visible GPU (up to 8). With exchange = "rccl": ncclCommInitAll + all-gather (one per member issue
    thread, or grouped on the calling thread) + device fold — every member must hold the same finals."""
    n_dev = min(ffi.device_count(), 8)
    assert n_dev >= 1
    _partitioned_sums_on_every_member(oracle, list(range(n_dev)), exchange, issue)


@pytest.mark.rehearsal
@pytest.mark.parametrize("members, issue", [(2, "threads"), (4, "threads"), (8, "threads"), (3, "caller"), (8, "caller")])
@pytest.mark.parametrize("exchange", ["rccl", "rccl-overlap", "rccl-overlap-lanes"])
def test_rehearsal_group_of_members_sharing_the_device(oracle, members, exchange, issue):
    """The same job with 2 / 4 / 8 members on device 0 through the loopback collective double: every member's all-gather waits
    on the GPU for its peers' (issued by eight threads at once, or grouped on the calling thread), in-stream, on side
    streams, and with two scan lanes — bit-identical finals on every member."""
    _partitioned_sums_on_every_member(oracle, [0] * members, exchange, issue)


def _partitioned_sums_on_every_member(oracle, devices, exchange, issue):
    from minarrow_amd.host import Group

    n_dev = len(devices)
    n = 3_000_017
    rng = np.random.default_rng(11)
    ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    flts = rng.standard_normal(n) * 1e12
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    n_distinct = len(set(devices))
    with Group(devices, exchange=exchange, issue=issue) as g:
        assert g.exchange_kind == exchange.split("-")[0] and g.issue_kind == issue
        assert ("overlapped on side streams" in g.exchange_note) == exchange.startswith("rccl-overlap")
        assert g.scan_lanes == exchange.endswith("lanes")
        note = g.exchange_note
        assert "instead of RCCL" not in note and "peer access:" in note and ("one thread per member" in note) == (issue == "threads")
        # several members on one device: only the loopback double takes that, and the note says what it is
        assert note.startswith("REHEARSAL") == (exchange != "host" and n_distinct < n_dev)
        # peer capability is probed between every pair of members at creation; a member always reaches itself
        assert all(g.peer_access(i, i) for i in range(n_dev)) and not g.peer_access(0, n_dev) and not g.peer_access(-1, 0)
        if n_distinct > 1:
            assert f"/{n_distinct * (n_distinct - 1)} ordered device pairs" in note
        ctxs = [g.member_ctx(i) for i in range(n_dev)]
        assert len({c.hip_device for c in ctxs}) == n_distinct
        chunks = row_chunks(n, n_dev)
        di, df, dm = _chunk_tables(ctxs, ints, flts, bits, chunks)
        lens = [b - a for a, b in chunks]
        offs = [a for a, _ in chunks]
        # several steps enqueued back to back, two columns sharing one exchange, one synchronize (with "rccl-overlap" the
        # exchange of step k runs on side streams while step k + 1 scans into the second record set)
        for _ in range(3):
            g.enqueue_sum("i64", 0, di, lens)
            g.enqueue_sum("f64", 0, df, lens)
            g.enqueue_sum("i64", 5, di, lens, dm, offs)
            g.enqueue_sum("f64", 5, df, lens, dm, offs)
            g.exchange()
        g.synchronize()
        if exchange != "host":
            assert g.exchange_stats()["rccl_ranks"] == n_dev
        valid = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
        for m in range(n_dev):
            isum, icnt, fsum, fcnt = g.result(0, m)
            assert (isum, icnt) == (oracle.sum_scalar(ints), n)
            exact = math.fsum(flts.tolist())
            assert fcnt == n and abs(fsum - exact) <= math.ulp(exact)
            isum, icnt, fsum, fcnt = g.result(5, m)
            assert (isum, icnt) == oracle.masked_sum(ints, bits, 0)
            exact = math.fsum(flts[valid].tolist())
            assert fcnt == int(valid.sum()) and abs(fsum - exact) <= math.ulp(exact)
            assert g.result(0, m) == g.result(0, 0) and g.result(5, m) == g.result(5, 0)
        # the synchronous one-call forms
        assert g.sum("i64", di, lens) == (oracle.sum_scalar(ints), n)
        f, c = g.sum("f64", df, lens, dm, offs)
        assert c == int(valid.sum()) and abs(f - exact) <= math.ulp(exact)
        for b in di + df + dm:
            b.free()


def test_group_rccl_needs_distinct_devices_and_can_fall_back():
    from minarrow_amd.host import Group

    with pytest.raises(ffi.MinarrowHipError) as e:
        Group([0, 0], exchange="rccl")
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED and "distinct devices" in e.value.message
    with Group([0, 0], exchange="rccl-or-host") as g:
        assert g.exchange_kind == "host" and "distinct devices" in g.exchange_note
        ctx0 = g.member_ctx(0)
        a = np.arange(1000, dtype=np.int64)
        d = ctx0.to_device(a, 64)
        assert g.sum("i64", [d, d.offset(4000)], [500, 500]) == (int(a.sum()), 1000)
        d.free()


def test_comm_one_rank(ctx, oracle):
    """ma_comm_*: ncclGetUniqueId + ncclCommInitRank with one rank on this box's GPU, the bare collectives and the
    record exchange (all-gather + ordered fold)."""
    from minarrow_amd.host import Comm

    lib = ffi.load_library()
    assert lib.ma_rccl_version() >= 20000
    comm = Comm(ctx, Comm.unique_id(), 0, 1)
    try:
        assert (lib.ma_comm_rank(comm.handle), lib.ma_comm_size(comm.handle)) == (0, 1)
        src = np.arange(256, dtype=np.int64) - 77
        d_src, d_dst = ctx.to_device(src), ctx.alloc(src.nbytes)
        comm.all_gather(d_src, d_dst, src.nbytes)
        np.testing.assert_array_equal(d_dst.download(np.int64, src.size), src)
        ctx.dev_memset(d_dst, 0, src.nbytes)
        comm.all_reduce_sum_i64(d_src, d_dst, src.size)
        np.testing.assert_array_equal(d_dst.download(np.int64, src.size), src)
        # two slots x three columns of records: the kernels write them, the exchange folds column c over the slots
        n = 100_003
        rng = np.random.default_rng(3)
        ints = rng.integers(-(1 << 60), 1 << 60, size=n, dtype=np.int64)
        flts = rng.standard_normal(n) * 1e6
        di, df = ctx.to_device(ints, 64), ctx.to_device(flts, 64)
        slots, cols = 2, 3
        local, gathered, finals = ctx.alloc(slots * cols * 64), ctx.alloc(slots * cols * 64), ctx.alloc(cols * 32)
        ctx.dev_memset(local, 0, slots * cols * 64)
        half = (n // 2 // 64) * 64
        for slot, (a, b) in enumerate(((0, half), (half, n))):
            r = local.ptr + 64 * (slot * cols + 1)  # column 1
            ctx.sum_into("i64", di.offset(a * 8), b - a, out_sum=r, out_count=r + 8)
            ctx.sum_into("f64", df.offset(a * 8), b - a, out_sum=r + 16, dd_lo=r + 24, out_count=r + 32)
        comm.sum_exchange(local, slots, cols, gathered, finals)
        f = finals.download(np.uint64, cols * 4).reshape(cols, 4)
        assert int(f[1][0]) == oracle.sum_scalar(ints) & ((1 << 64) - 1) and int(f[1][1]) == n and int(f[1][3]) == n
        exact = math.fsum(flts.tolist())
        assert abs(float(f[1][2:3].view(np.float64)[0]) - exact) <= math.ulp(exact)
        assert not f[0].any() and not f[2].any()  # untouched columns fold to zero
        np.testing.assert_array_equal(gathered.download(np.uint64, slots * cols * 8), local.download(np.uint64, slots * cols * 8))
        # the overlapped form: two record sets alternate, each exchange runs on the communicator's own stream behind the
        # scans that filled its set while the context's stream already works on the other set; slot_wait orders the reuse
        ctx.set_async(True)
        sets = [(ctx.alloc(64), ctx.alloc(64), ctx.alloc(32)) for _ in range(2)]
        wants = []
        for step in range(6):
            k = step % 2
            loc, gat, fin = sets[k]
            a, b = (step * 1000) % 50_000, n - (step * 777) % 40_000
            comm.slot_wait(k)
            ctx.sum_into("i64", di.offset(a * 8), b - a, out_sum=loc.ptr, out_count=loc.ptr + 8)
            ctx.sum_into("f64", df.offset(a * 8), b - a, out_sum=loc.ptr + 16, dd_lo=loc.ptr + 24, out_count=loc.ptr + 32)
            comm.sum_exchange_overlapped(k, loc, 1, 1, gat, fin)
            wants.append((k, oracle.sum_scalar(ints[a:b]) & ((1 << 64) - 1), b - a, math.fsum(flts[a:b].tolist())))
            if step >= 4:  # the last exchange of each set: read back after everything has drained
                pass
        comm.synchronize()
        ctx.set_async(False)
        for k, want_i, want_n, want_f in wants[-2:]:
            f = sets[k][2].download(np.uint64, 4)
            assert int(f[0]) == want_i and int(f[1]) == want_n and int(f[3]) == want_n
            assert abs(float(f[2:3].view(np.float64)[0]) - want_f) <= math.ulp(want_f)
        with pytest.raises(ffi.MinarrowHipError):
            comm.sum_exchange_overlapped(2, sets[0][0], 1, 1, sets[0][1], sets[0][2])
    finally:
        comm.close()


@pytest.mark.big
def test_group_one_billion_rows_with_nulls_partitioned_eight_ways(ctx):
    """BASELINE config 4 at full size through the group API: a 10^9-row i64 column with 10 % nulls, row-chunk
    partitioned over 8 members (sharing this box's GPU; host exchange) — the job's finals must equal the single-call
    scan of the whole column, and the f64 twin must stay within 1 ULP of it (size-independent property: a checksum of
    checksums over the partition)."""
    from minarrow_amd.host import Group

    n, members = 1_000_000_000, 8
    data, fdata = ctx.alloc(n * 8), ctx.alloc(n * 8)
    mask = ctx.alloc(n // 8 + 128)
    ctx.synth_iota("i64", data, n, 0)
    ctx.synth_iota("f64", fdata, n, 0)
    ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
    whole_i = ctx.sum("i64", data, n, mask=mask)
    whole_f = ctx.sum("f64", fdata, n, mask=mask)
    chunks = row_chunks(n, members)
    lens = [b - a for a, b in chunks]
    assert all(a % 64 == 0 for a, _ in chunks)
    with Group([0] * members, exchange="host") as g:
        pi = [data.ptr + a * 8 for a, _ in chunks]
        pf = [fdata.ptr + a * 8 for a, _ in chunks]
        pm = [mask.ptr] * members
        offs = [a for a, _ in chunks]
        for _ in range(2):  # two steps back to back, one synchronize
            g.enqueue_sum("i64", 0, pi, lens, pm, offs)
            g.enqueue_sum("f64", 0, pf, lens, pm, offs)
            g.exchange()
        g.synchronize()
        isum, icnt, fsum, fcnt = g.result(0)
        assert (isum, icnt) == whole_i
        assert fcnt == whole_f[1] and abs(fsum - whole_f[0]) <= math.ulp(whole_f[0])
        assert 0.0999 < 1 - icnt / n < 0.1001
    for b in (data, fdata, mask):
        b.free()


@pytest.mark.parametrize("members,exchange,distinct", [(1, "rccl-or-host", False), (3, "host", False), (8, "host", False),
                                                      (8, "rccl-or-host", True)])
def test_group_super_array_broadcast_fans_chunks_out(oracle, members, exchange, distinct):
    """ma_group_route_super_array_broadcast: chunk pair i runs on member i % G (here every member drives device 0 — the
    partition, the per-member tables and the scatter of the validity flags are what is under test; on a multi-GPU node
    the same code places member m on device m). Mixed mask presence, odd lengths, an empty chunk, more chunks than
    members; compared chunk by chunk with the oracle. A dense integer division by zero on one member is reported by
    synchronize(); a chunk resident on the wrong device is refused."""
    from minarrow_amd.host import Group

    rng = np.random.default_rng(100 + members)
    lens = [5, 64, 1000, 4097, 70_001, 0, 129, 8192, 33, 100_003, 1]
    k = len(lens)
    devices = [0] * members
    if distinct:  # one member per visible GPU (up to 8): every chunk's buffers live on ITS member's device
        devices = list(range(min(ffi.device_count(), 8)))
        members = len(devices)
    with Group(devices, exchange) as g:
        ctxs = [g.member_ctx(m) for m in range(members)]
        own = lambda i: ctxs[i % members]  # noqa: E731
        for fmt, dt in (("l", np.int64), ("g", np.float64), ("i", np.int32)):
            lhs = [rng.integers(1, 100, size=n).astype(dt) for n in lens]
            rhs = [rng.integers(1, 100, size=n).astype(dt) for n in lens]
            lm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 2 == 0 else None for i, n in enumerate(lens)]
            rm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 3 != 1 else None for i, n in enumerate(lens)]
            up = lambda xs: [own(i).to_device(x, 64) if x is not None else None for i, x in enumerate(xs)]  # noqa: E731
            dl, dr, dlm, drm = up(lhs), up(rhs), up(lm), up(rm)
            outs = [own(i).alloc(max(n, 1) * np.dtype(dt).itemsize + 64) for i, n in enumerate(lens)]
            oms = [own(i).alloc(((n + 63) // 64) * 8 + 8) for i, n in enumerate(lens)]
            for op, name in ((0, "add"), (2, "multiply"), (3, "divide")):
                has = g.route_super_array_broadcast(fmt, op, dl, dr, lens, lens, outs, dlm, drm, oms)
                g.synchronize()
                for i, n in enumerate(lens):
                    if lm[i] is not None and rm[i] is not None:
                        common = oracle.bitmask_union(oracle.pad_bits(lm[i], n), oracle.pad_bits(rm[i], n), n)
                    else:
                        common = lm[i] if lm[i] is not None else rm[i]
                    assert has[i] == (common is not None), (fmt, name, i)
                    if n == 0:
                        continue
                    if common is None:
                        fn = oracle.apply_float if np.dtype(dt).kind == "f" else oracle.apply_int
                        st, want, _, _ = fn(oracle.aligned_copy(lhs[i]), oracle.aligned_copy(rhs[i]), name)
                        np.testing.assert_array_equal(outs[i].download(dt, n), want, err_msg=f"{fmt} {name} chunk {i}")
                    else:
                        body = oracle.float_body if np.dtype(dt).kind == "f" else oracle.int_body
                        st, want, want_mask = body("masked_std", lhs[i], rhs[i], name, mask=oracle.pad_bits(common, n))
                        nb = ((n + 63) // 64) * 8
                        np.testing.assert_array_equal(outs[i].download(dt, n), want, err_msg=f"{fmt} {name} chunk {i}")
                        np.testing.assert_array_equal(oms[i].download(np.uint8, nb), want_mask[:nb], err_msg=f"{fmt} {name} chunk {i} validity")
        # length mismatch in ANY chunk is reported before a member starts
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.route_super_array_broadcast("i", 0, dl, dr, lens, lens[:-1] + [2], outs)
        assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "Chunk 10" in e.value.message
        # dense integer division by zero on one member: latched there, reported by the group's synchronize
        z = rhs[4].copy()
        z[17] = 0
        dz = own(4).to_device(z, 64)
        g.route_super_array_broadcast("i", 3, dl[:5], dr[:4] + [dz], lens[:5], lens[:5], outs[:5])
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.synchronize()
        assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
        g.synchronize()  # the latch is cleared by the report


@pytest.mark.parametrize("members,dest,distinct", [(1, 0, False), (3, 1, False), (8, 7, False), (8, 7, True)])
def test_group_consolidate_column_gathers_onto_one_member(ctx, oracle, members, dest, distinct):
    """ma_group_consolidate_column: chunk i on member i % G, the consolidated column (+ validity joined at bit
    granularity, chunks without a bitmap = all valid) on member `dest`. With every member on device 0 the peer copies
    are same-device copies; what is under test is the placement arithmetic, the staging of the validity windows (byte
    offsets, bit remainders), the stream ordering between the members and the join — against the single-context
    ma_consolidate_column on the same chunks, which is itself checked against the oracle and the reference's vectors."""
    from minarrow_amd.host import Group

    rng = np.random.default_rng(500 + members)
    devices = [0] * members
    if distinct:  # one member per visible GPU (up to 8): real peer copies over xGMI into the last member's HBM
        devices = list(range(min(ffi.device_count(), 8)))
        members = len(devices)
        dest = members - 1
    for dt, lens in ((np.int64, [5, 64, 1000, 0, 4097, 70_001, 129, 1, 8192, 33]), (np.int32, [100_003, 63, 65, 7]), (np.uint8, [3, 1000, 17, 4096, 9])):
        dt = np.dtype(dt)
        k, total = len(lens), sum(lens)
        data = [rng.integers(0, 100, size=n).astype(dt) for n in lens]
        offs = [int(o) for o in rng.integers(0, 50, size=k)]
        masks = [rng.integers(0, 256, size=(o + n) // 8 + 16, dtype=np.uint8) if i % 3 != 1 else None
                 for i, (n, o) in enumerate(zip(lens, offs))]
        nb = ((total + 63) // 64) * 8
        # single-context answer
        want_out, want_mask = ctx.alloc(total * dt.itemsize + 64), ctx.alloc(nb + 8)
        d1 = [ctx.to_device(x, 64) for x in data]
        m1 = [ctx.to_device(m, 16) if m is not None else None for m in masks]
        for with_masks in (False, True):
            has1 = ctx.consolidate_column(dt.itemsize, d1, lens, want_out, m1 if with_masks else None, offs if with_masks else None,
                                          want_mask if with_masks else None)
            with Group(devices, "host") as g:
                assert all(g.peer_access(m, dest) for m in range(members)), g.exchange_note
                own = [g.member_ctx(m) for m in range(members)]
                dd = [own[i % members].to_device(x, 64) for i, x in enumerate(data)]
                dm = [own[i % members].to_device(m, 16) if m is not None else None for i, m in enumerate(masks)]
                out = own[dest].alloc(total * dt.itemsize + 64)
                om = own[dest].alloc(nb + 8)
                for _ in range(2):  # twice: the staging arena is re-used in stream order
                    has = g.consolidate_column(dest, dt.itemsize, dd, lens, out, dm if with_masks else None,
                                               offs if with_masks else None, om if with_masks else None)
                g.synchronize()
                assert has == has1 == with_masks
                np.testing.assert_array_equal(out.download(dt, total), want_out.download(dt, total))
                if with_masks:
                    np.testing.assert_array_equal(om.download(np.uint8, nb), want_mask.download(np.uint8, nb))
        for b in d1 + [m for m in m1 if m is not None] + [want_out, want_mask]:
            b.free()
    with Group([0], "host") as g:
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.consolidate_column(0, 8, [], [], None)
        assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "empty SuperTable" in e.value.message
        host = np.zeros(16, dtype=np.int64)
        o = g.member_ctx(0).alloc(256)
        with pytest.raises(ffi.MinarrowHipError) as e:  # chunks must be device memory of their member
            g.consolidate_column(0, 8, [host], [16], o)
        assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


def test_group_refuses_chunks_resident_on_another_members_device(ctx):
    """The cross-device guards (one row chunk per GPU): member i's kernels dereference chunk i, so a chunk
    in another GPU's HBM must come back as MA_ERR_INVALID_ARGUMENT before anything is enqueued — on a multi-GPU node it
    would otherwise be a memory fault — and ma_group_consolidate_column must refuse an owner without peer access to the
    destination. With two or more GPUs the real thing is tested; on a one-GPU box ma_group_test_set_member_device makes
    member 1 LOOK as if it lived on HIP device 7 (nothing launches differently), which drives the same code."""
    from minarrow_amd.host import Group

    lib = ffi.load_library()
    a = np.arange(4096, dtype=np.int64)
    f = a.astype(np.float64)
    m = np.full(4096 // 8 + 16, 0xFF, dtype=np.uint8)
    assert lib.ma_pointer_device(None) == -1 and lib.ma_pointer_device(a.ctypes.data) == -1
    real = ffi.device_count() >= 2
    for issue in ("threads", "caller"):
        with Group([0, 1] if real else [0, 0], "host", issue=issue) as g:
            c0, c1 = g.member_ctx(0), g.member_ctx(1)
            if not real:
                g.test_set_member_device(1, 7, peer_capable=False)
                assert g.peer_access(0, 0) and not g.peer_access(0, 1) and not g.peer_access(1, 0)
            on0, on0f, on0m = c0.to_device(a, 64), c0.to_device(f, 64), c0.to_device(m, 16)
            assert lib.ma_pointer_device(on0.ptr) == c0.hip_device == lib.ma_pointer_device(on0.ptr + 8 * 100)
            good1 = c1.to_device(a, 64) if real else None
            # sums: chunk 1 handed to member 1 but resident on member 0's device
            for tag, buf in (("i64", on0), ("f64", on0f)):
                with pytest.raises(ffi.MinarrowHipError) as e:
                    g.enqueue_sum(tag, 0, [buf, buf], [4096, 4096])
                assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "chunk 1 belongs to member 1" in e.value.message
                assert f"resident on device {c0.hip_device}" in e.value.message
            # ... or only its validity bitmap is
            if real:
                with pytest.raises(ffi.MinarrowHipError) as e:
                    g.enqueue_sum("i64", 0, [on0, good1], [4096, 4096], [on0m, on0m], [0, 0])
                assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "validity bitmap" in e.value.message
            # host memory is reachable from every device: accepted (staged), and so is an empty chunk with any pointer
            assert g.sum("i64", [on0, a], [4096, 4096]) == (2 * int(a.sum()), 8192)
            assert g.sum("i64", [on0, on0], [4096, 0]) == (int(a.sum()), 4096)
            # chunk fan-out: same rule for operands, result and bitmaps
            out = c0.alloc(4096 * 8 + 64)
            with pytest.raises(ffi.MinarrowHipError) as e:
                g.route_super_array_broadcast("l", 0, [on0, on0], [on0, on0], [4096, 4096], [4096, 4096], [out, out])
            assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "chunk 1" in e.value.message
            # consolidate: chunk 1 must be resident on member 1 ...
            whole = c0.alloc(2 * 4096 * 8 + 64)
            with pytest.raises(ffi.MinarrowHipError) as e:
                g.consolidate_column(0, 8, [on0, on0], [4096, 4096], whole)
            assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "chunk 1 belongs to member 1" in e.value.message
            if not real:
                # ... and when it is (member 1 is looked at as device 0 again), an owner WITHOUT a link to the destination is
                # refused with MA_ERR_UNSUPPORTED instead of being left to whatever the runtime does without one
                g.test_set_member_device(1, c0.hip_device, peer_capable=False)
                with pytest.raises(ffi.MinarrowHipError) as e:
                    g.consolidate_column(0, 8, [on0, on0], [4096, 4096], whole)
                assert e.value.status == ffi.MA_ERR_UNSUPPORTED and "no peer access" in e.value.message
                g.test_set_member_device(1, c0.hip_device, peer_capable=True)
                g.consolidate_column(0, 8, [on0, on0], [4096, 4096], whole)
                g.synchronize()
                np.testing.assert_array_equal(whole.download(np.int64, 8192), np.concatenate([a, a]))
            else:
                g.consolidate_column(0, 8, [on0, good1], [4096, 4096], whole)
                g.synchronize()
                np.testing.assert_array_equal(whole.download(np.int64, 8192), np.concatenate([a, a]))


def test_group_issue_thread_handshake_spinning_and_sleeping():
    """The hand-off between a calling thread and the members' issue threads (ma_group.hip: one job posted to all, workers
    spin 200 us then sleep on a condition variable, the caller spins 100 us then sleeps): thousands of tiny group calls
    from TWO calling threads with pauses on either side of both thresholds, so that jobs find workers spinning, about to
    sleep and asleep, and callers wait both ways. Every call's result is checked; a lost wake-up is a hang (the join
    timeout), a mixed-up job a wrong sum."""
    import threading
    import time

    from minarrow_amd.host import Group

    rng = np.random.default_rng(5)
    data = rng.integers(-(1 << 40), 1 << 40, size=64 * 1024, dtype=np.int64)
    errors, done = [], []
    with Group([0] * 8, "host") as g:
        ctxs = [g.member_ctx(i) for i in range(8)]
        bufs = [c.to_device(data, 64) for c in ctxs]
        pre = np.concatenate([[0], np.cumsum(data)])

        def caller(seed):
            r = np.random.default_rng(seed)
            t_end = time.time() + float(os.environ.get("MA_STRESS_SECONDS", "4"))
            k = 0
            while time.time() < t_end and not errors:
                lens = [int(x) for x in r.integers(0, 4096, size=8)]
                offs = [int(x) & ~1 for x in r.integers(0, data.size - 4096, size=8)]
                got = g.sum("i64", [b.offset(o * 8) for b, o in zip(bufs, offs)], lens)
                want = sum(int(pre[o + n] - pre[o]) for o, n in zip(offs, lens))
                if got != (want, sum(lens)):
                    errors.append((seed, k, got, want))
                k += 1
                pause = r.choice([0.0, 0.0, 0.00005, 0.00015, 0.0003, 0.002])
                if pause:
                    time.sleep(float(pause))
            done.append(k)

        threads = [threading.Thread(target=caller, args=(s,)) for s in (1, 2)]
        [t.start() for t in threads]
        [t.join(timeout=120 + float(os.environ.get("MA_STRESS_SECONDS", "4"))) for t in threads]
        assert not any(t.is_alive() for t in threads), "a group call never returned (lost wake-up?)"
    assert not errors, errors[:3]
    assert sum(done) > 500, done
    print(f"{sum(done)} group calls from two threads checked, 0 errors")


@pytest.mark.parametrize("members,exchange,issue", [(1, "rccl", "threads"), (1, "rccl-overlap", "caller"), (4, "host", "caller"),
                                                    (8, "host", "threads"),
                                                    pytest.param(4, "rccl", "caller", marks=pytest.mark.rehearsal),
                                                    pytest.param(8, "rccl-overlap", "threads", marks=pytest.mark.rehearsal)])
def test_group_sum_of_a_chunked_column(ctx, oracle, members, exchange, issue):
    """ma_group_enqueue_sum_chunks: ONE column held as 3000 chunks (ragged, some empty, validity at odd bit offsets on two
    thirds), chunk i on member i % size; every member sums its chunks in one pass, one exchange: the total equals the sum
    over the consolidated column — bit-exact for i64 / i32, within 1 ULP for f64 / f32 — and every member holds the same
    finals. On a one-GPU box the members share device 0 (host exchange) or the group has one member (RCCL)."""
    from minarrow_amd.host import Group

    rng = np.random.default_rng(members * 5 + len(exchange))
    lens = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 8192, 8191], size=3000)]
    with Group([0] * members, exchange=exchange, issue=issue) as g:
        for fmt, dt, col in (("l", np.int64, 0), ("g", np.float64, 0), ("i", np.int32, 3), ("f", np.float32, 3)):
            if np.dtype(dt).kind == "f":
                cols = [(rng.standard_normal(n) * 1e6).astype(dt) for n in lens]
            else:
                info = np.iinfo(dt)
                cols = [rng.integers(info.min // 2, info.max // 2, size=n, dtype=dt) for n in lens]
            masks, offs, d_cols, d_masks = [], [], [], []
            for i, n in enumerate(lens):
                d_cols.append(ctx.to_device(cols[i], 64))
                if i % 3 == 0 or n == 0:
                    masks.append(None); offs.append(0); d_masks.append(None)
                else:
                    off = [0, 3, 64, 77][i % 4]
                    m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
                    masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
            for masked in (False, True):
                g.enqueue_sum_chunks(fmt, col, d_cols, lens, d_masks if masked else None, offs if masked else None)
                g.exchange()
                g.synchronize()
                sel = [c[np.unpackbits(m, bitorder="little")[o:o + n].astype(bool)] if (masked and m is not None) else c
                       for c, m, o, n in zip(cols, masks, offs, lens)]
                allv = np.concatenate(sel)
                is_float = np.dtype(dt).kind == "f"  # the exact value once, not once per member
                exact = math.fsum(allv.astype(np.float64).tolist()) if is_float else int(allv.astype(np.int64).sum(dtype=object))
                for m_ in range(members):
                    isum, icnt, fsum, fcnt = g.result(col, m_)
                    if is_float:
                        assert fcnt == allv.size and abs(fsum - exact) <= math.ulp(exact)
                    else:
                        assert icnt == allv.size and (isum - exact) % (1 << 64) == 0
            for b in d_cols + [m for m in d_masks if m is not None]:
                b.free()
        # an empty chunk list is a zero record on every member
        g.enqueue_sum_chunks("l", 7, [], [])
        g.exchange()
        g.synchronize()
        assert g.result(7, 0)[:2] == (0, 0)


@pytest.mark.parametrize("members, exchange", [(1, "host"), (3, "host"), (8, "host"), (1, "rccl"), (1, "rccl-overlap")] +
                         [pytest.param(m, x, marks=pytest.mark.rehearsal)
                          for m in (2, 4, 8) for x in ("rccl", "rccl-overlap", "rccl-overlap-lanes")])
def test_group_fused_table_step_equals_the_per_column_steps(ctx, oracle, members, exchange):
    """ma_group_enqueue_sum_table: the partitioned step as ONE launch per member (i64 + f64 chunk of each member, dense and
    Bitmask-gated) must give the finals of the two-launch form bit for bit, through either exchange, step after step."""
    from minarrow_amd.host import Group

    n = 3_000_011
    rng = np.random.default_rng(members * 11 + len(exchange))
    ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    flts = rng.standard_normal(n) * 1e9
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    with Group([0] * members, exchange=exchange) as g:
        ctxs = [g.member_ctx(i) for i in range(members)]
        chunks = row_chunks(n, members)
        lens = [b - a for a, b in chunks]
        up = _uploaders(ctxs)  # (the members share device 0 here)
        di = [up[i].to_device(ints[a:b], 64) for i, (a, b) in enumerate(chunks)]
        df = [up[i].to_device(flts[a:b], 64) for i, (a, b) in enumerate(chunks)]
        dm = [up[i].to_device(bits, 16) for i in range(members)]
        offs = [a for a, _ in chunks]
        for masks in (None, dm):
            for _ in range(3):  # overlapped exchanges alternate between two record sets
                g.enqueue_sum_table([("l", 2, di, lens, masks, offs), ("g", 2, df, lens, masks, offs)])
                g.exchange()
            g.synchronize_for(20_000)
            fused = g.result(2)
            for _ in range(3):
                g.enqueue_sum("i64", 5, di, lens, masks, offs if masks else None)
                g.enqueue_sum("f64", 5, df, lens, masks, offs if masks else None)
                g.exchange()
            g.synchronize_for(20_000)
            assert g.result(5) == fused
            assert all(g.result(5, m) == fused for m in range(members)), "every member folds the same gathered records"
            if masks is None:
                want_i, want_c = oracle.sum_scalar(ints), n
                exact = math.fsum(flts.tolist())
            else:
                want_i, want_c = oracle.masked_sum(ints, bits, 0)
                valid = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
                exact = math.fsum(flts[valid].tolist())
            assert (fused[0], fused[1], fused[3]) == (want_i, want_c, want_c)
            assert abs(fused[2] - exact) <= math.ulp(exact)
        if exchange.startswith("rccl-overlap") and g.exchange_kind == "rccl":
            # two record sets alternate: a slot that was not enqueued in the last exchange's step is refused, not served stale
            with pytest.raises(ffi.MinarrowHipError) as e:
                g.result(2)  # the last exchanges carried slot 5 only
            assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "not enqueued" in str(e.value)
            assert g.result(5) == fused
        st = g.exchange_stats()
        if exchange.startswith("rccl") and g.exchange_kind == "rccl":
            assert st["rccl_ranks"] == members and st["samples"] >= 1 and st["all_gather_us"] > 0 and st["fold_us"] > 0
        else:
            assert st["rccl_ranks"] == 0 and st["all_gather_us"] == 0.0
        # the argument checks: an unsupported format, two columns on the same half of one record
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.enqueue_sum_table([("i", 0, di, lens)])
        assert e.value.status == ffi.MA_ERR_UNSUPPORTED
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.enqueue_sum_table([("l", 1, di, lens), ("L", 1, di, lens)])
        assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("members, failing", [(1, 0), pytest.param(2, 1, marks=pytest.mark.rehearsal),
                                              pytest.param(8, 3, marks=pytest.mark.rehearsal)])
def test_group_exchange_failure_on_a_member_aborts_instead_of_hanging(members, failing):
    """Per-member issue threads enqueue their own rank's all-gather; should one member fail in front of its call, the
    others' collectives could never complete (rehearsal: 7 members' all-gathers really are on the GPU, spinning for the 8th).
    The group then aborts its communicators and reports MA_ERR_DEVICE from exchange and synchronize (no hang); a rebuild — or
    destroying it — remains."""
    from minarrow_amd.host import Group

    with Group([0] * members, exchange="rccl") as g:
        if g.exchange_kind != "rccl":
            pytest.skip("RCCL is not available on this box")
        c = g.member_ctx(failing)
        n = 1 << 20
        col = c.alloc(n * 8)
        c.synth_iota("i64", col, n, 0)
        cols, lens = [col] * members, [n if m == failing else 0 for m in range(members)]  # one non-empty chunk, the failing member's
        g.enqueue_sum("i64", 0, cols, lens)
        g.exchange()
        g.synchronize()
        assert all(g.result(0, m)[:2] == (n * (n - 1) // 2, n) for m in range(members))
        ffi.check(g.lib.ma_group_test_fail_next_exchange(g.handle, failing))
        g.enqueue_sum("i64", 0, cols, lens)
        t0 = time.perf_counter()
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.exchange()
        assert e.value.status == ffi.MA_ERR_DEVICE and "aborted" in str(e.value) and f"member {failing}" in str(e.value)
        assert time.perf_counter() - t0 < 8.0 and g.is_broken == 1, "the peers' all-gathers were ended by the abort"
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.synchronize()
        assert e.value.status == ffi.MA_ERR_DEVICE
        with pytest.raises(ffi.MinarrowHipError):
            g.exchange()
        c.synchronize()  # the member's own stream is intact
        if members > 1:  # the same members, a fresh exchange
            g.rebuild_exchange("rccl", issue="caller")
            g.enqueue_sum("i64", 0, cols, lens)
            g.exchange()
            g.synchronize_for(20_000)
            assert all(g.result(0, m)[:2] == (n * (n - 1) // 2, n) for m in range(members))
