"""First-contact safety of the multi-GPU forms: bounded waits (ma_group_synchronize_for / ma_comm_synchronize_for), the way
down (ma_group_rebuild_exchange, ma_group_set_handoff), the self-tests (ma_group_selftest / ma_comm_selftest) and the fault
hooks that drive them on a one-GPU box (ma_*_test_stall_next_exchange: an exchange that never completes;
ma_*_test_corrupt_next_exchange: finals that are wrong on one member). The job they guard is the reference's partitioned
reduction, benches/benchmark_parallel_simd.rs:81-125."""
import time

import numpy as np
import pytest
from conftest import run_rank_processes

from minarrow_amd import ffi
from minarrow_amd.host import (SELFTEST_EXCHANGE, SELFTEST_EXCHANGE_ALL_FORMS, SELFTEST_PEER_COPIES, SELFTEST_STAMPS, Comm, Context,
                               Group)

pytestmark = pytest.mark.gpu

N = 1 << 20
TRI = N * (N - 1) // 2


def _column(g, member=0):
    c = g.member_ctx(member)
    col_i, col_f = c.alloc(N * 8), c.alloc(N * 8)
    c.synth_iota("i64", col_i, N, 0)
    c.synth_iota("f64", col_f, N, 0)
    return c, col_i, col_f


def _step(g, col_i, col_f, fused=True):
    if fused:
        g.enqueue_sum_table([("l", 0, [col_i], [N]), ("g", 0, [col_f], [N])])
    else:
        g.enqueue_sum("i64", 0, [col_i], [N])
        g.enqueue_sum("f64", 0, [col_f], [N])
    g.exchange()


def _rccl_group(exchange, issue="threads", members=1):
    g = Group([0] * members, exchange=exchange, issue=issue)
    if g.exchange_kind != "rccl":
        g.close()
        pytest.skip("RCCL is not available on this box")
    return g


def test_stamp_words_are_waitable_and_say_what_they_are(ctx):
    s = ctx.stamp_alloc()
    try:
        assert ctx.lib.ma_stamp_is_signal(s) in (0, 1)  # 1: the runtime's signal memory (8 bytes), 0: a plain device word
        col = ctx.alloc(N * 8)
        out = ctx.alloc(64)
        ctx.synth_iota("i64", col, N, 0)
        call = ctx.prepare_sum_fused([("l", col, N, out.ptr)], stamp=s)
        call(7)
        ctx.synchronize()
        assert int(out.download(np.int64, 1)[0]) == TRI
    finally:
        ctx.stamp_free(s)
    assert ctx.lib.ma_stamp_is_signal(s) == -1


@pytest.mark.parametrize("exchange", ["rccl", "rccl-overlap", "host"])
@pytest.mark.parametrize("issue", ["threads", "caller"])
def test_synchronize_for_is_synchronize_when_nothing_hangs(exchange, issue):
    g = Group([0], exchange=exchange if exchange == "host" else exchange + "-or-host", issue=issue)
    try:
        c, col_i, col_f = _column(g)
        for _ in range(5):
            _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.is_broken == 0
        assert g.result(0) == (TRI, N, float(TRI), N)
        g.synchronize_for(0)  # <= 0: no deadline, plain synchronize
    finally:
        g.close()


@pytest.mark.parametrize("exchange, issue", [("rccl", "threads"), ("rccl", "caller"), ("rccl-overlap", "threads"),
                                             ("rccl-overlap", "caller"), ("rccl-overlap-lanes", "threads"), ("host", "threads")])
def test_a_stalled_exchange_ends_in_an_error_not_a_hang_and_the_group_can_be_rebuilt(exchange, issue):
    g = Group([0], exchange=exchange, issue=issue) if exchange == "host" else _rccl_group(exchange, issue)
    try:
        c, col_i, col_f = _column(g)
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        g.test_stall_next_exchange(0)
        _step(g, col_i, col_f)
        t0 = time.perf_counter()
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.synchronize_for(300)
        waited = time.perf_counter() - t0
        assert e.value.status == ffi.MA_ERR_DEVICE and "member 0" in str(e.value) and "did not finish within 300 ms" in str(e.value)
        assert 0.25 < waited < 8.0, waited
        assert g.is_broken == 1, "the abort released the stall: the streams must have run empty"
        for call in (g.exchange, g.synchronize, lambda: g.synchronize_for(100), lambda: g.selftest(100, raise_on_failure=True)):
            with pytest.raises(ffi.MinarrowHipError) as e2:
                call()
            assert e2.value.status == ffi.MA_ERR_DEVICE and "rebuild" in str(e2.value)
        c.synchronize()  # the member's own context is intact, and so are its buffers
        c.set_async(False)
        assert c.sum("i64", col_i, N) == (TRI, N)
        c.set_async(True)
        # one notch down, same members, same columns
        g.rebuild_exchange("rccl" if exchange.startswith("rccl") else "host", issue="caller")
        assert g.is_broken == 0 and not g.overlapped and g.issue_kind == "caller"
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.result(0) == (TRI, N, float(TRI), N)
        # ... and all the way down
        g.rebuild_exchange("host")
        assert g.exchange_kind == "host"
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.result(0) == (TRI, N, float(TRI), N)
        # a stall can be armed again after a rebuild (its word carries a sequence, not a flag)
        g.test_stall_next_exchange(0)
        _step(g, col_i, col_f)
        with pytest.raises(ffi.MinarrowHipError):
            g.synchronize_for(200)
        assert g.is_broken == 1
    finally:
        g.close()  # destroying a broken group must not block either


def test_destroying_a_group_with_an_armed_stall_releases_it():
    g = _rccl_group("rccl-overlap")
    c, col_i, col_f = _column(g)
    g.test_stall_next_exchange(0)
    _step(g, col_i, col_f)
    t0 = time.perf_counter()
    g.close()
    assert time.perf_counter() - t0 < 10.0


def test_the_library_s_own_destroy_is_bounded_too(ctx, monkeypatch):
    """ma_group_destroy / ma_comm_destroy as a C or Rust host calls them, without host.py's drain in front: a stalled exchange in
    flight must not keep them (MINARROW_HIP_DESTROY_WAIT_MS for work in flight — 10 s by default —, then the abort path)."""
    monkeypatch.setenv("MINARROW_HIP_DESTROY_WAIT_MS", "300")
    col_i, col_f = ctx.alloc(N * 8), ctx.alloc(N * 8)  # the fixture's context owns the columns: nothing dies with the group
    ctx.synth_iota("i64", col_i, N, 0)
    ctx.synth_iota("f64", col_f, N, 0)
    ctx.synchronize()
    g = _rccl_group("rccl-overlap")
    g.test_stall_next_exchange(0)
    _step(g, col_i, col_f)
    for m in g._members:
        m.handle = None  # views of contexts the group owns
    t0 = time.perf_counter()
    g.lib.ma_group_destroy(g.handle)
    g.handle = None
    assert time.perf_counter() - t0 < 8.0

    comm = _comm(ctx)
    local, gathered, final = _records(ctx)
    ctx.set_async(True)
    try:
        ctx.sum_into("i64", col_i, N, out_sum=local.ptr, out_count=local.ptr + 8)
        comm.test_stall_next_exchange()
        comm.sum_exchange(local, 1, 1, gathered, final)
        t0 = time.perf_counter()
        comm.lib.ma_comm_destroy(comm.handle)
        comm.handle = None
        assert time.perf_counter() - t0 < 8.0
        ctx.synchronize()  # the abort path released the held stream
    finally:
        ctx.set_async(False)


@pytest.mark.parametrize("exchange", ["rccl", "rccl-overlap", "host"])
def test_a_corrupted_exchange_is_visible_in_the_finals_and_only_once(exchange):
    g = Group([0], exchange=exchange) if exchange == "host" else _rccl_group(exchange)
    try:
        c, col_i, col_f = _column(g)
        g.test_corrupt_next_exchange(0)
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.is_broken == 0
        assert g.result(0)[0] != TRI, "the hook flips the integer sum"
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.result(0) == (TRI, N, float(TRI), N)
    finally:
        g.close()


def test_handoff_can_be_switched_without_a_rebuild():
    g = _rccl_group("rccl-overlap")
    try:
        c, col_i, col_f = _column(g)
        first = g.handoff
        assert first in ("stamp", "event")
        assert ("stamps in" in g.exchange_note) == (first == "stamp"), g.exchange_note
        for kind in ("event", "stamp", "event"):
            g.set_handoff(kind)
            assert g.handoff == ("event" if kind == "event" else first)
            for _ in range(3):
                _step(g, col_i, col_f)
            g.synchronize_for(20_000)
            assert g.result(0) == (TRI, N, float(TRI), N)
        assert g.lib.ma_group_set_handoff(g.handle, 7) == ffi.MA_ERR_INVALID_ARGUMENT
    finally:
        g.close()
    with Group([0], exchange="host") as h:
        assert h.handoff is None


@pytest.mark.parametrize("exchange, issue", [("rccl", "threads"), ("rccl", "caller"), ("rccl-overlap", "threads"),
                                             ("rccl-overlap", "caller"), ("rccl-overlap-lanes", "threads"), ("host", "threads")])
def test_selftest_passes_in_every_form_this_box_can_run(exchange, issue):
    g = Group([0], exchange=exchange, issue=issue) if exchange == "host" else _rccl_group(exchange, issue)
    try:
        rep = g.selftest(20_000)  # the configured form + stamps + peer copies
        assert rep["ok"] and rep["text"].startswith("PASS"), rep
        assert rep["members"] == 1 and rep["devices"] == 1 and rep["peer_pairs"] == 0
        assert len(rep["forms"]) == 1 and all(f["ok"] and f["us"] > 0 for f in rep["forms"].values()), rep
        if exchange.startswith("rccl-overlap"):
            (name,) = rep["forms"]
            assert name == f"overlap-{g.handoff}/{issue}"
            if g.handoff == "stamp":  # one wait per record set; with two scan lanes also the lanes' early-stamp gate, both ways
                assert rep["stamp_waits"] == rep["stamp_waits_ok"] == (4 if exchange.endswith("lanes") else 2)
            assert g.scan_lanes == exchange.endswith("lanes")
        if exchange.startswith("rccl"):
            assert rep["rccl_ranks"] == 1 and rep["exchange"] == "rccl"
        allf = g.selftest(20_000, SELFTEST_EXCHANGE | SELFTEST_EXCHANGE_ALL_FORMS)
        want = {"rccl": 2, "rccl-overlap": 4 if g.handoff == "stamp" else 2, "rccl-overlap-lanes": 4, "host": 1}[exchange]
        assert allf["ok"] and len(allf["forms"]) == want and all(f["ok"] for f in allf["forms"].values()), allf
        assert g.issue_kind == issue, "the self-test leaves the issue form as it found it"
        # the group still does its job afterwards, and the test left nothing behind in its records
        c, col_i, col_f = _column(g)
        _step(g, col_i, col_f)
        g.synchronize_for(20_000)
        assert g.result(0) == (TRI, N, float(TRI), N)
    finally:
        g.close()


def test_selftest_over_several_members_sharing_the_device():
    """[0] * n host-fold groups: the member-ordered fold of n tagged records on every member."""
    with Group([0] * 5, exchange="host") as g:
        rep = g.selftest(20_000)
        assert rep["ok"] and rep["members"] == 5 and rep["devices"] == 1 and rep["forms"]["host-fold"]["ok"], rep


def test_selftest_reports_a_stall_and_a_corruption():
    g = _rccl_group("rccl-overlap")
    try:
        g.test_corrupt_next_exchange(0)
        rep = g.selftest(20_000, SELFTEST_EXCHANGE)
        assert not rep["ok"] and rep["text"].startswith("FAIL") and rep["failed_member"] == 0 and not rep["timed_out"], rep
        assert "rank order" in rep["text"] or "fold" in rep["text"]
        assert g.is_broken == 0
        assert g.selftest(20_000)["ok"]  # the hook is spent; the group is fine
        g.test_stall_next_exchange(0)
        rep = g.selftest(300, SELFTEST_EXCHANGE)
        assert not rep["ok"] and rep["timed_out"] and rep["failed_form"].startswith("overlap-"), rep
        assert g.is_broken == 1
        g.rebuild_exchange("rccl-overlap")
        assert g.selftest(20_000)["ok"]
    finally:
        g.close()


# ---- the multi-process communicator ------------------------------------------------------------------------------------------


def _comm(ctx):
    try:
        return Comm(ctx, Comm.unique_id(), 0, 1)
    except ffi.MinarrowHipError as e:
        pytest.skip(f"RCCL is not available on this box: {e}")


def _records(ctx):
    local, gathered, final = ctx.alloc(64), ctx.alloc(64), ctx.alloc(32)
    for b, nbytes in ((local, 64), (gathered, 64), (final, 32)):
        ctx.dev_memset(b, 0, nbytes)
    return local, gathered, final


def test_comm_selftest_and_bounded_wait(ctx):
    comm = _comm(ctx)
    try:
        rep = comm.selftest(20_000)
        assert rep["ok"] and rep["rccl_ranks"] == 1, rep
        assert rep["forms"]["in-stream/caller"]["ok"] and rep["forms"]["overlap-event/caller"]["ok"], rep
        col = ctx.alloc(N * 8)
        ctx.synth_iota("i64", col, N, 0)
        local, gathered, final = _records(ctx)
        ctx.set_async(True)
        ctx.sum_into("i64", col, N, out_sum=local.ptr, out_count=local.ptr + 8)
        comm.sum_exchange(local, 1, 1, gathered, final)
        comm.synchronize_for(20_000)
        assert [int(v) for v in final.download(np.uint64, 2)] == [TRI, N]
        # corrupted once, then fine
        comm.test_corrupt_next_exchange()
        comm.sum_exchange(local, 1, 1, gathered, final)
        comm.synchronize_for(20_000)
        assert int(final.download(np.uint64, 1)[0]) != TRI
        comm.sum_exchange_overlapped(0, local, 1, 1, gathered, final)
        comm.synchronize_for(20_000)
        assert [int(v) for v in final.download(np.uint64, 2)] == [TRI, N]
        # a stall: error after the deadline, the communicator is dead, the context is not
        comm.test_stall_next_exchange()
        comm.sum_exchange_overlapped(1, local, 1, 1, gathered, final)
        t0 = time.perf_counter()
        with pytest.raises(ffi.MinarrowHipError) as e:
            comm.synchronize_for(300)
        assert e.value.status == ffi.MA_ERR_DEVICE and "exchange stream" in str(e.value) and time.perf_counter() - t0 < 8.0
        assert comm.is_broken == 1
        with pytest.raises(ffi.MinarrowHipError) as e:
            comm.sum_exchange(local, 1, 1, gathered, final)
        assert e.value.status == ffi.MA_ERR_DEVICE and "aborted" in str(e.value)
        ctx.set_async(False)
        assert ctx.sum("i64", col, N) == (TRI, N)
    finally:
        comm.close()
    # the ranks agree on a new communicator: a fresh id, the same context
    again = _comm(ctx)
    try:
        assert again.selftest(20_000)["ok"]
        again.abort()  # what a rank does when ANOTHER rank reports the deadline
        assert again.is_broken == 1
    finally:
        again.close()


def test_two_scan_lanes_give_every_step_its_own_results():
    """MA_GROUP_SCAN_LANES: consecutive stamped steps run on two scan streams per member, each gated on the early stamp of the one
    before. Steps over DIFFERENT columns in turn (so that a step served by the wrong lane's records, or started before its
    input was ready, would show), every step's finals checked; work the host enqueues itself on the member's context in between
    (a column rewritten in place) is seen by the step that follows it, on either lane; marks bracket the scan on whichever lane."""
    g = _rccl_group("rccl-overlap-lanes")
    try:
        assert g.scan_lanes and "two scan lanes" in g.exchange_note
        c = g.member_ctx(0)
        sizes = [N, N // 2 + 64, N - 128, 3 * N // 4]
        cols = []
        for k, n in enumerate(sizes):
            ci, cf = c.alloc(n * 8), c.alloc(n * 8)
            c.synth_iota("i64", ci, n, k)
            c.synth_iota("f64", cf, n, k)
            cols.append((ci, cf, n, k))
        want = lambda n, k: (n * (n - 1) // 2 + k * n, n, float(n * (n - 1) // 2 + k * n), n)  # noqa: E731
        for step in range(12):
            ci, cf, n, k = cols[step % len(cols)]
            if step % 5 == 4:
                g.mark_next_scan(2 * step, 2 * step + 1)
            g.enqueue_sum_table([("l", 0, [ci], [n]), ("g", 0, [cf], [n])])
            g.exchange()
            if step % 3 == 2:  # read every third step's finals; the others are overwritten two steps later, unread
                g.synchronize_for(20_000)
                assert g.result(0) == want(n, k), (step, g.result(0))
                if step % 5 == 4:
                    assert 0 < g.mark_elapsed_ms(0, 2 * step, 2 * step + 1) < 50
        # the host's own work on the member's context between two steps: the column the NEXT step scans is rewritten in place
        ci, cf, n, k = cols[0]
        for rounds in range(4):  # whichever lane the next step lands on
            c.synth_iota("i64", ci, n, 100 + rounds)
            c.synth_iota("f64", cf, n, 100 + rounds)
            g.enqueue_sum_table([("l", 0, [ci], [n]), ("g", 0, [cf], [n])])
            g.exchange()
            g.join_lanes()  # what follows on the member's context comes after the step, whichever lane ran it
        g.synchronize_for(20_000)
        assert g.result(0) == want(n, 103)
        # non-stamped enqueues still work on a group with lanes (they take the set's lane, ordered by events)
        g.enqueue_sum("i64", 1, [ci], [n])
        g.exchange()
        g.enqueue_sum("i64", 1, [cols[1][0]], [cols[1][2]])
        g.exchange()
        g.synchronize_for(20_000)
        assert g.result(1)[:2] == want(cols[1][2], 1)[:2]
    finally:
        g.close()


# ---- the same, with PEERS: several members on this box's one GPU through the loopback collective double -----------------------
# (tests/test_gpu_rehearsal.py starts the one session these run in; tests/loopback_rccl is the stand-in for RCCL)

REHEARSAL_FORMS = [("rccl", "threads"), ("rccl", "caller"), ("rccl-overlap", "threads"), ("rccl-overlap", "caller"),
                   ("rccl-overlap-lanes", "threads"), ("rccl-overlap-lanes", "caller")]


def _table(g):
    """Per member an i64 and an f64 column of N rows (0, 1, 2, ...) in that member's own allocations."""
    ctxs = [g.member_ctx(m) for m in range(g.size)]
    cols_i, cols_f = [c.alloc(N * 8) for c in ctxs], [c.alloc(N * 8) for c in ctxs]
    for c, a, b in zip(ctxs, cols_i, cols_f):
        c.synth_iota("i64", a, N, 0)
        c.synth_iota("f64", b, N, 0)
    return ctxs, cols_i, cols_f


def _table_step(g, cols_i, cols_f, fused=True):
    lens = [N] * g.size
    if fused:
        g.enqueue_sum_table([("l", 0, cols_i, lens), ("g", 0, cols_f, lens)])
    else:
        g.enqueue_sum("i64", 0, cols_i, lens)
        g.enqueue_sum("f64", 0, cols_f, lens)
    g.exchange()


def _want(g):
    k = g.size
    return (TRI * k, N * k, float(TRI * k), N * k)


def _same_on_every_member(g, want):
    got = [g.result(0, m) for m in range(g.size)]
    assert all(r == want for r in got), got


@pytest.mark.rehearsal
@pytest.mark.parametrize("members, stalled, exchange, issue", [(8, 3, x, i) for x, i in REHEARSAL_FORMS] +
                         [(2, 1, "rccl", "threads"), (2, 0, "rccl-overlap-lanes", "caller")])
def test_rehearsal_a_stalled_member_among_peers(members, stalled, exchange, issue):
    """Member 3 of 8 never starts its all-gather: the seven others' collectives are on the GPU, waiting for it — what a lost peer
    looks like over xGMI. ma_group_synchronize_for names the member within its deadline, the abort ends all eight communicators,
    every stream runs empty, and one notch down the same members and columns do the job."""
    g = _rccl_group(exchange, issue, members)
    try:
        ctxs, cols_i, cols_f = _table(g)
        for _ in range(3):
            _table_step(g, cols_i, cols_f)
        g.synchronize_for(20_000)
        _same_on_every_member(g, _want(g))
        g.test_stall_next_exchange(stalled)
        _table_step(g, cols_i, cols_f)
        t0 = time.perf_counter()
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.synchronize_for(400)
        waited = time.perf_counter() - t0
        text = str(e.value)
        assert e.value.status == ffi.MA_ERR_DEVICE and f"member {stalled} " in text and "did not finish within 400 ms" in text
        assert 0.35 < waited < 8.0, waited
        assert g.is_broken == 1, "the abort released the stall and ended the peers' collectives: every stream must have run empty"
        with pytest.raises(ffi.MinarrowHipError) as e2:
            g.exchange()
        assert "rebuild" in str(e2.value)
        for c, col in zip(ctxs, cols_i):  # every member's own context and buffers are intact
            c.synchronize()
            c.set_async(False)
            assert c.sum("i64", col, N) == (TRI, N)
            c.set_async(True)
        down = {"rccl-overlap-lanes": "rccl-overlap", "rccl-overlap": "rccl", "rccl": "rccl"}[exchange]
        g.rebuild_exchange(down, issue="caller" if exchange == "rccl" else issue)
        assert g.is_broken == 0 and g.exchange_kind == "rccl"
        for fused in (True, False, True):
            _table_step(g, cols_i, cols_f, fused)
        g.synchronize_for(20_000)
        _same_on_every_member(g, _want(g))
        assert g.exchange_stats()["rccl_ranks"] == members
    finally:
        g.close()


@pytest.mark.rehearsal
@pytest.mark.parametrize("members", [2, 4, 8])
@pytest.mark.parametrize("exchange, issue", REHEARSAL_FORMS)
def test_rehearsal_selftest_with_peers(members, exchange, issue):
    """ma_group_selftest where its peer columns are not "n/a": rank-tagged records of 2 / 4 / 8 members through the exchange in the
    configured form and in every form, the gathered blocks in rank order and the member-ordered fold on EVERY member."""
    g = _rccl_group(exchange, issue, members)
    try:
        rep = g.selftest(20_000)
        assert rep["ok"] and rep["text"].startswith("PASS"), rep
        assert rep["members"] == members and rep["devices"] == 1 and rep["rccl_ranks"] == members and rep["exchange"] == "rccl"
        assert f"{members} members on 1 device(s)" in rep["text"]
        assert len(rep["forms"]) == 1 and all(f["ok"] and f["us"] > 0 for f in rep["forms"].values()), rep
        if exchange.startswith("rccl-overlap") and g.handoff == "stamp":
            assert rep["stamp_waits"] == rep["stamp_waits_ok"] == members * (4 if exchange.endswith("lanes") else 2)
        allf = g.selftest(20_000, SELFTEST_EXCHANGE | SELFTEST_EXCHANGE_ALL_FORMS)
        want = {"rccl": 2, "rccl-overlap": 4 if g.handoff == "stamp" else 2, "rccl-overlap-lanes": 4}[exchange]
        assert allf["ok"] and len(allf["forms"]) == want and all(f["ok"] for f in allf["forms"].values()), allf
        assert g.issue_kind == issue
        ctxs, cols_i, cols_f = _table(g)
        _table_step(g, cols_i, cols_f)
        g.synchronize_for(20_000)
        _same_on_every_member(g, _want(g))
    finally:
        g.close()


@pytest.mark.rehearsal
@pytest.mark.parametrize("exchange", ["rccl", "rccl-overlap"])
def test_rehearsal_a_corrupted_member_is_the_only_one_with_wrong_finals(exchange):
    g = _rccl_group(exchange, "threads", 8)
    try:
        ctxs, cols_i, cols_f = _table(g)
        g.test_corrupt_next_exchange(5)  # member 5's copy of member 0's record is flipped in front of ITS fold
        _table_step(g, cols_i, cols_f)
        g.synchronize_for(20_000)
        got = [g.result(0, m) for m in range(8)]
        assert got[5][0] != _want(g)[0] and all(got[m] == _want(g) for m in range(8) if m != 5), got
        rep = None
        g.test_corrupt_next_exchange(6)
        rep = g.selftest(20_000, SELFTEST_EXCHANGE)
        assert not rep["ok"] and rep["failed_member"] == 6 and not rep["timed_out"] and g.is_broken == 0, rep
        g.test_stall_next_exchange(2)
        rep = g.selftest(400, SELFTEST_EXCHANGE)
        assert not rep["ok"] and rep["timed_out"] and g.is_broken == 1, rep
        g.rebuild_exchange(exchange)
        assert g.selftest(20_000)["ok"]
    finally:
        g.close()


@pytest.mark.rehearsal
def test_rehearsal_destroying_a_group_of_eight_with_a_stalled_member_is_bounded(monkeypatch):
    monkeypatch.setenv("MINARROW_HIP_DESTROY_WAIT_MS", "300")
    g = _rccl_group("rccl-overlap-lanes", "threads", 8)
    ctxs, cols_i, cols_f = _table(g)
    g.test_stall_next_exchange(7)
    _table_step(g, cols_i, cols_f)
    t0 = time.perf_counter()
    g.close()
    assert time.perf_counter() - t0 < 10.0


@pytest.mark.rehearsal
@pytest.mark.parametrize("issue", ["threads", "caller"])
def test_rehearsal_two_scan_lanes_in_front_of_a_rendezvous(issue):
    """MA_GROUP_SCAN_LANES with 4 members: consecutive stamped steps alternate between two scan streams per member, each gated on
    the early stamp of the step before, and every step's exchange is a rendezvous of all four. Steps over DIFFERENT columns in
    turn; every third step's finals are read on every member."""
    g = _rccl_group("rccl-overlap-lanes", issue, 4)
    try:
        assert g.scan_lanes and "two scan lanes" in g.exchange_note
        ctxs = [g.member_ctx(m) for m in range(4)]
        sizes = [N, N // 2 + 64, N - 128]
        tables = []
        for k, n in enumerate(sizes):
            ci, cf = [c.alloc(n * 8) for c in ctxs], [c.alloc(n * 8) for c in ctxs]
            for m, c in enumerate(ctxs):
                c.synth_iota("i64", ci[m], n, k + m)
                c.synth_iota("f64", cf[m], n, k + m)
            tables.append((ci, cf, n, k))

        def want(n, k):
            s = sum(n * (n - 1) // 2 + (k + m) * n for m in range(4))
            return (s, 4 * n, float(s), 4 * n)

        for step in range(18):
            ci, cf, n, k = tables[step % 3]
            g.enqueue_sum_table([("l", 0, ci, [n] * 4), ("g", 0, cf, [n] * 4)])
            g.exchange()
            if step % 3 == 2 or step == 17:
                g.synchronize_for(20_000)
                _same_on_every_member(g, want(n, k))
        g.set_scan_lanes(False)  # the same steps on one scan stream per member, without a rebuild
        for step in range(4):
            ci, cf, n, k = tables[step % 3]
            g.enqueue_sum_table([("l", 0, ci, [n] * 4), ("g", 0, cf, [n] * 4)])
            g.exchange()
        g.synchronize_for(20_000)
        _same_on_every_member(g, want(tables[0][2], 0))
    finally:
        g.close()


@pytest.mark.rehearsal
@pytest.mark.parametrize("world", [2, 3])
def test_rehearsal_comm_across_processes(world):
    """ma_comm_* with PEERS: 2 / 3 processes (one context each on device 0), the exchange in-stream, overlapped on an event and
    overlapped on the scan's stamp, the communicator's self-test in every form — every rank ends up with the same finals, folded
    in rank order."""
    outs = run_rank_processes(world, "exchange")
    n = 1 << 20
    total = n * world
    want = [total * (total - 1) // 2, total]
    for o in outs:
        assert o["selftest_ok"] and o["rccl_ranks"] == world and o["rehearsal"], o
        assert o["forms"] == ["in-stream/caller", "overlap-event/caller", "overlap-stamp/caller"], o
        for name in ("in_stream", "overlapped_event", "overlapped_stamp", "all_reduce"):
            assert o[name] == want, (name, o)
        assert o["f64_within_1ulp"], o
    assert len({tuple(o["finals_bits"]) for o in outs}) == 1, "bit-identical finals on every rank"


@pytest.mark.rehearsal
def test_rehearsal_a_stalled_rank_and_a_vanished_rank():
    """Rank 1's exchange never starts (stall hook): rank 0's collective waits on the GPU for it; ma_comm_synchronize_for ends in an
    error on BOTH ranks within its deadline, both abort, their contexts stay usable, and a new communicator from a fresh id works.
    Then rank 1 leaves without a word: rank 0's next exchange times out the same way instead of hanging."""
    outs = run_rank_processes(2, "stall")
    for o in outs:
        assert o["timed_out"] and o["broken"] == 1 and o["waited_s"] < 8.0 and o["ctx_ok"] and o["second_comm_ok"], o
    assert outs[0]["vanished_peer_timed_out"] and outs[0]["vanished_waited_s"] < 8.0 and outs[0]["ctx_ok_after_vanish"], outs[0]


@pytest.mark.rehearsal
@pytest.mark.parametrize("exchange, issue", [("rccl-overlap-lanes", "threads"), ("rccl-overlap", "caller"), ("rccl", "threads")])
def test_rehearsal_soak_of_steps_with_a_jittery_host(exchange, issue):
    """Ordering under a host that does not keep time: 8 members, 240 steps over THREE different tables in turn (a step served by the
    wrong record set, started before its input was ready, or folded from a peer's stale block shows as a wrong total), random
    pauses between enqueue, exchange and the occasional bounded wait, columns rewritten in place between steps on some members,
    the finals read on a random member every few steps."""
    rng = np.random.default_rng(len(exchange) * 7 + len(issue))
    g = _rccl_group(exchange, issue, 8)
    try:
        ctxs = [g.member_ctx(m) for m in range(8)]
        n = 1 << 18
        tables = []
        for k in range(3):
            ci, cf = [c.alloc(n * 8) for c in ctxs], [c.alloc(n * 8) for c in ctxs]
            for m, c in enumerate(ctxs):
                c.synth_iota("i64", ci[m], n, 10 * k + m)
                c.synth_iota("f64", cf[m], n, 10 * k + m)
            tables.append([ci, cf, [10 * k + m for m in range(8)]])

        def want(starts):
            s = sum(n * (n - 1) // 2 + st * n for st in starts)
            return (s, 8 * n, float(s), 8 * n)

        checked = 0
        for step in range(240):
            ci, cf, starts = tables[step % 3]
            if step % 17 == 16:  # the host rewrites one member's chunk of the table the NEXT use of which is three steps away ...
                m = int(rng.integers(0, 8))
                g.join_lanes()  # ... behind whatever lane still reads it
                starts[m] += 1000
                ctxs[m].synth_iota("i64", ci[m], n, starts[m])
                ctxs[m].synth_iota("f64", cf[m], n, starts[m])
            if rng.random() < 0.3:
                time.sleep(float(rng.choice([0.0001, 0.0005, 0.002])))
            g.enqueue_sum_table([("l", 0, ci, [n] * 8), ("g", 0, cf, [n] * 8)])
            if rng.random() < 0.3:
                time.sleep(float(rng.choice([0.0001, 0.001])))
            g.exchange()
            if rng.random() < 0.2 or step == 239:
                g.synchronize_for(20_000)
                assert g.result(0, int(rng.integers(0, 8))) == want(starts), step
                checked += 1
        assert checked >= 20 and g.is_broken == 0
    finally:
        g.close()
