"""GPU side of the rank-sharded SuperTable path (minarrow_amd/parallel.py): the device fold of per-(batch, column)
records and the bit-granular join of gathered validity pieces. The collectives themselves are covered over gloo in
tests/test_parallel_gloo.py; a GPU box has one card, so the exchange here is the one-rank copy.

These checks hold device tensors, so torch must own the process's HIP runtime: it has to be imported BEFORE
libminarrow_hip.so is loaded (as bench.py does), which cannot be arranged inside a pytest session whose other tests
have already loaded the library. The checks therefore run in ONE child process (`python tests/test_gpu_sharded_table.py`)
started by the single pytest test at the bottom."""
import math
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def make_env():
    import torch  # first: the library then shares torch's HIP runtime

    from minarrow_amd.host import Context

    dev = torch.device("cuda", 0)
    ctx = Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    return torch, dev, ctx


def check_per_batch_records_fold_in_batch_order(env):
    torch, dev, ctx = env
    from minarrow_amd.parallel import ScalarExchange

    rng = np.random.default_rng(5)
    slots, rows = 3, [70_001, 64, 1]
    ex = ScalarExchange(dev, n_columns=2, slots_per_rank=slots)
    want_i, want_c, all_f = 0, 0, []
    keep = []
    ctx.set_async(True)
    for slot in range(slots):
        n = rows[slot]
        ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
        flts = rng.standard_normal(n) * 10.0 ** rng.integers(0, 14, size=n)
        bits = rng.integers(0, 256, size=n // 8 + 24, dtype=np.uint8)
        valid = np.unpackbits(bits, bitorder="little")[5:5 + n].astype(bool)
        d_i, d_f, d_m = (torch.from_numpy(a).to(dev) for a in (ints, flts, bits))
        keep += [d_i, d_f, d_m]
        ctx.sum_into("i64", d_i, n, out_sum=ex.slot_ptr(0, 0, slot), out_count=ex.slot_ptr(1, 0, slot), mask=d_m, mask_bit_offset=5)
        ctx.sum_into("f64", d_f, n, out_sum=ex.slot_ptr(2, 1, slot), dd_lo=ex.slot_ptr(3, 1, slot),
                     out_count=ex.slot_ptr(4, 1, slot), mask=d_m, mask_bit_offset=5)
        want_i += int(ints[valid].astype(object).sum())
        want_c += int(valid.sum())
        all_f += flts[valid].tolist()
    ex.exchange()
    ex.fold_on_device(ctx)
    ctx.set_async(False)
    ctx.synchronize()
    (isum, icnt, _, _), (_, _, fsum, fcnt) = ex.column_results()
    exact = math.fsum(all_f)
    assert isum == want_i & ((1 << 64) - 1) and icnt == want_c == fcnt
    assert abs(fsum - exact) <= math.ulp(exact)
    # the host fold of the same gathered records (the gloo path) agrees bit for bit
    ex._folded_on_device = False
    assert ex.column_results() == [(isum, icnt, 0.0, 0), (0, 0, fsum, fcnt)]


def check_bit_granular_join_of_gathered_validity(env, rows):
    torch, dev, ctx = env
    from minarrow_amd.parallel import join_bit_pieces

    rng = np.random.default_rng(sum(rows))
    width = max((r + 7) // 8 for r in rows) + 8
    staged = np.zeros(len(rows) * width, dtype=np.uint8)
    valid = []
    for r, n in enumerate(rows):
        v = rng.random(n) > 0.3
        valid.append(v)
        packed = np.packbits(v, bitorder="little")
        staged[r * width:r * width + len(packed)] = packed
    total = sum(rows)
    out = torch.full(((total + 63) // 64 * 8,), 0xAA, dtype=torch.uint8, device=dev)
    join_bit_pieces(ctx, torch.from_numpy(staged).to(dev), width, rows, out)
    ctx.synchronize()
    want = np.zeros((total + 63) // 64 * 8, dtype=np.uint8)
    packed = np.packbits(np.concatenate(valid), bitorder="little")
    want[:len(packed)] = packed
    assert out.cpu().numpy().tobytes() == want.tobytes()


def check_gather_consolidated_single_rank(env):
    torch, dev, ctx = env
    from minarrow_amd.parallel import gather_consolidated

    vals = torch.arange(1000, dtype=torch.int64, device=dev)
    bits = torch.from_numpy(np.packbits(np.arange(1000) % 3 != 0, bitorder="little")).to(dev)
    out = torch.zeros(1000, dtype=torch.int64, device=dev)
    out_bits = torch.zeros(128, dtype=torch.uint8, device=dev)
    assert gather_consolidated(vals, [1000], out, bits, out_bits, ctx=ctx)
    assert torch.equal(out, vals) and torch.equal(out_bits[:125], bits)
    assert not gather_consolidated(vals, [1000], out)


def main():
    env = make_env()
    check_per_batch_records_fold_in_batch_order(env)
    for rows in ([13, 1, 64, 7], [1000, 999, 0, 1001], [5]):
        check_bit_granular_join_of_gathered_validity(env, rows)
    check_gather_consolidated_single_rank(env)
    env[2].close()
    print("sharded-table checks ok")


@pytest.mark.gpu
def test_sharded_table_checks_in_a_torch_first_process():
    r = subprocess.run([sys.executable, str(Path(__file__).resolve())], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded-table checks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.rehearsal
@pytest.mark.parametrize("world", [2, 3])  # (with the two pytest sessions that is 5 processes on the card: the pool allows 6)
def test_rehearsal_sharded_table_records_across_rank_processes(world):
    """The same per-(batch, column) records with PEERS: 2 / 3 rank processes on this box's one GPU, 3 batches x 2 columns of records
    per rank in ONE ma_comm_sum_exchange through the loopback collective double (tests/loopback_rccl), folded per column over
    (rank, batch) in that order on every rank: i64 bit-exact, f64 within 1 ULP of the exactly rounded sum, the same bits everywhere."""
    from conftest import run_rank_processes

    outs = run_rank_processes(world, "sharded")
    assert all(o["i64_ok"] and o["f64_ok"] for o in outs), outs
    assert len({tuple(o["finals_bits"]) for o in outs}) == 1


if __name__ == "__main__":
    main()
