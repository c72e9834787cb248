"""The multi-rank paths, run for real semantics on the one GPU this pool's boxes have.

Everything N > 1 in the library — eight issue threads each blocked in its own all-gather until the peers arrive, the stamp wait
in front of a collective that has peers, the multi-member abort and rebuild, two scan lanes gating in front of a real
rendezvous, communicators across processes — needs N ranks, and RCCL refuses two ranks on one device. The loopback collective
double (tests/loopback_rccl: the twelve RCCL entry points the library resolves, with collective LIVENESS semantics on one
device) stands in for it: MINARROW_HIP_RCCL_PATH makes the library open it, and says so wherever a host could mistake the result
for a multi-GPU figure (REHEARSAL in ma_rccl_path, ma_group_exchange_note, bench.py's line).

RCCL is opened once per process and the number of hardware queues is fixed when the HIP runtime starts, so the tests marked
`rehearsal` (in test_gpu_group.py, test_gpu_guard.py, test_gpu_sharded_table.py, test_gpu_bench_modes.py) run in ONE child
pytest session started here. The job being rehearsed: benches/benchmark_parallel_simd.rs:81-98 over the GPUs of a node."""
import re
import subprocess
import sys
from pathlib import Path

import pytest

from conftest import LOOPBACK_LIB, ROOT, rehearsal_env

pytestmark = pytest.mark.gpu
SELFCHECK = LOOPBACK_LIB.parent / "selfcheck.bin"


def test_the_loopback_double_by_itself():
    """Without the library: all-gathers and all-reduces among 8 ranks on device 0 (one thread per rank; one thread in group calls),
    7 ranks blocked on a peer that never posts until ncclCommAbort ends them, 3 processes over a shared segment."""
    assert SELFCHECK.exists(), f"{SELFCHECK} is missing: make -C tests/cpp"
    for mode in (["single", "8", "100"], ["grouped", "8", "30"], ["abort", "8"], ["procs", "3", "50"]):
        r = subprocess.run([str(SELFCHECK), *mode], capture_output=True, text=True, timeout=120, env=rehearsal_env({"GPU_MAX_HW_QUEUES": "16"}))
        assert r.returncode == 0 and r.stdout.startswith("ok "), (mode, r.stdout[-500:], r.stderr[-1500:])


def test_every_multi_rank_path_through_the_loopback_double():
    assert LOOPBACK_LIB.exists(), f"{LOOPBACK_LIB} is missing: make -C tests/cpp"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", "--durations=8"],
                       capture_output=True, text=True, timeout=1500, env=rehearsal_env(), cwd=str(ROOT))
    out = ROOT / "gpurun_out"
    if out.is_dir():  # the child's own account, for profiles/
        (out / "rehearsal_suite.log").write_text(r.stdout[-200_000:] + "\n--- stderr ---\n" + r.stderr[-50_000:])
    tail = r.stdout[-6000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 68, tail
    print(r.stdout.strip().splitlines()[-1])
