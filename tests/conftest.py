import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
# The library's fault hooks (include/minarrow_hip_testing.h) act only when this is in the environment when the library is LOADED:
# the test sessions — and the bench.py / rank processes they start — are the hosts that want them.
os.environ.setdefault("MINARROW_HIP_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "big: a gpu test that holds 16 GB of HBM or more")
    config.addinivalue_line("markers", "rehearsal: a gpu test of the MULTI-RANK paths with several ranks on this box's one GPU; it runs only in the "
                                       "child session tests/test_gpu_rehearsal.py starts (MINARROW_REHEARSAL=1), where the library opens "
                                       "the loopback collective double of tests/loopback_rccl instead of RCCL")


# The loopback collective double (test infrastructure) and the environment of the one child session that runs through it. RCCL is
# opened once per process, and the number of hardware queues is fixed when the HIP runtime starts: neither can change inside a
# pytest session that has already run single-rank tests against the real RCCL — hence a child process.
LOOPBACK_LIB = ROOT / "tests" / "loopback_rccl" / "libloopback_rccl.so"
REHEARSAL = os.environ.get("MINARROW_REHEARSAL") == "1"


def rehearsal_env(extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # 8 members: the streams that carry their collectives (which the library puts into the high priority class in a rehearsal) need a
    # hardware queue each — a rank's collective kernel spins until its peers' kernels have run, and streams in one hardware queue
    # run in order — and the process must stay under the 23 queues a device runs at a time (tests/loopback_rccl/selfcheck.cpp `slots`)
    env.update({"MINARROW_REHEARSAL": "1", "MINARROW_HIP_RCCL_PATH": str(LOOPBACK_LIB), "GPU_MAX_HW_QUEUES": "8",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra or {})
    return env


def run_rank_processes(world, scenario, timeout=240):
    """`world` processes of tests/rehearsal_ranks.py sharing device 0 (one context and one ma_comm rank each): ncclCommInitRank over
    the double's shared segment. Returns every rank's JSON line."""
    import json
    import subprocess

    from minarrow_amd.host import Comm

    script = ROOT / "tests" / "rehearsal_ranks.py"
    ids = [Comm.unique_id().hex() for _ in range(3)]  # the double's ncclGetUniqueId: a segment name, no GPU call
    rank_env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}  # one rank per process: the runtime's default pool will do
    procs = [subprocess.Popen([sys.executable, str(script), scenario, str(r), str(world), *ids], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=rank_env) for r in range(world)]
    outs = []
    for r, p in enumerate(procs):
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError(f"rank {r} of scenario {scenario} did not finish within {timeout} s")
        assert p.returncode == 0, (r, so[-1500:], se[-3000:])
        outs.append(json.loads(so.strip().splitlines()[-1]))
    return outs


def pytest_collection_modifyitems(config, items):
    """`rehearsal` tests run in the rehearsal session and nowhere else; that session runs nothing but them."""
    keep, drop = [], []
    for item in items:
        (keep if (item.get_closest_marker("rehearsal") is not None) == REHEARSAL else drop).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def ctx():
    """A library context on device 0. GPU tests fail loudly (never skip, never fall back) when the HIP
    library is missing or no device is visible."""
    from minarrow_amd.host import Context

    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.lib()
    return o
