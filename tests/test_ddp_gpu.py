"""GPU test of the data-parallel step logic: 2 ranks (gloo, both on cuda:0), each stepping its OWN clips, must reproduce a
one-process two-replica emulation of the same program (gradient arenas summed by hand, tests/ddp_worker.py "emulate2") —
eager (hook-driven bucketed reductions) and graph mode (phase graphs + reductions between)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, world, tmp_path, which="ganomaly", rccl_single=False):
    out = str(tmp_path / ("%s_%s_%d%s.json" % (which, mode, world, "_rccl" if rccl_single else "")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if rccl_single:
            env.update(VFD_DIST_SINGLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), mode, out, which], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o.decode()[-3000:]
    return json.load(open(out))


@pytest.mark.parametrize("mode", ["eager", "graph"])
@pytest.mark.parametrize("which", ["ganomaly", "anogan", "mygan"])
def test_two_ranks_match_single_process(which, mode, dev, tmp_path):
    """All three models: ganomaly / mygan (one backward per net and step), anogan (TWO backward passes into netD before
    its reduction: GradReducer.arm(passes=2)); graph mode runs the models' step_program() with asynchronous
    reductions between the captured graphs.  Ranks hold DIFFERENT clips (and noise): the expected values come from the
    collective-free two-replica emulation, and the workers assert after every reduction that their gradient arenas are equal."""
    one = _run("emulate2", 1, tmp_path, which)
    two = _run(mode, 2, tmp_path, which)
    assert two["world"] == 2
    for k, v in one["errors"].items():
        assert abs(two["errors"][k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, two["errors"][k], v)
    for k, v in one["sums"].items():
        assert abs(two["sums"][k] - v) <= 1e-4 * max(abs(v), 1.0), (k, two["sums"][k], v)


@pytest.mark.parametrize("mode", ["eager", "graph"])
@pytest.mark.parametrize("which", ["ganomaly", "anogan", "mygan"])
def test_single_rank_rccl_matches_plain(which, mode, dev, tmp_path):
    """The RCCL calls themselves on the one GPU this box has: a ONE-rank process group over backend "nccl"
    (VFD_DIST_SINGLE=1) with every broadcast, bucket all-reduce (asynchronous, from autograd hooks in eager mode, between the
    replayed phase graphs in graph mode, next to the filter-gradient side stream) and barrier issued for real; an
    all-reduce over one rank is the identity, so the step must equal the plain single-process run."""
    plain = _run(mode, 1, tmp_path, which)
    rccl = _run(mode, 1, tmp_path, which, rccl_single=True)
    for k, v in plain["errors"].items():
        assert abs(rccl["errors"][k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, rccl["errors"][k], v)
    for k, v in plain["sums"].items():
        assert abs(rccl["sums"][k] - v) <= 1e-4 * max(abs(v), 1.0), (k, rccl["sums"][k], v)
