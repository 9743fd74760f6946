"""The data-parallel step over RCCL with one fresh process per GPU (tests/rccl_worker.py).

Collected FIRST on purpose (file name): the ranks are child processes, and a process may only
start programs while it has not initialised the GPU itself -- so these tests neither use the
`cuda` fixture nor call anything that touches the device, and skip when an earlier test did.
  * two ranks: needs >= 2 visible GPUs (skipped on the 1-GPU boxes);
  * one rank: the same worker with WORLD_SIZE=1 (process group, FlatGradParallel, RCCL
    all-reduce of one rank), so that the worker itself is exercised wherever a GPU exists.
reference recipe: detection/GroupFree3D/train_GF_FSB.py:172-190, :250, :450-474."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, mode=None):
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU: it must not start programs")
    n = torch.cuda.device_count()          # (counting devices does not initialise them)
    if n < world:
        pytest.skip("%d GPU(s) visible, %d needed" % (n, world))
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BTR_DP="flat",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")] +
                                      ([mode] if mode else []),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    results = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, err[-3000:]
        results.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    return sorted(results, key=lambda r: r["rank"])


def _check(results, world):
    assert [r["rank"] for r in results] == list(range(world))
    for r in results:
        assert r["rccl_ranks"] == world
        assert r["replicas_identical"] and r["finite"] and r["buffers_per_replica"]
        # the averaged gradient against the hand-averaged local ones (two evaluations of a
        # float32 backward whose scatter order is not fixed: rounding level)
        assert r["grad_rel_err"] < 1e-3, r
    if world > 1:
        assert results[0]["losses"] != results[1]["losses"]   # different shards


def test_two_rank_rccl_pipelined_steps():
    _check(_launch(2), 2)


def test_one_rank_rccl_worker():
    _check(_launch(1), 1)


@pytest.mark.parametrize("mode", ["br", "gf"])
@pytest.mark.parametrize("world", [2, 1])
def test_rccl_back_to_reality_and_groupfree_steps(world, mode):
    """The two-forward Back-to-Reality step and the GroupFree3D step under FlatGradParallel over
    RCCL (train_GF_BR.py:330-331, 356: two forwards per backward with broadcast_buffers=False;
    train_GF_FSB.py:250, 316-319): replicas bit-identical after three pipelined steps.  Two ranks
    need two GPUs; the one-rank variant runs wherever a GPU exists."""
    results = _launch(world, mode)
    assert [r["rank"] for r in results] == list(range(world))
    for r in results:
        assert r["mode"] == mode and r["rccl_ranks"] == world
        assert r["replicas_identical"] and r["finite"], r
