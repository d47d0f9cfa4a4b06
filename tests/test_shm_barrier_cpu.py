"""CPU-only: the host-side barrier of the peer-direct transport's bootstrap (cleora_amd/csrc/shm_barrier.h, used by csrc/peer.hip)
with as many PROCESSES as a node has GPUs and more.  The GPU box has one GPU and runs the transport with two and three ranks
(tests/test_gpu_sharded_abi.py); a scaling run is the first time eight ranks meet at this barrier, so it is stressed here:
every round each rank publishes the round number, passes the barrier and must see that number in every slot.  A rank that never
arrives must cost the budget and an error, not a hang.  No reference counterpart (pycleora is single-process)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stress(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("shm") / "shm_barrier_stress")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "csrc", "shm_barrier_stress.cpp"),
                           "-o", exe, "-lrt"])
    return exe


@pytest.mark.parametrize("world,rounds", [(2, 200_000), (3, 100_000), (8, 100_000), (16, 20_000), (64, 2_000)])
def test_every_rank_sees_every_round(stress, world, rounds):
    p = subprocess.run([stress, str(world), str(rounds)], capture_output=True, text=True, timeout=300)
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and out == {"world": world, "rounds": rounds, "disagreements": 0, "timeouts": 0, "crashed": 0}, (p.returncode, out)


def test_an_absent_rank_is_a_timeout_on_every_other_rank(stress):
    p = subprocess.run([stress, "8", "10", "5"], capture_output=True, text=True, timeout=60)
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert p.returncode == 0 and out["timeouts"] == 7 and out["crashed"] == 0, (p.returncode, out)
