"""`bench.py --gpus 2` launches its own two ranks (python -m torch.distributed.run, one process per rank) and reports n_gpus = 2: the
multi-process path of BASELINE configs[4] (one cell per GPU, no data-path exchange) with the PRODUCT engine in every rank - two engines, two
HIP contexts and two sets of pinned threads on one box.  On a one-GPU box both ranks share the GPU (LSN_DIST_BACKEND=gloo: RCCL needs one
device per rank); the numbers are not a scaling result, the test checks the launch contract and that rank 0's stream still passes the parity
gate while another rank loads the same GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_spawns_two_product_ranks():
    env = dict(os.environ, LSN_DIST_BACKEND="gloo")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--step-sf", "800", "--nsf", "1600",
                          "--cpu-sample", "400", "--batch", "200", "--no-legs"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-1500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["cells"] == 2 and j["config"]["distinct_subframes"] == 1600
    assert j["value"] > 0
    # a 1 600-subframe capture is not the gated stream (no cached oracle blocks): the live oracle walks the first 400 subframes and rank 0's
    # cold-state blocks must equal its blocks; pcap_diff (which describes the timed region) stays null
    assert j["parity"]["warmup_equals_live_oracle_blocks"] is True and j["parity"]["oracle_subframes"] == 400 and j["pcap_diff"] is None
    assert j["cpu_baseline"]["value"] > 0
    # rank 1 replays its own cell: the head of its cold-state stream is checked against the oracle run live in ITS process, the verdict travels to rank 0
    assert j["parity"]["other_ranks_head_equals_live_oracle"] == [1]
    assert len(j["host"]["cores_busy_per_rank"]) == 2 and all(c > 0 for c in j["host"]["cores_busy_per_rank"])
