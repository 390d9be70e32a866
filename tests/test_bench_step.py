"""bench.py --step-sf 0 (the default): the step is chosen so that the whole run lies inside the cached oracle stream - the headline is never printed ungated
because the cache is shorter than the run (round 5: the 500 000-subframe cache of the driver's command was lost with a container; the line then showed pcap_diff null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_auto_step_keeps_the_run_inside_the_cached_oracle_stream():
    import bench
    f = bench.auto_step_sf
    assert f(20000, 20, 5, 500000) == 20000          # the driver's command on the full cache: 25 steps of one capture pass
    assert f(20000, 20, 5, 100000) == 4000           # ... on the round-4 cache: rounds 1-4's step
    assert f(20000, 5, 1, 100000) == 10000           # the default command
    assert f(20000, 3, 1, 80000) == 20000            # the 16 dB workload: one cold + three timed passes
    assert f(20000, 20, 5, 499999) == 10000
    assert f(20000, 20, 5, 3000) == 200 and f(20000, 20, 5, 0) == 200
    for cached in (5000, 12345, 100000, 250000, 500000, 10**7):
        for steps, warm in ((20, 5), (5, 1), (1, 0), (50, 10)):
            s = f(20000, steps, warm, cached)
            assert 20000 % s == 0 and s % 200 == 0
            assert (steps + warm) * s <= cached or s == 200


def test_committed_cache_covers_the_drivers_command():
    """the driver runs `bench.py --gpus 1 --steps 20 --warmup 5`: with the committed cache the step must be a whole pass of the capture (timed region 2 s)"""
    import bench
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg3_stream_oracle.json")))
    assert g["oracle_subframes"] == 200 * len(g["blocks"])
    s = bench.auto_step_sf(20000, 20, 5, g["oracle_subframes"])
    assert s * 25 <= g["oracle_subframes"]
    assert s == (20000 if g["oracle_subframes"] >= 500000 else s)
