#!/usr/bin/env python3
"""Run bench.py with the given arguments and print a compact one-line summary (development helper)."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print("bench failed:", out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
ps = d["detail"]["per_step"]
km = d["detail"]["kernel_ms_per_step"]
print("waits: front %.1f slot %.1f drain %.1f" % (ps.get("ms_wait_front", 0), ps.get("ms_wait_slot", 0), ps.get("ms_drain", 0)), end=" ")
print("%s | %.0f sf/s | %.2f ms/step | search %.1f (core %.1f rar %.1f) stageC %.1f commit %.1f | turbo %.2f vit %.2f demod %.2f | rm/map Gcyc %.2f/%.2f | cb %d it %d" % (
    os.environ.get("TAG", ""), d["value"], d["ms_per_step"], ps["ms_search"], ps["ms_search_core"], ps["ms_rar"], ps["ms_stage_c"], ps["ms_commit"],
    km["k_turbo<64>"] + km["k_turbo<128>"], km["k_viterbi"], km["k_pdsch_demod"], ps["turbo_cyc_rm"] / 1e9, ps["turbo_cyc_map"] / 1e9, ps["nof_cb_decodes"], ps.get("nof_turbo_iterations_run", ps["nof_turbo_iterations"])))
