#!/usr/bin/env python3
"""Ablations of the token-stationary QKV projection (debug library): HIP-event time of a 256 x 512-token bf16 forward with
RR_CE_PROJ_EXP = 0 (as shipped), 1 (no output), 2 (no staging), 3 (neither), 7 (+ no barrier), 8 (output rows stored with sc1), 16 (staging loads without their LDS stores), 32 (LDS stores without the loads).
The difference to the shipped form / 6 launches = what that part costs per launch.  One child process per variant."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in (0, 16, 32, 2, 0):
    env = dict(os.environ, RR_DEBUG_HARNESS="1", RR_CE_PROJ_EXP=str(v))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k5_bench.py"), "--precision", "bf16", "--reps", "9"], env=env,
                       capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if "forward_ms" in l]
    print(f"RR_CE_PROJ_EXP={v}: ", line[-1][:120] if line else p.stderr[-300:])
