#!/bin/bash
# interleaved A/B on one box (run from the repo root on the GPU box): the busy lists by kind (the Car's default) against one busy list,
# Car / push_box and Point / go_to_goal at 4 M, 2 M and 1 M envs; each line = one process (200 burn-in steps, 40 timed)
for rep in 1 2; do
for cfg in "push_box car 4194304" "go_to_goal point 4194304" "push_box car 2097152" "push_box car 1048576" "go_to_goal point 1048576"; do
  echo "$cfg kinds=1 $(SAG_BUSY_KINDS=1 python tools/busy_e_sweep.py one $cfg)"; echo "$cfg kinds=0 $(SAG_BUSY_KINDS=0 python tools/busy_e_sweep.py one $cfg)"
done; done
