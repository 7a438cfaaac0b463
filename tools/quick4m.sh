for cfg in "push_box car 4194304" "push_box car 1048576" "go_to_goal point 4194304"; do
  echo "$cfg default $(python tools/busy_e_sweep.py one $cfg)"; echo "$cfg kinds=0 $(SAG_BUSY_KINDS=0 python tools/busy_e_sweep.py one $cfg)"
done
