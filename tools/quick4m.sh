for cfg in "push_box car 4194304" "go_to_goal point 4194304"; do
  echo "$cfg default $(python tools/busy_e_sweep.py one $cfg)"
  for q in 32 48 64 80 96 128; do echo "$cfg cu_split=$q $(SAG_CU_SPLIT=$q python tools/busy_e_sweep.py one $cfg)"; done
done
