cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_quick -o q -- python3 $GRAFT_REPO_ROOT/tools/busy_e_sweep.py one go_to_goal point 4194304 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/prof_quick/*/*kernel_stats.csv gpurun_out/prof_quick/*kernel_stats.csv 2>/dev/null | head -1); grep -E "k_compact|k_step_busy|k_step_quiet" $f
