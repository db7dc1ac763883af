O=gpurun_out/r2j; mkdir -p $O
export DET_FULL=1 DET_N=16 DET_WS=40 DET_REPS=10 MVD_DEBUG_SUM=1 MVD_ONE_WAY_FORK=1
timeout 300 python tools/det_step.py > $O/a.out 2> $O/a.err; grep "\[gath\]" $O/a.err | cut -c1-700 | head -12
