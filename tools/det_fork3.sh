O=gpurun_out/r2i; mkdir -p $O
export DET_FULL=1 DET_N=16 DET_WS=40 DET_REPS=3 MVD_DEBUG_VOLUME=1
timeout 300 python tools/det_step.py > $O/a.out 2> $O/a.err; grep "\[volume\]" $O/a.err | head -8
MVD_NO_HALO=1 timeout 300 python tools/det_step.py > $O/b.out 2> $O/b.err; echo NO_HALO; grep "\[volume\]" $O/b.err | head -4
MVD_NO_SIDE_STREAM=1 timeout 300 python tools/det_step.py > $O/c.out 2> $O/c.err; echo NO_SIDE; grep "\[volume\]" $O/c.err | head -4
