O=gpurun_out/r2g; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_train.py -q -s 2>&1 | grep -E "parity|passed|failed|Error|error|assert|FAILED|^E " | tail -40) > $O/pytest.log 2>&1
export DET_FULL=1 DET_N=16 DET_WS=30 DET_REPS=16 MVD_DEBUG_SUM=1
MVD_ONE_WAY_FORK=1 timeout 300 python tools/det_step.py > $O/sum_oneway.out 2> $O/sum_oneway.err
MVD_ONE_WAY_FORK=1 MVD_NO_COMM_OVERLAP=1 timeout 300 python tools/det_step.py > $O/sum_oneway_nocomm.out 2> $O/sum_oneway_nocomm.err
tail -25 $O/pytest.log
