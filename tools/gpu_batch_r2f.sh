O=gpurun_out/r2f; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_vae.py "tests/test_gpu_model.py::test_full_width_config_variants_properties" -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -60) > $O/pytest.log 2>&1
export DET_FULL=1 DET_N=16 DET_WS=30 DET_REPS=12 MVD_DEBUG_SUM=1
MVD_ONE_WAY_FORK=1 timeout 300 python tools/det_step.py > $O/sum_oneway.out 2> $O/sum_oneway.err
timeout 300 python tools/det_step.py > $O/sum_twoway.out 2> $O/sum_twoway.err
MVD_ONE_WAY_FORK=1 MVD_NO_HALO=1 timeout 300 python tools/det_step.py > $O/sum_oneway_nohalo.out 2> $O/sum_oneway_nohalo.err
tail -30 $O/pytest.log; tail -2 $O/sum_oneway.out
