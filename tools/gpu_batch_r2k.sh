O=gpurun_out/r2k; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_vae.py tests/test_gpu_ops.py "tests/test_gpu_model.py::test_full_width_config_variants_properties" "tests/test_gpu_model.py::test_step_full_width_n16_vs_golden" -q -s 2>&1 | grep -E "parity\] (grad|decoded|eps)|property|passed|failed|Error|error|assert|FAILED|^E " | tail -50) > $O/pytest.log 2>&1
for i in 1 2; do
  for w in 0 -1; do
    echo "MVD_WMAJOR=$w" >> $O/ab.log
    MVD_WMAJOR=$w timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  N=16      ', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms')" >> $O/ab.log
    MVD_WMAJOR=$w timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --simulate-gpus 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  sim8 rank ', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms')" >> $O/ab.log
  done
done
MVD_LAYER_TIMING=1 timeout 300 python tools/layer_step.py 2> $O/layers.log > /dev/null
tail -32 $O/pytest.log; cat $O/ab.log
