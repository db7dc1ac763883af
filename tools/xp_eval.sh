mkdir -p gpurun_out/r2c
for L in 0 1 2 3 4; do
  echo "=== MVD_XP=$L" >> gpurun_out/r2c/xp.log
  MVD_XP=$L timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "unet_full_vs_golden or trained or step_full_width or smplx or small_persp or lat64" 2>&1 | grep -E "parity\] (unet_out|eps)|passed|failed" >> gpurun_out/r2c/xp.log
  MVD_XP=$L timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/s', round(d['value'],2), 'ms', round(d['ms_per_step'],3))" >> gpurun_out/r2c/xp.log
done
cat gpurun_out/r2c/xp.log
