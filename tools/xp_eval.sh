mkdir -p gpurun_out/r06_k
for L in 2 3 4; do
  echo "=== MVD_XP=$L" >> gpurun_out/r06_k/xp_levels.txt
  MVD_XP=$L python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> gpurun_out/r06_k/xp_levels.txt
  MVD_XP=$L timeout 600 python -m pytest tests/test_gpu_model.py -q -s -k "step_full_width_n16 or full_width_config_variants or stages_and_step_small" 2>&1 | grep -E "\] eps|passed|failed" >> gpurun_out/r06_k/xp_levels.txt
  MVD_XP=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline ms', round(d['ms_per_step'],3))" >> gpurun_out/r06_k/xp_levels.txt
  MVD_XP=$L python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline ms', round(d['ms_per_step'],3))" >> gpurun_out/r06_k/xp_levels.txt
done
cat gpurun_out/r06_k/xp_levels.txt
