# round-2 measurement batch: full GPU tests, bench (+CPU baseline), rocprof kernel stats, per-layer timing (1 GPU and simulated 8-way)
O=gpurun_out/r2d; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -200) > $O/pytest.log 2>&1
timeout 700 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r2d -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_prof.json 2> $O/prof.err
MVD_LAYER_TIMING=1 timeout 300 python tools/layer_step.py 2> $O/layers.log > /dev/null
MVD_LAYER_TIMING=1 timeout 300 python tools/layer_step.py --simulate-gpus 8 2> $O/layers_sim8.log > /dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --simulate-gpus 8 > $O/bench_sim8.json 2>/dev/null
tail -3 $O/pytest.log; head -c 1500 $O/bench.json; echo; find $O/prof -name "*stats*" | head
