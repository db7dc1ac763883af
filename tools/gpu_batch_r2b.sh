# round-2 second profile set: full tests, bench, rocprof family table, sim8, layer timing
O=gpurun_out/r2B; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -260) > $O/pytest.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 700 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --simulate-gpus 8 > $O/bench_sim8.json 2>/dev/null
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python tools/prof_summary.py $O/kernel_stats.csv $O/bench_prof.json > $O/family_table.txt
MVD_LAYER_TIMING=1 timeout 300 python tools/layer_step.py 2> $O/layers.log > /dev/null; python tools/layer_agg.py $O/layers.log 40 > $O/layers_step.txt 2>/dev/null
tail -4 $O/pytest.log; tail -2 $O/smoke.log; head -c 600 $O/bench.json; echo; python tools/fam_table.py $O/bench.json | head -8; python tools/fam_table.py $O/bench_sim8.json | head -3; head -12 $O/family_table.txt
