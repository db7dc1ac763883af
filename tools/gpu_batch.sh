# One script regenerates every number DESIGN §5 quotes:  gpurun --timeout 3000 -- 'bash tools/gpu_batch.sh <tag> [tests] [pmc] [train] [nocpu]'
# -> gpurun_out/<tag>/{pytest.log, smoke.log, bench.json, bench_sim8.json, kernel_stats.csv, family_table.txt, layers_step.txt,
#    pmc_mfma_busy.txt, pmc_hbm_traffic.txt, pmc_traffic.json, bench_train_b8.json, train_kernel_stats.txt}; copy what is to be
#    judged into profiles/<round>_<tag>_* (pmc_traffic.json -> profiles/pmc_traffic.json is what bench.py's roofline.traffic reads).
ulimit -c 0
T=${1:-batch}; shift; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
want() { case " $ARGS " in *" $1 "*) return 0;; esac; return 1; }
ARGS="$*"
if want tests; then
  (timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -400) > $O/pytest.log 2>&1
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
NC=""; if want nocpu; then NC="--no-cpu-baseline"; fi  # (the CPU-baseline leg is ~6 minutes of host time on the GPU box)
timeout 700 python bench.py --steps 20 --warmup 3 $NC > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --simulate-gpus 8 > $O/bench_sim8.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python tools/prof_summary.py $O/kernel_stats.csv $O/bench_prof.json $O/kernel_durations.json > $O/family_table.txt
MVD_LAYER_TIMING=1 timeout 300 python tools/layer_step.py 2> $O/layers.log > /dev/null; python tools/layer_agg.py $O/layers.log 40 > $O/layers_step.txt 2>/dev/null
if want pmc; then
  CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $CMD > $O/pmc_bench.json 2> $O/mfma.err
  f=$(find $O/mfma -name "*counter_collection.csv" | head -1); python tools/pmc_step.py $f > $O/pmc_mfma_busy.txt; rm -rf $O/mfma
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o c -- $CMD > /dev/null 2> $O/$C.err
    f=$(find $O/$C -name "*counter_collection.csv" | head -1); cp $f $O/counters_$C.csv; rm -rf $O/$C
  done
  python tools/pmc_traffic.py $O/counters_FETCH_SIZE.csv $O/counters_WRITE_SIZE.csv $O/pmc_bench.json 5 $O/pmc_traffic.json > $O/pmc_hbm_traffic.txt
  rm -f $O/counters_*.csv
fi
if want train; then
  timeout 600 python bench.py --config train --train-batch 8 --steps 5 --warmup 2 > $O/bench_train_b8.json 2> $O/train.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tprof -o t -- python bench.py --config train --train-batch 8 --steps 3 --warmup 1 > /dev/null 2> $O/tprof.err
  f=$(find $O/tprof -name "*kernel_stats.csv" | head -1); head -40 $f > $O/train_kernel_stats.txt; rm -rf $O/tprof
fi
tail -4 $O/pytest.log 2>/dev/null; tail -2 $O/smoke.log; head -c 600 $O/bench.json; echo; python tools/fam_table.py $O/bench.json | head -8; python tools/fam_table.py $O/bench_sim8.json | head -3; head -12 $O/family_table.txt
