# One short call for a last kernel change: parity subset, a one-round A/B of two builds, then the stamped PMC summary, a bench line
# that carries it, and the rocprofv3 kernel table, most important first:  tools/ab_then_profile.sh <tag> <old.so> <new.so> <family> "<pytest -k expr>"
ulimit -c 0
T=$1; A=$2; B=$3; F=$4; K=$5; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -s -k "$K" 2>&1 | grep -E "passed|failed|Error|^E |unet_out" | tail -6 > $O/pytest.log
cat $O/pytest.log
if ! grep -q " passed" $O/pytest.log || grep -q "failed" $O/pytest.log; then echo "PARITY NOT GREEN: stopping"; exit 1; fi
bash tools/ab_lib_family.sh $A $B "$F" 1 --steps 20 | tee $O/ab.txt
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o c -- $CMD > $O/pmc_bench.json 2> $O/$C.err
  f=$(find $O/$C -name "*counter_collection.csv" | head -1); cp $f $O/counters_$C.csv; rm -rf $O/$C
done
python tools/pmc_traffic.py $O/counters_FETCH_SIZE.csv $O/counters_WRITE_SIZE.csv $O/pmc_bench.json 5 $O/pmc_traffic.json > $O/pmc_hbm_traffic.txt
rm -f $O/counters_*.csv; cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; rm -rf $O/prof
python tools/prof_summary.py $O/kernel_stats.csv $O/bench_prof.json > $O/family_table.txt; head -12 $O/family_table.txt
