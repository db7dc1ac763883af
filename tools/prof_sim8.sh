# rocprofv3 kernel stats of the 2-views-per-rank step (one rank of an 8-GPU run simulated on one GPU): tools/prof_sim8.sh <tag>
O=gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --simulate-gpus 8 > $O/bench_sim8_prof.json 2> $O/prof8.err
f=$(find $O/prof8 -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_sim8.csv; rm -rf $O/prof8
python tools/prof_summary.py $O/kernel_stats_sim8.csv $O/bench_sim8_prof.json > $O/family_table_sim8.txt
head -30 $O/family_table_sim8.txt
