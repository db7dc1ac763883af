# training-step profile set:  gpurun -- 'bash tools/train_prof.sh <tag> [B...]'  -> gpurun_out/<tag>/
ulimit -c 0
T=${1:-train}; shift; O=gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp
for B in ${@:-8}; do
  timeout 600 python bench.py --config train --train-batch $B --steps 5 --warmup 2 > $O/bench_train_b$B.json 2> $O/train_b$B.err
  timeout 600 python tools/train_phases.py $B > $O/phases_b$B.txt 2> $O/phases_b$B.err
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tprof -o t -- python bench.py --config train --train-batch 8 --steps 4 --warmup 2 > /dev/null 2> $O/tprof.err
f=$(find $O/tprof -name "*kernel_stats.csv" | head -1); head -60 $f > $O/train_kernel_stats.txt; rm -rf $O/tprof
for B in ${@:-8}; do head -c 300 $O/bench_train_b$B.json; echo; tail -9 $O/phases_b$B.txt; done
