# A/B of environment switches on ONE box: bash tools/ab_env.sh <outfile> <rounds> "<ENV=1 ...>" "<ENV=...>" ...   (use X=1 for the default arm)
# each arm runs the headline bench and the 2-views-per-rank bench per round, arms interleaved
O=$1; R=$2; shift 2
mkdir -p $(dirname $O); : > $O
for r in $(seq 1 $R); do
  for arm in "$@"; do
    env $arm timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline [$arm]', round(d['ms_per_step'],3), 'ms', sum(f['launches_per_step'] for f in d['families']), 'launches')" >> $O
    env $arm timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --simulate-gpus 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sim8     [$arm]', round(d['ms_per_step'],3), 'ms', sum(f['launches_per_step'] for f in d['families']), 'launches')" >> $O
  done
done
sort $O | awk '{k=$1" "$2" "$3; s[k]+=$(NF-3); n[k]++} END {for (k in s) print k, s[k]/n[k], "ms mean of", n[k]}' | sort
