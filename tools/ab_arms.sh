# A/B of arbitrary arms (environment assignments, e.g. a library path and switches) on ONE box, interleaved:
#   bash tools/ab_arms.sh <outfile> <rounds> "<ENV=.. ENV=..>" ...     headline bench only
O=$1; R=$2; shift 2
mkdir -p $(dirname $O); : > $O
for r in $(seq 1 $R); do
  for arm in "$@"; do
    env $arm timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline [$arm]', round(d['ms_per_step'],3), 'ms')" >> $O
  done
done
cat $O
