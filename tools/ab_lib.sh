# A/B timing of two builds of the same ABI: tools/ab_lib.sh <libA.so> <libB.so> [extra bench args]
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    MVD_LIB_PATH=$PWD/$L timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms')"
  done
done
