# A/B of two builds of the same ABI with one family's time beside the step: tools/ab_lib_family.sh <libA.so> <libB.so> <family> [rounds] [extra bench args]
A=$1; B=$2; F=$3; R=${4:-3}; shift 4
for i in $(seq 1 $R); do
  for L in $A $B; do
    MVD_LIB_PATH=$PWD/$L timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f=[x for x in d['families'] if x['family']=='$F']
print('$L', round(d['ms_per_step'],3), 'ms/step;', '$F', round(f[0]['ms_per_step'],4) if f else None, 'ms/step in', f[0]['launches_per_step'] if f else None, 'launches')"
  done
done
