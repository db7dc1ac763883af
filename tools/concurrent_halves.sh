# Experiment: two processes, each running the 8-views-per-rank step (half of the N = 16 batch), concurrently on one GPU --
# does de-phasing two half-batch chains beat one full-batch chain?  tools/concurrent_halves.sh <tag>
O=gpurun_out/$1; mkdir -p $O
A="--steps 40 --warmup 5 --no-cpu-baseline --no-extras"
python bench.py $A > $O/full.json 2>/dev/null
python bench.py $A --simulate-gpus 2 > $O/half_alone.json 2>/dev/null
python bench.py $A --simulate-gpus 2 > $O/half_a.json 2>/dev/null &
P1=$!
python bench.py $A --simulate-gpus 2 > $O/half_b.json 2>/dev/null &
P2=$!
wait $P1 $P2
for f in full half_alone half_a half_b; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', round(d['ms_per_step'],3),'ms')"; done
