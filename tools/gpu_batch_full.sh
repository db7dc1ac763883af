# full verification: every GPU test, smoke(), the bench line with the CPU baseline
O=gpurun_out/${1:-full}; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -260) > $O/pytest.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 700 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -4 $O/pytest.log; tail -2 $O/smoke.log; head -c 1200 $O/bench.json; echo; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['cpu_baseline']); print(d['ddim50_wall_s'], d['vae_decode_ms_local_views'])"
