O=gpurun_out/r2e; mkdir -p $O
bash tools/det_fork.sh > /dev/null 2>&1
(timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_vae.py tests/test_gpu_errors.py "tests/test_gpu_model.py::test_full_width_config_variants_properties" -q -s 2>&1 | grep -E "parity|property|precision|passed|failed|Error|error|assert|FAILED|^E " | tail -60) > $O/pytest.log 2>&1
for c in n8 smplx32; do timeout 600 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r2e -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof.err
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -o r2e -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_mfma.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o r2e -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o r2e -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_write.err
cat $O/det.log; tail -5 $O/pytest.log; find $O -name "*.csv" | head -20; du -sh $O
