# PMC passes over a short bench run (separate passes, kernel-trace only beside --pmc): MFMA-busy and HBM-side traffic
O=gpurun_out/pmcB; mkdir -p $O; export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o m -- $CMD > $O/mfma.json 2> $O/mfma.err
f=$(find $O/mfma -name "*counter_collection.csv" | head -1); python tools/pmc_step.py $f > $O/pmc_mfma_busy.txt; rm -rf $O/mfma
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o c -- $CMD > $O/$C.json 2> $O/$C.err
  f=$(find $O/$C -name "*counter_collection.csv" | head -1); cp $f $O/counters_$C.csv; rm -rf $O/$C
done
head -14 $O/pmc_mfma_busy.txt; ls -la $O
