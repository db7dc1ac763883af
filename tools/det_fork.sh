# Side-stream determinism investigation: full width, N=16, 200 repetitions per variant.
O=gpurun_out/r2e; mkdir -p $O
export DET_FULL=1 DET_N=16 DET_WS=30 DET_REPS=200
run() { echo "--- $*" >> $O/det.log; env "$@" timeout 300 python tools/det_step.py 2>&1 | grep -E "mismatches|diff" | tail -4 >> $O/det.log; }
run MVD_DUMMY=1                                   # shipped: two-way handshake, comm overlap on
run MVD_ONE_WAY_FORK=1                            # the variant that was flaky in round 1
run MVD_ONE_WAY_FORK=1 MVD_NO_COMM_OVERLAP=1
run MVD_ONE_WAY_FORK=1 MVD_NO_CTX_FOLD=1
run MVD_ONE_WAY_FORK=1 MVD_XP=0
run MVD_ONE_WAY_FORK=1 DET_BVN=4
run MVD_ONE_WAY_FORK=1 MVD_NO_PARITY_BATCH=1
run MVD_ONE_WAY_FORK=1 MVD_NO_HALO=1
cat $O/det.log
