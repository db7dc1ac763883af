O=gpurun_out/r2h; mkdir -p $O
export DET_FULL=1 DET_N=16 DET_WS=40 DET_REPS=24 MVD_DEBUG_SUM=1 MVD_ONE_WAY_FORK=1
run() { echo "--- $*" >> $O/det.log; env "$@" timeout 300 python tools/det_step.py > $O/out.tmp 2> $O/err.tmp; grep mismatches $O/out.tmp >> $O/det.log;
        python - >> $O/det.log <<'PY'
import collections
rows=[dict(kv.split("=") for kv in l.split()[1:]) for l in open("gpurun_out/r2h/err.tmp") if l.startswith("[sum]")]
print("   distinct:", {k: len(set(r[k] for r in rows)) for k in ("volume","gath","film","x3","src0","eps")})
PY
}
run X=1
run MVD_FORK_SYNC=1
run MVD_FORK_PAD_MB=2048
run MVD_POSTFORK_PAD_MB=2048
run MVD_MAIN_SPIN=500000
run MVD_SIDE_SPIN=500000
run MVD_FORK_SYNC=1 MVD_POSTFORK_PAD_MB=2048 MVD_FORK_PAD_MB=2048
cat $O/det.log
