#!/bin/bash
# The N > 1 bench line at FULL size on ONE GPU box (VERDICT r5 #1c): `python bench.py --gpus N --same-gpu` -- the script starts its N ranks itself, every rank a process
# on device 0, the library's RCCL leg over the stand-in transport tests/fake_rccl (RCCL refuses two ranks on a device) -- 10 M-triangle soup cut into N tiles, the
# replicated Image scheduler on the un-cut soup, config 4 at 1900x1080, N weak tiles of 10 M triangles each, per-variant parity against rank 0's one-rank render,
# the CPU baseline on rank 0.  Records the wall time against the driver's 600 s limit.  The pool allows 6 processes on the card at once: N <= 6 here.
#   bash tools/multiproc_fullsize.sh [N=4] [tag=r06] [extra bench.py arguments]
N=${1:-4}; TAG=${2:-r06}; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out
python3 -c "from tests.test_gpu_multiproc import build_fake_rccl; print(build_fake_rccl())" || exit 1
export GVT_HIP_RCCL_LIB=$REPO/tests/fake_rccl/libfakerccl.so TMPDIR=/tmp/mpf_$$ FAKE_RCCL_TIMEOUT_S=600 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p $TMPDIR
OUT=gpurun_out/${TAG}_multiproc_fullsize_$N
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus $N --same-gpu "$@" > $OUT.json 2> $OUT.err
rc=$?
t1=$(date +%s)
echo "bench.py --gpus $N --same-gpu $@: exit status $rc, wall $((t1 - t0)) s" | tee $OUT.txt
python3 - $OUT.json >> $OUT.txt <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("no JSON line"); sys.exit(0)
j = json.loads(lines[-1])
print("value %.1f Mrays/s, %.3f ms per step (%s); transport %s; rehearsal: %s" % (j["value"], j["ms_per_step"], j["config"]["scheduler"], j["config"].get("transport"), j["config"].get("rehearsal")))
for k, v in j.get("variants", {}).items():
    print("  %-34s %s" % (k, ("%.1f Mrays/s %.3f ms, %.1f ticks" % (v["value"], v["ms_per_step"], v["ticks_per_step"])) if "value" in v else v))
for k in ("domain_async", "domain_bsp"):
    v = j.get("config4_bunny_grid", {}).get(k)
    if v: print("  config4 %-26s %.1f Mrays/s %.3f ms, %.1f ticks" % (k, v["value"], v["ms_per_step"], v["ticks_per_step"]))
w = j.get("weak_soup")
if w: print("  weak_soup %d tiles x %d tris, film %s: %.1f Mrays/s %.3f ms, %.1f ticks" % (w["tiles"], w["tris_per_tile"], w["film"], w["value"], w["ms_per_step"], w["ticks_per_step"]))
print("parity:", json.dumps(j.get("parity"), indent=1))
cb = j.get("cpu_baseline", {})
print("cpu_baseline: %s Mrays/s on %s cores (%s), %s" % (cb.get("value"), cb.get("cores"), cb.get("which"), cb.get("note")))
print("roofline:", {k: j["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")} if "roofline" in j else None)
print("legs_error:", j.get("legs_error"))
PY
tail -5 $OUT.err >> $OUT.txt
rm -rf $TMPDIR
exit $rc
