# Same-box A/B of the timed frame: the tree of an older commit (copied with its built library to gpurun_ab_old/, not tracked) against this
# tree in several ORB launch modes, no profiling events in either, alternating.  profiles/r06_orb_launch_modes.log
cd $GRAFT_REPO_ROOT
export SIVO_BENCH_NO_EVENTS=1
run() {  # label, dir, env..
  local label=$1 dir=$2; shift 2
  (cd $dir && env "$@" timeout 300 python bench.py --configs none --no-cpu-baseline --steps 60 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', d['value'], 'fps', d['ms_per_step'], 'ms; serial', d['config']['serial_fps'])"
}
for i in 1 2 3 4; do
  [ -d gpurun_ab_old ] && run "round-5 ORB (commit fad8803)" gpurun_ab_old X=1
  for m in 0 14 15; do run "this tree, ORB launch mode $m" . SIVO_BENCH_ORB_MODE=$m; done
done
