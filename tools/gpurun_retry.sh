# usage (in the build container): bash tools/gpurun_retry.sh <timeout_s> '<command>'  — gpurun, retried while every GPU slot of the pod is busy (exit code 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  if [ $rc -ne 3 ] && ! grep -q '"status": "transient"' /root/repo/gpurun_out/.last_call.json 2>/dev/null; then exit $rc; fi
  sleep 45
done
exit 3
