# Round-6 closing measurements (GPU box, repo root):  bash tools/closing_r6.sh A|B
#   A: the GPU test suite, the driver's bench command, the full profile of config 3 (kernel stats, HBM traffic, issue / LDS counters)
#      and the L2 -> fabric request-size counters of the same command (calibration of FETCH_SIZE for scattered gathers: ipres)
#   B: light profiles (kernel stats + HBM traffic) of config 3 hostile, config 5, 8 residual rules; the strong-scaling shares on one GPU
# Before the gpurun call:  git log -1 --format=%h -- pingoo_amd/csrc > .commit_id
R=$GRAFT_REPO_ROOT; cd $R
case $1 in
A)
  O=$R/gpurun_out/r6_final; mkdir -p $O
  python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; tail -2 $O/gputests.log
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
  bash tools/profile_round.sh r6_c3
  # which request sizes the L2 sends to the fabric, per kernel (FETCH_SIZE = requests x 64 B whatever their size)
  cd /tmp; export TMPDIR=/tmp
  NAMES=$(rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*_sum" | sort -u | head -4 | tr '\n' ' ')
  echo "request counters: $NAMES" > $R/gpurun_out/r6_c3/rdreq.log
  if [ -n "$NAMES" ]; then
    rocprofv3 --kernel-trace --pmc $NAMES -d $R/gpurun_out/r6_c3/pmc_RDREQ -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 >> $R/gpurun_out/r6_c3/rdreq.log 2>&1
    python $R/tools/pmc_rdreq.py $R/gpurun_out/r6_c3/pmc_RDREQ > $R/gpurun_out/r6_c3/rdreq.txt 2>> $R/gpurun_out/r6_c3/rdreq.log
    rm -rf $R/gpurun_out/r6_c3/pmc_RDREQ
    cat $R/gpurun_out/r6_c3/rdreq.txt
  fi
  ;;
B)
  PROFILE_LIGHT=1 bash tools/profile_round.sh r6_c3_adv --adversarial
  PROFILE_LIGHT=1 bash tools/profile_round.sh r6_c5 --config 5
  PROFILE_LIGHT=1 bash tools/profile_round.sh r6_res8 --residual 8
  O=$R/gpurun_out/r6_strong; mkdir -p $O
  for n in 10000000 5000000 2500000 1250000; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --requests $n --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 > $O/strong_$n.json 2> $O/strong_$n.err
  done
  ;;
esac
