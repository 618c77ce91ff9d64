# round-6 (GPU box): the GPU suite and the driver's bench command again (after a test / bench.py change that left pingoo_amd/csrc untouched)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc=$?" >> $O/gputests.log; grep -E "passed|failed" $O/gputests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
