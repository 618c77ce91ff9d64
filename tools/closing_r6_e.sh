# round-6 closing, part E (GPU box): what the driver runs, at HEAD — the GPU suite, smoke(), the bench command
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_final; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests_head.log 2>&1; echo "rc=$?" >> $O/gputests_head.log; grep -E "passed|failed|rc=" $O/gputests_head.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_head.json 2> $O/bench_head.err; tail -c 200 $O/bench_head.json
