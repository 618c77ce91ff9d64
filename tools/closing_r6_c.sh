# round-6 closing, part C (GPU box): the driver's bench command once more, the timelines (10M, 1.25M), every kernel alone (profiling build), GPU fuzz at new seeds
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_final; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench2.json 2> $O/bench2.err; tail -c 200 $O/bench2.json
BENCH_EXTRA="--no-config5" bash tools/timeline.sh > $O/timeline_10M.txt 2>&1; cat $O/timeline_10M.txt
BENCH_EXTRA="--no-config5 --requests 1250000" bash tools/timeline.sh > $O/timeline_1250000.txt 2>&1; cat $O/timeline_1250000.txt
TAG=r6_final bash tools/r6_alone.sh
python tools/gpufuzz.py 700000 200 90 > $O/gpufuzz.json 2> $O/gpufuzz.err; cat $O/gpufuzz.json
PWAF_RESOLVE_PARTS=1 python tools/gpufuzz.py 710000 120 0 > $O/gpufuzz_parts1.json 2> $O/gpufuzz_parts1.err; cat $O/gpufuzz_parts1.json
