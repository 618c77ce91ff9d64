# round-6 closing, part D (GPU box): smoke(), then a longer GPU fuzz campaign at new seeds on the closing commit (default launch shapes; one wave per slab forced)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python tools/gpufuzz.py 800000 420 240 > $O/gpufuzz_d1.json 2> $O/gpufuzz_d1.err; cat $O/gpufuzz_d1.json
PWAF_RESOLVE_PARTS=1 python tools/gpufuzz.py 820000 240 0 > $O/gpufuzz_d2.json 2> $O/gpufuzz_d2.err; cat $O/gpufuzz_d2.json
