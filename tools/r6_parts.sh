# round-6 (GPU box): resolve_kernel with one wave per slab against four (PWAF_RESOLVE_PARTS), GPU suite first, then the shares' step times and timelines
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
PWAF_RESOLVE_PARTS=4 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prefilter.py -m gpu -x -q > $O/tests_parts4.log 2>&1; echo "rc=$?" >> $O/tests_parts4.log; tail -2 $O/tests_parts4.log
for parts in 1 4; do
 for n in 1250000 2500000 10000000; do
  PWAF_RESOLVE_PARTS=$parts python bench.py --gpus 1 --steps 20 --warmup 5 --requests $n --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 > $O/b_${parts}_$n.json 2> $O/b_${parts}_$n.err
  python -c "
import json
d=json.load(open('$O/b_${parts}_$n.json')); print('parts $parts share $n', round(d['ms_per_step'],4), d['traffic_modes']['tuned_benign']['kernels_ms_per_step'])"
 done
done
for parts in 1 4; do
  PWAF_RESOLVE_PARTS=$parts BENCH_EXTRA="--no-config5 --requests 1250000" bash tools/timeline.sh > $O/tl_${parts}_1250000.txt 2>&1
done
tail -n 20 $O/tl_4_1250000.txt
