# round-6 experiment (GPU box): chunk-driven resolve — suite, the resolve-heavy files with one wave per slab forced, fuzz with it forced, step times
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6m; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log
PWAF_RESOLVE_PARTS=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prefilter.py tests/test_gpu_paths.py -m gpu -x -q > $O/tests_parts1.log 2>&1; echo "rc=$?" >> $O/tests_parts1.log; grep -E "passed|failed|rc=" $O/tests_parts1.log
PWAF_RESOLVE_PARTS=1 python tools/gpufuzz.py 720000 90 0 > $O/gpufuzz_parts1.json 2> $O/gpufuzz_parts1.err; cut -c1-200 $O/gpufuzz_parts1.json
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
for n in 10000000 2500000; do
  python bench.py $A --requests $n > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
d=json.load(open('$O/b_$n.json')); print('share $n', round(d['ms_per_step'],4), d['traffic_modes']['tuned_benign']['kernels_ms_per_step'])"
done
python bench.py $A --adversarial > $O/b_adv.json 2> $O/b_adv.err
python -c "
import json
d=json.load(open('$O/b_adv.json')); v=list(d['traffic_modes'].values())[0]; print('adv', round(d['ms_per_step'],4), v['kernels_ms_per_step'])"
python bench.py $A --config 5 > $O/b_c5.json 2> $O/b_c5.err
python -c "
import json
d=json.load(open('$O/b_c5.json')); v=list(d['traffic_modes'].values())[0]; print('c5', round(d['ms_per_step'],4), v['kernels_ms_per_step'])"
TAG=r6m bash tools/r6_alone.sh
