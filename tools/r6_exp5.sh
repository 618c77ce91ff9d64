# round-6 experiment (GPU box): list-scan items with a common entries-per-item floor — suite, then step times at 10M / 1.25M, hostile, alone table
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r6j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log
A="--steps 20 --warmup 5 --no-cpu-baseline --no-extra-modes --no-pcie --no-config5"
for n in 10000000 1250000; do
  python bench.py $A --requests $n > $O/b_$n.json 2> $O/b_$n.err
  python -c "
import json
d=json.load(open('$O/b_$n.json')); print('share $n', round(d['ms_per_step'],4), d['traffic_modes']['tuned_benign']['kernels_ms_per_step'])"
done
python bench.py $A --adversarial > $O/b_adv.json 2> $O/b_adv.err
python -c "
import json
d=json.load(open('$O/b_adv.json')); print('adv', round(d['ms_per_step'],4), d['traffic_modes'])" | cut -c1-900
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pcie > $O/b_full.json 2> $O/b_full.err
python -c "
import json
d=json.load(open('$O/b_full.json')); tm=d['traffic_modes']
for k,v in tm.items(): print(k, round(v['requests_per_s']/1e9,3), round(v['ms_per_step'],3), v.get('kernels_ms_per_step'))
c=d['config5']; print('c5', c['requests_per_s'], c['adversarial_over_benign'], c['kernels_ms_per_step'], c['adversarial']['kernels_ms_per_step'])"
TAG=r6j bash tools/r6_alone.sh
