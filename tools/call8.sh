cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c8
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/c8/pytest.log
cat gpurun_out/c8/pytest.log
bash tools/exp_round3.sh c8 p0:PWAF_PLACEMENT=0 p1:PWAF_PLACEMENT=1 p2:PWAF_PLACEMENT=2 p3:PWAF_PLACEMENT=3 > gpurun_out/c8/exp.log 2>&1
cat gpurun_out/c8/exp.log
BENCH_EXTRA=--adversarial bash tools/exp_round3.sh c8adv p0:PWAF_PLACEMENT=0 p1:PWAF_PLACEMENT=1 > gpurun_out/c8/exp_adv.log 2>&1
cat gpurun_out/c8/exp_adv.log
BENCH_EXTRA="--config 5" bash tools/exp_round3.sh c8c5 p0:PWAF_PLACEMENT=0 p1:PWAF_PLACEMENT=1 p2:PWAF_PLACEMENT=2 > gpurun_out/c8/exp_c5.log 2>&1
cat gpurun_out/c8/exp_c5.log
unset PWAF_LIB_VARIANT
timeout 900 python bench.py > gpurun_out/c8/bench.json 2> gpurun_out/c8/bench.err; tail -c 400 gpurun_out/c8/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c8/bench.json'))
print('HEADLINE', d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['traffic_modes'].get('adversarial_tuned_on_benign',{}).get('requests_per_s'), d.get('batcher'))
print('config5', {k:v for k,v in d['config5'].items() if k in ('requests_per_s','ms_per_step','frac','latency_ms','adversarial_over_benign')})
PY
