# Timing experiments of round 2 (needs the -DPWAF_PROFILING build of libpwaf.so: env switches below change results).
# usage on the GPU box: bash tools/exp_round2.sh <tag> [runs...]   (run = name:ENV=V,ENV=V)
TAG=${1:-exp}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 6 --warmup 2 --verbose --no-cpu-baseline --no-extra-modes --no-pcie $BENCH_EXTRA > $OUT/$name.json 2> $OUT/$name.err
  echo "== $name ($*)"; grep -E "avg" $OUT/$name.err | sed 's/^  //;s/  */ /g' | tr '\n' ';'; echo
  python -c "import json;d=json.load(open('$OUT/$name.json'));print('   ms/step',round(d['ms_per_step'],3),'Greq/s',round(d['value']/1e9,3),'roofline',d['roofline']['kernel'],round(d['roofline']['frac'],4))"
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  run $name $(echo $envs | tr ',' ' ')
done
