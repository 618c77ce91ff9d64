# Timing experiments of round 3 on the -DPWAF_PROFILING build (pingoo_amd/libpwaf_prof.so; env switches below change results).
# usage on the GPU box: bash tools/exp_round3.sh <tag> name:ENV=V,ENV=V ...
TAG=${1:-exp}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export PWAF_LIB_VARIANT=${PWAF_LIB_VARIANT:-prof}
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --verbose --no-cpu-baseline --no-extra-modes --no-pcie --no-config5 $BENCH_EXTRA > $OUT/$name.json 2> $OUT/$name.err
  echo "== $name ($*)"; grep -E "avg" $OUT/$name.err | sed 's/^  //;s/  */ /g' | tr '\n' ';'; echo
  python -c "import json;d=json.load(open('$OUT/$name.json'));print('   ms/step',round(d['ms_per_step'],3),'Greq/s',round(d['value']/1e9,3),'roofline',d['roofline']['kernel'],round(d['roofline']['frac'],4))" 2>/dev/null || tail -3 $OUT/$name.err
}
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  [ "$envs" = "$spec" ] && envs="X=1"
  run $name $(echo $envs | tr ',' ' ')
done
