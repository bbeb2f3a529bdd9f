# kernel times (rocprof) of the train step under knock-out builds of the weight-gradient kernel
root=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  export ABNET3_HIP_LIB=$root/tools/variants/lib_$v.so
  echo "== $v"
  PROF_ROWS=3 bash tools/prof.sh w_$v tools/step_prof.py | grep wgrad | cut -c1-150
done
