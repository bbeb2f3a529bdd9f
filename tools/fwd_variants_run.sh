# kernel time (rocprof) of the planes forward under knock-out builds:  bash tools/fwd_variants_run.sh base PEXP_NOW ...
root=${GRAFT_REPO_ROOT:-$(pwd)}
for prec in bf16x3 bf16; do
for v in "$@"; do
  export ABNET3_PRECISION=$prec ABNET3_HIP_LIB=$root/tools/variants/lib_$v.so
  echo "== $prec $v"
  PROF_ROWS=1 bash tools/prof.sh v_${prec}_$v tools/fwd_time.py | tail -1 | cut -c1-140
done; done
