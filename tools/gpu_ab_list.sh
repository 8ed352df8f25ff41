# A/B of side-by-side library builds (MLRT_TAG=<tag> python mu-lambda-raytracer_amd/build.py -> lib/libmlrt_hip<tag>.so):
#   bash tools/gpu_ab_list.sh "<tags, '' = the default build>" "<workloads of tools/gpu_perf.py>" [rounds]
# prints one line per (library, workload, precision, round), sorted so that the rounds of a variant sit together
L=$GRAFT_REPO_ROOT/mu-lambda-raytracer_amd/lib
TAGS=${1:-base}; WHICH=${2:-c2}; ROUNDS=${3:-2}
for round in $(seq $ROUNDS); do for t in $TAGS; do [ "$t" = base ] && t=""; for prec in ${PRECS:-1 0}; do
  MLRT_PREC=$prec MLRT_LIB=$L/libmlrt_hip$t.so timeout -k 10 300 python tools/gpu_perf.py $WHICH 2>&1 | grep -v amdgpu | sed "s/^/prec$prec /" | cut -c1-150
done; done; done | sort
