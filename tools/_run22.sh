set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
E=$PWD/skyfall-gs_amd/sfgs/_exp
for r in 1 2; do for v in cc32 cc64 cc256 cc1024; do SFGS_LIB=$E/lib_$v.so timeout 200 python tools/diag_placement2.py tiles 14 2>&1 | tail -1; done; done | tee $O/placement3.log
