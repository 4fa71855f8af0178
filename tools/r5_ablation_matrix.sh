#!/bin/bash
# round 5 (gpurun): composite_bwd ablation matrix, the registers-instead-of-DPP variant, LDS-conflict attribution
# -> profiles/r5_bwd_ablation_matrix_ab.txt. Build the variants first, in the authoring container (they travel with the snapshot):
#   P=tools/variants/bwd_lab_r5.patch
#   tools/build_variant.sh base "" ; tools/build_variant.sh noexp "-DSFGS_BWD_ABLATE=32" $P ; tools/build_variant.sh noexprcp "-DSFGS_BWD_ABLATE=96" $P
#   tools/build_variant.sh nop1 "-DSFGS_BWD_ABLATE=2" $P ; tools/build_variant.sh nop2 "-DSFGS_BWD_ABLATE=4" $P ; tools/build_variant.sh nop1p2 "-DSFGS_BWD_ABLATE=6" $P
#   tools/build_variant.sh skel "-DSFGS_BWD_ABLATE=30" $P ; tools/build_variant.sh greg2 "-DSFGS_BWD_GREG=2" $P
#   tools/build_variant.sh greg4o12 "-DSFGS_BWD_GREG=4 -DSFGS_BWD_OCC=12" $P
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c1; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
# parity of the candidate first (bit-identical sums expected: fmaf == v_fmac)
SFGS_LIB=$PWD/$E/lib_greg2.so timeout 600 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k "not soak" > $O/tests_greg2.txt 2>&1; tail -3 $O/tests_greg2.txt
ROUNDS=2 tools/ab.sh $E/lib_base.so $E/lib_noexp.so $E/lib_noexprcp.so $E/lib_nop1.so $E/lib_nop2.so $E/lib_nop1p2.so $E/lib_skel.so $E/lib_greg2.so $E/lib_greg4o12.so -- --steps 60 --warmup 10 > $O/ab.txt 2>&1
cat $O/ab.txt
for v in base nop1 nop2; do
  SFGS_LIB=$PWD/$E/lib_$v.so tools/collect_profiles.sh r5c1_$v "sq2" > $O/sq2_$v.txt 2>&1
  cp gpurun_out/prof_r5c1_$v/sq.json $O/sq_$v.json 2>/dev/null
done
python - <<'PY'
import json
for v in ("base","nop1","nop2"):
    try:
        d=json.load(open(f"gpurun_out/r5c1/sq_{v}.json"))
        for k,x in d.items():
            if "composite_bwd" in k: print(v, {a:b for a,b in x.items() if a!="raw"}, {a:round(b) for a,b in x["raw"].items() if "LDS" in a})
    except Exception as e: print(v, "failed", e)
PY
