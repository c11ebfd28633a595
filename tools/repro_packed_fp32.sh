#!/bin/bash
# tools/repro_packed_fp32.sh [launches] [soak rounds]   (GPU box, from the repo root)  -> gpurun_out/r06_packed_fp32_repro.txt
# The round-5 hazard: compiler-formed packed fp32 next to another kernel's bf16-MFMA waves.
#  (1) the minimal attempt: tools/ubench/pk_fma_beside_bf16_mfma.hip (built here: tools/_abl/pk_fma);
#  (2) in the product: bit-identity soak of four bf16 bags in flight (tools/soak_merged.py) with the shipped library (packed fp32
#      off everywhere) and with tools/_abl/librrt_pkstream.so = the same sources with packed fp32 ON in the streaming units
#      (crmsa.hip, ln_partition.hip, cast16.hip: `ABL_PACKED=crmsa.hip,ln_partition.hip,cast16.hip tools/build_ablation.sh pkstream`).
R=$PWD; OUT=$R/gpurun_out/r06_packed_fp32_repro.txt; mkdir -p $R/gpurun_out
{
  echo "== (1) tools/ubench/pk_fma_beside_bf16_mfma.hip"
  timeout 300 $R/tools/_abl/pk_fma ${1:-2000}
  for lib in rrt-mil_amd/librrt_hip.so tools/_abl/librrt_pkstream.so; do
    for dt in bf16 f16; do
      echo "== (2) soak, $lib, $dt, four bags in flight"
      RRT_HIP_LIB=$R/$lib SOAK_DTYPE=$dt timeout 300 python $R/tools/soak_merged.py ${2:-60} 4 2>&1 | grep -v amdgpu.ids | tail -2
      echo "== (2b) the same with stream 0 running exact-fp32 one-bag-in-flight forwards (crmsa_combine_parts_kernel) beside the $dt bags"
      RRT_HIP_LIB=$R/$lib SOAK_DTYPE=$dt SOAK_MIX=1 timeout 300 python $R/tools/soak_merged.py ${2:-60} 4 2>&1 | grep -v amdgpu.ids | tail -2
    done
  done
} > $OUT 2>&1
cat $OUT
