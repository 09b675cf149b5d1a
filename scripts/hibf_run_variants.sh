# usage: scripts/hibf_run_variants.sh "workload ..." "variant ..."   ("" = the product library; variants are built by scripts/hibf_variant.sh)
for w in ${1:-hibf64k_skew}; do
for v in "" $2; do
  if [ -z "$v" ]; then lib=""; else lib=$PWD/ganon_amd/csrc/_variants/libganon_hip_$v.so; fi
  echo "### $w ${v:-product}"
  GANON_HIP_LIB=$lib VARIANTS="" timeout 300 python scripts/hibf_probe.py $w 2> gpurun_out/var_$v.err
  grep "wave end" gpurun_out/var_$v.err | tail -3
done
done
