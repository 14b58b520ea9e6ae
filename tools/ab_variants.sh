# A/B of libfd_hip.so variants inside ONE gpurun call (boxes differ by up to 20 %): usage  bash tools/ab_variants.sh "" old4 old5 ...
R=$PWD
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" != "cur" ]; then export FD_HIP_LIB=$R/featuredetection_amd/alt/libfd_hip_$v.so; else unset FD_HIP_LIB; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --also ffp15 2>&1 | tail -1 > gpurun_out/ab_$v.json
  python - <<PY
import json
r=json.loads(open("gpurun_out/ab_$v.json").read())
print("variant [$v]", round(r["value"],1), "kernel_ms", round(r["roofline"]["kernel_ms"],4), "| ffp15", round(r["also"][0]["value"],1), round(r["also"][0]["roofline"]["kernel_ms"],4))
PY
done
done
