# A/B of kernel-library variants for wgrad on ONE device: tools/ab_wgrad.sh default variant ...
set -e
SHAPES=("40 256 32 64 256 3 1 1 zeros" "40 512 16 32 512 3 1 1 zeros" "40 256 16 32 256 3 1 1 reflect" "40 128 64 128 128 3 1 1 reflect" "40 64 128 256 64 3 1 1 zeros" "40 64 64 128 64 3 1 1 reflect" "40 128 64 128 64 3 1 1 reflect" "40 3 128 256 64 3 1 1 zeros" "8 64 64 128 64 4 2 1 reflect" "40 32 128 256 32 3 1 1 reflect")
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset C2M_AMD_LIB; else export C2M_AMD_LIB=$PWD/c2m_amd/lib/libc2m_hip_$v.so; fi
  for s in "${SHAPES[@]}"; do
    echo "[$v r$round] $s :: $(python tools/conv_microbench.py $s ${ITERS:-8} all 2>/dev/null | grep wgrad | sed 's/.*) //')"
  done
done
done
