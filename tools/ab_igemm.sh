# A/B of kernel-library variants on the gather-kernel shapes (stride-2 4x4 convs, 16x32-map reflect layers, 7x7 stem):
# tools/ab_igemm.sh default <variant> ...
set -e
SHAPES=("8 64 64 128 128 4 2 1 reflect" "40 128 32 64 256 4 2 1 reflect" "40 256 16 32 256 3 1 1 reflect" "40 32 128 256 64 4 2 1 reflect" "40 3 128 256 32 7 1 3 reflect" "8 21 128 256 32 4 2 1 reflect")
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = "default" ]; then unset C2M_AMD_LIB; else export C2M_AMD_LIB=$PWD/c2m_amd/lib/libc2m_hip_$v.so; fi
  for s in "${SHAPES[@]}"; do
    echo "[$v r$round] $s :: $(python tools/conv_microbench.py $s ${ITERS:-8} all 2>/dev/null | tr '\n' ' ' | cut -c1-600)"
  done
done
done
