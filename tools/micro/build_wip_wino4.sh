# Builds the work-in-progress F(4x4,3x3) kernel ALONE into tools/micro/_build/libwip_wino4.so (never part of libc2m_hip.so).
#   bash tools/micro/build_wip_wino4.sh [-DW4_ZERO_RECORDS]
set -e
cd "$(dirname "$0")"
mkdir -p _build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -fno-slp-vectorize -I ../../c2m_amd/csrc "$@" -shared -o _build/libwip_wino4.so wip_conv_wino4.hip
echo built _build/libwip_wino4.so
