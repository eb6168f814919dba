#!/usr/bin/env bash
# Builds a variant of libtunafock.so with extra compiler flags (A/B experiments on the GPU box: TUNAFOCK_LIB selects the library).
# usage: tools/build_variant.sh NAME "-DTF_SEG_PAD=16 ..."   ->  tuna_amd/libtunafock_NAME.so (git-ignored, travels with gpurun)
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/../tuna_amd/csrc"
NAME="$1"; shift
FLAGS="$*"
mkdir -p /tmp/tf_variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I/opt/rocm/include -Wno-unused-result --offload-arch=gfx950 $FLAGS -c tf_device.hip -o /tmp/tf_variants/tf_device_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I/opt/rocm/include -Wno-unused-result --offload-arch=gfx950 $FLAGS -c tf_eri_team.hip -o /tmp/tf_variants/tf_eri_team_$NAME.o
[ -f tf_host.o ] || make tf_host.o
[ -f tf_eri_teamc.o ] || make tf_eri_teamc.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/tf_variants/tf_device_$NAME.o /tmp/tf_variants/tf_eri_team_$NAME.o tf_eri_teamc.o tf_host.o -shared -L/opt/rocm/lib -lrocblas -lrocsolver -Wl,-rpath,/opt/rocm/lib -o ../libtunafock_$NAME.so
echo built tuna_amd/libtunafock_$NAME.so with: $FLAGS
