#!/bin/bash
# A/B builds of libfd_hip.so: tools/build_variant.sh <name> <-D flags for wvm.hip ...> -> featuredetection_amd/alt/libfd_hip_<name>.so
# (the other objects come from the regular build; select a variant with FD_HIP_LIB=<path>)
N=$1; shift
cd "$(dirname "$0")/../featuredetection_amd/csrc" && mkdir -p ../alt && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w "$@" -c wvm.hip -o /tmp/wvm_$N.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../alt/libfd_hip_$N.so ctx.o pyramid.o /tmp/wvm_$N.o svm.o hog.o whi.o rvm.o fhog.o sdm.o dist.o hostalgo.o -ldl && echo "built $N"
