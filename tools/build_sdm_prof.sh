#!/bin/bash
# -DFD_SDM_PROF build of libfd_hip.so for tools/sdm_phases.py -> featuredetection_amd/alt/libfd_hip_sdmprof.so
cd "$(dirname "$0")/../featuredetection_amd/csrc" && make >/dev/null && mkdir -p ../alt && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w -DFD_SDM_PROF -c sdm.hip -o /tmp/sdm_prof.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../alt/libfd_hip_sdmprof.so ctx.o pyramid.o wvm.o svm.o hog.o whi.o rvm.o fhog.o /tmp/sdm_prof.o dist.o hostalgo.o -ldl && echo built
