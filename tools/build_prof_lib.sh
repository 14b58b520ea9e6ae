#!/bin/bash
# -DFD_WVB_PROF build of libfd_hip.so for tools/wvb_phases.py (in-kernel timestamps of the stage-B kernels) -> featuredetection_amd/alt/
cd "$(dirname "$0")/../featuredetection_amd/csrc" && make >/dev/null && mkdir -p ../alt && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w -DFD_WVB_PROF -c wvm.hip -o /tmp/wvm_prof.o && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w -DFD_PYR_PROF -c pyramid.hip -o /tmp/pyramid_prof.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../alt/libfd_hip_wvbprof.so ctx.o /tmp/pyramid_prof.o /tmp/wvm_prof.o svm.o hog.o whi.o rvm.o fhog.o sdm.o dist.o hostalgo.o -ldl && echo built
