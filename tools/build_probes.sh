#!/bin/bash
# builds the standalone probes under tools/probes (run from the repo root; binaries are git-ignored but travel with gpurun)
set -e
F="-w --offload-arch=gfx950 -O3 -std=c++17"
hipcc $F -DTSA_STAMPS tools/probes/tsa_probe.hip svd_xtend_amd/csrc/common.cpp -o tools/probes/tsa_probe
hipcc $F -DTSA_STAMPS -DTSA_SKIP_QKV_STORE tools/probes/tsa_probe.hip svd_xtend_amd/csrc/common.cpp -o tools/probes/tsa_probe_nostore
hipcc $F tools/probes/ingest_probe.hip -o tools/probes/ingest_probe
hipcc $F tools/probes/regw_probe.hip -Lsvd_xtend_amd/csrc -lsvdx -Wl,-rpath,'$ORIGIN/../../svd_xtend_amd/csrc' -o tools/probes/regw_probe
hipcc $F tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe
hipcc $F tools/probes/band_probe.hip -Lsvd_xtend_amd/csrc -lsvdx -Wl,-rpath,'$ORIGIN/../../svd_xtend_amd/csrc' -o tools/probes/band_probe
