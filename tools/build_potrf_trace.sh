#!/bin/bash
# builds the phase-stamp harness of potrf_diag_kernel into tools/_bin/ (git-ignored; travels with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
C=slam_plus_plus_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DSPP_POTRF_TRACE $SPP_EXTRA_DEFS -DSPP_HAVE_SPARSE -DSPP_HAVE_ASSEMBLE -I $C -I include \
	-o tools/_bin/potrf_trace tools/potrf_trace.hip $C/spp_api.cpp $C/spp_symbolic.cpp $C/spp_schur.hip $C/spp_sparse.hip $C/spp_assemble.hip $C/spp_geometry.hip
