#!/bin/bash
# Registers / spills / occupancy of step_apply_xr_tile_kernel<7,false,NDT>, NDT = 1..6, compiled alone (seconds instead of a minute).
cd "$(dirname "$0")/../mgpreconditionedgcr_amd/csrc" || exit 1
cat > /tmp/xr_tile_only.hip <<'EOT'
#include "internal.h"
#include "reduce.h"
#include "spmv_dev.h"
#include "gcr_dev.h"
namespace mgcr {
struct DotVecs { const cplx *v[FND]; };
#include "gcr_fused_xr_tile.h"
#define INST(N) template __global__ void step_apply_xr_tile_kernel<7, false, N, APCV>(RowMat, const cplx *, const cplx *, cplx *, cplx *, DotVecs, int64_t, int, RowMap, double *, double *, DevState *, int, const double *, int, int, cplx *, int, LeanCoef *);
INST(1) INST(2) INST(3) INST(4) INST(5) INST(6)
}
EOT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -I../../include -Rpass-analysis=kernel-resource-usage -DAPCV=${APC:-false} $EXTRA -c /tmp/xr_tile_only.hip -o /tmp/xr_tile_only.o 2> /tmp/xr_tile_only.log
python3 - <<'EOP'
import re
t = open('/tmp/xr_tile_only.log').read()
if 'error' in t: print(t[:3000])
for b in re.split(r'(?=remark: [^\n]*Function Name:)', t):
    m = re.search(r'Function Name: (\S+)', b)
    if not m or 'step_apply_xr_tile' not in m.group(1): continue
    nd = re.search(r'ELb0ELi(\d+)ELb', m.group(1)).group(1)
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print('NDT', nd, 'VGPR', g('VGPRs'), 'spill', g('VGPRs Spill'), 'scratch B/lane', g(r'ScratchSize \[bytes/lane\]'), 'waves/SIMD', g(r'Occupancy \[waves/SIMD\]'))
EOP
