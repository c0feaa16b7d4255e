#!/bin/bash
# In-kernel phase stamps of the MARS tail kernels (csrc/mars_tail.hip built with -DDD_MARS_STAMPS into a private library; the product
# library is untouched).  Run from the repository root after `python -m deepdish_amd.build`:   bash scripts/experiments/mars_stamps.sh [crops]
set -e
N=${1:-7680}
OBJ=deepdish_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -mllvm -amdgpu-mfma-vgpr-form -DDD_MARS_STAMPS -x hip -c deepdish_amd/csrc/mars_tail.hip -o /tmp/mars_tail_stamps.o
OBJS=$(ls $OBJ/*.o | grep -v mars_tail.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/libdd_stamps.so $OBJS /tmp/mars_tail_stamps.o
DD_LIB=$PWD/gpurun_out/libdd_stamps.so python3 - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deepdish_amd import nets
from deepdish_amd.engine import Net
net = Net(nets.compile_mars(nets.synthetic_mars_weights()), max_batch=$N)
x = torch.randint(0, 256, ($N, 64, 32, 3), dtype=torch.uint8, device='cuda')
for r in range(3):
    net.forward(x); net.ctx.sync()
    print('---', file=sys.stderr)
PY
