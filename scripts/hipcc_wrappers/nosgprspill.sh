#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -amdgpu-spill-sgpr-to-vgpr=0
