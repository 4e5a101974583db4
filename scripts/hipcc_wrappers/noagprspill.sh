#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -amdgpu-spill-vgpr-to-agpr=0
