#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -amdgpu-waitcnt-forcezero
