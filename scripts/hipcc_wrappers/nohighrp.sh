#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -amdgpu-disable-unclustered-high-rp-reschedule
