#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -amdgpu-enable-rewrite-partial-reg-uses=0
