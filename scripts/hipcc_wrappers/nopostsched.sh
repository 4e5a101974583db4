#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -enable-post-misched=0
