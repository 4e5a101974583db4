#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -mllvm -enable-misched=0
