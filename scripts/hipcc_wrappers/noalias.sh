#!/bin/bash
exec /opt/rocm/bin/hipcc "$@" -fno-strict-aliasing
