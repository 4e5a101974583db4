#!/bin/bash
args=(); for a in "$@"; do [ "$a" = "-O3" ] && a="-O2"; args+=("$a"); done
exec /opt/rocm/bin/hipcc "${args[@]}"
