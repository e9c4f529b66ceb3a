#!/bin/bash
# usage: build_variant.sh NAME [extra hipcc flags]   -> variants/NAME.so (developer A/B builds, TG_DEV_MIN)
set -e
name=$1; shift
cd "$(dirname "$0")/.."; mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wno-comment -Wno-int-to-pointer-cast -DTG_DEV_MIN=0 "$@" any4_amd/csrc/tinygemm_hip.hip -o variants/$name.so
echo built variants/$name.so
