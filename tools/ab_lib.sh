#!/bin/bash
# A/B two builds of the library in one gpurun call (same box): $1 = old .so, rest = command
set -e
PKG=facerecognition-multiarchitecture-pipeline_amd
cp $PKG/libfrmap_hip.so /tmp/new.so
for round in 1 2; do
  cp "$1" $PKG/libfrmap_hip.so; echo "--- old"; "${@:2}"
  cp /tmp/new.so $PKG/libfrmap_hip.so; echo "--- new"; "${@:2}"
done
