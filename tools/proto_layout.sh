#!/bin/bash
# Node-layout prototype: the shipped library against the -DRTG_PROTO_LINK build in its layouts.
# Usage: tools/proto_layout.sh c2|c4 > gpurun_out/proto_layout_<what>.txt   (take build_ab out of .gpurunignore first)
set -e
what=$1
timeout -k 10 200 python tools/proto_layout.py $what shipped
export RTG_LIBRTG_PATH=$PWD/build_ab/librtg_link.so
for mode in 1 2; do
  RTG_PROTO_LAYOUT=$mode timeout -k 10 200 python tools/proto_layout.py $what link
done
for top in 12 15 17 19; do
  RTG_PROTO_LAYOUT=3 RTG_PROTO_TOP=$top timeout -k 10 200 python tools/proto_layout.py $what link
done
