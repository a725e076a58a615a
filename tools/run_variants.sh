#!/bin/bash
# time every library variant under tools/variants with the fusion probe
for f in tools/variants/*.so; do
  echo "== $f"
  MISTITCH_LIB=$PWD/$f timeout -k 10 120 python tools/fuse_probe.py --reps 3 || exit 1
done
