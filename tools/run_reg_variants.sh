#!/bin/bash
# time every library variant under tools/variants with the registration probe; arguments are passed on to it
for f in tools/variants/*.so; do
  echo "== $f"
  MISTITCH_LIB=$PWD/$f timeout -k 10 200 python tools/reg_probe.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
done
