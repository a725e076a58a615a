#!/bin/bash
# time the shipped library and every variant named on the command line with tools/reg_probe.py
# usage: run_reg_variants.sh <edge> <f64|f32> <variant>...
edge=$1; fft=$2; shift 2
timeout -k 10 120 python tools/reg_probe.py $edge $fft || exit 1
for v in "$@"; do
  MISTITCH_LIB=$PWD/tools/variants/$v.so timeout -k 10 120 python tools/reg_probe.py $edge $fft || exit 1
done
