#!/bin/bash
# Probe build (-DKTF_TILE_PROBE) with extra flags: tools/build_probe_variant.sh <name> "<flags>" -> kaldi_tflite_amd/libktf_<name>.so
exec $(dirname $0)/build_variant.sh "$1" "-DKTF_TILE_PROBE $2"
