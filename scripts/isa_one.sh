#!/bin/bash
# usage: scripts/isa_one.sh '<explicit instantiation, e.g. template __global__ void tsff::k_form_factor_2d<1, false, 1, false>(tsff::KStatic, const double*, const double*, int, double, double, int, long, long, double*, double*);>' out.s [-D...]
# device assembly of ONE kernel instantiation in seconds instead of the two minutes of the whole library
inst=$1; out=$2; shift 2
tmp=$(mktemp /tmp/isa_one_XXXX.hip)
cat > $tmp <<EOT
#define TSFF_NO_API
#include "$(cd $(dirname $0)/.. && pwd)/tsadar_amd/csrc/tsff_kernels.hip"
$inst
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I $(dirname $0)/../include "$@" -o $out $tmp 2>&1 | grep -E "error" ; rm -f $tmp
