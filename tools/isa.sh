#!/bin/bash
# tools/isa.sh [kernel substring] [extra hipcc flags] — compile fused_kernels.hip to ISA (/tmp/fused.s) and print the static report
cd "$(dirname "$0")/../fastq-rs_amd/csrc"
K=${1:-k_scan_statsILj5ELj16}; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall "$@" -S --cuda-device-only -o /tmp/fused.s fused_kernels.hip 2>&1 | grep -v "hip-link"
python ../../tools/isa_report.py /tmp/fused.s "$K" 2>/dev/null | sed -n '2,15p'
