#!/bin/bash
# usage: tools/kisa.sh file.hip 'kernel-substring' [extra flags]  -> prints the kernel's ISA to stdout
f=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I/root/repo/include -S --cuda-device-only "$f" "$@" -o /tmp/kisa.s 2>/dev/null
awk -v pat="$pat" 'index($0,pat) && /^_Z[^ ]*:/ && !p {p=1} p{print} p && /^\.Lfunc_end/{exit}' /tmp/kisa.s
